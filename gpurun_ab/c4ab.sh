cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; env $2 python bench.py --precision bf16 --model prior180_fov180 --batch 256 $3 --no-extra --no-cpu-baseline --no-kernel-events --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])'; }
run "graph default" "X=1" "--graph"
run "graph one stream" "CCVPE_EVAL_TWO_STREAMS=0 CCVPE_OVERLAP_DECODERS=0" "--graph"
run "graph enc two, dec one" "CCVPE_OVERLAP_DECODERS=0" "--graph"
run "eager default" "X=1" ""
run "eager one stream" "CCVPE_EVAL_TWO_STREAMS=0 CCVPE_OVERLAP_DECODERS=0" ""
run "graph default" "X=1" "--graph"
