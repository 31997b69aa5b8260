cd $GRAFT_REPO_ROOT
run() { echo -n "$1 [$2]: "; env $2 python bench.py --precision bf16 $3 --no-extra --no-cpu-baseline --no-kernel-events --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])'; }
for rep in 1 2; do
for e in "X=1" "CCVPE_OVERLAP_DECODERS=0" "CCVPE_EVAL_TWO_STREAMS=0" "CCVPE_EVAL_TWO_STREAMS=0 CCVPE_OVERLAP_DECODERS=0"; do
run "C1 B64" "$e" ""
run "C2 B32" "$e" "--model vigor20 --batch 32"
done
done
run "C1 fp32" "X=1" "--precision fp32"
run "C1 fp32" "CCVPE_EVAL_TWO_STREAMS=0" "--precision fp32"
