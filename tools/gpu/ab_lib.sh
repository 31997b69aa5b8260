#!/bin/bash
# A/B of two builds of the library behind the same host code: gpurun_ab/libccvpe_prev.so (built from another commit in a worktree)
# against the in-tree build.  usage: bash tools/gpu/ab_lib.sh [train]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() {  # label, extra args, env...
  label=$1; extra=$2; shift; shift
  line=$(env "$@" python3 bench.py $extra --no-extra --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1)
  echo "$label: $(echo $line | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms", d["value"], "pairs/s")')"
}
PREV="CCVPE_LIB=$R/gpurun_ab/libccvpe_prev.so CCVPE_LIB_ALLOW_MISSING=1"
for rep in 1 2; do
run "fp32 forward prev" "--steps 20 --warmup 5" $PREV
run "fp32 forward new " "--steps 20 --warmup 5" X=1
run "bf16 forward prev" "--precision bf16 --steps 20 --warmup 5" $PREV
run "bf16 forward new " "--precision bf16 --steps 20 --warmup 5" X=1
done
if [ "$1" = train ]; then
for rep in 1 2; do
run "train prev" "--train --model vigor20 --steps 5 --warmup 3" $PREV
run "train new " "--train --model vigor20 --steps 5 --warmup 3" X=1
done
fi
