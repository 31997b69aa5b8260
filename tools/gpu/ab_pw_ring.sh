#!/bin/bash
# A/B of the fp32 pointwise ring kernel (CCVPE_PW_RING=0 / 1) on the C1 fp32 forward B = 64 and the CVM_VIGOR training step.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() {  # label, extra args, env...
  label=$1; extra=$2; shift; shift
  line=$(env "$@" python3 bench.py $extra --no-extra --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -1)
  echo "$label: $(echo $line | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms", d["value"], "pairs/s")')"
}
for rep in 1 2; do
run "fp32 forward pw_gemm" "--steps 20 --warmup 5" CCVPE_PW_RING=0
run "fp32 forward pw_ring" "--steps 20 --warmup 5" CCVPE_PW_RING=1
done
if [ "$1" = train ]; then
run "train pw_gemm" "--train --model vigor20 --steps 5 --warmup 3" CCVPE_PW_RING=0
run "train pw_ring" "--train --model vigor20 --steps 5 --warmup 3" CCVPE_PW_RING=1
fi
if [ "$1" = bf16 ] || [ "$2" = bf16 ]; then
for rep in 1 2; do
run "bf16 forward pw_gemm" "--precision bf16 --steps 20 --warmup 5" CCVPE_PW_RING=0
run "bf16 forward pw_ring" "--precision bf16 --steps 20 --warmup 5" CCVPE_PW_RING=1
done
fi
