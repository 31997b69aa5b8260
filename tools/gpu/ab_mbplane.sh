#!/bin/bash
# A/B of the late-block MBConv front inside the forwards, one GPU call: CCVPE_MBPLANE = 0 (round-5 chain), 3 (slice kernel),
# 7 (band-owner kernel in bf16; default).  bash tools/gpu/ab_mbplane.sh [bf16|fp32]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
P=${1:-bf16}
run() {  # label, env..., -- args
  label=$1; shift
  line=$(env "$@" python3 bench.py --precision $P --no-extra --no-cpu-baseline --no-kernel-events --steps 20 --warmup 5 $EXTRA 2>/dev/null | tail -1)
  echo "$label: $(echo $line | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms", d["value"], "pairs/s")')"
}
for rep in 1 2; do
EXTRA="" run "C1 $P chain" CCVPE_MBPLANE=0
EXTRA="" run "C1 $P slice" CCVPE_MBPLANE=3
EXTRA="" run "C1 $P band " CCVPE_MBPLANE=7
done
if [ $P = bf16 ]; then
EXTRA="--model vigor20 --batch 32" run "C2 chain" CCVPE_MBPLANE=0
EXTRA="--model vigor20 --batch 32" run "C2 band " CCVPE_MBPLANE=7
EXTRA="--model prior180_fov180 --batch 256 --graph" run "C4 chain" CCVPE_MBPLANE=0
EXTRA="--model prior180_fov180 --batch 256 --graph" run "C4 band " CCVPE_MBPLANE=7
fi
