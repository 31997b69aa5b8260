#!/bin/bash
# A/B of the bf16 forwards in one GPU call: narrow-level kernels off / on, fused matching off / on (C1 B = 64, C2 B = 32, C4 B = 256 graph).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() {  # label, env..., -- args
  label=$1; shift
  line=$(env "$@" python3 bench.py --precision bf16 --no-extra --no-cpu-baseline --no-kernel-events --steps 20 --warmup 5 $EXTRA 2>/dev/null | tail -1)
  echo "$label: $(echo $line | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms", d["value"], "pairs/s")')"
}
for rep in 1 2; do
EXTRA="" run "C1 tiled            " CCVPE_NARROW=0
EXTRA="" run "C1 narrow, no fusion" CCVPE_NARROW=1 CCVPE_FUSE_MATCH=0
EXTRA="" run "C1 narrow + fusion  " CCVPE_NARROW=1
done
EXTRA="--model vigor20 --batch 32" run "C2 narrow" CCVPE_NARROW=1
EXTRA="--model vigor20 --batch 32" run "C2 tiled " CCVPE_NARROW=0
EXTRA="--model prior180_fov180 --batch 256 --graph" run "C4 narrow" CCVPE_NARROW=1
EXTRA="--model prior180_fov180 --batch 256 --graph" run "C4 tiled " CCVPE_NARROW=0
