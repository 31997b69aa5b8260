#!/bin/bash
# A/B of the matrix-core form of the matching kernel inside the 20 / 21-hypothesis forwards (C2 B = 32, C4 B = 256 graph), one GPU call.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() {  # label, env..., -- args
  label=$1; shift
  line=$(env "$@" python3 bench.py --precision bf16 --no-extra --no-cpu-baseline --no-kernel-events --steps 20 --warmup 5 $EXTRA 2>/dev/null | tail -1)
  echo "$label: $(echo $line | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms", d["value"], "pairs/s")')"
}
for rep in 1 2; do
EXTRA="--model vigor20 --batch 32" run "C2 valu" CCVPE_MATCH_MFMA=0
EXTRA="--model vigor20 --batch 32" run "C2 mfma" CCVPE_MATCH_MFMA=1
done
EXTRA="--model prior180_fov180 --batch 256 --graph" run "C4 valu" CCVPE_MATCH_MFMA=0
EXTRA="--model prior180_fov180 --batch 256 --graph" run "C4 mfma" CCVPE_MATCH_MFMA=1
EXTRA="--model vigor20 --batch 64 --precision fp32" run "vigor20 fp32 B64 valu" CCVPE_MATCH_MFMA=0
EXTRA="--model vigor20 --batch 64 --precision fp32" run "vigor20 fp32 B64 mfma" CCVPE_MATCH_MFMA=1
