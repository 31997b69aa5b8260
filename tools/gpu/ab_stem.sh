#!/bin/bash
# A/B of the fused stem + block-0 depthwise launch (CCVPE_FUSE_STEM=0 / 1) on the C1 forward, fp32 and bf16, B = 64.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() {  # label, precision, env...
  label=$1; prec=$2; shift; shift
  line=$(env "$@" python3 bench.py --precision $prec --no-extra --no-cpu-baseline --no-kernel-events --steps 20 --warmup 5 2>/dev/null | tail -1)
  echo "$label: $(echo $line | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms", d["value"], "pairs/s")')"
}
for rep in 1 2; do
run "bf16 two launches" bf16 CCVPE_FUSE_STEM=0
run "bf16 fused       " bf16 CCVPE_FUSE_STEM=1
run "fp32 two launches" fp32 CCVPE_FUSE_STEM=0
run "fp32 fused       " fp32 CCVPE_FUSE_STEM=1
done
