#!/bin/bash
# A/B of library builds on the fp32 forward and the training step: bash tools/gpu/ab_f32.sh <tagA> <tagB> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
for v in "$@"; do
  for cfg in "--precision fp32 --steps 15 --warmup 4" "--train --model vigor20 --steps 4 --warmup 2"; do
    echo -n "$v [$cfg] "
    CCVPE_LIB=$R/tools/ab/libccvpe_hip_$v.so python3 bench.py $cfg --no-cpu-baseline --no-extra --no-kernel-events 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
  done
done
done
