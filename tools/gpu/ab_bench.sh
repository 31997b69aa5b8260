#!/bin/bash
# A/B of library builds on whole forwards (same box): bash tools/gpu/ab_bench.sh <tagA> <tagB> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
for v in "$@"; do
  for cfg in "--model vigor20 --precision bf16 --batch 32" "--precision bf16" "--precision fp32"; do
    echo -n "$v [$cfg] "
    CCVPE_LIB=$R/tools/ab/libccvpe_hip_$v.so python3 bench.py $cfg --steps 15 --warmup 4 --no-cpu-baseline --no-extra --no-kernel-events 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
  done
done
done
