#!/bin/bash
# Ablation of the weight-gradient stage loop (diagnostics build): 1 = no global loads, 2 = no LDS stores, 4 = no fragment reads
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export CCVPE_LIB=$R/gpurun_ab/libccvpe_ablate.so
export PROBE_SHAPE=64,16,640,1344,3
for a in 0 1 2 3 4 7; do
  echo "== ablate=$a"; CCVPE_WG_ABLATE=$a python3 tools/wgrad_probe.py 10 2>&1 | grep -v amdgpu.ids
done
