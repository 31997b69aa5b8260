#!/bin/bash
# Ablation of the 3x3 convolution's stage loop with the diagnostics build (-DCCVPE_ABLATE -> gpurun_ab/libccvpe_ablate.so):
# which part of a stage keeps the MFMA pipe from its 32-cycle cadence.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export CCVPE_LIB=$R/gpurun_ab/libccvpe_ablate.so
for a in 0 256 32; do
  echo "== ablate=$a (8 = no fragment reads, 16 = no barriers, 32 = no W DMA, 64 = no halo traffic, 128 = no DMA wait)"
  CCVPE_C3_ABLATE=$a python3 tools/conv3_probe.py fp32 20 64,16,16,1344,640 2>&1 | grep -v amdgpu.ids
done
