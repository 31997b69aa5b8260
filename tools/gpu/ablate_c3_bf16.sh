#!/bin/bash
# Ablation of the bf16 3x3 convolution's stage loop with the diagnostics build (make EXTRA=-DCCVPE_ABLATE -> gpurun_ab/libccvpe_ablate.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export CCVPE_LIB=$R/gpurun_ab/libccvpe_ablate.so
for shape in 64,16,16,640,640 64,64,64,160,160; do
for a in 0 2 8 16 32 64 128 34 42 106 122; do
  echo "== shape=$shape ablate=$a (2 = no loads in the loop, 8 = no fragment reads, 16 = no barriers, 32 = no W DMA, 64 = no halo traffic, 128 = no DMA wait)"
  CCVPE_C3_ABLATE=$a python3 tools/conv3_probe.py bf16 20 $shape 2>&1 | grep -v amdgpu.ids
done
done
