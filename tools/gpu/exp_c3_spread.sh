#!/bin/bash
# Experiments on the 3x3 stage loop (diagnostics build): 256 = W requests spread over the stage's matrix instructions,
# 512 = second workgroup of a CU starts half a stage late
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export CCVPE_LIB=$R/gpurun_ab/libccvpe_ablate.so
for prec in bf16 fp32; do
for shape in 64,16,16,640,640 64,32,32,320,320 64,64,64,160,160; do
for a in 0 256 512 768; do
  echo -n "ablate=$a  "
  CCVPE_C3_ABLATE=$a python3 tools/conv3_probe.py $prec 20 $shape 2>&1 | grep -v amdgpu.ids
done
done
done
