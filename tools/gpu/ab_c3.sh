#!/bin/bash
# A/B of two library builds on the 3x3 probe shapes: bash tools/gpu/ab_c3.sh <tagA> <tagB>   (tools/ab/libccvpe_hip_<tag>.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for prec in bf16; do
for shape in 64,16,16,640,640 64,32,32,320,320 64,64,64,160,160 64,16,16,1344,640; do
for v in "$@"; do
  echo -n "$v  "
  CCVPE_LIB=$R/tools/ab/libccvpe_hip_$v.so python3 tools/conv3_probe.py $prec 20 $shape 2>&1 | grep -v amdgpu.ids
done
done
done
