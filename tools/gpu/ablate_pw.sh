#!/bin/bash
# Ablation of the pointwise GEMM (csrc/conv_pw_impl.h) with a diagnostics build of conv_pw_f32.hip / conv_pw_bf16.hip:
#   bash tools/gpu/ablate_pw.sh build   (here, CPU: writes gpurun_ab/libccvpe_pw_abl.so)
#   bash tools/gpu/ablate_pw.sh [fp32|bf16] ["M,K,N" ...]   (GPU box)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/../.. && pwd)}
cd $R
if [ "$1" = build ]; then
  mkdir -p gpurun_ab
  cd ccvpe_amd/csrc
  for f in conv_pw_f32 conv_pw_bf16 conv_pw2_f32; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DCCVPE_ABLATE -c $f.hip -o /tmp/${f}_abl.o || exit 1
  done
  OBJS=$(ls *.o | grep -v "conv_pw_f32.o\|conv_pw_bf16.o\|conv_pw2_f32.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/conv_pw_f32_abl.o /tmp/conv_pw_bf16_abl.o /tmp/conv_pw2_f32_abl.o -o $R/gpurun_ab/libccvpe_pw_abl.so || exit 1
  echo built $R/gpurun_ab/libccvpe_pw_abl.so
  exit 0
fi
PREC=${1:-fp32}; shift
SHAPES=${@:-"65536,112,672 65536,672,112 16384,192,1152 65536,80,480"}
export CCVPE_LIB=$R/gpurun_ab/libccvpe_pw_abl.so
for a in ${ABLS:-0 1 2 4 8 6 14 15}; do
  echo "== ablate=$a (pw_gemm: 1 no global stores, 2 no stage loads after the first, 4 no MFMAs, 8 no swish; pw_ring (CCVPE_PW_RING=1): 1 no global stores, 2 no epilogue, 4 no MFMAs, 8 no DMA after the prologue, 16 no waits / ring barriers)"
  for sh in $SHAPES; do CCVPE_PW_ABLATE=$a CCVPE_PW2_ABLATE=$a python3 tools/pw_probe.py $PREC 20 $sh 2>&1 | grep -v amdgpu.ids; done
done
