#!/bin/bash
# Ablation of the fused stem + block-0 depthwise kernel (csrc/stem_dw.hip) with a diagnostics build of that file only:
#   bash tools/gpu/ablate_stem_dw.sh build   (here, CPU: writes gpurun_ab/libccvpe_stem_abl.so)
#   bash tools/gpu/ablate_stem_dw.sh         (GPU box: tools/stem_probe.py per ablation mask)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/../.. && pwd)}
cd $R
if [ "$1" = build ]; then
  mkdir -p gpurun_ab
  cd ccvpe_amd/csrc
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DCCVPE_ABLATE -c stem_dw.hip -o /tmp/stem_abl.o || exit 1
  OBJS=$(ls *.o | grep -v stem_dw.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/stem_abl.o -o $R/gpurun_ab/libccvpe_stem_abl.so || exit 1
  echo built $R/gpurun_ab/libccvpe_stem_abl.so
  exit 0
fi
export CCVPE_LIB=$R/gpurun_ab/libccvpe_stem_abl.so
for a in ${ABLS:-0 1 2 4 8 16 32 6 20 24 30 62 63}; do
  echo "== ablate=$a (1 no patch loads after the first tile, 2 no stem MFMAs, 4 no stem swish, 8 no depthwise reads/FMAs, 16 no output swish, 32 no stores)"
  CCVPE_STEM_DW_ABLATE=$a STEM_PROBE_FEW=1 python3 tools/stem_probe.py 20 2>&1 | grep -v amdgpu.ids
done
