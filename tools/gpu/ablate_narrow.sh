#!/bin/bash
# Ablation of the narrow-level kernels (csrc/narrow_impl.h) with a diagnostics build of narrow_bf16.hip only:
#   bash tools/gpu/ablate_narrow.sh build   (here, CPU: writes gpurun_ab/libccvpe_narrow_abl.so)
#   bash tools/gpu/ablate_narrow.sh         (GPU box: tools/narrow_probe.py per ablation mask)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/../.. && pwd)}
cd $R
if [ "$1" = build ]; then
  mkdir -p gpurun_ab
  cd ccvpe_amd/csrc
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DCCVPE_ABLATE -c narrow_bf16.hip -o /tmp/narrow_abl.o || exit 1
  OBJS=$(ls *.o | grep -v narrow_bf16.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/narrow_abl.o -o $R/gpurun_ab/libccvpe_narrow_abl.so || exit 1
  echo built $R/gpurun_ab/libccvpe_narrow_abl.so
  exit 0
fi
export CCVPE_LIB=$R/gpurun_ab/libccvpe_narrow_abl.so
for a in 0 1 2 4 8 16 3 5 12 7 15; do
  echo "== ablate=$a (1 no halo requests, 2 no stores, 4 no MFMA, 8 no fragment reads, 16 no barrier/wait)"
  CCVPE_NARROW_ABLATE=$a NARROW_PROBE_FEW=1 python3 tools/narrow_probe.py 20 2>&1 | grep -v amdgpu.ids
done
