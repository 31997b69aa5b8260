#!/bin/bash
# Ablation of the late-block MBConv front (csrc/mbconv_plane.hip) with diagnostics builds of that file only (compile-time masks:
# a run-time switch would change the code being measured):
#   bash tools/gpu/ablate_mbplane.sh build   (here, CPU: writes gpurun_ab/libccvpe_mbp_<mask>.so)
#   bash tools/gpu/ablate_mbplane.sh         (GPU box: tools/mbp_probe.py per mask)
# masks: 1 no y stores, 2 no output swish, 4 no depthwise FMAs, 8 no window reads, 16 no expand swish, 32 no x loads, 64 no expand
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/../.. && pwd)}
cd $R
MASKS=${MASKS:-0 1 2 4 8 16 64 18 30 94 95}
if [ "$1" = build ]; then
  mkdir -p gpurun_ab
  cd ccvpe_amd/csrc
  OBJS=$(ls *.o | grep -v mbconv_plane.o)
  for m in $MASKS; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMBP_ABL=$m -c mbconv_plane.hip -o /tmp/mbp_abl_$m.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/mbp_abl_$m.o -o $R/gpurun_ab/libccvpe_mbp_$m.so || exit 1
    echo built $R/gpurun_ab/libccvpe_mbp_$m.so
  done
  exit 0
fi
for m in $MASKS; do
  echo "== mask=$m"
  CCVPE_LIB=$R/gpurun_ab/libccvpe_mbp_$m.so MBP_FEW=1 MBP_BAND_ONLY=${BAND_ONLY:-} python3 tools/mbp_probe.py 20 ${DT:-bf16} ${KB:-72} 2>&1 | grep -v amdgpu.ids
done
