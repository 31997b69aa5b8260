#!/bin/bash
# A/B of the matching kernel's lanes-per-pixel rule (csrc/matching.hip: match_lpp_div) on the C2 / C1 bf16 / C1 fp32 forwards.
#   bash tools/gpu/ab_match_lpp.sh build   (here, CPU: diagnostics build of csrc/matching.hip -> gpurun_ab/libccvpe_match_abl.so)
#   bash tools/gpu/ab_match_lpp.sh         (GPU box)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/../.. && pwd)}
cd $R
if [ "$1" = build ]; then
  mkdir -p gpurun_ab
  cd ccvpe_amd/csrc
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -DCCVPE_ABLATE -c matching.hip -o /tmp/matching_abl.o || exit 1
  OBJS=$(ls *.o | grep -v "^matching.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/matching_abl.o -o $R/gpurun_ab/libccvpe_match_abl.so || exit 1
  echo built $R/gpurun_ab/libccvpe_match_abl.so
  exit 0
fi
export CCVPE_LIB=$R/gpurun_ab/libccvpe_match_abl.so
run() { label=$1; extra=$2; shift; shift; line=$(env "$@" python3 bench.py $extra --no-extra --no-cpu-baseline --no-kernel-events --steps 20 --warmup 5 2>/dev/null | tail -1); echo "$label: $(echo $line | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms", d["value"], "pairs/s")')"; }
for div in 48 24 12; do
run "C2 bf16 lpp_div $div" "--precision bf16 --model vigor20 --batch 32" CCVPE_MATCH_LPP_DIV=$div
run "C1 bf16 lpp_div $div" "--precision bf16" CCVPE_MATCH_LPP_DIV=$div
run "C1 fp32 lpp_div $div" "" CCVPE_MATCH_LPP_DIV=$div
done
