#!/bin/bash
# A/B of the matching kernel's lanes-per-pixel rule (csrc/matching.hip: match_lpp_div) on the C2 / C1 bf16 / C1 fp32 forwards.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
# needs the diagnostics build of csrc/matching.hip (-DCCVPE_ABLATE -fno-slp-vectorize) linked as gpurun_ab/libccvpe_match_abl.so
export CCVPE_LIB=$R/gpurun_ab/libccvpe_match_abl.so
run() { label=$1; extra=$2; shift; shift; line=$(env "$@" python3 bench.py $extra --no-extra --no-cpu-baseline --no-kernel-events --steps 20 --warmup 5 2>/dev/null | tail -1); echo "$label: $(echo $line | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms", d["value"], "pairs/s")')"; }
for div in 48 24 12; do
run "C2 bf16 lpp_div $div" "--precision bf16 --model vigor20 --batch 32" CCVPE_MATCH_LPP_DIV=$div
run "C1 bf16 lpp_div $div" "--precision bf16" CCVPE_MATCH_LPP_DIV=$div
run "C1 fp32 lpp_div $div" "" CCVPE_MATCH_LPP_DIV=$div
done
