#!/bin/bash
# LDS bank-conflict pass (PMC, counters only with --kernel-trace) over the training step and the two forwards.
#   usage (through gpurun): bash tools/gpu/lds_pass.sh <tag> <commit> ["train f32 bf16"]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/lds_${1:-r03}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
F32="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-kernel-events"
BF1="python3 $R/bench.py --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-kernel-events"
TRN="python3 $R/bench.py --train --model vigor20 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events"
for w in ${3:-train f32 bf16}; do
  case $w in f32) CMD="$F32";; train) CMD="$TRN";; bf16) CMD="$BF1";; esac
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_$w -o p -- $CMD > $OUT/$w.log 2>&1
  python3 $R/tools/lds_conflicts.py $OUT/pmc_$w/p_counter_collection.csv $OUT/pmc_$w/p_kernel_trace.csv $OUT/lds_conflicts_$w.json ${2:-unknown} | head -30
done
rm -rf $OUT/pmc_*
