#!/bin/bash
# Round evidence: the default bench line, rocprofv3 kernel stats of the measured workloads, PMC traffic and MFMA-busy passes.
#   usage (on the GPU box, through gpurun): bash tools/gpu/evidence.sh <round tag, e.g. r03> <commit> [stats|pmc|all]
# Everything lands in gpurun_out/ev_<round>/; the summaries to be judged are then copied into profiles/<round>/ (tracked).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
OUT=$R/gpurun_out/ev_$TAG
COMMIT=${2:-unknown}
WHAT=${3:-all}
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_${TAG}_n1.json 2> $OUT/bench_${TAG}_n1.err
cd /tmp
F32="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-kernel-events"
BF1="python3 $R/bench.py --precision bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-kernel-events"
BF2="python3 $R/bench.py --precision bf16 --model vigor20 --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-kernel-events"
TRN="python3 $R/bench.py --train --model vigor20 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events"
if [ "$WHAT" != pmc ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/f32 -o f32 -- $F32 > $OUT/f32.json 2> $OUT/f32.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bf16_c1 -o bf16_c1 -- $BF1 > $OUT/bf16_c1.json 2> $OUT/bf16_c1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bf16_c2 -o bf16_c2 -- $BF2 > $OUT/bf16_c2.json 2> $OUT/bf16_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -o train -- $TRN > $OUT/train.json 2> $OUT/train.err
fi
[ "$WHAT" = stats ] && { find $OUT -name '*kernel_trace.csv' -size +8M -delete; ls -la $OUT; exit 0; }
# PMC passes (counters only with --kernel-trace; FETCH and WRITE in separate passes)
for w in f32 train bf16 bf16_c2; do
  case $w in f32) CMD="$F32";; train) CMD="$TRN";; bf16) CMD="$BF1";; bf16_c2) CMD="$BF2";; esac
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$w -o p -- $CMD > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$w -o p -- $CMD > /dev/null 2>&1
  [ $w != bf16_c2 ] && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma_$w -o p -- $CMD > /dev/null 2>&1
  python3 $R/tools/pmc_traffic.py $OUT/pmc_fetch_$w/p_counter_collection.csv $OUT/pmc_write_$w/p_counter_collection.csv $OUT/pmc_traffic_$w.json $COMMIT "$w" || true
  [ $w != bf16_c2 ] && { python3 $R/tools/mfma_busy.py $OUT/pmc_mfma_$w/p_counter_collection.csv $OUT/pmc_mfma_$w/p_kernel_trace.csv $OUT/mfma_busy_$w.json $COMMIT || true; }
done
# one table for bench.py (profiles/pmc_traffic.json): {workload: {kernel family: HBM bytes per launch, "#meta": {..., gb_per_step}}}
python3 - $OUT <<'PY'
import json, os, sys
out = {"#layout": "workload -> kernel family -> HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3 --pmc passes); "
                  "#meta.gb_per_step = all families x their launches per executed step"}
for w in ("f32", "train", "bf16", "bf16_c2"):
    p = os.path.join(sys.argv[1], "pmc_traffic_%s.json" % w)
    if os.path.isfile(p):
        out[w] = json.load(open(p))
json.dump(out, open(os.path.join(sys.argv[1], "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
PY
# per-dispatch trace of ONE forward on one stream (no two kernels share the chip): the per-family picture DESIGN reads
for prec in bf16 fp32; do
  bash $R/tools/gpu/fwd_trace.sh $prec > /dev/null 2>&1 && cp $R/gpurun_out/fwd_trace_$prec.csv $OUT/fwd_trace_$prec.csv
done
EXTRA="--model vigor20 --batch 32" bash $R/tools/gpu/fwd_trace.sh bf16 _c2 > /dev/null 2>&1 && cp $R/gpurun_out/fwd_trace_bf16_c2.csv $OUT/fwd_trace_bf16_c2.csv
# keep the merge small: drop the per-dispatch counter / trace CSVs of the PMC passes
rm -rf $OUT/pmc_fetch_* $OUT/pmc_write_* $OUT/pmc_mfma_*
find $OUT -name '*kernel_trace.csv' -size +8M -delete
ls -la $OUT
