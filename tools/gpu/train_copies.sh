#!/bin/bash
# Which device-to-device copies does a training step make?  (single stream; prints the large ones with their neighbours)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
OUT=$R/gpurun_out/train_copies
mkdir -p $OUT
export TMPDIR=/tmp
export CCVPE_TRAIN_TWO_STREAMS=0 CCVPE_TRAIN_DEFER_WGRAD=0 CCVPE_TRAIN_DECODER_STREAMS=0
rocprofv3 --kernel-trace --output-format csv -d $OUT -o train -- python3 $R/bench.py --train --model vigor20 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events > $OUT/train.json 2> $OUT/train.err
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
# the last step = the tail of the trace after the last adam_kernel but one
adam = [i for i, n in enumerate(names) if "adam_kernel" in n]
lo = adam[-2] + 1 if len(adam) > 1 else 0
hi = adam[-1] + 1
print("launches in the last step:", hi - lo, " kernel time %.1f ms" % (sum(dur[lo:hi]) / 1e3), " span %.1f ms" % ((int(rows[hi-1]["End_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e6))
cp = [i for i in range(lo, hi) if "copyBuffer" in names[i]]
print("copyBuffer launches:", len(cp), " total %.1f us" % sum(dur[i] for i in cp))
for i in cp:
    if dur[i] > 40:
        print("%8.1f us  after [%s]  before [%s]" % (dur[i], names[i-1][:70], names[i+1][:70]))
small = [dur[i] for i in cp if dur[i] <= 40]
print("small copies:", len(small), "total %.1f us" % sum(small))
# gaps between consecutive kernels in the step
gaps = [(int(rows[i+1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3 for i in range(lo, hi-1)]
print("idle between kernels: %.1f ms total, %d gaps > 20 us" % (sum(g for g in gaps if g > 0) / 1e3, sum(1 for g in gaps if g > 20)))
import collections
agg = collections.Counter()
for i in range(lo, hi): agg[names[i].split("(")[0][:60]] += dur[i]
for n, t in agg.most_common(40): print("%9.1f us  %s" % (t, n))
PY
find $OUT -name "*kernel_trace.csv" -delete
