#!/bin/bash
# rocprofv3 kernel stats of the VIGOR training bench (3 timed steps + 2 warm-up) -> gpurun_out/prof_train/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_train
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o train -- python3 $R/bench.py --train --model vigor20 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events > $OUT/train.json 2> $OUT/train.err
find $OUT -name '*kernel_trace.csv' -size +12M -delete
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/train_kernel_stats.csv')))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:45]:
    print("%-80s %6s %9.3f ms/step %8.1f us"%(r['Name'][:80],r['Calls'],float(r["TotalDurationNs"])/5e6,float(r['AverageNs'])/1e3))
PY
