#!/bin/bash
# per-dispatch durations of se_gate_kernel (and its neighbours) in one bf16 forward, in dispatch order
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
export CCVPE_EVAL_TWO_STREAMS=0 CCVPE_OVERLAP_DECODERS=0
cd /tmp
rm -rf /tmp/se_tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/se_tr -o t -- python3 $R/bench.py --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-kernel-events > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/se_tr/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last forward only: take the last 400 dispatches
rows = rows[-420:]
prev = None
for i, r in enumerate(rows):
    n = r['Kernel_Name']
    if 'se_gate' in n:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        p = rows[i - 1]
        pd = (int(p['End_Timestamp']) - int(p['Start_Timestamp'])) / 1e3
        gap = (int(r['Start_Timestamp']) - int(p['End_Timestamp'])) / 1e3
        print("se_gate %6.1f us  grid %s  | after %-40s %7.1f us, gap %5.1f us" % (d, r.get('Grid_Size_X', '?') + 'x' + r.get('Grid_Size_Y', '?'), p['Kernel_Name'][:40], pd, gap))
PY
