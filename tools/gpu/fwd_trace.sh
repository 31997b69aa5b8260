#!/bin/bash
# Per-dispatch durations of ONE forward, in dispatch order (single stream): bash tools/gpu/fwd_trace.sh [fp32|bf16] [tag] -> gpurun_out/fwd_trace_<prec><tag>.csv
# EXTRA="--model vigor20 --batch 32" selects another configuration (tag it: _c2)
R=${GRAFT_REPO_ROOT:-$(pwd)}
P=${1:-bf16}
TAG=${2:-}
export TMPDIR=/tmp
export CCVPE_EVAL_TWO_STREAMS=0 CCVPE_OVERLAP_DECODERS=0
cd /tmp
rm -rf /tmp/fw_tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/fw_tr -o t -- python3 $R/bench.py --precision $P --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-kernel-events $EXTRA > /dev/null 2>&1
python3 - "$R/gpurun_out/fwd_trace_$P$TAG.csv" <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/fw_tr/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last forward: from the last pair of stem launches (stem_dw_kernel since round 5, stem_conv_kernel with CCVPE_FUSE_STEM=0) onwards
idx = [i for i, r in enumerate(rows) if 'stem_conv' in r['Kernel_Name'] or 'stem_dw' in r['Kernel_Name']]
start = idx[-2] if len(idx) >= 2 else 0
with open(sys.argv[1], 'w') as out:
    out.write("n,us,gap_us,grid,wg,lds,vgpr,kernel\n")
    prev_end = None
    for n, r in enumerate(rows[start:]):
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
        prev_end = e
        out.write("%d,%.1f,%.1f,%sx%s,%s,%s,%s,%s\n" % (n, (e - s) / 1e3, gap, r.get('Grid_Size_X', ''), r.get('Grid_Size_Y', ''),
                  r.get('Workgroup_Size_X', ''), r.get('LDS_Block_Size', ''), r.get('VGPR_Count', ''), r['Kernel_Name'].replace(',', ';')[:90]))
PY
