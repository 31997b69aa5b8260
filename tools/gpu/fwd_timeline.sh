#!/bin/bash
# Kernel timeline of ONE forward in the DEFAULT schedule (two streams): bash tools/gpu/fwd_timeline.sh [fp32|bf16] [tag]
# -> gpurun_out/fwd_timeline_<prec><tag>.csv (start / end relative to the forward's first launch, in us) + a phase summary on stdout
R=${GRAFT_REPO_ROOT:-$(pwd)}
P=${1:-bf16}
TAG=${2:-}
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/fw_tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/fw_tl -o t -- python3 $R/bench.py --precision $P --steps 3 --warmup 2 --no-cpu-baseline --no-extra --no-kernel-events $EXTRA > /dev/null 2>&1
python3 - "$R/gpurun_out/fwd_timeline_$P$TAG.csv" <<'PY'
import csv, glob, sys, re
f = glob.glob('/tmp/fw_tl/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'stem_conv' in r['Kernel_Name'] or 'stem_dw' in r['Kernel_Name']]
start = idx[-2] if len(idx) >= 2 else 0
rows = rows[start:]
t0 = int(rows[0]['Start_Timestamp'])
def short(k):
    m = re.search(r'(stem_dw|se_gate|igemm_kernel|mbconv_front|mbconv_band|mbconv_plane|pw_gemm|pw2_kernel|dwconv_plane|splitk_finish|gdesc|match_kernel|upconv_dma|upconv_halo|conv3x3_kernel|c3n_kernel|up2_kernel|tail512|softmax_apply|cast)', k)
    return m.group(1) if m else k[:24]
ev = []
with open(sys.argv[1], 'w') as out:
    out.write("start_us,end_us,dur_us,queue,kernel\n")
    for r in rows:
        s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
        out.write("%.1f,%.1f,%.1f,%s,%s\n" % (s, e, e - s, r.get('Queue_Id', ''), short(r['Kernel_Name'])))
        ev.append((s, e, r.get('Queue_Id', ''), short(r['Kernel_Name'])))
end = max(e for s, e, q, k in ev)
# time with 0 / 1 / 2+ kernels in flight
pts = sorted([(s, 1) for s, e, q, k in ev] + [(e, -1) for s, e, q, k in ev])
busy = {0: 0.0, 1: 0.0, 2: 0.0}
cur, last = 0, 0.0
for t, d in pts:
    busy[min(cur, 2)] += t - last
    last = t
    cur += d
first_dec = min(s for s, e, q, k in ev if k in ('match_kernel', 'upconv_dma', 'upconv_halo'))
print("forward %.1f us; encoders until %.1f us; time with 0 / 1 / >=2 kernels in flight: %.0f / %.0f / %.0f us" % (end, first_dec, busy[0], busy[1], busy[2]))
PY
