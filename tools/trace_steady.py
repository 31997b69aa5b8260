"""Per-kernel time of the STEADY-STATE steps from a rocprofv3 kernel trace CSV (skips the warm-up / packing launches by
keeping the last `frac` of the trace):  python tools/trace_steady.py <kernel_trace.csv> <n_steps_total> <steps_to_keep>"""
import csv
import re
import sys
from collections import defaultdict

path, total_steps, keep = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steady-state window: the trace's last `keep / total_steps` share by launch COUNT of the dominant repeating kernel
names = [r["Kernel_Name"] for r in rows]
n = len(rows)
# find the boundary as the start of the `keep`-th last occurrence of the stem kernel on the aerial image
stem = [i for i, r in enumerate(rows) if "stem_conv_kernel" in r["Kernel_Name"]]
per_step = len(stem) // total_steps if stem else 0
start = stem[len(stem) - keep * per_step] if per_step else int(n * (1 - keep / total_steps))
agg = defaultdict(lambda: [0, 0])
t0 = int(rows[start]["Start_Timestamp"])
t1 = max(int(r["End_Timestamp"]) for r in rows[start:])
for r in rows[start:]:
    nm = re.sub(r"\(.*", "", r["Kernel_Name"])
    nm = re.sub(r"^void ", "", nm)
    d = agg[nm]
    d[0] += 1
    d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(d[1] for d in agg.values())
print("steady window: %d launches, wall %.3f ms/step, kernel-time sum %.3f ms/step" % (n - start, (t1 - t0) / keep / 1e6, tot / keep / 1e6))
for nm, d in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[4]) if len(sys.argv) > 4 else 45]:
    print("%-78s %5.1f/st %8.3f ms/st %5.1f%% avg %8.1f us" % (nm[:78], d[0] / keep, d[1] / keep / 1e6, 100.0 * d[1] / tot, d[1] / d[0] / 1e3))
