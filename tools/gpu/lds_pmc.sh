#!/bin/bash
# LDS bank conflicts per kernel family of one forward: bash tools/gpu/lds_pmc.sh [fp32|bf16]   (one --pmc pass, counters only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
PREC=${1:-fp32}
OUT=$R/gpurun_out/lds_pmc_$PREC
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p -o p -- python3 $R/bench.py --precision $PREC --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-kernel-events > /dev/null 2> $OUT/err.txt
python3 - $OUT <<'PY'
import csv, glob, os, re, sys, collections
f = glob.glob(os.path.join(sys.argv[1], "p", "**", "*counter_collection.csv"), recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(.*", "", r["Kernel_Name"])[:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
rows = []
for k, t in agg.items():
    act, conf, gui = t.get("SQ_ACTIVE_INST_LDS", 0), t.get("SQ_LDS_BANK_CONFLICT", 0), t.get("GRBM_GUI_ACTIVE", 0)
    if gui <= 0: continue
    rows.append((gui, k, conf / max(act, 1), (act + conf) / 256.0 / (gui / 8.0)))
print("%-72s %10s %9s %9s" % ("kernel", "cycles/8", "conf/act", "LDS busy"))
for gui, k, ratio, busy in sorted(rows, reverse=True)[:28]:
    print("%-72s %10.3g %9.2f %9.2f" % (k, gui / 8, ratio, busy))
PY
rm -rf $OUT/p
