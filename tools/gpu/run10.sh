#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_wg -o a -- python3 $R/tools/wgrad_probe.py 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INST_LEVEL_LDS --kernel-trace --output-format csv -d $OUT/pmc_wg -o b -- python3 $R/tools/wgrad_probe.py 2 > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/pmc_wg -o c -- python3 $R/tools/wgrad_probe.py 2 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ.get("GRAFT_REPO_ROOT", ".")
for f in sorted(glob.glob("%s/gpurun_out/pmc_wg/*counter_collection.csv"%R)):
    agg=collections.OrderedDict()
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"]
        if "conv_wgrad" not in k: continue
        key=(k[:52], row["Grid_Size"])
        agg.setdefault(key, collections.defaultdict(list))[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,d in agg.items():
        print(k, {c: round(sum(v)/len(v)) for c,v in d.items()})
PY
