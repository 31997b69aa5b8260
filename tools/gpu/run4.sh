#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
for pw in 1 0; do
  echo "== pw=$pw"; CCVPE_PW_GEMM=$pw python3 tools/pw_probe.py fp32 20 2>/dev/null
  CCVPE_PW_GEMM=$pw python3 tools/pw_probe.py bf16 20 2>/dev/null
done
cd /tmp
for pw in 1 0; do
export CCVPE_PW_GEMM=$pw
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_pw$pw -o a -- python3 $R/tools/pw_probe.py fp32 3 65536,112,672 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM GRBM_GUI_ACTIVE SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/pmc_pw$pw -o b -- python3 $R/tools/pw_probe.py fp32 3 65536,112,672 > /dev/null 2>&1
done
ls $OUT/pmc_pw1 $OUT/pmc_pw0
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ.get("GRAFT_REPO_ROOT", ".")
for pw in (1,0):
    for f in sorted(glob.glob("%s/gpurun_out/pmc_pw%d/*counter_collection.csv"%(R,pw))):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"]
            if "gemm" not in k: continue
            agg[k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k,d in agg.items():
            print("pw=%d"%pw, k, {c: round(sum(v)/len(v)) for c,v in d.items()})
PY
