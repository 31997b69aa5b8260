#!/bin/bash
# What bounds the narrow fp32 up-sampling tiles?  Counter passes (one group per run, counters only with --kernel-trace) over
# tools/up_probe.py on the N = 32 level (upconv_halo_kernel<float,4,1,2>) and the N = 160 level (<4,5,2>), and over the plain 3x3
# kernel at 160 -> 160 (conv3x3_kernel<float,4,5,2>, 0.86 MFMA-busy) for comparison:  bash tools/gpu/narrow_pmc.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/narrow_pmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
declare -A PMCG
PMCG[vmem]="SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
PMCG[lds]="SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"
PMCG[l2]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum"
PMCG[mfma]="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
run() {  # tag, group, command...
  tag=$1; g=$2; shift; shift
  rocprofv3 --pmc ${PMCG[$g]} --kernel-trace --output-format csv -d $OUT/${tag}_$g -o p -- "$@" > /dev/null 2> $OUT/${tag}_$g.err
}
for g in vmem lds l2 mfma; do
  PROBE_SHAPE=64,128,64,16,32 run up32 $g python3 $R/tools/up_probe.py 5
  PROBE_SHAPE=64,32,328,40,160 run up160 $g python3 $R/tools/up_probe.py 5
  run c3_160 $g python3 $R/tools/conv3_probe.py fp32 5 64,64,64,160,160
done
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for tag in ("up32", "up160", "c3_160"):
    tot = collections.defaultdict(float); n = 0; dur = 0.0
    for g in ("vmem", "lds", "l2", "mfma"):
        f = glob.glob(os.path.join(out, "%s_%s" % (tag, g), "**", "*counter_collection.csv"), recursive=True)
        if not f:
            print(tag, g, "no counters:", open(os.path.join(out, "%s_%s.err" % (tag, g))).read()[-300:]); continue
        for r in csv.DictReader(open(f[0])):
            k = r["Kernel_Name"]
            if "upconv" in k or "conv3x3" in k:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
    print(tag, {k: ("%.4g" % v) for k, v in sorted(tot.items())})
    t = tot
    if t.get("SQ_INSTS_VMEM_RD"): print("   mean VMEM read latency %.0f cycles" % (t["SQ_INST_LEVEL_VMEM"] / t["SQ_INSTS_VMEM_RD"]))
    if t.get("SQ_INSTS_LDS"): print("   mean LDS latency %.0f cycles; bank-conflict cycles / LDS-active cycles %.3f" % (t["SQ_INST_LEVEL_LDS"] / t["SQ_INSTS_LDS"], t["SQ_LDS_BANK_CONFLICT"] / max(t["SQ_ACTIVE_INST_LDS"], 1)))
    if t.get("TCC_REQ_sum"): print("   L2 hit rate %.3f; requests %.4g; EA reads %.4g" % (t["TCC_HIT_sum"] / max(t["TCC_HIT_sum"] + t["TCC_MISS_sum"], 1), t["TCC_REQ_sum"], t.get("TCC_EA_RDREQ_sum", 0)))
    if t.get("SQ_WAVE_CYCLES"): print("   MFMA busy / wave cycles %.3f; wait-any / wave cycles %.3f" % (t["SQ_VALU_MFMA_BUSY_CYCLES"] / t["SQ_WAVE_CYCLES"], t["SQ_WAIT_INST_ANY"] / t["SQ_WAVE_CYCLES"]))
PY
rm -rf $OUT/*_vmem $OUT/*_lds $OUT/*_l2 $OUT/*_mfma
