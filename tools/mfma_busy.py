"""MFMA-pipe utilisation per kernel family from ONE rocprofv3 --pmc pass (CSV):
    --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE  --kernel-trace
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
(SQ_VALU_MFMA_BUSY_CYCLES sums cycles over all SIMDs = 32 x the fp32 16x16x4 MFMA count; GRBM_GUI_ACTIVE sums the busy
cycles of the 8 XCDs — checked against SQ_INSTS_MFMA x 32 and against kernel duration x clock), effective clock =
GRBM_GUI_ACTIVE / 8 / kernel duration.   python tools/mfma_busy.py <counter_collection.csv> <kernel_trace.csv> <out.json> [commit]"""
import csv
import json
import re
import sys
from collections import defaultdict

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_traffic import family     # noqa: E402


def main():
    cc, kt, out = sys.argv[1], sys.argv[2], sys.argv[3]
    commit = sys.argv[4] if len(sys.argv) > 4 else "unknown"
    dur = {}
    for r in csv.DictReader(open(kt)):
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    per = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    seen = set()
    for r in csv.DictReader(open(cc)):
        fam = family(r["Kernel_Name"])
        if fam is None:
            continue
        per[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (fam, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            cnt[fam] += 1
            per[fam]["_ns"] += dur.get(r["Dispatch_Id"], 0)
    res = {"#meta": {"commit": commit, "formula": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8)",
                     "note": "counters summed over the launches of a kernel family in one profiled run (profiled kernels "
                             "run serialised and ~3-5 % below the un-profiled clock)"}}
    for fam, d in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
        gui = d.get("GRBM_GUI_ACTIVE", 0.0)
        if gui <= 0 or d.get("SQ_INSTS_MFMA", 0.0) <= 0:
            continue
        res[fam] = {"launches": cnt[fam], "mfma_busy": round(d["SQ_VALU_MFMA_BUSY_CYCLES"] * 8.0 / (1024.0 * gui), 4),
                    "cycles_per_mfma": round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_INSTS_MFMA"], 2),
                    "effective_clock_ghz": round(gui / 8.0 / d["_ns"], 3) if d["_ns"] else None,
                    "total_ms": round(d["_ns"] / 1e6, 3)}
    json.dump(res, open(out, "w"), indent=1)
    print("wrote", out, len(res) - 1, "kernel families")


if __name__ == "__main__":
    main()
