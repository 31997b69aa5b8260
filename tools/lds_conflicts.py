"""LDS bank-conflict share per kernel from ONE rocprofv3 --pmc pass (CSV):
    --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE  --kernel-trace
  conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE   (extra cycles / all LDS-array cycles, MI355X_MICROARCH.md)
  lds_share      = SQ_LDS_IDX_ACTIVE / (256 CUs x GRBM_GUI_ACTIVE / 8)   (how much of the kernel the LDS arrays were busy)
A kernel with a large lds_share AND a large conflict_share is losing time to its LDS layout (round 3: the matching backward's
a_i table at pitch 256 — four rows a wave reads together on one bank).
    python tools/lds_conflicts.py <counter_collection.csv> <kernel_trace.csv> <out.json> [commit]"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name).replace("void ", "").replace("ccvpe::", "")
    return name[:90]


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
        nm = r["Kernel_Name"]
        if "ccvpe" not in nm:
            continue
        fam = short(nm)
        per[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (fam, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            cnt[fam] += 1
            per[fam]["_ns"] += dur.get(r["Dispatch_Id"], 0)
    res = {"#meta": {"commit": commit, "conflict_share": "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE",
                     "lds_share": "SQ_LDS_IDX_ACTIVE / (256 * GRBM_GUI_ACTIVE / 8)",
                     "note": "summed over the launches of a kernel in one profiled run; sorted by conflict cycles"}}
    for fam, d in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_LDS_BANK_CONFLICT", 0.0)):
        act = d.get("SQ_LDS_IDX_ACTIVE", 0.0)
        gui = d.get("GRBM_GUI_ACTIVE", 0.0)
        if act <= 0 or gui <= 0:
            continue
        res[fam] = {"launches": cnt[fam], "total_ms": round(d["_ns"] / 1e6, 3),
                    "conflict_share": round(d.get("SQ_LDS_BANK_CONFLICT", 0.0) / act, 4),
                    "lds_share": round(act * 8.0 / (256.0 * gui), 4)}
    json.dump(res, open(out, "w"), indent=1)
    print("wrote", out, len(res) - 1, "kernels")
    for fam, v in list(res.items())[1:26]:
        print("%-92s x%-4d %8.2f ms  conflict %.3f  lds %.3f" % (fam, v["launches"], v["total_ms"], v["conflict_share"], v["lds_share"]))


if __name__ == "__main__":
    main()
