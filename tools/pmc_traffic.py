"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; CSV output) into
profiles/pmc_traffic.json: kernel family -> HBM bytes per launch.

Corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950
FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B/lane) coalesced read, which is how
every kernel here loads, so the read side is doubled.  WRITE_SIZE is taken as is (uncalibrated).
Usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [commit] [workload]
"""
import csv
import json
import re
import sys
from collections import defaultdict


def family(name):
    """rocprof kernel name -> the name bench.py's launch recorder uses.  bf16 instantiations arrive MANGLED (the profiler's
    demangler does not know the __bf16 type code DF16b): _ZN5ccvpe14conv3x3_kernelIDF16bLi4ELi5ELi2ELi4ELb1EEEv..."""
    # round 5: the narrow-level kernels (bf16 only; template arguments are integers / bools) and upconv_dma_kernel's PAIR flag
    m = re.search(r"c3n_kernelILi(\d+)ELi\d+ELi\d+ELb[01]ELb([01])E", name) or re.search(r"c3n_kernel<(\d+), *\d+, *\d+, *(?:true|false), *(true|false)", name)
    if m:
        return "c3n_kernel<bf16,%d%s>" % (8 * int(m.group(1)), ",match" if m.group(2) in ("1", "true") else "")
    m = re.search(r"up2_kernelILi(\d+)ELi(\d+)ELi(\d+)E", name) or re.search(r"up2_kernel<(\d+), *(\d+), *(\d+)", name)
    if m:
        c0, nt = 8 * int(m.group(1)), int(m.group(3))
        return "up2_kernel<bf16,%d,%d>" % (c0, {3: 40, 2: 32}.get(nt, 16 * nt))
    m = re.search(r"pw2_kernel<(float|__bf16), *(\d+), *(\d+), *(\d+)", name) or re.search(r"pw2_kernelI(f|DF16b)Li(\d+)ELi(\d+)ELi(\d+)E", name)
    if m:       # pointwise ring kernel: bench.py's recorder calls it by its route (ops.conv_route_name)
        t, a, b, c = m.groups()
        return ("pw_ring_f32_kernel<%s,%s,%s>" if t in ("float", "f") else "pw_ring_kernel<bf16,%s,%s,%s>") % (a, b, c)
    m = re.search(r"upconv_dma_kernelIDF16bLi(\d+)ELi(\d+)ELi(\d+)ELb([01])E", name)
    if m:
        return "upconv_dma_kernel<bf16,%s,%s,%s%s>" % (m.group(1), m.group(2), m.group(3), ",pair" if m.group(4) == "1" else "")
    m = re.search(r"upconv_dma_kernel<float, *(\d+), *(\d+), *(\d+), *(true|false)", name)
    if m:
        return "upconv_dma_kernel<f32,%s,%s,%s%s>" % (m.group(1), m.group(2), m.group(3), ",pair" if m.group(4) == "true" else "")
    m = re.search(r"(igemm_kernel|pw_gemm_kernel|conv3x3_kernel|upconv_kernel|upconv_halo_kernel)IDF16bLi(\d+)ELi(\d+)ELi(\d+)E", name)
    if m:
        k, a, b, c = m.groups()
        return "%s<bf16,%s,%s,%s>" % (k, a, b, c)
    m = re.search(r"(igemm_kernel|pw_gemm_kernel|conv3x3_kernel|upconv_kernel|upconv_halo_kernel)<(float|__bf16|bf16), *(\d+), *(\d+), *(\d+)[,>]", name)
    if m:
        k, t, a, b, c = m.groups()
        if t == "float":
            if k == "igemm_kernel":
                return "igemm_f32_kernel<%s,%s,%s>" % (a, b, c)
            if k == "pw_gemm_kernel":
                return "pw_gemm_f32_kernel<%s,%s,%s>" % (a, b, c)
            if k == "conv3x3_kernel":
                return "conv3x3_f32_kernel<%s,%s,%s>" % (a, b, c)
            return "%s<f32,%s,%s,%s>" % (k, a, b, c)
        return "%s<bf16,%s,%s,%s>" % (k, a, b, c)
    m = re.search(r"conv_wgrad_kernel<(\d+), *(\d+)>", name)
    if m:
        return "conv_wgrad_kernel<%s,%s>" % m.groups()
    m = re.search(r"_ZN5ccvpe\d+(\w+?_kernel)I", name)
    if m:
        return m.group(1)
    m = re.search(r"ccvpe::(\w+?)(<[^>]*>)?\(", name) or re.search(r"ccvpe::(\w+)", name)
    return m.group(1) if m else None


def collect(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            fam = family(row["Kernel_Name"])
            if fam is None:
                continue
            tot[fam] += float(row["Counter_Value"])
            cnt[fam] += 1
    return {k: tot[k] / cnt[k] for k in tot}, cnt


def main():
    fetch, n1 = collect(sys.argv[1], "FETCH_SIZE")
    write, n2 = collect(sys.argv[2], "WRITE_SIZE")
    out = {"#meta": {"commit": sys.argv[4] if len(sys.argv) > 4 else "unknown",
                     "workload": sys.argv[5] if len(sys.argv) > 5 else "",
                     "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (KiB); FETCH_SIZE x 2 "
                               "(gfx950 reports half of a 16 B/lane coalesced read, MI355X_MICROARCH.md HBM section), "
                               "WRITE_SIZE as is (uncalibrated); mean bytes per launch of each kernel family"}}
    # bytes a whole STEP moves according to the counters: every family's bytes per launch x its launches, over the executed
    # steps (counted from the stem launches: two per forward — ground and aerial)
    stem = max([n1.get(k, 0) for k in n1 if k.startswith("stem_dw_kernel") or k.startswith("stem_conv_kernel")] or [0])
    steps = stem / 2.0 if stem else 0.0
    total = 0.0
    for k in sorted(set(fetch) | set(write)):
        rd = 2.0 * fetch.get(k, 0.0) * 1024.0
        wr = write.get(k, 0.0) * 1024.0
        total += rd * n1.get(k, 0) + wr * n2.get(k, n1.get(k, 0))
        out[k] = round(rd + wr)
        out[k + "#detail"] = {"read_bytes_corrected_x2": round(rd), "write_bytes": round(wr),
                              "launches_sampled": int(n1.get(k, 0))}
    if steps:
        out["#meta"]["steps_executed"] = steps
        out["#meta"]["gb_per_step"] = round(total / steps / 1e9, 3)
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    print("wrote", sys.argv[3])


if __name__ == "__main__":
    main()
