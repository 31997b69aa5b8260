"""Per-kernel VGPR / scratch / occupancy / LDS summary of one HIP source, cross-compiled for gfx950 (no GPU needed):
    python tools/kres.py ccvpe_amd/csrc/matching_bwd.hip [name filter] [-DNAME=VALUE ...]
A kernel that shows scratch or a VGPR count at the 128 / 256 occupancy cliffs is the first thing to look at before a GPU run."""
import os
import re
import subprocess
import sys

src = sys.argv[1]
filt = [a for a in sys.argv[2:] if not a.startswith("-")]
defs = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage",
       "-c", os.path.basename(src), "-o", "/tmp/kres_%d.o" % os.getpid()] + defs
if os.path.basename(src).startswith("matching"):       # same per-file flag as csrc/Makefile ($(NOSLP))
    cmd.insert(1, "-fno-slp-vectorize")
res = subprocess.run(cmd, cwd=os.path.dirname(os.path.abspath(src)), capture_output=True, text=True)
cur = None
rows = []
for line in res.stderr.splitlines():
    m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" ")[0]] = v
try:
    os.remove("/tmp/kres_%d.o" % os.getpid())
except OSError:
    pass
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    if filt and not any(f in n for f in filt):
        continue
    print("%-70s vgpr %4s agpr %3s scratch %5s occ %s lds %s" % (n[:70], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"),
                                                                 r.get("Occupancy"), r.get("LDS")))
