"""Order of global loads / waits / barriers / LDS traffic / MFMA runs inside one kernel's ISA (from `hipcc -S --cuda-device-only`):
    python tools/isa_loop.py /tmp/ig.s <mangled-name-substring> [first_line] [n_lines]
Shows at a glance whether a stage's prefetch is waited for BEFORE the stage's matrix work (a select / mask applied at the load)
or only when it is written to LDS."""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cnt = int(sys.argv[4]) if len(sys.argv) > 4 else 200
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l and l.rstrip().split(":")[0].endswith("E") or (l.startswith("_ZN") and key in l))
body = []
for l in lines[start:]:
    body.append(l)
    if "s_endpgm" in l:
        break
pat = re.compile(r"^\s*(v_mfma\S*|s_waitcnt.*|s_barrier|global_load\S*|buffer_load\S*|ds_write\S*|ds_read\S*|global_store\S*|s_cbranch\S*\s+\S+|\.LBB\S+:|scratch_\S+)")
out = []
prev, n = None, 0
for i, l in enumerate(body):
    m = pat.match(l)
    if not m:
        continue
    tok = m.group(1)
    k = tok.split()[0] if not tok.startswith("s_waitcnt") and not tok.startswith("s_cbranch") and not tok.startswith(".LBB") else tok
    if k == prev:
        n += 1
        continue
    if prev is not None:
        out.append((first, prev, n))
    prev, n, first = k, 1, i
out.append((first, prev, n))
shown = 0
for first, k, n in out:
    if first < lo:
        continue
    print("%6d  %-44s x%d" % (first, k, n))
    shown += 1
    if shown >= cnt:
        break
print("total lines", len(body))
