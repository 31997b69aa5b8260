"""Static guard on the compiled kernels (no GPU needed): the build records hipcc's per-kernel resource remarks in
ccvpe_amd/csrc/<source>.res; no kernel of the library may use scratch memory (= spilled registers), with one known,
measured exception.  This is the check that would have caught `match_bwd_kernel<20>` before its GPU run (the SLP vectoriser
spilled its per-shift arrays: 2.9 KB of scratch per lane, 5x slower — DESIGN.md §4); `tools/kres.py` prints the same numbers."""
import glob
import os
import re

from ccvpe_amd import _lib

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ccvpe_amd", "csrc")
# mangled-name fragment -> bytes per lane allowed (the 8-wave 128-column 3x3 tile without DMA lives under a 128-VGPR cap)
ALLOWED = {"conv3x3_kernelIfLi4ELi4ELi2ELi8ELb0ELi1E": 32}


def _kernels():
    _lib.build(verbose=False)                     # incremental; writes the .res files next to the objects
    out = []
    for path in sorted(glob.glob(os.path.join(CSRC, "*.res"))):
        cur = None
        for line in open(path, errors="replace"):
            m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|Occupancy \[waves/SIMD\]): (\S+)", line)
            if not m:
                continue
            k, v = m.group(1).split(" ")[0], m.group(2)
            if k == "Function":
                cur = {"name": v, "file": os.path.basename(path)}
                out.append(cur)
            elif cur is not None:
                cur[m.group(1)] = v
    return out


def test_every_kernel_reports_resources_and_none_spills():
    ks = _kernels()
    assert len(ks) > 300, "resource remarks missing (is the Makefile still passing -Rpass-analysis=kernel-resource-usage?)"
    bad = []
    for k in ks:
        scratch = int(k.get("ScratchSize [bytes/lane]", "0"))
        limit = max([v for frag, v in ALLOWED.items() if frag in k["name"]] + [0])
        if scratch > limit:
            bad.append("%s: %s scratch %d B/lane (VGPRs %s, AGPRs %s)" % (k["file"], k["name"][:90], scratch, k.get("VGPRs"), k.get("AGPRs")))
    assert not bad, "kernels spilling registers to scratch:\n" + "\n".join(bad)


def test_hot_kernels_keep_two_waves_per_simd():
    """The MFMA kernels are sized for >= 2 waves per SIMD (two workgroups per CU); a change that pushes one of them over 256
    registers halves that silently."""
    ks = _kernels()
    hot = [k for k in ks if re.search(r"(conv3x3_kernel|igemm_kernel|upconv_kernel|upconv_halo_kernel|pw_gemm_kernel|conv_wgrad_kernel)I", k["name"])]
    assert len(hot) > 100
    low = [k["name"][:90] for k in hot if int(k.get("Occupancy [waves/SIMD]", "0")) < 2]
    assert not low, "occupancy below 2 waves/SIMD:\n" + "\n".join(low)


def test_product_library_reads_no_environment_switches():
    """A/B switches (CCVPE_CONV3_*, CCVPE_PW_GEMM, CCVPE_*_ABLATE) and the experiments behind them live under
    `#ifdef CCVPE_ABLATE` (diagnostics build: `make EXTRA=-DCCVPE_ABLATE`); the product library neither imports getenv nor
    carries the W-from-L2 experiment kernel."""
    import subprocess
    _lib.build(verbose=False)
    syms = subprocess.run(["nm", "-D", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in syms
    res = "".join(open(p, errors="replace").read() for p in glob.glob(os.path.join(CSRC, "conv3x3_*.res")))
    assert "conv3x3_kernel" in res and "conv3x3_wreg_kernel" not in res
    for path in glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")):
        depth = 0
        for n, line in enumerate(open(path), 1):
            s = line.strip()
            if s.startswith("#ifdef CCVPE_ABLATE"):
                depth += 1
            elif s.startswith("#if") and depth:
                depth += 1
            elif s.startswith("#endif") and depth:
                depth -= 1
            elif s.startswith("#else") and depth == 1:
                depth = 0                     # the product branch of an #ifdef CCVPE_ABLATE / #else pair
            code = line.split("//")[0]
            assert depth or "getenv(" not in code, "%s:%d reads the environment in the product build" % (os.path.basename(path), n)


def test_diagnostics_build_still_compiles():
    """`make EXTRA=-DCCVPE_ABLATE` (tools/gpu/ablate_*.sh) compiles code the product build never sees: the experiment kernels and
    their switches.  A syntax-only pass over the translation units that carry `#ifdef CCVPE_ABLATE` blocks keeps an edit of
    the shared kernel code from breaking them unnoticed (~2 s per file, no code generation)."""
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    units = [p for p in sorted(glob.glob(os.path.join(CSRC, "*.hip")))
             if "CCVPE_ABLATE" in open(p).read()
             or any("CCVPE_ABLATE" in open(os.path.join(CSRC, h)).read()
                    for h in re.findall(r'#include "(\w+\.h)"', open(p).read()) if os.path.isfile(os.path.join(CSRC, h)))]
    assert any(os.path.basename(u).startswith("conv3x3_") for u in units) and len(units) >= 3
    for u in units:
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-DCCVPE_ABLATE", "-fsyntax-only", u],
                           capture_output=True, text=True, cwd=CSRC)
        assert r.returncode == 0, "%s does not compile with -DCCVPE_ABLATE:\n%s" % (os.path.basename(u), r.stderr[-2000:])
