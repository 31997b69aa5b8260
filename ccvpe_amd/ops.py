"""Python-side operator layer: torch tensors in, C-ABI calls out (include/ccvpe_hip.h).

Each function enqueues exactly one library call on torch's current HIP stream.  torch is used
for device memory and streams only; every arithmetic step runs in libccvpe_hip.so.  No
function here has a CPU or eager fallback.
"""
import ctypes

import torch

from . import _lib
from ._lib import ConvDesc, check

ACT_NONE, ACT_RELU, ACT_SWISH, ACT_RELU_MASK = 0, 1, 2, 3
SPLIT_K_CALLS = 0        # number of conv_igemm calls that took the split-K path (tests / diagnostics)
SPLIT_K = __import__("os").environ.get("CCVPE_SPLIT_K", "1") == "1"      # split-K for small-batch GEMMs (0 disables)
OUT_NHWC, OUT_DECONV2X = 0, 1

# Optional launch recorder (bench.py): when set, every library call is bracketed by HIP events on
# the launching stream and reported as (kernel family, tag, algorithmic flops, algorithmic bytes).
_recorder = None


def set_recorder(rec):
    global _recorder
    _recorder = rec


class LaunchRecorder(object):
    """Collects (name, tag, flops, bytes, start_event, end_event) per library call."""

    def __init__(self):
        self.items = []

    def begin(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        return ev

    def end(self, name, tag, flops, nbytes, ev0):
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record(torch.cuda.current_stream())
        self.items.append((name, tag, flops, nbytes, ev0, ev1))

    def summary(self):
        """name -> dict(calls, ms, flops, bytes); call after torch.cuda.synchronize()."""
        out = {}
        for name, tag, flops, nbytes, e0, e1 in self.items:
            d = out.setdefault(name, dict(calls=0, ms=0.0, flops=0.0, bytes=0.0))
            d["calls"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
            d["bytes"] += nbytes
        return out


def igemm_tile(n, is3x3=False):
    """Mirror of pick_cfg() in csrc/conv_igemm.hip: (BM, BN) chosen for a GEMM N (reporting only)."""
    npad = (n + 15) // 16 * 16
    cfgs = [(4, 5, 2), (4, 4, 2), (4, 3, 2), (4, 2, 2), (4, 1, 2), (4, 5, 1), (4, 3, 1), (4, 1, 1), (2, 7, 1)]
    best, best_cost = None, None
    for mt, nt, wn in cfgs:
        bn = 16 * nt * wn
        tiles = (npad + bn - 1) // bn
        cost = tiles * bn * 1000 + (1000 - bn)
        if best_cost is None or cost < best_cost:
            best, best_cost = (mt, nt, wn), cost
    mt, nt, wn = best
    return "%s<%d,%d,%d>" % ("conv3x3_f32_kernel" if is3x3 else "igemm_f32_kernel", mt, nt, wn)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


ROUTE_FAMILIES = {0: "igemm", 1: "pw_gemm", 2: "conv3x3", 3: "c3n", 4: "pw_ring", 5: "pwn"}


def conv_route(desc, is_bf16, out_f32=False):
    """(family, MT, NT, WN) of the kernel ccvpe_conv_igemm_f32 / _bf16 runs for `desc` (ccvpe_conv_igemm_route: the
    library's own dispatch, nothing is launched)."""
    r = _lib.load().ccvpe_conv_igemm_route(ctypes.byref(desc), int(bool(is_bf16)), int(bool(out_f32)))
    if r < 0:
        check(int(r), "ccvpe_conv_igemm_route")
    return ROUTE_FAMILIES[r & 0xff], (r >> 8) & 0xf, (r >> 12) & 0xf, (r >> 16) & 0xf


UPROUTE_KERNELS = {0: "upconv_kernel", 1: "upconv_halo_kernel", 2: "upconv_dma_kernel", 3: "upconv_dma_kernel", 4: "up2_kernel"}


def upconv_route(desc, is_bf16):
    """(kernel, MT, NT, WN, pair) ccvpe_upconv3x3_f32 / _bf16 runs for `desc` (nothing is launched)."""
    r = _lib.load().ccvpe_upconv3x3_route(ctypes.byref(desc), int(bool(is_bf16)))
    if r < 0:
        check(int(r), "ccvpe_upconv3x3_route")
    return UPROUTE_KERNELS[r & 0xff], (r >> 8) & 0xf, (r >> 12) & 0xf, (r >> 16) & 0xf, (r & 0xff) == 3


def upconv_route_name(desc, is_bf16):
    """Kernel name as rocprofv3 prints it (minus namespace): the key of profiles/pmc_traffic.json."""
    k, mt, nt, wn, pair = upconv_route(desc, is_bf16)
    ty = "bf16" if is_bf16 else "f32"
    if k == "up2_kernel":
        return "up2_kernel<bf16,%d,%d>" % (desc.c0, desc.n)
    return "%s<%s,%d,%d,%d%s>" % (k, ty, mt, nt, wn, ",pair" if pair else "")


def conv_route_name(desc, is_bf16, out_f32=False):
    """Kernel family name of the launch the library would make for this descriptor, as the launch recorder and
    profiles/pmc_traffic.json spell it: `<family>_f32_kernel<MT,NT,WN>` in fp32, `<family>_kernel<bf16,MT,NT,WN>` in bf16
    (the narrow 3x3 kernel c3n exists in bf16 only and is named by its column count)."""
    fam, mt, nt, wn = conv_route(desc, is_bf16, out_f32)
    if fam == "c3n":
        return "c3n_kernel<bf16,%d>" % desc.n
    if is_bf16:
        return "%s_kernel<bf16,%d,%d,%d>" % (fam, mt, nt, wn)
    return "%s_f32_kernel<%d,%d,%d>" % (fam, mt, nt, wn)


# Plan recording (ccvpe_amd/plan.py): while a forward is being recorded every tensor this layer allocates comes from the
# recorder's workspace and every tensor whose pointer goes to the library is remembered (weights are collected from those).
_record = None


def _empty(shape, device=None, dtype=torch.float32):
    if _record is not None:
        return _record.empty(shape, dtype, device)
    return torch.empty(shape, device=device, dtype=dtype)


def _ptr(t):
    if t is None:
        return None
    if _record is not None:
        _record.note(t)
    return ctypes.c_void_p(t.data_ptr())


def _chk(t, name, dtype=torch.float32):
    if t is None:
        return
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise ValueError("%s must be a contiguous %s device tensor (got %s %s contiguous=%s)" % (
            name, dtype, t.device, t.dtype, t.is_contiguous()))


def _act_dtype(t):
    """Activation storage type of a call: fp32 or bf16 (selects the _f32 / _bf16 entry point)."""
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("activations must be fp32 or bf16, got %s" % t.dtype)
    return t.dtype


def conv_igemm(src0, c0, w_packed, n, *, batch, in_h, in_w, kh=1, kw=1, stride=1, pad=0,
               src1=None, c1=0, gate=None, scale=None, shift=None, residual=None, act=ACT_NONE,
               out_mode=OUT_NHWC, dst=None, ldd=None, ld0=None, ld1=None, algo_k=None, out_f32=False, route_only=False):
    """Implicit-GEMM conv / deconv / linear (ccvpe_conv_igemm_f32 / _bf16 by src0.dtype).
    src tensors are NHWC.  out_f32 (bf16 only): write an fp32 result."""
    lib = _lib.load()
    dt = _act_dtype(src0)
    odt = torch.float32 if (out_f32 or dt == torch.float32) else dt
    for t, nm in ((src0, "src0"), (src1, "src1"), (w_packed, "w"), (residual, "residual")):
        _chk(t, nm, dt)
    for t, nm in ((gate, "gate"), (scale, "scale"), (shift, "shift")):
        _chk(t, nm)
    _chk(dst, "dst", odt)
    ld0 = ld0 if ld0 is not None else src0.shape[-1]
    ld1 = ld1 if ld1 is not None else (src1.shape[-1] if src1 is not None else 0)
    ho = (in_h + 2 * pad - kh) // stride + 1
    wo = (in_w + 2 * pad - kw) // stride + 1
    if out_mode == OUT_DECONV2X:
        cout = n // 4
        oshape = (batch, 2 * ho, 2 * wo)
    else:
        cout = n
        oshape = (batch, ho, wo)
    if dst is None:
        ldd = ldd if ldd is not None else cout
        dst = _empty(oshape + (ldd,), device=src0.device, dtype=odt)
    elif ldd is None:
        ldd = dst.shape[-1]
    d = ConvDesc()
    d.src0, d.src1, d.gate, d.w = _ptr(src0), _ptr(src1), _ptr(gate), _ptr(w_packed)
    d.scale, d.shift, d.residual, d.dst = _ptr(scale), _ptr(shift), _ptr(residual), _ptr(dst)
    d.c0, d.ld0, d.c1, d.ld1 = c0, ld0, c1, ld1
    d.batch, d.in_h, d.in_w = batch, in_h, in_w
    d.kh, d.kw, d.stride, d.pad = kh, kw, stride, pad
    d.n, d.kpad = n, w_packed.shape[1]
    d.ldd = ldd
    d.ldres = residual.shape[-1] if residual is not None else 0
    d.act, d.out_mode = act, out_mode
    if route_only:        # (family, MT, NT, WN) the one-pass entry point would run for this call; nothing is launched
        return conv_route(d, dt != torch.float32, bool(out_f32))
    rec = _recorder
    ev0 = rec.begin() if rec is not None else None
    want = lib.ccvpe_conv_igemm_splitk_floats(ctypes.byref(d), int(dt != torch.float32)) if SPLIT_K else 0
    if want < 0:
        check(int(want), "ccvpe_conv_igemm_splitk_floats")
    if want > 0:          # small-batch GEMM: K cut into slices + deterministic second pass (csrc/conv_igemm.hip)
        global SPLIT_K_CALLS
        SPLIT_K_CALLS += 1
        scratch = _empty((want,), device=src0.device, dtype=torch.float32)
        if dt == torch.float32:
            check(lib.ccvpe_conv_igemm_splitk_f32(ctypes.byref(d), _ptr(scratch), _stream()), "ccvpe_conv_igemm_splitk_f32")
        else:
            check(lib.ccvpe_conv_igemm_splitk_bf16(ctypes.byref(d), int(bool(out_f32)), _ptr(scratch), _stream()),
                  "ccvpe_conv_igemm_splitk_bf16")
    elif dt == torch.float32:
        check(lib.ccvpe_conv_igemm_f32(ctypes.byref(d), _stream()), "ccvpe_conv_igemm_f32")
    else:
        check(lib.ccvpe_conv_igemm_bf16(ctypes.byref(d), int(bool(out_f32)), _stream()), "ccvpe_conv_igemm_bf16")
    if rec is not None:
        m = batch * ho * wo
        k_alg = algo_k if algo_k is not None else kh * kw * (c0 + c1)
        flops = 2.0 * m * n * k_alg
        # algorithmic bytes: input read once, output written once, weights once
        esz = 4.0 if dt == torch.float32 else 2.0
        nbytes = esz * (batch * in_h * in_w * (c0 + c1) + m * n + n * k_alg
                        + (m * n if residual is not None else 0))
        if want > 0:      # split-K: always the generic gather kernel with the tile pick_cfg() chose
            name = igemm_tile(n, False)
        else:             # ask the library which kernel it ran (no Python mirror of the dispatch rules)
            name = conv_route_name(d, dt != torch.float32, bool(out_f32))
        if want > 0 and dt != torch.float32:
            name = name.replace("_f32_kernel<", "_kernel<bf16,")
        rec.end(name, "%dx%d s%d M%d N%d K%d" % (kh, kw, stride, m, n, k_alg), flops, nbytes, ev0)
    return dst


def upconv3x3(src0, c0, w_packed, shift9, n, *, batch, h1, w1, src1=None, c1=0, act=ACT_NONE, algo_flops=None, route_only=False):
    """ConvTranspose2d(k2,s2) folded into the following 3x3 conv (ccvpe_upconv3x3_f32 / _bf16).
    src0 [B,h1,w1,ld0] low-res, src1 [B,2h1,2w1,ld1] skip; returns [B,2h1,2w1,n]."""
    lib = _lib.load()
    dt = _act_dtype(src0)
    for t, nm in ((src0, "src0"), (src1, "src1"), (w_packed, "w")):
        _chk(t, nm, dt)
    _chk(shift9, "shift9")
    dst = _empty((batch, 2 * h1, 2 * w1, n), device=src0.device, dtype=dt)
    d = _lib.UpconvDesc()
    d.src0, d.src1, d.w, d.shift9, d.dst = _ptr(src0), _ptr(src1), _ptr(w_packed), _ptr(shift9), _ptr(dst)
    d.c0, d.ld0, d.c1, d.ld1 = c0, src0.shape[-1], c1, (src1.shape[-1] if src1 is not None else 0)
    d.batch, d.h1, d.w1 = batch, h1, w1
    d.n, d.kpad, d.ldd, d.act = n, w_packed.shape[-1], n, act
    if route_only:
        return upconv_route(d, dt != torch.float32)
    rec = _recorder
    ev0 = rec.begin() if rec is not None else None
    fn = lib.ccvpe_upconv3x3_f32 if dt == torch.float32 else lib.ccvpe_upconv3x3_bf16
    check(fn(ctypes.byref(d), _stream()), "ccvpe_upconv3x3")
    if rec is not None:
        m = batch * h1 * w1 * 4
        k_eff = 4 * c0 + 9 * c1
        flops = algo_flops if algo_flops is not None else 2.0 * m * n * k_eff
        esz = 4.0 if dt == torch.float32 else 2.0
        nbytes = esz * (batch * h1 * w1 * c0 + m * c1 + m * n + 4 * n * k_eff)
        name = upconv_route_name(d, dt != torch.float32)                  # the library's own dispatch (ccvpe_upconv3x3_route)
        rec.end(name, "up3x3 M%d N%d Keff%d" % (m, n, k_eff), flops, nbytes, ev0)
    return dst


def tail512(x, c0, w_packed, shift9, w2, b2, cout, normalize, *, batch, h1, w1, split=False, want_softmax=False):
    """The whole 512 x 512 decoder level in one launch (ccvpe_tail512_f32 / _bf16): folded deconv + conv.0 + ReLU + conv.2
    (+ F.normalize for cout = 2).  x [B,h1,w1,ld0]; w_packed / shift9 from models._pack_upconv (n = 16, no skip);
    w2 [cout,3,3,16], b2 [cout] fp32; returns [B,cout,2h1,2w1] fp32."""
    lib = _lib.load()
    dt = _act_dtype(x)
    _chk(x, "x", dt)
    _chk(w_packed, "w", dt)
    for t, nm in ((shift9, "shift9"), (w2, "w2"), (b2, "b2")):
        _chk(t, nm)
    if tuple(w2.shape) != (cout, 3, 3, 16) or tuple(shift9.shape) != (9, 16) or w_packed.shape[0] != 4 or w_packed.shape[1] != 16:
        raise ValueError("tail512: w [4,16,kpad], shift9 [9,16], w2 [cout,3,3,16] expected")
    out = _empty((batch, cout, 2 * h1, 2 * w1), device=x.device, dtype=torch.float32)
    d = _lib.TailDesc()
    d.x, d.w, d.shift9, d.w2, d.b2, d.out = _ptr(x), _ptr(w_packed), _ptr(shift9), _ptr(w2), _ptr(b2), _ptr(out)
    d.batch, d.h1, d.w1 = batch, h1, w1
    d.c0, d.ld0, d.kpad = c0, x.shape[-1], w_packed.shape[-1]
    d.cout, d.normalize, d.split = cout, int(bool(normalize)), int(bool(split))
    smx = None
    if want_softmax:                         # cout = 1: per-(tile, wave) softmax partials for softmax_apply()
        npart = lib.ccvpe_tail512_partials(ctypes.byref(d), int(dt != torch.float32))
        if npart <= 0:
            check(int(npart) if npart < 0 else -1, "ccvpe_tail512_partials")
        smx = _empty((batch, npart, 2), device=x.device, dtype=torch.float32)
        d.softmax_partial = _ptr(smx)
    rec = _recorder
    ev0 = rec.begin() if rec is not None else None
    fn = lib.ccvpe_tail512_f32 if dt == torch.float32 else lib.ccvpe_tail512_bf16
    check(fn(ctypes.byref(d), _stream()), "ccvpe_tail512")
    if rec is not None:
        m = batch * h1 * w1 * 4
        esz = 4.0 if dt == torch.float32 else 2.0
        rec.end("tail512_kernel<%s,%d>" % (("f32" if not split else "f32 as bf16 hi+lo") if dt == torch.float32 else "bf16", cout), "tail M%d Keff%d" % (m, 4 * c0),
                2.0 * m * 16 * (4 * c0 + 9 * cout), esz * batch * h1 * w1 * c0 + 4.0 * m * cout, ev0)
    return (out, smx) if want_softmax else out


def softmax_apply(logits, partials):
    """Softmax(dim=-1) of logits [rows, n] from the (max, sum exp) partials tail512 wrote ([rows, P, 2])."""
    lib = _lib.load()
    _chk(logits, "logits")
    _chk(partials, "partials")
    rows, n = logits.shape
    out = _empty(logits.shape, device=logits.device, dtype=logits.dtype)
    check(lib.ccvpe_softmax_apply_f32(_ptr(logits), _ptr(partials), partials.shape[1], _ptr(out), rows, n, _stream()),
          "ccvpe_softmax_apply_f32")
    return out


def tail512_ok(h1, w1, n_a):
    """Shapes the fused 512 x 512 level handles (everything else runs ccvpe_upconv3x3 + ccvpe_head_conv3x3)."""
    return n_a == 16 and h1 % 16 == 0 and w1 % 16 == 0


def stem_conv(x_nchw, w, scale, shift, circular, out_dtype=torch.float32):
    lib = _lib.load()
    for t, nm in ((x_nchw, "x"), (w, "w"), (scale, "scale"), (shift, "shift")):
        _chk(t, nm)
    b, c, h, wd = x_nchw.shape
    if c != 3:
        raise ValueError("stem expects 3 input channels")
    ho, wo = (h + 1 - 3) // 2 + 1, (wd + 1 - 3) // 2 + 1
    y = _empty((b, ho, wo, 32), device=x_nchw.device, dtype=out_dtype)
    fn = lib.ccvpe_stem_conv_f32 if out_dtype == torch.float32 else lib.ccvpe_stem_conv_bf16
    check(fn(_ptr(x_nchw), _ptr(w), _ptr(scale), _ptr(shift), _ptr(y), b, h, wd, int(bool(circular)), _stream()),
          "ccvpe_stem_conv")
    return y


def stem_dw_supported(in_h, in_w, circular):
    """SE partial rows per sample of the fused stem + block-0 depthwise launch, 0 when the shape runs unfused."""
    return _lib.load().ccvpe_stem_dw_nblk(in_h, in_w, int(bool(circular)))


def stem_dw(x_nchw, w, s0, b0, w_dw, s1, b1, circular, out_dtype=torch.float32):
    """Stem conv + BN + swish -> block-0 depthwise 3x3 + BN + swish + SE squeeze partials, the stem tensor kept in LDS
    (csrc/stem_dw.hip); returns (y [B,Ho,Wo,32], se_partial [B,nblk,32])."""
    lib = _lib.load()
    for t, nm in ((x_nchw, "x"), (w, "w"), (s0, "s0"), (b0, "b0"), (w_dw, "w_dw"), (s1, "s1"), (b1, "b1")):
        _chk(t, nm)
    b, c, h, wd = x_nchw.shape
    if c != 3 or w_dw.numel() != 9 * 32:
        raise ValueError("stem_dw expects 3 input channels and a 3x3x32 depthwise weight")
    nblk = stem_dw_supported(h, wd, circular)
    if nblk <= 0:
        raise _lib.CcvpeError("stem_dw: unsupported shape %s" % (tuple(x_nchw.shape),))
    ho, wo = (h + 1 - 3) // 2 + 1, (wd + 1 - 3) // 2 + 1
    y = _empty((b, ho, wo, 32), device=x_nchw.device, dtype=out_dtype)
    part = _empty((b, nblk, 32), device=x_nchw.device, dtype=torch.float32)
    rec = _recorder
    ev0 = rec.begin() if rec is not None else None
    fn = lib.ccvpe_stem_dw_f32 if out_dtype == torch.float32 else lib.ccvpe_stem_dw_bf16
    check(fn(_ptr(x_nchw), _ptr(w), _ptr(s0), _ptr(b0), _ptr(w_dw), _ptr(s1), _ptr(b1), _ptr(y), _ptr(part), b, h, wd,
             int(bool(circular)), _stream()), "ccvpe_stem_dw")
    if rec is not None:
        flops = 2.0 * b * ho * wo * 32 * (27 + 9)
        nbytes = 4.0 * b * 3 * h * wd + (4.0 if out_dtype == torch.float32 else 2.0) * b * ho * wo * 32
        rec.end("stem_dw_kernel", "in %dx%dx3" % (h, wd), flops, nbytes, ev0)
    return y, part


def dwconv(x, w, scale, shift, k, stride, circular):
    """Depthwise conv + BN + swish; returns (y, se_partial [B,nblk,C])."""
    lib = _lib.load()
    dt = _act_dtype(x)
    _chk(x, "x", dt)
    for t, nm in ((w, "w"), (scale, "scale"), (shift, "shift")):
        _chk(t, nm)
    b, h, wd, c = x.shape
    tot = (k - 1) if stride == 1 else (k - 2)
    ho, wo = (h + tot - k) // stride + 1, (wd + tot - k) // stride + 1
    nblk = lib.ccvpe_dwconv_nblk(h, wd, c, stride)
    if nblk <= 0:
        raise _lib.CcvpeError("ccvpe_dwconv_nblk rejected shape %s" % (tuple(x.shape),))
    y = _empty((b, ho, wo, c), device=x.device, dtype=dt)
    part = _empty((b, nblk, c), device=x.device, dtype=torch.float32)
    fn = lib.ccvpe_dwconv_f32 if dt == torch.float32 else lib.ccvpe_dwconv_bf16
    check(fn(_ptr(x), _ptr(w), _ptr(scale), _ptr(shift), _ptr(y), _ptr(part), b, h, wd, c, k, stride,
             int(bool(circular)), _stream()), "ccvpe_dwconv")
    return y, part


def mbconv_front_supported(in_h, in_w, cin, mid, k, stride):
    n = _lib.load().ccvpe_mbconv_front_nblk(in_h, in_w, cin, mid, k, stride)
    if n < 0:
        raise _lib.CcvpeError("ccvpe_mbconv_front_nblk rejected k=%d stride=%d" % (k, stride))
    return n


def mbconv_front(x, w_exp, s0, b0, w_dw, s1, b1, mid, k, stride, circular):
    """Fused expand + depthwise (+BN+swish each) + SE squeeze partials; returns (y, se_partial)."""
    lib = _lib.load()
    dt = _act_dtype(x)
    _chk(x, "x", dt)
    _chk(w_exp, "w_exp", dt)
    for t, nm in ((s0, "s0"), (b0, "b0"), (w_dw, "w_dw"), (s1, "s1"), (b1, "b1")):
        _chk(t, nm)
    b, h, wd, cin = x.shape
    nblk = mbconv_front_supported(h, wd, cin, mid, k, stride)
    if nblk == 0:
        raise _lib.CcvpeError("mbconv_front: unsupported shape %s" % (tuple(x.shape),))
    tot = (k - 1) if stride == 1 else (k - 2)
    ho, wo = (h + tot - k) // stride + 1, (wd + tot - k) // stride + 1
    y = _empty((b, ho, wo, mid), device=x.device, dtype=dt)
    part = _empty((b, nblk, mid), device=x.device, dtype=torch.float32)
    rec = _recorder
    ev0 = rec.begin() if rec is not None else None
    fn = lib.ccvpe_mbconv_front_f32 if dt == torch.float32 else lib.ccvpe_mbconv_front_bf16
    check(fn(_ptr(x), _ptr(w_exp), w_exp.shape[1], _ptr(s0), _ptr(b0), _ptr(w_dw), _ptr(s1), _ptr(b1), _ptr(y),
             _ptr(part), b, h, wd, cin, mid, k, stride, int(bool(circular)), _stream()), "ccvpe_mbconv_front")
    if rec is not None:
        flops = 2.0 * b * h * wd * cin * mid + 2.0 * b * ho * wo * mid * k * k
        nbytes = (4.0 if dt == torch.float32 else 2.0) * (b * h * wd * cin + b * ho * wo * mid)
        route = lib.ccvpe_mbconv_front_route(h, wd, cin, mid, k, stride, int(dt != torch.float32), b)
        kname = {1: "mbconv_front_kernel", 2: "mbconv_plane_kernel", 3: "mbconv_band_kernel"}.get(route, "mbconv_front_kernel")
        rec.end("%s<%d,%d>" % (kname, k, stride), "in %dx%dx%d mid %d" % (h, wd, cin, mid), flops, nbytes, ev0)
    return y, part


def se_gate(part, hw, w1, b1, w2, b2):
    lib = _lib.load()
    for t, nm in ((part, "part"), (w1, "w1"), (b1, "b1"), (w2, "w2"), (b2, "b2")):
        _chk(t, nm)
    b, nblk, c = part.shape
    cs = w1.shape[0]
    gate = _empty((b, c), device=part.device, dtype=torch.float32)
    check(lib.ccvpe_se_gate_f32(_ptr(part), nblk, 1.0 / float(hw), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2),
                                _ptr(gate), b, c, cs, _stream()), "ccvpe_se_gate_f32")
    return gate


def ground_descriptor(y1, wh, bh, cd):
    lib = _lib.load()
    for t, nm in ((y1, "y1"), (wh, "wh"), (bh, "bh")):
        _chk(t, nm)
    b, h, w, ld = y1.shape
    cd_arr = (ctypes.c_int * 6)(*cd)
    out = _empty((b, w * sum(cd)), device=y1.device, dtype=torch.float32)
    check(lib.ccvpe_ground_descriptor_f32(_ptr(y1), ld, _ptr(wh), _ptr(bh), cd_arr, _ptr(out), b, h, w,
                                          _stream()), "ccvpe_ground_descriptor_f32")
    return out


def conv3x3_match1(src0, c0, w_packed, n, g, L, shift, stride, ldo, *, batch, in_h, in_w, bias, window_offset=0, out_f32=False,
                   query_only=False):
    """convK.2 (3x3, bias, no activation) of a decoder level WITH the next level's one-hypothesis matching in its epilogue
    (ccvpe_conv3x3_match1_bf16, csrc/narrow_impl.h c3n_kernel<MATCH>): returns (scores [B,1,H,W] fp32, cat [B,H,W,ldo] bf16 / fp32)
    = what conv_igemm(...) followed by match_level(x, g, L, [shift], 1, 0, stride, ldo) returns; x is never written.
    query_only: True / False — does the library serve this layer and shape (else the caller runs the two ops)."""
    lib = _lib.load()
    dt = _act_dtype(src0)
    if dt != torch.bfloat16:
        if query_only:
            return False
        raise ValueError("conv3x3_match1: bf16 storage only")
    _chk(src0, "src0", dt)
    _chk(w_packed, "w", dt)
    _chk(bias, "bias")
    d = ConvDesc()
    d.src0, d.src1, d.gate, d.w = _ptr(src0), None, None, _ptr(w_packed)
    d.scale, d.shift, d.residual = None, _ptr(bias), None
    d.c0, d.ld0, d.c1, d.ld1 = c0, src0.shape[-1], 0, 0
    d.batch, d.in_h, d.in_w = batch, in_h, in_w
    d.kh, d.kw, d.stride, d.pad = 3, 3, 1, 1
    d.n, d.kpad, d.ldd, d.ldres = n, w_packed.shape[1], ldo, 0
    d.act, d.out_mode = ACT_NONE, OUT_NHWC
    if query_only:
        return lib.ccvpe_conv3x3_match1_ok(ctypes.byref(d), int(bool(out_f32)), L) == 1
    if not g.is_cuda or g.dtype != torch.float32 or g.stride(-1) != 1:
        raise ValueError("g must be an fp32 device tensor with unit inner stride")
    scores = _empty((batch, 1, in_h, in_w), device=src0.device, dtype=torch.float32)
    cat = _empty((batch, in_h, in_w, ldo), device=src0.device, dtype=torch.float32 if out_f32 else dt)
    d.dst = _ptr(cat)
    rec = _recorder
    ev0 = rec.begin() if rec is not None else None
    check(lib.ccvpe_conv3x3_match1_bf16(ctypes.byref(d), int(bool(out_f32)), _ptr(g), g.stride(0), L, int(shift), int(stride),
                                        int(window_offset), _ptr(scores), _stream()), "ccvpe_conv3x3_match1_bf16")
    if rec is not None:
        m = batch * in_h * in_w
        rec.end("c3n_kernel<bf16,%d,match>" % n, "3x3+match M%d N%d K%d" % (m, n, 9 * c0), 2.0 * m * n * 9 * c0,
                2.0 * m * c0 + (4.0 if out_f32 else 2.0) * m * ldo + 4.0 * m + 2.0 * n * 9 * c0, ev0)
    return scores, cat


def match_level(x, g, L, shifts, n_max, n_tail, stride, ldo, channels=None, window_offset=0):
    """Fused rotational matching.  x [B,H,W,ldx]; g [B,ldg] view with row stride ldg.
    Returns (scores [B,n_shifts,H,W], dstx [B,H,W,ldo])."""
    lib = _lib.load()
    dt = _act_dtype(x)
    _chk(x, "x", dt)
    if not g.is_cuda or g.dtype != torch.float32 or g.stride(-1) != 1:
        raise ValueError("g must be an fp32 device tensor with unit inner stride")
    b, h, w, ldx = x.shape
    c = channels if channels is not None else ldx
    n = len(shifts)
    sh = (ctypes.c_int * n)(*shifts)
    scores = _empty((b, n, h, w), device=x.device, dtype=torch.float32)
    dstx = _empty((b, h, w, ldo), device=x.device, dtype=dt)
    fn = lib.ccvpe_match_level_f32 if dt == torch.float32 else lib.ccvpe_match_level_bf16
    check(fn(_ptr(x), ldx, _ptr(g), g.stride(0), L, sh, n, n_max, n_tail, stride, window_offset, _ptr(scores), _ptr(dstx),
             ldo, b, h * w, c, _stream()), "ccvpe_match_level")
    return scores, dstx


def head_conv3x3(x, w, bias, cout, normalize):
    lib = _lib.load()
    dt = _act_dtype(x)
    _chk(x, "x", dt)
    for t, nm in ((w, "w"), (bias, "bias")):
        _chk(t, nm)
    b, h, wd, c = x.shape
    if c != 16:
        raise ValueError("head conv expects 16 input channels")
    out = _empty((b, cout, h, wd), device=x.device, dtype=torch.float32)
    fn = lib.ccvpe_head_conv3x3_f32 if dt == torch.float32 else lib.ccvpe_head_conv3x3_bf16
    check(fn(_ptr(x), _ptr(w), _ptr(bias), _ptr(out), b, h, wd, cout, int(bool(normalize)), _stream()),
          "ccvpe_head_conv3x3")
    return out


def cast_f32(x):
    """bf16 activation tensor -> fp32 (ccvpe_cast_bf16_f32); an fp32 tensor is returned as is."""
    if x.dtype == torch.float32:
        return x
    lib = _lib.load()
    _chk(x, "x", torch.bfloat16)
    out = _empty(x.shape, device=x.device, dtype=torch.float32)
    check(lib.ccvpe_cast_bf16_f32(_ptr(x), _ptr(out), x.numel(), _stream()), "ccvpe_cast_bf16_f32")
    return out


def softmax_rows(logits):
    lib = _lib.load()
    _chk(logits, "logits")
    rows, n = logits.shape
    out = _empty(logits.shape, device=logits.device, dtype=logits.dtype)
    check(lib.ccvpe_softmax_rows_f32(_ptr(logits), _ptr(out), rows, n, _stream()), "ccvpe_softmax_rows_f32")
    return out


# ----------------------------------------------------------------------------------------------
# train-mode pieces
# ----------------------------------------------------------------------------------------------
def stem_conv_raw(x_nchw, w, circular):
    """Stem convolution only (train mode: BN needs the batch statistics first)."""
    lib = _lib.load()
    _chk(x_nchw, "x")
    _chk(w, "w")
    b, c, h, wd = x_nchw.shape
    ho, wo = (h + 1 - 3) // 2 + 1, (wd + 1 - 3) // 2 + 1
    y = _empty((b, ho, wo, 32), device=x_nchw.device, dtype=torch.float32)
    check(lib.ccvpe_stem_conv_raw_f32(_ptr(x_nchw), _ptr(w), _ptr(y), b, h, wd, int(bool(circular)), _stream()),
          "ccvpe_stem_conv_raw_f32")
    return y


def dwconv_raw(x, w, k, stride, circular):
    lib = _lib.load()
    _chk(x, "x")
    _chk(w, "w")
    b, h, wd, c = x.shape
    tot = (k - 1) if stride == 1 else (k - 2)
    ho, wo = (h + tot - k) // stride + 1, (wd + tot - k) // stride + 1
    y = _empty((b, ho, wo, c), device=x.device, dtype=torch.float32)
    check(lib.ccvpe_dwconv_raw_f32(_ptr(x), _ptr(w), _ptr(y), b, h, wd, c, k, stride, int(bool(circular)), _stream()),
          "ccvpe_dwconv_raw_f32")
    return y


def bn_stats(x, run_mean=None, run_var=None, momentum=0.01):
    """Per-channel batch mean / biased variance of an NHWC tensor; updates the running statistics in place
    (momentum as torch: new = (1-m)*old + m*batch, unbiased variance)."""
    lib = _lib.load()
    _chk(x, "x")
    _chk(run_mean, "run_mean")
    _chk(run_var, "run_var")
    c = x.shape[-1]
    rows = x.numel() // c
    mean = _empty((c,), device=x.device, dtype=torch.float32)
    var = _empty((c,), device=x.device, dtype=torch.float32)
    scratch = _empty((lib.ccvpe_bn_stats_nblk(rows) * 3 * c,), device=x.device, dtype=torch.float32)
    check(lib.ccvpe_bn_stats_f32(_ptr(x), rows, c, _ptr(mean), _ptr(var), _ptr(run_mean), _ptr(run_var),
                                 float(momentum), _ptr(scratch), _stream()), "ccvpe_bn_stats_f32")
    return mean, var


def bn_act(x, mean, var, gamma, beta, eps, act, residual=None, dc_scale=None, want_se=False):
    """Normalise + activation (+ drop-connect scale, + residual); optionally SE squeeze partials."""
    lib = _lib.load()
    for t, nm in ((x, "x"), (mean, "mean"), (var, "var"), (gamma, "gamma"), (beta, "beta"), (residual, "residual"),
                  (dc_scale, "dc_scale")):
        _chk(t, nm)
    b, c = x.shape[0], x.shape[-1]
    rps = x.numel() // (b * c)
    y = _empty(x.shape, device=x.device, dtype=x.dtype)
    part = None
    if want_se:
        part = _empty((b, lib.ccvpe_bn_act_nblk(rps), c), device=x.device, dtype=torch.float32)
    check(lib.ccvpe_bn_act_f32(_ptr(x), _ptr(mean), _ptr(var), _ptr(gamma), _ptr(beta), float(eps), act, _ptr(residual),
                               _ptr(dc_scale), _ptr(y), _ptr(part), b, rps, c, _stream()), "ccvpe_bn_act_f32")
    return (y, part) if want_se else y


def eval_postprocess(heatmap, ori):
    """train_VIGOR.py:294-324 on the device: returns [B,6] = (y, x, cos, sin, angle_deg, prob)."""
    lib = _lib.load()
    _chk(heatmap, "heatmap")
    _chk(ori, "ori")
    b, _, h, w = heatmap.shape
    out = _empty((b, 6), device=heatmap.device, dtype=torch.float32)
    check(lib.ccvpe_eval_postprocess_f32(_ptr(heatmap), _ptr(ori), _ptr(out), b, h, w, _stream()),
          "ccvpe_eval_postprocess_f32")
    return out


def _loss_common(fn, name, tensors, extra):
    lib = _lib.load()
    for i, t in enumerate(tensors):
        _chk(t, "%s arg %d" % (name, i))
    b = tensors[0].shape[0]
    loss = _empty((1,), device=tensors[0].device, dtype=torch.float32)
    scratch = _empty((2 * b,), device=tensors[0].device, dtype=torch.float32)
    return lib, loss, scratch, b


def infonce_loss(scores, labels, temperature=0.1, want_den=False, want_rows=False):
    """losses.py:4 infoNCELoss — forward value only (device scalar).  want_den: also return the batch's label mass
    sum(labels > 1e-2) (the loss's denominator, losses.py:18); want_rows: also return the per-sample statistics
    [4*B + 4] the backward kernel reads (ccvpe_infonce_loss_bwd_f32)."""
    lib = _lib.load()
    _chk(scores, "infonce scores")
    _chk(labels, "infonce labels")
    b, n = scores.shape
    loss = _empty((1,), device=scores.device, dtype=torch.float32)
    rows = _empty((4 * b + 4,), device=scores.device, dtype=torch.float32)
    nfl = lib.ccvpe_infonce_scratch_floats(b, n)
    if nfl <= 0:
        raise _lib.CcvpeError("ccvpe_infonce_scratch_floats rejected [%d, %d]" % (b, n))
    scratch = _empty((nfl,), device=scores.device, dtype=torch.float32)
    check(lib.ccvpe_infonce_loss_f32(_ptr(scores), _ptr(labels), float(temperature), _ptr(loss), _ptr(rows), _ptr(scratch),
                                     b, n, _stream()), "ccvpe_infonce_loss_f32")
    out = (loss[0],)
    if want_den:
        out = out + (rows[4 * b],)
    if want_rows:
        out = out + (rows,)
    return out if len(out) > 1 else out[0]


def cross_entropy_loss(logits, labels):
    """losses.py:23 cross_entropy_loss — forward value only."""
    lib, loss, scratch, b = _loss_common(None, "ce", (logits, labels), None)
    check(lib.ccvpe_cross_entropy_loss_f32(_ptr(logits), _ptr(labels), _ptr(loss), _ptr(scratch), b,
                                           logits.shape[1], _stream()), "ccvpe_cross_entropy_loss_f32")
    return loss[0]


def orientation_loss(ori, gt_orientation, gt):
    """losses.py:28 orientation_loss — forward value only.  ori/gt_orientation [B,2,H,W], gt [B,1,H,W]."""
    lib, loss, scratch, b = _loss_common(None, "ori", (ori, gt_orientation, gt), None)
    hw = ori.shape[2] * ori.shape[3]
    check(lib.ccvpe_orientation_loss_f32(_ptr(ori), _ptr(gt_orientation), _ptr(gt), _ptr(loss), _ptr(scratch),
                                         b, hw, _stream()), "ccvpe_orientation_loss_f32")
    return loss[0]
