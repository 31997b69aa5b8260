"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol that
include/ccvpe_hip.h declares (no compute calls without a GPU), the ctypes table covers the
header, and the drop-in modules keep the reference's state_dict layout."""
import os
import re

import pytest
import torch

from ccvpe_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "ccvpe_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ccvpe_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    if not os.path.isfile(_lib.LIB_PATH):
        _lib.build()
    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), "libccvpe_hip.so does not export %s" % n
    assert sorted(_lib.PROTOTYPES) == names, "ctypes table and header disagree"
    assert lib.ccvpe_abi_version() == 7


def test_ctypes_prototypes_match_header_signatures():
    """Arity and pointer/int/float kind of every ctypes prototype vs the C declaration."""
    import ctypes
    src = open(os.path.join(ROOT, "include", "ccvpe_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = re.findall(r"\b(?:int|const char\*)\s+(ccvpe_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src)
    assert len(decls) == len(_lib.PROTOTYPES)
    for name, params in decls:
        params = [q.strip() for q in params.split(",") if q.strip() and q.strip() != "void"]
        argtypes = _lib.PROTOTYPES[name][1]
        assert len(params) == len(argtypes), "%s: header has %d params, ctypes %d" % (name, len(params), len(argtypes))
        for q, a in zip(params, argtypes):
            is_ptr = "*" in q
            if is_ptr:
                assert a is ctypes.c_void_p or issubclass(a, ctypes._Pointer), (name, q, a)
            elif q.startswith("float"):
                assert a is ctypes.c_float, (name, q, a)
            elif q.startswith("double"):
                assert a is ctypes.c_double, (name, q, a)
            elif q.startswith("long"):
                assert a is ctypes.c_long, (name, q, a)
            else:
                assert a is ctypes.c_int, (name, q, a)


def test_conv_desc_matches_header_field_order():
    src = open(os.path.join(ROOT, "include", "ccvpe_hip.h")).read()
    body = src[src.index("typedef struct ccvpe_conv_desc"):src.index("} ccvpe_conv_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(None, 1)[1] if not decl.startswith("const") else decl.split(None, 2)[2]
        for n in names.split(","):
            fields.append(n.strip().lstrip("*").strip())
    assert fields == [f[0] for f in _lib.ConvDesc._fields_]


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libccvpe_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


@pytest.mark.parametrize("kind", ["vigor", "kitti", "oxford"])
def test_state_dict_layout_and_roundtrip(kind, synth_sd):
    from ccvpe_amd import models
    net = {"kitti": lambda: models.CVM_KITTI("cpu"), "oxford": lambda: models.CVM_OxfordRobotCar("cpu"),
           "vigor": lambda: models.CVM_VIGOR("cpu", True)}[kind]()
    spec = synth.state_dict_spec(kind)
    sd = net.state_dict()
    assert list(sd.keys()) == [k for k, _, _ in spec]
    assert len(sd) == 818
    for k, shape, _ in spec:
        assert tuple(sd[k].shape) == tuple(shape), k
    # dead _fc weights must round-trip
    assert "grd_efficientnet._fc.weight" in sd and sd["sat_efficientnet._fc.weight"].shape == (1000, 1280)
    src = synth_sd(kind, 3 if kind == "vigor" else 4)
    net.load_state_dict(src, strict=True)
    back = net.state_dict()
    for k in src:
        assert torch.equal(back[k], src[k]), k
    # parameters vs buffers as in the reference (BN statistics are buffers)
    pnames = {n for n, _ in net.named_parameters()}
    assert "conv6.0.weight" in pnames and "grd_efficientnet._bn0.running_mean" not in pnames


def test_vigor_and_ori_prior_share_checkpoints(synth_sd):
    from ccvpe_amd import models
    a = models.CVM_VIGOR("cpu", True)
    b = models.CVM_VIGOR_ori_prior("cpu", 72, True)
    b.load_state_dict(a.state_dict(), strict=True)
    assert list(a.state_dict().keys()) == list(b.state_dict().keys())


def test_cpu_inputs_are_rejected_not_emulated():
    from ccvpe_amd import models
    net = models.CVM_VIGOR_ori_prior("cpu", 0).eval()
    grd, sat = synth.synthetic_pair(1, "vigor", 1)
    with pytest.raises(RuntimeError, match="MI355X"):
        net(grd, sat)


def test_c_abi_rejects_bad_arguments_without_a_gpu():
    """Every entry point validates before launching: bad shapes / alignment come back as CCVPE_EINVAL
    with a message (no exception crosses the ABI, nothing is enqueued), so this runs on CPU."""
    import ctypes
    lib = _lib.load()
    EINVAL = -1
    assert lib.ccvpe_conv_igemm_f32(None, None) == EINVAL
    assert b"null desc" in lib.ccvpe_last_error()
    d = _lib.ConvDesc()
    d.src0, d.w, d.dst = 256, 512, 1024            # fake 16-byte aligned "pointers": never dereferenced
    d.c0, d.ld0, d.batch, d.in_h, d.in_w = 12, 12, 1, 4, 4       # c0 not a multiple of 8
    d.kh = d.kw = d.stride = 1
    d.n, d.kpad, d.ldd = 16, 16, 16
    assert lib.ccvpe_conv_igemm_f32(ctypes.byref(d), None) == EINVAL
    assert b"multiples of 8" in lib.ccvpe_last_error()
    d.c0, d.ld0, d.src0 = 16, 16, 260                              # misaligned pointer
    assert lib.ccvpe_conv_igemm_f32(ctypes.byref(d), None) == EINVAL
    assert b"aligned" in lib.ccvpe_last_error()
    d.src0, d.kpad = 256, 8                                        # kpad too small / not a stage multiple
    assert lib.ccvpe_conv_igemm_bf16(ctypes.byref(d), 0, None) == EINVAL
    assert lib.ccvpe_upconv3x3_f32(None, None) == EINVAL
    assert lib.ccvpe_dwconv_f32(256, 256, 256, 256, 256, 256, 1, 8, 8, 6, 3, 1, 0, None) == EINVAL      # C % 4
    assert lib.ccvpe_dwconv_f32(256, 256, 256, 256, 256, 256, 1, 8, 8, 8, 4, 1, 0, None) == EINVAL      # k = 4
    sh = (ctypes.c_int * 2)(0, 1)
    assert lib.ccvpe_match_level_f32(256, 40, 256, 48, 48, sh, 2, 2, 0, 2, 0, 256, 256, 48, 1, 64, 40, None) == EINVAL
    assert b"bad L" in lib.ccvpe_last_error()                                                           # window wider than C
    assert lib.ccvpe_match_level_f32(256, 40, 256, 40, 40, sh, 99, 2, 0, 2, 0, 256, 256, 48, 1, 64, 40, None) == EINVAL
    assert lib.ccvpe_head_conv3x3_f32(256, 256, 256, 256, 1, 8, 8, 3, 0, None) == EINVAL               # cout = 3
    assert lib.ccvpe_softmax_rows_f32(256, 256, 0, 8, None) == EINVAL
    assert lib.ccvpe_mbconv_front_nblk(8, 8, 16, 96, 4, 1) == EINVAL
    assert lib.ccvpe_mbconv_front_nblk(16, 16, 192, 1152, 3, 1) == 1        # late block: csrc/mbconv_plane.hip, the plane is one band
    assert lib.ccvpe_mbconv_front_nblk(32, 32, 112, 672, 5, 1) == 2         # 32 x 32 plane: two bands of 16 rows
    assert lib.ccvpe_mbconv_front_nblk(16, 16, 96, 576, 3, 1) == 0          # valid but unfused shape (Cin not instantiated)
    assert lib.ccvpe_dwconv_nblk(32, 32, 672, 1) == 2 and lib.ccvpe_dwconv_nblk(7, 9, 480, 1) == 1
    assert lib.ccvpe_mbconv_front_nblk(64, 64, 40, 240, 5, 1) == 32         # squeeze-partial rows = 8 x 16 output tiles: 8 x 4
    assert lib.ccvpe_mbconv_front_nblk(256, 256, 16, 96, 3, 2) == 128       # block 1: 128 x 128 outputs in 8 x 16 tiles
    assert lib.ccvpe_mbconv_front_nblk(64, 64, 40, 240, 5, 2) == 0          # k5 s2 with Cin > 32: not instantiated -> unfused


def test_planning_entry_points_without_a_gpu():
    """The size / split planners are pure host code: callable on a box without a GPU."""
    import ctypes
    EINVAL = -1
    lib = _lib.load()
    d = _lib.ConvDesc()
    d.src0 = d.w = d.dst = 256
    d.c0, d.ld0, d.batch, d.in_h, d.in_w = 1344, 1344, 1, 16, 16
    d.kh = d.kw = 3
    d.stride, d.pad, d.n, d.kpad, d.ldd = 1, 1, 640, 9 * 1344, 640
    want = lib.ccvpe_conv_igemm_splitk_floats(ctypes.byref(d), 0)          # B = 1, level-6 3x3: 8 tiles, 756 K stages
    assert want > 0 and want % (256 * 640) == 0 and 2 <= want // (256 * 640) <= 32
    d.batch = 64                                                            # B = 64: 512 tiles -> one pass
    assert lib.ccvpe_conv_igemm_splitk_floats(ctypes.byref(d), 0) == 0
    # bf16 3x3 with >= 128 halo tiles stays UN-split (the LDS-DMA 3x3 kernel wins there); fp32 and smaller batches still split
    d.batch, d.kpad = 32, 9 * 1344
    assert lib.ccvpe_conv_igemm_splitk_floats(ctypes.byref(d), 1) == 0          # 32 x 2 x 1 x 4 = 256 tiles
    assert lib.ccvpe_conv_igemm_splitk_floats(ctypes.byref(d), 0) > 0           # fp32: 256 gather tiles < 320 -> split
    d.batch = 8
    assert lib.ccvpe_conv_igemm_splitk_floats(ctypes.byref(d), 1) > 0           # 64 tiles: split
    d.batch, d.c0 = 1, 1343
    assert lib.ccvpe_conv_igemm_splitk_floats(ctypes.byref(d), 0) == EINVAL
    assert lib.ccvpe_conv_igemm_splitk_f32(ctypes.byref(d), None, None) == EINVAL      # no scratch
    # matching backward: channel slices only when there are few pixel workgroups
    assert lib.ccvpe_match_bwd_nblk(64, 8, 2048) > lib.ccvpe_match_bwd_nblk(64, 8, 64) >= 1
    assert lib.ccvpe_match_bwd_nblk(65536, 64, 32) == 256
    # BatchNorm statistics: partial rows + one group row per 64 partials, bounded for huge tensors
    assert lib.ccvpe_bn_stats_nblk(256) == 8                               # small tensors: 32-row workgroups
    assert 4096 <= lib.ccvpe_bn_stats_nblk(64 * 512 * 512) <= 4096 + 64 + 1
    assert lib.ccvpe_dwconv_wgrad_nblk(64, 64, 3, 1) == 64 and lib.ccvpe_dwconv_wgrad_nblk(256, 256, 3, 2) == 64
    assert lib.ccvpe_dwconv_wgrad_nblk(16, 16, 5, 1) == 4 and lib.ccvpe_dwconv_wgrad_nblk(32, 32, 5, 2) == 4      # planes <= 1 024 px: dw_wgrad_rows_kernel
    assert lib.ccvpe_conv_wgrad_scratch_floats(2, 16, 16, 3, 3, 1, 1, 1344, 640) > 0
    assert lib.ccvpe_adam_chunk_elems() == 4096 and lib.ccvpe_train_targets_nblk(512, 512) == 256


def test_round5_queries_without_a_gpu():
    """Host-only: the fused-stem planner and the pointwise-ring route (csrc/stem_dw.hip, csrc/conv_pw2_impl.h: pw2_supported)."""
    import ctypes
    lib = _lib.load()
    assert lib.ccvpe_stem_dw_nblk(512, 512, 0) == 32 * 8                     # aerial: 256 x 256 outputs in 8 x 32 tiles
    assert lib.ccvpe_stem_dw_nblk(320, 640, 1) == 20 * 10                    # ground, circular padding
    assert lib.ccvpe_stem_dw_nblk(154, 231, 0) == 10 * 4                     # Oxford: 77 x 115 outputs, ragged tiles
    assert lib.ccvpe_stem_dw_nblk(154, 231, 1) == 0                          # circular padding needs an even width: two launches
    assert lib.ccvpe_stem_dw_nblk(2, 2, 0) == 0
    d = _lib.ConvDesc()
    d.src0 = d.w = d.dst = d.scale = d.shift = 256
    d.c0, d.ld0, d.batch, d.in_h, d.in_w = 112, 112, 2, 32, 32
    d.kh = d.kw = 1
    d.stride, d.pad, d.n, d.kpad, d.ldd, d.act = 1, 0, 672, 112, 672, 2     # expand conv of blocks 9-11: BN + swish
    fam = lambda r: r & 0xff
    for bf16 in (0, 1):
        d.kpad = 128 if bf16 else 112                                        # packed K: whole 64-byte pieces of the element type
        assert fam(lib.ccvpe_conv_igemm_route(ctypes.byref(d), bf16, 0)) == 4, "pw_ring expected"
    d.kpad = 112
    prev = lib.ccvpe_set_pw_ring_kernels(0)
    try:
        assert fam(lib.ccvpe_conv_igemm_route(ctypes.byref(d), 0, 0)) == 1  # the switch brings pw_gemm_kernel back
    finally:
        lib.ccvpe_set_pw_ring_kernels(prev)
    d.batch, d.in_h, d.in_w = 3, 19, 23                                      # M = 1311: not a multiple of the 128-row tile
    assert fam(lib.ccvpe_conv_igemm_route(ctypes.byref(d), 0, 0)) == 1
    d.batch, d.in_h, d.in_w, d.gate = 64, 20, 40, 256                        # ground encoder: a gated tile would span two samples
    d.c0, d.ld0, d.kpad, d.n, d.ldd, d.act = 672, 672, 672, 112, 112, 0
    assert fam(lib.ccvpe_conv_igemm_route(ctypes.byref(d), 0, 0)) == 1
    d.in_h, d.in_w = 32, 32                                                  # aerial: 1 024 pixels per sample = 8 tiles
    assert fam(lib.ccvpe_conv_igemm_route(ctypes.byref(d), 0, 0)) == 4
    assert fam(lib.ccvpe_conv_igemm_route(ctypes.byref(d), 1, 0)) == 4
    assert fam(lib.ccvpe_conv_igemm_route(ctypes.byref(d), 1, 1)) == 1      # bf16 storage writing fp32: pw_gemm_kernel


def test_pw_ring_is_never_routed_with_fewer_than_three_k_stages():
    """csrc/conv_pw2_impl.h: the ring's DMA requests run three K stages ahead of the matrix instructions and the per-tile vectors
    (two LDS copies by tile parity) of tile t + 2 must not be requested before every wave is past tile t's epilogue — a hazard that
    is excluded by ROUTING (pw2_supported), so the route is what this pins: a bf16 layer with K = 64 is exactly two 32-channel stages
    and must stay on pw_gemm_kernel, K = 96 (three stages) may take the ring; fp32 needs K >= 64 = four 16-channel stages anyway."""
    import ctypes
    lib = _lib.load()
    fam = lambda r: r & 0xff
    d = _lib.ConvDesc()
    d.src0 = d.w = d.dst = d.scale = d.shift = 256
    d.batch, d.in_h, d.in_w = 4, 32, 32
    d.kh = d.kw = 1
    d.stride, d.pad, d.n, d.ldd, d.act = 1, 0, 384, 384, 2                    # an expand-shaped layer: BN + swish, N = 384
    for c0, want in ((64, 1), (96, 4), (128, 4)):                             # 2, 3, 4 bf16 K stages
        d.c0, d.ld0, d.kpad = c0, c0, c0
        r = lib.ccvpe_conv_igemm_route(ctypes.byref(d), 1, 0)
        assert fam(r) == want, "bf16 K = %d: route %d, expected %d" % (c0, fam(r), want)
    d.c0, d.ld0, d.kpad = 64, 64, 64
    assert fam(lib.ccvpe_conv_igemm_route(ctypes.byref(d), 0, 0)) == 4        # fp32 K = 64: four stages
    d.c0, d.ld0, d.kpad = 48, 48, 48
    assert fam(lib.ccvpe_conv_igemm_route(ctypes.byref(d), 0, 0)) == 1        # fp32 K = 48 (three stages): below the ring's minimum K
