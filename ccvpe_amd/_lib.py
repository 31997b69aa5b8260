"""ctypes binding of libccvpe_hip.so (C ABI declared in include/ccvpe_hip.h).

There is NO fallback: if the library is missing or a symbol is absent this raises, and every
operator in ccvpe_amd.ops goes through it.  `import torch` must precede the load so that the
library binds to the HIP runtime torch already mapped (same SONAME libamdhip64.so.7) — streams
and device pointers are then shared with torch.
"""
import ctypes
import os
import subprocess

import torch  # noqa: F401  (must be imported before the HIP library is mapped)

_HERE = os.path.dirname(os.path.abspath(__file__))
# CCVPE_LIB: load another build of the same C ABI (A/B runs of compiler flags); the default is the in-tree library
LIB_PATH = os.environ.get("CCVPE_LIB") or os.path.join(_HERE, "libccvpe_hip.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

c_float_p = ctypes.c_void_p   # device pointers travel as integers
c_int = ctypes.c_int
c_float = ctypes.c_float
c_void_p = ctypes.c_void_p


class ConvDesc(ctypes.Structure):
    """struct ccvpe_conv_desc (include/ccvpe_hip.h)."""
    _fields_ = [
        ("src0", c_void_p), ("src1", c_void_p), ("gate", c_void_p), ("w", c_void_p),
        ("scale", c_void_p), ("shift", c_void_p), ("residual", c_void_p), ("dst", c_void_p),
        ("c0", c_int), ("ld0", c_int), ("c1", c_int), ("ld1", c_int),
        ("batch", c_int), ("in_h", c_int), ("in_w", c_int),
        ("kh", c_int), ("kw", c_int), ("stride", c_int), ("pad", c_int),
        ("n", c_int), ("kpad", c_int), ("ldd", c_int), ("ldres", c_int),
        ("act", c_int), ("out_mode", c_int),
    ]


class UpconvDesc(ctypes.Structure):
    """struct ccvpe_upconv_desc (include/ccvpe_hip.h)."""
    _fields_ = [
        ("src0", c_void_p), ("src1", c_void_p), ("w", c_void_p), ("shift9", c_void_p), ("dst", c_void_p),
        ("c0", c_int), ("ld0", c_int), ("c1", c_int), ("ld1", c_int),
        ("batch", c_int), ("h1", c_int), ("w1", c_int),
        ("n", c_int), ("kpad", c_int), ("ldd", c_int), ("act", c_int),
    ]


class TailDesc(ctypes.Structure):
    """struct ccvpe_tail_desc (include/ccvpe_hip.h)."""
    _fields_ = [
        ("x", c_void_p), ("w", c_void_p), ("shift9", c_void_p), ("w2", c_void_p), ("b2", c_void_p), ("out", c_void_p),
        ("batch", c_int), ("h1", c_int), ("w1", c_int),
        ("c0", c_int), ("ld0", c_int), ("kpad", c_int),
        ("cout", c_int), ("normalize", c_int), ("split", c_int), ("softmax_partial", c_void_p),
    ]


# name -> (restype, argtypes); must list EVERY symbol include/ccvpe_hip.h declares
# (tests/test_abi.py parses the header and checks this table and the .so against it).
PROTOTYPES = {
    "ccvpe_last_error": (ctypes.c_char_p, []),
    "ccvpe_abi_version": (c_int, []),
    "ccvpe_multi_copy_f32": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), ctypes.POINTER(c_int), c_int, c_void_p]),
    "ccvpe_gather_repack_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ccvpe_gather_repack_chunk": (c_int, []),
    "ccvpe_ctx_create": (c_int, [c_void_p, ctypes.c_longlong, c_void_p, c_void_p, ctypes.POINTER(c_void_p)]),
    "ccvpe_ctx_destroy": (c_int, [c_void_p]),
    "ccvpe_ctx_info": (c_int, [c_void_p] + [ctypes.POINTER(ctypes.c_longlong)] * 4 + [ctypes.POINTER(c_int)] * 2),
    "ccvpe_ctx_output": (c_int, [c_void_p, c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(ctypes.c_longlong),
                                  ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong)]),
    "ccvpe_forward": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_void_p), c_void_p]),
    "ccvpe_tail512_partials": (c_int, [ctypes.POINTER(TailDesc), c_int]),
    "ccvpe_softmax_apply_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "ccvpe_tail512_f32": (c_int, [ctypes.POINTER(TailDesc), c_void_p]),
    "ccvpe_tail512_bf16": (c_int, [ctypes.POINTER(TailDesc), c_void_p]),
    "ccvpe_conv_igemm_f32": (c_int, [ctypes.POINTER(ConvDesc), c_void_p]),
    "ccvpe_conv_igemm_splitk_floats": (c_int, [ctypes.POINTER(ConvDesc), c_int]),
    "ccvpe_conv_igemm_route": (c_int, [ctypes.POINTER(ConvDesc), c_int, c_int]),
    "ccvpe_set_narrow_kernels": (c_int, [c_int]),
    "ccvpe_set_pw_ring_kernels": (c_int, [c_int]),
    "ccvpe_set_mbconv_plane_kernels": (c_int, [c_int]),
    "ccvpe_mbconv_front_route": (c_int, [c_int] * 8),
    "ccvpe_set_match_mfma": (c_int, [c_int]),
    "ccvpe_set_pwn_kernels": (c_int, [c_int]),
    "ccvpe_conv3x3_match1_ok": (c_int, [ctypes.POINTER(ConvDesc), c_int, c_int]),
    "ccvpe_conv3x3_match1_bf16": (c_int, [ctypes.POINTER(ConvDesc), c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ccvpe_conv_igemm_splitk_f32": (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p]),
    "ccvpe_conv_igemm_splitk_bf16": (c_int, [ctypes.POINTER(ConvDesc), c_int, c_void_p, c_void_p]),
    "ccvpe_upconv3x3_f32": (c_int, [ctypes.POINTER(UpconvDesc), c_void_p]),
    "ccvpe_upconv3x3_bf16": (c_int, [ctypes.POINTER(UpconvDesc), c_void_p]),
    "ccvpe_upconv3x3_route": (c_int, [ctypes.POINTER(UpconvDesc), c_int]),
    "ccvpe_stem_conv_f32": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_void_p]),
    "ccvpe_dwconv_nblk": (c_int, [c_int] * 4),
    "ccvpe_dwconv_f32": (c_int, [c_void_p] * 6 + [c_int] * 7 + [c_void_p]),
    "ccvpe_mbconv_front_nblk": (c_int, [c_int] * 6),
    "ccvpe_stem_dw_nblk": (c_int, [c_int] * 3),
    "ccvpe_stem_dw_f32": (c_int, [c_void_p] * 9 + [c_int] * 4 + [c_void_p]),
    "ccvpe_stem_dw_bf16": (c_int, [c_void_p] * 9 + [c_int] * 4 + [c_void_p]),
    "ccvpe_mbconv_front_f32": (c_int, [c_void_p, c_void_p, c_int] + [c_void_p] * 7 + [c_int] * 8 + [c_void_p]),
    "ccvpe_se_gate_f32": (c_int, [c_void_p, c_int, c_float] + [c_void_p] * 5 + [c_int] * 3 + [c_void_p]),
    "ccvpe_ground_descriptor_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, ctypes.POINTER(c_int),
                                            c_void_p, c_int, c_int, c_int, c_void_p]),
    "ccvpe_match_level_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, ctypes.POINTER(c_int),
                                      c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                                      c_int, c_int, c_void_p]),
    "ccvpe_head_conv3x3_f32": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p]),
    "ccvpe_cast_bf16_f32": (c_int, [c_void_p, c_void_p, ctypes.c_long, c_void_p]),
    "ccvpe_softmax_rows_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ccvpe_eval_postprocess_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ccvpe_infonce_scratch_floats": (c_int, [c_int, c_int]),
    "ccvpe_infonce_loss_f32": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ccvpe_cross_entropy_loss_f32": (c_int, [c_void_p] * 4 + [c_int, c_int, c_void_p]),
    "ccvpe_orientation_loss_f32": (c_int, [c_void_p] * 5 + [c_int, c_int, c_void_p]),
    "ccvpe_stem_conv_raw_f32": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p]),
    "ccvpe_dwconv_raw_f32": (c_int, [c_void_p] * 3 + [c_int] * 7 + [c_void_p]),
    "ccvpe_bn_stats_nblk": (c_int, [c_int]),
    "ccvpe_bn_stats_f32": (c_int, [c_void_p, c_int, c_int] + [c_void_p] * 4 + [c_float, c_void_p, c_void_p]),
    "ccvpe_bn_act_nblk": (c_int, [c_int]),
    "ccvpe_bn_act_f32": (c_int, [c_void_p] * 5 + [c_float, c_int] + [c_void_p] * 4 + [c_int] * 3 + [c_void_p]),
    "ccvpe_conv_wgrad_tile": (c_int, [c_int, c_int]),
    "ccvpe_conv_wgrad_scratch_floats": (c_int, [c_int] * 9),
    "ccvpe_conv_wgrad_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p,
                                     c_void_p] + [c_int] * 8 + [c_void_p]),
    "ccvpe_conv_wgrad_gated_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                           c_int, c_void_p]),
    "ccvpe_conv_wgrad_bias_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p,
                                          c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]),
    "ccvpe_colsum_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "ccvpe_bn_bwd_nblk": (c_int, [c_int]),
    "ccvpe_bn_act_bwd_f32": (c_int, [c_void_p] * 9 + [c_float, c_int] + [c_void_p] * 4 + [c_int] * 3 + [c_void_p]),
    "ccvpe_se_bn_bwd_reduce_f32": (c_int, [c_void_p] * 6 + [c_float, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ccvpe_se_bn_bwd_apply_f32": (c_int, [c_void_p] * 8 + [c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                          c_void_p]),
    "ccvpe_se_dgate_f32": (c_int, [c_void_p] * 6 + [c_float, c_int, c_void_p] + [c_int] * 3 + [c_void_p]),
    "ccvpe_se_bwd_f32": (c_int, [c_void_p, c_int, c_float, c_void_p, c_int] + [c_void_p] * 10 + [c_int] * 3 + [c_void_p]),
    "ccvpe_dwconv_dgrad_f32": (c_int, [c_void_p] * 3 + [c_int] * 7 + [c_void_p]),
    "ccvpe_dwconv_wgrad_nblk": (c_int, [c_int] * 4),
    "ccvpe_dwconv_wgrad_f32": (c_int, [c_void_p] * 4 + [c_int] * 7 + [c_void_p]),
    "ccvpe_relu_bwd_f32": (c_int, [c_void_p] * 3 + [c_int, c_void_p]),
    "ccvpe_gate_mul_f32": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_void_p]),
    "ccvpe_softmax_bwd_f32": (c_int, [c_void_p] * 4 + [c_int, c_int, c_void_p]),
    "ccvpe_l2norm2_bwd_f32": (c_int, [c_void_p] * 3 + [c_int, c_int, c_void_p]),
    "ccvpe_head_conv3x3_bwd_f32": (c_int, [c_void_p] * 7 + [c_int] * 5 + [c_void_p]),
    "ccvpe_ground_descriptor_bwd_f32": (c_int, [c_void_p, c_int, c_void_p, ctypes.POINTER(c_int)] + [c_void_p] * 4 +
                                        [c_int] * 3 + [c_void_p]),
    "ccvpe_add_cols_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ccvpe_stem_wgrad_nblk": (c_int, [c_int] * 3),
    "ccvpe_stem_conv_wgrad_f32": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    "ccvpe_match_bwd_nblk": (c_int, [c_int, c_int, c_int]),
    "ccvpe_match_level_bwd_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, ctypes.POINTER(c_int), c_int, c_int,
                                          c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                          c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ccvpe_infonce_loss_bwd_f32": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ccvpe_cross_entropy_loss_bwd_f32": (c_int, [c_void_p] * 4 + [c_int, c_int, c_void_p]),
    "ccvpe_orientation_loss_bwd_f32": (c_int, [c_void_p] * 5 + [c_int, c_int, c_void_p]),
    "ccvpe_train_targets_nblk": (c_int, [c_int, c_int]),
    "ccvpe_train_targets_f32": (c_int, [c_void_p, c_void_p, c_int, c_float] + [c_void_p] * 10 + [c_int] * 3 + [c_void_p]),
    "ccvpe_train_targets_ordered_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float] + [c_void_p] * 10 + [c_int] * 3 + [c_void_p]),
    "ccvpe_adam_chunk_elems": (c_int, []),
    "ccvpe_adam_hyper_floats": (c_int, []),
    "ccvpe_adam_step_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p]),
    "ccvpe_preprocess_u8_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                        c_void_p, c_int, c_int, c_int, c_int, ctypes.POINTER(c_float), ctypes.POINTER(c_float),
                                        c_void_p]),
    "ccvpe_conv_igemm_bf16": (c_int, [ctypes.POINTER(ConvDesc), c_int, c_void_p]),
    "ccvpe_stem_conv_bf16": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_void_p]),
    "ccvpe_dwconv_bf16": (c_int, [c_void_p] * 6 + [c_int] * 7 + [c_void_p]),
    "ccvpe_mbconv_front_bf16": (c_int, [c_void_p, c_void_p, c_int] + [c_void_p] * 7 + [c_int] * 8 + [c_void_p]),
    "ccvpe_match_level_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, ctypes.POINTER(c_int),
                                       c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                                       c_int, c_int, c_void_p]),
    "ccvpe_head_conv3x3_bf16": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p]),
}

_lib = None
# bumped by anything that rewrites parameters behind torch's back (ccvpe_amd.optim.Adam updates them in place with a HIP
# kernel, which does not touch torch's version counters); the models' packed-weight cache keys on it
weights_epoch = 0


def build(verbose=False):
    """Compile libccvpe_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC_DIR, "-j8"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise RuntimeError("building libccvpe_hip.so failed (see output above)")
    return LIB_PATH


def load():
    """Map the library and attach prototypes.  Raises if it is missing — no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            "libccvpe_hip.so not found at %s — run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C ccvpe_amd/csrc`).  ccvpe_amd has no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lax = bool(os.environ.get("CCVPE_LIB")) and os.environ.get("CCVPE_LIB_ALLOW_MISSING") == "1"   # tools/ A/B runs against an older build
    for name, (res, args) in PROTOTYPES.items():
        if lax and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)       # AttributeError if the symbol is missing: loud by design
        fn.restype = res
        fn.argtypes = args
    if os.environ.get("CCVPE_NARROW") == "0" and hasattr(lib, "ccvpe_set_narrow_kernels"):
        lib.ccvpe_set_narrow_kernels(0)           # A/B runs (tools/gpu/*.sh): the tiled kernels for the narrow decoder levels
    if os.environ.get("CCVPE_PW_RING") == "0" and hasattr(lib, "ccvpe_set_pw_ring_kernels"):
        lib.ccvpe_set_pw_ring_kernels(0)          # A/B runs: pw_gemm_kernel for every pointwise layer
    if os.environ.get("CCVPE_MATCH_MFMA") == "0" and hasattr(lib, "ccvpe_set_match_mfma"):
        lib.ccvpe_set_match_mfma(0)               # A/B runs: the vector-ALU matching kernel for every configuration
    if os.environ.get("CCVPE_PWN") == "0" and hasattr(lib, "ccvpe_set_pwn_kernels"):
        lib.ccvpe_set_pwn_kernels(0)              # A/B runs: the generic kernel for the narrow projections
    if os.environ.get("CCVPE_MBPLANE") is not None and hasattr(lib, "ccvpe_set_mbconv_plane_kernels"):
        lib.ccvpe_set_mbconv_plane_kernels(int(os.environ["CCVPE_MBPLANE"]))   # A/B runs: 0 = pointwise GEMM + dwconv_plane_kernel
    _lib = lib
    return lib


class CcvpeError(RuntimeError):
    pass


def check(status, what):
    if status != 0:
        msg = load().ccvpe_last_error()
        raise CcvpeError("%s failed (%d): %s" % (what, status, msg.decode() if msg else "?"))
