"""Host-side pieces of the whole-forward plan (no GPU): the lifetime arena and the loader's rejection of foreign bytes."""
import ctypes

from ccvpe_amd import _lib, plan


def test_arena_first_fit_reuse_and_coalescing():
    a = plan._Arena(1 << 20)
    o1, n1 = a.alloc(1000)
    o2, n2 = a.alloc(5000)
    o3, n3 = a.alloc(300)
    assert (o1, n1) == (0, 1024) and o2 == 1024 and n2 == 5120 and o3 == 6144 and n3 == 512
    assert a.peak == 6656
    a.release(o2, n2)
    o4, n4 = a.alloc(4000)                       # fits the hole the dead tensor left
    assert o4 == 1024 and n4 == 4096
    a.release(o1, n1)
    a.release(o4, n4)                            # coalesces with the block before it and the rest of the hole after it
    assert a.free[0] == (0, 6144)
    o5, _ = a.alloc(6000)
    assert o5 == 0 and a.peak == 6656
    a.release(o3, n3)
    a.release(o5, 6144)
    assert a.free == [(0, 1 << 20)]


def test_ctx_create_rejects_foreign_bytes():
    lib = _lib.load()
    ctx = ctypes.c_void_p()
    junk = ctypes.create_string_buffer(b"not a plan" * 20, 200)
    assert lib.ccvpe_ctx_create(ctypes.cast(junk, ctypes.c_void_p), 200, None, None, ctypes.byref(ctx)) != 0
    assert b"bad magic" in lib.ccvpe_last_error()
    assert lib.ccvpe_ctx_create(None, 0, None, None, ctypes.byref(ctx)) != 0
    assert lib.ccvpe_forward(None, None, None, None, None) != 0
    assert lib.ccvpe_ctx_destroy(None) == 0


def test_ctx_create_bounds_arithmetic_does_not_wrap():
    """A plan FILE is untrusted bytes (tools/plan_run.cpp): a relocation field of 0xFFFFFFFC, an output whose off + bytes
    wraps around 2^64 and a weights size beyond the file must be rejected by the parser — all before any device call."""
    import struct
    lib = _lib.load()
    abi = lib.ccvpe_abi_version()

    def plan_bytes(outputs=b"", n_out=0, blobs=b"", n_blobs=0, weights_bytes=0, ws=4096):
        head = b"CCVPLAN1" + struct.pack("<II", abi, 0) + struct.pack("<QQQQ", ws, weights_bytes, 64, 64)
        head += struct.pack("<IIII", 0, n_blobs, n_out, 0) + outputs + blobs
        return head + b"\0" * ((-len(head)) % 256) + b"\0" * 256

    def create(data):
        ctx = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(data, len(data))
        rc = lib.ccvpe_ctx_create(ctypes.cast(buf, ctypes.c_void_p), len(data), None, None, ctypes.byref(ctx))
        return rc, lib.ccvpe_last_error()

    reloc = struct.pack("<II", 16, 1) + struct.pack("<IIQ", 0xFFFFFFFC, plan.K_WORKSPACE, 0) + b"\0" * 16
    rc, err = create(plan_bytes(blobs=reloc, n_blobs=1))
    assert rc != 0 and b"relocation outside its blob" in err
    out = struct.pack("<QQII4Q4Q", 2 ** 64 - 8, 16, 0, 1, 4, 1, 1, 1, 1, 0, 0, 0)
    rc, err = create(plan_bytes(outputs=out, n_out=1))
    assert rc != 0 and b"output outside the workspace" in err
    rc, err = create(plan_bytes(weights_bytes=2 ** 64 - 64))
    assert rc != 0 and b"truncated weights" in err


def test_plan_launch_list_matches_the_c_registry():
    """Every entry point plan.py may record is one csrc/plan.hip replays (and the other way round)."""
    import os
    import re
    src = open(os.path.join(os.path.dirname(_lib.CSRC_DIR), "csrc", "plan.hip")).read()
    reg = set(re.findall(r"CCVPE_REG\((ccvpe_[a-z0-9_]+)\)", src))
    assert reg == set(plan.LAUNCHES)
    assert all(n in _lib.PROTOTYPES for n in plan.LAUNCHES + plan.QUERIES)
