"""Whole-forward plans: the eval forward of a ccvpe_amd model as ONE C call (include/ccvpe_hip.h: ccvpe_ctx_create /
ccvpe_forward, csrc/plan.hip) — SURVEY.md section 8(b)'s "opaque ccvpe_ctx per (model kind, B, grd H x W, N_rot set, dtype)".

`record(net, grd, sat)` runs the Python forward (models.py:150-343 / :448-652 / :752-950 as ccvpe_amd/models.py lays it out)
once against a recording allocator and a recording view of the library (the forward keeps its two HIP streams: the ground
encoder — and in bf16 storage the orientation decoder — run on a side stream; calls carry their stream, stream waits are
recorded as fork / join marks and the ctx replays them on the caller's stream + one stream of its own):
  * every intermediate and output tensor comes from ONE workspace, placed by lifetime (a tensor's bytes are handed out again
    once nothing references it: program order is lifetime order within a stream; main- and side-stream tensors live in two
    regions, and bytes released while the OTHER stream may still be reading them are quarantined until the next join);
  * every library call is logged with its arguments; each pointer becomes (region, offset) with region = the workspace, the
    caller's grd / sat images, or the WEIGHTS blob the recorder assembles from whatever packed-weight tensors the calls touch.
The result serialises to bytes (`Plan.to_bytes()`, `Plan.save(path)`): a caller without Python loads them (tools/plan_run.cpp).
`PlannedForward(net, grd, sat)` is the Python binding: forward = one ctypes call, outputs are views of the workspace
(valid until the next call), bit-identical to the eager forward (tests/test_plan_gpu.py)."""
import ctypes
import struct
import weakref

import torch

from . import _lib, ops

K_INT, K_FLT, K_NULL, K_WEIGHTS, K_WORKSPACE, K_GRD, K_SAT, K_STREAM, K_BLOB = range(9)
ALIGN = 256

# entry points a plan may contain (csrc/plan.hip's registry); everything else the forward calls is a host-side query
LAUNCHES = ("ccvpe_conv_igemm_f32", "ccvpe_conv_igemm_bf16", "ccvpe_conv_igemm_splitk_f32", "ccvpe_conv_igemm_splitk_bf16",
            "ccvpe_conv3x3_match1_bf16",
            "ccvpe_upconv3x3_f32", "ccvpe_upconv3x3_bf16", "ccvpe_tail512_f32", "ccvpe_tail512_bf16", "ccvpe_stem_conv_f32",
            "ccvpe_stem_conv_bf16", "ccvpe_stem_dw_f32", "ccvpe_stem_dw_bf16", "ccvpe_dwconv_f32", "ccvpe_dwconv_bf16", "ccvpe_mbconv_front_f32", "ccvpe_mbconv_front_bf16",
            "ccvpe_se_gate_f32", "ccvpe_ground_descriptor_f32", "ccvpe_match_level_f32", "ccvpe_match_level_bf16",
            "ccvpe_head_conv3x3_f32", "ccvpe_head_conv3x3_bf16", "ccvpe_softmax_rows_f32", "ccvpe_softmax_apply_f32", "ccvpe_cast_bf16_f32",
            "ccvpe_eval_postprocess_f32")
QUERIES = ("ccvpe_tail512_partials", "ccvpe_conv3x3_match1_ok", "ccvpe_conv_igemm_splitk_floats", "ccvpe_conv_igemm_route", "ccvpe_upconv3x3_route", "ccvpe_dwconv_nblk", "ccvpe_mbconv_front_nblk", "ccvpe_mbconv_front_route", "ccvpe_stem_dw_nblk",
           "ccvpe_last_error", "ccvpe_abi_version")


class _Arena(object):
    """First-fit allocator over [0, size): 256-byte aligned blocks, coalescing free list (a few hundred calls per forward)."""

    def __init__(self, size):
        self.size, self.free, self.peak = size, [(0, size)], 0

    def alloc(self, nbytes):
        n = max(ALIGN, (nbytes + ALIGN - 1) // ALIGN * ALIGN)
        for i, (off, sz) in enumerate(self.free):
            if sz >= n:
                if sz == n:
                    del self.free[i]
                else:
                    self.free[i] = (off + n, sz - n)
                self.peak = max(self.peak, off + n)
                return off, n
        raise MemoryError("plan arena exhausted")

    def release(self, off, n):
        fr = self.free
        lo, hi = 0, len(fr)
        while lo < hi:
            mid = (lo + hi) // 2
            if fr[mid][0] < off:
                lo = mid + 1
            else:
                hi = mid
        fr.insert(lo, (off, n))
        if lo + 1 < len(fr) and fr[lo][0] + fr[lo][1] == fr[lo + 1][0]:
            fr[lo] = (fr[lo][0], fr[lo][1] + fr[lo + 1][1])
            del fr[lo + 1]
        if lo > 0 and fr[lo - 1][0] + fr[lo - 1][1] == fr[lo][0]:
            fr[lo - 1] = (fr[lo - 1][0], fr[lo - 1][1] + fr[lo][1])
            del fr[lo]


def _tensor_ranges(obj, out=None, seen=None):
    """[start, end) byte ranges of every tensor reachable from `obj` (the packed-weights object of a model: nested plain
    objects / dicts / lists / tuples of tensors)."""
    out = [] if out is None else out
    seen = set() if seen is None else seen
    if id(obj) in seen:
        return out
    seen.add(id(obj))
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda and obj.numel():
            st = obj.untyped_storage()
            out.append((st.data_ptr(), st.data_ptr() + st.nbytes()))
    elif isinstance(obj, dict):
        for v in obj.values():
            _tensor_ranges(v, out, seen)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _tensor_ranges(v, out, seen)
    elif hasattr(obj, "__dict__"):
        _tensor_ranges(vars(obj), out, seen)
    return out


class _Recorder(object):
    def __init__(self, device, arena_bytes, grd, sat, side_fraction=0.45, constants=None):
        self.ws = torch.empty((arena_bytes,), dtype=torch.uint8, device=device)
        self.base = self.ws.data_ptr()
        # two regions: [0, split) for tensors allocated under the main stream, [split, size) for the side stream
        self.split = int(arena_bytes * (1.0 - side_fraction)) // ALIGN * ALIGN
        self.arenas = (_Arena(self.split), _Arena(arena_bytes - self.split))
        self.size = arena_bytes
        self.grd, self.sat = grd, sat
        self.calls, self.blobs = [], []
        self.weights, self.weights_bytes = {}, 0         # data_ptr -> (offset, tensor)
        self.seen = {}                                   # data_ptr -> tensor, for every NON-workspace tensor handed to ops._ptr()
        # byte ranges that may be frozen into the weights blob: the packed weights as they existed BEFORE the recording forward.
        # Anything else outside the workspace / inputs is an input-dependent intermediate that bypassed ops._empty (torch.empty,
        # torch.cat, .contiguous() inside the forward): freezing it would replay the recording inputs' values silently.
        self.constants = sorted(constants) if constants is not None else None
        self.main = torch.cuda.current_stream().cuda_stream
        self.quarantine = []                             # (region, off, size): released under the other stream, free at the next join
        self.n_waits = 0

    def sid(self):
        return 0 if torch.cuda.current_stream().cuda_stream == self.main else 1

    @property
    def peak(self):
        return max(self.arenas[0].peak, self.split + self.arenas[1].peak if self.arenas[1].peak else 0)

    # -- allocation (ops._empty) ------------------------------------------------------------------------------------
    def empty(self, shape, dtype, device):
        shape = tuple(int(s) for s in shape)
        n = 1
        for s in shape:
            n *= s
        item = torch.empty((), dtype=dtype).element_size()
        region = self.sid()
        off, size = self.arenas[region].alloc(n * item)
        goff = off + (self.split if region else 0)
        # a ROOT tensor on the workspace's storage (not a view of self.ws): its views keep IT alive through ._base, so the
        # finalizer below runs only when the last alias of these bytes is gone
        t = torch.empty((0,), dtype=dtype, device=device).set_(self.ws.untyped_storage(), goff // item, shape)
        weakref.finalize(t, self._release, region, off, size)
        return t

    def _release(self, region, off, size):
        if self.sid() == region:
            self.arenas[region].release(off, size)       # in-order on its own stream: the bytes may be handed out again
        else:
            self.quarantine.append((region, off, size))  # the other stream may still be reading them: wait for a join

    def wait(self, waiter, waitee):
        """torch.cuda.Stream.wait_stream hook: `waiter` waits for everything enqueued on `waitee` so far."""
        a = 0 if waiter.cuda_stream == self.main else 1
        b = 0 if waitee.cuda_stream == self.main else 1
        if a == b:
            return
        self.calls.append(("@wait", [(K_INT, a), (K_INT, b)], a))
        self.n_waits += 1
        if a == 0:                                       # a JOIN (main waits for side): quarantined bytes are safe again.
            for region, off, size in self.quarantine:    # (a fork does not release: the side stream has only caught up with
                self.arenas[region].release(off, size)   #  main, main has not waited for side)
            self.quarantine = []

    # -- pointers ---------------------------------------------------------------------------------------------------
    def note(self, t):
        """ops._ptr hook: remember tensors that are NOT workspace tensors (packed weights) — holding a reference to a workspace
        tensor would keep its bytes from being handed out again."""
        p = t.data_ptr()
        if not (self.base <= p < self.base + self.size):
            self._check_constant(p, t.numel() * t.element_size())
            self.seen[p] = t

    def _is_input(self, addr):
        for t in (self.grd, self.sat):
            p = t.data_ptr()
            if p <= addr < p + t.numel() * t.element_size():
                return True
        return False

    def _check_constant(self, addr, nbytes=1):
        if self.constants is None or self._is_input(addr):
            return
        import bisect
        i = bisect.bisect_right(self.constants, (addr, float("inf"))) - 1
        if i >= 0 and self.constants[i][0] <= addr and addr + max(nbytes, 1) <= self.constants[i][1]:
            return
        raise RuntimeError("plan: pointer 0x%x (%d bytes) is neither workspace, nor an input, nor part of the weights packed before "
                           "recording: an intermediate allocated outside ops._empty would be frozen as a constant" % (addr, nbytes))

    def classify(self, addr):
        if addr is None or addr == 0:
            return K_NULL, 0
        if self.base <= addr < self.base + self.size:
            return K_WORKSPACE, addr - self.base
        for kind, t in ((K_GRD, self.grd), (K_SAT, self.sat)):
            p = t.data_ptr()
            if p <= addr < p + t.numel() * t.element_size():
                return kind, addr - p
        return K_WEIGHTS, self._weight_offset(addr)

    def _weight_offset(self, addr):
        self._check_constant(addr)
        if addr in self.weights:
            return self.weights[addr][0]
        t = self.seen.get(addr)
        if t is None:
            # a pointer INTO a known tensor (a slice of a packed weight)
            for p, tt in self.seen.items():
                if p <= addr < p + tt.numel() * tt.element_size() and p in self.weights:
                    return self.weights[p][0] + (addr - p)
            raise RuntimeError("plan: pointer 0x%x is neither in the workspace, nor an input, nor a tensor passed through ops._ptr" % addr)
        if not t.is_contiguous():
            raise RuntimeError("plan: non-contiguous weight tensor")
        off = (self.weights_bytes + ALIGN - 1) // ALIGN * ALIGN
        self.weights[addr] = (off, t)
        self.weights_bytes = off + t.numel() * t.element_size()
        return off

    # -- calls ------------------------------------------------------------------------------------------------------
    def log(self, name, argtypes, args):
        out = []
        for i, (ty, a) in enumerate(zip(argtypes, args)):
            if isinstance(a, ctypes.Array):                         # host array (shift tables, descriptor widths)
                self.blobs.append((bytes(a), []))
                out.append((K_BLOB, len(self.blobs) - 1))
            elif type(a).__name__ == "CArgObject":                  # ctypes.byref(struct)
                out.append((K_BLOB, self._struct_blob(a._obj)))
            elif ty is ctypes.c_void_p:
                v = a.value if isinstance(a, ctypes.c_void_p) else a
                if i == len(argtypes) - 1:                          # every launch ends in `void* stream`
                    out.append((K_STREAM, 0))
                else:
                    out.append(self.classify(v))
            elif ty is ctypes.c_float:
                out.append((K_FLT, struct.unpack("<I", struct.pack("<f", float(a)))[0]))
            else:
                out.append((K_INT, int(a) & 0xffffffffffffffff))
        self.calls.append((name, out, self.sid()))

    def _struct_blob(self, obj):
        relocs = []
        for fname, ftype in obj._fields_:
            if ftype is ctypes.c_void_p:
                kind, off = self.classify(getattr(obj, fname))
                relocs.append((getattr(type(obj), fname).offset, kind, off))
        self.blobs.append((bytes(obj), relocs))
        return len(self.blobs) - 1


class _RecordingLib(object):
    """Stands in for the ctypes library while a forward is being recorded: launches are logged, then executed."""

    def __init__(self, lib, rec):
        self._lib, self._rec = lib, rec

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name in QUERIES:
            return fn
        if name not in LAUNCHES:
            raise RuntimeError("plan: %s is not an entry point an eval forward may contain" % name)
        argtypes = _lib.PROTOTYPES[name][1]

        def call(*args):
            self._rec.log(name, argtypes, args)
            return fn(*args)
        return call


class Plan(object):
    """A recorded forward: calls, blobs, workspace / weights sizes, the weights blob (uint8 device tensor) and the outputs."""

    def __init__(self, rec, outputs, grd, sat):
        # compact the two regions: the side region starts where the main region's high-water mark ends
        main_peak = (rec.arenas[0].peak + ALIGN - 1) // ALIGN * ALIGN
        shift = rec.split - main_peak

        def fix(off):
            return off - shift if off >= rec.split else off
        self.calls = [(n, [(k, fix(v)) if k == K_WORKSPACE else (k, v) for k, v in a], sid) for n, a, sid in rec.calls]
        self.blobs = [(d, [(f, k, fix(v)) if k == K_WORKSPACE else (f, k, v) for f, k, v in r]) for d, r in rec.blobs]
        self.workspace_bytes = main_peak + (rec.arenas[1].peak + ALIGN - 1) // ALIGN * ALIGN
        self.n_waits = rec.n_waits
        self.weights_bytes = (rec.weights_bytes + ALIGN - 1) // ALIGN * ALIGN
        self.grd_shape, self.sat_shape = tuple(grd.shape), tuple(sat.shape)
        self.outputs = []
        for t in outputs:
            if t.dtype != torch.float32 or t.dim() > 4:
                raise RuntimeError("plan: outputs must be fp32 tensors of at most 4 dimensions")
            kind, off = rec.classify(t.data_ptr())
            if kind != K_WORKSPACE:
                raise RuntimeError("plan: an output lives outside the workspace")
            # (an output may be a strided view: ori_prior returns a channel slice of the level-1 score volume, models.py:501-511)
            extent = 4 * (1 + sum((n - 1) * st for n, st in zip(t.shape, t.stride())))
            self.outputs.append((fix(off), extent, tuple(t.shape), tuple(t.stride())))
        blob = torch.zeros((max(self.weights_bytes, ALIGN),), dtype=torch.uint8, device=grd.device)
        for off, t in rec.weights.values():
            n = t.numel() * t.element_size()
            blob[off:off + n] = t.reshape(-1).view(torch.uint8)
        self.weights = blob

    def to_bytes(self):
        lib = _lib.load()
        head = [b"CCVPLAN1", struct.pack("<II", lib.ccvpe_abi_version(), 0),
                struct.pack("<QQQQ", self.workspace_bytes, self.weights_bytes,
                            4 * _numel(self.grd_shape), 4 * _numel(self.sat_shape)),
                struct.pack("<IIII", len(self.calls), len(self.blobs), len(self.outputs), 0)]
        for off, nbytes, shape, strides in self.outputs:
            dims = list(shape) + [1] * (4 - len(shape))
            strd = list(strides) + [0] * (4 - len(strides))
            head.append(struct.pack("<QQII4Q4Q", off, nbytes, 0, len(shape), *(dims + strd)))
        for data, relocs in self.blobs:
            head.append(struct.pack("<II", len(data), len(relocs)))
            for field, kind, value in relocs:
                head.append(struct.pack("<IIQ", field, kind, value))
            head.append(data + b"\0" * ((-len(data)) % 8))
        for name, args, sid in self.calls:
            head.append(name.encode().ljust(48, b"\0"))
            head.append(struct.pack("<II", len(args), sid))
            for kind, value in args:
                head.append(struct.pack("<IIQ", kind, 0, value))
        body = b"".join(head)
        body += b"\0" * ((-len(body)) % ALIGN)
        return body + bytes(self.weights[:self.weights_bytes].cpu().numpy().tobytes())

    def save(self, path):
        with open(path, "wb") as f:
            f.write(self.to_bytes())


def _numel(shape):
    n = 1
    for s in shape:
        n *= int(s)
    return n


def record(net, grd, sat, arena_bytes=None):
    """Runs ONE eval forward of `net` on (grd, sat) and returns (Plan, outputs of that forward)."""
    if net.training:
        raise RuntimeError("plan.record: eval mode only (net.eval())")
    if not (grd.is_cuda and sat.is_cuda and grd.dtype == torch.float32 and sat.dtype == torch.float32
            and grd.is_contiguous() and sat.is_contiguous()):
        raise ValueError("plan.record: grd / sat must be contiguous fp32 device tensors")
    dev = grd.device
    with torch.no_grad():
        if arena_bytes is None:                      # size the arena from an ordinary forward (also packs the weights)
            torch.cuda.synchronize(dev)
            torch.cuda.reset_peak_memory_stats(dev)
            before = torch.cuda.memory_allocated(dev)
            net(grd, sat)
            torch.cuda.synchronize(dev)
            arena_bytes = int(1.5 * (torch.cuda.max_memory_allocated(dev) - before)) + (64 << 20)
        constants = _tensor_ranges([net._packed(), net])   # packed weights (+ tables cached on the model) exist BEFORE recording: nothing else may be frozen
    lib = _lib.load()
    saved = (ops._record, _lib.load, torch.cuda.Stream.wait_stream)
    for attempt in range(4):
        rec = _Recorder(dev, arena_bytes, grd, sat, constants=constants)
        orig_wait = saved[2]

        def wait_stream(self, other, _rec=rec, _orig=orig_wait):
            _rec.wait(self, other)
            return _orig(self, other)
        try:
            ops._record = rec
            proxy = _RecordingLib(lib, rec)
            _lib.load = lambda: proxy
            torch.cuda.Stream.wait_stream = wait_stream
            with torch.no_grad():
                outs = net(grd, sat)
            break
        except MemoryError:                          # a region was too small (first-fit fragmentation): grow and record again
            arena_bytes = int(arena_bytes * 1.5)
            outs = None
        finally:
            ops._record, _lib.load, torch.cuda.Stream.wait_stream = saved
            torch.cuda.synchronize(dev)
    if outs is None:
        raise MemoryError("plan.record: workspace regions exhausted after growing the arena")
    torch.cuda.synchronize(dev)
    return Plan(rec, outs, grd, sat), outs


class PlannedForward(object):
    """forward(grd, sat) as one C call (ccvpe_forward).  Built for fixed input shapes, precision and weights (re-build after a
    weight update); the nine returned tensors are views of the ctx's workspace and are overwritten by the next call."""

    def __init__(self, net, grd, sat):
        self.plan, _ = record(net, grd, sat)
        lib = _lib.load()
        self._blob = self.plan.to_bytes()            # host copy: ccvpe_ctx_create uploads the weights from it
        self.workspace = torch.empty((max(self.plan.workspace_bytes, ALIGN),), dtype=torch.uint8, device=grd.device)
        self.weights = torch.empty((max(self.plan.weights_bytes, ALIGN),), dtype=torch.uint8, device=grd.device)
        ctx = ctypes.c_void_p()
        host = ctypes.create_string_buffer(self._blob, len(self._blob))
        _lib.check(lib.ccvpe_ctx_create(ctypes.cast(host, ctypes.c_void_p), len(self._blob), ctypes.c_void_p(self.weights.data_ptr()),
                                        ctypes.c_void_p(self.workspace.data_ptr()), ctypes.byref(ctx)), "ccvpe_ctx_create")
        self._ctx = ctx
        self._lib = lib
        self._fin = weakref.finalize(self, lib.ccvpe_ctx_destroy, ctx)
        self.outputs = []
        for off, nbytes, shape, strides in self.plan.outputs:
            self.outputs.append(torch.as_strided(self.workspace[off:off + nbytes].view(torch.float32), shape, strides))
        self.grd_shape, self.sat_shape = self.plan.grd_shape, self.plan.sat_shape

    def __call__(self, grd, sat):
        if tuple(grd.shape) != self.grd_shape or tuple(sat.shape) != self.sat_shape:
            raise ValueError("PlannedForward was built for grd %s / sat %s" % (self.grd_shape, self.sat_shape))
        if not (grd.is_cuda and sat.is_cuda and grd.dtype == torch.float32 and sat.dtype == torch.float32
                and grd.is_contiguous() and sat.is_contiguous()):
            raise ValueError("PlannedForward: contiguous fp32 device inputs required")
        _lib.check(self._lib.ccvpe_forward(self._ctx, ctypes.c_void_p(grd.data_ptr()), ctypes.c_void_p(sat.data_ptr()), None,
                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "ccvpe_forward")
        return tuple(self.outputs)
