"""Train-mode weight re-pack in ONE kernel launch.

train_VIGOR.py:146-150: `optimizer.step()` rewrites every weight every iteration, so every training step re-lays all of them
out for the forward and backward GEMMs (models._pack_model(fold=False): ~1 000 tiny torch launches — permute, pad, flip, cat,
copy — 4.2 ms of GPU time per B = 64 step even when replayed as one hipGraph).  In train mode that re-pack is PURE DATA MOVEMENT
(no BatchNorm fold, no fp64 deconv fold): element j of a packed tensor is 0 or one element of one live parameter.  The plan is
found by running the very same pack code on tensors that carry their own element indices (and, in a second run, their source's
number) instead of weights; the per-step work is then `ccvpe_gather_repack_f32`: dst[j] = idx[j] ? src[idx[j] - 1] : 0.

A plan is tied to the parameters' storage addresses (the caller re-plans when they move) and is VERIFIED when it is built: the
gathered tensors must equal a fresh eager pack bit for bit, otherwise build() returns None and the caller keeps its old path.
Host logic (index propagation, chunking) runs on any device; only run() needs the HIP library.
"""
import numpy as np
import torch

MAX_EXACT = 1 << 24          # float32 carries integers exactly up to 2^24: per-tensor element indices must stay below


def walk(obj, prefix=""):
    """(path, tensor) leaves of a pack object tree (attribute objects, lists, dicts), in a deterministic order."""
    if torch.is_tensor(obj):
        yield prefix, obj
    elif isinstance(obj, dict):
        for k in sorted(obj):
            yield from walk(obj[k], "%s[%s]" % (prefix, k))
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            yield from walk(v, "%s[%d]" % (prefix, i))
    elif hasattr(obj, "__dict__"):
        for k in sorted(vars(obj)):
            yield from walk(getattr(obj, k), "%s.%s" % (prefix, k))


def _inside(t, sources):
    """True if tensor t lives inside the storage of one of the live source tensors (an alias: always current, nothing to do)."""
    a = t.data_ptr()
    for s in sources:
        b = s.data_ptr()
        if b <= a < b + max(s.numel() * s.element_size(), 1):
            return True
    return False


def index_maps(sd, build):
    """Run `build` on index-carrying stand-ins of the float32 tensors of sd.  Returns (names, pk_idx, pk_tid): names[t] is the
    source numbered t + 1; a leaf of pk_idx holds (element index + 1) or 0 for padding, the same leaf of pk_tid the source number."""
    names = [k for k, v in sd.items() if v.dtype == torch.float32]
    sd_idx, sd_tid = dict(sd), dict(sd)
    for t, k in enumerate(names):
        v = sd[k]
        if v.numel() >= MAX_EXACT:
            raise ValueError("repack: %s has %d elements (>= 2^24)" % (k, v.numel()))
        sd_idx[k] = torch.arange(1, v.numel() + 1, device=v.device, dtype=torch.float32).view(v.shape)
        sd_tid[k] = torch.full(tuple(v.shape), float(t + 1), device=v.device, dtype=torch.float32)
    return names, build(sd_idx), build(sd_tid)


def chunks_of(idx, tid, chunk):
    """Cut one packed tensor (flat int arrays idx, tid; 0 = padding) into (start, count, source number) pieces that read ONE source
    and hold at most `chunk` elements.  Padding joins the run in front of it (or behind it at the start of the tensor)."""
    n = idx.shape[0]
    if n == 0:
        return []
    filled = tid.copy()
    nz = np.flatnonzero(filled)
    if nz.size == 0:
        runs = [(0, n, 0)]
    else:
        # forward-fill zeros with the previous non-zero source (the leading zeros take the first one)
        pos = np.zeros(n, dtype=np.int64)
        pos[nz] = nz
        np.maximum.accumulate(pos, out=pos)
        filled = filled[pos]
        filled[:nz[0]] = filled[nz[0]]
        cuts = np.flatnonzero(np.diff(filled)) + 1
        bounds = np.concatenate(([0], cuts, [n]))
        runs = [(int(bounds[i]), int(bounds[i + 1]), int(filled[bounds[i]])) for i in range(len(bounds) - 1)]
    out = []
    for a, b, t in runs:
        for s in range(a, b, chunk):
            out.append((s, min(chunk, b - s), t))
    return out


class Plan(object):
    """Device tables of one re-pack: run(stream) re-derives every packed tensor of `pk` from the live parameters."""

    def __init__(self, pk, dsts, srcs, idx_off, counts, idx, n_elems, n_alias):
        self.pk, self.dsts, self.srcs, self.idx_off, self.counts, self.idx = pk, dsts, srcs, idx_off, counts, idx
        self.n_chunks = int(counts.numel())
        self.n_elems, self.n_alias = n_elems, n_alias

    def run(self):
        from . import _lib
        from .ops import _ptr, _stream, check
        check(_lib.load().ccvpe_gather_repack_f32(_ptr(self.dsts), _ptr(self.srcs), _ptr(self.idx_off), _ptr(self.counts),
                                                   _ptr(self.idx), self.n_chunks, _stream()), "ccvpe_gather_repack_f32")
        return self.pk


def tables(sd, build, pk, chunk):
    """Host side of build(): the chunk tables as numpy arrays + the list of destination leaves.  Raises ValueError when the pack
    is not a pure re-layout this scheme can express (a non-float32 leaf, a tensor >= 2^24 elements, a shape mismatch)."""
    names, pk_idx, pk_tid = index_maps(sd, build)
    sources = [sd[k] for k in names]
    leaves, li, lt = list(walk(pk)), list(walk(pk_idx)), list(walk(pk_tid))
    if [p for p, _ in leaves] != [p for p, _ in li] or [p for p, _ in leaves] != [p for p, _ in lt]:
        raise ValueError("repack: the pack's structure depends on the weight VALUES")
    dst_leaf, src_no, starts, counts, idx_parts, idx_off = [], [], [], [], [], []
    total = 0
    n_alias = 0
    for (path, t), (_, ti), (_, tt) in zip(leaves, li, lt):
        if _inside(t, sources):
            n_alias += 1
            continue
        if t.dtype != torch.float32 or not t.is_contiguous() or t.shape != ti.shape:
            raise ValueError("repack: leaf %s is %s / contiguous=%s" % (path, t.dtype, t.is_contiguous()))
        ix = ti.reshape(-1).round().to(torch.int32).cpu().numpy()
        td = tt.reshape(-1).round().to(torch.int32).cpu().numpy()
        for s, n, src in chunks_of(ix, td, chunk):
            dst_leaf.append(len(idx_parts))
            src_no.append(src)
            starts.append(s)
            counts.append(n)
            idx_off.append(total + s)
        idx_parts.append((t, ix))
        total += ix.shape[0]
    idx_all = np.concatenate([ix for _, ix in idx_parts]) if idx_parts else np.zeros((0,), np.int32)
    return names, [t for t, _ in idx_parts], np.asarray(dst_leaf), np.asarray(src_no), np.asarray(starts, dtype=np.int64), \
        np.asarray(counts, dtype=np.int32), np.asarray(idx_off, dtype=np.int64), idx_all, n_alias


def build(sd, pack, verify=True):
    """sd: name -> live (detached) tensors; pack(sd_like) -> the pack object tree.  Returns a Plan (whose .pk has just been
    re-derived by the gather kernel) or None when the pack cannot be expressed / does not verify."""
    from . import _lib
    chunk = _lib.load().ccvpe_gather_repack_chunk()
    pk = pack(sd)
    try:
        names, leaves, dst_leaf, src_no, starts, counts, idx_off, idx_all, n_alias = tables(sd, pack, pk, chunk)
    except ValueError:
        return None
    if len(leaves) == 0:
        return None
    dev = leaves[0].device
    dptr = np.asarray([leaves[l].data_ptr() for l in dst_leaf], dtype=np.int64) + starts * 4
    # padding-only chunks carry source 0: any valid pointer will do, it is never dereferenced
    anyp = sd[names[0]].data_ptr()
    sptr = np.asarray([sd[names[s - 1]].data_ptr() if s > 0 else anyp for s in src_no], dtype=np.int64)
    plan = Plan(pk, torch.from_numpy(dptr).to(dev), torch.from_numpy(sptr).to(dev), torch.from_numpy(idx_off).to(dev),
                torch.from_numpy(counts).to(dev), torch.from_numpy(idx_all).to(dev), int(idx_all.shape[0]), n_alias)
    plan._keep = (leaves, [sd[k] for k in names])          # the tables hold raw addresses of these tensors
    if verify:
        for t in leaves:
            t.fill_(float("nan"))
        plan.run()
        ref = pack(sd)
        for (path, a), (_, b) in zip(walk(pk), walk(ref)):
            if a.dtype != b.dtype or a.shape != b.shape or not torch.equal(a, b):
                return None
    return plan
