"""Host logic of the one-launch train-mode re-pack (ccvpe_amd/repack.py): the chunk tables derived by running the pack code on
index-carrying tensors must reproduce the eager pack exactly when the gather is emulated with numpy (no GPU needed)."""
import numpy as np
import pytest
import torch

from ccvpe_amd import models, repack


def emulate(sd, names, leaves, dst_leaf, src_no, starts, counts, idx_off, idx_all):
    out = [np.full((t.numel(),), np.nan, dtype=np.float32) for t in leaves]
    flat = {k: sd[k].reshape(-1).numpy() for k in names}
    for l, s, a, n, o in zip(dst_leaf, src_no, starts, counts, idx_off):
        ix = idx_all[o:o + n]
        src = flat[names[s - 1]] if s > 0 else np.zeros((1,), np.float32)
        assert s > 0 or not ix.any()
        out[l][a:a + n] = np.where(ix > 0, src[np.maximum(ix, 1) - 1], np.float32(0))
    return out


@pytest.mark.parametrize("kind", ["vigor", "kitti"])
def test_gather_tables_reproduce_the_eager_train_pack(kind):
    torch.manual_seed(7)
    net = (models.CVM_VIGOR("cpu", True) if kind == "vigor" else models.CVM_KITTI("cpu"))
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    n_tail = models.MODEL_SPECS[net.kind]["n_rot"]

    def pack(src):
        with torch.no_grad():
            return models._pack_model(src, net.kind, n_tail, torch.float32, fold=False)

    pk = pack(sd)
    chunk = 4096
    names, leaves, dst_leaf, src_no, starts, counts, idx_off, idx_all, n_alias = repack.tables(sd, pack, pk, chunk)
    assert counts.max() <= chunk and counts.min() >= 1
    assert n_alias > 0 and len(leaves) > 300            # biases are aliases of the live parameters; weights are re-laid out
    # every destination element is written exactly once
    cover = [np.zeros((t.numel(),), np.int32) for t in leaves]
    for l, a, n in zip(dst_leaf, starts, counts):
        cover[l][a:a + n] += 1
    assert all((c == 1).all() for c in cover)
    got = emulate(sd, names, leaves, dst_leaf, src_no, starts, counts, idx_off, idx_all)
    for t, g in zip(leaves, got):
        assert np.array_equal(t.reshape(-1).numpy(), g)
    # the backward-layout weights are part of the same plan
    assert pk.bwd is not None and any(p.startswith(".bwd[") for p, _ in repack.walk(pk))


def test_chunks_split_at_source_changes_and_padding_joins_a_neighbour():
    idx = np.array([0, 0, 1, 2, 0, 3, 1, 2, 0, 0], dtype=np.int32)
    tid = np.array([0, 0, 5, 5, 0, 5, 9, 9, 0, 0], dtype=np.int32)
    assert repack.chunks_of(idx, tid, 4) == [(0, 4, 5), (4, 2, 5), (6, 4, 9)]
    assert repack.chunks_of(np.zeros(5, np.int32), np.zeros(5, np.int32), 4) == [(0, 4, 0), (4, 1, 0)]
    assert repack.chunks_of(np.zeros(0, np.int32), np.zeros(0, np.int32), 4) == []
