"""SURVEY.md 8(f)-4, device half: ccvpe_amd.datasets.DeviceBatches — decoded uint8 images + three scalars per sample in,
normalised / rolled / cropped fp32 batches and the training ground truth out, all as kernels on the MI355X — against the
oracle's PIL-exact transform and ground-truth restatement (both pinned to the reference's dataset class in
tests/test_oracle_vs_reference.py / tests/test_datasets.py)."""
import os

import numpy as np
import pytest
import torch

from ccvpe_amd import datasets as DS
from oracle import ccvpe_oracle as O
from test_datasets import GOLDEN, make_vigor_tree

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fov,jpeg", [(360, True), (180, False)])
def test_device_batches_match_the_oracle_pipeline(tmp_path, fov, jpeg):
    make_vigor_tree(str(tmp_path), jpeg=jpeg)
    orient = np.load(os.path.join(GOLDEN, "samearea_orientation_test_head256.npy"))
    ds = DS.VIGORPairs(str(tmp_path), split="samearea", train=True, ori_noise=180, random_orientation=orient, strict_orientation=False)
    seen = []
    for batch in DS.DeviceBatches(ds, 5, device="cuda", workers=3, fov=fov):
        b = len(batch.indices)
        keep = int(fov / 360 * 640)
        assert tuple(batch.grd.shape) == (b, 3, 320, keep) and tuple(batch.sat.shape) == (b, 3, 512, 512)
        for j, i in enumerate(batch.indices):
            s = ds.sample(i)
            want_g = O.preprocess_reference(s["grd_u8"], (320, 640), roll=s["roll"])[:, :, :keep]
            assert torch.equal(batch.grd[j].cpu(), want_g), "ground image of sample %d" % i
            assert torch.equal(batch.sat[j].cpu(), O.preprocess_reference(s["sat_u8"], (512, 512)))
            g, flat, ori, labs = O.train_targets([list(s["center"])], [s["angle_deg"]], 20)
            # (the device takes the angle as fp32: the orientation-bin weights (angle % 18) / 18 carry ~1e-6 of that)
            assert (batch.gt[j].cpu() - g[0]).abs().max() < 1e-5 and (batch.gt_ori[j].cpu() - ori[0]).abs().max() < 1e-5
            assert (batch.gt_flat[j].cpu() - flat[0]).abs().max() < 2e-6 * float(flat[0].max())      # (normalised to sum 1: peak ~1e-2)
            for a, w in zip(batch.labels, labs):
                assert (a[j].cpu() - w[0]).abs().max() < 1e-5
            assert batch.cities[j] == ds.city_of[i] and abs(float(batch.angle_deg[j]) - s["angle_deg"]) < 1e-3
        seen += batch.indices
    assert seen == list(range(12))


def test_device_batches_shards_and_model_forward(tmp_path, synth_sd):
    """Two ranks' shards are disjoint and complete, and a batch goes straight into the model."""
    from ccvpe_amd import models
    make_vigor_tree(str(tmp_path))
    ds = DS.VIGORPairs(str(tmp_path), split="crossarea", train=False, ori_noise=180,
                       random_orientation=os.path.join(GOLDEN, "crossarea_orientation_test_head256.npy"), strict_orientation=False)
    got = [sum((b.indices for b in DS.DeviceBatches(ds, 2, device="cuda", rank=r, world=2, targets=False)), []) for r in range(2)]
    assert sorted(got[0] + got[1]) == list(range(6)) and not set(got[0]) & set(got[1])
    net = models.CVM_VIGOR_ori_prior("cuda", 180, True)
    net.load_state_dict(synth_sd("vigor", 0), strict=True)
    net = net.to("cuda:0").eval()
    batch = next(iter(DS.DeviceBatches(ds, 2, device="cuda", targets=False)))
    out = net(batch.grd, batch.sat)
    assert tuple(out[0].shape) == (2, 512 * 512) and torch.isfinite(out[0]).all()


def test_unreadable_aerial_tile_raises_in_the_consumer(tmp_path):
    make_vigor_tree(str(tmp_path))
    ds = DS.VIGORPairs(str(tmp_path), split="samearea", train=True)
    os.remove(ds.sat_paths[ds.labels[3, 0]])
    with pytest.raises((OSError, FileNotFoundError)):
        for _ in DS.DeviceBatches(ds, 4, device="cuda"):
            pass


def test_kitti_device_batches_match_the_oracle_pipeline_and_feed_the_model(tmp_path, synth_sd):
    """KITTIPairs (listed test perturbations: deterministic) -> DeviceBatches with the KITTI geometry (256 x 1024 camera image,
    16 orientation bins) against the oracle's transform / ground-truth restatement; a batch goes straight into CVM_KITTI."""
    from ccvpe_amd import models
    from test_datasets import make_kitti_tree
    train_file, test_file, names = make_kitti_tree(str(tmp_path))
    ds = DS.KITTIPairs(str(tmp_path), test_file, 20, 20, 10, test=True)
    seen = []
    last = None
    for batch in DS.DeviceBatches(ds, 3, device="cuda", workers=2, grd_hw=(256, 1024), n_bins=16):
        b = len(batch.indices)
        assert tuple(batch.grd.shape) == (b, 3, 256, 1024) and tuple(batch.sat.shape) == (b, 3, 512, 512)
        assert tuple(batch.labels[0].shape) == (b, 16, 8, 8)
        for j, i in enumerate(batch.indices):
            s = ds.sample(i)
            assert torch.equal(batch.grd[j].cpu(), O.preprocess_reference(s["grd_u8"], (256, 1024)))
            assert torch.equal(batch.sat[j].cpu(), O.preprocess_reference(s["sat_u8"], (512, 512)))
            g, flat, ori, labs = O.train_targets([list(s["center"])], [s["angle_deg"]], 16)
            assert (batch.gt[j].cpu() - g[0]).abs().max() < 1e-5 and (batch.gt_ori[j].cpu() - ori[0]).abs().max() < 1e-5
            for a, w in zip(batch.labels, labs):
                assert (a[j].cpu() - w[0]).abs().max() < 1e-5
        seen += batch.indices
        last = batch
    assert seen == list(range(4))
    net = models.CVM_KITTI("cuda")
    net.load_state_dict(synth_sd("kitti", 0), strict=True)
    net = net.to("cuda:0").eval()
    out = net(last.grd, last.sat)
    assert tuple(out[0].shape) == (1, 512 * 512) and torch.isfinite(out[0]).all()


def test_oxford_device_batches_ascending_bins_and_model(tmp_path, synth_sd):
    """OxfordPairs (val split: deterministic grid patches) -> DeviceBatches with the Oxford geometry (154 x 231 camera image, 800 x 800
    map patch resized to 512, 20 bins counted UP) against the oracle; a batch goes straight into CVM_OxfordRobotCar."""
    from ccvpe_amd import models
    from test_datasets import make_oxford_tree
    g, sat_path = make_oxford_tree(str(tmp_path))
    ds = DS.OxfordPairs(g, sat_path, split="val")
    batch = next(iter(DS.DeviceBatches(ds, 3, device="cuda", workers=2, grd_hw=(154, 231), n_bins=20, ascending_bins=True)))
    assert tuple(batch.grd.shape) == (3, 3, 154, 231) and tuple(batch.sat.shape) == (3, 3, 512, 512)
    for j, i in enumerate(batch.indices):
        s = ds.sample(i)
        assert torch.equal(batch.grd[j].cpu(), O.preprocess_reference(s["grd_u8"], (154, 231)))
        assert torch.equal(batch.sat[j].cpu(), O.preprocess_reference(s["sat_u8"], (512, 512)))
        gg, flat, ori, labs = O.train_targets([list(s["center"])], [s["angle_deg"]], 20, ascending=True)
        desc = O.train_targets([list(s["center"])], [s["angle_deg"]], 20)[3]
        assert (batch.gt[j].cpu() - gg[0]).abs().max() < 1e-5
        for a, w, d in zip(batch.labels, labs, desc):
            assert (a[j].cpu() - w[0]).abs().max() < 1e-5
            assert (w[0] - d[0]).abs().max() > 1e-3            # the two bin orders really differ for these angles
    net = models.CVM_OxfordRobotCar("cuda")
    net.load_state_dict(synth_sd("oxford", 0), strict=True)
    net = net.to("cuda:0").eval()
    out = net(batch.grd, batch.sat)
    assert tuple(out[0].shape) == (3, 512 * 512) and torch.isfinite(out[0]).all()
