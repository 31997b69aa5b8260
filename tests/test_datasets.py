"""the host side of the VIGOR input pipeline (ccvpe_amd/datasets.py: split files -> index, PIL decode,
the sample's orientation / positive-tile choice, sharded iteration) — CPU tests against the oracle's restatement of the
reference dataset's arithmetic and, where /root/reference exists (the build container), against the reference's own
`VIGORDataset.__getitem__` on the same synthetic dataset directory.  The device half (resize / normalise / roll / crop /
ground truth kernels) is in tests/test_datasets_gpu.py."""
import os

import numpy as np
import pytest
import torch

from ccvpe_amd import datasets as DS
from oracle import ccvpe_oracle as O
from ref_import import reference_available, import_reference_datasets

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CITIES = ("NewYork", "Seattle", "SanFrancisco", "Chicago")


def make_vigor_tree(root, per_city=3, sat_size=64, pano_hw=(96, 192), jpeg=True, seed=3):
    """A tiny dataset directory in the VIGOR layout (4 cities, `per_city` panoramas each, 4 aerial tiles per city, the
    three split files).  Returns {city: [(pano name, [(tile, drow, dcol)] * 4)]}."""
    from PIL import Image
    rng = np.random.RandomState(seed)
    ext = "jpg" if jpeg else "png"
    spec = {}
    for ci, city in enumerate(CITIES):
        os.makedirs(os.path.join(root, "splits_new", city))
        os.makedirs(os.path.join(root, city, "satellite"))
        os.makedirs(os.path.join(root, city, "panorama"))
        tiles = ["s_%s_%d.%s" % (city, k, ext) for k in range(4)]           # (VIGOR tile names are unique across cities)
        for t in tiles:
            Image.fromarray((rng.rand(sat_size, sat_size, 3) * 255).astype(np.uint8), "RGB").save(os.path.join(root, city, "satellite", t))
        with open(os.path.join(root, "splits_new", city, "satellite_list.txt"), "w") as f:
            f.write("".join(t + "\n" for t in tiles))
        rows = []
        for p in range(per_city):
            name = "p%d.%s" % (p, ext)
            Image.fromarray((rng.rand(pano_hw[0], pano_hw[1], 3) * 255).astype(np.uint8), "RGB").save(os.path.join(root, city, "panorama", name))
            order = rng.permutation(4)
            # first tile: the positive (inside the tile); the others: some inside, some >= 320 raw pixels away
            offs = [(7.0 * ci - 5.0 + p, -3.0 * ci + 11.0 - 2 * p), (12.5, -20.25), (400.0, 3.0), (-15.0, -330.0)]
            rows.append((name, [(tiles[order[k]], offs[k][0], offs[k][1]) for k in range(4)]))
        spec[city] = rows
        text = "".join("%s %s\n" % (n, " ".join("%s %f %f" % t for t in ts)) for n, ts in rows)
        for fn in ("same_area_balanced_train.txt", "same_area_balanced_test.txt", "pano_label_balanced.txt"):
            with open(os.path.join(root, "splits_new", city, fn), "w") as f:
                f.write(text)
    return spec


def test_index_follows_the_split_files(tmp_path):
    spec = make_vigor_tree(str(tmp_path))
    ds = DS.VIGORPairs(str(tmp_path), split="samearea", train=True)
    assert len(ds) == 12 and len(ds.sat_paths) == 16
    i = 0
    for ci, city in enumerate(CITIES):
        for name, tiles in spec[city]:
            assert ds.grd_paths[i] == os.path.join(str(tmp_path), city, "panorama", name) and ds.city_of[i] == city
            for k, (t, dr, dc) in enumerate(tiles):
                assert ds.sat_paths[ds.labels[i, k]] == os.path.join(str(tmp_path), city, "satellite", t)
                assert ds.deltas[i, k, 0] == pytest.approx(dr) and ds.deltas[i, k, 1] == pytest.approx(dc)
            i += 1
    # cross-area: train = NewYork + Seattle, test = SanFrancisco + Chicago, tile indices local to the split
    tr = DS.VIGORPairs(str(tmp_path), split="crossarea", train=True)
    te = DS.VIGORPairs(str(tmp_path), split="crossarea", train=False)
    assert (len(tr), len(te), len(tr.sat_paths), len(te.sat_paths)) == (6, 6, 8, 8)
    assert set(tr.city_of) == {"NewYork", "Seattle"} and set(te.city_of) == {"SanFrancisco", "Chicago"}
    with pytest.raises(ValueError):
        DS.VIGORPairs(str(tmp_path), split="elsewhere")
    with pytest.raises(ValueError):
        DS.VIGORPairs(str(tmp_path), random_orientation=np.zeros(3))          # fewer angles than samples


def test_orientation_fixture_and_random_choices(tmp_path):
    make_vigor_tree(str(tmp_path))
    head = os.path.join(GOLDEN, "samearea_orientation_test_head256.npy")      # first 256 entries of the reference's fixture
    ds = DS.VIGORPairs(str(tmp_path), split="samearea", train=False, ori_noise=180, random_orientation=head, strict_orientation=False)
    angles = np.load(head)
    assert angles.dtype == np.float64 and 0.0 <= angles.min() and angles.max() < 360.0
    for i in range(len(ds)):
        s = ds.sample(i)
        assert s["angle_deg"] == pytest.approx(angles[i], abs=1e-9)
        assert s["roll"] == int(torch.round(torch.as_tensor(angles[i] / 360) * 640).int())
        assert s["grd_u8"].dtype == np.uint8 and s["grd_u8"].shape == (96, 192, 3) and s["sat_u8"].shape == (64, 64, 3)
    # no fixture: ori_noise >= 180 -> anywhere on the circle; ori_noise = 36 -> within +-36 degrees; seeded
    a = [DS.VIGORPairs(str(tmp_path), ori_noise=180, seed=5).rotation_fraction(0) for _ in range(2)]
    assert a[0] == a[1] and 0.0 <= a[0] < 1.0
    ds36 = DS.VIGORPairs(str(tmp_path), ori_noise=36, seed=1)
    fr = np.array([ds36.rotation_fraction(0) for _ in range(200)])
    assert np.abs(fr).max() <= 0.1 and fr.min() < -0.05 and fr.max() > 0.05
    # semi-positives: only tiles with the ground truth inside (|offset| < 320 raw pixels) are ever drawn
    dsp = DS.VIGORPairs(str(tmp_path), pos_only=False, seed=2)
    seen = set()
    for _ in range(100):
        k, dr, dc = dsp.positive(0)
        assert abs(dr) < 320 and abs(dc) < 320
        seen.add(k)
    assert seen == {0, 1}


def test_sample_matches_the_oracle_restatement(tmp_path):
    """Decoded images + (roll, centre, angle) reproduce, through the oracle's PIL-exact transform and ground-truth
    restatement (pinned to the reference in test_oracle_vs_reference.py), what the reference's dataset returns."""
    spec = make_vigor_tree(str(tmp_path), jpeg=False)
    orient = np.linspace(0.0, 359.0, 12)
    ds = DS.VIGORPairs(str(tmp_path), split="samearea", train=True, ori_noise=180, random_orientation=orient)
    from PIL import Image
    for i in (0, 5, 11):
        s = ds.sample(i)
        city = ds.city_of[i]
        name, tiles = spec[city][i % 3]
        pano = np.asarray(Image.open(os.path.join(str(tmp_path), city, "panorama", name)).convert("RGB"))
        assert np.array_equal(s["grd_u8"], pano)
        assert s["center"] == (float(np.round(tiles[0][2] / 64 * 512)), float(-np.round(tiles[0][1] / 64 * 512)))
        g, flat, ori, labs = O.train_targets([list(s["center"])], [s["angle_deg"]], 20)
        assert g.shape == (1, 1, 512, 512) and abs(float(flat.sum()) - 1.0) < 1e-5
        peak = int(g[0, 0].argmax())
        assert abs(peak % 512 - (256 - s["center"][0])) <= 1 and abs(peak // 512 - (256 - s["center"][1])) <= 1


@pytest.mark.skipif(not reference_available(), reason="reference not present")
def test_sample_matches_the_reference_dataset_live(tmp_path):
    """The reference's own VIGORDataset on the same directory: identical sample order, orientation angle, rolled +
    normalised panorama, aerial image and ground truth."""
    import torch.nn.functional as F
    make_vigor_tree(str(tmp_path), jpeg=True)
    D = import_reference_datasets()

    def tf(hw):
        return lambda im: O.preprocess_reference(np.asarray(im.convert("RGB")), hw)
    orient = np.load(os.path.join(GOLDEN, "crossarea_orientation_test_head256.npy"))
    for split, train in (("samearea", False), ("crossarea", True), ("crossarea", False)):
        ref = D.VIGORDataset(str(tmp_path), split=split, train=train, transform=(tf((320, 640)), tf((512, 512))), pos_only=True,
                             ori_noise=180, random_orientation=orient)
        ours = DS.VIGORPairs(str(tmp_path), split=split, train=train, pos_only=True, ori_noise=180, random_orientation=orient, strict_orientation=False)
        assert len(ours) == len(ref)
        assert list(ours.grd_paths) == list(ref.grd_list) and list(ours.sat_paths) == list(ref.sat_list)
        assert np.array_equal(ours.labels, ref.label) and np.array_equal(ours.deltas, ref.delta)
        for i in range(0, len(ours), 2):
            grd, sat, gt, gt_with_ori, orientation, city, angle = ref[i]
            s = ours.sample(i)
            assert s["city"] == city and abs(s["angle_deg"] - angle) < 1e-9
            assert torch.equal(O.preprocess_reference(s["grd_u8"], (320, 640), roll=s["roll"]), grd)
            assert torch.equal(O.preprocess_reference(s["sat_u8"], (512, 512)), sat)
            g, flat, ori, labs = O.train_targets([list(s["center"])], [s["angle_deg"]], 20)
            assert torch.equal(g[0], gt) and torch.equal(ori[0], orientation)
            if train:
                for k, lab in zip((64, 32, 16, 8, 4, 2), labs):
                    assert torch.equal(lab[0], F.max_pool2d(gt_with_ori[None], k, stride=k)[0])


def test_device_batches_shard_without_overlap(tmp_path):
    """Rank shards of the (optionally shuffled) index list are disjoint and cover it; no GPU needed for the planning."""
    make_vigor_tree(str(tmp_path))
    ds = DS.VIGORPairs(str(tmp_path), split="samearea", train=True)
    parts = [DS.DeviceBatches(ds, 4, device="cpu", shuffle=True, seed=9, rank=r, world=3).indices for r in range(3)]
    allidx = np.concatenate(parts)
    assert sorted(allidx.tolist()) == list(range(12)) and [len(p) for p in parts] == [4, 4, 4]
    assert not np.array_equal(np.concatenate(parts), np.arange(12))            # shuffled, identically on every rank
    assert len(DS.DeviceBatches(ds, 5, device="cpu")) == 3 and len(DS.DeviceBatches(ds, 5, device="cpu", drop_last=True)) == 2


def test_training_shards_run_the_same_number_of_batches_on_every_rank():
    """A training step all-reduces gradients inside the backward: a rank with one batch more would block until the RCCL
    timeout.  "pad" (the default with targets and world > 1) and "truncate" make len() rank-independent by construction;
    "exact" (evaluation) covers every sample once and may differ by a batch."""
    class Fake(object):
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

    for n, world, bs, drop in ((129, 2, 64, False), (127, 2, 64, True), (130, 8, 4, False), (5, 8, 2, False), (1000, 3, 64, True)):
        for mode in ("pad", "truncate"):
            its = [DS.DeviceBatches(Fake(n), bs, device="cpu", rank=r, world=world, drop_last=drop, shard=mode) for r in range(world)]
            assert len({len(it) for it in its}) == 1 and len({len(it.indices) for it in its}) == 1, (n, world, bs, drop, mode)
            seen = set(np.concatenate([it.indices for it in its]).tolist())
            if mode == "pad":
                assert seen == set(range(n))                              # every sample at least once
            else:
                assert len(seen) == (n // world) * world                  # no sample twice
        default = [DS.DeviceBatches(Fake(n), bs, device="cpu", rank=r, world=world, drop_last=drop) for r in range(world)]
        assert all(it.shard == "pad" for it in default) and len({len(it) for it in default}) == 1
    # the case of the advisor's finding: exact shards of 129 samples over 2 ranks at B = 64 give 2 and 1 batches
    ev = [DS.DeviceBatches(Fake(129), 64, device="cpu", rank=r, world=2, targets=False) for r in range(2)]
    assert [it.shard for it in ev] == ["exact", "exact"] and [len(it) for it in ev] == [2, 1]
    assert sorted(np.concatenate([it.indices for it in ev]).tolist()) == list(range(129))
    assert DS.DeviceBatches(Fake(7), 2, device="cpu", fov=70).keep_w == int(640 * 70 / 360)     # train_VIGOR.py:177
    # the default follows the SPLIT, not `targets`: a test split with ground truth (validation loss) is still never padded
    class FakeSplit(Fake):
        def __init__(self, n, **kw):
            Fake.__init__(self, n)
            self.__dict__.update(kw)
    for kw, want in ((dict(train=False), "exact"), (dict(train=True), "pad"), (dict(test=True), "exact"), (dict(test=False), "pad"),
                     (dict(split="val"), "exact"), (dict(split="train"), "pad")):
        its = [DS.DeviceBatches(FakeSplit(129, **kw), 64, device="cpu", rank=r, world=2, targets=True) for r in range(2)]
        assert [it.shard for it in its] == [want, want], kw
        if want == "exact":
            assert sorted(np.concatenate([it.indices for it in its]).tolist()) == list(range(129))     # no sample twice


def test_seeded_draws_do_not_depend_on_the_decode_threads(tmp_path):
    """The random choices of a chunk are drawn on the producer thread in index order, so two seeded iterations with 8 decode
    threads agree with each other and with a sequential pass."""
    make_vigor_tree(str(tmp_path))

    def run(workers):
        ds = DS.VIGORPairs(str(tmp_path), split="samearea", train=True, pos_only=False, ori_noise=180, seed=5)
        it = DS.DeviceBatches(ds, 4, device="cpu", workers=workers, targets=False)
        out = []
        with __import__("concurrent.futures").futures.ThreadPoolExecutor(max_workers=workers) as pool:
            for i in range(0, len(it.indices), 4):
                out += [(s["index"], s["roll"], s["angle_deg"], s["center"]) for s in it._decode_batch(pool, it.indices[i:i + 4])]
        return out

    a, b, c = run(8), run(8), run(1)
    assert a == b == c and len(a) == 12
    ds = DS.VIGORPairs(str(tmp_path), split="samearea", train=True, pos_only=False, ori_noise=180, seed=5)
    seq = [ds.sample(i) for i in range(12)]
    assert [(s["index"], s["roll"], s["angle_deg"], s["center"]) for s in seq] == a


def test_malformed_label_line_and_fixture_length_raise(tmp_path):
    import pytest
    make_vigor_tree(str(tmp_path))
    with pytest.raises(ValueError, match="random_orientation has 13 entries for 12"):
        DS.VIGORPairs(str(tmp_path), split="samearea", train=False, random_orientation=np.zeros(13))
    path = os.path.join(str(tmp_path), "splits_new", DS.CITIES[("samearea", True)][0], "same_area_balanced_train.txt")
    lines = open(path).read().splitlines()
    open(path, "w").write("\n".join(lines[:1] + ["p_short.jpg s_x.png 1.0"] + lines[1:]) + "\n")
    with pytest.raises(ValueError, match=r"same_area_balanced_train.txt:2: expected 13 fields"):
        DS.VIGORPairs(str(tmp_path), split="samearea", train=True)


# ---- KITTI ------------------------------------------------------------------------------------------------------------
KITTI_DRIVES = ("2011_09_26/2011_09_26_drive_0001_sync/", "2011_09_30/2011_09_30_drive_0028_sync/")


def make_kitti_tree(root, per_drive=2, sat_size=640, grd_hw=(94, 310), seed=5):
    """A tiny dataset directory in the KITTI layout of the reference's loader (satmap/, raw_data/<drive>/oxts/data,
    raw_data/<drive>/image_02/data) + a training file and a test file.  Returns (train file, test file, names)."""
    from PIL import Image
    rng = np.random.RandomState(seed)
    names = []
    for d in KITTI_DRIVES:
        assert len(d) == 38
        os.makedirs(os.path.join(root, "satmap", d))
        os.makedirs(os.path.join(root, "raw_data", d, "oxts/data"))
        os.makedirs(os.path.join(root, "raw_data", d, "image_02/data"))
        for k in range(per_drive):
            frame = "%010d.png" % (k * 7)
            Image.fromarray((rng.rand(sat_size, sat_size, 3) * 255).astype(np.uint8), "RGB").save(os.path.join(root, "satmap", d, frame))
            Image.fromarray((rng.rand(grd_hw[0], grd_hw[1], 3) * 255).astype(np.uint8), "RGB").save(
                os.path.join(root, "raw_data", d, "image_02/data", frame))
            with open(os.path.join(root, "raw_data", d, "oxts/data", frame.replace(".png", ".txt")), "w") as f:
                f.write("49.01 8.43 112.0 0.02 0.01 %.6f 1.0 2.0\n" % rng.uniform(-3.1, 3.1))
            names.append(d + frame)
    train_file, test_file = os.path.join(root, "train_files.txt"), os.path.join(root, "test1_files.txt")
    with open(train_file, "w") as f:
        f.write("".join(n + "\n" for n in names))
    with open(test_file, "w") as f:
        f.write("".join("%s %.4f %.4f %.4f\n" % (n, rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-1, 1)) for n in names))
    return train_file, test_file, names


def test_kitti_index_paths_and_listed_perturbations(tmp_path):
    train_file, test_file, names = make_kitti_tree(str(tmp_path))
    tr = DS.KITTIPairs(str(tmp_path), train_file)
    te = DS.KITTIPairs(str(tmp_path), test_file, rotation_range=10, test=True)
    assert len(tr) == len(te) == 4 and tr.lines == names
    sat, oxts, grd = tr.paths(names[2])
    assert sat == os.path.join(str(tmp_path), "satmap", names[2]) and oxts.endswith("oxts/data/0000000000.txt")
    assert grd == os.path.join(str(tmp_path), "raw_data", KITTI_DRIVES[1], "image_02/data", "0000000000.png")
    _, sx, sy, th = te.lines[1].split(" ")
    assert te.perturbation(1) == (-float(sx), -float(sy), float(th) * 10)
    assert abs(tr.meter_per_pixel - 0.19582850865) < 1e-9 and abs(tr.shift_px_lat - 20 / tr.meter_per_pixel) < 1e-12
    s = te.sample(1)
    assert s["sat_u8"].shape == (512, 512, 3) and s["grd_u8"].shape == (94, 310, 3) and s["roll"] == 0
    assert 0.0 <= s["angle_deg"] <= 360.0 and s["city"] == KITTI_DRIVES[0]
    # seeded generator instead of numpy's global one: reproducible draws
    a = DS.KITTIPairs(str(tmp_path), train_file, rng=np.random.default_rng(3)).perturbation(0)
    b = DS.KITTIPairs(str(tmp_path), train_file, rng=np.random.default_rng(3)).perturbation(0)
    assert a == b and all(-1 <= v <= 1 for v in a[:2]) and abs(a[2]) <= 10


@pytest.mark.skipif(not reference_available(), reason="reference not present")
def test_kitti_sample_matches_the_reference_datasets_live(tmp_path):
    """The reference's own SatGrdDataset / SatGrdDatasetTest on the same directory: the aligned + perturbed aerial crop, the camera
    image, the orientation angle and — through the oracle's ground-truth restatement with 16 bins — gt, gt_with_ori (max-pooled)
    and the orientation map.  torchvision is not installed in this image: the reference's one use of it here,
    `TF.center_crop` on a PIL image, is supplied with torchvision 0.x's published rule (top/left = int(round((size - crop) / 2)))."""
    import torch.nn.functional as F
    train_file, test_file, names = make_kitti_tree(str(tmp_path))
    D = import_reference_datasets()

    def center_crop(img, size):
        w, h = img.size
        left, top = int(round((w - size) / 2.0)), int(round((h - size) / 2.0))
        return img.crop((left, top, left + size, top + size))
    D.TF.center_crop = center_crop

    def to_u8(im):
        return torch.from_numpy(np.array(im)).permute(2, 0, 1)
    for test in (False, True):
        cls = D.SatGrdDatasetTest if test else D.SatGrdDataset
        ref = cls(str(tmp_path), test_file if test else train_file, transform=(to_u8, to_u8), shift_range_lat=20, shift_range_lon=20,
                  rotation_range=(10 if test else 180))
        ours = DS.KITTIPairs(str(tmp_path), test_file if test else train_file, 20, 20, 10 if test else 180, test=test)
        assert len(ours) == len(ref)
        for i in range(len(ours)):
            np.random.seed(100 + i)
            sat, grd, gt, gt_with_ori, orientation, angle = ref[i]
            np.random.seed(100 + i)
            s = ours.sample(i)
            assert abs(s["angle_deg"] - angle) < 1e-9
            assert np.array_equal(s["sat_u8"], sat.permute(1, 2, 0).numpy())
            assert np.array_equal(s["grd_u8"].astype(np.float32), grd.permute(1, 2, 0).numpy().astype(np.float32))
            g, flat, ori, labs = O.train_targets([list(s["center"])], [s["angle_deg"]], 16)
            assert torch.equal(g[0], gt) and torch.equal(ori[0], orientation)
            for k, lab in zip((64, 32, 16, 8, 4, 2), labs):
                assert torch.equal(lab[0], F.max_pool2d(gt_with_ori[None], k, stride=k)[0])


# ---- Oxford RobotCar ----------------------------------------------------------------------------------------------------
def make_oxford_tree(root, n=3, seed=8):
    """A tiny Oxford RobotCar directory: list files + yaw fixtures + camera images, and an aerial 'map' that covers the
    top-left 2600 x 2800 pixels of the real one (the vehicles sit ~60-80 m south-east of the map's north-west control point)."""
    from PIL import Image
    rng = np.random.RandomState(seed)
    g = os.path.join(root, "grd") + "/"
    os.makedirs(g)
    yy, xx = np.mgrid[0:2800, 0:2600]
    sat = np.stack([(xx * 7 + yy * 3) % 251, (xx // 16 * 13 + yy // 16 * 29) % 241, (xx * yy // 64) % 239], -1).astype(np.uint8)
    sat_path = os.path.join(root, "satellite_map_new.png")
    Image.fromarray(sat, "RGB").save(sat_path)
    for fn, yaw in (("training.txt", "train_yaw.npy"), ("validation.txt", "val_yaw.npy"), ("test1_j.txt", None),
                    ("test2_j.txt", None), ("test3_j.txt", "test_yaw.npy")):
        rows = []
        for k in range(n):
            name = "%s_%d.png" % (fn.split(".")[0], k)
            Image.fromarray((rng.rand(60, 90, 3) * 255).astype(np.uint8), "RGB").save(g + name)
            rows.append("%s %d %.4f %.4f\n" % (name, 1000 + k, 619400 + rng.uniform(55, 85), 5736195 - rng.uniform(60, 90)))
        with open(g + fn, "w") as f:
            f.write("".join(rows))
        if yaw:
            np.save(g + yaw, rng.uniform(0, 2 * np.pi, size=(3 * n if fn == "test3_j.txt" else n)))
    return g, sat_path


def test_oxford_index_and_grid_patches(tmp_path):
    g, sat_path = make_oxford_tree(str(tmp_path))
    te = DS.OxfordPairs(g, sat_path, split="test")
    assert len(te) == 9 and te.list_lengths == [3, 3, 3] and te.rows[3][0] == "test2_j_0.png"
    for i in range(9):
        s = te.sample(i)
        assert s["sat_u8"].shape == (800, 800, 3) and s["grd_u8"].shape == (60, 90, 3)
        assert abs(s["center"][0]) <= 128 and abs(s["center"][1]) <= 128       # the vehicle sits in the middle half of a grid patch
        assert 0 <= s["angle_deg"] < 360
    import random
    tr = DS.OxfordPairs(g, sat_path, split="train", rng=random.Random(4))
    offs = [tr.offset() for _ in range(200)]
    assert max(np.hypot(a, b) for a, b in offs) <= 200 * np.sqrt(2) and len(set(offs)) > 150


@pytest.mark.skipif(not reference_available(), reason="reference not present")
def test_oxford_sample_matches_the_reference_dataset_live(tmp_path):
    """The reference's OxfordRobotCarDataset on the same directory (Python's `random` seeded identically on both sides): camera
    image, the 800 x 800 map patch, orientation angle and — through the oracle's restatement with ASCENDING bins — gt, gt_with_ori
    (max-pooled) and the orientation map."""
    import random
    import torch.nn.functional as F
    g, sat_path = make_oxford_tree(str(tmp_path))
    D = import_reference_datasets()

    def to_u8(im):
        return torch.from_numpy(np.array(im.convert("RGB"))).permute(2, 0, 1)

    def to_512(im):          # the reference reads the ground-truth size from the transformed aerial image: give it 512 x 512
        return torch.from_numpy(np.array(im.convert("RGB").resize((512, 512)))).permute(2, 0, 1)
    for split in ("train", "val", "test"):
        ref = D.OxfordRobotCarDataset(g, sat_path, split=split, transform=(to_u8, to_512))
        ours = DS.OxfordPairs(g, sat_path, split=split)
        assert len(ours) == len(ref)
        for i in range(len(ours)):
            random.seed(50 + i)
            grd, sat, gt, gt_with_ori, orientation, angle = ref[i]
            random.seed(50 + i)
            s = ours.sample(i)
            assert abs(s["angle_deg"] - angle) < 1e-9
            assert np.array_equal(s["grd_u8"], grd.permute(1, 2, 0).numpy())
            from PIL import Image
            assert np.array_equal(np.array(Image.fromarray(s["sat_u8"]).resize((512, 512))), sat.permute(1, 2, 0).numpy())
            gg, flat, ori, labs = O.train_targets([list(s["center"])], [s["angle_deg"]], 20, ascending=True)
            assert torch.equal(gg[0], gt) and torch.equal(ori[0], orientation)
            for k, lab in zip((64, 32, 16, 8, 4, 2), labs):
                assert torch.equal(lab[0], F.max_pool2d(gt_with_ori[None], k, stride=k)[0])
