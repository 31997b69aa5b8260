"""VIGOR and KITTI input side of the path: dataset indexing, image decoding, iteration — feeding the
device pipeline (ccvpe_amd/preprocess.py: PIL-exact resize + normalise + panorama roll + FoV crop; ccvpe_amd/targets.py:
ground truth from three scalars per sample).

What the reference does per sample on the host (/root/reference/datasets.py:18-177, train_VIGOR.py:57-93): read the VIGOR
split files, PIL-decode the panorama and the aerial tile, torchvision Resize / ToTensor / Normalize, roll the panorama by the
sample's orientation (random, or the fixed per-sample angles of `samearea_orientation_test.npy` /
`crossarea_orientation_test.npy` for the test sets), build a 1 + 20 + 2 channel 512 x 512 ground truth in numpy (24 MB per
sample) and hand everything to a DataLoader.  Here the host keeps only what has to be on the host — the directory index and
the JPEG decode (PIL, like the reference; on a thread pool, PIL releases the GIL while decoding) — and ships the DECODED
uint8 images (3-6 MB per pair) plus three scalars per sample to the device, where the rest runs as kernels.

    pairs = VIGORPairs(root, split="samearea", train=False, ori_noise=180, random_orientation="samearea_orientation_test.npy")
    for batch in DeviceBatches(pairs, batch_size=64, device="cuda", rank=rank, world=world, shard="exact"):   # evaluation
        out = net(batch.grd, batch.sat)                     # batch.gt, batch.gt_flat, batch.gt_ori, batch.labels for training

KITTI (datasets.py:354-640, train_KITTI.py:46-100): `KITTIPairs(root, file, ...)` reads the split file, the OXTS heading and
the two images of a sample and performs the aerial image's geometric alignment — rotate to the vehicle heading, shift to the
camera, the random (training) or listed (test files) shift and rotation, centre crop — with the same PIL calls as the reference
(PIL's affine resampling is host work either way; the result is the 512 x 512 uint8 crop).  Resize + normalise of both images
and the 1 + 16 + 2 channel ground truth then run on the device exactly as for VIGOR:

    pairs = KITTIPairs(root, "train_files.txt", shift_range_lat=20, shift_range_lon=20, rotation_range=10)
    for batch in DeviceBatches(pairs, 64, grd_hw=(256, 1024), n_bins=16, rank=rank, world=world): ...

Oxford RobotCar (datasets.py:183-353): `OxfordPairs(grd_image_root, sat_path, split)` — list files + yaw fixtures, the UTM ->
map-pixel fit, an 800 x 800 patch of the one large aerial map per sample; `DeviceBatches(pairs, B, grd_hw=(154, 231),
ascending_bins=True)`.

Sharding: DeviceBatches gives rank r a contiguous shard of the index list (after the optional seeded shuffle, identical on
every rank).  Evaluation (`shard="exact"`): harness.shard_indices — no sample is seen twice, no collective is needed, ranks
may differ by one batch.  Training (`shard="pad"`, the default for a training split and world > 1): every rank iterates the SAME
number of batches (short shards wrap around, as torch's DistributedSampler pads), because each step all-reduces gradients.
A sample's random choices (`pairs.draw(i)`) are drawn in index order on the producer thread, decoding runs on the pool.
"""
import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

CITIES = {("samearea", True): ("NewYork", "Seattle", "SanFrancisco", "Chicago"),
          ("samearea", False): ("NewYork", "Seattle", "SanFrancisco", "Chicago"),
          ("crossarea", True): ("NewYork", "Seattle"),                    # datasets.py:32-38
          ("crossarea", False): ("SanFrancisco", "Chicago")}


def _label_file(split, train):
    if split == "samearea":
        return "same_area_balanced_train.txt" if train else "same_area_balanced_test.txt"    # datasets.py:66-70
    return "pano_label_balanced.txt"                                                          # datasets.py:71-72


class VIGORPairs(object):
    """Host-side index of a VIGOR split + per-sample decode.  `sample(i)` returns the DECODED images and the scalars the
    device pipeline needs; nothing here touches the GPU."""

    def __init__(self, root, label_root="splits_new", split="samearea", train=True, pos_only=True, ori_noise=180.0,
                 random_orientation=None, seed=0, strict_orientation=True):
        if split not in ("samearea", "crossarea"):
            raise ValueError("split must be 'samearea' or 'crossarea'")
        self.root, self.split, self.train, self.pos_only = root, split, bool(train), bool(pos_only)
        self.ori_noise = float(ori_noise)
        self.cities = CITIES[(split, self.train)]
        if isinstance(random_orientation, str):                   # the reference's .npy fixtures (train_VIGOR.py:73-79)
            random_orientation = np.load(random_orientation)
        self.random_orientation = None if random_orientation is None else np.asarray(random_orientation, dtype=np.float64)
        self._rng = np.random.default_rng(seed)
        self._rng_lock = threading.Lock()

        # aerial tiles: one global index over the cities of the split, in file order (datasets.py:41-55)
        sat_paths, sat_index = [], {}
        for city in self.cities:
            with open(os.path.join(root, label_root, city, "satellite_list.txt")) as f:
                for line in f:
                    name = line.rstrip("\n")
                    sat_index[name] = len(sat_paths)
                    sat_paths.append(os.path.join(root, city, "satellite", name))
        self.sat_paths = sat_paths
        # panoramas: "<pano> <sat0> <drow0> <dcol0> <sat1> ... <sat3> <drow3> <dcol3>" (positive first, then 3 semi-positives)
        grd_paths, labels, deltas, city_of = [], [], [], []
        for city in self.cities:
            path = os.path.join(root, label_root, city, _label_file(split, self.train))
            with open(path) as f:
                for lineno, line in enumerate(f, 1):
                    tok = line.split(" ")
                    if len(tok) < 13:           # the reference indexes tok[1..12] and raises here; a skipped line would shift
                        raise ValueError("%s:%d: expected 13 fields, got %d" % (path, lineno, len(tok)))   # every later fixture entry
                    grd_paths.append(os.path.join(root, city, "panorama", tok[0]))
                    labels.append([sat_index[tok[i]] for i in (1, 4, 7, 10)])
                    deltas.append([[float(tok[i]), float(tok[i + 1])] for i in (2, 5, 8, 11)])
                    city_of.append(city)
        self.grd_paths, self.city_of = grd_paths, city_of
        self.labels = np.asarray(labels, dtype=np.int64).reshape(-1, 4)
        self.deltas = np.asarray(deltas, dtype=np.float64).reshape(-1, 4, 2)       # (row, col) offsets in raw aerial pixels
        # the fixture is indexed by sample position: it must describe exactly this index (strict_orientation=False accepts a
        # longer array, as the reference silently does — used by the tests that pair the fixture's head with a small tree)
        n_ori = None if self.random_orientation is None else len(self.random_orientation)
        if n_ori is not None and (n_ori < len(grd_paths) or (strict_orientation and n_ori != len(grd_paths))):
            raise ValueError("random_orientation has %d entries for %d samples" % (n_ori, len(grd_paths)))

    def __len__(self):
        return len(self.grd_paths)

    # ---- the sample's three random choices (datasets.py:107-116,124-135) ------------------------------------------------
    def rotation_fraction(self, idx):
        """Panorama roll as a fraction of its width; orientation angle = 360 * fraction (north = 0, counter-clockwise)."""
        if self.random_orientation is not None:
            return float(self.random_orientation[idx]) / 360.0
        with self._rng_lock:
            if self.ori_noise >= 180:
                return float(self._rng.uniform(0.0, 1.0))
            r = self.ori_noise / 360.0
            return float(self._rng.uniform(-r, r))

    def positive(self, idx):
        """(aerial tile index among the sample's four, row offset, column offset) in raw aerial pixels: the positive, or —
        pos_only=False — a random one of the four whose ground-truth location lies inside the tile."""
        if self.pos_only:
            return 0, self.deltas[idx, 0, 0], self.deltas[idx, 0, 1]
        ok = [k for k in range(4) if abs(self.deltas[idx, k, 0]) < 320 and abs(self.deltas[idx, k, 1]) < 320]
        if not ok:
            raise ValueError("sample %d has no tile with the ground truth inside it" % idx)
        with self._rng_lock:
            k = int(ok[self._rng.integers(len(ok))])
        return k, self.deltas[idx, k, 0], self.deltas[idx, k, 1]

    def draw(self, idx):
        """The sample's random choices, in the order sample() makes them.  DeviceBatches calls this on ONE thread in index order
        and hands the result to sample(draws=...), so a seeded iteration is reproducible whatever the decode threads do."""
        return self.rotation_fraction(idx), self.positive(idx)

    def sample(self, idx, sat_hw=(512, 512), grd_hw=(320, 640), draws=None):
        """Decode sample idx.  Returns a dict: grd_u8 / sat_u8 (HWC uint8 numpy, as PIL decodes them), roll (pixels of the
        RESIZED panorama, datasets.py:121), angle_deg, center (cx, cy) of the Gaussian on the resized aerial grid
        (cx = column offset, cy = -row offset, both rescaled and rounded as datasets.py:141-142 does), city."""
        from PIL import Image
        try:
            with Image.open(self.grd_paths[idx]) as im:
                grd = np.array(im.convert("RGB"))             # owning, writable: torch.from_numpy needs that
        except (OSError, ValueError):                                  # unreadable panorama -> blank image (datasets.py:103-105)
            grd = np.zeros((grd_hw[0], grd_hw[1], 3), dtype=np.uint8)
        rot, (k, drow, dcol) = self.draw(idx) if draws is None else draws
        with Image.open(self.sat_paths[self.labels[idx, k]]) as im:
            sat = np.array(im.convert("RGB"))
        h_raw, w_raw = sat.shape[0], sat.shape[1]
        row = np.round(drow / h_raw * sat_hw[0])
        col = np.round(dcol / w_raw * sat_hw[1])
        # torch.round (half to even) of rotation * width, like datasets.py:121
        roll = int(torch.round(torch.as_tensor(rot) * grd_hw[1]).int().item())
        return dict(grd_u8=grd, sat_u8=sat, roll=roll, angle_deg=rot * 360.0,
                    center=(float(col), float(-row)), city=self.city_of[idx], index=int(idx))


# ---- KITTI (datasets.py:354-372: constants of the reference's loader) ---------------------------------------------------
KITTI_LAT, KITTI_ZOOM = 49.015, 18
KITTI_CAMERA_SHIFT = (1.08, 0.26)          # metres: GPS antenna -> left colour camera (datasets.py:366)
KITTI_CROP = 512


def kitti_meter_per_pixel(lat=KITTI_LAT, zoom=KITTI_ZOOM, scale=1.0):
    """Ground resolution of the aerial maps (datasets.py:368-372: web-mercator resolution at `lat`, maps fetched at scale 2)."""
    return 156543.03392 * np.cos(lat * np.pi / 180.0) / (2 ** zoom) / 2.0 / scale


class KITTIPairs(object):
    """Host-side index of a KITTI split file + per-sample decode and aerial alignment (datasets.py:374-503 `SatGrdDataset`,
    :506-640 `SatGrdDatasetTest`).  A line of a TRAINING file is `<day>/<drive>/<frame>.png`; a line of a TEST file carries the
    sample's fixed perturbation: `<name> <shift_x> <shift_y> <theta>` (test=True).  Training perturbations are three uniform
    draws in [-1, 1) per sample — from numpy's global generator when rng is None (so `np.random.seed(s)` reproduces the
    reference's sequence sample by sample when samples are drawn in order: sample() called sequentially, or DeviceBatches, which
    calls draw() in index order on one thread), else from `rng` (a numpy Generator / RandomState; draws are serialised by a lock)."""

    def __init__(self, root, file, shift_range_lat=20.0, shift_range_lon=20.0, rotation_range=10.0, test=False, rng=None):
        self.root, self.test = root, bool(test)
        self.meter_per_pixel = kitti_meter_per_pixel()
        self.shift_px_lat = shift_range_lat / self.meter_per_pixel
        self.shift_px_lon = shift_range_lon / self.meter_per_pixel
        self.rotation_range = float(rotation_range)
        with open(file) as f:
            self.lines = [ln[:-1] for ln in f.readlines()]                    # datasets.py:399 (drops the line's last character)
        self._rng, self._rng_lock = rng, threading.Lock()

    def __len__(self):
        return len(self.lines)

    def paths(self, name):
        """(aerial map, OXTS record, left colour image) of `<day>/<drive>/<frame>.png` (datasets.py:410-435)."""
        drive, frame = name[:38], name[38:]
        base = os.path.join(self.root, "raw_data", drive)
        return (os.path.join(self.root, "satmap", name),
                os.path.join(base, "oxts/data", frame.lower().replace(".png", ".txt")),
                os.path.join(base, "image_02/data", frame.lower()))

    def perturbation(self, idx):
        """(shift_x, shift_y) in units of the shift range (right / up positive, along / across the heading) and the rotation in
        degrees (counter-clockwise; 0 = the vehicle heads east in the aerial image)."""
        if self.test:
            _, sx, sy, th = self.lines[idx].split(" ")
            return -float(sx), -float(sy), float(th) * self.rotation_range          # datasets.py:585-595
        with self._rng_lock:
            u = np.random.uniform if self._rng is None else self._rng.uniform
            sx, sy, r = u(-1, 1), u(-1, 1), u(-1, 1)                                # datasets.py:452-453,463 (this order)
        return float(sx), float(sy), float(r) * self.rotation_range

    def draw(self, idx):
        """The sample's random choices (see VIGORPairs.draw)."""
        return self.perturbation(idx)

    def sample(self, idx, sat_hw=(512, 512), grd_hw=(256, 1024), draws=None):
        """Decode + align sample idx.  Returns the dict DeviceBatches consumes: grd_u8 (the left camera image as decoded),
        sat_u8 (the aligned, perturbed 512 x 512 aerial crop), roll = 0, angle_deg (orientation ground truth in [0, 360]),
        center = (x_offset, y_offset) of the Gaussian (datasets.py:475-476), city = the drive, index."""
        from PIL import Image
        name = self.lines[idx].split(" ")[0] if self.test else self.lines[idx]
        sat_path, oxts_path, grd_path = self.paths(name)
        with Image.open(sat_path, "r") as im:
            sat = im.convert("RGB")
        with open(oxts_path) as f:
            heading = float(f.readline().split(" ")[5])
        with Image.open(grd_path, "r") as im:
            grd = np.array(im.convert("RGB"))
        sx, sy, ori = self.perturbation(idx) if draws is None else draws
        mpp = self.meter_per_pixel
        sat = sat.rotate(-heading / np.pi * 180)                                    # east = the vehicle heading
        sat = sat.transform(sat.size, Image.AFFINE, (1, 0, KITTI_CAMERA_SHIFT[0] / mpp, 0, 1, KITTI_CAMERA_SHIFT[1] / mpp),
                            resample=Image.BILINEAR)
        sat = sat.transform(sat.size, Image.AFFINE, (1, 0, sx * self.shift_px_lon, 0, 1, -sy * self.shift_px_lat),
                            resample=Image.BILINEAR)
        sat = sat.rotate(ori)
        w, h = sat.size
        if w < KITTI_CROP or h < KITTI_CROP:
            raise ValueError("%s: aerial map %dx%d is smaller than the %d crop" % (sat_path, w, h, KITTI_CROP))
        left, top = int(round((w - KITTI_CROP) / 2.0)), int(round((h - KITTI_CROP) / 2.0))     # torchvision center_crop
        sat = np.array(sat.crop((left, top, left + KITTI_CROP, top + KITTI_CROP)))
        c, s_ = np.cos(ori / 180 * np.pi), np.sin(ori / 180 * np.pi)
        x_off = int(sx * self.shift_px_lon * c - sy * self.shift_px_lat * s_)       # int(): truncation, as the reference
        y_off = int(-sy * self.shift_px_lat * c - sx * self.shift_px_lon * s_)
        angle = 90 - ori
        if angle < 0:
            angle += 360
        elif angle > 360:
            angle -= 360
        return dict(grd_u8=grd, sat_u8=sat, roll=0, angle_deg=float(angle), center=(float(x_off), float(y_off)),
                    city=name[:38], index=int(idx))


# ---- Oxford RobotCar (datasets.py:183-353, train_OxfordRobotCar.py:49-72) ---------------------------------------------
# Ground control points of the reference's aerial map: UTM (easting, northing) -> (column, row) of `satellite_map_new.png`
# (datasets.py:252-263: four corners + centre; the affine map is their least-squares fit).
OXFORD_UTM = ((619400., 5736195.), (619400., 5734600.), (620795., 5736195.), (620795., 5734600.), (620100., 5735400.))
OXFORD_PIX = ((900., 900.), (492., 18168.), (15966., 1260.), (15553., 18528.), (8255., 9688.))
OXFORD_LISTS = {"train": (("training.txt",), "train_yaw.npy"), "val": (("validation.txt",), "val_yaw.npy"),
                "test": (("test1_j.txt", "test2_j.txt", "test3_j.txt"), "test_yaw.npy")}


class OxfordPairs(object):
    """Host-side index of an Oxford RobotCar split + per-sample decode: `<image> <...> <easting> <northing>` lines, the yaw
    fixtures (`*_yaw.npy`, radians, 0 = west, clockwise), ONE large aerial map from which an 800 x 800 patch is cut per sample —
    around the vehicle plus a random offset of up to 200 * sqrt(2) map pixels (train: two draws from Python's `random` module,
    or from `rng`, a `random.Random`), or the half-overlapping grid patch that holds the vehicle at least 200 pixels from its
    border (val / test).  The Gaussian's centre is the vehicle's position in the 512 x 512 resized patch; the orientation bins of
    this loader count UP (DeviceBatches(..., ascending_bins=True))."""

    def __init__(self, grd_image_root, sat_path, split="train", rng=None):
        from PIL import Image
        if split not in OXFORD_LISTS:
            raise ValueError("split must be 'train', 'val' or 'test'")
        self.root, self.split = grd_image_root, split
        Image.MAX_IMAGE_PIXELS = None                                  # the real map is 16 k x 19 k pixels
        self.map = Image.open(sat_path)
        self.map.load()
        files, yaw = OXFORD_LISTS[split]
        self.rows, self.list_lengths = [], []
        for fn in files:
            with open(os.path.join(grd_image_root, fn)) as f:
                rows = [ln[:-1].split(" ") for ln in f.readlines()]
            self.list_lengths.append(len(rows))
            self.rows += rows
        self.yaw = np.load(os.path.join(grd_image_root, yaw))
        if len(self.yaw) < len(self.rows):
            raise ValueError("%s has %d entries for %d samples" % (yaw, len(self.yaw), len(self.rows)))
        utm = np.asarray([r[2:] for r in self.rows], dtype=np.float64)            # [n, 2]: easting, northing
        src = np.hstack([np.asarray(OXFORD_UTM), np.ones((5, 1))])
        dst = np.hstack([np.asarray(OXFORD_PIX), np.ones((5, 1))])
        self.utm = utm
        self.affine = np.linalg.lstsq(src, dst, rcond=None)[0]                   # 3 x 3; its last column maps the padding 1 to 1
        self._rng, self._rng_lock = rng, threading.Lock()

    def __len__(self):
        return len(self.rows)

    def offset(self):
        """Training: the (row, column) offset of the patch centre from the vehicle, in map pixels (datasets.py:296-300)."""
        import math
        import random
        with self._rng_lock:
            u = random.random if self._rng is None else self._rng.random
            alpha, r = 2 * math.pi * u(), 200 * np.sqrt(2) * u()
        return int(r * math.cos(alpha)), int(r * math.sin(alpha))

    def draw(self, idx):
        """The sample's random choice (see VIGORPairs.draw): the patch offset in training, nothing otherwise."""
        return self.offset() if self.split == "train" else None

    def sample(self, idx, sat_hw=(512, 512), grd_hw=(154, 231), draws=None):
        from PIL import Image
        with Image.open(os.path.join(self.root, self.rows[idx][0])) as im:
            grd = np.array(im.convert("RGB"))
        # one sample at a time, as the reference evaluates it (a batched product may round differently in the last place)
        col, row = (float(v) for v in np.dot(np.hstack([self.utm[idx:idx + 1], np.ones((1, 1))]), self.affine)[0, :2])
        if self.split == "train":
            drow, dcol = self.offset() if draws is None else draws
            r0, c0 = int(row + drow), int(col + dcol)
            box = (c0 - 400, r0 - 400, c0 + 400, r0 + 400)
            cy = int(np.round((400 + drow) / 800 * 512 - 256))
            cx = int(np.round((400 + dcol) / 800 * 512 - 256))
        else:
            def grid(v):                    # patch origin = a multiple of 400 such that the vehicle sits in [200, 600) of the patch
                k = int(v // 400)
                if np.round(v - 400 * k) < 200:
                    k -= 1
                return k, int(np.round(v - 400 * k))
            kc, pc = grid(col)
            kr, pr = grid(row)
            box = (kc * 400, kr * 400, kc * 400 + 800, kr * 400 + 800)
            cy = int(-(pr / 800 * 512 - 256))
            cx = int(-(pc / 800 * 512 - 256))
        sat = np.array(self.map.crop(box).convert("RGB"))
        angle = float(self.yaw[idx]) / np.pi * 180 - 90                          # 0 = north, clockwise (datasets.py:334-337)
        if angle < 0:
            angle += 360
        return dict(grd_u8=grd, sat_u8=sat, roll=0, angle_deg=float(angle), center=(float(cx), float(cy)), city="oxford",
                    index=int(idx))


def shard_positions(n, world, rank, mode="exact"):
    """Positions of range(n) that rank `rank` of `world` iterates (see DeviceBatches: "exact" / "pad" / "truncate")."""
    from . import harness
    if mode == "exact":
        return np.asarray(harness.shard_indices(n, world, rank), dtype=np.int64)
    if mode == "truncate":
        per = n // world
        return np.arange(rank * per, (rank + 1) * per, dtype=np.int64)
    if mode == "pad":
        per = (n + world - 1) // world
        if n == 0:
            return np.zeros((0,), dtype=np.int64)
        return np.arange(rank * per, (rank + 1) * per, dtype=np.int64) % n
    raise ValueError("shard must be 'exact', 'pad' or 'truncate'")


def pairs_is_training(pairs):
    """True / False when the index object says which split it holds (VIGORPairs.train, KITTIPairs.test, OxfordPairs.split),
    None otherwise.  DeviceBatches keys its default shard mode on it: an EVALUATION set must never be padded (wrapped-around
    samples would be counted twice in the gathered metrics), whatever `targets` is."""
    if isinstance(getattr(pairs, "train", None), bool):
        return pairs.train
    if isinstance(getattr(pairs, "test", None), bool):
        return not pairs.test
    if isinstance(getattr(pairs, "split", None), str):
        return pairs.split == "train"
    return None


class Batch(object):
    """One device batch: grd [B,3,h,w_fov], sat [B,3,H,W] (normalised fp32), angle_deg [B], center [B,2], cities, indices and —
    with targets — gt [B,1,H,W], gt_flat [B,H*W], gt_ori [B,2,H,W], labels (six max-pooled orientation-binned maps)."""
    __slots__ = ("grd", "sat", "angle_deg", "center", "cities", "indices", "gt", "gt_flat", "gt_ori", "labels")


class DeviceBatches(object):
    """Iterates a VIGORPairs / KITTIPairs / OxfordPairs index in device batches (KITTI: grd_hw=(256, 1024), n_bins=16; Oxford:
    grd_hw=(154, 231), ascending_bins=True).  Decoding runs `workers` samples ahead on a thread pool; resize,
    normalisation, roll, FoV crop and the ground truth are kernels on `device` (no CPU fallback: preprocess raises without
    the HIP library).  fov < 360 keeps the first fov/360 of the rolled panorama's columns (train_VIGOR.py:177-178).

    Rank shards (`shard`): "exact" = the balanced contiguous shards of harness.shard_indices — every sample exactly once over the
    ranks, shard sizes differ by at most one, so ranks may run a DIFFERENT number of batches: evaluation only (no collective).
    "pad" = every rank gets ceil(n / world) samples, the short shards wrap around to the start of the index list
    (torch's DistributedSampler rule); "truncate" = every rank gets floor(n / world).  With "pad" / "truncate" len(self) is
    the same on every rank BY CONSTRUCTION — required for training, where each step all-reduces gradients inside the
    backward and a rank with one batch more would wait for its peers until the RCCL timeout.  Default: "pad" for a TRAINING
    split (`pairs_is_training`; an index object that does not say: when targets are produced) and world > 1, else "exact" —
    a test split is never padded by default, with or without targets."""

    def __init__(self, pairs, batch_size, device="cuda", indices=None, shuffle=False, seed=0, rank=0, world=1, workers=8,
                 prefetch=2, grd_hw=(320, 640), sat_hw=(512, 512), fov=360, targets=True, n_bins=20, drop_last=False,
                 ascending_bins=False, shard=None):
        self.pairs, self.batch_size, self.device = pairs, int(batch_size), torch.device(device)
        idx = np.arange(len(pairs)) if indices is None else np.asarray(indices)
        if shuffle:
            idx = idx[np.random.default_rng(seed).permutation(len(idx))]      # same permutation on every rank
        if shard is None:
            training = pairs_is_training(pairs)
            if training is None:                      # an index object that does not say: ground truth requested = training
                training = bool(targets)
            shard = "pad" if (training and world > 1) else "exact"
        self.shard = shard
        self.indices = idx[shard_positions(len(idx), world, rank, shard)]
        self.workers, self.prefetch = int(workers), int(prefetch)
        self.grd_hw, self.sat_hw, self.fov = tuple(grd_hw), tuple(sat_hw), fov
        self.keep_w = int(grd_hw[1] * fov / 360)                              # train_VIGOR.py:177: int(w * FoV / 360)
        self.targets, self.n_bins, self.drop_last, self.ascending_bins = targets, n_bins, drop_last, bool(ascending_bins)

    def __len__(self):
        n = len(self.indices)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _decode_batch(self, pool, chunk):
        # the random choices of the chunk are drawn HERE, on the producer thread, in index order; the pool only decodes
        draws = [self.pairs.draw(int(i)) for i in chunk]
        return list(pool.map(lambda a: self.pairs.sample(int(a[0]), self.sat_hw, self.grd_hw, draws=a[1]), zip(chunk, draws)))

    def _to_device(self, samples):
        from . import preprocess, targets
        dev = self.device
        b = len(samples)
        out = Batch()
        out.grd = torch.empty((b, 3, self.grd_hw[0], self.keep_w), device=dev, dtype=torch.float32)
        out.sat = torch.empty((b, 3) + self.sat_hw, device=dev, dtype=torch.float32)
        for i, s in enumerate(samples):
            g = torch.from_numpy(s["grd_u8"]).to(dev, non_blocking=True)
            a = torch.from_numpy(s["sat_u8"]).to(dev, non_blocking=True)
            preprocess.preprocess(g, self.grd_hw, dst=out.grd[i], roll=s["roll"], keep_w=self.keep_w)
            preprocess.preprocess(a, self.sat_hw, dst=out.sat[i])
        out.angle_deg = torch.tensor([s["angle_deg"] for s in samples], device=dev, dtype=torch.float32)
        out.center = torch.tensor([s["center"] for s in samples], device=dev, dtype=torch.float32)
        out.cities = [s["city"] for s in samples]
        out.indices = [s["index"] for s in samples]
        out.gt = out.gt_flat = out.gt_ori = out.labels = None
        if self.targets:
            out.gt, out.gt_flat, out.gt_ori, out.labels = targets.train_targets(out.center, out.angle_deg % 360.0, self.n_bins,
                                                                               self.sat_hw[0], self.sat_hw[1],
                                                                               ascending=self.ascending_bins)
        return out

    def __iter__(self):
        chunks = [self.indices[i:i + self.batch_size] for i in range(0, len(self.indices), self.batch_size)]
        if self.drop_last and chunks and len(chunks[-1]) < self.batch_size:
            chunks.pop()
        q = queue.Queue(maxsize=max(1, self.prefetch))
        stop = threading.Event()

        def producer():
            try:
                with ThreadPoolExecutor(max_workers=max(1, self.workers)) as pool:
                    for chunk in chunks:
                        if stop.is_set():
                            return
                        q.put(("ok", self._decode_batch(pool, chunk)))
                q.put(("end", None))
            except BaseException as ex:                 # noqa: BLE001 — hand the failure to the consumer
                q.put(("err", ex))

        t = threading.Thread(target=producer, daemon=True)
        t.start()
        try:
            while True:
                kind, payload = q.get()
                if kind == "end":
                    return
                if kind == "err":
                    raise payload
                yield self._to_device(payload)
        finally:
            stop.set()
            while not q.empty():                       # unblock a producer waiting on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    break
