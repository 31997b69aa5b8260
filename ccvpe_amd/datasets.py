"""VIGOR input side of the path: dataset indexing, JPEG decoding, iteration — feeding the
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
    for batch in DeviceBatches(pairs, batch_size=64, device="cuda", rank=rank, world=world):
        out = net(batch.grd, batch.sat)                     # batch.gt, batch.gt_flat, batch.gt_ori, batch.labels for training

Sharding: DeviceBatches gives rank r the contiguous shard harness.shard_indices(len, world, r) of the index list (after the
optional seeded shuffle, identical on every rank): no sample is seen twice, no collective is needed.
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
                 random_orientation=None, seed=0):
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
            with open(os.path.join(root, label_root, city, _label_file(split, self.train))) as f:
                for line in f:
                    tok = line.split(" ")
                    if len(tok) < 13:
                        continue
                    grd_paths.append(os.path.join(root, city, "panorama", tok[0]))
                    labels.append([sat_index[tok[i]] for i in (1, 4, 7, 10)])
                    deltas.append([[float(tok[i]), float(tok[i + 1])] for i in (2, 5, 8, 11)])
                    city_of.append(city)
        self.grd_paths, self.city_of = grd_paths, city_of
        self.labels = np.asarray(labels, dtype=np.int64).reshape(-1, 4)
        self.deltas = np.asarray(deltas, dtype=np.float64).reshape(-1, 4, 2)       # (row, col) offsets in raw aerial pixels
        if self.random_orientation is not None and len(self.random_orientation) < len(grd_paths):
            raise ValueError("random_orientation has %d entries for %d samples" % (len(self.random_orientation), len(grd_paths)))

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

    def sample(self, idx, sat_hw=(512, 512), grd_hw=(320, 640)):
        """Decode sample idx.  Returns a dict: grd_u8 / sat_u8 (HWC uint8 numpy, as PIL decodes them), roll (pixels of the
        RESIZED panorama, datasets.py:121), angle_deg, center (cx, cy) of the Gaussian on the resized aerial grid
        (cx = column offset, cy = -row offset, both rescaled and rounded as datasets.py:141-142 does), city."""
        from PIL import Image
        try:
            with Image.open(self.grd_paths[idx]) as im:
                grd = np.array(im.convert("RGB"))             # owning, writable: torch.from_numpy needs that
        except (OSError, ValueError):                                  # unreadable panorama -> blank image (datasets.py:103-105)
            grd = np.zeros((grd_hw[0], grd_hw[1], 3), dtype=np.uint8)
        rot = self.rotation_fraction(idx)
        k, drow, dcol = self.positive(idx)
        with Image.open(self.sat_paths[self.labels[idx, k]]) as im:
            sat = np.array(im.convert("RGB"))
        h_raw, w_raw = sat.shape[0], sat.shape[1]
        row = np.round(drow / h_raw * sat_hw[0])
        col = np.round(dcol / w_raw * sat_hw[1])
        # torch.round (half to even) of rotation * width, like datasets.py:121
        roll = int(torch.round(torch.as_tensor(rot) * grd_hw[1]).int().item())
        return dict(grd_u8=grd, sat_u8=sat, roll=roll, angle_deg=rot * 360.0,
                    center=(float(col), float(-row)), city=self.city_of[idx], index=int(idx))


class Batch(object):
    """One device batch: grd [B,3,h,w_fov], sat [B,3,H,W] (normalised fp32), angle_deg [B], center [B,2], cities, indices and —
    with targets — gt [B,1,H,W], gt_flat [B,H*W], gt_ori [B,2,H,W], labels (six max-pooled orientation-binned maps)."""
    __slots__ = ("grd", "sat", "angle_deg", "center", "cities", "indices", "gt", "gt_flat", "gt_ori", "labels")


class DeviceBatches(object):
    """Iterates a VIGORPairs index in device batches.  Decoding runs `workers` samples ahead on a thread pool; resize,
    normalisation, roll, FoV crop and the ground truth are kernels on `device` (no CPU fallback: preprocess raises without
    the HIP library).  fov < 360 keeps the first fov/360 of the rolled panorama's columns (train_VIGOR.py:177-178)."""

    def __init__(self, pairs, batch_size, device="cuda", indices=None, shuffle=False, seed=0, rank=0, world=1, workers=8,
                 prefetch=2, grd_hw=(320, 640), sat_hw=(512, 512), fov=360, targets=True, n_bins=20, drop_last=False):
        from . import harness
        self.pairs, self.batch_size, self.device = pairs, int(batch_size), torch.device(device)
        idx = np.arange(len(pairs)) if indices is None else np.asarray(indices)
        if shuffle:
            idx = idx[np.random.default_rng(seed).permutation(len(idx))]      # same permutation on every rank
        mine = harness.shard_indices(len(idx), world, rank)
        self.indices = idx[mine]
        self.workers, self.prefetch = int(workers), int(prefetch)
        self.grd_hw, self.sat_hw, self.fov = tuple(grd_hw), tuple(sat_hw), fov
        self.keep_w = int(fov / 360 * grd_hw[1])
        self.targets, self.n_bins, self.drop_last = targets, n_bins, drop_last

    def __len__(self):
        n = len(self.indices)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _decode_batch(self, pool, chunk):
        return list(pool.map(lambda i: self.pairs.sample(int(i), self.sat_hw, self.grd_hw), chunk))

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
                                                                               self.sat_hw[0], self.sat_hw[1])
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
