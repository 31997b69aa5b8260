"""Input pipeline after JPEG decoding, on the MI355X: Resize + ToTensor + Normalize + panorama roll +
FoV crop of train_VIGOR.py:57-70,177-178 / datasets.py:98-121 as ccvpe_preprocess_u8_f32 on the decoded uint8 image.

torchvision's `transforms.Resize` on a PIL image is `PIL.Image.resize(size, BILINEAR)`: Pillow's antialiased resampler
in 8-bit fixed point.  `resample_tables` restates its coefficient set-up (src/libImaging/Resample.c: precompute_coeffs()
and normalize_coeffs_8bpc(), Pillow 12.2) in float64, the kernels do the two integer passes; the result equals PIL's
bit for bit (tests/test_preprocess_gpu.py runs PIL itself as the reference)."""
import ctypes
import functools
import math

import numpy as np
import torch

from . import _lib, ops
from ._lib import check

PRECISION_BITS = 32 - 8 - 2
IMAGENET_MEAN = (0.485, 0.456, 0.406)      # train_VIGOR.py:60
IMAGENET_STD = (0.229, 0.224, 0.225)


@functools.lru_cache(maxsize=64)
def resample_tables(in_size, out_size):
    """Pillow's precompute_coeffs() for the BILINEAR filter over the whole axis (box = [0, in_size]) followed by
    normalize_coeffs_8bpc(): returns (bounds int32 [out,2] = (xmin, count), coef int32 [out,ksize], ksize)."""
    scale = float(in_size - 0) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale                       # bilinear support = 1
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coef = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = []
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            a = -a if a < 0.0 else a
            w = 1.0 - a if a < 1.0 else 0.0
            k.append(w)
            ww += w
        for x in range(xmax):
            v = k[x] / ww if ww != 0.0 else k[x]
            coef[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, coef, ksize


_dev_tables = {}


def _tables_on(device, in_size, out_size):
    key = (str(device), in_size, out_size)
    if key not in _dev_tables:
        b, c, k = resample_tables(in_size, out_size)
        _dev_tables[key] = (torch.from_numpy(b).to(device), torch.from_numpy(c).to(device), k)
    return _dev_tables[key]


def preprocess(img_u8, out_hw, dst=None, roll=0, keep_w=None, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """img_u8: decoded image [H,W,3] uint8 on the device.  Writes / returns dst [3, h, keep_w] fp32 (e.g. batch[i]):
    Resize([h, w]) -> ToTensor -> Normalize -> torch.roll(shifts=roll, dims=W) -> [..., :keep_w]."""
    lib = _lib.load()
    if not (img_u8.is_cuda and img_u8.dtype == torch.uint8 and img_u8.is_contiguous() and img_u8.dim() == 3
            and img_u8.shape[2] == 3):
        raise ValueError("img_u8 must be a contiguous [H,W,3] uint8 device tensor (no CPU fallback)")
    h_in, w_in = int(img_u8.shape[0]), int(img_u8.shape[1])
    h, w = out_hw
    keep = w if keep_w is None else int(keep_w)
    if dst is None:
        dst = torch.empty((3, h, keep), device=img_u8.device, dtype=torch.float32)
    ops._chk(dst, "dst")
    if tuple(dst.shape) != (3, h, keep):
        raise ValueError("dst must be [3,%d,%d]" % (h, keep))
    xb, xc, xk = _tables_on(img_u8.device, w_in, w)
    yb, yc, yk = _tables_on(img_u8.device, h_in, h)
    tmp = torch.empty((h_in, w, 3), device=img_u8.device, dtype=torch.uint8)
    m = (ctypes.c_float * 3)(*mean)
    s = (ctypes.c_float * 3)(*std)
    check(lib.ccvpe_preprocess_u8_f32(img_u8.data_ptr(), h_in, w_in, xb.data_ptr(), xc.data_ptr(), xk, yb.data_ptr(),
                                      yc.data_ptr(), yk, tmp.data_ptr(), dst.data_ptr(), h, w, keep, int(roll), m, s,
                                      ops._stream()), "ccvpe_preprocess_u8_f32")
    return dst


def preprocess_batch(images_u8, out_hw, rolls=None, keep_w=None, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """List of decoded [H,W,3] uint8 device images (sizes may differ) -> one NCHW fp32 batch [B,3,h,keep_w], each sample
    written in place by its own pair of launches (no intermediate float image, no host round trip)."""
    h, w = out_hw
    keep = w if keep_w is None else int(keep_w)
    out = torch.empty((len(images_u8), 3, h, keep), device=images_u8[0].device, dtype=torch.float32)
    for i, img in enumerate(images_u8):
        preprocess(img, out_hw, dst=out[i], roll=0 if rolls is None else int(rolls[i]), keep_w=keep, mean=mean, std=std)
    return out
