"""Input pipeline.  CPU part: the host-side restatement of Pillow's coefficient set-up, applied
with numpy integer arithmetic, against PIL itself (bit-exact).  GPU part: the HIP kernels against PIL + torchvision's
ToTensor / Normalize arithmetic + torch.roll + FoV crop (bit-exact on the uint8 resample, fp32-exact after)."""
import numpy as np
import pytest
import torch

from ccvpe_amd import preprocess as P
from ccvpe_amd import synth
from oracle import ccvpe_oracle as O


def _image(h, w, seed):
    # smooth structure + noise so that antialiasing and rounding are both exercised
    yy, xx = np.mgrid[0:h, 0:w]
    base = 127 + 90 * np.sin(xx / 17.0 + seed) * np.cos(yy / 11.0) + 30 * synth.normal((h, w), seed).numpy()
    img = np.stack([base, base[::-1], 255 - base], axis=2)
    return np.clip(img, 0, 255).astype(np.uint8)


def _numpy_resample(img, out_hw):
    """Two integer passes with resample_tables(): what csrc/preprocess.hip computes."""
    h, w = out_hw
    xb, xc, _ = P.resample_tables(img.shape[1], w)
    yb, yc, _ = P.resample_tables(img.shape[0], h)
    half = 1 << (P.PRECISION_BITS - 1)
    tmp = np.empty((img.shape[0], w, 3), dtype=np.uint8)
    src = img.astype(np.int64)
    for xx in range(w):
        x0, n = xb[xx]
        acc = half + (src[:, x0:x0 + n, :] * xc[xx, :n].astype(np.int64)[None, :, None]).sum(1)
        tmp[:, xx, :] = np.clip(acc >> P.PRECISION_BITS, 0, 255)
    out = np.empty((h, w, 3), dtype=np.uint8)
    t64 = tmp.astype(np.int64)
    for yy in range(h):
        y0, n = yb[yy]
        acc = half + (t64[y0:y0 + n] * yc[yy, :n].astype(np.int64)[:, None, None]).sum(0)
        out[yy] = np.clip(acc >> P.PRECISION_BITS, 0, 255)
    return out


@pytest.mark.parametrize("in_hw,out_hw", [((1024, 2048), (320, 640)), ((640, 640), (512, 512)), ((375, 1242), (256, 1024)),
                                          ((100, 150), (154, 231)), ((320, 640), (320, 640))])
def test_resample_tables_reproduce_pillow_bit_for_bit(in_hw, out_hw):
    from PIL import Image
    img = _image(in_hw[0], in_hw[1], 3)
    want = np.asarray(Image.fromarray(img, "RGB").resize((out_hw[1], out_hw[0]), Image.BILINEAR))
    got = _numpy_resample(img, out_hw)
    assert np.array_equal(got, want), "max diff %d" % np.abs(got.astype(int) - want.astype(int)).max()


@pytest.mark.gpu
@pytest.mark.parametrize("in_hw,out_hw,roll,keep", [((1024, 2048), (320, 640), 0, None), ((1024, 2048), (320, 640), 213, None),
                                                   ((1024, 2048), (320, 640), -77, 320), ((640, 640), (512, 512), 0, None),
                                                   ((375, 1242), (256, 1024), 0, None), ((480, 640), (154, 231), 0, None)])
def test_preprocess_kernels_match_pil_pipeline(in_hw, out_hw, roll, keep):
    img = _image(in_hw[0], in_hw[1], 5)
    want = O.preprocess_reference(img, out_hw, roll, keep)
    batch = torch.zeros((2, 3, out_hw[0], keep or out_hw[1]), device="cuda")
    got = P.preprocess(torch.from_numpy(img).cuda(), out_hw, dst=batch[1], roll=roll, keep_w=keep)
    torch.cuda.synchronize()
    assert got.data_ptr() == batch[1].data_ptr()
    assert torch.equal(batch[1].cpu(), want), "max diff %g" % (batch[1].cpu() - want).abs().max().item()
    assert float(batch[0].abs().max()) == 0.0
