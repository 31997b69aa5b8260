"""Host-side weight re-packing (ccvpe_amd/models.py) checked on the CPU with plain torch: the packed matrices, applied the
way the kernels' K order and epilogues are documented in include/ccvpe_hip.h, must reproduce the reference operators
(F.conv2d, F.conv_transpose2d and the deconv + cat + 3x3 pair that the upconv kernels fold into one GEMM)."""
import pytest
import torch
import torch.nn.functional as F

from ccvpe_amd import synth
from ccvpe_amd.models import _pack_conv, _pack_deconv, _pack_upconv


def _ours_from_ref(x_ref, col_map, cp):
    """reference channel order [B,Cin,h,w] -> this implementation's NHWC concat order [B,h,w,cp] (pad columns zero)."""
    b, _, h, w = x_ref.shape
    out = x_ref.new_zeros((b, h, w, cp))
    for d0, s0, n in col_map:
        out[..., d0:d0 + n] = x_ref[:, s0:s0 + n].permute(0, 2, 3, 1)
    return out


@pytest.mark.parametrize("cin,col_map,cp", [(17, [(0, 1, 16), (16, 0, 1)], 24),          # loc branch: [max, X] -> [X, max, pad]
                                            (28, [(0, 20, 8), (9, 0, 20)], 32),          # ori level 6: [scores, X] -> [X, max, scores]
                                            (16, [(0, 0, 16)], 16)])
def test_folded_deconv_conv_weights_reproduce_the_pair(cin, col_map, cp):
    b, h, w, cd, c1, co = 2, 5, 6, 12, 8, 10
    wd = synth.normal((cin, cd, 2, 2), 11, 0.3).double()
    bd = synth.normal((cd,), 12, 0.2).double()
    w3 = synth.normal((co, cd + c1, 3, 3), 13, 0.2).double()
    b3 = synth.normal((co,), 14, 0.2).double()
    x_ref = synth.normal((b, cin, h, w), 15).double()
    skip = synth.normal((b, c1, 2 * h, 2 * w), 16).double()
    want = F.conv2d(torch.cat([F.conv_transpose2d(x_ref, wd, bd, stride=2), skip], 1), w3, b3, padding=1)

    fw, shift9 = _pack_upconv(wd.float(), bd.float(), col_map, cp, w3.float(), b3.float())
    fw, shift9 = fw.double(), shift9.double()
    x = _ours_from_ref(x_ref, col_map, cp)
    xp = F.pad(x, (0, 0, 1, 1, 1, 1))                                   # low-res zero halo: index +1
    sp = F.pad(skip.permute(0, 2, 3, 1), (0, 0, 1, 1, 1, 1))            # high-res zero halo
    got = torch.zeros((b, 2 * h, 2 * w, co), dtype=torch.float64)
    for py in range(2):
        for px in range(2):
            wpar = fw[py * 2 + px, :co]
            for y1 in range(h):
                for x1 in range(w):
                    cols = [xp[:, y1 + du - 1 + py + 1, x1 + dv - 1 + px + 1] for du in range(2) for dv in range(2)]
                    yy, xx = 2 * y1 + py, 2 * x1 + px
                    cols += [sp[:, yy + ky - 1 + 1, xx + kx - 1 + 1] for ky in range(3) for kx in range(3)]
                    k = torch.cat(cols, 1)                              # [B, 4*cp + 9*c1]
                    rc = 0 if yy == 0 else (2 if yy == 2 * h - 1 else 1)
                    cc = 0 if xx == 0 else (2 if xx == 2 * w - 1 else 1)
                    got[:, yy, xx] = k @ wpar[:, :k.shape[1]].t() + shift9[rc * 3 + cc]
    err = (got.permute(0, 3, 1, 2) - want).abs().max().item()
    assert err <= 2e-5 * want.abs().max().item(), err


def test_pack_conv_and_pack_deconv_are_the_documented_gemms():
    # conv: rows = output channels, K order (ky, kx, ci)
    w = synth.normal((7, 24, 3, 3), 21).double()
    x = synth.normal((2, 24, 6, 5), 22).double()
    wp = _pack_conv(w.float()).double()
    assert wp.shape[0] % 16 == 0 and wp.shape[1] % 16 == 0
    cols = F.unfold(x, 3, padding=1).reshape(2, 24, 9, -1).permute(0, 3, 2, 1).reshape(2, -1, 9 * 24)     # (ky,kx,ci)
    got = (cols @ wp[:7, :9 * 24].t()).permute(0, 2, 1).reshape(2, 7, 6, 5)
    assert (got - F.conv2d(x, w, padding=1)).abs().max().item() < 1e-5
    assert float(wp[7:].abs().max()) == 0.0 and float(wp[:, 9 * 24:].abs().max()) == 0.0                # zero padding
    # deconv: rows n = (dy*2+dx)*Cout + co, columns in this implementation's input order; output pixel (2y+dy, 2x+dx)
    cin, cout, col_map, ldo = 9, 8, [(0, 1, 8), (8, 0, 1)], 16
    wd = synth.normal((cin, cout, 2, 2), 23).double()
    bd = synth.normal((cout,), 24).double()
    xr = synth.normal((2, cin, 4, 3), 25).double()
    wp, bp = _pack_deconv(wd.float(), bd.float(), col_map, ldo)
    y = _ours_from_ref(xr, col_map, ldo).reshape(2, 12, ldo) @ wp.double()[:4 * cout, :ldo].t() + bp.double()
    got = y.reshape(2, 4, 3, 2, 2, cout).permute(0, 5, 1, 3, 2, 4).reshape(2, cout, 8, 6)
    assert (got - F.conv_transpose2d(xr, wd, bd, stride=2)).abs().max().item() < 1e-5
