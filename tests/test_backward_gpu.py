"""Backward building blocks (ccvpe_amd/backward.py) against torch autograd on the CPU (fp32).
Tolerance 2e-4 of the gradient scale (the weight gradient reduces over thousands of pixels)."""
import pytest
import torch
import torch.nn.functional as F

from ccvpe_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bw():
    from ccvpe_amd import backward, _lib
    _lib.load()
    return backward


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def close(got, want, tol, what):
    got, want = got.detach().cpu().double(), want.detach().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = want.abs().max().item() + 1e-30
    err = (got - want).abs().max().item()
    assert err <= tol * scale, "%s: max err %.3e vs scale %.3e" % (what, err, scale)


@pytest.mark.parametrize("c0,c1,n,k,stride,pad,h,w", [(40, 16, 40, 3, 1, 1, 12, 14), (96, 0, 24, 1, 1, 0, 9, 13),
                                                       (16, 0, 96, 1, 1, 0, 17, 11), (64, 0, 48, 2, 2, 0, 8, 8),
                                                       (320, 112, 320, 3, 1, 1, 6, 6), (16, 0, 16, 3, 1, 1, 33, 20),
                                                       (1024, 320, 640, 3, 1, 1, 4, 4), (1280, 0, 126, 1, 1, 0, 10, 20)])
def test_conv_wgrad_and_dgrad_vs_autograd(bw, c0, c1, n, k, stride, pad, h, w):
    b = 3
    x = synth.normal((b, c0 + c1, h, w), 800 + c0).requires_grad_(True)
    wt = synth.normal((n, c0 + c1, k, k), 801, (1.0 / ((c0 + c1) * k * k)) ** 0.5).requires_grad_(True)
    bias = synth.normal((n,), 802, 0.1).requires_grad_(True)
    y = F.conv2d(x, wt, bias, stride=stride, padding=pad)
    dy = synth.normal(tuple(y.shape), 803)
    y.backward(dy)
    xs = nhwc(x.detach())
    x0 = xs[..., :c0].contiguous().cuda()
    x1 = xs[..., c0:].contiguous().cuda() if c1 else None
    dyd = nhwc(dy).cuda()
    if n % 4:                                   # pixel stride must be a multiple of 4 floats: pad like the model
        dyd = F.pad(dyd, (0, 4 - n % 4)).contiguous()
    close(bw.conv_wgrad(x0, dyd, n, k, k, stride, pad, x1), wt.grad, 2e-4, "wgrad")
    close(bw.bias_grad(dyd)[:n], bias.grad, 2e-4, "bias grad")
    dyd = dyd[..., :n].contiguous() if n % 4 == 0 else None
    if dyd is None:
        return                                  # dgrad of the padded case is the plain 1x1 path tested above
    wd = wt.detach().cuda()
    if k == 1:
        dx = bw.conv1x1_dgrad(dyd, wd)
    elif k == 3:
        dx = bw.conv3x3_dgrad(dyd, wd)
    else:
        dx = bw.conv2x2s2_dgrad(dyd, wd)
    close(dx.permute(0, 3, 1, 2), x.grad, 2e-4, "dgrad")


@pytest.mark.parametrize("cin,cout,h", [(48, 16, 9), (648, 320, 3), (168, 40, 5)])
def test_deconv_backward_vs_autograd(bw, cin, cout, h):
    b = 2
    x = synth.normal((b, cin, h, h), 810 + cin).requires_grad_(True)
    wt = synth.normal((cin, cout, 2, 2), 811, (1.0 / cin) ** 0.5).requires_grad_(True)
    y = F.conv_transpose2d(x, wt, None, stride=2)
    dy = synth.normal(tuple(y.shape), 812)
    y.backward(dy)
    dyd = nhwc(dy).cuda()
    close(bw.deconv_wgrad(nhwc(x.detach()).cuda(), dyd), wt.grad, 2e-4, "deconv wgrad")
    close(bw.deconv_dgrad(dyd, wt.detach().cuda()).permute(0, 3, 1, 2), x.grad, 2e-4, "deconv dgrad")
