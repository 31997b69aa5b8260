"""ccvpe_ctx / ccvpe_forward (csrc/plan.hip, ccvpe_amd/plan.py): the whole eval forward as one C call.
  * PlannedForward == the eager Python forward, bit for bit, on the recorded pair AND on new inputs through the same ctx
    (fp32 ori_prior, bf16 CVM_VIGOR with its split-bf16 fp32 tail, CVM_KITTI);
  * the workspace is laid out by lifetime (far smaller than the sum of all intermediates);
  * a caller WITHOUT Python — tools/plan_run.cpp, built here with hipcc against include/ccvpe_hip.h — loads the serialised plan
    from a file and reproduces the nine outputs bit for bit."""
import json
import os
import subprocess

import numpy as np
import pytest
import torch

from ccvpe_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _net(synth_sd, which, precision):
    from ccvpe_amd import models
    if which == "kitti":
        net, kind, gshape = models.CVM_KITTI("cuda"), "kitti", "kitti"
    elif which == "prior0":
        net, kind, gshape = models.CVM_VIGOR_ori_prior("cuda", 0, True), "vigor", "vigor"
    else:
        net, kind, gshape = models.CVM_VIGOR("cuda", True), "vigor", "vigor"
    net.load_state_dict(synth_sd(kind, 0), strict=True)
    return net.to("cuda:0").eval().set_precision(precision), gshape


@pytest.mark.parametrize("which,precision,batch", [("prior0", "fp32", 2), ("vigor20", "bf16", 3), ("kitti", "fp32", 1)])
def test_planned_forward_is_the_eager_forward(synth_sd, which, precision, batch):
    from ccvpe_amd import plan
    net, gshape = _net(synth_sd, which, precision)
    grd, sat = synth.synthetic_pair(batch, gshape, 31)
    grd, sat = grd.cuda(), sat.cuda()
    eager = [t.clone() for t in net(grd, sat)]
    pf = plan.PlannedForward(net, grd, sat)
    got = pf(grd, sat)
    torch.cuda.synchronize()
    assert len(got) == 9
    for k, (a, b) in enumerate(zip(got, eager)):
        assert a.shape == b.shape and torch.equal(a, b), "output %d differs from the eager forward" % k
    # new inputs through the same ctx; the eager forward in between must not disturb the ctx's workspace
    grd2, sat2 = synth.synthetic_pair(batch, gshape, 32)
    grd2, sat2 = grd2.cuda(), sat2.cuda()
    eager2 = [t.clone() for t in net(grd2, sat2)]
    got2 = pf(grd2, sat2)
    torch.cuda.synchronize()
    for k, (a, b) in enumerate(zip(got2, eager2)):
        assert torch.equal(a, b), "output %d differs on new inputs" % k
    # lifetime layout: every intermediate + the outputs in much less than their sum
    assert pf.plan.workspace_bytes < 0.6 * _sum_of_allocations(net, grd, sat)
    assert len(pf.plan.calls) > 100


def _sum_of_allocations(net, grd, sat):
    from ccvpe_amd import ops
    seen = []
    real = ops._empty

    def counting(shape, device=None, dtype=torch.float32):
        t = real(shape, device=device, dtype=dtype)
        seen.append(t.numel() * t.element_size())
        return t
    ops._empty = counting
    try:
        net(grd, sat)
    finally:
        ops._empty = real
    return sum(seen)


def test_a_caller_without_python_reproduces_the_forward(synth_sd, tmp_path):
    from ccvpe_amd import plan
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "plan_run")
    libdir = os.path.join(ROOT, "ccvpe_amd")
    subprocess.run([hipcc, "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "plan_run.cpp"), "-L", libdir,
                    "-lccvpe_hip", "-Wl,-rpath," + libdir, "-o", exe], check=True, capture_output=True, timeout=600)
    net, gshape = _net(synth_sd, "prior0", "fp32")
    grd, sat = synth.synthetic_pair(2, gshape, 77)
    grd, sat = grd.cuda(), sat.cuda()
    pl, outs = plan.record(net, grd, sat)
    want = np.concatenate([t.detach().cpu().numpy().reshape(-1) for t in outs])
    pl.save(str(tmp_path / "m.plan"))
    grd.cpu().numpy().tofile(str(tmp_path / "grd.f32"))
    sat.cpu().numpy().tofile(str(tmp_path / "sat.f32"))
    res = subprocess.run([exe, str(tmp_path / "m.plan"), str(tmp_path / "grd.f32"), str(tmp_path / "sat.f32"),
                          str(tmp_path / "out.f32"), "3"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    info = json.loads(res.stdout.strip().splitlines()[-1])
    assert info["outputs"] == 9 and info["calls"] == len(pl.calls)
    got = np.fromfile(str(tmp_path / "out.f32"), dtype=np.float32)
    assert got.shape == want.shape and np.array_equal(got, want), "the C++ caller's outputs differ from the Python forward"


def test_ccvpe_forward_is_capturable_in_a_hipgraph(synth_sd):
    """ccvpe_forward enqueues on the caller's stream and on the ctx's own side stream, joined through events: a capture on the
    caller's stream takes the whole forward (both streams) into the graph; the replay reproduces the eager outputs."""
    from ccvpe_amd import plan
    net, gshape = _net(synth_sd, "vigor20", "bf16")
    grd, sat = synth.synthetic_pair(2, gshape, 41)
    grd, sat = grd.cuda(), sat.cuda()
    eager = [t.clone() for t in net(grd, sat)]
    pf = plan.PlannedForward(net, grd, sat)
    assert pf.plan.n_waits >= 2                            # the recorded fork / join of the two streams
    pf(grd, sat)
    torch.cuda.synchronize()
    sg, ss = grd.clone(), sat.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        pf(sg, ss)
    grd2, sat2 = synth.synthetic_pair(2, gshape, 42)
    sg.copy_(grd2.cuda())
    ss.copy_(sat2.cuda())
    g.replay()
    torch.cuda.synchronize()
    want = net(grd2.cuda(), sat2.cuda())
    for k, (a, b) in enumerate(zip(pf.outputs, want)):
        assert torch.equal(a, b), "graph replay of ccvpe_forward differs in output %d" % k
    del eager


def test_ctx_rejects_what_it_cannot_replay(synth_sd):
    import ctypes
    from ccvpe_amd import _lib, plan
    lib = _lib.load()
    net, gshape = _net(synth_sd, "prior0", "fp32")
    grd, sat = synth.synthetic_pair(1, gshape, 5)
    pl, _ = plan.record(net, grd.cuda(), sat.cuda())
    blob = bytearray(pl.to_bytes())
    ctx = ctypes.c_void_p()

    def create(b):
        host = ctypes.create_string_buffer(bytes(b), len(b))
        return lib.ccvpe_ctx_create(ctypes.cast(host, ctypes.c_void_p), len(b), None, None, ctypes.byref(ctx))
    assert create(blob[:4096]) != 0 and b"truncated" in lib.ccvpe_last_error()
    bad = bytearray(blob)
    bad[8] = 99                                            # ABI stamp
    assert create(bad) != 0 and b"ABI" in lib.ccvpe_last_error()
    pf = plan.PlannedForward(net, grd.cuda(), sat.cuda())
    with pytest.raises(ValueError, match="was built for"):
        pf(torch.zeros((2,) + tuple(grd.shape[1:]), device="cuda"), sat.cuda().repeat(2, 1, 1, 1))
