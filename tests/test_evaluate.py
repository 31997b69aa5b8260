"""The sharded evaluation loop (ccvpe_amd/evaluate.py): metric formulas against a restatement of the
reference's host loop (train_VIGOR.py:294-336), 2-rank gloo run == single-process run (CPU, stub
forward), and on the GPU the real model + device post-processing."""
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_samples(n, hw=32):
    g = torch.Generator().manual_seed(7)
    out = []
    for i in range(n):
        ang = float(torch.rand((), generator=g)) * 2 * math.pi
        out.append(dict(grd=torch.randn(3, 4, 8, generator=g), sat=torch.randn(3, hw, hw, generator=g),
                        gt_yx=(int(torch.randint(0, hw, (), generator=g)), int(torch.randint(0, hw, (), generator=g))),
                        gt_cos_sin=(math.cos(ang), math.sin(ang)), metres_per_pixel=0.11 + 0.01 * (i % 3)))
    return out


def stub_forward(grd, sat):
    """deterministic fake 9-tuple from the inputs (CPU): heat-map = softmax of a channel mix"""
    b, _, h, w = sat.shape
    logits = (sat[:, 0] * 1.7 - sat[:, 1] + 0.3 * sat[:, 2]).reshape(b, -1)
    heat = torch.softmax(logits, 1).reshape(b, 1, h, w)
    ori = torch.nn.functional.normalize(sat[:, :2] + 0.1, dim=1)
    return (logits, heat, ori) + (None,) * 6


def cpu_post(heat, ori):
    from oracle import ccvpe_oracle as O
    return O.eval_postprocess(heat, ori)


def reference_style_metrics(samples):
    """train_VIGOR.py:294-336 restated with numpy, one sample at a time."""
    pd, md, oe = [], [], []
    for s in samples:
        _, heat, ori = stub_forward(s["grd"][None], s["sat"][None])[:3]
        heat, ori = heat.numpy(), ori.numpy()
        loc = np.unravel_index(heat[0].argmax(), heat[0].shape)
        d = np.sqrt((s["gt_yx"][0] - loc[1]) ** 2 + (s["gt_yx"][1] - loc[2]) ** 2)
        pd.append(d)
        md.append(d * s["metres_per_pixel"])
        c, sn = ori[0, :, loc[1], loc[2]]
        if abs(c) <= 1 and abs(sn) <= 1:
            a = math.acos(c)
            ap = math.degrees(-a) % 360 if sn < 0 else math.degrees(a)
            ag = math.acos(max(-1, min(1, s["gt_cos_sin"][0])))
            ag = math.degrees(-ag) % 360 if s["gt_cos_sin"][1] < 0 else math.degrees(ag)
            oe.append(min(abs(ag - ap), 360 - abs(ag - ap)))
    return np.mean(pd), np.median(pd), np.mean(md), np.median(md), np.mean(oe), np.median(oe)


def test_metrics_match_reference_style_loop():
    sys.path.insert(0, ROOT)
    from ccvpe_amd import evaluate as E
    samples = make_samples(11)
    res = E.evaluate(stub_forward, cpu_post, samples, batch_size=4)
    want = reference_style_metrics(samples)
    got = (res["mean_pixel_error"], res["median_pixel_error"], res["mean_metre_error"], res["median_metre_error"],
           res["mean_orientation_error"], res["median_orientation_error"])
    assert res["n"] == 11
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-4)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ccvpe_amd import evaluate as E
    res = E.evaluate(stub_forward, cpu_post, make_samples(11), batch_size=3)
    if rank == 1:                       # every rank holds the global result; check the non-zero one
        torch.save({k: v for k, v in res.items()}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_evaluation_equals_single_process(tmp_path):
    sys.path.insert(0, ROOT)
    from ccvpe_amd import evaluate as E
    single = E.evaluate(stub_forward, cpu_post, make_samples(11), batch_size=3)
    out = str(tmp_path / "r.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    multi = torch.load(out, weights_only=False)
    assert multi["n"] == single["n"] == 11
    np.testing.assert_array_equal(multi["rows"], single["rows"])
    assert multi["median_metre_error"] == single["median_metre_error"]


@pytest.mark.gpu
def test_evaluate_on_gpu_with_device_postprocess(synth_sd):
    from ccvpe_amd import evaluate as E, models, ops, synth
    from oracle import ccvpe_oracle as O
    net = models.CVM_VIGOR_ori_prior("cuda", 0, True)
    net.load_state_dict(synth_sd("vigor", 0), strict=True)
    net = net.to("cuda:0").eval()
    grd, sat = synth.synthetic_pair(3, "vigor", 555)
    samples = [dict(grd=grd[i], sat=sat[i], gt_yx=(100 + 50 * i, 300 - 40 * i), gt_cos_sin=(0.6, -0.8),
                    metres_per_pixel=0.113248 / 512 * 640) for i in range(3)]
    res = E.evaluate(net, ops.eval_postprocess, samples, batch_size=2, device="cuda:0")
    with torch.no_grad():
        ref = O.forward(synth_sd("vigor", 0), grd, sat, "vigor", True, ori_noise=0)
    post = O.eval_postprocess(ref[1], ref[2])
    for i in range(3):
        pd, md, oe, pr = E.sample_metrics(post[i], samples[i]["gt_yx"], samples[i]["gt_cos_sin"], samples[i]["metres_per_pixel"])
        assert abs(res["rows"][i, 0] - pd) < 1e-6 and abs(res["rows"][i, 1] - md) < 1e-6
        assert abs(res["rows"][i, 2] - oe) < 0.05
