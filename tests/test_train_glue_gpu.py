"""Training-step glue on the device: ground-truth construction and the one-launch Adam."""
import numpy as np
import pytest
import torch

from ccvpe_amd import synth
from oracle import ccvpe_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_bins,angles", [(20, [0.0, 7.5, 18.0, 200.25, 359.9]), (16, [0.0, 22.5, 100.0, 337.6, 359.0])])
def test_train_targets_vs_dataset_restatement(n_bins, angles):
    from ccvpe_amd import targets
    centers = [[0.0, 0.0], [37.0, -120.0], [-200.0, 55.0], [255.0, 255.0], [-3.0, 1.0]]
    want = O.train_targets(centers, angles, n_bins)
    got = targets.train_targets(torch.tensor(centers).cuda(), torch.tensor(angles).cuda(), n_bins)
    names = ("gt", "gt_flattened", "gt_orientation")
    for nm, g, w in zip(names, got[:3], want[:3]):
        g = g.cpu()
        assert g.shape == w.shape, nm
        assert (g - w).abs().max().item() <= 2e-6 * w.abs().max().item() + 1e-12, nm
    for l, (g, w) in enumerate(zip(got[3], want[3])):
        g = g.cpu()
        assert g.shape == w.shape
        assert (g - w).abs().max().item() <= 1e-5, "pyramid level %d" % (l + 1)
        assert torch.equal(g.amax(dim=(2, 3)) > 0, w.amax(dim=(2, 3)) > 0)             # same orientation bins are populated
    # the device pyramid is the max-pool of the device gt (separable cell maximum): check the dominant bin of level 6
    for b in range(len(angles)):
        pooled = torch.nn.functional.max_pool2d(got[0], 2, 2)[b, 0]
        lab = got[3][5][b]
        top = int(lab.amax(dim=(1, 2)).argmax())
        w = lab[top].max() / pooled.max()
        assert torch.allclose(lab[top], pooled * w, rtol=2e-6, atol=1e-30)


def test_adam_matches_torch_adam():
    from ccvpe_amd import optim
    shapes = [(160, 336, 3, 3), (17,), (1, 10, 1, 1), (4099,), (96, 1, 5, 5)]      # 118 chunks of 4 096 elements in the first one
    ref_p = [torch.nn.Parameter(synth.normal(s, 2000 + i, 0.1)) for i, s in enumerate(shapes)]
    our_p = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref_p]
    ref = torch.optim.Adam(ref_p, lr=1e-4, betas=(0.9, 0.999))
    ours = optim.Adam(our_p, lr=1e-4, betas=(0.9, 0.999))
    for step in range(5):
        for i, (a, b) in enumerate(zip(ref_p, our_p)):
            if i == 1:                      # a parameter that never receives a gradient is left untouched
                continue
            g = synth.normal(tuple(a.shape), 3000 + 10 * step + i, 1e-3 * (1 + i))
            a.grad = g.clone()
            b.grad = g.permute(*reversed(range(g.dim()))).contiguous().permute(*reversed(range(g.dim()))).cuda() if i == 0 else g.cuda()
        ref.step()
        ours.step()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(ref_p, our_p)):
        d = (a.detach() - b.detach().cpu()).abs().max().item()
        moved = (a.detach() - synth.normal(shapes[i], 2000 + i, 0.1)).abs().max().item()
        assert d <= 2e-3 * moved + 1e-9, (i, d, moved)
    assert torch.equal(our_p[1].detach().cpu(), synth.normal(shapes[1], 2001, 0.1))
    sd = ours.state_dict()
    assert set(sd["state"].keys()) == {0, 2, 3, 4} and float(sd["state"][0]["step"]) == 5.0
    rs = ref.state_dict()["state"]
    assert torch.allclose(sd["state"][3]["exp_avg_sq"].cpu(), rs[3]["exp_avg_sq"], rtol=1e-5, atol=1e-12)
    twin = optim.Adam(our_p, lr=1e-4)
    twin.load_state_dict(sd)
    assert float(twin.state[our_p[0]]["step"]) == 5.0 and torch.equal(twin.state[our_p[0]]["exp_avg"], ours.state[our_p[0]]["exp_avg"])
    with pytest.raises(ValueError):                      # torch's own validation: wrong number of parameters
        optim.Adam(our_p[:2], lr=1e-4).load_state_dict(sd)


def test_adam_late_parameter_scheduler_and_grad_scale():
    """torch.optim.Adam semantics the first version lacked: a parameter that starts receiving gradients later
    keeps its own step count / bias correction, torch LR schedulers drive the optimizer, per-group hyper-parameters;
    plus grad_scale (SUM all-reduce -> mean folded into the update)."""
    from ccvpe_amd import optim
    shapes = [(33, 7), (4096,), (5, 3, 3, 3)]
    ref_p = [torch.nn.Parameter(synth.normal(s, 2100 + i, 0.1)) for i, s in enumerate(shapes)]
    our_p = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref_p]
    ref = torch.optim.Adam([dict(params=ref_p[:2]), dict(params=ref_p[2:], lr=3e-3, betas=(0.8, 0.99))], lr=1e-3)
    ours = optim.Adam([dict(params=our_p[:2]), dict(params=our_p[2:], lr=3e-3, betas=(0.8, 0.99))], lr=1e-3)
    rs = torch.optim.lr_scheduler.StepLR(ref, step_size=2, gamma=0.5)
    os_ = torch.optim.lr_scheduler.StepLR(ours, step_size=2, gamma=0.5)          # TypeError before: not an Optimizer
    ours.grad_scale = 0.25
    for step in range(6):
        for i, (a, b) in enumerate(zip(ref_p, our_p)):
            if i == 1 and step < 3:            # starts receiving gradients at step 3
                a.grad, b.grad = None, None
                continue
            g = synth.normal(tuple(a.shape), 3100 + 10 * step + i, 1e-2)
            a.grad = g.clone() * 0.25
            b.grad = g.cuda()
        ref.step(); ours.step()
        rs.step(); os_.step()
    torch.cuda.synchronize()
    assert float(ours.state[our_p[1]]["step"]) == 3.0 and float(ours.state[our_p[0]]["step"]) == 6.0
    for i, (a, b) in enumerate(zip(ref_p, our_p)):
        d = (a.detach() - b.detach().cpu()).abs().max().item()
        moved = (a.detach() - synth.normal(shapes[i], 2100 + i, 0.1)).abs().max().item()
        assert d <= 2e-3 * moved + 1e-9, (i, d, moved)


def test_targets_angle_wrap():
    """a negative / >= 360 angle must land in the same bins as its wrapped value (datasets.py:483-487 wraps
    before binning), never produce all-zero labels."""
    from ccvpe_amd import targets
    center = torch.tensor([[3.0, -7.0]] * 4, device="cuda")
    ang = torch.tensor([-30.0, 330.0, 365.5, 5.5], device="cuda")
    labs = targets.train_targets(center, ang, 16)[3]
    for l in labs:
        assert torch.equal(l[0], l[1]) and torch.equal(l[2], l[3])
        assert float(l[0].sum()) > 0


def test_multi_copy_many_small_vectors():
    """ccvpe_multi_copy_f32 (gradients -> all-reduce arena slots): 230 vectors of ragged sizes, more than one launch batch."""
    from ccvpe_amd import harness
    g = torch.Generator().manual_seed(5)
    sizes = [int(torch.randint(1, 3000, (1,), generator=g)) for _ in range(229)] + [65536]
    srcs = [torch.randn((n,), generator=g).cuda() for n in sizes]
    flat = torch.full((sum(sizes) + 7,), -1.0, device="cuda")
    views, off = [], 3
    for n in sizes:
        views.append(flat[off:off + n])
        off += n
    harness._multi_copy([(s_.data_ptr(), v.data_ptr(), n) for s_, v, n in zip(srcs, views, sizes)])
    torch.cuda.synchronize()
    for s_, v in zip(srcs, views):
        assert torch.equal(s_, v)
    assert bool((flat[:3] == -1).all()) and bool((flat[off:] == -1).all())      # nothing written outside the slots
