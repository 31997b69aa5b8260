"""synth's integer-hash generators give the SAME BITS on the MI355X as on the CPU (ccvpe_amd/synth.py: hash_u32) — the large test
batches and bench.py's inputs are generated on the device (B = 64 pairs take 15 s of host time, milliseconds there)."""
import pytest
import torch

from ccvpe_amd import synth

pytestmark = pytest.mark.gpu


def test_device_generation_is_bit_identical_to_the_cpu():
    for shape, seed in (((3, 5, 7), 11), ((2, 3, 64, 129), 1234), ((1 << 20,), 99)):
        assert torch.equal(synth.normal(shape, seed, device="cuda").cpu(), synth.normal(shape, seed))
        assert torch.equal(synth.normal(shape, seed, 0.3, -1.5, device="cuda").cpu(), synth.normal(shape, seed, 0.3, -1.5))
        assert torch.equal(synth.uniform(shape, seed, -2.0, 5.0, device="cuda").cpu(), synth.uniform(shape, seed, -2.0, 5.0))
    for kind in ("vigor", "vigor_fov180", "kitti", "oxford"):
        g0, s0 = synth.synthetic_pair(2, kind, 4321)
        g1, s1 = synth.synthetic_pair(2, kind, 4321, device="cuda")
        assert torch.equal(g0, g1.cpu()) and torch.equal(s0, s1.cpu())
    # the index space beyond 2^32 elements is not used; a large batch's last sample agrees with the same sample generated alone
    # through its offset in the stream (nothing but the arange differs)
    g, s = synth.synthetic_pair(24, "vigor", 7, device="cuda")
    g_cpu, s_cpu = synth.synthetic_pair(24, "vigor", 7)
    assert torch.equal(g[-1].cpu(), g_cpu[-1]) and torch.equal(s[-1].cpu(), s_cpu[-1])
