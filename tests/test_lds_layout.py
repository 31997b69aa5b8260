"""The LDS bank-conflict model of tools/lds_layout.py (lane groups and bank widths of MI355X_MICROARCH.md) applied to the layouts
DESIGN section 4 "LDS bank conflicts" talks about: what the PMC pass measured (2-way conflicts on the pitch-20 fragment reads and
on the old weight-gradient pitch) and what the derived layouts promise (pitch 24 / pitch 16 + XOR swizzle conflict-free for EVERY
window base, the weight-gradient pitch now in the code conflict-free)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import lds_layout as L      # noqa: E402


def test_pitch_20_fragment_reads_are_two_way_conflicts_as_measured():
    c, ideal = L.cycles("read_b128", L.fragment_b128(20))
    assert (c, ideal) == (8, 4)                 # SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50 for igemm_kernel (profiles/r03/lds_conflicts_*.json)
    assert L.cycles("read_b32", L.wgrad_b32(68)) == (4, 2) and L.cycles("read_b32", L.wgrad_b32(84)) == (4, 2)


def test_derived_layouts_are_conflict_free_for_every_window_base():
    for base in range(64):
        assert L.cycles("read_b128", L.fragment_b128(24, base=base)) == (4, 4)
        assert L.cycles("read_b128", L.fragment_b128(16, base=base, swizzle=L.xor_swizzle)) == (4, 4)
    assert L.cycles("write_b128", L.staging_b128(16, swizzle=L.xor_swizzle)) == (8, 8)
    assert L.cycles("write_b128", L.staging_b128(24, swap_rows=True)) == (8, 8)
    assert L.cycles("write_b128", L.staging_b128(24)) == (16, 8)        # (as conflicted as pitch 20: nothing lost by the padding)


def test_weight_gradient_pitch_in_the_source_is_conflict_free():
    src = open(os.path.join(ROOT, "ccvpe_amd", "csrc", "conv_wgrad.hip")).read()
    m = re.search(r"constexpr int wgrad_pitch\(int width, bool square\) \{ return square \? width \+ 4 : \(width \+ 15\) / 32 \* 32 \+ 16; \}", src)
    assert m, "wgrad_pitch changed: update this test's restatement"
    for width in (16, 32, 48, 64, 80):
        pitch = (width + 15) // 32 * 32 + 16
        assert pitch >= width and pitch % 32 == 16
        assert L.cycles("read_b32", L.wgrad_b32(pitch)) == (2, 2)
