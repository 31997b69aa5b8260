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


def test_bf16_layouts_in_the_source_are_conflict_free():
    """Round 4: the GEMM kernels (both storage types) stage unpadded, XOR-swizzled panels (conv_common.h PanelLayout / panel_swz), the 3x3
    halo is 64-byte pixel rows in 24 slots per halo row with the same swizzle on the halo column, the pointwise kernel has
    8 pieces per row swizzled by (row >> 1) & 7.  Restated here from the source and checked against the lane-group model."""
    common = open(os.path.join(ROOT, "ccvpe_amd", "csrc", "conv_common.h")).read()
    assert "__device__ __forceinline__ int panel_swz(int row) { return ((row >> 2) & 1) << 1; }" in common
    assert "static constexpr int LD = SWZ ? 16 : LDS_LD;" in common
    c3 = open(os.path.join(ROOT, "ccvpe_amd", "csrc", "conv3x3_impl.h")).read()
    assert "static constexpr bool SWZ = true;" in common
    assert "static constexpr bool HSW = !(NW == 8 && sizeof(T) == 4 && !DMA);" in c3 and "static constexpr int HCP = HSW ? 24 : 18;" in c3 and "(((hx >> 2) & 1) << 1)" in c3
    pw = open(os.path.join(ROOT, "ccvpe_amd", "csrc", "conv_pw_impl.h")).read()
    assert "static constexpr bool SWZ = true;" in pw and "static constexpr int LDF = SWZ ? 16 * KP : 16 * KP + 4;" in pw and "((frow >> 1) & 7)" in pw
    swz = lambda row: ((row >> 2) & 1) << 1
    for base in range(0, 256, 16):                      # row-indexed panels: fragment rows 16 j + (lane & 15)
        assert L.cycles("read_b128", L.fragment_b128(16, base=base, swizzle=swz)) == (4, 4)
    # 3x3 halo: pixel (hy, hx) at slot hy * 24 + hx, fragment lanes at hx = (lane & 15) + kx, kx = 0..2, any halo row
    for hy in range(6):
        for kx in range(3):
            def addr(lane, hy=hy, kx=kx):
                hx = (lane % 16) + kx
                return (hy * 24 + hx) * 16 + 4 * ((lane // 16) ^ swz(hx))
            assert L.cycles("read_b128", addr) == (4, 4)
    # pointwise kernel: rows of 32 floats (two 64-byte K pieces), piece index 4 kp + (lane >> 4) XOR (row >> 1) & 7
    for kp in range(2):
        for base in (0, 16, 48):
            def addr(lane, kp=kp, base=base):
                row = base + lane % 16
                return row * 32 + 4 * ((4 * kp + lane // 16) ^ ((row >> 1) & 7))
            assert L.cycles("read_b128", addr) == (4, 4)
    # ... and what they replace: pitch 36 (pointwise) and pitch 20 are 2-way conflicts on every fragment read
    assert L.cycles("read_b128", L.fragment_b128(36)) == (8, 4)
    # staging writes of the swizzled panels: one row = 4 (8) contiguous pieces per lane group
    assert L.cycles("write_b128", L.staging_b128(16, swizzle=swz)) == (8, 8)


def test_paired_level6_halo_is_conflict_free_with_a_gap_of_eight_slots():
    """upconv_dma_kernel PAIR form: lanes 0-7 of a fragment read image A's halo (slots c .. c + 7), lanes 8-15 image B's (slots
    16 + c ..): with the swizzle keyed on the slot, a gap of 8 keeps the read conflict-free, gaps of 2 / 4 / 6 do not."""
    up = open(os.path.join(ROOT, "ccvpe_amd", "csrc", "upconv_impl.h")).read()
    assert "const int hx = frow + c + (PAIR ? 8 * (frow >> 3) : 0);" in up and "HCP = PAIR ? 32 : 24" in up
    swz = lambda row: ((row >> 2) & 1) << 1
    def cyc(gap, c):
        def addr(lane):
            frow = lane % 16
            hx = frow + c + gap * (frow >> 3)
            return hx * 16 + 4 * ((lane // 16) ^ swz(hx))
        return L.cycles("read_b128", addr)
    for c in range(3):
        assert cyc(8, c) == (4, 4)
        assert cyc(2, c) == (8, 4)


def test_round5_layouts_in_the_source_are_conflict_free():
    """Round 5: the ring GEMM's 64-byte panel rows (csrc/conv_pw2_impl.h: piece ^ (-(row >> 2) & 3), the LDS side written lane-linearly by
    the DMA) and the fused stem's tile (csrc/stem_dw.hip: 36 floats per pixel, 40 pixels per row; lane = channel group + 8 x row)."""
    src = open(os.path.join(ROOT, "ccvpe_amd", "csrc", "conv_pw2_impl.h")).read()
    assert "const int fcol = ((lane >> 4) ^ ((0 - (frow >> 2)) & 3)) * 4;" in src, "the ring GEMM's fragment swizzle changed: update this test"
    swz = lambda row: (0 - (row >> 2)) & 3
    for base in range(0, 256, 16):                                  # fragment rows (wm MT + i) 16 + lane % 16: tile bases are multiples of 16
        assert L.cycles("read_b128", L.fragment_b128(16, base=base, swizzle=swz)) == (4, 4)
    assert L.cycles("read_b128", L.fragment_b128(16)) == (8, 4)     # (the same rows without the swizzle: 2-way conflicts)
    src = open(os.path.join(ROOT, "ccvpe_amd", "csrc", "stem_dw.hip")).read()
    assert "constexpr int SD_PP = 36;" in src and "constexpr int SD_RP = 40;" in src, "the fused stem's tile pitches changed: update this test"
    PP, RP = 36, 40
    for col in range(0, 40, 3):                                     # depthwise window reads: lane = cg + 8 orow, any window column
        rd = lambda lane, col=col: ((lane >> 3) * RP + col) * PP + (lane & 7) * 4
        assert L.cycles("read_b128", rd) == (4, 4)
    for first in (0, 5, 16, 34):                                    # stem epilogue: 16 consecutive halo pixels x 4 channel quads per store
        wr = lambda lane, first=first: (first + (lane & 15)) * PP + 4 * (lane >> 4)
        assert L.cycles("write_b128", wr) == (8, 8)
    assert L.cycles("write_b128", lambda lane: (lane & 15) * 32 + 4 * (lane >> 4)) == (64, 8)   # (pitch 32: every group on one bank set)


def test_tail512_wide_slot_geometry_in_the_source_is_conflict_free():
    """csrc/tail512.hip stage 1 (round 6): a tile's 16 lanes are 16 CONSECUTIVE positions of a (TY+1) x (TX+1) grid, i.e. they
    straddle two halo rows, and read their 2 x 2 taps at per-lane addresses.  With the 80-byte slot in rows of TX + 2 that is a 2-way
    conflict on every read (SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS = 1.5-2.1 measured, profiles/r06/probes_second_half.txt); the
    geometry the source takes for the bf16-path tiles — read from the source here — has none, for every tile and tap."""
    src = open(os.path.join(ROOT, "ccvpe_amd", "csrc", "tail512.hip")).read()
    m = re.search(r"static constexpr int LD = WIDE_SLOT \? (\d+) : (\d+);", src)
    n = re.search(r"static constexpr int HCP = WIDE_SLOT \? (\d+) : HC;", src)
    assert m and n, "TailGeom's slot geometry moved: update this test"
    ld_wide, ld_plain, hcp_wide = int(m.group(1)), int(m.group(2)), int(n.group(1))
    ty, tx = 8, 16
    pw, hc, npos = tx + 1, tx + 2, (ty + 1) * (tx + 1)

    def frag(ld, hcp, tile, tap):
        def addr(lane):
            pc = min(tile * 16 + lane % 16, npos - 1)
            iy, ix = pc // pw, pc % pw
            return ((iy + (tap >> 1)) * hcp + ix + (tap & 1)) * ld + 4 * (lane // 16)
        return addr

    nfull = npos // 16                                     # the last tile of a parity is ragged (clamped lanes): not asserted
    wide = [L.cycles("read_b128", frag(ld_wide, hcp_wide, t, tap))[0] for t in range(nfull) for tap in range(4)]
    plain = [L.cycles("read_b128", frag(ld_plain, hc, t, tap))[0] for t in range(nfull) for tap in range(4)]
    assert max(wide) == 4, wide                            # conflict-free: 4 LDS cycles per ds_read_b128
    assert sum(plain) / len(plain) > 7.0                   # the 80-byte slot: ~2-way everywhere (what the counters saw)
    assert (ty + 2) * hcp_wide * ld_wide * 4 * 2 <= 48 * 1024      # two chunk buffers of the 8 x 16 tile: three workgroups per CU
