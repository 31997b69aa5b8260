// Narrow decoder levels, bf16 storage (narrow_impl.h): instantiations + dispatch.
#include "narrow_impl.h"

namespace ccvpe {

int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else n = 256;
  }
  return n;
}

// Which (CPT, NT) instantiation serves a 3x3 layer, or 0: 40 -> <= 48 (level 2 of the localisation decoder), 32 -> 32 (level 2 of
// the orientation decoder, KITTI's level 2), 64 -> 64 (level 3 of the orientation decoder), 80 -> 80 (level 3 of the localisation
// decoder: 276 weight registers per wave with the output channels split over two groups of waves).
int c3n_supported(const IgemmParams& p, int batch) {
  const int ctot = p.c0 + p.c1;
  if (p.c1 != 0 || p.residual || p.scale || p.gate) return 0;
  if (p.act != CCVPE_ACT_NONE && p.act != CCVPE_ACT_RELU) return 0;
  if (p.W % 16 || p.W < 16 || p.ld0 % 8 || p.N % 8 || p.ldd % (p.out_f32 ? 4 : 8)) return 0;      // (16-byte stores of 8 bf16 / 4 fp32 channels)
  int id = 0, th = 16;
  if (ctot == 40 && p.Npad == 48) id = 1;
  else if (ctot == 32 && p.Npad == 32) id = 2;
  else if (ctot == 64 && p.Npad == 64) id = 3;
  else if (ctot == 80 && p.Npad == 80) { id = 4; th = 8; }    // the channels split over two groups of waves: 8 x 16 pixel tiles
  if (!id || p.N <= p.Npad - 16) return 0;
  if (p.H % th) return 0;
  const int nch = (9 * (ctot / 8) + 3) / 4;
  if (p.Kpad < 32 * nch) return 0;
  const long tiles = (long)batch * (p.H / th) * (p.W / 16);
  if (tiles < 2L * num_cus()) return 0;                        // persistent workgroups: at least two tiles each, or the tiled kernel
  // (no size limit on the tensor: the DMA offsets are relative to the tile's first pixel, always < 4 GB)
  return id;
}

static int c3n_launch_id(int id, const NarrowParams& q, int batch, bool f32out, bool match, hipStream_t stream) {
  if (id == 1) return launch_c3n<5, 3, 4>(q, batch, f32out, match, stream);
  if (id == 2) return launch_c3n<4, 2, 4>(q, batch, f32out, match, stream);
  if (id == 3) return launch_c3n<8, 4, 4, false, false>(q, batch, f32out, match, stream);
  return launch_c3n<10, 3, 4, false, false, 2>(q, batch, f32out, match, stream);
}

int c3n_dispatch(const IgemmParams& p, int batch, hipStream_t stream) {
  const int id = c3n_supported(p, batch);
  if (!id) return fail(CCVPE_EINVAL, "c3n: unsupported layer");
  NarrowParams q{};
  q.src = p.src0; q.w = p.w; q.shift = p.shift; q.dst = p.dst;
  q.H = p.H; q.W = p.W; q.ld = p.ld0; q.N = p.N; q.Npad = p.Npad; q.Kpad = p.Kpad; q.ldd = p.ldd;
  q.act_floor = p.act == CCVPE_ACT_RELU ? 0.f : -__builtin_huge_valf();
  return c3n_launch_id(id, q, batch, p.out_f32 != 0, false, stream);
}

// 3x3 conv + the next level's one-hypothesis matching in its epilogue (c3n_kernel<MATCH>).  `p` describes the conv with dst = the
// decoder input rows (pitch p.ldd >= N + 1, a multiple of 8) — c3n_supported() is asked about the conv alone (ldd = N).
int c3n_match_supported(const IgemmParams& p, int batch, int L) {
  IgemmParams c = p;
  c.ldd = p.out_f32 ? (p.N + 3) / 4 * 4 : (p.N + 7) / 8 * 8;
  const int id = c3n_supported(c, batch);
  if (!id || id >= 3) return 0;                               // (64 / 80 channels: no registers left for the tables, or channels split over waves)
  const int nt = id == 1 ? 3 : 2;
  if (p.N % 4 || p.ldd < p.N + 1 || p.ldd > 16 * nt || p.ldd % (p.out_f32 ? 4 : 8) || L < 1 || L > p.N) return 0;
  if (p.act != CCVPE_ACT_NONE) return 0;
  const int lds = (id == 1 ? C3nGeom<5, 3, 4>::LDS_BYTES : C3nGeom<4, 2, 4>::LDS_BYTES) + batch * (2 * 16 * nt + 1) * 4;
  if (lds > 160 * 1024) return 0;
  return id;
}

int c3n_match_dispatch(const IgemmParams& p, int batch, const float* g, int ldg, int L, int off, float* scores, hipStream_t stream) {
  const int id = c3n_match_supported(p, batch, L);
  if (!id) return fail(CCVPE_EINVAL, "conv3x3_match1: unsupported layer / shape");
  NarrowParams q{};
  q.src = p.src0; q.w = p.w; q.shift = p.shift; q.dst = p.dst;
  q.H = p.H; q.W = p.W; q.ld = p.ld0; q.N = p.N; q.Npad = p.Npad; q.Kpad = p.Kpad; q.ldd = p.ldd;
  q.act_floor = -__builtin_huge_valf();
  q.g = g; q.ldg = ldg; q.L = L; q.off = off; q.scores = scores;
  return c3n_launch_id(id, q, batch, p.out_f32 != 0, true, stream);
}

// Folded deconv + 3x3 on the narrow levels (up2_kernel): level 2 of the localisation decoder (81 -> 40 | 16 skip channels),
// level 2 of the orientation decoder (64 -> 32 | 16), KITTI's localisation level 2 (129 -> 32 | 16).
int up2_supported(int c0, int c1, int n, int kpad, int h1, int w1, int batch) {
  const int npad = (n + 15) / 16 * 16;
  int id = 0;
  if (c0 == 88 && c1 == 16 && npad == 48) id = 1;
  else if (c0 == 64 && c1 == 16 && npad == 32) id = 2;
  else if (c0 == 136 && c1 == 16 && npad == 32) id = 3;
  if (!id) return 0;
  constexpr int MT = 8;
  if (w1 % 16 || w1 < 16 || h1 % MT) return 0;
  const int nch = c0 / 8 + (9 * (c1 / 8) + 3) / 4;
  if (kpad < 32 * nch) return 0;
  const long tiles = (long)batch * (h1 / MT) * (w1 / 16);
  if (tiles < 2L * num_cus()) return 0;
  return id;
}

int up2_dispatch(const void* src0, const void* src1, const void* w, const float* shift9, void* dst, int c0, int ld0, int c1, int ld1,
                 int h1, int w1, int n, int kpad, int ldd, int act, int batch, hipStream_t stream) {
  const int id = up2_supported(c0, c1, n, kpad, h1, w1, batch);
  if (!id) return fail(CCVPE_EINVAL, "up2: unsupported layer");
  if (act != CCVPE_ACT_NONE && act != CCVPE_ACT_RELU) return fail(CCVPE_EINVAL, "up2: activation");
  Up2Params q{};
  q.src0 = src0; q.src1 = src1; q.w = w; q.shift9 = shift9; q.dst = dst;
  if (n % 8 || ldd % 8) return fail(CCVPE_EINVAL, "up2: n and ldd must be multiples of 8");
  q.H1 = h1; q.W1 = w1; q.ld0 = ld0; q.ld1 = ld1; q.N = n; q.Npad = (n + 15) / 16 * 16; q.Kpad = kpad; q.ldd = ldd;
  q.act_floor = act == CCVPE_ACT_RELU ? 0.f : -__builtin_huge_valf();
  if (id == 1) return launch_up2<11, 2, 3, 8>(q, batch, stream);
  if (id == 2) return launch_up2<8, 2, 2, 8>(q, batch, stream);
  return launch_up2<17, 2, 2, 8>(q, batch, stream);
}

}  // namespace ccvpe
