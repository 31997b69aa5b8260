// Narrow decoder levels, bf16 storage (narrow_impl.h): instantiations + dispatch.
#include "narrow_impl.h"
#include "tail2_impl.h"

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
// the orientation decoder, KITTI's level 2), 64 -> 64 (level 3 of the orientation decoder).
int c3n_supported(const IgemmParams& p, int batch) {
  const int ctot = p.c0 + p.c1;
  if (p.c1 != 0 || p.residual || p.scale || p.gate) return 0;
  if (p.act != CCVPE_ACT_NONE && p.act != CCVPE_ACT_RELU) return 0;
  if (p.W % 16 || p.W < 16 || p.ld0 % 8 || p.N % 8 || p.ldd % (p.out_f32 ? 4 : 8)) return 0;      // (16-byte stores of 8 bf16 / 4 fp32 channels)
  int id = 0, th = 16;
  if (ctot == 40 && p.Npad == 48) id = 1;
  else if (ctot == 32 && p.Npad == 32) id = 2;
  else if (ctot == 64 && p.Npad == 64) id = 3;
  if (!id || p.N <= p.Npad - 16) return 0;
  if (p.H % th) return 0;
  const int nch = (9 * (ctot / 8) + 3) / 4;
  if (p.Kpad < 32 * nch) return 0;
  const long tiles = (long)batch * (p.H / th) * (p.W / 16);
  if (tiles < 2L * num_cus()) return 0;                        // persistent workgroups: at least two tiles each, or the tiled kernel
  if ((long)p.in_pixels * p.ld0 * 2 >= (1L << 32)) {}          // (DMA offsets are relative to the tile: always < 4 GB)
  return id;
}

static int c3n_launch_id(int id, const NarrowParams& q, int batch, bool f32out, bool match, hipStream_t stream) {
  if (id == 1) return launch_c3n<5, 3, 4>(q, batch, f32out, match, stream);
  if (id == 2) return launch_c3n<4, 2, 4>(q, batch, f32out, match, stream);
  return launch_c3n<8, 4, 4, false, false>(q, batch, f32out, match, stream);
}

int c3n_dispatch(const IgemmParams& p, int batch, hipStream_t stream) {
  const int id = c3n_supported(p, batch);
  if (!id) return fail(CCVPE_EINVAL, "c3n: unsupported layer");
  NarrowParams q{};
  q.src = p.src0; q.w = p.w; q.shift = p.shift; q.dst = p.dst;
  q.H = p.H; q.W = p.W; q.ld = p.ld0; q.N = p.N; q.Kpad = p.Kpad; q.ldd = p.ldd;
  q.act_floor = p.act == CCVPE_ACT_RELU ? 0.f : -__builtin_huge_valf();
  return c3n_launch_id(id, q, batch, p.out_f32 != 0, false, stream);
}

// 3x3 conv + the next level's one-hypothesis matching in its epilogue (c3n_kernel<MATCH>).  `p` describes the conv with dst = the
// decoder input rows (pitch p.ldd >= N + 1, a multiple of 8) — c3n_supported() is asked about the conv alone (ldd = N).
int c3n_match_supported(const IgemmParams& p, int batch, int L) {
  IgemmParams c = p;
  c.ldd = p.out_f32 ? (p.N + 3) / 4 * 4 : (p.N + 7) / 8 * 8;
  const int id = c3n_supported(c, batch);
  if (!id || id == 3) return 0;                               // (64 channels: 4 column tiles + tables do not fit the register budget: not built)
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
  q.H = p.H; q.W = p.W; q.ld = p.ld0; q.N = p.N; q.Kpad = p.Kpad; q.ldd = p.ldd;
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

// The 512 x 512 level (tail2_kernel): the bf16 orientation tail (32 channels -> 2 + F.normalize).  The kernel also exists for the
// localisation tails — bf16 (48 -> 1) and fp32 operands as hi + lo planes — and is correct there (measured at B = 64: 445 vs 376 us
// and 984 vs 639 us against tail512_kernel: at one wave per SIMD the ~1 400 non-matrix instructions per tile of the two epilogues,
// the in-LDS hi / lo conversion and the softmax partials are not hidden by anything), so only the form that wins is dispatched:
// 353 vs 393 us with two workgroups per CU.
int tail2_supported(int is_bf16, int split, int cout, int c0, int ld0, int h1, int w1, int batch) {
  if (!(is_bf16 && !split && c0 == 32 && cout == 2)) return 0;
  if (h1 % 8 || w1 % 16 || w1 < 16 || ld0 % 8) return 0;
  if ((long)batch * (h1 / 8) * (w1 / 16) < 2L * num_cus()) return 0;
  return 1;
}

int tail2_dispatch(int id, const void* x, const void* w, const float* shift9, const float* w2, const float* b2, float* out, float* smx,
                   int batch, int h1, int w1, int c0, int ld0, int kpad, int normalize, hipStream_t stream) {
  Tail2Params q{};
  q.x = x; q.w = w; q.shift9 = shift9; q.w2 = w2; q.b2 = b2; q.out = out; q.smx = smx;
  q.H1 = h1; q.W1 = w1; q.c0 = c0; q.ld0 = ld0; q.Kpad = kpad; q.normalize = normalize;
  if (id == 1) return launch_tail2<4, 2, false, 8>(q, batch, stream);
  return fail(CCVPE_EINVAL, "tail2: unsupported layer");
}

}  // namespace ccvpe
