// Implicit-GEMM convolution on the gfx950 fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
//   out[m, n] = act((sum_k A[m, k] * Wp[n, k]) * scale[n] + shift[n]) (+ residual[m, n])
//
// m = output pixel (b, oy, ox); k = (tap, concat channel) walked in 8-channel chunks so that a
// K-stage of 16 floats may straddle a tap or the boundary between the two concatenated sources.
// Replaces F.conv2d / F.conv_transpose2d / nn.Linear call sites of the reference:
// efficientnet_pytorch/model.py:104-106,121-130,299 ; models.py:42-47,57-97,102-148,173-184.
//
// Tile: 256 threads = 4 waves laid out WM x WN; each wave owns (16*MT) x (16*NT) outputs as MT*NT
// accumulators of the 16x16x4 MFMA.  A and W stages ([rows][16 k] fp32) go global -> VGPR -> LDS
// (register staging: the gather needs per-row predication, zero fill and the SE gate multiply,
// which a lane-linear LDS-DMA cannot do) and are double buffered: the loads of stage s+1 are
// issued before the MFMAs of stage s, the LDS writes after them, one barrier per stage.
// K permutation: lane group q = lane>>4 consumes k in {4q..4q+3} over the 4 MFMAs of a stage, so
// each operand fragment is ONE ds_read_b128; A and W use the same map, so the sum is unchanged.
// LDS rows are padded 16 -> 20 floats (80 B: 16-byte aligned, breaks the 64 B power-of-two stride).
//
// The fp32 MFMA runs at the fp32 vector rate (157 TF peak); it is a bitwise k-ordered fmaf
// chain, i.e. this kernel is exact fp32 (no TF32-like shortcut exists on gfx950).
#include "common.h"

namespace ccvpe {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct IgemmParams {
  const float* src0;
  const float* src1;
  const float* gate;
  const float* w;
  const float* scale;
  const float* shift;
  const float* residual;
  float* dst;
  int c0, ld0, c1, ld1;
  int H, W, Ho, Wo;
  int kw, stride, pad;
  int N, Kpad, Npad;
  int cpt0, cpt, total_chunks, stages;
  int ldd, ldres, act, out_mode, cout;
  int M;
  int tiles_n, tiles_total;
};

constexpr int LDS_LD = 20;  // floats per staged row (16 + 4 pad)

template <int MT, int NT, int WN>
__global__ __launch_bounds__(256) void igemm_f32_kernel(const IgemmParams p) {
  constexpr int WM = 4 / WN;
  constexpr int BM = 16 * MT * WM;
  constexpr int BN = 16 * NT * WN;
  constexpr int A_IT = BM / 64;            // float4 loads per thread per stage (A)
  constexpr int B_IT = (BN + 63) / 64;     // float4 loads per thread per stage (W)

  __shared__ __attribute__((aligned(16))) float As[2][BM][LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN][LDS_LD];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave % WN;

  // XCD-aware tile order: consecutive workgroup ids round-robin over the 8 XCDs, so give each XCD
  // a contiguous run of tiles (n fastest): the N-tiles that re-read one A panel share an L2.
  int tile;
  {
    const int bid = blockIdx.x;
    const int nx = 8;
    const int q = p.tiles_total / nx, r = p.tiles_total % nx;
    const int xcd = bid % nx, loc = bid / nx;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tm = tile / p.tiles_n;
  const int tn = tile % p.tiles_n;
  const int m0 = tm * BM;
  const int n0 = tn * BN;

  // ---- per-thread staging coordinates -------------------------------------------------------
  const int srow = tid >> 2;   // 0..63
  const int ssub = tid & 3;    // which float4 of the 16-float stage row
  const int chunk_in_stage = ssub >> 1;
  const int half = ssub & 1;

  int a_b[A_IT], a_y[A_IT], a_x[A_IT];
  bool a_ok[A_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int m = m0 + srow + 64 * it;
    a_ok[it] = m < p.M;
    const int mm = a_ok[it] ? m : 0;
    const int hw = p.Ho * p.Wo;
    const int b = mm / hw;
    const int rem = mm - b * hw;
    const int oy = rem / p.Wo;
    const int ox = rem - oy * p.Wo;
    a_b[it] = b;
    a_y[it] = oy * p.stride - p.pad;
    a_x[it] = ox * p.stride - p.pad;
  }
  // chunk cursor of this thread: kc = 2*stage + chunk_in_stage, decoded as (ky, kx, r)
  int kc = chunk_in_stage;
  int r = chunk_in_stage, ky = 0, kx = 0;
  while (r >= p.cpt) {  // cpt may be 1 (never for our shapes, but keep it correct)
    r -= p.cpt;
    if (++kx == p.kw) { kx = 0; ++ky; }
  }

  f32x4 a_reg[A_IT], b_reg[B_IT];

  auto load_stage = [&](int s) {
    // A operand
    const bool kvalid = kc < p.total_chunks;
    const bool from0 = r < p.cpt0;
    const float* base = from0 ? p.src0 : p.src1;
    const int ld = from0 ? p.ld0 : p.ld1;
    const int ch = (from0 ? r : r - p.cpt0) * 8 + half * 4;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int iy = a_y[it] + ky;
      const int ix = a_x[it] + kx;
      const bool ok = a_ok[it] && kvalid && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) {
        const size_t off = ((size_t)(a_b[it] * p.H + iy) * p.W + ix) * ld + ch;
        v = *reinterpret_cast<const f32x4*>(base + off);
        if (p.gate != nullptr && from0) {
          const f32x4 gv = *reinterpret_cast<const f32x4*>(p.gate + (size_t)a_b[it] * p.c0 + ch);
          v *= gv;
        }
      }
      a_reg[it] = v;
    }
    // W operand
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int nrow = srow + 64 * it;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (nrow < BN && n0 + nrow < p.Npad)
        v = *reinterpret_cast<const f32x4*>(p.w + (size_t)(n0 + nrow) * p.Kpad + s * 16 + ssub * 4);
      b_reg[it] = v;
    }
    // advance the chunk cursor by one stage (2 chunks)
    kc += 2;
    r += 2;
    while (r >= p.cpt) {
      r -= p.cpt;
      if (++kx == p.kw) { kx = 0; ++ky; }
    }
  };

  auto store_stage = [&](int buf) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it)
      *reinterpret_cast<f32x4*>(&As[buf][srow + 64 * it][ssub * 4]) = a_reg[it];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int nrow = srow + 64 * it;
      if (nrow < BN) *reinterpret_cast<f32x4*>(&Bs[buf][nrow][ssub * 4]) = b_reg[it];
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;
  const int fk = (lane >> 4) * 4;

  load_stage(0);
  store_stage(0);
  __syncthreads();

  for (int s = 0; s < p.stages; ++s) {
    const int buf = s & 1;
    const bool more = s + 1 < p.stages;
    if (more) load_stage(s + 1);

    f32x4 af[MT], bf[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
      af[i] = *reinterpret_cast<const f32x4*>(&As[buf][(wm * MT + i) * 16 + frow][fk]);
#pragma unroll
    for (int j = 0; j < NT; ++j)
      bf[j] = *reinterpret_cast<const f32x4*>(&Bs[buf][(wn * NT + j) * 16 + frow][fk]);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][kk], bf[j][kk], acc[i][j], 0, 0, 0);

    if (more) store_stage(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C layout of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg -----------
  const int ecol = lane & 15;
  const int erow = (lane >> 4) * 4;
  float sc[NT], sh[NT];
  int ncol[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + (wn * NT + j) * 16 + ecol;
    ncol[j] = n;
    const bool ok = n < p.N;
    sc[j] = (ok && p.scale) ? p.scale[n] : 1.0f;
    sh[j] = (ok && p.shift) ? p.shift[n] : 0.0f;
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int m = m0 + (wm * MT + i) * 16 + erow + rg;
      if (m >= p.M) continue;
      size_t obase;
      if (p.out_mode == CCVPE_OUT_NHWC) {
        obase = (size_t)m * p.ldd;
      } else {
        const int hw = p.Ho * p.Wo;
        const int b = m / hw;
        const int rem = m - b * hw;
        const int y = rem / p.Wo;
        const int x = rem - y * p.Wo;
        obase = ((size_t)(b * 2 * p.Ho + 2 * y) * (2 * p.Wo) + 2 * x) * p.ldd;
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = ncol[j];
        if (n >= p.N) continue;
        float v = acc[i][j][rg] * sc[j] + sh[j];
        if (p.act == CCVPE_ACT_RELU) v = fmaxf(v, 0.0f);
        else if (p.act == CCVPE_ACT_SWISH) v = v / (1.0f + expf(-v));
        if (p.out_mode == CCVPE_OUT_NHWC) {
          if (p.residual) v += p.residual[(size_t)m * p.ldres + n];
          p.dst[obase + n] = v;
        } else {
          const int quad = n / p.cout;
          const int co = n - quad * p.cout;
          const size_t o = obase + ((size_t)(quad >> 1) * (2 * p.Wo) + (quad & 1)) * p.ldd + co;
          p.dst[o] = v;
        }
      }
    }
  }
}

template <int MT, int NT, int WN>
static int launch(const IgemmParams& p0, hipStream_t stream) {
  constexpr int WM = 4 / WN;
  constexpr int BM = 16 * MT * WM;
  constexpr int BN = 16 * NT * WN;
  IgemmParams p = p0;
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.Npad + BN - 1) / BN;
  p.tiles_total = tiles_m * p.tiles_n;
  hipLaunchKernelGGL((igemm_f32_kernel<MT, NT, WN>), dim3(p.tiles_total), dim3(256), 0, stream, p);
  return check_launch("igemm_f32_kernel");
}

// Pick the N tile that wastes the fewest MFMA columns, then the widest.
struct TileCfg { int mt, nt, wn; };
static const TileCfg kCfgs[] = {
    {4, 5, 2}, {4, 4, 2}, {4, 3, 2}, {4, 2, 2}, {4, 1, 2},  // BN 160,128,96,64,32  BM 128
    {4, 5, 1}, {4, 3, 1}, {4, 1, 1},                        // BN 80,48,16          BM 256
    {2, 7, 1},                                              // BN 112               BM 128
};

static int pick_cfg(int npad16) {
  int best = 0;
  long best_cost = -1;
  for (int i = 0; i < (int)(sizeof(kCfgs) / sizeof(kCfgs[0])); ++i) {
    const int bn = 16 * kCfgs[i].nt * kCfgs[i].wn;
    const int tiles = (npad16 + bn - 1) / bn;
    // cost: padded columns first, then prefer wide tiles (fewer A re-reads)
    const long cost = (long)tiles * bn * 1000 + (1000 - bn);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = i; }
  }
  return best;
}

}  // namespace ccvpe

using namespace ccvpe;

extern "C" int ccvpe_conv_igemm_f32(const ccvpe_conv_desc* d, void* stream) {
  if (!d) return fail(CCVPE_EINVAL, "conv_igemm: null desc");
  if (d->c0 <= 0 || d->c0 % 8 || d->c1 < 0 || d->c1 % 8)
    return fail(CCVPE_EINVAL, "conv_igemm: c0/c1 must be multiples of 8 (got %d,%d)", d->c0, d->c1);
  if (d->c1 > 0 && !d->src1) return fail(CCVPE_EINVAL, "conv_igemm: c1>0 but src1 null");
  if (d->ld0 % 4 || (d->c1 && d->ld1 % 4) || d->kpad % 16)
    return fail(CCVPE_EINVAL, "conv_igemm: ld0/ld1 %% 4, kpad %% 16 required");
  if (!aligned16(d->src0) || (d->src1 && !aligned16(d->src1)) || !aligned16(d->w) ||
      (d->gate && !aligned16(d->gate)))
    return fail(CCVPE_EINVAL, "conv_igemm: pointers must be 16-byte aligned");
  if (d->gate && (d->kh != 1 || d->c1 != 0)) return fail(CCVPE_EINVAL, "conv_igemm: gate only for 1x1 single-source");
  if (d->stride < 1 || d->kh < 1 || d->kw < 1) return fail(CCVPE_EINVAL, "conv_igemm: bad kernel/stride");
  IgemmParams p;
  p.src0 = d->src0; p.src1 = d->src1; p.gate = d->gate; p.w = d->w;
  p.scale = d->scale; p.shift = d->shift; p.residual = d->residual; p.dst = d->dst;
  p.c0 = d->c0; p.ld0 = d->ld0; p.c1 = d->c1; p.ld1 = d->ld1;
  p.H = d->in_h; p.W = d->in_w;
  p.Ho = (d->in_h + 2 * d->pad - d->kh) / d->stride + 1;
  p.Wo = (d->in_w + 2 * d->pad - d->kw) / d->stride + 1;
  p.kw = d->kw; p.stride = d->stride; p.pad = d->pad;
  p.N = d->n; p.Kpad = d->kpad; p.Npad = (d->n + 15) / 16 * 16;
  p.cpt0 = d->c0 / 8; p.cpt = (d->c0 + d->c1) / 8;
  p.total_chunks = p.cpt * d->kh * d->kw;
  if (p.total_chunks * 8 > p.Kpad) return fail(CCVPE_EINVAL, "conv_igemm: kpad %d < K %d", p.Kpad, p.total_chunks * 8);
  p.stages = (p.total_chunks + 1) / 2;
  p.ldd = d->ldd; p.ldres = d->ldres; p.act = d->act; p.out_mode = d->out_mode;
  p.cout = (d->out_mode == CCVPE_OUT_DECONV2X) ? d->n / 4 : d->n;
  if (d->out_mode == CCVPE_OUT_DECONV2X && (d->n % 4 || d->residual))
    return fail(CCVPE_EINVAL, "conv_igemm: deconv mode needs n%%4==0 and no residual");
  const long M = (long)d->batch * p.Ho * p.Wo;
  if (M <= 0 || M > 0x7fffffffL) return fail(CCVPE_EINVAL, "conv_igemm: bad M");
  p.M = (int)M;
  p.tiles_n = p.tiles_total = 0;
  hipStream_t st = (hipStream_t)stream;
  const TileCfg c = kCfgs[pick_cfg(p.Npad)];
#define CCVPE_CASE(MT_, NT_, WN_) \
  if (c.mt == MT_ && c.nt == NT_ && c.wn == WN_) return launch<MT_, NT_, WN_>(p, st);
  CCVPE_CASE(4, 5, 2) CCVPE_CASE(4, 4, 2) CCVPE_CASE(4, 3, 2) CCVPE_CASE(4, 2, 2) CCVPE_CASE(4, 1, 2)
  CCVPE_CASE(4, 5, 1) CCVPE_CASE(4, 3, 1) CCVPE_CASE(4, 1, 1) CCVPE_CASE(2, 7, 1)
#undef CCVPE_CASE
  return fail(CCVPE_EINVAL, "conv_igemm: no tile config");
}
