// Convolution weight gradient on the fp32 matrix cores:
//   dW[n][tap][c] = sum_{m = (b,oy,ox)} dY[m][n] * X[b, oy*stride - pad + ky, ox*stride - pad + kx][c]
// (backward of F.conv2d w.r.t. its weight for every dense conv of the path: models.py:42-47,57-104 and
// efficientnet_pytorch/model.py:62,86,209; ConvTranspose2d's weight gradient is the same reduction with
// the roles of input and output gradient swapped, see ccvpe_amd/backward.py).
//
// GEMM view: rows = output channels n, columns = input channels c, reduction = OUTPUT PIXELS (millions),
// one GEMM per tap.  Both operands are pixel-major in HBM (NHWC), i.e. the reduction index is the slow
// one, so fragments are read from LDS one dword per lane (lane = (channel, pixel-in-group)): with the
// 32-cycle fp32 MFMA that costs nothing.  The pixel range is split over S workgroups per (tile, tap);
// partials go to scratch and a second kernel adds them in fixed order (deterministic, no atomics).
#include "common.h"

namespace ccvpe {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WgradParams {
  const float* src0;
  const float* src1;
  const float* dy;
  float* part;          // [S][N][taps][Ctot]
  int c0, ld0, c1, ld1;
  int H, W, Ho, Wo;
  int kw, stride, pad, taps;
  int N, ldy;
  int M;                // batch * Ho * Wo
  int S, pix_per_split;
  int tiles_n, tiles_c;
};

constexpr int WG_T = 64;    // tile: 64 output channels x 64 input channels
constexpr int WG_BP = 32;   // pixels per stage
constexpr int WG_LD = 68;   // LDS row stride (floats): 64 + 4

__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
  __shared__ __attribute__((aligned(16))) float Ys[2][WG_BP][WG_LD];   // dY  [pixel][n]
  __shared__ __attribute__((aligned(16))) float Xs[2][WG_BP][WG_LD];   // X   [pixel][c]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wc = wave & 1;          // wave tile: 32 n x 32 c
  int bid = blockIdx.x;
  const int tc = bid % p.tiles_c; bid /= p.tiles_c;
  const int tn = bid % p.tiles_n; bid /= p.tiles_n;
  const int tap = bid % p.taps;
  const int split = bid / p.taps;
  const int ky = tap / p.kw, kx = tap - ky * p.kw;
  const int n0 = tn * WG_T, cc0 = tc * WG_T;
  const int ctot = p.c0 + p.c1;
  const int m_begin = split * p.pix_per_split;
  const int m_end = min(m_begin + p.pix_per_split, p.M);

  // staging: thread -> (pixel row = tid >> 3 (0..31), 16-byte piece pair = tid & 7 -> pieces 2*(tid&7), +1)
  const int prow = tid >> 3;
  const int pc = (tid & 7) * 8;                    // first channel of this thread's two float4 pieces
  f32x4 yr[2], xr[2];

  auto load_stage = [&](int m0) {
    const int m = m0 + prow;
    const bool ok = m < m_end;
    int b = 0, oy = 0, ox = 0;
    if (ok) {
      const int hw = p.Ho * p.Wo;
      b = m / hw;
      const int rem = m - b * hw;
      oy = rem / p.Wo;
      ox = rem - oy * p.Wo;
    }
    const int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
    const bool inb = ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int n = n0 + pc + q * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok && n < p.N) {
        if (n + 3 < p.N) v = *reinterpret_cast<const f32x4*>(p.dy + (size_t)m * p.ldy + n);
        else
          for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] = p.dy[(size_t)m * p.ldy + n + r];
      }
      yr[q] = v;
      const int c = cc0 + pc + q * 4;
      f32x4 u = {0.f, 0.f, 0.f, 0.f};
      if (inb && c < ctot) {
        const size_t pix = (size_t)(b * p.H + iy) * p.W + ix;
        u = (c < p.c0) ? *reinterpret_cast<const f32x4*>(p.src0 + pix * p.ld0 + c)
                       : *reinterpret_cast<const f32x4*>(p.src1 + pix * p.ld1 + (c - p.c0));
      }
      xr[q] = u;
    }
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      *reinterpret_cast<f32x4*>(&Ys[buf][prow][pc + q * 4]) = yr[q];
      *reinterpret_cast<f32x4*>(&Xs[buf][prow][pc + q * 4]) = xr[q];
    }
  };

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fi = lane & 15, fq = lane >> 4;       // fragment: channel-in-tile, pixel-in-group
  const int nstage = (m_end - m_begin + WG_BP - 1) / WG_BP;
  if (nstage > 0) {
    load_stage(m_begin);
    store_stage(0);
  }
  __syncthreads();
  for (int s = 0; s < nstage; ++s) {
    const int buf = s & 1;
    const bool more = s + 1 < nstage;
    if (more) load_stage(m_begin + (s + 1) * WG_BP);
#pragma unroll
    for (int g = 0; g < WG_BP / 4; ++g) {
      float a[2], bb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = Ys[buf][g * 4 + fq][(wn * 2 + i) * 16 + fi];
#pragma unroll
      for (int j = 0; j < 2; ++j) bb[j] = Xs[buf][g * 4 + fq][(wc * 2 + j) * 16 + fi];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bb[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_stage(buf ^ 1);
    __syncthreads();
  }
  // D: row (n) = (lane>>4)*4 + reg, col (c) = lane&15
  float* out = p.part + (size_t)split * p.N * p.taps * ctot;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = cc0 + (wc * 2 + j) * 16 + (lane & 15);
      if (c >= ctot) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + (wn * 2 + i) * 16 + (lane >> 4) * 4 + r;
        if (n < p.N) out[((size_t)n * p.taps + tap) * ctot + c] = acc[i][j][r];
      }
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                           long n_elem, int S) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_elem) return;
  float s = 0.f;
  for (int k = 0; k < S; ++k) s += part[(size_t)k * n_elem + i];
  dw[i] = s;
}

// per-channel column sums (bias gradients): partial per 256-row block then fixed-order reduce
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int rows, int C, int ld,
                                                            float* __restrict__ part) {
  const int r0 = blockIdx.x * 256, r1 = min(r0 + 256, rows);
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += x[(size_t)r * ld + c];
    part[(size_t)blockIdx.x * C + c] = s;
  }
}

}  // namespace ccvpe

using namespace ccvpe;

static int wgrad_splits(int M, int tiles, int taps) {
  int S = 2048 / (tiles * taps > 0 ? tiles * taps : 1);
  if (S < 1) S = 1;
  const int maxS = (M + WG_BP * 8 - 1) / (WG_BP * 8);
  if (S > maxS) S = maxS;
  if (S > 256) S = 256;
  return S < 1 ? 1 : S;
}

extern "C" int ccvpe_conv_wgrad_scratch_floats(int batch, int in_h, int in_w, int kh, int kw, int stride, int pad,
                                               int ctot, int n) {
  const int Ho = (in_h + 2 * pad - kh) / stride + 1, Wo = (in_w + 2 * pad - kw) / stride + 1;
  const long M = (long)batch * Ho * Wo;
  if (M <= 0 || M > 0x7fffffffL || ctot <= 0 || n <= 0) return CCVPE_EINVAL;
  const int tiles = ((n + WG_T - 1) / WG_T) * ((ctot + WG_T - 1) / WG_T);
  const int S = wgrad_splits((int)M, tiles, kh * kw);
  const long fl = (long)S * n * kh * kw * ctot;
  return fl > 0x7fffffffL ? CCVPE_EINVAL : (int)fl;
}

extern "C" int ccvpe_conv_wgrad_f32(const float* src0, int c0, int ld0, const float* src1, int c1, int ld1,
                                    const float* dy, int ldy, float* dw, float* scratch, int batch, int in_h,
                                    int in_w, int kh, int kw, int stride, int pad, int n, void* stream) {
  if (c0 <= 0 || c0 % 4 || c1 < 0 || c1 % 4 || ld0 % 4 || (c1 && ld1 % 4)) return fail(CCVPE_EINVAL, "conv_wgrad: channels/ld %% 4");
  if (c1 > 0 && !src1) return fail(CCVPE_EINVAL, "conv_wgrad: c1>0 but src1 null");
  if (!aligned16(src0) || (src1 && !aligned16(src1)) || !aligned16(dy) || ldy % 4) return fail(CCVPE_EINVAL, "conv_wgrad: alignment");
  WgradParams p;
  p.src0 = src0; p.src1 = src1; p.dy = dy; p.part = scratch;
  p.c0 = c0; p.ld0 = ld0; p.c1 = c1; p.ld1 = ld1;
  p.H = in_h; p.W = in_w;
  p.Ho = (in_h + 2 * pad - kh) / stride + 1;
  p.Wo = (in_w + 2 * pad - kw) / stride + 1;
  p.kw = kw; p.stride = stride; p.pad = pad; p.taps = kh * kw;
  p.N = n; p.ldy = ldy;
  const long M = (long)batch * p.Ho * p.Wo;
  if (M <= 0 || M > 0x7fffffffL) return fail(CCVPE_EINVAL, "conv_wgrad: bad M");
  p.M = (int)M;
  const int ctot = c0 + c1;
  p.tiles_n = (n + WG_T - 1) / WG_T;
  p.tiles_c = (ctot + WG_T - 1) / WG_T;
  p.S = wgrad_splits(p.M, p.tiles_n * p.tiles_c, p.taps);
  p.pix_per_split = ((p.M + p.S - 1) / p.S + WG_BP - 1) / WG_BP * WG_BP;
  hipStream_t st = (hipStream_t)stream;
  const long blocks = (long)p.tiles_n * p.tiles_c * p.taps * p.S;
  hipLaunchKernelGGL(conv_wgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
  const long n_elem = (long)n * p.taps * ctot;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n_elem + 255) / 256)), dim3(256), 0, st, scratch, dw, n_elem, p.S);
  return check_launch("conv_wgrad");
}

extern "C" int ccvpe_colsum_f32(const float* x, int rows, int channels, int ld, float* out, float* scratch,
                                void* stream) {
  if (rows <= 0 || channels <= 0 || ld < channels) return fail(CCVPE_EINVAL, "colsum: bad shape");
  const int nblk = (rows + 255) / 256;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), 0, st, x, rows, channels, ld, scratch);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((channels + 255) / 256), dim3(256), 0, st, scratch, out, (long)channels, nblk);
  return check_launch("colsum");
}
