// Fused MBConv front half: expand 1x1 + BN0 + swish -> depthwise KxK (stride S) + BN1 + swish, plus
// the SE squeeze partial sums.  efficientnet_pytorch/model.py:102-110,114.
//
// Why: the 6x-expanded tensor is the largest in the network (25 MB fp32 per aerial sample at block 1)
// and the unfused path writes it and reads it back (about half of all encoder HBM bytes).  Here it
// lives only in LDS.
//
// Mapping: a workgroup owns (sample b, 16-channel chunk of `mid`, band of RB output rows).
//   * expand phase: one FULL-WIDTH input row at a time.  Wave w takes 16-pixel tiles w, w+4, ...;
//     the x fragment is loaded straight from HBM in MFMA operand layout (lane = pixel l&15, k group
//     l>>4 -> one dwordx4 of 4 consecutive input channels; 16 pixels x 64 B contiguous), the 16xCin
//     weight fragment stays in registers, v_mfma_f32_16x16x4_f32 accumulates, BN0+swish, and the
//     result is written as one ds_write_b128 per lane into a ring of K rows [W][16] in LDS.
//     Rows outside the image are ZERO rows (the depthwise conv pads the EXPANDED tensor).
//   * depthwise phase: thread = (4 channels, pixel); KxK taps read the ring (horizontal circular
//     padding is an index wrap because the ring holds full rows; no halo recompute), BN1+swish,
//     dwordx4 store, running per-channel sum for the squeeze.
// Vertical overlap between bands costs (K-S)/(RB*S) extra expand rows.
#include "common.h"
#include <type_traits>

namespace ccvpe {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MbFrontParams {
  const void* x;
  const void* w_exp;
  const float* s0;
  const float* b0;
  const float* w_dw;
  const float* s1;
  const float* b1;
  void* y;
  float* se_partial;
  int H, W, Cin, kpad, mid, Ho, Wo, circular;
  int rb, nbands, nchunks, total_blocks;
};

constexpr int MBF_MAX_KK = 3;   // Cin <= 48
constexpr int MBF_MAXT = 5;     // 16-pixel tiles per wave per row: W <= 320
// The kernel is instantiated per (tiles per wave T, k-chunks NKK) so that the register-resident row
// costs T*NKK dwordx4 only: narrow rows keep their occupancy (a fixed T=5 needed 199 VGPRs).

// TE = storage type of x / w_exp / y (float or bf16).  NKK counts 64-byte K pieces: 16 fp32 or 32 bf16
// input channels each; the ring (expanded tensor) and all depthwise math stay fp32 either way.
template <typename TE, int K, int S, int T, int NKK>
__global__ __launch_bounds__(256) void mbconv_front_kernel(const MbFrontParams p) {
  constexpr int E = 16 / sizeof(TE);
  constexpr int SK = 4 * E;
  constexpr int PB = (S == 1) ? (K - 1) / 2 : (K - 2) / 2;   // pad before (224-schedule SAME)
  extern __shared__ __attribute__((aligned(16))) float sm[];
  // ring rows are PADDED by the horizontal halo (PB pixels before, K-1-PB after): the pads hold zeros (zero padding) or
  // the wrapped pixels (circular padding), so the depthwise taps read ring[ox*S + kx] with NO bounds test — a per-tap
  // `if (ix < W)` made every tap its own basic block (exec-mask branch, LDS read, s_waitcnt, FMA): ~240 cycles per tap.
  const int RW = p.W + K - 1;                        // ring row width in pixels
  float* ring = sm;                                  // [K][RW][16]
  float* wdw = ring + max(K * RW * 16, 1024);        // [K*K][16] (ring region >= 1024 floats: red aliases it)
  float* red = ring;                                 // [256][4], aliases the ring after the last row

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  int lb;
  {
    const int q = p.total_blocks / 8, r = p.total_blocks % 8;
    const int xcd = blockIdx.x % 8, loc = blockIdx.x / 8;
    lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int chunk = lb % p.nchunks;                  // chunk fastest: neighbours re-read the same x rows
  const int band = (lb / p.nchunks) % p.nbands;
  const int b = lb / (p.nchunks * p.nbands);
  const int c0 = chunk * 16;
  const int oy0 = band * p.rb;
  const int oy1 = min(oy0 + p.rb, p.Ho);
  const int r_begin = oy0 * S - PB;

  // depthwise weights of this chunk -> LDS
  for (int i = tid; i < K * K * 16; i += 256) wdw[i] = p.w_dw[(size_t)(i >> 4) * p.mid + c0 + (i & 15)];

  // expand weights (MFMA A operand: row n = lane&15, k group = lane>>4) and BN0 in registers
  const int q4 = (lane >> 4) * 4;
  constexpr int nkk = NKK;
  f32x4 wf[NKK];
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) {
    wf[kk] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (kk < nkk)
      wf[kk] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const TE*>(p.w_exp) + (size_t)(c0 + (lane & 15)) * p.kpad +
                                               kk * SK + (lane >> 4) * E);
  }
  const f32x4 sc0 = *reinterpret_cast<const f32x4*>(p.s0 + c0 + q4);
  const f32x4 sh0 = *reinterpret_cast<const f32x4*>(p.b0 + c0 + q4);

  const int ntile = (p.W + 15) >> 4;
  const TE* xb = reinterpret_cast<const TE*>(p.x) + (size_t)b * p.H * p.W * p.Cin;
  const int last_needed = (oy1 - 1) * S - PB + K - 1;

  // Register-resident x fragments of D = S input rows for this wave's tiles (<= MBF_MAXT tiles x <= 3 k-chunks each): row r
  // lives in set (r - r_begin) % D and the loads of row r + D are issued right after the MFMAs of row r consumed that set.
  // An output row consumes S new input rows: with one set (the first version) the SECOND row of a stride-2 pair was loaded
  // a few hundred cycles before its use (one exposed L2/HBM round trip per output row: expand was 3 000 cycles per input
  // row against ~900 of work); with S sets every row's loads are issued one whole depthwise phase ahead.
  // Loads are branch-free (clamped address, result AND-ed with a lane mask): the compiler turns a guarded load into an
  // exec-mask branch per load.
  constexpr int D = S;
  constexpr int KY_UNROLL = K == 3 ? 3 : 1;
  f32x4 xv[D][T][NKK];
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  auto prefetch_row = [&](int iy, auto set_tag) {
    constexpr int SET = decltype(set_tag)::value;
    const bool rowok = (unsigned)iy < (unsigned)p.H && iy <= last_needed;
    const TE* xr = xb + (size_t)(rowok ? iy : 0) * p.W * p.Cin;
#pragma unroll
    for (int ti = 0; ti < T; ++ti) {
      const int px = (wave + 4 * ti) * 16 + (lane & 15);
      const int pxc = px < p.W ? px : p.W - 1;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        const int ch = kk * SK + (lane >> 4) * E;
        const int m = (rowok && px < p.W && ch < p.Cin) ? -1 : 0;
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + (size_t)pxc * p.Cin + (ch < p.Cin ? ch : 0));
        xv[SET][ti][kk] = __builtin_bit_cast(f32x4, __builtin_bit_cast(i32x4, v) & (i32x4){m, m, m, m});
      }
    }
  };

  auto produce_row_t = [&](int iy, auto set_tag) {
    constexpr int SET = decltype(set_tag)::value;
    float* dst = ring + (size_t)((iy - r_begin) % K) * RW * 16 + PB * 16;      // pixel 0 of the row (after the left pad)
    if ((unsigned)iy >= (unsigned)p.H) {
      for (int i = tid; i < RW * 4; i += 256) *reinterpret_cast<f32x4*>(dst - PB * 16 + i * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
      prefetch_row(iy + D, set_tag);                     // keep the pipeline going across padding rows
      return;
    }
    f32x4 acc[T];
#pragma unroll
    for (int ti = 0; ti < T; ++ti) {
      acc[ti] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (wave + 4 * ti < ntile) {
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk)
          if (kk < nkk) {
            if (sizeof(TE) == 4) {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                acc[ti] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kk][r], xv[SET][ti][kk][r], acc[ti], 0, 0, 0);
            } else {
              acc[ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cc_bf16x8, wf[kk]),
                                                                __builtin_bit_cast(cc_bf16x8, xv[SET][ti][kk]), acc[ti], 0, 0, 0);
            }
          }
      }
    }
    prefetch_row(iy + D, set_tag);
    // D: row = channel 4*(lane>>4)+reg, col = pixel lane&15
#pragma unroll
    for (int ti = 0; ti < T; ++ti) {
      const int px = (wave + 4 * ti) * 16 + (lane & 15);
      if (wave + 4 * ti < ntile && px < p.W) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = swishf(acc[ti][r] * sc0[r] + sh0[r]);
        *reinterpret_cast<f32x4*>(dst + (size_t)px * 16 + q4) = o;
        if (p.circular) {                               // wrapped copies into the pads (utils.py:350: F.pad(..., 'circular') on W)
          if (px < K - 1 - PB) *reinterpret_cast<f32x4*>(dst + (size_t)(px + p.W) * 16 + q4) = o;
          if (px >= p.W - PB) *reinterpret_cast<f32x4*>(dst + (size_t)(px - p.W) * 16 + q4) = o;
        }
      }
    }
  };
  auto produce_row = [&](int iy) {
    if (D == 1 || ((iy - r_begin) & 1) == 0) produce_row_t(iy, std::integral_constant<int, 0>{});
    else produce_row_t(iy, std::integral_constant<int, D - 1>{});
  };
  // the pads of all K ring rows start as zeros (only circular padding ever overwrites them)
  for (int i = tid; i < (p.circular ? 0 : K * (K - 1) * 4); i += 256) {
    const int row = i / ((K - 1) * 4), rem = i - row * (K - 1) * 4;
    const int pp = rem >> 2, q = rem & 3;                           // pad pixel 0..K-2: first PB on the left, rest on the right
    const int col = pp < PB ? pp : p.W + pp;
    *reinterpret_cast<f32x4*>(ring + ((size_t)row * RW + col) * 16 + q * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  prefetch_row(r_begin, std::integral_constant<int, 0>{});   // fill the pipeline (padding rows load nothing)
  if (D > 1) prefetch_row(r_begin + 1, std::integral_constant<int, D - 1>{});

  // depthwise coordinates
  const int cg = tid & 3;
  const int pxl = tid >> 2;
  const f32x4 sc1 = *reinterpret_cast<const f32x4*>(p.s1 + c0 + cg * 4);
  const f32x4 sh1 = *reinterpret_cast<const f32x4*>(p.b1 + c0 + cg * 4);
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};

  int next_row = r_begin;
  for (int oy = oy0; oy < oy1; ++oy) {
    const int need = oy * S - PB + K - 1;
    while (next_row <= need) {
      produce_row(next_row);
      ++next_row;
    }
    __syncthreads();
    for (int ox = pxl; ox < p.Wo; ox += 64) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      // k = 5: one kernel row (5 taps = 10 LDS reads in flight) at a time — fully unrolled, the branch-free loop lets the
      // scheduler hoist all 50 reads and the kernel needs 200-256 VGPRs (1-2 waves per SIMD)
#pragma unroll KY_UNROLL
      for (int ky = 0; ky < K; ++ky) {
        const int iy = oy * S - PB + ky;
        const float* rrow = ring + (size_t)((iy - r_begin) % K) * RW * 16 + (size_t)(ox * S) * 16 + cg * 4;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(rrow + kx * 16);
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wdw + (ky * K + kx) * 16 + cg * 4);
          acc += v * wv;
        }
      }
      f32x4 o = acc * sc1 + sh1;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = swishf(o[r]);
      st4<TE>(reinterpret_cast<TE*>(p.y) + ((size_t)(b * p.Ho + oy) * p.Wo + ox) * p.mid + c0 + cg * 4, o);
      sum += o;
    }
    __syncthreads();
  }

  // squeeze partial of this (band, chunk): fixed-order reduction over the 64 pixel lanes
  *reinterpret_cast<f32x4*>(red + tid * 4) = sum;   // safe: the loop ended with a barrier, ring is dead
  __syncthreads();
  if (tid < 4) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 64; ++i) t += *reinterpret_cast<const f32x4*>(red + (i * 4 + tid) * 4);
    *reinterpret_cast<f32x4*>(p.se_partial + ((size_t)b * p.nbands + band) * p.mid + c0 + tid * 4) = t;
  }
}

constexpr int MBF_RB = 16;

static bool mbf_supported(int W, int cin, int mid, int k, int sk = 16) {
  return cin % 8 == 0 && cin <= sk * MBF_MAX_KK && mid % 16 == 0 && W >= 1 &&
         W <= 64 * MBF_MAXT && (size_t)((k * (W + k - 1) * 16 > 1024 ? k * (W + k - 1) * 16 : 1024) + k * k * 16) * 4 <= 64 * 1024;
}

}  // namespace ccvpe

using namespace ccvpe;

extern "C" int ccvpe_mbconv_front_nblk(int in_h, int in_w, int cin, int mid, int k, int stride) {
  if (!(k == 3 || k == 5) || !(stride == 1 || stride == 2)) return CCVPE_EINVAL;
  if (!mbf_supported(in_w, cin, mid, k)) return 0;           // 0 => use the unfused kernels
  const int total_pad = (stride == 1) ? (k - 1) : (k - 2);
  const int Ho = (in_h + total_pad - k) / stride + 1;
  return (Ho + MBF_RB - 1) / MBF_RB;
}

template <typename TE>
static int mbconv_front_any(const void* x, const void* w_exp, int kpad, const float* s0, const float* b0,
                            const float* w_dw, const float* s1, const float* b1, void* y, float* se_partial, int B,
                            int H, int W, int cin, int mid, int k, int stride, int circular, void* stream) {
  constexpr int SK = 4 * (16 / (int)sizeof(TE));
  if (!(k == 3 || k == 5) || !(stride == 1 || stride == 2)) return fail(CCVPE_EINVAL, "mbconv_front: k/stride unsupported");
  if (!mbf_supported(W, cin, mid, k, SK)) return fail(CCVPE_EINVAL, "mbconv_front: shape not supported (W=%d cin=%d mid=%d)", W, cin, mid);
  if (kpad % SK || kpad < cin) return fail(CCVPE_EINVAL, "mbconv_front: bad kpad");
  if (!aligned16(x) || !aligned16(w_exp) || !aligned16(s0) || !aligned16(b0) || !aligned16(s1) || !aligned16(b1) ||
      !aligned16(y) || !aligned16(se_partial))
    return fail(CCVPE_EINVAL, "mbconv_front: pointers must be 16-byte aligned");
  MbFrontParams p;
  p.x = x; p.w_exp = w_exp; p.s0 = s0; p.b0 = b0; p.w_dw = w_dw; p.s1 = s1; p.b1 = b1; p.y = y; p.se_partial = se_partial;
  p.H = H; p.W = W; p.Cin = cin; p.kpad = kpad; p.mid = mid; p.circular = circular;
  const int total_pad = (stride == 1) ? (k - 1) : (k - 2);
  p.Ho = (H + total_pad - k) / stride + 1;
  p.Wo = (W + total_pad - k) / stride + 1;
  p.rb = MBF_RB;
  p.nbands = (p.Ho + MBF_RB - 1) / MBF_RB;
  p.nchunks = mid / 16;
  const long total = (long)p.nbands * p.nchunks * B;
  if (total > 0x7fffffffL) return fail(CCVPE_EINVAL, "mbconv_front: grid too large");
  p.total_blocks = (int)total;
  const size_t smem = (size_t)((k * (W + k - 1) * 16 > 1024 ? k * (W + k - 1) * 16 : 1024) + k * k * 16) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  const int tneed = ((W + 15) / 16 + 3) / 4;          // 16-pixel tiles per wave per row
  const int tsel = tneed <= 1 ? 1 : (tneed <= 2 ? 2 : (tneed <= 3 ? 3 : 5));
  const int nkk = (cin + SK - 1) / SK;
#define MBF_LAUNCH(K_, S_, T_, N_) \
  hipLaunchKernelGGL((mbconv_front_kernel<TE, K_, S_, T_, N_>), dim3(p.total_blocks), dim3(256), smem, st, p)
#define MBF_N(K_, S_, T_)                    \
  if (nkk == 1) MBF_LAUNCH(K_, S_, T_, 1);   \
  else if (nkk == 2) MBF_LAUNCH(K_, S_, T_, 2); \
  else MBF_LAUNCH(K_, S_, T_, 3)
#define MBF_T(K_, S_)                        \
  if (tsel == 1) { MBF_N(K_, S_, 1); }       \
  else if (tsel == 2) { MBF_N(K_, S_, 2); }  \
  else if (tsel == 3) { MBF_N(K_, S_, 3); }  \
  else { MBF_N(K_, S_, 5); }
  if (k == 3 && stride == 1) { MBF_T(3, 1) }
  else if (k == 3 && stride == 2) { MBF_T(3, 2) }
  else if (k == 5 && stride == 1) { MBF_T(5, 1) }
  else { MBF_T(5, 2) }
#undef MBF_T
#undef MBF_N
#undef MBF_LAUNCH
  return check_launch("mbconv_front_kernel");
}

extern "C" int ccvpe_mbconv_front_f32(const float* x, const float* w_exp, int kpad, const float* s0, const float* b0,
                                      const float* w_dw, const float* s1, const float* b1, float* y,
                                      float* se_partial, int B, int H, int W, int cin, int mid, int k, int stride,
                                      int circular, void* stream) {
  return mbconv_front_any<float>(x, w_exp, kpad, s0, b0, w_dw, s1, b1, y, se_partial, B, H, W, cin, mid, k, stride,
                                 circular, stream);
}
extern "C" int ccvpe_mbconv_front_bf16(const void* x, const void* w_exp, int kpad, const float* s0, const float* b0,
                                       const float* w_dw, const float* s1, const float* b1, void* y,
                                       float* se_partial, int B, int H, int W, int cin, int mid, int k, int stride,
                                       int circular, void* stream) {
  return mbconv_front_any<cc_bf16>(x, w_exp, kpad, s0, b0, w_dw, s1, b1, y, se_partial, B, H, W, cin, mid, k, stride,
                                   circular, stream);
}
