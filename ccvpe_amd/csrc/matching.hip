// Rotational matching + LMU concat, fused into one pass over the aerial feature volume.
//
// Reference pattern (models.py:186-205, repeated at :211-228 ... :299-315; ori_prior :484-514;
// KITTI :788-806): per rotation hypothesis i it materialises roll(X, -i*stride), slices the first
// L channels, takes a norm, a repeat()-ed product, a sum and a division, then stacks, maxes,
// F.normalize()s X again and cats — about 20x the algorithmic traffic.
//
// Here: one kernel reads X once from HBM (second touch of the tile comes from L2), and writes
// the score volume (NCHW, returned to the caller) and the decoder input
// [X/||X||, max score, (level-1 orientation scores), 0-pad] exactly once.
//
// HBM-bound: bytes/pixel = 4*C read + 4*ldo write + 4*n_shifts scores; the arithmetic
// (n_shifts*C FMAs per pixel) is <1% of the forward's FLOPs, so it stays on the VALU.
//
// Mapping: lane = pixel.  A 256-pixel x 32-channel tile is loaded coalesced (128 B per pixel row)
// and transposed into LDS so that the per-lane channel walk is conflict-free; the ground
// descriptor is shared by every pixel of a sample, so it sits in LDS as a doubled table
// gg[k] = g_ext[k mod C] (k < 2C, g_ext = g zero-extended to C) and each FMA's second operand is
// a wave-uniform LDS broadcast:  dot_i = sum_c' x[c'] * gg[c' + off_i],  off_i = (-i*stride) mod C.
// For partial windows (L < C: FoV < 360, KITTI) the window norm uses the same trick with a 0/1
// table; for L == C it is the pixel's total norm.
#include "common.h"

namespace ccvpe {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct MatchOffsets {
  int off[CCVPE_MAX_SHIFTS];
};

constexpr int MCK = 32;  // channels per staged tile

// TX = storage type of X and of the concat output (float or bf16); scores, g and all math are fp32.
template <typename TX, int NPAD, bool PARTIAL, int VEC>
__global__ __launch_bounds__(256) void match_kernel(const TX* __restrict__ x, int ldx, const float* __restrict__ g,
                                                    int ldg, int L, const MatchOffsets mo, int n_shifts, int n_max,
                                                    int n_tail, float* __restrict__ scores,
                                                    TX* __restrict__ dstx, int ldo, int hw, int C) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int TPB = blockDim.x;
  const int XLD = TPB + 1;
  float* gg = sm;                                   // [2C]
  float* ww = gg + 2 * C;                           // [2C] (PARTIAL only)
  float* xs = ww + (PARTIAL ? 2 * C : 0);           // [MCK][XLD]
  float* inv_s = xs + MCK * XLD;                    // [TPB]
  float* red = inv_s + TPB;                         // [4]

  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * TPB;
  const int p = p0 + tid;
  const bool pvalid = p < hw;
  const TX* xb = x + (size_t)b * hw * ldx;

  // descriptor tables + ||g||
  float gsq = 0.f;
  for (int k = tid; k < 2 * C; k += TPB) {
    const int kk = k < C ? k : k - C;
    const float v = kk < L ? g[(size_t)b * ldg + kk] : 0.f;
    gg[k] = v;
    if (PARTIAL) ww[k] = kk < L ? 1.f : 0.f;
    if (k < C) gsq = fmaf(v, v, gsq);
  }
  gsq = wave_sum(gsq);
  if ((tid & 63) == 0) red[tid >> 6] = gsq;
  __syncthreads();
  float gnorm = 0.f;
  for (int i = 0; i < (TPB >> 6); ++i) gnorm += red[i];
  gnorm = sqrtf(gnorm);

  float acc[NPAD], nrm[PARTIAL ? NPAD : 1];
#pragma unroll
  for (int i = 0; i < NPAD; ++i) acc[i] = 0.f;
#pragma unroll
  for (int i = 0; i < (PARTIAL ? NPAD : 1); ++i) nrm[i] = 0.f;
  float tot = 0.f;

  const int f4_per_row = MCK / 4;  // 8 float4 per pixel row of the tile
  for (int c0 = 0; c0 < C; c0 += MCK) {
    const int ck = min(MCK, C - c0);
    // cooperative, coalesced tile load -> transposed LDS
    for (int idx = tid; idx < TPB * f4_per_row; idx += TPB) {
      const int pp = idx / f4_per_row;
      const int cq = (idx - pp * f4_per_row) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (p0 + pp < hw && cq < ck) v = ld4<TX>(xb + (size_t)(p0 + pp) * ldx + c0 + cq);
      xs[(cq + 0) * XLD + pp] = v[0];
      xs[(cq + 1) * XLD + pp] = v[1];
      xs[(cq + 2) * XLD + pp] = v[2];
      xs[(cq + 3) * XLD + pp] = v[3];
    }
    __syncthreads();
    for (int cc = 0; cc < ck; cc += 4) {
      float xv[4], x2[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xv[j] = xs[(cc + j) * XLD + tid];
        x2[j] = xv[j] * xv[j];
        tot += x2[j];
      }
#pragma unroll
      for (int i = 0; i < NPAD; ++i) {
        const int k = c0 + cc + mo.off[i];
        float gv[4];
        if (VEC == 4) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(gg + k);
          gv[0] = t[0]; gv[1] = t[1]; gv[2] = t[2]; gv[3] = t[3];
        } else if (VEC == 2) {
          const f32x2 t0 = *reinterpret_cast<const f32x2*>(gg + k);
          const f32x2 t1 = *reinterpret_cast<const f32x2*>(gg + k + 2);
          gv[0] = t0[0]; gv[1] = t0[1]; gv[2] = t1[0]; gv[3] = t1[1];
        } else {      // odd table offsets (centred windows of CVM_OxfordRobotCar, models.py:1094)
          gv[0] = gg[k]; gv[1] = gg[k + 1]; gv[2] = gg[k + 2]; gv[3] = gg[k + 3];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i] = fmaf(xv[j], gv[j], acc[i]);
        if (PARTIAL) {
          float wv[4];
          if (VEC == 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(ww + k);
            wv[0] = t[0]; wv[1] = t[1]; wv[2] = t[2]; wv[3] = t[3];
          } else if (VEC == 2) {
            const f32x2 t0 = *reinterpret_cast<const f32x2*>(ww + k);
            const f32x2 t1 = *reinterpret_cast<const f32x2*>(ww + k + 2);
            wv[0] = t0[0]; wv[1] = t0[1]; wv[2] = t1[0]; wv[3] = t1[1];
          } else {
            wv[0] = ww[k]; wv[1] = ww[k + 1]; wv[2] = ww[k + 2]; wv[3] = ww[k + 3];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) nrm[PARTIAL ? i : 0] = fmaf(x2[j], wv[j], nrm[PARTIAL ? i : 0]);
        }
      }
    }
    __syncthreads();
  }

  // scores, max (NaN-propagating like torch.max), extras
  float mx = 0.f;
  if (pvalid) {
    TX* drow = dstx + ((size_t)b * hw + p) * ldo;
    const int tail0 = n_shifts - n_tail;
#pragma unroll
    for (int i = 0; i < NPAD; ++i) {
      if (i < n_shifts) {
        const float wn = sqrtf(PARTIAL ? nrm[PARTIAL ? i : 0] : tot);
        const float s = acc[i] / (wn * gnorm);
        scores[((size_t)b * n_shifts + i) * hw + p] = s;
        if (i == 0) mx = s;
        else if (i < n_max && (s > mx || s != s)) mx = s;
        if (i >= tail0) drow[C + 1 + (i - tail0)] = (TX)s;
      }
    }
    drow[C] = (TX)mx;
    for (int c = C + 1 + n_tail; c < ldo; ++c) drow[c] = (TX)0.f;
  }
  inv_s[tid] = 1.0f / fmaxf(sqrtf(tot), 1e-12f);
  __syncthreads();

  // second (coalesced) pass: X * inv_norm -> dstx[:, 0:C]; the tile was just read, so it is L2-hot
  const int c4n = C >> 2;
  const int npx = min(TPB, hw - p0);
  for (int idx = tid; idx < npx * c4n; idx += TPB) {
    const int pp = idx / c4n;
    const int c4 = (idx - pp * c4n) * 4;
    f32x4 v = ld4<TX>(xb + (size_t)(p0 + pp) * ldx + c4);
    v *= inv_s[pp];
    st4<TX>(dstx + ((size_t)b * hw + p0 + pp) * ldo + c4, v);
  }
}

}  // namespace ccvpe

using namespace ccvpe;

template <typename TX, int NPAD, bool PARTIAL, int VEC>
static int launch_match(const TX* x, int ldx, const float* g, int ldg, int L, const MatchOffsets& mo, int n_shifts,
                        int n_max, int n_tail, float* scores, TX* dstx, int ldo, int B, int hw, int C,
                        hipStream_t st) {
  const int tpb = hw >= 256 ? 256 : ((hw + 63) / 64) * 64;
  const size_t smem = sizeof(float) * ((size_t)2 * C * (PARTIAL ? 2 : 1) + (size_t)MCK * (tpb + 1) + tpb + 4);
  if (smem > 160 * 1024) return fail(CCVPE_EINVAL, "match_level: C=%d needs %zu B of LDS", C, smem);
  auto kern = match_kernel<TX, NPAD, PARTIAL, VEC>;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return fail(CCVPE_ELAUNCH, "match_level: set smem attr: %s", hipGetErrorString(e));
  }
  dim3 grid((hw + tpb - 1) / tpb, B);
  hipLaunchKernelGGL(kern, grid, dim3(tpb), smem, st, x, ldx, g, ldg, L, mo, n_shifts, n_max, n_tail, scores, dstx,
                     ldo, hw, C);
  return check_launch("match_kernel");
}

template <typename TX>
static int match_any(const TX* x, int ldx, const float* g, int ldg, int L, const int* shifts, int n_shifts, int n_max,
                     int n_tail, int stride, int window_offset, float* scores, TX* dstx, int ldo, int B, int hw, int C,
                     void* stream) {
  if (n_shifts < 1 || n_shifts > CCVPE_MAX_SHIFTS) return fail(CCVPE_EINVAL, "match_level: n_shifts %d out of range", n_shifts);
  if (n_max < 1 || n_max > n_shifts || n_tail < 0 || n_tail > n_shifts) return fail(CCVPE_EINVAL, "match_level: bad n_max/n_tail");
  if (C % 8 || ldx % 4 || ldo % 4 || ldo < C + 1 + n_tail) return fail(CCVPE_EINVAL, "match_level: C%%8, ldx%%4, ldo%%4, ldo>=C+1+n_tail required");
  if (L < 1 || L > C || L > ldg) return fail(CCVPE_EINVAL, "match_level: bad L");
  if (!aligned16(x) || !aligned16(dstx)) return fail(CCVPE_EINVAL, "match_level: x/dstx must be 16-byte aligned");
  MatchOffsets mo;
  for (int i = 0; i < CCVPE_MAX_SHIFTS; ++i) {
    long o = 0;
    if (i < n_shifts) {
      o = (-((long)shifts[i] * stride + window_offset)) % C;
      if (o < 0) o += C;
    }
    mo.off[i] = (int)o;
  }
  const bool partial = L < C;
  int vec = 4;                                   // widest aligned read of the doubled descriptor table
  for (int i = 0; i < n_shifts; ++i) {
    if (mo.off[i] % 4 && vec > 2) vec = 2;
    if (mo.off[i] % 2) vec = 1;
  }
  hipStream_t st = (hipStream_t)stream;
#define M_ARGS x, ldx, g, ldg, L, mo, n_shifts, n_max, n_tail, scores, dstx, ldo, B, hw, C, st
#define M_DISPATCH(NP)                                              \
  if (n_shifts <= NP) {                                             \
    if (partial) {                                                  \
      if (vec == 4) return launch_match<TX, NP, true, 4>(M_ARGS);   \
      if (vec == 2) return launch_match<TX, NP, true, 2>(M_ARGS);   \
      return launch_match<TX, NP, true, 1>(M_ARGS);                 \
    }                                                               \
    if (vec == 4) return launch_match<TX, NP, false, 4>(M_ARGS);    \
    if (vec == 2) return launch_match<TX, NP, false, 2>(M_ARGS);    \
    return launch_match<TX, NP, false, 1>(M_ARGS);                  \
  }
  M_DISPATCH(1) M_DISPATCH(8) M_DISPATCH(16) M_DISPATCH(24) M_DISPATCH(48)
#undef M_DISPATCH
#undef M_ARGS
  return fail(CCVPE_EINVAL, "match_level: unreachable");
}

extern "C" int ccvpe_match_level_f32(const float* x, int ldx, const float* g, int ldg, int L, const int* shifts,
                                     int n_shifts, int n_max, int n_tail, int stride, int window_offset, float* scores,
                                     float* dstx, int ldo, int B, int hw, int C, void* stream) {
  return match_any<float>(x, ldx, g, ldg, L, shifts, n_shifts, n_max, n_tail, stride, window_offset, scores, dstx, ldo, B,
                          hw, C, stream);
}
extern "C" int ccvpe_match_level_bf16(const void* x, int ldx, const float* g, int ldg, int L, const int* shifts,
                                      int n_shifts, int n_max, int n_tail, int stride, int window_offset, float* scores,
                                      void* dstx, int ldo, int B, int hw, int C, void* stream) {
  return match_any<cc_bf16>(reinterpret_cast<const cc_bf16*>(x), ldx, g, ldg, L, shifts, n_shifts, n_max, n_tail,
                            stride, window_offset, scores, reinterpret_cast<cc_bf16*>(dstx), ldo, B, hw, C, stream);
}
