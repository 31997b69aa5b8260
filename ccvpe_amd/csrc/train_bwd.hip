// Backward of the train-mode BatchNorm + activation (+ drop_connect scale, + SE gating) and small
// elementwise backward helpers.  Forward (csrc/train_ops.hip):
//     xhat = (x - mean) * istd ;  z = xhat*gamma + beta ;  y = act(z) * dcs[b] (+ residual)
// and, when the BN output u = act(z) feeds squeeze-excite:  v = u * gate[b,c] (model.py:113-118).
// Given the gradient dv w.r.t. the consumer's input (v, or y when there is no SE):
//     du   = dv * gate[b,c] + dmean[b,c] / HW        (SE: gate product + the avg-pool branch)
//     g    = du * dcs[b] * act'(z)
//     dbeta = sum g ; dgamma = sum g*xhat ; dx = gamma*istd*(g - dbeta/M - xhat*dgamma/M)
// Two kernels: a per-channel reduction (partials per workgroup, merged in fixed order: deterministic) and
// an elementwise apply that recomputes g.  The residual branch's gradient is dv itself (no kernel).
// swish'(z) = s*(1 + z*(1-s)), s = sigmoid(z)   (efficientnet_pytorch/utils.py:71-75).
#include "common.h"
#include <type_traits>

namespace ccvpe {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct BnBwdParams {
  const float* x;       // raw conv output [B, rows_per_sample, C]
  const float* dv;      // incoming gradient, same shape
  const float* mean;
  const float* var;
  const float* gamma;
  const float* beta;
  const float* gate;    // [B,C] or null
  const float* dmean;   // [B,C] or null (already divided by HW)
  const float* dcs;     // [B] or null
  float eps;
  int act;
  int rows_per_sample, C, rows_per_block, nblk;   // nblk = workgroups per sample
};

__device__ __forceinline__ float act_grad(int act, float z) {
  if (act == CCVPE_ACT_SWISH) {
    const float s = sigmoidf(z);
    return s * (1.0f + z * (1.0f - s));
  }
  if (act == CCVPE_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  return 1.f;
}

// g for 4 channels of one row
__device__ __forceinline__ void bn_bwd_g(const BnBwdParams& p, int b, size_t off, int c, const f32x4& mu, const f32x4& istd,
                                         const f32x4& ga, const f32x4& be, float dcs, f32x4& g, f32x4& xh) {
  const f32x4 xv = *reinterpret_cast<const f32x4*>(p.x + off);
  f32x4 du = *reinterpret_cast<const f32x4*>(p.dv + off);
  if (p.gate) du *= *reinterpret_cast<const f32x4*>(p.gate + (size_t)b * p.C + c);
  if (p.dmean) du += *reinterpret_cast<const f32x4*>(p.dmean + (size_t)b * p.C + c);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    xh[j] = (xv[j] - mu[j]) * istd[j];
    const float z = xh[j] * ga[j] + be[j];
    g[j] = du[j] * dcs * act_grad(p.act, z);
  }
}

// The same arithmetic for the two streaming kernels below, shaped for memory-level parallelism: the activation is a
// compile-time constant (a runtime `act` inside the element loop is a scalar compare + branch per value), the per-(sample,
// channel) gate / mean-branch vectors are loaded once per channel group instead of once per row, and the row loop handles
// two rows per trip with their four loads issued together (one row per trip = one load in flight per thread).
template <int ACT>
__device__ __forceinline__ float act_grad_t(float z) {
  if (ACT == CCVPE_ACT_SWISH) {
    const float s = sigmoidf(z);
    return s * (1.0f + z * (1.0f - s));
  }
  if (ACT == CCVPE_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  return 1.f;
}
template <int ACT>
__device__ __forceinline__ void bn_bwd_g_t(f32x4 xv, f32x4 du, const f32x4& gatev, const f32x4& dmeanv, const f32x4& mu,
                                           const f32x4& istd, const f32x4& ga, const f32x4& be, float dcs, f32x4& g, f32x4& xh) {
  du = du * gatev + dmeanv;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    xh[j] = (xv[j] - mu[j]) * istd[j];
    const float z = xh[j] * ga[j] + be[j];
    g[j] = du[j] * dcs * act_grad_t<ACT>(z);
  }
}
#define CCVPE_BN_ACT_DISPATCH(act, body)                                              \
  do {                                                                                \
    if ((act) == CCVPE_ACT_SWISH) body(std::integral_constant<int, CCVPE_ACT_SWISH>{});       \
    else if ((act) == CCVPE_ACT_RELU) body(std::integral_constant<int, CCVPE_ACT_RELU>{});    \
    else body(std::integral_constant<int, CCVPE_ACT_NONE>{});                         \
  } while (0)

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const BnBwdParams p, float* __restrict__ part /*[B*nblk][2][C]*/) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [P][cgx][2] float4
  const int b = blockIdx.y;
  const int cg4 = p.C >> 2;
  const int cgx = cg4 < 256 ? cg4 : 256;
  const int P = 256 / cgx;
  const int tid = threadIdx.x;
  const int cgl = tid % cgx, pl = tid / cgx;
  const int r0 = blockIdx.x * p.rows_per_block;
  const int r1 = min(r0 + p.rows_per_block, p.rows_per_sample);
  const float dcs = p.dcs ? p.dcs[b] : 1.0f;
  float* out = part + ((size_t)b * p.nblk + blockIdx.x) * 2 * p.C;
  for (int cc = 0; cc < cg4; cc += cgx) {
    const int cg = cc + cgl;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    if (pl < P && cg < cg4) {
      const int c = cg * 4;
      const f32x4 mu = *reinterpret_cast<const f32x4*>(p.mean + c);
      const f32x4 vv = *reinterpret_cast<const f32x4*>(p.var + c);
      const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
      const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + c);
      f32x4 istd;
#pragma unroll
      for (int j = 0; j < 4; ++j) istd[j] = 1.0f / sqrtf(vv[j] + p.eps);
      const f32x4 one4 = {1.f, 1.f, 1.f, 1.f}, zero4 = {0.f, 0.f, 0.f, 0.f};
      const f32x4 gatev = p.gate ? *reinterpret_cast<const f32x4*>(p.gate + (size_t)b * p.C + c) : one4;
      const f32x4 dmeanv = p.dmean ? *reinterpret_cast<const f32x4*>(p.dmean + (size_t)b * p.C + c) : zero4;
      const float* __restrict__ xb = p.x + ((size_t)b * p.rows_per_sample) * p.C + c;
      const float* __restrict__ db_ = p.dv + ((size_t)b * p.rows_per_sample) * p.C + c;
      auto rows = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
        int r = r0 + pl;
        for (; r + P < r1; r += 2 * P) {
          const f32x4 xa = *reinterpret_cast<const f32x4*>(xb + (size_t)r * p.C);
          const f32x4 da = *reinterpret_cast<const f32x4*>(db_ + (size_t)r * p.C);
          const f32x4 xc = *reinterpret_cast<const f32x4*>(xb + (size_t)(r + P) * p.C);
          const f32x4 dc = *reinterpret_cast<const f32x4*>(db_ + (size_t)(r + P) * p.C);
          f32x4 g, xh, g2, xh2;
          bn_bwd_g_t<ACT>(xa, da, gatev, dmeanv, mu, istd, ga, be, dcs, g, xh);
          bn_bwd_g_t<ACT>(xc, dc, gatev, dmeanv, mu, istd, ga, be, dcs, g2, xh2);
          s0 += g;
          s1 += g * xh;
          s0 += g2;
          s1 += g2 * xh2;
        }
        if (r < r1) {
          f32x4 g, xh;
          bn_bwd_g_t<ACT>(*reinterpret_cast<const f32x4*>(xb + (size_t)r * p.C), *reinterpret_cast<const f32x4*>(db_ + (size_t)r * p.C),
                          gatev, dmeanv, mu, istd, ga, be, dcs, g, xh);
          s0 += g;
          s1 += g * xh;
        }
      };
      CCVPE_BN_ACT_DISPATCH(p.act, rows);
    }
    f32x4* red4 = reinterpret_cast<f32x4*>(red);
    if (pl < P) {
      red4[(pl * cgx + cgl) * 2] = s0;
      red4[(pl * cgx + cgl) * 2 + 1] = s1;
    }
    __syncthreads();
    if (pl == 0 && cg < cg4) {
      f32x4 t0 = red4[cgl * 2], t1 = red4[cgl * 2 + 1];
      for (int q = 1; q < P; ++q) {
        t0 += red4[(q * cgx + cgl) * 2];
        t1 += red4[(q * cgx + cgl) * 2 + 1];
      }
      *reinterpret_cast<f32x4*>(out + cg * 4) = t0;
      *reinterpret_cast<f32x4*>(out + p.C + cg * 4) = t1;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnBwdParams p, const float* __restrict__ dbeta,
                                                           const float* __restrict__ dgamma, float inv_m,
                                                           float* __restrict__ dx) {
  const int b = blockIdx.y;
  const int cg4 = p.C >> 2;
  const int cgx = cg4 < 256 ? cg4 : 256;
  const int P = 256 / cgx;
  const int tid = threadIdx.x;
  const int cgl = tid % cgx, pl = tid / cgx;
  const int r0 = blockIdx.x * p.rows_per_block;
  const int r1 = min(r0 + p.rows_per_block, p.rows_per_sample);
  const float dcs = p.dcs ? p.dcs[b] : 1.0f;
  for (int cc = 0; cc < cg4; cc += cgx) {
    const int cg = cc + cgl;
    if (pl < P && cg < cg4) {
      const int c = cg * 4;
      const f32x4 mu = *reinterpret_cast<const f32x4*>(p.mean + c);
      const f32x4 vv = *reinterpret_cast<const f32x4*>(p.var + c);
      const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
      const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + c);
      const f32x4 db = *reinterpret_cast<const f32x4*>(dbeta + c) * inv_m;
      const f32x4 dg = *reinterpret_cast<const f32x4*>(dgamma + c) * inv_m;
      f32x4 istd;
#pragma unroll
      for (int j = 0; j < 4; ++j) istd[j] = 1.0f / sqrtf(vv[j] + p.eps);
      const f32x4 one4 = {1.f, 1.f, 1.f, 1.f}, zero4 = {0.f, 0.f, 0.f, 0.f};
      const f32x4 gatev = p.gate ? *reinterpret_cast<const f32x4*>(p.gate + (size_t)b * p.C + c) : one4;
      const f32x4 dmeanv = p.dmean ? *reinterpret_cast<const f32x4*>(p.dmean + (size_t)b * p.C + c) : zero4;
      const size_t base = ((size_t)b * p.rows_per_sample) * p.C + c;
      const float* __restrict__ xb = p.x + base;
      const float* __restrict__ dvb = p.dv + base;
      float* __restrict__ dxb = dx + base;
      const f32x4 k = ga * istd;
      auto rows = [&](auto act_tag) {
        constexpr int ACT = decltype(act_tag)::value;
        int r = r0 + pl;
        for (; r + P < r1; r += 2 * P) {
          const f32x4 xa = *reinterpret_cast<const f32x4*>(xb + (size_t)r * p.C);
          const f32x4 da = *reinterpret_cast<const f32x4*>(dvb + (size_t)r * p.C);
          const f32x4 xc = *reinterpret_cast<const f32x4*>(xb + (size_t)(r + P) * p.C);
          const f32x4 dc = *reinterpret_cast<const f32x4*>(dvb + (size_t)(r + P) * p.C);
          f32x4 g, xh, g2, xh2;
          bn_bwd_g_t<ACT>(xa, da, gatev, dmeanv, mu, istd, ga, be, dcs, g, xh);
          bn_bwd_g_t<ACT>(xc, dc, gatev, dmeanv, mu, istd, ga, be, dcs, g2, xh2);
          *reinterpret_cast<f32x4*>(dxb + (size_t)r * p.C) = k * (g - db - xh * dg);
          *reinterpret_cast<f32x4*>(dxb + (size_t)(r + P) * p.C) = k * (g2 - db - xh2 * dg);
        }
        if (r < r1) {
          f32x4 g, xh;
          bn_bwd_g_t<ACT>(*reinterpret_cast<const f32x4*>(xb + (size_t)r * p.C), *reinterpret_cast<const f32x4*>(dvb + (size_t)r * p.C),
                          gatev, dmeanv, mu, istd, ga, be, dcs, g, xh);
          *reinterpret_cast<f32x4*>(dxb + (size_t)r * p.C) = k * (g - db - xh * dg);
        }
      };
      CCVPE_BN_ACT_DISPATCH(p.act, rows);
    }
  }
}

// dgate[b,c] = sum_px dv[b,px,c] * u[b,px,c],  u = act(bn(x))  (the SE product's gradient w.r.t. the gate)
__global__ __launch_bounds__(256) void se_dgate_kernel(const BnBwdParams p, float* __restrict__ part /*[B][nblk][C]*/) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int b = blockIdx.y;
  const int cg4 = p.C >> 2;
  const int cgx = cg4 < 256 ? cg4 : 256;
  const int P = 256 / cgx;
  const int tid = threadIdx.x;
  const int cgl = tid % cgx, pl = tid / cgx;
  const int r0 = blockIdx.x * p.rows_per_block;
  const int r1 = min(r0 + p.rows_per_block, p.rows_per_sample);
  for (int cc = 0; cc < cg4; cc += cgx) {
    const int cg = cc + cgl;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f};
    if (pl < P && cg < cg4) {
      const int c = cg * 4;
      const f32x4 mu = *reinterpret_cast<const f32x4*>(p.mean + c);
      const f32x4 vv = *reinterpret_cast<const f32x4*>(p.var + c);
      const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
      const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + c);
      for (int r = r0 + pl; r < r1; r += P) {
        const size_t off = ((size_t)b * p.rows_per_sample + r) * p.C + c;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(p.x + off);
        const f32x4 dv = *reinterpret_cast<const f32x4*>(p.dv + off);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float z = (xv[j] - mu[j]) / sqrtf(vv[j] + p.eps) * ga[j] + be[j];
          const float u = p.act == CCVPE_ACT_SWISH ? swishf(z) : (p.act == CCVPE_ACT_RELU ? fmaxf(z, 0.f) : z);
          s0[j] = fmaf(dv[j], u, s0[j]);
        }
      }
    }
    f32x4* red4 = reinterpret_cast<f32x4*>(red);
    if (pl < P) red4[pl * cgx + cgl] = s0;
    __syncthreads();
    if (pl == 0 && cg < cg4) {
      f32x4 t = red4[cgl];
      for (int q = 1; q < P; ++q) t += red4[q * cgx + cgl];
      *reinterpret_cast<f32x4*>(part + ((size_t)b * p.nblk + blockIdx.x) * p.C + cg * 4) = t;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// BatchNorm-with-squeeze-excite backward in TWO passes over (x, dv) instead of three.
// The SE gradient needs dgate[b,c] = sum_px dv*u BEFORE the BatchNorm reduction can run, because the reduction's g =
// (dv*gate + dmean) * act'(z) contains dmean, which comes out of the SE backward.  But g is LINEAR in (gate, dmean), both
// constant over the pixels of a (sample, channel):
//     sum_px g        = gate * sum dv*a'      + dmean * sum a'
//     sum_px g * xhat = gate * sum dv*a'*xhat + dmean * sum a'*xhat            (a' = act'(z))
// so ONE pass accumulates the five per-(sample, channel) sums  A = (dv*u, dv*a', a', dv*a'*xhat, a'*xhat),  the SE backward
// runs on A[0], and dbeta / dgamma are finished from A[1..4] with a [B,C]-sized kernel.  (se_dgate_kernel + bn_bwd_reduce
// were two full passes over the depthwise-conv-sized tensors of all 32 MBConv blocks: 2.3 + 6.8 ms of the B = 64 step.)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void se_bn_bwd_reduce_kernel(const BnBwdParams p, float* __restrict__ part /*[B*nblk][5][C]*/) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [P][cgx][5] float4
  const int b = blockIdx.y;
  const int cg4 = p.C >> 2;
  const int cgx = cg4 < 256 ? cg4 : 256;
  const int P = 256 / cgx;
  const int tid = threadIdx.x;
  const int cgl = tid % cgx, pl = tid / cgx;
  const int r0 = blockIdx.x * p.rows_per_block;
  const int r1 = min(r0 + p.rows_per_block, p.rows_per_sample);
  float* out = part + ((size_t)b * p.nblk + blockIdx.x) * 5 * p.C;
  for (int cc = 0; cc < cg4; cc += cgx) {
    const int cg = cc + cgl;
    f32x4 s[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) s[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (pl < P && cg < cg4) {
      const int c = cg * 4;
      const f32x4 mu = *reinterpret_cast<const f32x4*>(p.mean + c);
      const f32x4 vv = *reinterpret_cast<const f32x4*>(p.var + c);
      const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
      const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + c);
      f32x4 istd;
#pragma unroll
      for (int j = 0; j < 4; ++j) istd[j] = 1.0f / sqrtf(vv[j] + p.eps);
      const float* __restrict__ xb = p.x + ((size_t)b * p.rows_per_sample) * p.C + c;
      const float* __restrict__ db_ = p.dv + ((size_t)b * p.rows_per_sample) * p.C + c;
      auto one = [&](auto act_tag, f32x4 xv, f32x4 dv) {
        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = (xv[j] - mu[j]) * istd[j];
          const float z = xh * ga[j] + be[j];
          float u, ap;
          if (ACT == CCVPE_ACT_SWISH) {
            const float sg = sigmoidf(z);
            u = z * sg;
            ap = sg * (1.0f + z * (1.0f - sg));
          } else if (ACT == CCVPE_ACT_RELU) {
            u = fmaxf(z, 0.f);
            ap = z > 0.f ? 1.f : 0.f;
          } else {
            u = z;
            ap = 1.f;
          }
          const float da = dv[j] * ap;
          s[0][j] = fmaf(dv[j], u, s[0][j]);
          s[1][j] += da;
          s[2][j] += ap;
          s[3][j] = fmaf(da, xh, s[3][j]);
          s[4][j] = fmaf(ap, xh, s[4][j]);
        }
      };
      auto rows = [&](auto act_tag) {
        int r = r0 + pl;
        for (; r + P < r1; r += 2 * P) {             // two rows per trip: four loads in flight
          const f32x4 xa = *reinterpret_cast<const f32x4*>(xb + (size_t)r * p.C);
          const f32x4 da = *reinterpret_cast<const f32x4*>(db_ + (size_t)r * p.C);
          const f32x4 xc = *reinterpret_cast<const f32x4*>(xb + (size_t)(r + P) * p.C);
          const f32x4 dc = *reinterpret_cast<const f32x4*>(db_ + (size_t)(r + P) * p.C);
          one(act_tag, xa, da);
          one(act_tag, xc, dc);
        }
        if (r < r1) one(act_tag, *reinterpret_cast<const f32x4*>(xb + (size_t)r * p.C), *reinterpret_cast<const f32x4*>(db_ + (size_t)r * p.C));
      };
      CCVPE_BN_ACT_DISPATCH(p.act, rows);
    }
    f32x4* red4 = reinterpret_cast<f32x4*>(red);
    if (pl < P) {
#pragma unroll
      for (int q = 0; q < 5; ++q) red4[(pl * cgx + cgl) * 5 + q] = s[q];
    }
    __syncthreads();
    if (pl == 0 && cg < cg4) {
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        f32x4 t = red4[cgl * 5 + q];
        for (int w = 1; w < P; ++w) t += red4[(w * cgx + cgl) * 5 + q];
        *reinterpret_cast<f32x4*>(out + (size_t)q * p.C + cg * 4) = t;
      }
    }
    __syncthreads();
  }
}

// A[q][b][c] = sum over the sample's nblk partial rows (fixed order)
__global__ __launch_bounds__(256) void se_bn_merge_kernel(const float* __restrict__ part, int nblk, int C, int B,
                                                          float* __restrict__ A) {
  const int e = blockIdx.x * 256 + threadIdx.x;      // q * C + c
  const int b = blockIdx.y;
  if (e >= 5 * C) return;
  const float* src = part + (size_t)b * nblk * 5 * C + e;
  float s0 = 0.f, s1 = 0.f;
  int k = 0;
  for (; k + 1 < nblk; k += 2) {
    s0 += src[(size_t)k * 5 * C];
    s1 += src[(size_t)(k + 1) * 5 * C];
  }
  if (k < nblk) s0 += src[(size_t)k * 5 * C];
  const int q = e / C, c = e - q * C;
  A[((size_t)q * B + b) * C + c] = s0 + s1;
}

// dbeta[c] = sum_b gate*A1 + dmean*A2 ; dgamma[c] = sum_b gate*A3 + dmean*A4.  Workgroup = 64 channels x 4 batch lanes (lane q
// adds samples q, q+4, ...), combined through LDS in lane order: fixed summation order, 16 dependent steps instead of 64.
__global__ __launch_bounds__(256) void se_bn_finish_kernel(const float* __restrict__ A, const float* __restrict__ gate,
                                                           const float* __restrict__ dmean, int B, int C,
                                                           float* __restrict__ dbeta, float* __restrict__ dgamma) {
  __shared__ float red[2][4][64];
  const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const size_t plane = (size_t)B * C;
  float sb = 0.f, sg = 0.f;
  if (c < C) {
#pragma unroll 4
    for (int b = q; b < B; b += 4) {
      const size_t i = (size_t)b * C + c;
      const float g = gate[i], dm = dmean[i];
      sb += g * A[plane + i] + dm * A[2 * plane + i];
      sg += g * A[3 * plane + i] + dm * A[4 * plane + i];
    }
  }
  red[0][q][cl] = sb;
  red[1][q][cl] = sg;
  __syncthreads();
  if (q == 0 && c < C) {
    dbeta[c] = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
    dgamma[c] = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
  }
}

// dx = dy where y > 0 else 0   (ReLU between the two convs of double_conv, models.py:45)
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                       float* __restrict__ dx, long n4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4 yv = reinterpret_cast<const f32x4*>(y)[i];
  f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
#pragma unroll
  for (int j = 0; j < 4; ++j) g[j] = yv[j] > 0.f ? g[j] : 0.f;
  reinterpret_cast<f32x4*>(dx)[i] = g;
}

}  // namespace ccvpe

using namespace ccvpe;

static int bn_rows_per_block(int rows_per_sample) {
  int rpb = rows_per_sample / 32;
  return rpb < 8 ? 8 : rpb;
}

static int fill_bn_params(BnBwdParams& p, const float* x, const float* dv, const float* mean, const float* var,
                          const float* gamma, const float* beta, const float* gate, const float* dmean,
                          const float* dcs, float eps, int act, int batch, int rows_per_sample, int channels) {
  if (batch <= 0 || rows_per_sample <= 0 || channels <= 0 || channels % 4) return fail(CCVPE_EINVAL, "bn_bwd: bad shape");
  if (!aligned16(x) || !aligned16(dv) || !aligned16(mean) || !aligned16(var) || !aligned16(gamma) || !aligned16(beta) ||
      (gate && !aligned16(gate)) || (dmean && !aligned16(dmean)))
    return fail(CCVPE_EINVAL, "bn_bwd: pointers must be 16-byte aligned");
  p.x = x; p.dv = dv; p.mean = mean; p.var = var; p.gamma = gamma; p.beta = beta;
  p.gate = gate; p.dmean = dmean; p.dcs = dcs; p.eps = eps; p.act = act;
  p.rows_per_sample = rows_per_sample; p.C = channels;
  p.rows_per_block = bn_rows_per_block(rows_per_sample);
  p.nblk = (rows_per_sample + p.rows_per_block - 1) / p.rows_per_block;
  return CCVPE_OK;
}

extern "C" int ccvpe_bn_bwd_nblk(int rows_per_sample) {
  const int rpb = bn_rows_per_block(rows_per_sample);
  return (rows_per_sample + rpb - 1) / rpb;
}

extern "C" int ccvpe_bn_act_bwd_f32(const float* x, const float* dv, const float* mean, const float* var,
                                    const float* gamma, const float* beta, const float* gate, const float* dmean,
                                    const float* dc_scale, float eps, int act, float* dx, float* dgamma, float* dbeta,
                                    float* scratch, int batch, int rows_per_sample, int channels, void* stream) {
  BnBwdParams p;
  const int rc = fill_bn_params(p, x, dv, mean, var, gamma, beta, gate, dmean, dc_scale, eps, act, batch, rows_per_sample,
                                channels);
  if (rc) return rc;
  const int cg4 = channels / 4, cgx = cg4 < 256 ? cg4 : 256, P = 256 / cgx;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(p.nblk, batch);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, grid, dim3(256), (size_t)P * cgx * 2 * 16, st, p, scratch);
  if (dgamma == dbeta + channels) {   // adjacent outputs (what the host side allocates): the partial rows are [dbeta | dgamma] too
    launch_sum_parts(scratch, p.nblk * batch, 2L * channels, 2 * channels, dbeta, st);
  } else {
    launch_sum_parts(scratch, p.nblk * batch, 2L * channels, channels, dbeta, st);
    launch_sum_parts(scratch + channels, p.nblk * batch, 2L * channels, channels, dgamma, st);
  }
  hipLaunchKernelGGL(bn_bwd_apply_kernel, grid, dim3(256), 0, st, p, dbeta, dgamma,
                     1.0f / ((float)batch * (float)rows_per_sample), dx);
  return check_launch("bn_act_bwd");
}

extern "C" int ccvpe_se_bn_bwd_reduce_f32(const float* x, const float* dv, const float* mean, const float* var,
                                          const float* gamma, const float* beta, float eps, int act, float* A,
                                          float* scratch, int batch, int rows_per_sample, int channels, void* stream) {
  BnBwdParams p;
  const int rc = fill_bn_params(p, x, dv, mean, var, gamma, beta, nullptr, nullptr, nullptr, eps, act, batch, rows_per_sample,
                                channels);
  if (rc) return rc;
  if (!A || !scratch) return fail(CCVPE_EINVAL, "se_bn_bwd_reduce: null output");
  const int cg4 = channels / 4, cgx = cg4 < 256 ? cg4 : 256, P = 256 / cgx;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(se_bn_bwd_reduce_kernel, dim3(p.nblk, batch), dim3(256), (size_t)P * cgx * 5 * 16, st, p, scratch);
  hipLaunchKernelGGL(se_bn_merge_kernel, dim3((5 * channels + 255) / 256, batch), dim3(256), 0, st, scratch, p.nblk, channels,
                     batch, A);
  return check_launch("se_bn_bwd_reduce");
}

extern "C" int ccvpe_se_bn_bwd_apply_f32(const float* x, const float* dv, const float* mean, const float* var,
                                         const float* gamma, const float* beta, const float* gate, const float* dmean,
                                         float eps, int act, const float* A, float* dx, float* dgamma, float* dbeta,
                                         int batch, int rows_per_sample, int channels, void* stream) {
  BnBwdParams p;
  const int rc = fill_bn_params(p, x, dv, mean, var, gamma, beta, gate, dmean, nullptr, eps, act, batch, rows_per_sample,
                                channels);
  if (rc) return rc;
  if (!gate || !dmean || !A) return fail(CCVPE_EINVAL, "se_bn_bwd_apply: gate, dmean and A are required");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(se_bn_finish_kernel, dim3((channels + 63) / 64), dim3(256), 0, st, A, gate, dmean, batch, channels, dbeta,
                     dgamma);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(p.nblk, batch), dim3(256), 0, st, p, dbeta, dgamma,
                     1.0f / ((float)batch * (float)rows_per_sample), dx);
  return check_launch("se_bn_bwd_apply");
}

extern "C" int ccvpe_se_dgate_f32(const float* x, const float* dv, const float* mean, const float* var,
                                  const float* gamma, const float* beta, float eps, int act, float* part, int batch,
                                  int rows_per_sample, int channels, void* stream) {
  BnBwdParams p;
  const int rc = fill_bn_params(p, x, dv, mean, var, gamma, beta, nullptr, nullptr, nullptr, eps, act, batch,
                                rows_per_sample, channels);
  if (rc) return rc;
  const int cg4 = channels / 4, cgx = cg4 < 256 ? cg4 : 256, P = 256 / cgx;
  hipLaunchKernelGGL(se_dgate_kernel, dim3(p.nblk, batch), dim3(256), (size_t)P * cgx * 16, (hipStream_t)stream, p, part);
  return check_launch("se_dgate_kernel");
}

extern "C" int ccvpe_relu_bwd_f32(const float* y, const float* dy, float* dx, int n_elems, void* stream) {
  if (n_elems <= 0 || n_elems % 4) return fail(CCVPE_EINVAL, "relu_bwd: n %% 4");
  const long n4 = n_elems / 4;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, dy, dx, n4);
  return check_launch("relu_bwd_kernel");
}

// ---------------------------------------------------------------------------------------------
// Depthwise conv backward (efficientnet_pytorch/model.py:181-182; padding utils.py:265-277 / 341-353).
// Forward: y[b,oy,ox,c] = sum_{ky,kx} xpad[b, oy*S+ky, ox*S+kx, c] * w[ky,kx,c], xpad = x padded by PB before
// (zero rows; zero or circular columns).
//   dgrad: dx[b,iy,ix,c] = sum over the (ky,kx,oy,ox) whose padded coordinate maps onto (iy,ix).
//   wgrad: dw[ky,kx,c]   = sum_{b,oy,ox} dy[b,oy,ox,c] * xpad[b, oy*S+ky, ox*S+kx, c]   (partials + merge).
// ---------------------------------------------------------------------------------------------
namespace ccvpe {

template <int K, int S>
__global__ __launch_bounds__(256) void dw_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                       float* __restrict__ dx, int B, int H, int W, int C, int Ho,
                                                       int Wo, int circular) {
  constexpr int PB = (S == 1) ? (K - 1) / 2 : (K - 2) / 2;
  const int cg4 = C >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)B * H * W * cg4;
  if (idx >= total) return;
  const int cg = (int)(idx % cg4);
  long px = idx / cg4;
  const int ix = (int)(px % W);
  px /= W;
  const int iy = (int)(px % H);
  const int b = (int)(px / H);
  const int c = cg * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float* dyb = dy + (size_t)b * Ho * Wo * C + c;
  const int mlo = circular ? -1 : 0, mhi = circular ? 1 : 0;
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    const int ty = iy + PB - ky;
    if (ty < 0 || (ty % S) != 0) continue;
    const int oy = ty / S;
    if (oy >= Ho) continue;
    for (int m = mlo; m <= mhi; ++m) {
      const int jp = ix + PB + m * W;          // padded column this input column appears at
      if (jp < 0 || jp >= W + ((S == 1) ? (K - 1) : (K - 2))) continue;
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const int tx = jp - kx;
        if (tx < 0 || (tx % S) != 0) continue;
        const int ox = tx / S;
        if (ox >= Wo) continue;
        const f32x4 g = *reinterpret_cast<const f32x4*>(dyb + ((size_t)oy * Wo + ox) * C);
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (size_t)(ky * K + kx) * C + c);
        acc += g * wv;
      }
    }
  }
  *reinterpret_cast<f32x4*>(dx + (((size_t)b * H + iy) * W + ix) * C + c) = acc;
}

// output rows per workgroup: small, so that even the 16x64 late blocks launch >= 1-2 K workgroups (the per-thread
// loop is a chain of dependent gathers: it needs occupancy, not long rows)
static inline int dww_rows(int Ho) { return Ho <= 64 ? 1 : (Ho <= 128 ? 2 : 4); }

template <int K, int S>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                       float* __restrict__ part, int H, int W, int C, int Ho, int Wo,
                                                       int circular, int nblk, int rows) {
  constexpr int PB = (S == 1) ? (K - 1) / 2 : (K - 2) / 2;
  extern __shared__ __attribute__((aligned(16))) float red[];   // [K][P][cgx] float4: one kernel row of taps per round
  const int b = blockIdx.y;
  const int cg4 = C >> 2;
  const int cgx = cg4 < 64 ? cg4 : 64;
  const int P = 256 / cgx;
  const int tid = threadIdx.x;
  const int cgl = tid % cgx, pl = tid / cgx;
  const int oy0 = blockIdx.x * rows;
  float* out = part + ((size_t)b * nblk + blockIdx.x) * K * K * C;
  const float* xb = x + (size_t)b * H * W * C;
  const float* dyb = dy + ((size_t)b * Ho + oy0) * Wo * C;
  f32x4* red4 = reinterpret_cast<f32x4*>(red);
  for (int cc = 0; cc < cg4; cc += cgx) {
    const int cg = cc + cgl;
    const bool live = pl < P && cg < cg4;
    f32x4 acc[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (live) {
      // strip of CW output columns per thread: the K input rows of a strip are loaded once as a (CW-1)*S+K wide window
      // and every tap takes its operand from the registers (about half the loads of a per-tap gather)
      constexpr int CW = 4;
      constexpr int XW = (CW - 1) * S + K;
      const int nrow = min(rows, Ho - oy0);
      const int nstrip = (Wo + CW - 1) / CW;
      for (int o = pl; o < nrow * nstrip; o += P) {
        const int r = o / nstrip, ox0 = (o - r * nstrip) * CW;
        const int oy = oy0 + r;
        f32x4 g[CW];
#pragma unroll
        for (int j = 0; j < CW; ++j) {
          g[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (ox0 + j < Wo) g[j] = *reinterpret_cast<const f32x4*>(dyb + ((size_t)r * Wo + ox0 + j) * C + cg * 4);
        }
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
          const int iy = oy * S - PB + ky;
          if ((unsigned)iy >= (unsigned)H) continue;
          f32x4 xw[XW];
#pragma unroll
          for (int t = 0; t < XW; ++t) {
            int ix = ox0 * S - PB + t;
            if (circular) {
              if (ix < 0) ix += W;
              else if (ix >= W) ix -= W;
            }
            xw[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if ((unsigned)ix < (unsigned)W) xw[t] = *reinterpret_cast<const f32x4*>(xb + ((size_t)iy * W + ix) * C + cg * 4);
          }
#pragma unroll
          for (int kx = 0; kx < K; ++kx)
#pragma unroll
            for (int j = 0; j < CW; ++j) acc[ky * K + kx] += g[j] * xw[j * S + kx];
        }
      }
    }
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      if (pl < P) {
#pragma unroll
        for (int kx = 0; kx < K; ++kx) red4[(kx * P + pl) * cgx + cgl] = acc[ky * K + kx];
      }
      __syncthreads();
      // K * cgx outputs, P partials each: spread over the workgroup
      for (int i = tid; i < K * cgx; i += 256) {
        const int kx = i / cgx, cl = i - kx * cgx;
        if (cc + cl < cg4) {
          f32x4 s = red4[(kx * P) * cgx + cl];
          for (int q = 1; q < P; ++q) s += red4[(kx * P + q) * cgx + cl];
          *reinterpret_cast<f32x4*>(out + (size_t)(ky * K + kx) * C + (cc + cl) * 4) = s;
        }
      }
      __syncthreads();
    }
  }
}

// Depthwise weight gradient  dW[ky][kx][c] = sum_{b,oy,ox} dy[b,oy,ox,c] * x[b, oy*S-PB+ky, ox*S-PB+kx, c]
// (efficientnet_pytorch/model.py:69-75's _depthwise_conv under autograd) for the LATE blocks (planes of <= 1 024 pixels, 480-1 152
// channels; round 6): a WAVE owns one kernel row ky, a lane four channels, so a thread carries K accumulators instead of K*K
// (dw_wgrad_kernel above sits at 216 VGPRs = 2 waves per SIMD for K = 5, and on these planes a workgroup has a handful of strips
// per thread: every strip a full load latency with little else resident; tools/dww_probe.py, B = 64, us: 32 x 32 k3 133 -> 94,
// k5 275 -> 150 (480 ch) / 277 -> 221 (672 ch), 16 x 16 k5 158 -> 83, k3 125 -> 58.  On the large planes of blocks 0-4 it is
// SLOWER than dw_wgrad_kernel — 404 -> 559 us at 144 ch x 128 x 128: its channel chunks cut a pixel's channel run into pieces
// that are not cache-line multiples — so those keep the all-taps kernel).  Per strip of CW
// output columns a thread issues CW dy loads + (CW-1)*S+K x loads together — branch-free buffer loads, out-of-image taps
// steered to an out-of-range offset that returns zeros — and does K*CW multiply-adds.  Workgroup = K x NW waves on `rows`
// output rows of one sample and one chunk of <= 64 channel groups (blockIdx.z: the channel split costs no extra partial
// rows, unlike a finer row split); one LDS merge per workgroup, fixed order (deterministic).
static inline int dww_rows_late(int Ho) { return 4; }

// channel groups per chunk: the split of cg4 into n equal chunks of <= 64 that fills a wave best (lanes = cgx * (64 / cgx))
static inline int dww_chunk(int cg4) {
  int best = cg4 < 64 ? cg4 : 64;
  double bu = 0.0;
  for (int n = 1; n <= 16; ++n) {
    const int cgx = (cg4 + n - 1) / n;
    if (cgx > 64) continue;
    const double u = (double)cg4 / (n * cgx) * (double)(cgx * (64 / cgx)) / 64.0;
    if (u > bu + 1e-9) { bu = u; best = cgx; }
    if (cgx == 1) break;
  }
  return best;
}

template <int K, int S, int NW>
__global__ __launch_bounds__(K * NW * 64) void dw_wgrad_rows_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                float* __restrict__ part, int H, int W, int C, int Ho, int Wo,
                                                                int circular, int nblk, int rows, int cgx) {
  constexpr int PB = (S == 1) ? (K - 1) / 2 : (K - 2) / 2;
  constexpr int CW = 4;
  constexpr int XW = (CW - 1) * S + K;
  extern __shared__ __attribute__((aligned(16))) float red[];   // [K][K][NW * Pw][cgx] float4
  const int b = blockIdx.y;
  const int cg4 = C >> 2;
  const int Pw = 64 / cgx;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int ky = wave / NW, slot = wave - ky * NW;
  const int cgl = lane % cgx, pl = lane / cgx;
  const int cg = blockIdx.z * cgx + cgl;
  const bool live = pl < Pw && cg < cg4;
  const int oy0 = blockIdx.x * rows;
  const int nrow = min(rows, Ho - oy0);
  const int nstrip = (Wo + CW - 1) / CW;
  const unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (size_t)b * H * W * C), 0, H * W * C * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy + (size_t)b * Ho * Wo * C), 0, Ho * Wo * C * 4, 0x00020000);
  f32x4 acc[K];
#pragma unroll
  for (int t = 0; t < K; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (live) {
    for (int o = slot * Pw + pl; o < nrow * nstrip; o += NW * Pw) {
      const int r = o / nstrip, ox0 = (o - r * nstrip) * CW;
      const int oy = oy0 + r;
      const int iy = oy * S - PB + ky;
      const bool rowok = (unsigned)iy < (unsigned)H;
      f32x4 g[CW], xw[XW];
#pragma unroll
      for (int j = 0; j < CW; ++j) {
        const unsigned off = (ox0 + j < Wo && rowok) ? (unsigned)(((oy * Wo + ox0 + j) * C + cg * 4) * 4) : OOB;
        g[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, off, 0, 0));
      }
#pragma unroll
      for (int t = 0; t < XW; ++t) {
        int ix = ox0 * S - PB + t;
        if (circular) ix = ix < 0 ? ix + W : (ix >= W ? ix - W : ix);
        const unsigned off = ((unsigned)ix < (unsigned)W && rowok) ? (unsigned)(((iy * W + ix) * C + cg * 4) * 4) : OOB;
        xw[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
      }
#pragma unroll
      for (int kx = 0; kx < K; ++kx)
#pragma unroll
        for (int j = 0; j < CW; ++j) acc[kx] += g[j] * xw[j * S + kx];
    }
  }
  f32x4* red4 = reinterpret_cast<f32x4*>(red);
  const int Q = NW * Pw;
  if (pl < Pw) {
#pragma unroll
    for (int kx = 0; kx < K; ++kx) red4[((ky * K + kx) * Q + slot * Pw + pl) * cgx + cgl] = acc[kx];
  }
  __syncthreads();
  float* out = part + ((size_t)b * nblk + blockIdx.x) * K * K * C;
  for (int i = tid; i < K * K * cgx; i += K * NW * 64) {
    const int tap = i / cgx, cl = i - tap * cgx;
    const int cgo = blockIdx.z * cgx + cl;
    if (cgo < cg4) {
      f32x4 sacc = red4[(tap * Q) * cgx + cl];
      for (int q = 1; q < Q; ++q) sacc += red4[(tap * Q + q) * cgx + cl];
      *reinterpret_cast<f32x4*>(out + (size_t)tap * C + cgo * 4) = sacc;
    }
  }
}

}  // namespace ccvpe

static void dw_out_dims(int H, int W, int k, int stride, int* Ho, int* Wo) {
  const int tot = stride == 1 ? k - 1 : k - 2;
  *Ho = (H + tot - k) / stride + 1;
  *Wo = (W + tot - k) / stride + 1;
}

extern "C" int ccvpe_dwconv_dgrad_f32(const float* dy, const float* w, float* dx, int batch, int in_h, int in_w,
                                      int channels, int k, int stride, int circular, void* stream) {
  if (batch <= 0 || in_h <= 0 || in_w <= 0 || channels <= 0 || channels % 4) return fail(CCVPE_EINVAL, "dw_dgrad: bad shape");
  if (!((k == 3 || k == 5) && (stride == 1 || stride == 2))) return fail(CCVPE_EINVAL, "dw_dgrad: k in {3,5}, stride in {1,2}");
  int Ho, Wo;
  dw_out_dims(in_h, in_w, k, stride, &Ho, &Wo);
  const long total = (long)batch * in_h * in_w * (channels / 4);
  dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
#define DG(K_, S_) hipLaunchKernelGGL((dw_dgrad_kernel<K_, S_>), grid, dim3(256), 0, st, dy, w, dx, batch, in_h, in_w, channels, Ho, Wo, circular)
  if (k == 3 && stride == 1) DG(3, 1);
  else if (k == 3) DG(3, 2);
  else if (stride == 1) DG(5, 1);
  else DG(5, 2);
#undef DG
  return check_launch("dw_dgrad_kernel");
}

// planes of <= 1 024 pixels (blocks 5-15 at 512 x 512 / 320 x 640 images): the row-per-wave kernel
static inline bool dww_late(int in_h, int in_w) { return (long)in_h * in_w <= 1024; }

extern "C" int ccvpe_dwconv_wgrad_nblk(int in_h, int in_w, int k, int stride) {
  int Ho, Wo;
  dw_out_dims(in_h, in_w, k, stride, &Ho, &Wo);
  const int rows = dww_late(in_h, in_w) ? dww_rows_late(Ho) : dww_rows(Ho);
  return (Ho + rows - 1) / rows;
}

extern "C" int ccvpe_dwconv_wgrad_f32(const float* x, const float* dy, float* dw, float* scratch, int batch, int in_h,
                                      int in_w, int channels, int k, int stride, int circular, void* stream) {
  if (batch <= 0 || in_h <= 0 || in_w <= 0 || channels <= 0 || channels % 4) return fail(CCVPE_EINVAL, "dw_wgrad: bad shape");
  if (!((k == 3 || k == 5) && (stride == 1 || stride == 2))) return fail(CCVPE_EINVAL, "dw_wgrad: k in {3,5}, stride in {1,2}");
  int Ho, Wo;
  dw_out_dims(in_h, in_w, k, stride, &Ho, &Wo);
  const bool late = dww_late(in_h, in_w);
  const int rows = late ? dww_rows_late(Ho) : dww_rows(Ho);
  const int nblk = (Ho + rows - 1) / rows;
  const int cg4 = channels / 4;
  hipStream_t st = (hipStream_t)stream;
  if (late) {
    const int cgx = dww_chunk(cg4), nchunk = (cg4 + cgx - 1) / cgx;
    dim3 grid(nblk, batch, nchunk);
#define WGR(K_, S_, NW_)                                                                                                           \
  hipLaunchKernelGGL((dw_wgrad_rows_kernel<K_, S_, NW_>), grid, dim3(K_ * NW_ * 64), (size_t)K_ * K_ * NW_ * (64 / cgx) * cgx * 16, st, x, \
                     dy, scratch, in_h, in_w, channels, Ho, Wo, circular, nblk, rows, cgx)
    if (k == 3 && stride == 1) WGR(3, 1, 4);
    else if (k == 3) WGR(3, 2, 4);
    else if (stride == 1) WGR(5, 1, 2);
    else WGR(5, 2, 2);
#undef WGR
  } else {
    const int cgx = cg4 < 64 ? cg4 : 64, P = 256 / cgx;
    dim3 grid(nblk, batch);
    const size_t lds = (size_t)k * P * cgx * 16;
#define WG(K_, S_) hipLaunchKernelGGL((dw_wgrad_kernel<K_, S_>), grid, dim3(256), lds, st, x, dy, scratch, in_h, in_w, channels, Ho, Wo, circular, nblk, rows)
    if (k == 3 && stride == 1) WG(3, 1);
    else if (k == 3) WG(3, 2);
    else if (stride == 1) WG(5, 1);
    else WG(5, 2);
#undef WG
  }
  const int n = k * k * channels;
  launch_sum_parts(scratch, nblk * batch, n, n, dw, st);
  return check_launch("dw_wgrad_kernel");
}

// ---------------------------------------------------------------------------------------------
// Squeeze-excite backward (efficientnet_pytorch/model.py:113-118).  Forward per sample:
//   m = mean_px(u) ; z1 = W1 m + b1 ; a = swish(z1) ; z2 = W2 a + b2 ; gate = sigmoid(z2)
// se_bwd_sample_kernel (one workgroup per sample) recomputes the forward from the squeeze partials and emits
//   dz2 [B][C], dz1 [B][Cs], a [B][Cs], m [B][C] and dmean [B][C] = (W1^T dz1) / HW
// se_bwd_weights_kernel then reduces over the batch: dW2[c][s] = sum_b dz2[b,c] a[b,s] (reference layout
// [C][Cs]), dW1[s][c] = sum_b dz1[b,s] m[b,c], db2 = sum_b dz2, db1 = sum_b dz1.
// ---------------------------------------------------------------------------------------------
namespace ccvpe {

// out[c] = scale * sum_k rows[k][c] for one sample's [nrows][C] partial rows (C % 4 == 0): a thread owns four channels, the
// 256 / (C/4) row lanes walk the rows with four 16-byte loads in flight and are combined through LDS in lane order (fixed
// assignment: deterministic).  One channel per thread walked all rows serially: ~16 dependent L2 round trips for the early
// MBConv blocks (50-128 partial rows), most of this kernel's 54 us.
__device__ __forceinline__ void se_sum_rows(const float* __restrict__ rows, int nrows, int C, float scale, float* __restrict__ out,
                                            f32x4* __restrict__ red4, int tid) {
  const int C4 = C >> 2;
  const int cw = C4 < 256 ? C4 : 256;
  const int R = 256 / cw;
  for (int c0 = 0; c0 < C4; c0 += cw) {
    const int cl = tid % cw, rr = tid / cw;
    const int c4 = c0 + cl;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (rr < R && c4 < C4) {
      const f32x4* p = reinterpret_cast<const f32x4*>(rows) + c4;
      int q = rr;
      for (; q + 3 * R < nrows; q += 4 * R) {
        s0 += p[(size_t)q * C4];
        s1 += p[(size_t)(q + R) * C4];
        s2 += p[(size_t)(q + 2 * R) * C4];
        s3 += p[(size_t)(q + 3 * R) * C4];
      }
      for (; q < nrows; q += R) s0 += p[(size_t)q * C4];
    }
    red4[tid] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rr == 0 && c4 < C4) {
      f32x4 t = red4[cl];
      for (int j = 1; j < R; ++j) t += red4[j * cw + cl];
      *reinterpret_cast<f32x4*>(out + 4 * c4) = t * scale;
    }
    __syncthreads();
  }
}

// y[s] = sum_c W[s][c] v[c] for s < Cs (W row-major [Cs][C], v in LDS): four rows per pass and wave with independent accumulators
// (their loads are in flight together), lanes stride over c.  The result of row s is handed to `fin(s, sum)` on lane 0.
template <typename Fin>
__device__ __forceinline__ void se_matvec_rows(const float* __restrict__ W, const float* __restrict__ v, int C, int Cs, int lane,
                                               int wv, Fin fin) {
  for (int j0 = wv * 4; j0 < Cs; j0 += 16) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const int j1 = min(j0 + 1, Cs - 1), j2 = min(j0 + 2, Cs - 1), j3 = min(j0 + 3, Cs - 1);
#pragma unroll 6
    for (int c = lane; c < C; c += 64) {
      const float m = v[c];
      s0 = fmaf(W[(size_t)j0 * C + c], m, s0);
      s1 = fmaf(W[(size_t)j1 * C + c], m, s1);
      s2 = fmaf(W[(size_t)j2 * C + c], m, s2);
      s3 = fmaf(W[(size_t)j3 * C + c], m, s3);
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
    if (lane == 0) {
      fin(j0, s0);
      if (j0 + 1 < Cs) fin(j0 + 1, s1);
      if (j0 + 2 < Cs) fin(j0 + 2, s2);
      if (j0 + 3 < Cs) fin(j0 + 3, s3);
    }
  }
}

__global__ __launch_bounds__(256) void se_bwd_sample_kernel(const float* __restrict__ se_partial, int nblk, float inv_hw,
                                                            const float* __restrict__ dg_partial, int ndg,
                                                            const float* __restrict__ w1, const float* __restrict__ b1,
                                                            const float* __restrict__ w2t, const float* __restrict__ b2,
                                                            float* __restrict__ dz2_o, float* __restrict__ dz1_o,
                                                            float* __restrict__ a_o, float* __restrict__ m_o,
                                                            float* __restrict__ dmean_o, int C, int Cs) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Cs4 = (Cs + 3) & ~3;
  float* m = sm;            // [C]      (C % 4 == 0: checked by the launcher)
  float* dz2 = m + C;       // [C]
  float* dgs = dz2 + C;     // [C]      sum of the dgate partial rows
  float* z1 = dgs + C;      // [Cs^4]
  float* a = z1 + Cs4;      // [Cs^4]
  float* dz1 = a + Cs4;     // [Cs^4]
  f32x4* red4 = reinterpret_cast<f32x4*>(dz1 + Cs4);   // [256]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  se_sum_rows(se_partial + (size_t)b * nblk * C, nblk, C, inv_hw, m, red4, tid);
  se_sum_rows(dg_partial + (size_t)b * ndg * C, ndg, C, 1.0f, dgs, red4, tid);
  for (int c = tid; c < C; c += 256) m_o[(size_t)b * C + c] = m[c];
  se_matvec_rows(w1, m, C, Cs, lane, wv, [&](int s, float acc) {
    const float z = acc + b1[s];
    z1[s] = z;
    a[s] = swishf(z);
    a_o[(size_t)b * Cs + s] = a[s];
  });
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float z = b2[c];
#pragma unroll 8
    for (int s = 0; s < Cs; ++s) z = fmaf(w2t[(size_t)s * C + c], a[s], z);
    const float g = sigmoidf(z);
    const float d = dgs[c] * g * (1.0f - g);
    dz2[c] = d;
    dz2_o[(size_t)b * C + c] = d;
  }
  __syncthreads();
  se_matvec_rows(w2t, dz2, C, Cs, lane, wv, [&](int s, float acc) {
    const float z = z1[s], sg = sigmoidf(z);
    const float d = acc * sg * (1.0f + z * (1.0f - sg));
    dz1[s] = d;
    dz1_o[(size_t)b * Cs + s] = d;
  });
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float dm = 0.f;
#pragma unroll 8
    for (int s = 0; s < Cs; ++s) dm = fmaf(w1[(size_t)s * C + c], dz1[s], dm);
    dmean_o[(size_t)b * C + c] = dm * inv_hw;
  }
}

__global__ __launch_bounds__(256) void se_bwd_weights_kernel(const float* __restrict__ dz2, const float* __restrict__ dz1,
                                                             const float* __restrict__ a, const float* __restrict__ m,
                                                             float* __restrict__ dw1, float* __restrict__ db1,
                                                             float* __restrict__ dw2, float* __restrict__ db2, int B,
                                                             int C, int Cs) {
  // (four independent accumulators, batch loop unrolled: the first version was one dependent load -> fma chain of B = 64 steps
  // per output, ~90 us per launch, 32 launches per training step; fixed order b = q, q+4, ... then (0+1)+(2+3))
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = C * Cs;
  auto dot_b = [&](const float* __restrict__ p, int ps, int pi, const float* __restrict__ q, int qs, int qi) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int b = 0;
#pragma unroll 2
    for (; b + 3 < B; b += 4) {
      a0 = fmaf(p[(size_t)b * ps + pi], q[(size_t)b * qs + qi], a0);
      a1 = fmaf(p[(size_t)(b + 1) * ps + pi], q[(size_t)(b + 1) * qs + qi], a1);
      a2 = fmaf(p[(size_t)(b + 2) * ps + pi], q[(size_t)(b + 2) * qs + qi], a2);
      a3 = fmaf(p[(size_t)(b + 3) * ps + pi], q[(size_t)(b + 3) * qs + qi], a3);
    }
    for (; b < B; ++b) a0 = fmaf(p[(size_t)b * ps + pi], q[(size_t)b * qs + qi], a0);
    return (a0 + a1) + (a2 + a3);
  };
  auto sum_b = [&](const float* __restrict__ p, int ps, int pi) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int b = 0;
#pragma unroll 2
    for (; b + 3 < B; b += 4) {
      a0 += p[(size_t)b * ps + pi];
      a1 += p[(size_t)(b + 1) * ps + pi];
      a2 += p[(size_t)(b + 2) * ps + pi];
      a3 += p[(size_t)(b + 3) * ps + pi];
    }
    for (; b < B; ++b) a0 += p[(size_t)b * ps + pi];
    return (a0 + a1) + (a2 + a3);
  };
  if (i < n) {
    dw2[i] = dot_b(dz2, C, i / Cs, a, Cs, i % Cs);      // dW2 [C][Cs]
    dw1[i] = dot_b(dz1, Cs, i / C, m, C, i % C);        // dW1 [Cs][C]
  }
  if (i < C) db2[i] = sum_b(dz2, C, i);
  if (i < Cs) db1[i] = sum_b(dz1, Cs, i);
}

}  // namespace ccvpe

extern "C" int ccvpe_se_bwd_f32(const float* se_partial, int nblk, float inv_hw, const float* dgate_partial, int ndg,
                                const float* w1, const float* b1, const float* w2t, const float* b2, float* dmean,
                                float* dw1, float* db1, float* dw2, float* db2, float* scratch, int batch, int channels,
                                int squeezed, void* stream) {
  if (batch <= 0 || channels <= 0 || squeezed <= 0 || nblk <= 0 || ndg <= 0) return fail(CCVPE_EINVAL, "se_bwd: bad shape");
  if (channels % 4 || !aligned16(se_partial) || !aligned16(dgate_partial))
    return fail(CCVPE_EINVAL, "se_bwd: channels %% 4 == 0 and 16-byte aligned partial rows required");
  hipStream_t st = (hipStream_t)stream;
  float* dz2 = scratch;                              // [B][C]
  float* m = dz2 + (size_t)batch * channels;         // [B][C]
  float* dz1 = m + (size_t)batch * channels;         // [B][Cs]
  float* a = dz1 + (size_t)batch * squeezed;         // [B][Cs]
  const size_t lds = (size_t)(3 * channels + 3 * ((squeezed + 3) & ~3) + 1024) * sizeof(float);
  hipLaunchKernelGGL(se_bwd_sample_kernel, dim3(batch), dim3(256), lds, st, se_partial, nblk, inv_hw, dgate_partial, ndg, w1,
                     b1, w2t, b2, dz2, dz1, a, m, dmean, channels, squeezed);
  hipLaunchKernelGGL(se_bwd_weights_kernel, dim3((channels * squeezed + 255) / 256), dim3(256), 0, st, dz2, dz1, a, m, dw1,
                     db1, dw2, db2, batch, channels, squeezed);
  return check_launch("se_bwd");
}

// v[b,px,c] = u[b,px,c] * gate[b,c]: the SE product materialised (train mode only: the projection conv's
// weight gradient needs the gated tensor; in eval the gate is applied inside the GEMM's operand load).
namespace ccvpe {
__global__ __launch_bounds__(256) void gate_mul_kernel(const float* __restrict__ u, const float* __restrict__ gate,
                                                       float* __restrict__ v, long rows_per_sample, int C, long total4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int cg4 = C >> 2;
  const int c = (int)(i % cg4) * 4;
  const long b = (i / cg4) / rows_per_sample;
  reinterpret_cast<f32x4*>(v)[i] = reinterpret_cast<const f32x4*>(u)[i] * *reinterpret_cast<const f32x4*>(gate + b * C + c);
}
}  // namespace ccvpe

extern "C" int ccvpe_gate_mul_f32(const float* u, const float* gate, float* v, int batch, int rows_per_sample, int channels,
                                  void* stream) {
  if (batch <= 0 || rows_per_sample <= 0 || channels <= 0 || channels % 4) return fail(CCVPE_EINVAL, "gate_mul: bad shape");
  const long total4 = (long)batch * rows_per_sample * (channels / 4);
  hipLaunchKernelGGL(gate_mul_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u, gate, v,
                     (long)rows_per_sample, channels, total4);
  return check_launch("gate_mul_kernel");
}
