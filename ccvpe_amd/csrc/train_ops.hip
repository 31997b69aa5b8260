// Train-mode BatchNorm for the EfficientNet encoders (efficientnet_pytorch/model.py:63,73,87,182,210 with
// momentum 0.01, eps 1e-3 — utils.py:665-666) and drop_connect (utils.py:129-154).
//
// Eval mode folds BN into the conv epilogues; in train mode the statistics are those of the batch, so the
// conv writes its raw output, `bn_stats` reduces it per channel and `bn_act` normalises.
//
//   bn_stats : per-channel (count, mean, M2) partials per workgroup (shifted sums over the workgroup's rows,
//              fixed order), merged k-way (bn_stats_fold_kernel, Chan's formula) -> mean, biased variance, and
//              the running-statistics update (unbiased variance, momentum m):  deterministic, and free
//              of the E[x^2]-E[x]^2 cancellation.
//   bn_act   : y = act((x-mean)*rsqrt(var+eps)*gamma+beta) [* dc_scale[b]] [+ residual], optional
//              per-(sample, workgroup) channel sums of y for the SE squeeze.
// All HBM-bound elementwise/reduction work on fp32 NHWC rows.
#include "common.h"

namespace ccvpe {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BNS_ROWS = 256;   // rows per workgroup in the stats kernel

// thread = 4 consecutive channels (one 16-byte load per row) x row lane: thread t handles channel group
// (t % cw) for rows r = t / cw, +R, ...
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ x, long rows, int C,
                                                               float* __restrict__ part /*[nblk][3][C]*/, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [R][cw][3] float4
  const int cg4 = C >> 2;
  const int cw = cg4 < 256 ? cg4 : 256;
  const int R = 256 / cw;
  const int tid = threadIdx.x;
  const int cl = tid % cw, rr = tid / cw;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  f32x4* sm4 = reinterpret_cast<f32x4*>(sm);
  for (int c0 = 0; c0 < cg4; c0 += cw) {
    const int cg = c0 + cl;
    f32x4 n = {0.f, 0.f, 0.f, 0.f}, mean = n, m2 = n;
    if (rr < R && cg < cg4 && r0 + rr < r1) {
      // shifted sums around the slice's first value (a sample of the data, so |x - pivot| ~ sigma: no E[x^2]-E[x]^2
      // cancellation) — 2 FMAs per element instead of Welford's divide
      const float* xc = x + cg * 4;
      const f32x4 pivot = *reinterpret_cast<const f32x4*>(xc + (r0 + rr) * C);
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = s;
      float cnt = 0.f;
      long r = r0 + rr;
      for (; r + 3L * R < r1; r += 4L * R) {           // four rows per trip: four loads in flight per thread, fixed order
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(xc + r * C);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(xc + (r + R) * C);
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(xc + (r + 2L * R) * C);
        const f32x4 v3 = *reinterpret_cast<const f32x4*>(xc + (r + 3L * R) * C);
        const f32x4 d0 = v0 - pivot, d1 = v1 - pivot, d2 = v2 - pivot, d3 = v3 - pivot;
        s += (d0 + d1) + (d2 + d3);
        q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        cnt += 4.f;
      }
      for (; r < r1; r += R) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(xc + r * C) - pivot;
        s += d;
        q += d * d;
        cnt += 1.f;
      }
      const float inv = 1.0f / cnt;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        n[j] = cnt;
        mean[j] = pivot[j] + s[j] * inv;
        m2[j] = fmaxf(q[j] - s[j] * s[j] * inv, 0.f);
      }
    }
    if (rr < R) {
      sm4[(rr * cw + cl) * 3 + 0] = n;
      sm4[(rr * cw + cl) * 3 + 1] = mean;
      sm4[(rr * cw + cl) * 3 + 2] = m2;
    }
    __syncthreads();
    if (rr == 0 && cg < cg4) {                           // Chan merge over the R row slices, fixed order
      f32x4 na = sm4[cl * 3], ma = sm4[cl * 3 + 1], sa = sm4[cl * 3 + 2];
      for (int j = 1; j < R; ++j) {
        const f32x4 nb = sm4[(j * cw + cl) * 3], mb = sm4[(j * cw + cl) * 3 + 1], sb = sm4[(j * cw + cl) * 3 + 2];
        if (nb[0] > 0.f) {                               // the count is the same for the 4 channels of a thread
          const float nt = na[0] + nb[0], f = nb[0] / nt, g = na[0] * nb[0] / nt;
          const f32x4 d = mb - ma;
          ma += d * f;
          sa += sb + d * d * g;
          na = (f32x4){nt, nt, nt, nt};
        }
      }
      float* p = part + (size_t)blockIdx.x * 3 * C + cg * 4;
      *reinterpret_cast<f32x4*>(p) = na;
      *reinterpret_cast<f32x4*>(p + C) = ma;
      *reinterpret_cast<f32x4*>(p + 2 * C) = sa;
    }
    __syncthreads();
  }
}

// Merge of (count, mean, M2) partial rows: workgroup (x, g) folds rows [g*group, (g+1)*group) of channels
// [16x, 16x+16) into one row — or, when FINAL, into mean / biased variance and the running-statistics update.
// 256 threads = 16 channels x 16 row lanes; every lane walks its rows twice, with no divide inside the loops:
//   N = sum n_j,  mean = m_ref + sum n_j (m_j - m_ref) / N,  M2 = sum [M2_j + n_j (m_j - mean)^2]
// (the k-way form of Chan's update; m_ref = the group's first mean keeps the first sum small).  The lane sums meet in
// LDS and are added in lane order by every thread: deterministic.  The serial pairwise merge this replaces walked 64
// rows per thread with a divide per row: 37 + 12 us per BatchNorm, 4.8 ms of a 98-BatchNorm training step.
template <bool FINAL>
__global__ __launch_bounds__(256) void bn_stats_fold_kernel(const float* __restrict__ part, int nblk, int C, int group,
                                                            float* __restrict__ out, float* __restrict__ mean,
                                                            float* __restrict__ var, float* __restrict__ run_mean,
                                                            float* __restrict__ run_var, float momentum) {
  __shared__ float sm[2][16][16];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const bool live = c < C;
  const int cc = live ? c : C - 1;
  const int j0 = blockIdx.y * group, j1 = min(j0 + group, nblk);
  const float mref = part[(size_t)j0 * 3 * C + C + cc];
  float n = 0.f, s = 0.f;
  for (int j = j0 + rl; j < j1; j += 16) {
    const float* p = part + (size_t)j * 3 * C;
    const float nb = p[cc];
    n += nb;
    s = fmaf(nb, p[C + cc] - mref, s);
  }
  sm[0][rl][cl] = n;
  sm[1][rl][cl] = s;
  __syncthreads();
  float nt = 0.f, st = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    nt += sm[0][q][cl];
    st += sm[1][q][cl];
  }
  const float mu = mref + st / nt;
  __syncthreads();
  float m2 = 0.f;
  for (int j = j0 + rl; j < j1; j += 16) {
    const float* p = part + (size_t)j * 3 * C;
    const float d = p[C + cc] - mu;
    m2 += fmaf(p[cc] * d, d, p[2 * C + cc]);
  }
  sm[0][rl][cl] = m2;
  __syncthreads();
  if (rl != 0 || !live) return;
  float m2t = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) m2t += sm[0][q][cl];
  if (!FINAL) {
    float* o = out + (size_t)blockIdx.y * 3 * C;
    o[c] = nt;
    o[C + c] = mu;
    o[2 * C + c] = m2t;
  } else {
    mean[c] = mu;
    var[c] = m2t / nt;                                          // biased: what the normalisation uses
    if (run_mean) {
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mu;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * (nt > 1.f ? m2t / (nt - 1.f) : m2t);   // unbiased
    }
  }
}

// rows of one sample are contiguous: grid = (chunks per sample, B); thread = 4 channels
__global__ __launch_bounds__(256) void bn_act_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                     const float* __restrict__ var, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, int act,
                                                     const float* __restrict__ residual,
                                                     const float* __restrict__ dc_scale, float* __restrict__ y,
                                                     float* __restrict__ se_partial, int rows_per_sample, int C,
                                                     int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [P][cgx] float4 (SE partials)
  const int b = blockIdx.y;
  const int cg4 = C >> 2;
  const int cgx = cg4 < 256 ? cg4 : 256;
  const int P = 256 / cgx;
  const int tid = threadIdx.x;
  const int cgl = tid % cgx, pl = tid / cgx;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(r0 + rows_per_block, rows_per_sample);
  const float dcs = dc_scale ? dc_scale[b] : 1.0f;
  for (int cc = 0; cc < cg4; cc += cgx) {
    const int cg = cc + cgl;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    if (pl < P && cg < cg4) {
      const int c = cg * 4;
      const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c);
      const f32x4 vv = *reinterpret_cast<const f32x4*>(var + c);
      const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
      const f32x4 be = *reinterpret_cast<const f32x4*>(beta + c);
      f32x4 sc, sh;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sc[j] = ga[j] / sqrtf(vv[j] + eps);
        sh[j] = be[j] - mu[j] * sc[j];
      }
      auto one = [&](size_t off, f32x4 xv, f32x4 rv) {
        f32x4 v = xv * sc + sh;
        if (act == CCVPE_ACT_SWISH) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = swishf(v[j]);
        } else if (act == CCVPE_ACT_RELU) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        v *= dcs;
        if (residual) v += rv;
        *reinterpret_cast<f32x4*>(y + off) = v;
        sum += v;
      };
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      int r = r0 + pl;
      for (; r + P < r1; r += 2 * P) {                 // two rows per trip: their loads are issued together
        const size_t o0 = ((size_t)b * rows_per_sample + r) * C + c, o1 = o0 + (size_t)P * C;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(x + o0), x1 = *reinterpret_cast<const f32x4*>(x + o1);
        const f32x4 q0 = residual ? *reinterpret_cast<const f32x4*>(residual + o0) : zero4;
        const f32x4 q1 = residual ? *reinterpret_cast<const f32x4*>(residual + o1) : zero4;
        one(o0, x0, q0);
        one(o1, x1, q1);
      }
      if (r < r1) {
        const size_t o0 = ((size_t)b * rows_per_sample + r) * C + c;
        one(o0, *reinterpret_cast<const f32x4*>(x + o0), residual ? *reinterpret_cast<const f32x4*>(residual + o0) : zero4);
      }
    }
    if (se_partial) {
      f32x4* red4 = reinterpret_cast<f32x4*>(red);
      if (pl < P) red4[pl * cgx + cgl] = sum;
      __syncthreads();
      if (pl == 0 && cg < cg4) {
        f32x4 t = red4[cgl];
        for (int q = 1; q < P; ++q) t += red4[q * cgx + cgl];
        *reinterpret_cast<f32x4*>(se_partial + ((size_t)b * gridDim.x + blockIdx.x) * C + cg * 4) = t;
      }
      __syncthreads();
    }
  }
}

}  // namespace ccvpe

using namespace ccvpe;

// rows per workgroup grows with the tensor so that there are at most ~4096 partial rows; those are merged in two
// levels (groups of BNS_GROUP, then the groups) by bn_stats_fold_kernel.
constexpr int BNS_GROUP = 64;
static void bn_stats_geometry(long rows, int* rpb, int* nblk, int* ngroups) {
  const long mult = (rows + (long)BNS_ROWS * 4096 - 1) / ((long)BNS_ROWS * 4096);
  *rpb = (int)(BNS_ROWS * (mult < 1 ? 1 : mult));
  if (rows < 65536) *rpb = 32;       // small tensors (late blocks, small batches): more, shorter workgroups — a handful of
                                     // workgroups walking 256 rows each is pure load latency
  *nblk = (int)((rows + *rpb - 1) / *rpb);
  *ngroups = *nblk > BNS_GROUP ? (*nblk + BNS_GROUP - 1) / BNS_GROUP : 0;
}

extern "C" int ccvpe_bn_stats_nblk(int rows) {
  int rpb, nblk, ng;
  bn_stats_geometry(rows, &rpb, &nblk, &ng);
  return nblk + ng;
}

extern "C" int ccvpe_bn_stats_f32(const float* x, int rows, int channels, float* mean, float* var, float* run_mean,
                                  float* run_var, float momentum, float* scratch, void* stream) {
  if (rows <= 0 || channels <= 0 || channels % 4) return fail(CCVPE_EINVAL, "bn_stats: bad shape (channels %% 4)");
  if (!aligned16(x) || !aligned16(scratch)) return fail(CCVPE_EINVAL, "bn_stats: 16-byte alignment");
  if ((run_mean == nullptr) != (run_var == nullptr)) return fail(CCVPE_EINVAL, "bn_stats: run_mean/run_var both or none");
  int rpb, nblk, ng;
  bn_stats_geometry(rows, &rpb, &nblk, &ng);
  const int cg4 = channels / 4;
  const int cw = cg4 < 256 ? cg4 : 256;
  const int R = 256 / cw;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(nblk), dim3(256), (size_t)R * cw * 3 * 16, st, x, (long)rows,
                     channels, scratch, rpb);
  const float* fin = scratch;
  int nfin = nblk;
  if (ng > 0) {
    float* lvl1 = scratch + (size_t)nblk * 3 * channels;
    hipLaunchKernelGGL(bn_stats_fold_kernel<false>, dim3((channels + 15) / 16, ng), dim3(256), 0, st, scratch, nblk, channels,
                       BNS_GROUP, lvl1, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, 0.f);
    fin = lvl1;
    nfin = ng;
  }
  hipLaunchKernelGGL(bn_stats_fold_kernel<true>, dim3((channels + 15) / 16, 1), dim3(256), 0, st, fin, nfin, channels, nfin,
                     (float*)nullptr, mean, var, run_mean, run_var, momentum);
  return check_launch("bn_stats");
}

extern "C" int ccvpe_bn_act_nblk(int rows_per_sample) {
  int rpb = rows_per_sample / 32;
  if (rpb < 8) rpb = 8;
  return (rows_per_sample + rpb - 1) / rpb;
}

extern "C" int ccvpe_bn_act_f32(const float* x, const float* mean, const float* var, const float* gamma,
                                const float* beta, float eps, int act, const float* residual, const float* dc_scale,
                                float* y, float* se_partial, int batch, int rows_per_sample, int channels,
                                void* stream) {
  if (batch <= 0 || rows_per_sample <= 0 || channels <= 0 || channels % 4) return fail(CCVPE_EINVAL, "bn_act: bad shape");
  if (!aligned16(x) || !aligned16(y) || (residual && !aligned16(residual)) || !aligned16(mean) || !aligned16(var) ||
      !aligned16(gamma) || !aligned16(beta))
    return fail(CCVPE_EINVAL, "bn_act: pointers must be 16-byte aligned");
  int rpb = rows_per_sample / 32;
  if (rpb < 8) rpb = 8;
  const int nblk = (rows_per_sample + rpb - 1) / rpb;
  const int cg4 = channels / 4;
  const int cgx = cg4 < 256 ? cg4 : 256;
  const int P = 256 / cgx;
  hipLaunchKernelGGL(bn_act_kernel, dim3(nblk, batch), dim3(256), (size_t)P * cgx * 16, (hipStream_t)stream, x, mean, var,
                     gamma, beta, eps, act, residual, dc_scale, y, se_partial, rows_per_sample, channels, rpb);
  return check_launch("bn_act_kernel");
}
