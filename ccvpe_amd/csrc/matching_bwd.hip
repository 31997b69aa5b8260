// Backward of the fused rotational matching + LMU concat (csrc/matching.hip; models.py:186-205 ...).
// Forward per pixel p of sample b (gg / ww = doubled descriptor / window tables, off_i = (-shift_i*stride) mod C):
//     num_i = sum_c x[c] gg[c+off_i] ;  n_i^2 = sum_c x[c]^2 ww[c+off_i] ;  s_i = num_i / (n_i G),  G = ||g||
//     dst[0:C] = x / max(||x||, 1e-12) ; dst[C] = max_{i<n_max} s_i ; dst[C+1+j] = s_{n-n_tail+j}
// Backward, with ds_i = dscores_i + [i == argmax] ddst[C] + [i >= n-n_tail] ddst[C+1+i-(n-n_tail)]:
//     a_i = ds_i / (n_i G),  b_i = ds_i s_i / n_i^2
//     dx[c]  = normalize_bwd(ddst[0:C])[c] + sum_i ( a_i gg[c+off_i] - b_i x[c] ww[c+off_i] )
//     dg[k]  = sum_p sum_i a_i x[c : (c+off_i) mod C == k]  -  g[k]/G^2 * sum_p sum_i ds_i s_i
// Same mapping as the forward (lane = pixel, 32-channel tiles transposed through LDS, wave-uniform LDS
// broadcasts of the tables).  dg is a reduction over pixels: every (channel, shift) term is summed across
// the workgroup's pixels by a small MFMA GEMM out of LDS (G[i][c] = sum_px a_i x_c: M = shifts, N = the 16 channels of the staged
// tile, K = pixels, one 64-pixel K range per wave) and folded into one LDS vector; the workgroup writes one partial row [L+1] and a
// finishing kernel reduces the rows in fixed order (deterministic).
// LDS pitches: every [row][pixel] array has pitch TPB + 4 floats (pitch mod 64 banks = 4): the transposing tile writes, the
// per-lane column reads AND the MFMA operand reads (lane -> row l%16, pixel l/16: bank 4*row + pixel) are all conflict-free.
// (Pitch TPB for the a_i table put the 4 shifts a wave reads together on ONE bank: the old per-thread G loop spent most of the
// kernel in 4-way conflicts — 2.7 ms for the 256 x 256 level of a B = 64 step.)
#include "common.h"

namespace ccvpe {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MatchOffsetsB {
  int off[CCVPE_MAX_SHIFTS];
};

constexpr int MBK = 16;   // channels per staged tile: two tiles (x, ddst) + the a_i table must leave room for >= 3 workgroups per CU

template <int NPAD>
__global__ __launch_bounds__(256) void match_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ g,
                                                        int ldg, int L, const MatchOffsetsB mo, int n_shifts, int n_max,
                                                        int n_tail, const float* __restrict__ scores,
                                                        const float* __restrict__ dscores, const float* __restrict__ ddst,
                                                        int ldo, float* __restrict__ dx, int lddx,
                                                        float* __restrict__ part, int nblk, int hw, int C, int nslice) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int MT = (NPAD + 15) / 16;  // 16-shift MFMA tiles
  const int TPB = blockDim.x;
  // (opaque to the optimiser on purpose: with the pitch KNOWN to be a multiple of 4 the compiler rewrote the channel loops below
  // into wide vector form and spilled — 256 VGPR + 256 AGPR + 2.9 KB of scratch per lane, the pathology `#pragma unroll 1` guards)
  const int XLD = __builtin_amdgcn_readfirstlane(TPB + 4);
  const int NW = TPB >> 6;
  float* gg = sm;                       // [2C]
  float* ww = gg + 2 * C;               // [2C]
  float* xs = ww + 2 * C;               // [MBK][XLD]
  float* dd = xs + MBK * XLD;           // [MBK][XLD]
  float* as = dd + MBK * XLD;           // [n_shifts][XLD]  a_i of every pixel of the workgroup
  float* Gs = as + n_shifts * XLD;      // [4 waves][16 MT][MBK]  per-wave G[i][c] = sum_px a_i(px) x(px, c) of the current tile
  float* dgw = Gs + 4 * 16 * MT * MBK;  // [L+1]
  float* red = dgw + (L + 1);           // [4]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * TPB;
  const int p = p0 + tid;
  const bool pvalid = p < hw;
  const bool partial = L < C;
  const float* xb = x + (size_t)b * hw * ldx;
  const float* db = ddst + (size_t)b * hw * ldo;

  float gsq = 0.f;
  for (int k = tid; k < 2 * C; k += TPB) {
    const int kk = k < C ? k : k - C;
    const float gv = g[(size_t)b * ldg + min(kk, L - 1)];       // unconditional load (see matching.hip)
    const float v = kk < L ? gv : 0.f;
    gg[k] = v;
    ww[k] = kk < L ? 1.f : 0.f;
    if (k < C) gsq = fmaf(v, v, gsq);
  }
  for (int k = tid; k < L + 1; k += TPB) dgw[k] = 0.f;
  gsq = wave_sum(gsq);
  if (lane == 0) red[wv] = gsq;
  __syncthreads();
  float gnorm = 0.f;
  for (int i = 0; i < NW; ++i) gnorm += red[i];
  gnorm = sqrtf(gnorm);

  // ---- pass A: ||x||^2, <x, ddst[0:C]>, window norms ------------------------------------------------
  float nrm[NPAD];
#pragma unroll
  for (int i = 0; i < NPAD; ++i) nrm[i] = 0.f;
  float tot = 0.f, xd = 0.f;
  constexpr int F4 = MBK / 4;
  for (int c0 = 0; c0 < C; c0 += MBK) {
    const int ck = min(MBK, C - c0);
    for (int idx = tid; idx < TPB * F4; idx += TPB) {
      const int pp = idx / F4;
      const int cq = (idx - pp * F4) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f};
      if (p0 + pp < hw && cq < ck) {
        v = *reinterpret_cast<const f32x4*>(xb + (size_t)(p0 + pp) * ldx + c0 + cq);
        d = *reinterpret_cast<const f32x4*>(db + (size_t)(p0 + pp) * ldo + c0 + cq);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xs[(cq + j) * XLD + pp] = v[j];
        dd[(cq + j) * XLD + pp] = d[j];
      }
    }
    __syncthreads();
#pragma unroll 1   // the body is NPAD independent LDS reads + FMAs: unrolling the channel walk as well only spills
    for (int cc = 0; cc < ck; ++cc) {
      const float xv = xs[cc * XLD + tid];
      const float x2 = xv * xv;
      tot += x2;
      xd = fmaf(xv, dd[cc * XLD + tid], xd);
      if (partial) {          // (shifts i >= n_shifts have offset 0 and are never used: no per-shift guard in the loop)
#pragma unroll
        for (int i = 0; i < NPAD; ++i) nrm[i] = fmaf(x2, ww[c0 + cc + mo.off[i]], nrm[i]);
      }
    }
    __syncthreads();
  }

  // ---- per-pixel coefficients ---------------------------------------------------------------------------
  float a[NPAD], bq[NPAD];
  float T = 0.f;
  {
    const int tail0 = n_shifts - n_tail;
    int amax = 0;
    float mx = 0.f;
    if (pvalid) {
#pragma unroll
      for (int i = 0; i < NPAD; ++i) {
        if (i < n_max) {
          const float s = scores[((size_t)b * n_shifts + i) * hw + p];
          if (i == 0) mx = s;
          else if (s > mx || s != s) { mx = s; amax = i; }
        }
      }
    }
    const float* drow = db + (size_t)p * ldo;
#pragma unroll
    for (int i = 0; i < NPAD; ++i) {
      a[i] = 0.f;
      bq[i] = 0.f;
      if (i < n_shifts && pvalid) {
        const float s = scores[((size_t)b * n_shifts + i) * hw + p];
        float ds = dscores ? dscores[((size_t)b * n_shifts + i) * hw + p] : 0.f;
        if (i == amax) ds += drow[C];
        if (i >= tail0) ds += drow[C + 1 + (i - tail0)];
        const float n2 = partial ? nrm[i] : tot;
        a[i] = ds / (sqrtf(n2) * gnorm);
        bq[i] = ds * s / n2;
        T = fmaf(ds, s, T);
      }
    }
  }
  float bqsum = 0.f;                       // full windows (L == C): sum_i b_i x[c] ww[.] = x[c] * sum_i b_i
#pragma unroll
  for (int i = 0; i < NPAD; ++i) {
    bqsum += bq[i];
    if (i < n_shifts) as[i * XLD + tid] = a[i];
  }
  const float xn = sqrtf(tot);
  const float k1 = 1.0f / fmaxf(xn, 1e-12f);
  const float k2 = xn > 1e-12f ? xd * k1 * k1 * k1 : 0.f;

  // ---- pass B: dx and the dg partials -------------------------------------------------------------------
  // blockIdx.z = channel slice: with few pixels per sample (decoder levels 6-5: one workgroup per sample) the channel
  // walk is split over nslice workgroups; each recomputes the (cheap) pass A and owns tiles [t0, t1) of pass B
  const int ntile = (C + MBK - 1) / MBK;
  const int t0 = (int)((long)ntile * blockIdx.z / nslice), t1 = (int)((long)ntile * (blockIdx.z + 1) / nslice);
  for (int c0 = t0 * MBK; c0 < min(t1 * MBK, C); c0 += MBK) {
    const int ck = min(MBK, C - c0);
    for (int idx = tid; idx < TPB * F4; idx += TPB) {
      const int pp = idx / F4;
      const int cq = (idx - pp * F4) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f};
      if (p0 + pp < hw && cq < ck) {
        v = *reinterpret_cast<const f32x4*>(xb + (size_t)(p0 + pp) * ldx + c0 + cq);
        d = *reinterpret_cast<const f32x4*>(db + (size_t)(p0 + pp) * ldo + c0 + cq);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xs[(cq + j) * XLD + pp] = v[j];
        dd[(cq + j) * XLD + pp] = d[j];
      }
    }
    __syncthreads();
    // BRANCH-FREE shift loop: a[i] = b[i] = 0 and offset 0 for the padding shifts i >= n_shifts, gg[] is already zero outside
    // the window, so sum_i a_i gg[c + off_i] needs no test at all and its NPAD LDS broadcast reads are issued together;
    // the b-term is one FMA with sum_i b_i for full windows and uses the 0/1 window table for partial ones.  (With
    // `if (i < n_shifts)` / `if (kk < L)` inside, every shift was its own basic block: one exposed LDS round trip per
    // (channel, shift) — 270 k cycles per 256-pixel workgroup at level 6.)
#pragma unroll 1   // (left to itself the compiler unrolls this 16x: 256 VGPRs + 2.9 KB of scratch per lane, 5x slower)
    for (int cc = 0; cc < ck; ++cc) {
      const float xv = xs[cc * XLD + tid];
      float acc = pvalid ? dd[cc * XLD + tid] * k1 - xv * k2 : 0.f;
      const float* ggc = gg + c0 + cc;
      float s0 = 0.f, s1 = 0.f;
      if (!partial) {
#pragma unroll
        for (int i = 0; i < NPAD; i += 2) {
          s0 = fmaf(a[i], ggc[mo.off[i]], s0);
          if (i + 1 < NPAD) s1 = fmaf(a[i + 1], ggc[mo.off[i + 1]], s1);
        }
        acc = fmaf(-bqsum, xv, acc + (s0 + s1));
      } else {
        const float* wwc = ww + c0 + cc;
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int i = 0; i < NPAD; i += 2) {
          s0 = fmaf(a[i], ggc[mo.off[i]], s0);
          t0 = fmaf(bq[i], wwc[mo.off[i]], t0);
          if (i + 1 < NPAD) {
            s1 = fmaf(a[i + 1], ggc[mo.off[i + 1]], s1);
            t1 = fmaf(bq[i + 1], wwc[mo.off[i + 1]], t1);
          }
        }
        acc = fmaf(-(t0 + t1), xv, acc + (s0 + s1));
      }
      dd[cc * XLD + tid] = acc;
    }
    // dg: G[i][c] = sum over the workgroup's pixels of a_i(px) * x(px, c): v_mfma_f32_16x16x4_f32 with A[m = shift][k = pixel] from
    // the a_i table and B[k = pixel][n = channel] from the transposed x tile (rows >= ck of the tile and pixels >= hw are zeros,
    // a_i = 0 for invalid pixels); wave wv owns pixels [64 wv, 64 wv + 64) and leaves its partial in Gs[wv].  (The first version
    // summed every (c, i) term across the wave with a butterfly: 5 ms per launch on dependent ds_bpermute chains; the second
    // gave every thread a few (i, c) entries and walked the pixels in LDS: two LDS reads per FMA, 4-way conflicts.)
    {
      const int fm = lane & 15, fk = lane >> 4;
      f32x4 gacc[MT][2];                 // even / odd K steps: two independent accumulation chains per tile
#pragma unroll
      for (int t = 0; t < MT; ++t) gacc[t][0] = gacc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* xrow = xs + fm * XLD + wv * 64 + fk;
      const float* arow[MT];
      bool aok[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int i = 16 * t + fm;
        aok[t] = i < n_shifts;
        arow[t] = as + (aok[t] ? i : 0) * XLD + wv * 64 + fk;
      }
#pragma unroll 2
      for (int ks = 0; ks < 16; ks += 2) {
        const float bv0 = xrow[4 * ks], bv1 = xrow[4 * ks + 4];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const float a0 = arow[t][4 * ks], a1 = arow[t][4 * ks + 4];
          gacc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aok[t] ? a0 : 0.f, bv0, gacc[t][0], 0, 0, 0);
          gacc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aok[t] ? a1 : 0.f, bv1, gacc[t][1], 0, 0, 0);
        }
      }
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) Gs[((wv * MT + t) * 16 + 4 * fk + j) * MBK + fm] = gacc[t][0][j] + gacc[t][1][j];
    }
    __syncthreads();
    if (tid < ck) {                    // one wave, lock-step: for a fixed shift the ck channels hit distinct dg entries
      for (int i = 0; i < n_shifts; ++i) {
        const int k = c0 + tid + mo.off[i];
        const int kk = k >= C ? k - C : k;
        float gsum = Gs[i * MBK + tid];
        for (int w = 1; w < NW; ++w) gsum += Gs[(w * MT * 16 + i) * MBK + tid];
        if (kk < L) dgw[kk] += gsum;
      }
    }
    for (int idx = tid; idx < TPB * F4; idx += TPB) {
      const int pp = idx / F4;
      const int cq = (idx - pp * F4) * 4;
      if (p0 + pp < hw && cq < ck) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = dd[(cq + j) * XLD + pp];
        *reinterpret_cast<f32x4*>(dx + ((size_t)b * hw + p0 + pp) * lddx + c0 + cq) = o;
      }
    }
    __syncthreads();
  }
  T = wave_sum(T);
  if (lane == 0) red[wv] = T;
  __syncthreads();
  float* out = part + (((size_t)b * nblk + blockIdx.x) * nslice + blockIdx.z) * (L + 1);
  for (int k = tid; k < L; k += TPB) out[k] = dgw[k];
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < NW; ++w) t += red[w];
    out[L] = blockIdx.z == 0 ? t : 0.f;          // the scalar term is the same in every slice: count it once
  }
}

__global__ __launch_bounds__(256) void match_dg_finish_kernel(const float* __restrict__ part, int nblk,
                                                              const float* __restrict__ g, int ldg, int L,
                                                              float* __restrict__ dg, int ldg_out) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  float gsq = 0.f;
  for (int k = tid; k < L; k += 256) {
    const float v = g[(size_t)b * ldg + k];
    gsq = fmaf(v, v, gsq);
  }
  gsq = wave_sum(gsq);
  if ((tid & 63) == 0) red[tid >> 6] = gsq;
  __syncthreads();
  const float g2 = red[0] + red[1] + red[2] + red[3];
  const float* pb = part + (size_t)b * nblk * (L + 1);
  float T = 0.f;
  for (int q = 0; q < nblk; ++q) T += pb[(size_t)q * (L + 1) + L];
  for (int k = tid; k < L; k += 256) {
    float s = 0.f;
    for (int q = 0; q < nblk; ++q) s += pb[(size_t)q * (L + 1) + k];
    dg[(size_t)b * ldg_out + k] = s - T * g[(size_t)b * ldg + k] / g2;
  }
}

}  // namespace ccvpe

using namespace ccvpe;

static int match_bwd_tpb(int hw) { return hw >= 256 ? 256 : ((hw + 63) / 64) * 64; }

// channel slices per pixel workgroup: enough workgroups to cover the chip, at least 4 tiles (64 channels) per slice
static int match_bwd_slices(int hw, int batch, int C) {
  const int tpb = match_bwd_tpb(hw);
  const long wgs = (long)((hw + tpb - 1) / tpb) * batch;
  int s = (int)((512 + wgs - 1) / wgs);
  const int smax = C / (4 * MBK);
  if (s > smax) s = smax;
  if (s > 32) s = 32;
  return s < 1 ? 1 : s;
}

extern "C" int ccvpe_match_bwd_nblk(int hw, int batch, int channels) {
  const int tpb = match_bwd_tpb(hw);
  return ((hw + tpb - 1) / tpb) * match_bwd_slices(hw, batch, channels);
}

template <int NPAD>
static int launch_match_bwd(const float* x, int ldx, const float* g, int ldg, int L, const MatchOffsetsB& mo, int n_shifts,
                            int n_max, int n_tail, const float* scores, const float* dscores, const float* ddst, int ldo,
                            float* dx, int lddx, float* part, int B, int hw, int C, hipStream_t st) {
  const int tpb = match_bwd_tpb(hw);
  const int nblk = (hw + tpb - 1) / tpb;
  const int nslice = match_bwd_slices(hw, B, C);
  const size_t smem = sizeof(float) * ((size_t)4 * C + (size_t)2 * MBK * (tpb + 4) + (size_t)n_shifts * (tpb + 4) +
                                       (size_t)4 * 16 * ((NPAD + 15) / 16) * MBK + (L + 1) + 4);
  if (smem > 160 * 1024) return fail(CCVPE_EINVAL, "match_level_bwd: C=%d needs %zu B of LDS", C, smem);
  auto kern = match_bwd_kernel<NPAD>;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return fail(CCVPE_ELAUNCH, "match_level_bwd: set smem attr: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(kern, dim3(nblk, B, nslice), dim3(tpb), smem, st, x, ldx, g, ldg, L, mo, n_shifts, n_max, n_tail, scores,
                     dscores, ddst, ldo, dx, lddx, part, nblk, hw, C, nslice);
  return check_launch("match_bwd_kernel");
}

extern "C" int ccvpe_match_level_bwd_f32(const float* x, int ldx, const float* g, int ldg, int L, const int* shifts,
                                         int n_shifts, int n_max, int n_tail, int stride, int window_offset,
                                         const float* scores, const float* dscores, const float* ddst, int ldo, float* dx, int lddx, float* dg,
                                         int ldg_out, float* scratch, int B, int hw, int C, void* stream) {
  if (n_shifts < 1 || n_shifts > CCVPE_MAX_SHIFTS) return fail(CCVPE_EINVAL, "match_level_bwd: n_shifts %d out of range", n_shifts);
  if (n_max < 1 || n_max > n_shifts || n_tail < 0 || n_tail > n_shifts) return fail(CCVPE_EINVAL, "match_level_bwd: bad n_max/n_tail");
  if (C % 8 || ldx % 4 || ldo % 4 || lddx % 4 || ldo < C + 1 + n_tail || lddx < C)
    return fail(CCVPE_EINVAL, "match_level_bwd: C%%8, ldx%%4, ldo%%4, lddx%%4, ldo>=C+1+n_tail required");
  if (L < 1 || L > C || L > ldg || L > ldg_out) return fail(CCVPE_EINVAL, "match_level_bwd: bad L");
  if (!aligned16(x) || !aligned16(ddst) || !aligned16(dx)) return fail(CCVPE_EINVAL, "match_level_bwd: 16-byte alignment");
  MatchOffsetsB mo;
  for (int i = 0; i < CCVPE_MAX_SHIFTS; ++i) {
    long o = 0;
    if (i < n_shifts) {
      o = (-((long)shifts[i] * stride + window_offset)) % C;
      if (o < 0) o += C;
    }
    mo.off[i] = (int)o;
  }
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (n_shifts <= 8)
    rc = launch_match_bwd<8>(x, ldx, g, ldg, L, mo, n_shifts, n_max, n_tail, scores, dscores, ddst, ldo, dx, lddx, scratch, B, hw, C, st);
  else if (n_shifts <= 16)
    rc = launch_match_bwd<16>(x, ldx, g, ldg, L, mo, n_shifts, n_max, n_tail, scores, dscores, ddst, ldo, dx, lddx, scratch, B, hw, C, st);
  else if (n_shifts <= 20)
    rc = launch_match_bwd<20>(x, ldx, g, ldg, L, mo, n_shifts, n_max, n_tail, scores, dscores, ddst, ldo, dx, lddx, scratch, B, hw, C, st);
  else if (n_shifts <= 24)
    rc = launch_match_bwd<24>(x, ldx, g, ldg, L, mo, n_shifts, n_max, n_tail, scores, dscores, ddst, ldo, dx, lddx, scratch, B, hw, C, st);
  else
    rc = launch_match_bwd<48>(x, ldx, g, ldg, L, mo, n_shifts, n_max, n_tail, scores, dscores, ddst, ldo, dx, lddx, scratch, B, hw, C, st);
  if (rc) return rc;
  hipLaunchKernelGGL(match_dg_finish_kernel, dim3(B), dim3(256), 0, st, scratch, ccvpe_match_bwd_nblk(hw, B, C), g, ldg, L, dg, ldg_out);
  return check_launch("match_dg_finish_kernel");
}
