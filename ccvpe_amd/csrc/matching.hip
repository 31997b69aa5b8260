// Rotational matching + LMU concat, fused into one pass over the aerial feature volume.
//
// Reference pattern (models.py:186-205, repeated at :211-228 ... :299-315; ori_prior :484-514;
// KITTI :788-806): per rotation hypothesis i it materialises roll(X, -i*stride), slices the first
// L channels, takes a norm, a repeat()-ed product, a sum and a division, then stacks, maxes,
// F.normalize()s X again and cats — about 20x the algorithmic traffic.
//
// Here: one kernel reads X ONCE from HBM — a (TP pixels x all C channels) tile is staged in LDS and serves the
// norms, the scores AND the normalised copy — and writes the score volume (NCHW, returned to the caller) and the decoder
// input [X/||X||, max score, (level-1 orientation scores), 0-pad] exactly once, as whole rows.
// (Round 2 walked the channels in 32-wide slabs and re-read the tile from global memory for the normalised copy: the PMC
// pass showed 2.5x the algorithmic read traffic at B = 64 — the "second touch is L2-hot" assumption did not hold, and a
// 128-byte slab of a 160-byte pixel row touches two cache lines per row per slab.)
//
// HBM-bound: bytes/pixel = esz*C read + esz*ldo write + 4*n_shifts scores; the arithmetic
// (n_shifts*C FMAs per pixel) is <1% of the forward's FLOPs, so it stays on the VALU.
//
// Mapping: 256 threads; LPP lanes share one pixel (LPP = 1 for C <= 24, 2 for C = 40 ... 32 for C >= 400), so a workgroup owns
// TP = 256/LPP pixels and the tile is ~22 KB whatever the level (4+ workgroups per CU, ~6 16-byte loads in flight per
// thread).  The tile is stored pixel-major with a row pitch S chosen so that S/4 = LPP * odd: the ds_read_b128 of a lane
// group (16 lanes = 16/LPP pixels x LPP consecutive 16-byte pieces) then covers all 64 banks exactly once.  Lane `sub` of
// a pixel walks the 4-channel groups sub, sub + LPP, ...; the ground descriptor is shared by every pixel of a sample, so it
// sits in LDS as a doubled table gg[k] = g_ext[k mod C] (k < 2C, g_ext = g zero-extended to C):
//   dot_i = sum_c' x[c'] * gg[c' + off_i],  off_i = (-i*stride - window_offset) mod C;
// the LPP partial sums are combined with a butterfly.  For partial windows (L < C: FoV < 360, KITTI) the window norm uses the
// same trick with a 0/1 table; for L == C it is the pixel's total norm.
//
// Round 6 — the N_rot >= 9 configurations on the matrix cores (MFMA = true).  With 20 / 21 hypotheses the per-pixel work is a
// dense contraction: scores[i][p] = sum_c G[i][c] X[c][p] with the CIRCULANT G[i][c] = gg[c + off_i] of the sample's ground
// descriptor — on the VALU it was 20 x C scalar FMAs per pixel, each group of four fed by its own 16-byte
// read of the doubled table (the kernel ran at a third of the rate of its one-hypothesis form: LDS-issue bound).  Now phase 2 is
// v_mfma_f32_16x16x4_f32 (exact fp32, an fmaf chain: no precision change): A = 16 hypotheses x 4 channels read straight from
// the doubled table (lane = hypothesis row, the lane's offset off_i folded into its address: one 16- or 8-byte read per four
// matrix instructions), B = 16 pixels x 4 channels = the tile row piece the VALU form read anyway; the window norms of partial
// windows (FoV < 360, KITTI) are the same product with the 0/1 table and x^2.  A wave owns pixel tiles (C <= 80) or a K slice
// of one tile (C >= 160: partial sums meet in LDS in fixed order).  Norms, max, normalised copy and the stores are unchanged.
#include "common.h"

namespace ccvpe {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct MatchOffsets {
  int off[CCVPE_MAX_SHIFTS];
};

// lanes per pixel and tile row pitch (floats) for C channels
// Channels per lane before a pixel gets another lane.  24 since round 5 (48 before: one lane per pixel at C = 40, 256 pixels and
// 70 KB of LDS per workgroup, two workgroups per CU): with 20 / 21 rotation hypotheses the three phases of a workgroup (tile in,
// 20 x C FMAs per pixel, rows out) are each a few microseconds and two workgroups do not cover them — halving the tile doubles the
// workgroups per CU: C2 (N_rot = 20, B = 32) 5.47 -> 5.36 ms, the one-hypothesis configurations unchanged (tools/gpu/ab_match_lpp.sh).
#ifdef CCVPE_ABLATE   // diagnostics build: CCVPE_MATCH_LPP_DIV overrides it
static inline int match_lpp_div() { static const int d = getenv("CCVPE_MATCH_LPP_DIV") ? atoi(getenv("CCVPE_MATCH_LPP_DIV")) : 24; return d; }
#else
static inline int match_lpp_div() { return 24; }
#endif
static inline int match_lpp(int C) { int l = 1; while (l < 32 && C > match_lpp_div() * l) l *= 2; return l; }
static inline int match_pitch(int C, int lpp) {
  if (lpp >= 16) return C + 4;                    // a lane group reads 256 contiguous bytes of ONE pixel: any pitch works
  int s = C + 4;                                  // (C % 8 == 0)  ->  s/4 = lpp * odd
  while ((s / 4) % lpp || ((s / 4) / lpp) % 2 == 0) s += 4;
  return s;
}

static inline int match_pitch_mfma(int C) { return ((C + 15) & ~15) + 4; }   // rows zero-padded to 16 channels; S/4 odd

// TX = storage type of X and of the concat output (float or bf16); scores, g and all math are fp32.
// MFMA: phase 2 on the matrix cores (NPAD = 16 or 32 hypothesis rows, VEC >= 2); see the header.
template <typename TX, int NPAD, bool PARTIAL, int VEC, bool MFMA = false>
__global__ __launch_bounds__(256) void match_kernel(const TX* __restrict__ x, int ldx, const float* __restrict__ g,
                                                    int ldg, int L, const MatchOffsets mo, int n_shifts, int n_max,
                                                    int n_tail, float* __restrict__ scores,
                                                    TX* __restrict__ dstx, int ldo, int hw, int C, int lpp_log2, int S) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int LPP = 1 << lpp_log2;
  const int TP = 256 >> lpp_log2;                   // pixels per workgroup
  const int TL = MFMA ? 2 * C + 16 : 2 * C;          // table length (MFMA: the zero-padded last 16-channel block reads past 2C)
  float* gg = sm;                                   // [TL]
  float* ww = gg + TL;                              // [TL] (PARTIAL only)
  float* xs = ww + (PARTIAL ? TL : 0);              // [TP][S]   the tile, pixel-major
  float* sc_s = xs + TP * S;                        // [n_shifts][TP] scores of the tile (written out coalesced)
  float* inv_s = sc_s + n_shifts * TP;              // [TP]  1 / max(||x||, 1e-12)
  float* mx_s = inv_s + TP;                         // [TP]  max score
  float* red = mx_s + TP;                           // [4]
  float* pbuf = red + 4;                            // MFMA, K slices: [4 waves][(1 + PARTIAL) * NPAD + 1][16] partial sums

  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * TP;
  const int npx = min(TP, hw - p0);
  const TX* xb = x + ((size_t)b * hw + p0) * ldx;

  // ---- phase 1: the tile, global -> LDS (whole pixel rows, 16 bytes per request — 4 fp32 or 8 bf16 channels —, 4 requests
  // in flight per thread; with 8-byte bf16 requests the bf16 kernel ran at HALF the fp32 kernel's rate: 1.33 vs 0.65 ms) ----
  constexpr int EPR = 16 / (int)sizeof(TX);         // elements per request
  const int c4n = C >> 2;
  {
    constexpr int LIF = MFMA ? 8 : 4;                // requests in flight per thread
    const int gpr = C / EPR;                         // requests per pixel row (C % 8 == 0)
    const int total = npx * gpr;
    for (int i0 = tid; i0 < total; i0 += LIF * 256) {
      f32x4 v[LIF];
      int dsto[LIF];
#pragma unroll
      for (int u = 0; u < LIF; ++u) {
        const int idx = min(i0 + u * 256, total - 1);
        const int pp = idx / gpr;
        const int cq = (idx - pp * gpr) * EPR;
        v[u] = *reinterpret_cast<const f32x4*>(xb + (size_t)pp * ldx + cq);      // 16 raw bytes
        dsto[u] = pp * S + cq;
      }
#pragma unroll
      for (int u = 0; u < LIF; ++u)
        if (i0 + u * 256 < total) {
          if (sizeof(TX) == 4) {
            *reinterpret_cast<f32x4*>(xs + dsto[u]) = v[u];
          } else {
            const cc_bf16x8 h = __builtin_bit_cast(cc_bf16x8, v[u]);
            f32x4 lo, hi;
#pragma unroll
            for (int j = 0; j < 4; ++j) { lo[j] = (float)h[j]; hi[j] = (float)h[j + 4]; }
            *reinterpret_cast<f32x4*>(xs + dsto[u]) = lo;
            *reinterpret_cast<f32x4*>(xs + dsto[u] + 4) = hi;
          }
        }
    }
  }
  if (MFMA) {                                        // zero the pad channels C .. roundup16(C) of every tile row + the table tails
    const int padc = ((C + 15) & ~15) - C;
    for (int i = tid; i < TP * padc; i += 256) xs[(i / padc) * S + C + (i % padc)] = 0.f;
    if (tid < 16) {
      gg[2 * C + tid] = 0.f;
      if (PARTIAL) ww[2 * C + tid] = 0.f;
    }
  }
  // descriptor tables + ||g||
  float gsq = 0.f;
#pragma unroll 4
  for (int k = tid; k < 2 * C; k += 256) {
    const int kk = k < C ? k : k - C;
    const float gv = g[(size_t)b * ldg + min(kk, L - 1)];       // unconditional (a guarded load is an exec-mask branch per load:
    const float v = kk < L ? gv : 0.f;                          // the 10 loads of a thread at C = 1280 ran one after the other)
    gg[k] = v;
    if (PARTIAL) ww[k] = kk < L ? 1.f : 0.f;
    if (k < C) gsq = fmaf(v, v, gsq);
  }
  gsq = wave_sum(gsq);
  if ((tid & 63) == 0) red[tid >> 6] = gsq;
  __syncthreads();
  const float gnorm = sqrtf(red[0] + red[1] + red[2] + red[3]);

  // ---- phase 2: lane (pixel pl, part sub) walks its 4-channel groups ---------------------------------------------------
  if constexpr (MFMA) {
    // ---- phase 2 on the matrix cores ----------------------------------------------------------------------------------
    constexpr int MTL = NPAD / 16;                  // 16-row hypothesis tiles
    constexpr int NR = (PARTIAL ? 2 : 1) * NPAD + 1;  // rows of a partial-sum block: scores | window norms | total norm
    const int lane = tid & 63, wave = tid >> 6;
    const int pxl = lane & 15, kq = lane >> 4;
    const int NPT = TP >= 16 ? TP >> 4 : 1;         // 16-pixel tiles of the workgroup (TP = 8: half a tile is padding)
    const int KS = NPT >= 4 ? 1 : 4 / NPT;          // K slices per tile (4 waves)
    const int nb16 = (C + 15) >> 4;
    int aoff[MTL];                                  // A operand: lane & 15 = hypothesis row; its table offset + this lane's k group
#pragma unroll
    for (int m = 0; m < MTL; ++m) {                 // (a select chain: a lane-indexed read of the kernel argument would go through scratch)
      int o = mo.off[0];
#pragma unroll
      for (int i = 1; i < 16; ++i)
        if (16 * m + i < CCVPE_MAX_SHIFTS) o = (pxl == i && 16 * m + i < n_shifts) ? mo.off[16 * m + i] : o;
      if (m > 0 && 16 * m < CCVPE_MAX_SHIFTS) o = (pxl == 0 && 16 * m < n_shifts) ? mo.off[16 * m] : o;
      aoff[m] = o + 4 * kq;
    }
    const int slice = KS == 1 ? 0 : (NPT == 2 ? wave >> 1 : wave);
    const int kb0 = slice * nb16 / KS, kb1 = (slice + 1) * nb16 / KS;
    for (int tile = (KS == 1 ? wave : (NPT == 2 ? wave & 1 : 0)); tile < NPT; tile += 4) {
      const int pp = tile * 16 + pxl;               // rows past npx hold stale LDS: a pixel's column never mixes with another's
      const float* xrow = xs + min(pp, TP - 1) * S + 4 * kq;
      f32x4 acc[MTL], nrm[PARTIAL ? MTL : 1];
#pragma unroll
      for (int m = 0; m < MTL; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < (PARTIAL ? MTL : 1); ++m) nrm[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
      float tot = 0.f;
      for (int blk = kb0; blk < kb1; ++blk) {
        const f32x4 xq = *reinterpret_cast<const f32x4*>(xrow + 16 * blk);
        f32x4 x2;
#pragma unroll
        for (int j = 0; j < 4; ++j) { x2[j] = xq[j] * xq[j]; tot += x2[j]; }
#pragma unroll
        for (int m = 0; m < MTL; ++m) {
          const float* gp = gg + 16 * blk + aoff[m];
          f32x4 ga;
          if (VEC == 4) {
            ga = *reinterpret_cast<const f32x4*>(gp);
          } else {
            const f32x2 t0 = *reinterpret_cast<const f32x2*>(gp), t1 = *reinterpret_cast<const f32x2*>(gp + 2);
            ga = (f32x4){t0[0], t0[1], t1[0], t1[1]};
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[j], xq[j], acc[m], 0, 0, 0);
          if (PARTIAL) {
            const float* wp = ww + 16 * blk + aoff[m];
            f32x4 wa;
            if (VEC == 4) {
              wa = *reinterpret_cast<const f32x4*>(wp);
            } else {
              const f32x2 t0 = *reinterpret_cast<const f32x2*>(wp), t1 = *reinterpret_cast<const f32x2*>(wp + 2);
              wa = (f32x4){t0[0], t0[1], t1[0], t1[1]};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) nrm[PARTIAL ? m : 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j], x2[j], nrm[PARTIAL ? m : 0], 0, 0, 0);
          }
        }
      }
      tot += __shfl_xor(tot, 16, 64);               // the four k groups of a pixel
      tot += __shfl_xor(tot, 32, 64);
      if (KS > 1) {                                 // publish this slice's block; the slice-0 wave of the tile adds them up below
        float* pb = pbuf + wave * NR * 16;
#pragma unroll
        for (int m = 0; m < MTL; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pb[(16 * m + 4 * kq + r) * 16 + pxl] = acc[m][r];
            if (PARTIAL) pb[(NPAD + 16 * m + 4 * kq + r) * 16 + pxl] = nrm[PARTIAL ? m : 0][r];
          }
        if (kq == 0) pb[(NR - 1) * 16 + pxl] = tot;
        __syncthreads();                            // (KS > 1: exactly one tile per wave, so every wave arrives once)
        if (slice == 0) {
          const int wstep = NPT == 2 ? 2 : 1;       // waves holding the other slices of this tile
          for (int q = 1; q < KS; ++q) {
            const float* ob = pbuf + (wave + q * wstep) * NR * 16;
#pragma unroll
            for (int m = 0; m < MTL; ++m)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                acc[m][r] += ob[(16 * m + 4 * kq + r) * 16 + pxl];
                if (PARTIAL) nrm[PARTIAL ? m : 0][r] += ob[(NPAD + 16 * m + 4 * kq + r) * 16 + pxl];
              }
            tot += ob[(NR - 1) * 16 + pxl];
          }
        }
      }
      if (slice == 0 && pp < npx) {
        // lane (pixel pp, k group kq) holds hypotheses 16 m + 4 kq + r
        float mx = -__builtin_inff();
#pragma unroll
        for (int m = 0; m < MTL; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * m + 4 * kq + r;
            if (i < n_shifts) {
              const float wn = sqrtf(PARTIAL ? nrm[PARTIAL ? m : 0][r] : tot);
              const float sv = acc[m][r] / (wn * gnorm);
              sc_s[i * TP + pp] = sv;
              if (i < n_max && (sv > mx || sv != sv)) mx = sv;
            }
          }
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {         // NaN-propagating like torch.max, any order
          const float other = __shfl_xor(mx, o, 64);
          mx = (other > mx || other != other) ? other : mx;
        }
        if (kq == 0) {
          mx_s[pp] = mx;
          inv_s[pp] = 1.0f / fmaxf(sqrtf(tot), 1e-12f);
        }
      }
    }
  } else {
  const int sub = tid & (LPP - 1);
  const int pl = tid >> lpp_log2;
  float acc[NPAD], nrm[PARTIAL ? NPAD : 1];
#pragma unroll
  for (int i = 0; i < NPAD; ++i) acc[i] = 0.f;
#pragma unroll
  for (int i = 0; i < (PARTIAL ? NPAD : 1); ++i) nrm[i] = 0.f;
  float tot = 0.f;
  if (pl < npx) {
    const float* xrow = xs + pl * S;
    for (int t = sub; t < c4n; t += LPP) {
      const int c = t * 4;
      const f32x4 xq = *reinterpret_cast<const f32x4*>(xrow + c);
      float xv[4], x2[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xv[j] = xq[j];
        x2[j] = xv[j] * xv[j];
        tot += x2[j];
      }
#pragma unroll
      for (int i = 0; i < NPAD; ++i) {
        const int k = c + mo.off[i];
        float gv[4];
        if (VEC == 4) {
          const f32x4 t4 = *reinterpret_cast<const f32x4*>(gg + k);
          gv[0] = t4[0]; gv[1] = t4[1]; gv[2] = t4[2]; gv[3] = t4[3];
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
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(ww + k);
            wv[0] = t4[0]; wv[1] = t4[1]; wv[2] = t4[2]; wv[3] = t4[3];
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
  }
  // combine the LPP parts of a pixel (adjacent lanes; fixed butterfly order)
  for (int o = 1; o < LPP; o <<= 1) {
    tot += __shfl_xor(tot, o, 64);
#pragma unroll
    for (int i = 0; i < NPAD; ++i) acc[i] += __shfl_xor(acc[i], o, 64);
    if (PARTIAL) {
#pragma unroll
      for (int i = 0; i < NPAD; ++i) nrm[PARTIAL ? i : 0] += __shfl_xor(nrm[PARTIAL ? i : 0], o, 64);
    }
  }
  // scores, max (NaN-propagating like torch.max) -> LDS
  if (sub == 0 && pl < npx) {
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < NPAD; ++i) {
      if (i < n_shifts) {
        const float wn = sqrtf(PARTIAL ? nrm[PARTIAL ? i : 0] : tot);
        const float s = acc[i] / (wn * gnorm);
        sc_s[i * TP + pl] = s;
        if (i == 0) mx = s;
        else if (i < n_max && (s > mx || s != s)) mx = s;
      }
    }
    mx_s[pl] = mx;
    inv_s[pl] = 1.0f / fmaxf(sqrtf(tot), 1e-12f);
  }
  }
  __syncthreads();

  // ---- phase 3: whole output rows [X * inv_norm | max | tail scores | 0-pad], and the score volume (NCHW), coalesced -----
  const int tail0 = n_shifts - n_tail;
  {
    // a thread writes 16 bytes of an output row: 4 fp32 or (ldo % 8 == 0) 8 bf16 channels
    const int EO = (sizeof(TX) == 2 && (ldo & 7) == 0) ? 8 : 4;
    const int ogn = ldo / EO;
    TX* drow0 = dstx + ((size_t)b * hw + p0) * ldo;
    auto piece = [&](int pp, int c) -> f32x4 {       // channels c .. c+3 of output row pp
      f32x4 v;
      if (c < C) {
        v = *reinterpret_cast<const f32x4*>(xs + pp * S + c) * inv_s[pp];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int e = c + j - C;                   // 0: max score; 1..n_tail: orientation scores; beyond: zero padding
          v[j] = e == 0 ? mx_s[pp] : (e <= n_tail ? sc_s[(tail0 + e - 1) * TP + pp] : 0.f);
        }
      }
      return v;
    };
    for (int idx = tid; idx < npx * ogn; idx += 256) {
      const int pp = idx / ogn;
      const int c = (idx - pp * ogn) * EO;
      if (EO == 4) {
        st4<TX>(drow0 + (size_t)pp * ldo + c, piece(pp, c));
      } else {
        const f32x4 a = piece(pp, c), bq = piece(pp, c + 4);
        cc_bf16x8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[j] = (cc_bf16)a[j]; o[j + 4] = (cc_bf16)bq[j]; }
        *reinterpret_cast<cc_bf16x8*>(reinterpret_cast<cc_bf16*>(drow0) + (size_t)pp * ldo + c) = o;
      }
    }
  }
  for (int idx = tid; idx < n_shifts * TP; idx += 256) {
    const int i = idx >> (8 - lpp_log2);           // idx / TP
    const int pp = idx & (TP - 1);
    if (pp < npx) scores[((size_t)b * n_shifts + i) * hw + p0 + pp] = sc_s[i * TP + pp];
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Streaming form for the NARROW levels (C <= 80: the 128 x 128 and 256 x 256 levels, where almost all of the matching's bytes are),
// 9 <= N_rot <= 32, no tail scores.  The tiled kernel above runs three phases per workgroup in series (tile in -> product -> rows
// out) and reaches 3.1-3.6 TB/s there; with C this small a pixel's whole channel vector fits a few registers, so — as in
// csrc/pwn.hip — every WAVE streams its own contiguous run of 16-pixel tiles with no LDS and no barrier: x pieces come from
// global memory straight in MFMA B layout (lane = pixel l & 15, channels 16 blk + 4 (l >> 4) .. + 3), PF tiles ahead; the circulant A
// fragments of the current sample's ground descriptor (and the 0/1 window table for partial windows) are built in registers from g
// itself (any offset parity) and rebuilt when the run reaches the next sample; |x|^2 by two cross-lane adds; scores / max as in the
// tiled kernel; the normalised row, the max column and the zero padding leave from the registers that hold x.
// ---------------------------------------------------------------------------------------------------------------------------------
struct MatchStreamParams {
  const void* x;
  const float* g;
  float* scores;
  void* dstx;
  MatchOffsets mo;
  int ldx, ldg, L, n_shifts, n_max, ldo, hw, C, tiles, tpw;
};

template <typename TX, int MTL, bool PARTIAL, int NB>
__global__ __launch_bounds__(256) void match_stream_kernel(const MatchStreamParams p) {
  constexpr int PF = NB >= 4 ? 2 : 3;                // x register sets (tiles in flight)
  const int tid = threadIdx.x, lane = tid & 63;
  const int pxl = lane & 15, kq = lane >> 4;
  // a workgroup owns a contiguous run of 4 tpw tiles and its four waves take them INTERLEAVED (tile = run start + 4 k + wave): the
  // 64-byte pieces the four waves store into one NCHW score row at about the same time are one 256-byte run (L2 merges them)
  constexpr int TS = 4;
  const int t0 = blockIdx.x * 4 * p.tpw + (tid >> 6);
  const int t1 = min((int)(blockIdx.x + 1) * 4 * p.tpw, p.tiles);
  if (t0 >= t1) return;
  const TX* x = reinterpret_cast<const TX*>(p.x);
  TX* dstx = reinterpret_cast<TX*>(p.dstx);
  const int C = p.C, L = p.L;
  const int tiles_per_sample = p.hw >> 4;

  // this lane's table offsets as a hypothesis ROW (A operand: row = lane & 15 of tile m)
  int roff[MTL];
#pragma unroll
  for (int m = 0; m < MTL; ++m) {
    int o = p.mo.off[0];
#pragma unroll
    for (int i = 1; i < 16; ++i)
      if (16 * m + i < CCVPE_MAX_SHIFTS) o = (pxl == i && 16 * m + i < p.n_shifts) ? p.mo.off[16 * m + i] : o;
    if (m > 0 && 16 * m < CCVPE_MAX_SHIFTS) o = (pxl == 0 && 16 * m < p.n_shifts) ? p.mo.off[16 * m] : o;
    roff[m] = o;
  }
  f32x4 ga[MTL][NB], wa[PARTIAL ? MTL : 1][PARTIAL ? NB : 1];
  float gnorm = 1.f;
  auto load_tables = [&](int b) {
    const float* gb = p.g + (size_t)b * p.ldg;
    float gsq = 0.f;
    for (int k = lane; k < L; k += 64) gsq = fmaf(gb[k], gb[k], gsq);
    gnorm = sqrtf(wave_sum(gsq));
#pragma unroll
    for (int m = 0; m < MTL; ++m)
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int k = 16 * blk + 4 * kq + j + roff[m];   // index into the doubled table gg[k] = g_ext[k mod C]
          k = k >= C ? k - C : k;
          k = k >= C ? k - C : k;                    // (pad channels of the last block: k < 2C + 16)
          const float gv = gb[min(k, L - 1)];
          ga[m][blk][j] = k < L ? gv : 0.f;
          if (PARTIAL) wa[PARTIAL ? m : 0][PARTIAL ? blk : 0][j] = k < L ? 1.f : 0.f;
        }
      }
  };
  int cur_b = t0 / tiles_per_sample;
  load_tables(cur_b);

  f32x4 xf[PF][NB];
  auto load_tile = [&](int s, int t) {
    const size_t pi = (size_t)(t < t1 ? t : t0) * 16 + pxl;          // (past the run: a valid re-read, never used)
    const TX* src = x + pi * p.ldx + 4 * kq;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      const bool in = 16 * blk + 4 * kq < C;          // pad channels of the last block (C % 16 = 8): zero
      const f32x4 v = ld4<TX>(src + (in ? 16 * blk : 0));
      xf[s][blk] = in ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
#pragma unroll
  for (int s = 0; s < PF - 1; ++s) load_tile(s, t0 + TS * s);

  for (int t = t0; t < t1; t += TS * PF) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      const int tt = t + TS * s;
      if (tt < t1) {                                  // wave-uniform
        load_tile((s + PF - 1) % PF, tt + TS * (PF - 1));
        const int b = tt / tiles_per_sample;
        if (b != cur_b) {                             // wave-uniform: next sample's descriptor
          cur_b = b;
          load_tables(b);
        }
        f32x4 acc[MTL], nrm[PARTIAL ? MTL : 1];
#pragma unroll
        for (int m = 0; m < MTL; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < (PARTIAL ? MTL : 1); ++m) nrm[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float tot = 0.f;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
          const f32x4 xq = xf[s][blk];
          f32x4 x2;
#pragma unroll
          for (int j = 0; j < 4; ++j) { x2[j] = xq[j] * xq[j]; tot += x2[j]; }
#pragma unroll
          for (int m = 0; m < MTL; ++m) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[m][blk][j], xq[j], acc[m], 0, 0, 0);
            if (PARTIAL) {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                nrm[PARTIAL ? m : 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[PARTIAL ? m : 0][PARTIAL ? blk : 0][j], x2[j], nrm[PARTIAL ? m : 0], 0, 0, 0);
            }
          }
        }
        tot += __shfl_xor(tot, 16, 64);
        tot += __shfl_xor(tot, 32, 64);
        const size_t pi = (size_t)tt * 16 + pxl;      // global pixel index (all samples); hw % 16 == 0: the tile is inside sample b
        const size_t pin = pi - (size_t)b * p.hw;     // pixel inside the sample
        float mx = -__builtin_inff();
#pragma unroll
        for (int m = 0; m < MTL; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * m + 4 * kq + r;
            if (i < p.n_shifts) {
              const float wn = sqrtf(PARTIAL ? nrm[PARTIAL ? m : 0][r] : tot);
              const float sv = acc[m][r] / (wn * gnorm);
              p.scores[((size_t)b * p.n_shifts + i) * p.hw + pin] = sv;
              if (i < p.n_max && (sv > mx || sv != sv)) mx = sv;
            }
          }
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
          const float other = __shfl_xor(mx, o, 64);
          mx = (other > mx || other != other) ? other : mx;
        }
        const float inv = 1.0f / fmaxf(sqrtf(tot), 1e-12f);
        TX* drow = dstx + pi * p.ldo + 4 * kq;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
          if (16 * blk + 4 * kq < C) st4<TX>(drow + 16 * blk, xf[s][blk] * inv);
        // [max | zero padding]: ldo - C floats in 16-byte pieces, piece e written by the lane with kq == e & 3
        const int extra = (p.ldo - C) >> 2;
        for (int e = kq; e < extra; e += 4)
          st4<TX>(dstx + pi * p.ldo + C + 4 * e, (f32x4){e == 0 ? mx : 0.f, 0.f, 0.f, 0.f});
      }
    }
  }
}

int num_cus();   // narrow_bf16.hip

}  // namespace ccvpe

using namespace ccvpe;

template <typename TX, int NPAD, bool PARTIAL, int VEC, bool MFMA = false>
static int launch_match(const TX* x, int ldx, const float* g, int ldg, int L, const MatchOffsets& mo, int n_shifts,
                        int n_max, int n_tail, float* scores, TX* dstx, int ldo, int B, int hw, int C,
                        hipStream_t st) {
  const int lpp = match_lpp(C);
  int lpp_log2 = 0;
  while ((1 << lpp_log2) < lpp) ++lpp_log2;
  const int S = MFMA ? match_pitch_mfma(C) : match_pitch(C, lpp);
  const int tp = 256 / lpp;
  const int tl = MFMA ? 2 * C + 16 : 2 * C;
  const int pbuf = (MFMA && tp < 64) ? 4 * ((PARTIAL ? 2 : 1) * NPAD + 1) * 16 : 0;
  const size_t smem = sizeof(float) * ((size_t)tl * (PARTIAL ? 2 : 1) + (size_t)tp * S + (size_t)n_shifts * tp + 2 * tp + 4 + pbuf);
  if (smem > 160 * 1024) return fail(CCVPE_EINVAL, "match_level: C=%d needs %zu B of LDS", C, smem);
  auto kern = match_kernel<TX, NPAD, PARTIAL, VEC, MFMA>;
  static size_t attr_smem = 64 * 1024;             // per instantiation: the largest size asked for so far
  if (smem > attr_smem) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return fail(CCVPE_ELAUNCH, "match_level: set smem attr: %s", hipGetErrorString(e));
    attr_smem = smem;
  }
  if (B > 65535) return fail(CCVPE_EINVAL, "match_level: batch > 65535");
  dim3 grid((hw + tp - 1) / tp, B);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, x, ldx, g, ldg, L, mo, n_shifts, n_max, n_tail, scores, dstx,
                     ldo, hw, C, lpp_log2, S);
  return check_launch("match_kernel");
}

static int g_match_mfma = 2;   // 0 vector-ALU form everywhere, 1 + the tiled matrix-core form, 2 + the streaming form of the narrow levels

template <typename TX>
static int match_any(const TX* x, int ldx, const float* g, int ldg, int L, const int* shifts, int n_shifts, int n_max,
                     int n_tail, int stride, int window_offset, float* scores, TX* dstx, int ldo, int B, int hw, int C,
                     void* stream) {
  if (n_shifts < 1 || n_shifts > CCVPE_MAX_SHIFTS) return fail(CCVPE_EINVAL, "match_level: n_shifts %d out of range", n_shifts);
  if (n_max < 1 || n_max > n_shifts || n_tail < 0 || n_tail > n_shifts) return fail(CCVPE_EINVAL, "match_level: bad n_max/n_tail");
  if (C % 8 || ldx % 4 || ldo % 4 || ldo < C + 1 + n_tail) return fail(CCVPE_EINVAL, "match_level: C%%8, ldx%%4, ldo%%4, ldo>=C+1+n_tail required");
  if (sizeof(TX) == 2 && ldx % 8) return fail(CCVPE_EINVAL, "match_level: bf16 rows must be 16-byte aligned (ldx %% 8)");
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
  // narrow levels with many hypotheses: every wave streams its own run of 16-pixel tiles (no LDS tile, no barriers)
  if (g_match_mfma >= 2 && n_shifts >= 9 && n_shifts <= 32 && n_tail == 0 && C <= 80 && hw % 16 == 0 && (long)B * hw >= 65536 &&
      ldo - C <= 64 && (sizeof(TX) == 4 || ldo % 4 == 0)) {
    MatchStreamParams q;
    q.x = x; q.g = g; q.scores = scores; q.dstx = dstx; q.mo = mo;
    q.ldx = ldx; q.ldg = ldg; q.L = L; q.n_shifts = n_shifts; q.n_max = n_max; q.ldo = ldo; q.hw = hw; q.C = C;
    q.tiles = (int)((long)B * hw / 16);
    const int waves = 2 * num_cus() * 4;
    q.tpw = (q.tiles + waves - 1) / waves;
    const int wgs = ((q.tiles + q.tpw - 1) / q.tpw + 3) / 4;
    const int nb = (C + 15) / 16;
#define MS_GO(MTL_, P_, NB_) hipLaunchKernelGGL((match_stream_kernel<TX, MTL_, P_, NB_>), dim3(wgs), dim3(256), 0, st, q)
#define MS_NB(MTL_, P_) \
    if (nb <= 2) MS_GO(MTL_, P_, 2); else if (nb == 3) MS_GO(MTL_, P_, 3); else if (nb == 4) MS_GO(MTL_, P_, 4); else MS_GO(MTL_, P_, 5)
    if (n_shifts <= 16) { if (partial) { MS_NB(1, true); } else { MS_NB(1, false); } }
    else { if (partial) { MS_NB(2, true); } else { MS_NB(2, false); } }
#undef MS_NB
#undef MS_GO
    return check_launch("match_stream_kernel");
  }
  if (g_match_mfma && n_shifts >= 9 && n_shifts <= 32 && vec >= 2) {       // many hypotheses: the circulant product on the matrix cores
#define M_MFMA(NP)                                                          \
    if (partial) {                                                          \
      if (vec == 4) return launch_match<TX, NP, true, 4, true>(M_ARGS);     \
      return launch_match<TX, NP, true, 2, true>(M_ARGS);                   \
    }                                                                       \
    if (vec == 4) return launch_match<TX, NP, false, 4, true>(M_ARGS);      \
    return launch_match<TX, NP, false, 2, true>(M_ARGS);
    if (n_shifts <= 16) { M_MFMA(16) }
    M_MFMA(32)
#undef M_MFMA
  }
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

extern "C" int ccvpe_set_match_mfma(int on) {
  const int old = g_match_mfma;
  g_match_mfma = on < 0 ? 0 : (on > 2 ? 2 : on);
  return old;
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
