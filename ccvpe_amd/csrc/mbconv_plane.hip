// MBConv front half for the LATE blocks (planes of <= 1024 pixels: EfficientNet-B0 blocks 6-15 at 32x32 ... 10x20):
// expand 1x1 + BN0 + swish -> depthwise KxK (stride S) + BN1 + swish + SE squeeze sums, the expanded tensor kept in LDS.
// efficientnet_pytorch/model.py:102-110,114 (padding utils.py:254-358).
//
// Round 6.  Before: pointwise GEMM (writes the 6x tensor) -> dwconv_plane_kernel (reads it back; bf16 -> fp32 conversion and
// border masks on EVERY window read: 8 vector instructions per 16-byte read beside 5 packed FMAs) — 20 + 20 launches of
// 20-80 us per forward, the depthwise ones at 2x their vector-ALU floor.  The tile-owning mbconv_front_kernel of the early
// blocks does not extend here: with Cin = 80-192 its x halo no longer fits the registers, and a 16-column tile of a 16-32
// pixel wide plane leaves waves idle (measured in round 1).
//
// Mapping: the work is split over CHANNELS, not space — a depthwise conv never mixes channels, so a workgroup that owns
// (sample, CH mid channels, a band of BH whole output rows) needs NO halo columns from a neighbour and recomputes only the
// K - S halo rows between bands (none when the plane is one band):
//   phase 0  the out-of-image rows / columns of the LDS plane are zeroed (the depthwise conv pads the EXPANDED tensor);
//   phase 1  expand: the band's real pixels, numbered linearly, in 16-pixel MFMA tiles; x fragments straight from global
//            memory in operand layout (16 bytes per lane; the CH-channel slices of a sample re-read x from L2: neighbouring
//            slices run on one XCD), W fragments resident in registers, BN0 + swish, one ds_write_b128 per lane into the
//            fp32 plane [row][column + pad][CH] — fp32 so that the window reads below need no conversion;
//            (depthwise-only form: the expanded tensor is loaded and converted ONCE instead)
//   phase 1b circular padding: the wrapped columns are copied inside LDS;
//   phase 2  depthwise: thread = (4 channels, NOUT adjacent output columns of one row), window slid through registers,
//            unmasked 16-byte reads at a conflict-free pixel pitch (tools/lds_layout.py model, chosen per shape on the
//            host: mbp_row_pitch), BN1 + swish, stores, squeeze sums by a fixed butterfly.
// One squeeze-partial row per (sample, band): deterministic.
#include "common.h"
#include "mbconv_plane.h"

// diagnostics builds only (tools/gpu/ablate_mbplane.sh): bit 0 no y stores, 1 no output swish, 2 no depthwise FMAs, 3 no window
// reads, 4 no expand swish, 5 no x loads, 6 no expand phase at all
#ifndef MBP_ABL
#define MBP_ABL 0
#endif

namespace ccvpe {

int num_cus();   // narrow_bf16.hip

int g_mbplane_mode = 7;

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MbPlaneParams {
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
  int BH, nbands, nchunks, PWp, total_blocks;
  unsigned w_magic, xg_magic;      // q / W = (q * w_magic) >> 20, it / XG likewise (q, it < 4096)
};

template <int CH> struct MbpPitch { static constexpr int V = CH == 16 ? 20 : (CH == 32 ? 40 : 64); };

template <typename TE, int K, int S, int CH, int NKK, bool EXPAND>
__global__ __launch_bounds__(256, 2) void mbconv_plane_kernel(const MbPlaneParams p) {
  constexpr int E = 16 / (int)sizeof(TE);            // elements per 16-byte piece
  constexpr int SK = 4 * E;                          // input channels per 64-byte K piece
  constexpr int PB = (S == 1) ? (K - 1) / 2 : (K - 2) / 2;
  constexpr int TP = (S == 1) ? (K - 1) : (K - 2);   // total padding per axis (224-schedule SAME)
  constexpr int NOUT = (S == 1) ? 4 : 2;             // adjacent outputs per depthwise thread (window step 4 pixels either way)
  constexpr int NCOL = (NOUT - 1) * S + K;
  constexpr int CG = CH / 4;
  constexpr int NTL = CH / 16;
  constexpr int PITCH = MbpPitch<CH>::V;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int IH = (p.BH - 1) * S + K;
  float* plane = sm;                                 // [IH][PWp][PITCH]
  float* wl = sm + IH * p.PWp * PITCH;               // [K*K][CH]
  float* red = wl + K * K * CH;                      // [4][CG][4]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int lb;
  {
    const int q = p.total_blocks / 8, r = p.total_blocks % 8;
    const int xcd = blockIdx.x % 8, loc = blockIdx.x / 8;
    lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int chunk = lb % p.nchunks;                  // the channel slices of one (sample, band) are neighbours: they share x
  lb /= p.nchunks;
  const int band = lb % p.nbands;
  const int b = lb / p.nbands;
  const int c0 = chunk * CH;
  const int oy0 = band * p.BH;
  const int bh = min(p.BH, p.Ho - oy0);
  const int IHb = (bh - 1) * S + K;
  const int iy0 = oy0 * S - PB;
  const int r_lo = max(0, -iy0), r_hi = min(IHb, p.H - iy0);
  const int PW = p.W + TP;

  // ---- phase 0: zeros of the padded plane; depthwise weights ----------------------------------------------------------------
  {
    const int j = lane;                              // PW <= 64 (planes of <= 1024 pixels are checked on the host)
    if (j < PW) {
      const bool colpad = !p.circular && (j < PB || j >= PB + p.W);
      for (int r = wave; r < IHb; r += 4) {
        if (colpad || r < r_lo || r >= r_hi) {
          float* d = plane + (r * p.PWp + j) * PITCH;
#pragma unroll
          for (int g = 0; g < CG; ++g) *reinterpret_cast<f32x4*>(d + 4 * g) = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    for (int i = tid; i < K * K * CH; i += 256) wl[i] = p.w_dw[(size_t)(i / CH) * p.mid + c0 + (i % CH)];
  }

  const int npx = (r_hi - r_lo) * p.W;               // real pixels of this band: a contiguous range of the sample
  const size_t gp0 = (size_t)b * p.H * p.W + (size_t)(iy0 + r_lo) * p.W;

  if (EXPAND) {
    // ---- phase 1: expand -----------------------------------------------------------------------------------------------------
    const int px = lane & 15;
    const int kq = (lane >> 4) * E;                  // first channel of this lane's k group inside a 64-byte piece
    const int q4 = (lane >> 4) * 4;                  // D rows (channels) of this lane
    const TE* xb = reinterpret_cast<const TE*>(p.x) + gp0 * p.Cin;
    f32x4 wf[NTL][NKK], sc0[NTL], sh0[NTL];
#pragma unroll
    for (int n = 0; n < NTL; ++n) {
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk)
        wf[n][kk] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const TE*>(p.w_exp) + (size_t)(c0 + 16 * n + px) * p.kpad + kk * SK + kq);
      sc0[n] = *reinterpret_cast<const f32x4*>(p.s0 + c0 + 16 * n + q4);
      sh0[n] = *reinterpret_cast<const f32x4*>(p.b0 + c0 + 16 * n + q4);
    }
    unsigned koff[NKK];                              // byte offset of the lane's piece kk (pieces beyond Cin re-read channel 0..:
#pragma unroll                                       // they meet the zero padding of the packed weights)
    for (int kk = 0; kk < NKK; ++kk) {
      const int ch = kk * SK + kq;
      koff[kk] = (unsigned)(ch < p.Cin ? ch : 0) * (unsigned)sizeof(TE);
    }
    const unsigned rowb = (unsigned)p.Cin * (unsigned)sizeof(TE);
    const int ntile = (npx + 15) >> 4;
    auto load = [&](f32x4 (&f)[NKK], int t) {
      const int q = min(16 * t + px, npx - 1);       // clamped: a valid address, the result is not stored
      const char* src = reinterpret_cast<const char*>(xb) + (size_t)q * rowb;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        if (MBP_ABL & 32) f[kk] = (f32x4){1.f, 2.f, 3.f, (float)q};
        else f[kk] = *reinterpret_cast<const f32x4*>(src + koff[kk]);
      }
    };
    auto compute = [&](const f32x4 (&f)[NKK], int t) {
      const int q = 16 * t + px;
      const int r = (int)(((unsigned)q * p.w_magic) >> 20);
      const int c = q - r * p.W;
      float* dst = plane + ((r_lo + r) * p.PWp + PB + c) * PITCH + q4;
#pragma unroll
      for (int n = 0; n < NTL; ++n) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
          if (sizeof(TE) == 4) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n][kk][rr], f[kk][rr], acc, 0, 0, 0);
          } else {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cc_bf16x8, wf[n][kk]), __builtin_bit_cast(cc_bf16x8, f[kk]), acc, 0, 0, 0);
          }
        }
        f32x4 o;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) o[rr] = (MBP_ABL & 16) ? acc[rr] * sc0[n][rr] + sh0[n][rr] : swishf(acc[rr] * sc0[n][rr] + sh0[n][rr]);
        if (q < npx) *reinterpret_cast<f32x4*>(dst + 16 * n) = o;
      }
    };
    // two fragment sets: the loads of the next tile are in flight under the matrix + swish work of this one
    f32x4 fa[NKK], fb[NKK];
    int t = (MBP_ABL & 64) ? ntile : wave;
    if (t < ntile) load(fa, t);
    for (; t < ntile; t += 8) {
      load(fb, min(t + 4, ntile - 1));
      compute(fa, t);
      load(fa, min(t + 8, ntile - 1));
      if (t + 4 < ntile) compute(fb, t + 4);
    }
  } else {
    // ---- phase 1 (depthwise-only form): the band's slice of the expanded tensor, converted once -----------------------------
    constexpr int PP = CH / E;                       // 16-byte pieces per pixel
    const TE* xb = reinterpret_cast<const TE*>(p.x) + gp0 * p.mid + c0;
    const int n = npx * PP;
    for (int i0 = tid; i0 < n; i0 += 4 * 256) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = min(i0 + 256 * u, n - 1);
        v[u] = *reinterpret_cast<const f32x4*>(xb + (size_t)(i / PP) * p.mid + (i % PP) * E);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 256 * u;
        if (i < n) {
          const int q = i / PP, pc = i % PP;
          const int r = (int)(((unsigned)q * p.w_magic) >> 20);
          const int c = q - r * p.W;
          float* dst = plane + ((r_lo + r) * p.PWp + PB + c) * PITCH + pc * E;
          if (sizeof(TE) == 4) {
            *reinterpret_cast<f32x4*>(dst) = v[u];
          } else {
            const cc_bf16x8 h = __builtin_bit_cast(cc_bf16x8, v[u]);
            *reinterpret_cast<f32x4*>(dst) = (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
            *reinterpret_cast<f32x4*>(dst + 4) = (f32x4){(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
          }
        }
      }
    }
  }
  __syncthreads();
  if (p.circular) {
    // ---- phase 1b: wrapped columns (plane column j holds image column j - PB) ---------------------------------------------------
    const int nrow = r_hi - r_lo;
    for (int i = tid; i < nrow * TP * CG; i += 256) {
      const int g = i % CG, jr = i / CG;
      const int jp = jr % TP, r = r_lo + jr / TP;
      const int j = jp < PB ? jp : p.W + jp;         // pad column
      const int js = jp < PB ? j + p.W : j - p.W;    // its source
      *reinterpret_cast<f32x4*>(plane + (r * p.PWp + j) * PITCH + 4 * g) =
          *reinterpret_cast<const f32x4*>(plane + (r * p.PWp + js) * PITCH + 4 * g);
    }
    __syncthreads();
  }

  // ---- phase 2: depthwise ----------------------------------------------------------------------------------------------------
  const int cg = tid % CG;
  const f32x4 sc1 = *reinterpret_cast<const f32x4*>(p.s1 + c0 + 4 * cg);
  const f32x4 sh1 = *reinterpret_cast<const f32x4*>(p.b1 + c0 + 4 * cg);
  const int XG = (p.Wo + NOUT - 1) / NOUT;
  const int nitem = bh * XG;
  TE* yb = reinterpret_cast<TE*>(p.y) + ((size_t)b * p.Ho + oy0) * p.Wo * p.mid + c0 + 4 * cg;
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  // depthwise weights in registers for the whole workgroup (LDS returns in order: a weight read behind the next row's window reads,
  // needed by the very next FMA, made every kernel row wait for the prefetch it was meant to overlap with — see mbconv_band_kernel)
  f32x4 wr[K * K];
#pragma unroll
  for (int q = 0; q < K * K; ++q) wr[q] = *reinterpret_cast<const f32x4*>(wl + q * CH + 4 * cg);
  for (int it = tid / CG; it < nitem; it += 256 / CG) {
    const int oyl = (int)(((unsigned)it * p.xg_magic) >> 20);
    const int ox0 = (it - oyl * XG) * NOUT;
    const float* trow = plane + ((oyl * S) * p.PWp + ox0 * S) * PITCH + 4 * cg;
    f32x4 acc[NOUT];
#pragma unroll
    for (int t = 0; t < NOUT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // one kernel row of window reads ahead of the row being accumulated (two register sets; the fence keeps the
    // scheduler from hoisting all K rows: 160 registers for k = 5)
    f32x4 col[2][NCOL];
#pragma unroll
    for (int j = 0; j < NCOL; ++j) col[0][j] = (MBP_ABL & 8) ? (f32x4){1.f, 2.f, 3.f, (float)it} : *reinterpret_cast<const f32x4*>(trow + j * PITCH);
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      if (ky + 1 < K) {
#pragma unroll
        for (int j = 0; j < NCOL; ++j)
          col[(ky + 1) & 1][j] = (MBP_ABL & 8) ? (f32x4){1.f, 2.f, 3.f, (float)(it + ky)} : *reinterpret_cast<const f32x4*>(trow + ((ky + 1) * p.PWp + j) * PITCH);
      }
#pragma unroll
      for (int kx = 0; kx < ((MBP_ABL & 4) ? 1 : K); ++kx) {
        const f32x4 wv = wr[ky * K + kx];
#pragma unroll
        for (int t = 0; t < NOUT; ++t) acc[t] += col[ky & 1][t * S + kx] * wv;
      }
      if (MBP_ABL & 4) {
#pragma unroll
        for (int j = 0; j < NCOL; ++j) acc[j % NOUT] += col[ky & 1][j];   // keeps every window read alive
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    TE* yrow = yb + (size_t)oyl * p.Wo * p.mid;
#pragma unroll
    for (int t = 0; t < NOUT; ++t) {
      f32x4 o = acc[t] * sc1 + sh1;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (MBP_ABL & 2) ? o[r] : swishf(o[r]);
      if (!(MBP_ABL & 1)) st4<TE>(yrow + (size_t)(ox0 + t) * p.mid, o);   // Wo % NOUT == 0 (host): no per-output test — a branch here makes
      sum += o;                                       // the compiler sink each output's FMAs behind ALL the window reads
    }
  }
  // ---- squeeze sums of this (sample, band, slice): butterfly over the lanes that share cg, then the four waves -----------------
#pragma unroll
  for (int o = CG; o < 64; o <<= 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) sum[r] += __shfl_xor(sum[r], o, 64);
  }
  if (lane < CG) *reinterpret_cast<f32x4*>(red + (wave * CG + lane) * 4) = sum;
  __syncthreads();
  if (tid < CG) {
    f32x4 t = *reinterpret_cast<const f32x4*>(red + tid * 4);
#pragma unroll
    for (int w = 1; w < 4; ++w) t += *reinterpret_cast<const f32x4*>(red + (w * CG + tid) * 4);
    *reinterpret_cast<f32x4*>(p.se_partial + ((size_t)b * p.nbands + band) * p.mid + c0 + 4 * tid) = t;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Band-owner form (bf16 storage, fused expand + depthwise): what the ablation of the kernel above showed (tools/gpu/
// ablate_mbplane.sh, block 9 of the aerial encoder, 102.8 us): with EVERYTHING removed 30.7 us remain (5 376 workgroups, each
// starting with a dependent weight load in front of its first barrier: 1.4 us of latency per workgroup generation), the expand
// phase costs 37 us although its swish costs nothing (ten x tiles per wave fetched from L2 one tile ahead: latency), the window
// reads 18, the FMAs 9, the stores 8.  So:
//   * a workgroup (512 threads) owns (sample, band, a RANGE of channel slices) and loops over the slices: launch, zero fill and
//     the x fetch are paid once;
//   * waves 0-3 are PRODUCERS: they keep the band's x fragments in REGISTERS for the whole kernel (x is the same for every
//     slice: up to 10 tiles x 4 pieces = 160 VGPRs) and per slice run MFMA + BN0 + swish -> ds_write into plane[slice & 1];
//     waves 4-7 are CONSUMERS: they run the depthwise phase of the PREVIOUS slice from plane[(slice - 1) & 1].  One barrier per
//     slice; a SIMD always holds one wave of each kind, so the producers' matrix / LDS-write phases and the consumers' window
//     reads + FMAs overlap by construction instead of by luck;
//   * no global load sits in front of anything: the consumers (who have registers to spare) fetch the NEXT slice's expand
//     weights, BN vectors and depthwise weights during their depthwise phase and park them in LDS (expand weights / BN0 double
//     buffered, depthwise weights / BN1 triple buffered: a consumer wave may still read slice s - 1's while another stores
//     slice s + 1's);
//   * circular padding: a producer lane whose pixel sits within the wrap distance of a border stores it twice.
// CH = 16 for every plane size here (a step is half as long on the 16 x 16 planes, but both planes + parameters stay < 90 KB).
// ---------------------------------------------------------------------------------------------------------------------------------
struct MbBandParams {
  const cc_bf16* x;
  const cc_bf16* w_exp;
  const float* s0;
  const float* b0;
  const float* w_dw;
  const float* s1;
  const float* b1;
  cc_bf16* y;
  float* se_partial;
  int H, W, Cin, kpad, mid, Ho, Wo, circular;
  int BH, nbands, nchunks, PWp, total_blocks, ngrp, cpg;
  unsigned w_magic, xg_magic;
};

template <int K, int NKK> struct MbBandLds {
  static constexpr int CH = 16;
  static constexpr int WPITCH = 16 * NKK + 4;        // dwords per expand-weight row (64 NKK bytes + 16: rows on different banks)
  static constexpr int WEXP = CH * WPITCH;           // dwords per buffer
  static constexpr int WL = K * K * CH + 2 * CH;     // depthwise weights | BN1 scale | BN1 shift
  static constexpr int PARAM_FLOATS = 2 * WEXP + 2 * 2 * CH + 3 * WL + 2 * 4 * 4 * 4;
};

template <int K, int S, int NKK, int TPW, int RY>
__global__ __launch_bounds__(512) void mbconv_band_kernel(const MbBandParams p) {
  constexpr int CH = 16, CG = 4, PITCH = 20, SK = 32, E = 8;
  constexpr int PB = (S == 1) ? (K - 1) / 2 : (K - 2) / 2;
  constexpr int TP = (S == 1) ? (K - 1) : (K - 2);
  constexpr int NOUT = (S == 1) ? 4 : 2;
  constexpr int NCOL = (NOUT - 1) * S + K;
  using L = MbBandLds<K, NKK>;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int IH = (p.BH - 1) * S + K;
  const int plane_floats = IH * p.PWp * PITCH;
  float* plane0 = sm;                                // [2][IH][PWp][PITCH]
  float* wexp = sm + 2 * plane_floats;               // [2][CH][WPITCH] (bf16 rows)
  float* bn0 = wexp + 2 * L::WEXP;                   // [2][scale CH | shift CH]
  float* wlb = bn0 + 2 * 2 * CH;                     // [3][K*K*CH | s1 CH | b1 CH]
  float* red = wlb + 3 * L::WL;                      // [2][4 waves][CG][4]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int lb;
  {
    const int q = p.total_blocks / 8, r = p.total_blocks % 8;
    const int xcd = blockIdx.x % 8, loc = blockIdx.x / 8;
    lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int grp = lb % p.ngrp;
  lb /= p.ngrp;
  const int band = lb % p.nbands;
  const int b = lb / p.nbands;
  const int chunk_begin = grp * p.cpg;
  const int nch = min(p.nchunks, chunk_begin + p.cpg) - chunk_begin;
  const int oy0 = band * p.BH;
  const int bh = min(p.BH, p.Ho - oy0);
  const int IHb = (bh - 1) * S + K;
  const int iy0 = oy0 * S - PB;
  const int r_lo = max(0, -iy0), r_hi = min(IHb, p.H - iy0);
  const int PW = p.W + TP;
  const int npx = (r_hi - r_lo) * p.W;
  const size_t gp0 = (size_t)b * p.H * p.W + (size_t)(iy0 + r_lo) * p.W;

  // ---- zeros of BOTH planes (the same band for every slice of this workgroup: written once) ---------------------------------------
  for (int idx = tid; idx < IHb * 64; idx += 512) {
    const int r = idx >> 6, j = idx & 63;
    if (j < PW && ((!p.circular && (j < PB || j >= PB + p.W)) || r < r_lo || r >= r_hi)) {
      float* d = plane0 + (r * p.PWp + j) * PITCH;
#pragma unroll
      for (int g = 0; g < CG; ++g) {
        *reinterpret_cast<f32x4*>(d + 4 * g) = (f32x4){0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(d + plane_floats + 4 * g) = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
  }

  if (wave < 4) {
    // =============================== producers ===============================================================================
    const int px = lane & 15;
    const int kq = (lane >> 4) * E;
    const int q4 = (lane >> 4) * 4;
    const cc_bf16* xb = p.x + gp0 * p.Cin;
    const int ntile = (npx + 15) >> 4;
    f32x4 xr[TPW][NKK];
    int dst[TPW], alt[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int t = wave + 4 * i;
      const int qq = 16 * t + px;
      const int q = min(qq, npx - 1);
      const cc_bf16* src = xb + (size_t)q * p.Cin;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        const int ch = kk * SK + kq;
        xr[i][kk] = *reinterpret_cast<const f32x4*>(src + (ch < p.Cin ? ch : 0));
      }
      const bool valid = qq < npx;
      const int r = (int)(((unsigned)q * p.w_magic) >> 20);
      const int c = q - r * p.W;
      const int rowb = (r_lo + r) * p.PWp;
      dst[i] = valid ? (rowb + PB + c) * PITCH + q4 : -1;
      int ac = -1;
      if (p.circular) ac = c < TP - PB ? PB + p.W + c : (c >= p.W - PB ? c - (p.W - PB) : -1);
      alt[i] = (valid && ac >= 0) ? (rowb + ac) * PITCH + q4 : -1;
    }
    __syncthreads();                                 // slice 0's expand weights are in LDS (consumers' prologue)
    for (int s = 0; s <= nch; ++s) {
      if (s < nch) {
        const float* we = wexp + (s & 1) * L::WEXP;
        f32x4 wf[NKK];
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) wf[kk] = *reinterpret_cast<const f32x4*>(we + px * L::WPITCH + kk * 16 + (lane >> 4) * 4);
        const f32x4 sc0 = *reinterpret_cast<const f32x4*>(bn0 + (s & 1) * 2 * CH + q4);
        const f32x4 sh0 = *reinterpret_cast<const f32x4*>(bn0 + (s & 1) * 2 * CH + CH + q4);
        float* pl = plane0 + (s & 1) * plane_floats;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
          if (wave + 4 * i < ntile && !(MBP_ABL & 64)) {   // (computing the tiles past the last one unconditionally measured 3 % slower)
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk)
              acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cc_bf16x8, wf[kk]), __builtin_bit_cast(cc_bf16x8, xr[i][kk]), acc, 0, 0, 0);
            f32x4 o;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) o[rr] = (MBP_ABL & 16) ? acc[rr] * sc0[rr] + sh0[rr] : swishf(acc[rr] * sc0[rr] + sh0[rr]);
            if (dst[i] >= 0) *reinterpret_cast<f32x4*>(pl + dst[i]) = o;
            if (alt[i] >= 0) *reinterpret_cast<f32x4*>(pl + alt[i]) = o;
          }
        }
      }
      __syncthreads();
    }
  } else {
    // =============================== consumers ===============================================================================
    const int ct = tid - 256, cwave = wave - 4;
    const int cg = ct % CG;
    const int XG = p.Wo / NOUT;
    const int nitem = (bh / RY) * XG;               // RY = 2: the host guarantees an even band height
    cc_bf16* yb = p.y + ((size_t)b * p.Ho + oy0) * p.Wo * p.mid + 4 * cg;
    float* sp = p.se_partial + ((size_t)b * p.nbands + band) * p.mid;
    // the parameters of one slice as this thread's share: NKK / 4 .. pieces of the expand weights, one piece of the depthwise
    // weights (threads < K*K*4), BN0 / BN1 (threads < 8 / the next 8)
    constexpr int WPT = (CH * NKK * 4 + 255) / 256;  // 16-byte pieces of w_exp per thread (a row = NKK * 4 pieces)
    f32x4 pw[WPT], pd, pb;
    auto fetch = [&](int cl) {                       // cl: slice index inside this workgroup's range (clamped)
      const int c0 = (chunk_begin + min(cl, nch - 1)) * CH;
#pragma unroll
      for (int u = 0; u < WPT; ++u) {
        const int pc = min(ct + 256 * u, CH * NKK * 4 - 1);
        pw[u] = *reinterpret_cast<const f32x4*>(p.w_exp + (size_t)(c0 + pc / (NKK * 4)) * p.kpad + (pc % (NKK * 4)) * E);
      }
      const int wi = ct < K * K * CG ? ct : 0;
      pd = *reinterpret_cast<const f32x4*>(p.w_dw + (size_t)(wi / CG) * p.mid + c0 + (wi % CG) * 4);
      const int bi = ct & 15;                        // 0-3 s0, 4-7 b0, 8-11 s1, 12-15 b1
      const float* bsrc = bi < 4 ? p.s0 : (bi < 8 ? p.b0 : (bi < 12 ? p.s1 : p.b1));
      pb = *reinterpret_cast<const f32x4*>(bsrc + c0 + (bi & 3) * 4);
    };
    auto park = [&](int cl) {                        // into the buffers slice cl will be read from
      float* we = wexp + (cl & 1) * L::WEXP;
#pragma unroll
      for (int u = 0; u < WPT; ++u) {
        const int pc = ct + 256 * u;
        if (pc < CH * NKK * 4) *reinterpret_cast<f32x4*>(we + (pc / (NKK * 4)) * L::WPITCH + (pc % (NKK * 4)) * 4) = pw[u];
      }
      float* wl = wlb + (cl % 3) * L::WL;
      if (ct < K * K * CG) *reinterpret_cast<f32x4*>(wl + ct * 4) = pd;
      if (ct < 16) {
        const int bi = ct;
        float* bd = bi < 8 ? bn0 + (cl & 1) * 2 * CH + (bi >> 2) * CH + (bi & 3) * 4 : wl + K * K * CH + ((bi - 8) >> 2) * CH + (bi & 3) * 4;
        *reinterpret_cast<f32x4*>(bd) = pb;
      }
    };
    fetch(0);
    park(0);
    fetch(1);
    __syncthreads();
    for (int s = 0; s <= nch; ++s) {
      if (s >= 1) {
        const int cl = s - 1;
        const int c0 = (chunk_begin + cl) * CH;
        if (s >= 2 && ct < CG) {                     // squeeze sums of slice s - 2 (its four wave rows were published at the last barrier)
          const float* rd = red + ((s - 2) & 1) * 64;
          f32x4 t = *reinterpret_cast<const f32x4*>(rd + ct * 4);
#pragma unroll
          for (int w = 1; w < 4; ++w) t += *reinterpret_cast<const f32x4*>(rd + (w * CG + ct) * 4);
          *reinterpret_cast<f32x4*>(sp + (chunk_begin + s - 2) * CH + 4 * ct) = t;
        }
        const float* pl = plane0 + (cl & 1) * plane_floats;
        const float* wl = wlb + (cl % 3) * L::WL;
        const f32x4 sc1 = *reinterpret_cast<const f32x4*>(wl + K * K * CH + 4 * cg);
        const f32x4 sh1 = *reinterpret_cast<const f32x4*>(wl + K * K * CH + CH + 4 * cg);
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        // an item = RY adjacent output ROWS x NOUT adjacent columns x 4 channels: the (RY - 1) S + K window rows are read once
        // and feed every output row they belong to (RY = 2, k = 5: 6 x 8 window reads for 8 outputs instead of 2 x 5 x 8 — the
        // window reads were the largest single cost of this kernel: tools/gpu/ablate_mbplane.sh, 38 of 100 us on block 9)
        constexpr int NR = (RY - 1) * S + K;
        f32x4 wr[K * K];
#pragma unroll
        for (int q = 0; q < K * K; ++q) wr[q] = *reinterpret_cast<const f32x4*>(wl + q * CH + 4 * cg);
        for (int it = ct / CG; it < nitem; it += 256 / CG) {
          const int oyl = RY * (int)(((unsigned)it * p.xg_magic) >> 20);
          const int ox0 = (it - (oyl / RY) * XG) * NOUT;
          const float* trow = pl + ((oyl * S) * p.PWp + ox0 * S) * PITCH + 4 * cg;
          f32x4 acc[RY][NOUT];
#pragma unroll
          for (int r = 0; r < RY; ++r)
#pragma unroll
            for (int t = 0; t < NOUT; ++t) acc[r][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
          // window row iy + 1 is requested before row iy is accumulated (two register sets).  The depthwise weights are NOT read
          // inside this loop: LDS returns in order, so a weight read issued behind the next row's window reads — and needed by the
          // very next FMA — made every row wait for the whole prefetch it was meant to overlap with (4 200 cycles per item for
          // ~1 500 cycles of issue); they sit in registers for the whole slice (wr[], loaded in front of the item loop)
          f32x4 col[2][NCOL];
#pragma unroll
          for (int j = 0; j < NCOL; ++j)
            col[0][j] = (MBP_ABL & 8) ? (f32x4){1.f, 2.f, 3.f, (float)it} : *reinterpret_cast<const f32x4*>(trow + j * PITCH);
#pragma unroll
          for (int iy = 0; iy < NR; ++iy) {
            if (iy + 1 < NR) {
#pragma unroll
              for (int j = 0; j < NCOL; ++j)
                col[(iy + 1) & 1][j] = (MBP_ABL & 8) ? (f32x4){1.f, 2.f, 3.f, (float)(it + iy)} : *reinterpret_cast<const f32x4*>(trow + ((iy + 1) * p.PWp + j) * PITCH);
            }
#pragma unroll
            for (int r = 0; r < RY; ++r) {
              const int ky = iy - r * S;               // this window row is kernel row ky of output row r
              if (ky >= 0 && ky < K) {
#pragma unroll
                for (int kx = 0; kx < ((MBP_ABL & 4) ? 1 : K); ++kx) {
                  const f32x4 wv = wr[ky * K + kx];
#pragma unroll
                  for (int t = 0; t < NOUT; ++t) acc[r][t] += col[iy & 1][t * S + kx] * wv;
                }
              }
            }
            if (MBP_ABL & 4) {
#pragma unroll
              for (int j = 0; j < NCOL; ++j) acc[0][j % NOUT] += col[iy & 1][j];
            }
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int r = 0; r < RY; ++r) {
            cc_bf16* yrow = yb + (size_t)(oyl + r) * p.Wo * p.mid + c0;
#pragma unroll
            for (int t = 0; t < NOUT; ++t) {
              f32x4 o = acc[r][t] * sc1 + sh1;
#pragma unroll
              for (int q = 0; q < 4; ++q) o[q] = (MBP_ABL & 2) ? o[q] : swishf(o[q]);
              if (!(MBP_ABL & 1)) st4<cc_bf16>(yrow + (size_t)(ox0 + t) * p.mid, o);
              sum += o;
            }
          }
        }
#pragma unroll
        for (int o = CG; o < 64; o <<= 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) sum[r] += __shfl_xor(sum[r], o, 64);
        }
        if (lane < CG) *reinterpret_cast<f32x4*>(red + (cl & 1) * 64 + (cwave * CG + lane) * 4) = sum;
      }
      if (s + 1 < nch) {                             // slice s + 1: read by the producers at step s + 1, by us at step s + 2
        park(s + 1);
        fetch(s + 2);
      }
      __syncthreads();
    }
    if (ct < CG) {                                   // the last slice's squeeze sums (published at the final barrier)
      const int cl = nch - 1;
      const float* rd = red + (cl & 1) * 64;
      f32x4 t = *reinterpret_cast<const f32x4*>(rd + ct * 4);
#pragma unroll
      for (int w = 1; w < 4; ++w) t += *reinterpret_cast<const f32x4*>(rd + (w * CG + ct) * 4);
      *reinterpret_cast<f32x4*>(sp + (chunk_begin + cl) * CH + 4 * ct) = t;
    }
  }
}

// ---- host: geometry ------------------------------------------------------------------------------------------------------------
// ds_read_b128 lane groups and 64 four-byte banks (MI355X_MICROARCH.md, LDS section; tools/lds_layout.py is the same model)
static int mbp_read_cycles(int CG, int pitch, int PWp, int XG, int S, int nitem) {
  static const int grp[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                 {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                 {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                 {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
  int total = 0;
  for (int wave = 0; wave < 4; ++wave)
    for (int g = 0; g < 4; ++g) {
      int addr_of_bank[64][16], cnt[64];
      for (int i = 0; i < 64; ++i) cnt[i] = 0;
      int worst = 0;
      for (int li = 0; li < 16; ++li) {
        const int tid = 64 * wave + grp[g][li];
        const int cg = tid % CG, it = tid / CG;
        if (it >= nitem) continue;
        const int row = it / XG, xg = it % XG;
        const int a = ((row * S) * PWp + xg * 4) * pitch + 4 * cg;   // window step = NOUT * S = 4 pixels
        for (int d = 0; d < 4; ++d) {
          const int bank = (a + d) % 64;
          bool seen = false;
          for (int k = 0; k < cnt[bank]; ++k) seen |= addr_of_bank[bank][k] == a + d;
          if (!seen) addr_of_bank[bank][cnt[bank]++] = a + d;
          if (cnt[bank] > worst) worst = cnt[bank];
        }
      }
      total += worst;
    }
  return total;
}

int g_mbp_plane_kb = 72;                      // LDS budget of the plane: two workgroups per CU (tuning: mode >> 8)

struct MbpGeom {
  int CH, BH, nbands, PWp, lds_bytes;
};

constexpr int MBP_LDS_MAX = 80 * 1024;

static bool mbp_geometry_compute(int H, int W, int mid, int k, int stride, MbpGeom* g);

// the row-pitch search runs the bank model ~16 x 256 lane evaluations: microseconds of host time per call, and this sits on the
// launch path of ~40 back-to-back kernels of 20-60 us — remembered per shape (thread-local: no lock)
static bool mbp_geometry(int H, int W, int mid, int k, int stride, MbpGeom* g) {
  struct Entry { int key[6]; bool ok; MbpGeom g; };
  static thread_local Entry cache[32];
  static thread_local int used = 0, next = 0;
  const int key[6] = {H, W, mid, k, stride, g_mbp_plane_kb};
  for (int i = 0; i < used; ++i) {
    const Entry& e = cache[i];
    if (e.key[0] == key[0] && e.key[1] == key[1] && e.key[2] == key[2] && e.key[3] == key[3] && e.key[4] == key[4] && e.key[5] == key[5]) {
      *g = e.g;
      return e.ok;
    }
  }
  Entry& e = cache[next];
  for (int i = 0; i < 6; ++i) e.key[i] = key[i];
  e.g = MbpGeom{0, 0, 0, 0, 0};
  e.ok = mbp_geometry_compute(H, W, mid, k, stride, &e.g);
  *g = e.g;
  next = (next + 1) % 32;
  if (used < 32) ++used;
  return e.ok;
}

static bool mbp_geometry_compute(int H, int W, int mid, int k, int stride, MbpGeom* g) {
  if (!(k == 3 || k == 5) || !(stride == 1 || stride == 2)) return false;
  if (H * W > 1024 || W + 4 > 64 || H < 5 || W < 5 || mid % 16) return false;
  const int tp = stride == 1 ? k - 1 : k - 2;
  const int Ho = (H + tp - k) / stride + 1, Wo = (W + tp - k) / stride + 1;
  const int ch = (H * W <= 256 && mid % 32 == 0) ? 32 : 16;   // by the INPUT plane: it is what sits in LDS
  const int pitch = ch == 16 ? 20 : 40;
  const int nout = stride == 1 ? 4 : 2;
  if (Wo % nout) return false;
  const int XG = Wo / nout;
  // bands are chosen for k = 5 whatever k is (ccvpe_dwconv_nblk does not know k), the row pitch for the real k
  const int tp5 = stride == 1 ? 4 : 3;
  for (int nb = 1; nb <= Ho; ++nb) {
    const int BH = (Ho + nb - 1) / nb;
    if ((Ho + BH - 1) / BH != nb) continue;
    const int IH5 = (BH - 1) * stride + 5;
    if ((size_t)IH5 * (W + tp5) * pitch * 4 > (size_t)g_mbp_plane_kb * 1024) continue;
    const int IH = (BH - 1) * stride + k, PW = W + tp;
    int best = -1, bestc = 1 << 30;
    for (int pwp = PW; pwp < PW + 16; ++pwp) {
      if ((size_t)IH * pwp * pitch * 4 > (size_t)g_mbp_plane_kb * 1024 + 4096) break;
      const int c = mbp_read_cycles(ch / 4, pitch, pwp, XG, stride, BH * XG);
      if (c < bestc) { bestc = c; best = pwp; }
    }
    if (best < 0) continue;
    if ((IH * best * pitch + k * k * ch + 4 * (ch / 4) * 4) * 4 > MBP_LDS_MAX) continue;
    g->CH = ch; g->BH = BH; g->nbands = nb; g->PWp = best;
    g->lds_bytes = (IH * best * pitch + k * k * ch + 4 * (ch / 4) * 4) * 4;
    return true;
  }
  return false;
}

// input-channel counts whose K-piece counts are instantiated for BOTH storage types (80 / 112 / 192: EfficientNet-B0's late blocks)
static bool mbp_cin_ok(int cin) { return cin == 80 || cin == 112 || cin == 192; }

int mbplane_nblk(int H, int W, int cin, int mid, int k, int stride) {
  MbpGeom g;
  if (cin > 0 && !mbp_cin_ok(cin)) return 0;
  if (!mbp_geometry(H, W, mid, k, stride, &g)) return 0;
  if (stride == 2 && g.CH == 32) return 0;           // not instantiated
  return g.nbands;
}

// the band-owner kernel (bf16, fused form): same bands as the geometry above (the squeeze-partial rows must not depend on the
// kernel), its own CH = 16 row pitch, chunk groups so that the launch is about one workgroup per CU
static int mbband_launch(const MbpGeom& g, const void* x, const void* w_exp, int kpad, const float* s0, const float* b0,
                         const float* w_dw, const float* s1, const float* b1, void* y, float* se_partial, int B, int H, int W,
                         int cin, int mid, int k, int stride, int circular, void* stream) {
  const int tp = stride == 1 ? k - 1 : k - 2;
  const int Ho = (H + tp - k) / stride + 1, Wo = (W + tp - k) / stride + 1;
  const int nout = stride == 1 ? 4 : 2;
  const int XG = Wo / nout;
  const int IH = (g.BH - 1) * stride + k, PW = W + tp;
  const int rows = IH < H ? IH : H;
  const int ntile = (rows * W + 15) / 16;
  const int tpw = (ntile + 3) / 4;
  const int nkk = kpad / 32;
  if (k == 3 && stride == 2) return -1000;
  if (!((tpw <= 10 && (nkk == 3 || nkk == 4)) || (tpw <= 4 && nkk == 6))) return -1000;
  struct PitchKey { int k[5]; int pwp; };
  static thread_local PitchKey pk[16];
  static thread_local int pk_used = 0, pk_next = 0;
  // two output rows per depthwise thread when every band has an even height and the plane is large enough to keep the 256
  // consumer threads busy (RY x fewer items)
  const int ry = (tpw > 4 && g.BH % 2 == 0 && Ho % g.BH == 0 && (g.BH / 2) * XG * 4 >= 256) ? 2 : 1;
  int pwp = -1;
  for (int i = 0; i < pk_used; ++i)
    if (pk[i].k[0] == PW && pk[i].k[1] == XG && pk[i].k[2] == stride * ry && pk[i].k[3] == g.BH && pk[i].k[4] == IH) pwp = pk[i].pwp;
  if (pwp < 0) {
    int bestc = 1 << 30;
    for (int c = PW; c < PW + 16; ++c) {
      const int cyc = mbp_read_cycles(4, 20, c, XG, stride * ry, (g.BH / ry) * XG);
      if (cyc < bestc) { bestc = cyc; pwp = c; }
    }
    pk[pk_next] = PitchKey{{PW, XG, stride * ry, g.BH, IH}, pwp};
    pk_next = (pk_next + 1) % 16;
    if (pk_used < 16) ++pk_used;
  }
  const int param_floats = nkk == 3 ? (k == 3 ? MbBandLds<3, 3>::PARAM_FLOATS : MbBandLds<5, 3>::PARAM_FLOATS)
                         : nkk == 4 ? (k == 3 ? MbBandLds<3, 4>::PARAM_FLOATS : MbBandLds<5, 4>::PARAM_FLOATS)
                                    : (k == 3 ? MbBandLds<3, 6>::PARAM_FLOATS : MbBandLds<5, 6>::PARAM_FLOATS);
  const int lds = (2 * IH * pwp * 20 + param_floats) * 4;
  if (lds > 160 * 1024) return -1000;
  MbBandParams p;
  p.x = reinterpret_cast<const cc_bf16*>(x); p.w_exp = reinterpret_cast<const cc_bf16*>(w_exp);
  p.s0 = s0; p.b0 = b0; p.w_dw = w_dw; p.s1 = s1; p.b1 = b1; p.y = reinterpret_cast<cc_bf16*>(y); p.se_partial = se_partial;
  p.H = H; p.W = W; p.Cin = cin; p.kpad = kpad; p.mid = mid; p.Ho = Ho; p.Wo = Wo; p.circular = circular;
  p.BH = g.BH; p.nbands = g.nbands; p.nchunks = mid / 16; p.PWp = pwp;
  // chunk groups: fewest rounds of (one workgroup per CU) x slices per workgroup
  const long base = (long)B * p.nbands;
  const int cus = num_cus();
  int best_g = 1;
  long best_cost = 1L << 60;
  for (int ng = 1; ng <= p.nchunks; ++ng) {
    const int cpg = (p.nchunks + ng - 1) / ng;
    const int ngr = (p.nchunks + cpg - 1) / cpg;
    const long rounds = (base * ngr + cus - 1) / cus;
    const long cost = rounds * (cpg + 2);            // + 2: prologue / drain steps of a workgroup
    if (cost < best_cost) { best_cost = cost; best_g = ngr; }
  }
  p.cpg = (p.nchunks + best_g - 1) / best_g;
  p.ngrp = (p.nchunks + p.cpg - 1) / p.cpg;
  const long total = base * p.ngrp;
  if (total > 0x7fffffffL) return fail(CCVPE_EINVAL, "mbconv_band: grid too large");
  p.total_blocks = (int)total;
  p.w_magic = (unsigned)((1048576 + W - 1) / W);
  p.xg_magic = (unsigned)((1048576 + XG - 1) / XG);
  hipStream_t st = (hipStream_t)stream;
  int rc = CCVPE_OK;
  bool launched = false;
  auto go = [&](auto kern) {
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) { rc = fail(CCVPE_ELAUNCH, "mbconv_band: set smem attr: %s", hipGetErrorString(e)); return; }
      attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.total_blocks), dim3(512), lds, st, p);
    launched = true;
  };
#define MBB(K_, S_, NKK_, TPW_, RY_) if (k == K_ && stride == S_ && nkk == NKK_ && tpw <= TPW_ && ry == RY_ && !launched && !rc) go(mbconv_band_kernel<K_, S_, NKK_, TPW_, RY_>);
  MBB(3, 1, 3, 10, 2) MBB(5, 1, 3, 10, 2) MBB(3, 1, 4, 10, 2) MBB(5, 1, 4, 10, 2) MBB(5, 2, 3, 10, 2) MBB(5, 2, 4, 10, 2)
  MBB(3, 1, 3, 10, 1) MBB(5, 1, 3, 10, 1) MBB(3, 1, 4, 10, 1) MBB(5, 1, 4, 10, 1) MBB(5, 2, 3, 10, 1) MBB(5, 2, 4, 10, 1)
  MBB(3, 1, 6, 4, 1) MBB(5, 1, 6, 4, 1) MBB(5, 2, 6, 4, 1)
#undef MBB
  if (rc) return rc;
  if (!launched) return -1000;
  return check_launch("mbconv_band_kernel");
}

bool mbband_takes(int H, int W, int cin, int mid, int k, int stride, int B) {
  MbpGeom g;
  if (B <= 0 || !mbp_geometry(H, W, mid, k, stride, &g) || mbplane_nblk(H, W, cin, mid, k, stride) <= 0) return false;
  const int tp = stride == 1 ? k - 1 : k - 2;
  const int IH = (g.BH - 1) * stride + k;
  const int rows = IH < H ? IH : H;
  const int tpw = ((rows * W + 15) / 16 + 3) / 4;
  const int nkk = (cin + 31) / 32;
  if (k == 3 && stride == 2) return false;
  if (!((tpw <= 10 && (nkk == 3 || nkk == 4)) || (tpw <= 4 && nkk == 6))) return false;
  (void)tp;
  return true;                                        // (the launcher re-checks the exact LDS size)
}

template <typename TE>
static int mbplane_any(int expand, const void* x, const void* w_exp, int kpad, const float* s0, const float* b0, const float* w_dw,
                       const float* s1, const float* b1, void* y, float* se_partial, int B, int H, int W, int cin, int mid, int k,
                       int stride, int circular, void* stream) {
  constexpr int SK = 4 * (16 / (int)sizeof(TE));
  MbpGeom g;
  if (B <= 0 || !mbp_geometry(H, W, mid, k, stride, &g) || mbplane_nblk(H, W, expand ? cin : 0, mid, k, stride) <= 0)
    return fail(CCVPE_EINVAL, "mbconv_plane: shape not supported (H=%d W=%d cin=%d mid=%d k=%d s=%d)", H, W, cin, mid, k, stride);
  if (expand && (kpad % SK || kpad < cin || kpad != ((cin + SK - 1) / SK) * SK)) return fail(CCVPE_EINVAL, "mbconv_plane: bad kpad");
  if (!aligned16(x) || !aligned16(w_dw) || !aligned16(s1) || !aligned16(b1) || !aligned16(y) || !aligned16(se_partial) ||
      (expand && (!aligned16(w_exp) || !aligned16(s0) || !aligned16(b0))))
    return fail(CCVPE_EINVAL, "mbconv_plane: pointers must be 16-byte aligned");
  if (circular && W < k) return fail(CCVPE_EINVAL, "mbconv_plane: W too small for circular wrap");
  if constexpr (sizeof(TE) == 2) {
    if (expand && (g_mbplane_mode & 4)) {
      const int rc2 = mbband_launch(g, x, w_exp, kpad, s0, b0, w_dw, s1, b1, y, se_partial, B, H, W, cin, mid, k, stride, circular, stream);
      if (rc2 != -1000) return rc2;                  // -1000: the band kernel does not take the shape
    }
  }
  MbPlaneParams p;
  p.x = x; p.w_exp = w_exp; p.s0 = s0; p.b0 = b0; p.w_dw = w_dw; p.s1 = s1; p.b1 = b1; p.y = y; p.se_partial = se_partial;
  p.H = H; p.W = W; p.Cin = cin; p.kpad = kpad; p.mid = mid; p.circular = circular;
  const int tp = stride == 1 ? k - 1 : k - 2;
  p.Ho = (H + tp - k) / stride + 1;
  p.Wo = (W + tp - k) / stride + 1;
  p.BH = g.BH; p.nbands = g.nbands; p.nchunks = mid / g.CH; p.PWp = g.PWp;
  const long total = (long)B * p.nbands * p.nchunks;
  if (total > 0x7fffffffL) return fail(CCVPE_EINVAL, "mbconv_plane: grid too large");
  p.total_blocks = (int)total;
  const int nout = stride == 1 ? 4 : 2;
  const int XG = (p.Wo + nout - 1) / nout;
  p.w_magic = (unsigned)((1048576 + W - 1) / W);
  p.xg_magic = (unsigned)((1048576 + XG - 1) / XG);
  const int nkk = expand ? kpad / SK : 1;
  hipStream_t st = (hipStream_t)stream;
  int rc = CCVPE_OK;
  bool launched = false;
  auto go = [&](auto kern) {
    static bool attr_set = false;                    // one per instantiation (generic lambda): a driver call, not per launch
    if (g.lds_bytes > 48 * 1024 && !attr_set) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, MBP_LDS_MAX);
      if (e != hipSuccess) { rc = fail(CCVPE_ELAUNCH, "mbconv_plane: set smem attr: %s", hipGetErrorString(e)); return; }
      attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.total_blocks), dim3(256), g.lds_bytes, st, p);
    launched = true;
  };
  constexpr int N0 = sizeof(TE) == 2 ? 3 : 5, N1 = sizeof(TE) == 2 ? 4 : 7, N2 = sizeof(TE) == 2 ? 6 : 12;
#define MBP_E(K_, S_, CH_)                                                                        \
  if (k == K_ && stride == S_ && g.CH == CH_) {                                                   \
    if (!expand) go(mbconv_plane_kernel<TE, K_, S_, CH_, 1, false>);                              \
    else if (nkk == N0) go(mbconv_plane_kernel<TE, K_, S_, CH_, N0, true>);                       \
    else if (nkk == N1) go(mbconv_plane_kernel<TE, K_, S_, CH_, N1, true>);                       \
    else if (nkk == N2) go(mbconv_plane_kernel<TE, K_, S_, CH_, N2, true>);                       \
  }
  MBP_E(3, 1, 16) else MBP_E(5, 1, 16) else MBP_E(5, 2, 16) else MBP_E(3, 2, 16) else MBP_E(3, 1, 32) else MBP_E(5, 1, 32)
#undef MBP_E
  if (rc) return rc;
  if (!launched) return fail(CCVPE_EINVAL, "mbconv_plane: no instantiation for k=%d s=%d ch=%d nkk=%d", k, stride, g.CH, nkk);
  return check_launch("mbconv_plane_kernel");
}

int mbplane_launch(int is_bf16, int expand, const void* x, const void* w_exp, int kpad, const float* s0, const float* b0,
                   const float* w_dw, const float* s1, const float* b1, void* y, float* se_partial, int B, int H, int W,
                   int cin, int mid, int k, int stride, int circular, void* stream) {
  return is_bf16 ? mbplane_any<cc_bf16>(expand, x, w_exp, kpad, s0, b0, w_dw, s1, b1, y, se_partial, B, H, W, cin, mid, k, stride,
                                        circular, stream)
                 : mbplane_any<float>(expand, x, w_exp, kpad, s0, b0, w_dw, s1, b1, y, se_partial, B, H, W, cin, mid, k, stride,
                                      circular, stream);
}

}  // namespace ccvpe

extern "C" int ccvpe_set_mbconv_plane_kernels(int mode) {
  const int old = ccvpe::g_mbplane_mode;
  ccvpe::g_mbplane_mode = mode & 7;
  if ((mode >> 8) > 0) ccvpe::g_mbp_plane_kb = (mode >> 8) > 76 ? 76 : (mode >> 8);   // measurements only: LDS budget in KB
  return old;
}
