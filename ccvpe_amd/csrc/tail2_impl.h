// tail2_kernel: the 512 x 512 level of a decoder (the math of tail512.hip: folded deconv1 + conv1.0 -> ReLU -> conv1.2 [+ F.normalize],
// models.py:124-127,145-148,319,341) in the structure of the narrow-level kernels (narrow_impl.h).  Included by narrow_bf16.hip.
//
// tail512_kernel (round 4) keeps W in registers too, but stages its halo through registers one 64-byte chunk at a time, runs conv1.2
// as fp32 16x16x4 MFMAs on the accumulators (stage 2: 4 x the cycles of its bf16 stage 1) and gathers 9 tap planes from LDS with
// five barriers per 8 x 16 tile: traced at 397 us (bf16 orientation) / 610 us (hi + lo localisation) per launch at B = 64, 13 000
// cycles per tile for 3 200 cycles of matrix work and 15 KB of HBM traffic.  Here, per MT x 16 low-res tile (persistent workgroups):
//   stage 1  wave = output parity; all of x's halo tile ((MT + 2) x 18 low-res pixels, every channel) is in LDS, requested by LDS-DMA
//            two tiles ahead; K is flat over (tap, 16-byte piece); positions of the parity's (MT + 1) x 17 grid (one-pixel apron
//            for conv1.2) are numbered linearly and cut into 16-position MFMA tiles; bias (border class) + ReLU, zero outside the
//            image; the 16-channel mid pixels go to an LDS image of the (2 MT + 2) x 34 hi-res grid;
//   stage 2  conv1.2 as a second flat-K MFMA pass over the mid image: K = 9 taps x 16 channels, rows = output channels (1 or 2 of
//            16 used: ~40 more MFMAs per wave, nothing next to the alternative of gathering tap planes); the q = 0 lanes hold the
//            results: + bias, F.normalize for the orientation head, fp32 NCHW stores, softmax partials for the heat-map head.
// Two barriers per tile.  SPLIT (fp32 operands on the bf16 matrix cores as hi + lo planes, tail512.hip's arithmetic: hi.hi + hi.lo
// + lo.hi): x arrives as fp32 by DMA and is converted IN PLACE in LDS ([16 fp32] -> [16 hi | 16 lo], the same 64 bytes) by one
// thread per chunk under stage 2 of the previous tile; mid is kept as hi + lo planes as well, so conv1.2 is fp32-class too.
#pragma once
#include "narrow_impl.h"

namespace ccvpe {

struct Tail2Params {
  const void* x;       // [B,H1,W1,ld0]: bf16, or fp32 (SPLIT)
  const void* w;       // [4][16][Kpad] folded weights (bf16, or fp32 for SPLIT), K order tap * c0 + channel
  const float* shift9; // [9][16]
  const float* w2;     // [COUT][3][3][16]
  const float* b2;     // [COUT]
  float* out;          // [B,COUT,2H1,2W1]
  float* smx;          // COUT == 1, optional: [B][tiles][4][2] softmax partials
  int H1, W1, c0, ld0, Kpad, normalize;
  TileIndex ti;
  int tiles_xy;
};

// PIECES0 = 16-byte pieces of one pixel of x (bf16: c0 / 8; SPLIT: c0 / 4 — one 64-byte chunk = 16 channels = 4 pieces)
template <int PIECES0, int COUT, bool SPLIT, int MT>
struct Tail2Geom {
  static constexpr int HR = MT + 2, HC = 18;
  static constexpr int PP = odd_pitch(PIECES0);
  static constexpr int PX = HR * HC * PP;
  static constexpr int NDX = (PX + 255) / 256;
  static constexpr int XBUF = NDX * 4096;
  static constexpr int PW = 17, P = (MT + 1) * PW, NM = (P + 15) / 16;     // stage-1 positions per parity, 16-position tiles
  static constexpr int NCH1 = SPLIT ? 4 * (PIECES0 / 4) : (4 * PIECES0 + 3) / 4;  // stage-1 k-steps (SPLIT: one per (tap, 16-channel chunk))
  static constexpr int SH = 2 * MT + 2, SW = 34;
  static constexpr int MP = SPLIT ? 5 : 3;                                // mid pixel pitch in pieces (4 / 2 used)
  static constexpr int MID_BASE = 2 * XBUF;
  static constexpr int MID_BYTES = SH * SW * MP * 16;
  static constexpr int SHIFT_BASE = MID_BASE + MID_BYTES;
  static constexpr int LDS_BYTES = SHIFT_BASE + 9 * 16 * 4;
  static constexpr int NCH2 = SPLIT ? 9 : 5;                              // stage-2 k-steps (bf16: 18 octets)
  static constexpr int NM2 = (2 * MT * 32 / 16) / 4;                      // stage-2 16-pixel tiles per wave
};

__device__ __forceinline__ void split8(const float* v, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    hi[i] = (bf16_t)v[i];
    lo[i] = (bf16_t)(v[i] - (float)hi[i]);
  }
}

template <int PIECES0, int COUT, bool SPLIT, int MT>
__global__ __launch_bounds__(256, (SPLIT || PIECES0 > 4) ? 1 : 2) void tail2_kernel(const Tail2Params p) {
  using G = Tail2Geom<PIECES0, COUT, SPLIT, MT>;
  constexpr int HC = G::HC, PP = G::PP, NDX = G::NDX, NM = G::NM, NCH1 = G::NCH1, NCH2 = G::NCH2, NM2 = G::NM2, MP = G::MP;
  constexpr int XP = PP * 16;
  constexpr int NW1 = SPLIT ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) char nsm[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = sgpr(tid >> 6);
  const int py = wave >> 1, px = wave & 1;
  const int f = lane & 15, q = lane >> 4;

  // ---- stage-1 weights of this wave's parity (rows = 16 mid channels) ----------------------------------------------------------
  f32x4 w1[NCH1][NW1];
  if constexpr (!SPLIT) {
    const bf16_t* wp = reinterpret_cast<const bf16_t*>(p.w) + ((size_t)wave * 16 + f) * p.Kpad;
#pragma unroll
    for (int j = 0; j < NCH1; ++j) {
      const int k = 32 * j + 8 * q;                              // flat K: tap * c0 + channel; beyond 4 c0: no such piece
      w1[j][0] = keep_if(*reinterpret_cast<const f32x4*>(wp + (k < 4 * p.c0 ? k : 0)), k < 4 * p.c0);
    }
  } else {
    const float* wp = reinterpret_cast<const float*>(p.w) + ((size_t)wave * 16 + f) * p.Kpad;
#pragma unroll
    for (int j = 0; j < NCH1; ++j) {                             // k-step j = (tap, 16-channel chunk): channels 16 j + 8 (q & 1) .. + 7
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = wp[16 * j + 8 * (q & 1) + e];
      bf16x8 hi, lo;
      split8(v, hi, lo);
      w1[j][0] = __builtin_bit_cast(f32x4, hi);                  // [w_hi | w_hi] against [x_hi | x_lo]
      w1[j][1] = keep_if(__builtin_bit_cast(f32x4, lo), q < 2);  // [w_lo | 0]
    }
  }
  // ---- stage-2 weights: conv1.2 as [output channel][tap][16 channels]; rows >= COUT are zero ---------------------------------------
  constexpr int NW2 = SPLIT ? 2 : 1;
  f32x4 w2r[NCH2][NW2];
#pragma unroll
  for (int j = 0; j < NCH2; ++j) {
    float v[8];
    int tap, c8;
    bool ok;
    if constexpr (SPLIT) { tap = j; c8 = q & 1; ok = f < COUT; }
    else { const int o8 = 4 * j + q; tap = o8 >> 1; c8 = o8 & 1; ok = f < COUT && o8 < 18; }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = ok ? p.w2[((ok ? f : 0) * 9 + (ok ? tap : 0)) * 16 + 8 * c8 + e] : 0.f;
    bf16x8 hi, lo;
    split8(v, hi, lo);
    w2r[j][0] = __builtin_bit_cast(f32x4, hi);
    if constexpr (SPLIT) w2r[j][1] = keep_if(__builtin_bit_cast(f32x4, lo), q < 2);
  }
  float b2v[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) b2v[o] = p.b2[o];

  // ---- stage-1 fragment bases: position pt = 16 m + f of the (MT + 1) x 17 grid -> halo pixel (iy, ix); k-step offsets -------------
  int fb[NM], mw[NM];                                            // mw: byte offset of the position's mid pixel, -1 beyond the grid
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const int pt = 16 * m + f;
    const int pc = pt < G::P ? pt : G::P - 1;
    const int iy = pc / G::PW, ix = pc - iy * G::PW;
    fb[m] = (iy * HC + ix) * XP;
    mw[m] = pt < G::P ? ((2 * iy - py + 1) * G::SW + 2 * ix - px + 1) * (MP * 16) : -1;
  }
  int koff[NCH1];
#pragma unroll
  for (int j = 0; j < NCH1; ++j) {
    int tap, piece;
    if constexpr (SPLIT) { tap = j / (PIECES0 / 4); piece = 4 * (j - tap * (PIECES0 / 4)) + q; }
    else { const int o = 4 * j + q; const int oo = o < 4 * PIECES0 ? o : 0; tap = oo / PIECES0; piece = oo - tap * PIECES0; }
    koff[j] = (((tap >> 1) * HC + (tap & 1)) * PP + piece) * 16;
  }
  // stage-2 fragment base of this lane: wave w owns hi-res rows 4 w' ... of the tile (NM2 tiles: rows x two 16-column halves)
  constexpr int ROWS2 = 2 * MT / 4;                              // hi-res rows per wave
  const int mb2 = ((wave * ROWS2) * G::SW + f) * (MP * 16) + (SPLIT ? q * 16 : 0);
  int k2off[NCH2];
#pragma unroll
  for (int j = 0; j < NCH2; ++j) {
    if constexpr (SPLIT) k2off[j] = ((j / 3) * G::SW + (j % 3)) * (MP * 16);
    else { const int o8 = 4 * j + q; const int oo = o8 < 18 ? o8 : 0; const int tap = oo >> 1; k2off[j] = ((tap / 3) * G::SW + tap % 3) * (MP * 16) + (oo & 1) * 16; }
  }
  const bool tail2_ok = SPLIT || 4 * (NCH2 - 1) + q < 18;

  // ---- DMA pieces of x (tile-invariant) -----------------------------------------------------------------------------------------
  constexpr int ESZ = SPLIT ? 4 : 2, EPP = 16 / ESZ;             // element size, elements per piece
  unsigned xv[NDX];
  int xh[NDX];
#pragma unroll
  for (int k = 0; k < NDX; ++k) {
    const int pidx = min(k * 256 + tid, G::PX - 1);
    const int pix = pidx / PP, pc = pidx - pix * PP;
    const int hy = pix / HC, hx = pix - hy * HC;
    xv[k] = (unsigned)(((hy * p.W1 + hx) * p.ld0 + (pc < PIECES0 ? pc : 0) * EPP) * ESZ);
    xh[k] = (hy << 8) | hx;
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)nsm;
  const int H1s = sgpr(p.H1), W1s = sgpr(p.W1);
  const int H2 = 2 * H1s, W2 = 2 * W1s;
  for (int e = tid; e < 9 * 16; e += 256) reinterpret_cast<float*>(nsm + G::SHIFT_BASE)[e] = p.shift9[e];

  auto stage = [&](int t, int buf) {
    int b, ty, tx;
    tile_decode(p.ti, t, b, ty, tx);
    const int y0 = ty * MT, x0 = tx * 16;
    const char* sb0 = reinterpret_cast<const char*>(p.x) + ((long)(b * H1s + y0 - 1) * W1s + (x0 - 1)) * (long)(p.ld0 * ESZ);
    const bool interior = y0 > 0 && y0 + MT < H1s && x0 > 0 && x0 + 16 < W1s;
    if (interior) {
#pragma unroll
      for (int k = 0; k < NDX; ++k) {
        const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * G::XBUF + (k * 256 + wave * 64) * 16));
        if ((k + 1) * 256 <= G::PX || k * 256 + tid < G::PX) dma16(ldsw, xv[k], sb0);
      }
    } else {
#pragma unroll
      for (int k = 0; k < NDX; ++k) {
        if ((k + 1) * 256 <= G::PX || k * 256 + tid < G::PX) {
          const int iy = y0 - 1 + (xh[k] >> 8), ix = x0 - 1 + (xh[k] & 255);
          const bool ok = (unsigned)iy < (unsigned)H1s && (unsigned)ix < (unsigned)W1s;
          const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * G::XBUF + (k * 256 + wave * 64) * 16));
          if (ok) dma16(ldsw, xv[k], sb0);
          else *reinterpret_cast<f32x4*>(nsm + buf * G::XBUF + (k * 256 + tid) * 16) = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
    }
  };
  // SPLIT: fp32 chunks of buffer `buf` -> [16 hi | 16 lo] in place, one thread per 64-byte chunk
  auto convert = [&](int buf) {
    if constexpr (SPLIT) {
      constexpr int NCHUNK = G::HR * HC * (PIECES0 / 4);
      for (int u = tid; u < NCHUNK; u += 256) {
        const int pix = u / (PIECES0 / 4), c = u - pix * (PIECES0 / 4);
        char* ch = nsm + buf * G::XBUF + (pix * PP + 4 * c) * 16;
        float v[16];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x4 t4 = *reinterpret_cast<const f32x4*>(ch + 16 * e);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[4 * e + r] = t4[r];
        }
        bf16x8 h0, l0, h1, l1;
        split8(v, h0, l0);
        split8(v + 8, h1, l1);
        *reinterpret_cast<bf16x8*>(ch) = h0;
        *reinterpret_cast<bf16x8*>(ch + 16) = h1;
        *reinterpret_cast<bf16x8*>(ch + 32) = l0;
        *reinterpret_cast<bf16x8*>(ch + 48) = l1;
      }
    }
  };

  // ---- stage 1 -------------------------------------------------------------------------------------------------------------------
  auto stage1 = [&](auto buf_tag, int t) {
    constexpr int BUF = decltype(buf_tag)::value;
    const char* hb = nsm + BUF * G::XBUF;
    f32x4 acc[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int HALF = (NM + 1) / 2;                           // fragments of half the position tiles in flight at a time
    f32x4 a[2][HALF];
    auto rd = [&](int s, f32x4* dst) {                           // step s = (k-step j, half h)
      const int j = s >> 1, h = s & 1;
#pragma unroll
      for (int i = 0; i < HALF; ++i)
        if (h * HALF + i < NM) dst[i] = *reinterpret_cast<const f32x4*>(hb + fb[h * HALF + i] + koff[j]);
    };
    rd(0, a[0]);
#pragma unroll
    for (int s = 0; s < 2 * NCH1; ++s) {
      const int j = s >> 1, h = s & 1, cur = s & 1;
      if (s + 1 < 2 * NCH1) rd(s + 1, a[cur ^ 1]);
      __builtin_amdgcn_sched_barrier(0);
      if (!SPLIT && j == NCH1 - 1 && (4 * PIECES0) % 4 != 0) {
#pragma unroll
        for (int i = 0; i < HALF; ++i) a[cur][i] = keep_if(a[cur][i], 4 * j + q < 4 * PIECES0);
      }
#pragma unroll
      for (int i = 0; i < HALF; ++i)
        if (h * HALF + i < NM) {
#pragma unroll
          for (int hh = 0; hh < NW1; ++hh) acc[h * HALF + i] = mfma_stage<bf16_t>(w1[j][hh], a[cur][i], acc[h * HALF + i]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // bias by border class + ReLU, zero outside the image -> mid image (this lane: channels 4 q .. 4 q + 3 of its position)
    int b, ty, tx;
    tile_decode(p.ti, t, b, ty, tx);
    const int y0 = ty * MT, x0 = tx * 16;
    char* mid = nsm + G::MID_BASE;
    const float* sh9 = reinterpret_cast<const float*>(nsm + G::SHIFT_BASE);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const int pt = 16 * m + f;
      const int pc = pt < G::P ? pt : G::P - 1;
      const int iy = pc / G::PW, ix = pc - iy * G::PW;
      const int Y = 2 * (y0 + iy) - py, X = 2 * (x0 + ix) - px;
      const bool inside = (unsigned)Y < (unsigned)H2 && (unsigned)X < (unsigned)W2;
      const int rc = Y <= 0 ? 0 : (Y >= H2 - 1 ? 2 : 1);
      const int cc = X <= 0 ? 0 : (X >= W2 - 1 ? 2 : 1);
      const f32x4 sh = *reinterpret_cast<const f32x4*>(sh9 + (rc * 3 + cc) * 16 + q * 4);
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = inside ? fmaxf(acc[m][r] + sh[r], 0.f) : 0.f;
      if (mw[m] >= 0) {
        bf16x4 hi;
#pragma unroll
        for (int r = 0; r < 4; ++r) hi[r] = (bf16_t)v[r];
        *reinterpret_cast<bf16x4*>(mid + mw[m] + q * 8) = hi;
        if constexpr (SPLIT) {
          bf16x4 lo;
#pragma unroll
          for (int r = 0; r < 4; ++r) lo[r] = (bf16_t)(v[r] - (float)hi[r]);
          *reinterpret_cast<bf16x4*>(mid + mw[m] + 32 + q * 8) = lo;
        }
      }
    }
  };

  // ---- stage 2: conv1.2 over the mid image; results in the q = 0 lanes (rows 0 .. COUT-1 of the MFMA tile) -----------------------------
  float res[COUT][NM2];                                        // results of the last stage 2: stored one phase LATER (see the schedule)
  auto stage2 = [&](int t) {
    const char* mid = nsm + G::MID_BASE;
    f32x4 acc[NM2];
#pragma unroll
    for (int m = 0; m < NM2; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 a[2][NM2];
    auto rd = [&](int j, f32x4* dst) {
#pragma unroll
      for (int m = 0; m < NM2; ++m)               // tile m: row m >> 1 of this wave's rows, column half m & 1
        dst[m] = *reinterpret_cast<const f32x4*>(mid + mb2 + k2off[j] + ((m >> 1) * G::SW + 16 * (m & 1)) * (MP * 16));
    };
    rd(0, a[0]);
#pragma unroll
    for (int j = 0; j < NCH2; ++j) {
      const int cur = j & 1;
      if (j + 1 < NCH2) rd(j + 1, a[cur ^ 1]);
      __builtin_amdgcn_sched_barrier(0);
      if (!SPLIT && j == NCH2 - 1) {
#pragma unroll
        for (int m = 0; m < NM2; ++m) a[cur][m] = keep_if(a[cur][m], tail2_ok);
      }
#pragma unroll
      for (int m = 0; m < NM2; ++m)
#pragma unroll
        for (int hh = 0; hh < NW2; ++hh) acc[m] = mfma_stage<bf16_t>(w2r[j][hh], a[cur][m], acc[m]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < NM2; ++m) {
      float v0 = acc[m][0] + b2v[0];
      if constexpr (COUT == 2) {
        float v1 = acc[m][1] + b2v[1];
        if (p.normalize) {                                     // F.normalize(p=2, dim=1, eps=1e-12): v / max(|v|, 1e-12)
          const float inv = __builtin_amdgcn_rsqf(fmaxf(v0 * v0 + v1 * v1, 1e-24f));      // (bf16 path: 1 ulp of rsq is far inside its tolerance)
          v0 *= inv;
          v1 *= inv;
        }
        res[1][m] = v1;
      }
      res[0][m] = v0;
    }
    int b, ty, tx;
    tile_decode(p.ti, t, b, ty, tx);
    if (COUT == 1 && p.smx) {                                  // softmax partials of this wave's logits (tail512.hip)
      float mx = -__builtin_huge_valf();
      if (q == 0) {
#pragma unroll
        for (int m = 0; m < NM2; ++m) mx = fmaxf(mx, res[0][m]);
      }
      mx = wave_max(mx);
      float e = 0.f;
      if (q == 0) {
#pragma unroll
        for (int m = 0; m < NM2; ++m) e += expf(res[0][m] - mx);
      }
      e = wave_sum(e);
      if (lane == 0) {
        float* dst = p.smx + ((size_t)((b * p.ti.dy.d + ty) * p.ti.dx.d + tx) * 4 + wave) * 2;
        dst[0] = mx;
        dst[1] = e;
      }
    }
  };
  auto store_results = [&](int t) {                            // the q = 0 lanes hold 16 consecutive pixels of a row per position tile
    int b, ty, tx;
    tile_decode(p.ti, t, b, ty, tx);
    const int Y0 = 2 * ty * MT + wave * ROWS2, X0 = 2 * tx * 16 + f;
    if (q == 0) {
#pragma unroll
      for (int m = 0; m < NM2; ++m)
#pragma unroll
        for (int o = 0; o < COUT; ++o)
          p.out[((size_t)(b * COUT + o) * H2 + Y0 + (m >> 1)) * W2 + X0 + 16 * (m & 1)] = res[o][m];
    }
  };

  // Schedule per tile s of this workgroup's sequence (buffer s & 1):
  //   stage1(s) | wait, barrier A | request DMA(s+2) into buffer s & 1 (just freed), STORE the results of stage2(s-1),
  //   convert buffer(s+1) [SPLIT], stage2(s) | barrier B
  // Everything the wait in front of barrier A covers — DMA(s+1), the stores of tile s-2 — was issued a whole tile earlier.  (First
  // version: DMA and result stores issued right before a stage 1 that is 40 matrix instructions long in bf16, i.e. the full memory
  // latency exposed at every barrier: 521 us against tail512_kernel's 398.)
  int t = blockIdx.x;
  const int step = gridDim.x;
  const int total = p.ti.total;
  if (t >= total) return;
  stage(t, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  convert(0);
  if (t + step < total) stage(t + step, 1);
  __syncthreads();
  int tprev = -1;
  while (true) {
    stage1(std::integral_constant<int, 0>{}, t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                          // A: mid complete, buffer 1 landed, everyone done reading buffer 0
    if (t + 2 * step < total) stage(t + 2 * step, 0);
    if (tprev >= 0) store_results(tprev);
    if (t + step < total) convert(1);
    stage2(t);
    __syncthreads();                                          // B: mid free, buffer 1 converted
    tprev = t;
    t += step;
    if (t >= total) break;
    stage1(std::integral_constant<int, 1>{}, t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 2 * step < total) stage(t + 2 * step, 1);
    store_results(tprev);
    if (t + step < total) convert(0);
    stage2(t);
    __syncthreads();
    tprev = t;
    t += step;
    if (t >= total) break;
  }
  store_results(tprev);
}

template <int PIECES0, int COUT, bool SPLIT, int MT>
static int launch_tail2(Tail2Params p, int batch, hipStream_t stream) {
  using G = Tail2Geom<PIECES0, COUT, SPLIT, MT>;
  static_assert(G::LDS_BYTES <= 160 * 1024, "tail2_kernel: LDS");
  const int tiles_x = p.W1 / 16, tiles_y = p.H1 / MT;
  const long total = (long)tiles_x * tiles_y * batch;
  if (total > 0x7fffffffL || total * (tiles_x > tiles_y ? tiles_x : tiles_y) >= (1L << 32)) return fail(CCVPE_EINVAL, "tail2: grid too large");
  p.ti = make_tile_index(tiles_x, tiles_y, (int)total);
  p.tiles_xy = tiles_x * tiles_y;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)tail2_kernel<PIECES0, COUT, SPLIT, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    if (e != hipSuccess) return fail(CCVPE_ELAUNCH, "tail2_kernel: set smem attr: %s", hipGetErrorString(e));
    attr_set = true;
  }
  const int per_cu = (!SPLIT && PIECES0 <= 4 && 2 * G::LDS_BYTES <= 160 * 1024) ? 2 : 1;     // the 32-channel bf16 form fits 256 registers: two workgroups share a CU
  const long slots = (long)per_cu * num_cus();
  const int grid = (int)(total < slots ? total : slots);
  hipLaunchKernelGGL((tail2_kernel<PIECES0, COUT, SPLIT, MT>), dim3(grid), dim3(256), G::LDS_BYTES, stream, p);
  return check_launch("tail2_kernel");
}

}  // namespace ccvpe
