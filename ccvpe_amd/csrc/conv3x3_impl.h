// conv3x3_kernel (and, in diagnostics builds, conv3x3_wreg_kernel): see the comment blocks below.  Included by conv3x3_f32.hip / conv3x3_bf16.hip.
#pragma once
#include "conv_common.h"

namespace ccvpe {

// Stage s = (chunk, tap): chunk = 16 consecutive concat channels, tap = ky*3+kx.
// ---------------------------------------------------------------------------------------------
// NW = waves per workgroup (4 or 8).  The 8-wave form (512 threads, 2x the pixel tile) stages the same W
// tile per K-stage for twice the MFMAs: the ablation (tools/ablation) attributes 11 % of the 4-wave
// kernel's time to W staging, 2 % to barriers, 5 % to LDS fragment reads (MFMA-only ceiling 140 TF).
// DMA = W tile staged by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write, no wait before
// the LDS store).  The DMA writes lane-linear (base + lane*16 B), so the W stage is an UNPADDED [BN][64 B]
// image and the bank-conflict fix is an XOR swizzle of the 16-byte piece index, applied to the per-lane
// SOURCE address and to the fragment read: slot(r, c) = 4r + (c ^ perm[(r>>2)&3]), perm = (0,2,3,1)
// (conflict-free for the four ds_read_b128 lane groups).  Needs Npad % BN == 0 (no row guard possible).
#ifdef CCVPE_ABLATE   // diagnostics build only (make EXTRA=-DCCVPE_ABLATE): A/B switches read from the environment at load
// CCVPE_CONV3_TPS=1: one tap per stage instead of a row of taps; CCVPE_CONV3_WREG=1: fp32 3x3 convolutions through
// conv3x3_wreg_kernel (W fragments straight from L2); CCVPE_CONV3_NW8=0: never the 8-wave form of the 128-column tile
static const bool g_conv3_wreg = getenv("CCVPE_CONV3_WREG") && getenv("CCVPE_CONV3_WREG")[0] == '1';
static const bool g_conv3_nw8 = !(getenv("CCVPE_CONV3_NW8") && getenv("CCVPE_CONV3_NW8")[0] == '0');
static const int g_conv3_tps = (getenv("CCVPE_CONV3_TPS") && getenv("CCVPE_CONV3_TPS")[0] == '1') ? 1 : 3;
#else                 // product build: the measured defaults, no environment reads
constexpr bool g_conv3_nw8 = true;
constexpr int g_conv3_tps = 3;
#endif

// TPS = taps per stage.  TPS = 1: one (16-channel chunk, tap) per stage.  TPS = 3 (W by LDS-DMA only): a stage is one ROW of
// taps (ky; kx = 0..2) of a chunk — three W panels land per stage, the fragments of tap kx+1 are read from LDS while the
// MFMAs of tap kx run (two fragment register sets), and the closing barrier, the DMA's vmcnt(0) and the exposed first LDS
// round trip are paid once per 240 matrix instructions instead of once per 80.
template <typename T, int MT, int NT, int WN, int NW, bool DMA, int TPS>
struct Conv3Geom {
  static constexpr int BLD = DMA ? 16 : LDS_LD;
  static constexpr int WM = NW / WN;
  static constexpr int BM = 16 * MT * WM;
  static constexpr int BN = 16 * NT * WN;
  static constexpr int TH = BM / 16;
  static constexpr int HPX = (TH + 2) * 18;
  // Halo pixel rows of 64 bytes (no padding) in 24 slots per halo row, the 16-byte piece index XOR-ed by
  // ((column >> 2) & 1) << 1: conflict-free ds_read_b128 fragments (tools/lds_layout.py) at 7 % more LDS instead of 2-way
  // conflicts on every read; with 24 = 0 mod 8 slots per row the swizzle term depends on (lane, kx) only -> three per-lane
  // address registers, rows and ky by immediate offsets.  (HSW = false is the round-3 layout: 18 slots of 80 bytes.)
  // (the 8-wave fp32 form without W DMA — a fallback for column counts that are not a tile multiple — lives under a 128-VGPR
  // cap and spills with the three extra address registers: it keeps the round-3 layout)
  static constexpr bool HSW = !(NW == 8 && sizeof(T) == 4 && !DMA);
  static constexpr int HCP = HSW ? 24 : 18;
  static constexpr int HLD = HSW ? 16 : LDS_LD;
  static constexpr int HS_FLOATS = (TH + 2) * HCP * HLD;
  static constexpr int BS_FLOATS = 2 * TPS * BN * BLD;
  static constexpr int LDS_BYTES = (HS_FLOATS + BS_FLOATS) * 4;
};

template <typename T, int MT, int NT, int WN, int NW, bool DMA, int TPS>
__global__ __launch_bounds__(64 * NW, (NW == 8 && sizeof(T) == 4) ? 4 : 2) void conv3x3_kernel(const IgemmParams p) {
  static_assert(TPS == 1 || (TPS == 3 && DMA), "a row of taps per stage needs the DMA W path");
  using G = Conv3Geom<T, MT, NT, WN, NW, DMA, TPS>;
  constexpr int BLD = G::BLD;                      // floats per W stage row
  constexpr int NTHR = 64 * NW;
  constexpr int RPP = NTHR / 4;                    // staged rows per pass
  constexpr int E = ElemTraits<T>::E;
  constexpr int SK = 4 * E;                        // channels per chunk: 16 (fp32) or 32 (bf16)
  constexpr int WM = NW / WN;
  constexpr int BM = 16 * MT * WM;
  constexpr int BN = 16 * NT * WN;
  constexpr int TH = BM / 16;
  constexpr int HR = TH + 2;                       // halo rows
  constexpr int HC = 18;                           // halo columns
  constexpr int HPX = HR * HC;
  constexpr int H_IT = (HPX * 4 + NTHR - 1) / NTHR;  // float4 loads per thread per chunk (halo)
  constexpr int B_IT = (BN + RPP - 1) / RPP;
  constexpr int NG = 9 / TPS;                      // stages per chunk
  constexpr bool HSW = G::HSW;
  constexpr int HCP = G::HCP, HLD = G::HLD;

  // halo is single-buffered (one extra barrier per chunk) to keep LDS small -> 2-4 blocks/CU
  extern __shared__ __attribute__((aligned(16))) float c3_sm[];
  float* Hs = c3_sm;                               // [HPX][LDS_LD]
  float* Bs = c3_sm + G::HS_FLOATS;                // [2][TPS][BN][BLD]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = sgpr(tid >> 6);               // wave-uniform by construction: keep it (and wm, wn) in scalar registers
  const int wm = wave / WN;
  const int wn = wave % WN;

  const int tile = xcd_tile(blockIdx.x, p.tiles_total);
  const int tn = tile % p.tiles_n;
  const int ts = tile / p.tiles_n;                 // spatial tile id: x fastest, then y, then sample
  const int tx = ts % p.tiles_x;
  const int ty = (ts / p.tiles_x) % p.tiles_y;
  const int b = ts / (p.tiles_x * p.tiles_y);
  const int y0 = ty * TH, x0 = tx * 16;
  const int n0 = tn * BN;
  const int ctot = p.c0 + p.c1;
  const int nchunks = (ctot + SK - 1) / SK;
  const int nstages = nchunks * NG;
  const T* src0 = reinterpret_cast<const T*>(p.src0);
  const T* src1 = reinterpret_cast<const T*>(p.src1);
  const T* wp = reinterpret_cast<const T*>(p.w);

  // halo staging coordinates (fixed per thread)
  int h_off[H_IT];     // LDS float offset, -1 if this slot is unused
  int h_pix[H_IT];     // global pixel index (b*H+iy)*W+ix, -1 if outside the image
  int h_sub[H_IT];
#pragma unroll
  for (int it = 0; it < H_IT; ++it) {
    const int idx = tid + NTHR * it;
    const int px = idx >> 2, sub = idx & 3;
    h_sub[it] = sub;
    if (px < HPX) {
      const int hy = px / HC, hx = px - hy * HC;
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      h_off[it] = (hy * HCP + hx) * HLD + (HSW ? (sub ^ (((hx >> 2) & 1) << 1)) : sub) * 4;
      h_pix[it] = ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) ? (b * p.H + iy) * p.W + ix : -1;
    } else {
      h_off[it] = -1;
      h_pix[it] = -1;
    }
  }
  const int srow = tid >> 2, ssub = tid & 3;

  f32x4 h_reg[H_IT], b_reg[B_IT];
  int h_chunk = 0;                         // chunk held in h_reg (workgroup-uniform: lives in a scalar register)
  const int ld0s = sgpr(p.ld0), ld1s = sgpr(p.ld1);

  auto load_halo = [&](int chunk) {        // raw loads from clamped addresses; masked in store_halo (STAGING RULE)
    h_chunk = chunk;
#pragma unroll
    for (int it = 0; it < H_IT; ++it) {
      const int ch = chunk * SK + h_sub[it] * E;
      const bool ok = h_pix[it] >= 0 && ch < ctot;
      const bool from0 = !ok || ch < p.c0;
      const T* base = from0 ? src0 : src1;
      const size_t off = ok ? (size_t)h_pix[it] * (from0 ? ld0s : ld1s) + (from0 ? ch : ch - p.c0) : 0;
      h_reg[it] = *reinterpret_cast<const f32x4*>(base + off);
    }
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int it = 0; it < H_IT; ++it)
      if (h_off[it] >= 0)
        *reinterpret_cast<f32x4*>(Hs + h_off[it]) =
            keep_if(h_reg[it], h_pix[it] >= 0 && h_chunk * SK + h_sub[it] * E < ctot);
  };
  // DMA lanes: one wave-instruction moves 16 W rows x 64 bytes; lane -> (row = lane>>2, swizzled 16-byte piece).  A wave owns
  // row groups g = wave, wave + NW, ...; the per-lane byte offset of each group is stage-invariant (one VGPR per group) and
  // the stage's K offset is added to the SCALAR base, so requesting a panel costs no vector ALU work at all — with the
  // address rebuilt per instruction (64-bit multiply-adds) the requests of a stage were a ~500-cycle burst of VALU work in
  // front of every MFMA block (ablation: 136 TF without the W requests, 118 with them, the same with their wait removed).
  constexpr int NSLOT = DMA ? (BN / 16 + NW - 1) / NW : 1;
  unsigned wvoff[NSLOT];
  if constexpr (DMA) {
    const int rl = lane >> 2;
    const int c = (lane & 3) ^ w_swz(rl);
#pragma unroll
    for (int q = 0; q < NSLOT; ++q) {
      const int g = min(wave + NW * q, BN / 16 - 1);
      wvoff[q] = ((unsigned)(n0 + g * 16 + rl) * (unsigned)p.Kpad + (unsigned)(c * E)) * (unsigned)sizeof(T);
    }
  }
  const unsigned bs_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)Bs;
  auto load_w = [&](int chunk, int tg, int dbuf) {   // the TPS taps tg*TPS .. of `chunk` -> Bs[dbuf][0..TPS)
    if constexpr (DMA) {
#pragma unroll
      for (int t = 0; t < TPS; ++t) {
        const char* sbase = reinterpret_cast<const char*>(wp) + ((size_t)(tg * TPS + t) * ctot + (size_t)chunk * SK) * sizeof(T);
#pragma unroll
        for (int q = 0; q < NSLOT; ++q) {
          const int g = wave + NW * q;
          if (g < BN / 16) {
            // Inline assembly: for the builtin the compiler waits vmcnt(0) in front of every later LDS read it cannot prove
            // disjoint from the DMA's target, which forces the request to the END of a stage.  Written this way the panels
            // of the NEXT stage are requested at the START of the stage; dma_wait() below is the matching vmcnt(0).
            const unsigned lds = __builtin_amdgcn_readfirstlane(bs_lds + (unsigned)(((dbuf * TPS + t) * BN + g * 16) * BLD * 4));
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(wvoff[q]), "s"(sbase)
                         : "memory", "m0");
          }
        }
      }
    } else {
      const int ch = chunk * SK + ssub * E;
      const int kcol = ch < ctot ? tg * ctot + ch : 0;         // beyond the channel range the halo piece is zero anyway
#pragma unroll
      for (int it = 0; it < B_IT; ++it) {
        const int nr = min(n0 + srow + RPP * it, p.Npad - 1);
        b_reg[it] = *reinterpret_cast<const f32x4*>(wp + (size_t)nr * p.Kpad + kcol);
      }
    }
  };
  auto store_w = [&](int buf) {
    if constexpr (!DMA) {
#pragma unroll
      for (int it = 0; it < B_IT; ++it) {
        const int nrow = srow + RPP * it;
        if (nrow < BN) *reinterpret_cast<f32x4*>(&Bs[(buf * BN + nrow) * BLD + ssub * 4]) = b_reg[it];
      }
    }
  };
  auto dma_wait = [&]() {                  // the W panels requested by load_w have landed in LDS (this wave's share)
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;
  const int fk = (lane >> 4) * 4;
  const int bcol = DMA ? (((lane >> 4) ^ w_swz(frow)) * 4) : fk;
  int acol[3];                                     // per-lane float offset of the A fragment for kx = 0..2 (row 0 of this wave)
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int hx = frow + kx;
    acol[kx] = (wm * MT * HCP + hx) * HLD + (HSW ? (((lane >> 4) ^ (((hx >> 2) & 1) << 1)) * 4) : fk);
  }

  load_halo(0);
  load_w(0, 0, 0);
  store_halo();
  store_w(0);
  dma_wait();
  __syncthreads();

#ifdef CCVPE_ABLATE   // diagnostics build only (tools/gpu/ablate_c3.sh): 2 = no loads in the loop, 8 = no fragment reads, 16 = no barriers,
                      // 32 = no W DMA, 64 = no halo traffic, 128 = no wait for the DMA
  const int abl = p.ablate;
#else
  constexpr int abl = 0;
#endif
  int chunk = 0, tg = 0;
  for (int s = 0; s < nstages; ++s) {
    const bool more = (s + 1 < nstages) && !(abl & 2);
    int nchunk = chunk, ntg = tg + 1;
    if (ntg == NG) { ntg = 0; ++nchunk; }
    const bool next_halo = (tg == 0) && (chunk + 1 < nchunks) && !(abl & 2);

    constexpr int FS = TPS > 1 ? 2 : 1;            // fragment register sets
    f32x4 af[FS][MT], bf[FS][NT];
    auto read_frag = [&](int t, f32x4* a, f32x4* w_) {
      const int tap = tg * TPS + t;
      const int ky = TPS == 3 ? tg : tap / 3, kx = TPS == 3 ? t : tap - ky * 3;
      int ab;                                        // (TPS = 3: kx is a compile-time index into acol; else selected)
      if constexpr (TPS == 3) ab = acol[t];
      else ab = kx == 0 ? acol[0] : (kx == 1 ? acol[1] : acol[2]);
#pragma unroll
      for (int i = 0; i < MT; ++i)
        if constexpr (HSW) a[i] = *reinterpret_cast<const f32x4*>(Hs + ab + (i + ky) * HCP * HLD);
        else a[i] = *reinterpret_cast<const f32x4*>(Hs + (((wm * MT + i) + ky) * HC + frow + kx) * LDS_LD + fk);
#pragma unroll
      for (int j = 0; j < NT; ++j)
        w_[j] = *reinterpret_cast<const f32x4*>(&Bs[((((s & 1) * TPS + t) * BN) + (wn * NT + j) * 16 + frow) * BLD + bcol]);
    };
    if (!(abl & 8) || s == 0) read_frag(0, af[0], bf[0]);
#pragma unroll
    for (int t = 0; t < TPS; ++t) {
      const int cur = t % FS;
      if (t + 1 < TPS) {
        if (!(abl & 8) || s == 0)
        read_frag(t + 1, af[(t + 1) % FS], bf[(t + 1) % FS]);   // next tap's fragments: in flight under this tap's MFMAs
      }
      if (t == 0) {
        // the next stage's W panels and (once per chunk) halo: requested under the whole stage's matrix work
        if (next_halo && !(abl & 64)) load_halo(chunk + 1);   // first: re-using h_reg makes the compiler wait for what is in flight
        if (more && !(abl & 32)) load_w(nchunk, ntg, (s + 1) & 1);
      }
      if (TPS > 1) __builtin_amdgcn_sched_barrier(0);           // keep those reads / loads ABOVE this tap's MFMAs
      if (sizeof(T) == 4) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[cur][j][kk], af[cur][i][kk], acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = mfma_stage<T>(bf[cur][j], af[cur][i], acc[i][j]);
      }
    }

    // keep the closing barrier (and the vmcnt(0) the DMA needs in front of it) BELOW the matrix work: the scheduler moves
    // s_barrier freely among MFMAs and had put it after the first one
    __builtin_amdgcn_sched_barrier(0);
    if (more) store_w((s + 1) & 1);
    if (!(abl & 128)) dma_wait();
    if (!(abl & 16)) __syncthreads();
    if (tg == NG - 1 && more && !(abl & 64)) {   // chunk boundary: every wave is done reading the halo -> overwrite it
      store_halo();
      if (!(abl & 16)) __syncthreads();
    }
    chunk = nchunk;
    tg = ntg;
  }

  // ---- epilogue (scale/shift are loaded per column group, not hoisted: keeping the kernel under
  // 128 VGPRs lets 4 workgroups share a CU, which is what keeps the MFMA pipe fed) ---------------
  const int epix = lane & 15;
  const int en = (lane >> 4) * 4;
  const int ox = x0 + epix;
  auto epilogue = [&](auto act_tag) {
  constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + (wn * NT + j) * 16 + en;
    if (n >= p.N) continue;
    float sc[4], sh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool ok = n + q < p.N;
      sc[q] = (ok && p.scale) ? p.scale[n + q] : 1.0f;
      sh[q] = (ok && p.shift) ? p.shift[n + q] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int oy = y0 + wm * MT + i;
      if (oy >= p.H || ox >= p.W) continue;
      const size_t m = (size_t)(b * p.H + oy) * p.W + ox;
      store4<T, ACT>(p, acc[i][j], n, m * p.ldd + n, m * p.ldres, sc, sh);
    }
  }
  };
  CCVPE_ACT_DISPATCH(p.act, epilogue);
}

#ifdef CCVPE_ABLATE   // the W-from-L2 experiment (measured the same or slower): diagnostics build only
// ---------------------------------------------------------------------------------------------
// 3x3 convolution, W FRAGMENTS STRAIGHT FROM L2 (no LDS for W, one barrier per 16-channel chunk).
//
// conv3x3_kernel above shares the W panel of a stage through LDS: every stage ends in a barrier, and with two waves per
// SIMD (the accumulators of a 128 x 160 tile leave room for no more) the four SIMDs of a CU keep waiting for each other —
// rocprof shows the MFMA pipe 77-80 % busy under a perfectly clean instruction stream.  Here a wave fetches its own W
// fragments (lane = row, 16-byte K piece: exactly the MFMA operand layout) from global memory one tap ahead, into the second
// of two fragment register sets; the two waves that share a row block fetch the same 64-byte pieces (L1 / L2 hits, ~8 B
// per cycle per CU).  LDS holds only the activation halo, double-buffered: chunk c+1 is loaded to registers during tap 0
// of chunk c, stored during tap 4, and ONE barrier at the start of tap 8 both publishes it and retires the buffer of
// chunk c-1.  Between barriers a wave runs 720 MFMAs on its own.
// ---------------------------------------------------------------------------------------------
template <typename T, int MT, int NT, int WN>
__global__ __launch_bounds__(256, 2) void conv3x3_wreg_kernel(const IgemmParams p) {
  constexpr int NW = 4, NTHR = 256;
  constexpr int E = ElemTraits<T>::E;
  constexpr int SK = 4 * E;                        // channels per chunk: 16 (fp32) or 32 (bf16)
  constexpr int WM = NW / WN;
  constexpr int BM = 16 * MT * WM;
  constexpr int BN = 16 * NT * WN;
  constexpr int TH = BM / 16;
  constexpr int HR = TH + 2, HC = 18, HPX = HR * HC;
  constexpr int H_IT = (HPX * 4 + NTHR - 1) / NTHR;

  __shared__ __attribute__((aligned(16))) float Hs[2][HPX * LDS_LD];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave % WN;

  const int tile = xcd_tile(blockIdx.x, p.tiles_total);
  const int tn = tile % p.tiles_n;
  const int ts = tile / p.tiles_n;
  const int tx = ts % p.tiles_x;
  const int ty = (ts / p.tiles_x) % p.tiles_y;
  const int b = ts / (p.tiles_x * p.tiles_y);
  const int y0 = ty * TH, x0 = tx * 16;
  const int n0 = tn * BN;
  const int ctot = p.c0 + p.c1;
  const int nchunks = (ctot + SK - 1) / SK;
  const int nstages = nchunks * 9;
  const T* src0 = reinterpret_cast<const T*>(p.src0);
  const T* src1 = reinterpret_cast<const T*>(p.src1);
  const T* wp = reinterpret_cast<const T*>(p.w);

  int h_off[H_IT], h_pix[H_IT], h_sub[H_IT];
#pragma unroll
  for (int it = 0; it < H_IT; ++it) {
    const int idx = tid + NTHR * it;
    const int px = idx >> 2, sub = idx & 3;
    h_sub[it] = sub;
    if (px < HPX) {
      const int hy = px / HC, hx = px - hy * HC;
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      h_off[it] = px * LDS_LD + sub * 4;         // this experiment keeps the round-3 halo layout: [HPX][LDS_LD], no swizzle
      h_pix[it] = ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) ? (b * p.H + iy) * p.W + ix : -1;
    } else {
      h_off[it] = -1;
      h_pix[it] = -1;
    }
  }
  f32x4 h_reg[H_IT];
  int h_chunk = 0;
  const int ld0s = sgpr(p.ld0), ld1s = sgpr(p.ld1);
  auto load_halo = [&](int chunk) {        // raw loads from clamped addresses; masked in store_halo (STAGING RULE)
    h_chunk = chunk;
#pragma unroll
    for (int it = 0; it < H_IT; ++it) {
      const int ch = chunk * SK + h_sub[it] * E;
      const bool ok = h_pix[it] >= 0 && ch < ctot;
      const bool from0 = !ok || ch < p.c0;
      const T* base = from0 ? src0 : src1;
      const size_t off = ok ? (size_t)h_pix[it] * (from0 ? ld0s : ld1s) + (from0 ? ch : ch - p.c0) : 0;
      h_reg[it] = *reinterpret_cast<const f32x4*>(base + off);
    }
  };
  auto store_halo = [&](int buf) {
#pragma unroll
    for (int it = 0; it < H_IT; ++it)
      if (h_off[it] >= 0)
        *reinterpret_cast<f32x4*>(&Hs[buf][h_off[it]]) =
            keep_if(h_reg[it], h_pix[it] >= 0 && h_chunk * SK + h_sub[it] * E < ctot);
  };

  const int frow = lane & 15;
  const int fk = (lane >> 4) * 4;
  // W fragment of column block j: row n0 + (wn*NT + j)*16 + frow (rows past Npad: any valid row, never stored), K piece
  // lane>>4 of the stage's 64 bytes.  Pieces past the channel range meet zeroed halo pieces.
  unsigned woff[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
    woff[j] = (unsigned)min(n0 + (wn * NT + j) * 16 + frow, p.Npad - 1) * (unsigned)p.Kpad + (lane >> 4) * E;
  auto load_b = [&](int chunk, int tap, f32x4* dst) {
    const T* base = wp + tap * ctot + chunk * SK;
#pragma unroll
    for (int j = 0; j < NT; ++j) dst[j] = *reinterpret_cast<const f32x4*>(base + woff[j]);
  };
  auto read_a = [&](int hbuf, int tap, f32x4* dst) {
    const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
    for (int i = 0; i < MT; ++i)
      dst[i] = *reinterpret_cast<const f32x4*>(&Hs[hbuf][(((wm * MT + i) + ky) * HC + frow + kx) * LDS_LD + fk]);
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 af[2][MT], bf[2][NT];
  load_halo(0);
  load_b(0, 0, bf[0]);
  store_halo(0);
  __syncthreads();
  read_a(0, 0, af[0]);

  int chunk = 0, tap = 0;
  // one tap; `cur` (the fragment set it computes from) is a compile-time constant so that af / bf stay in registers
  auto step = [&](int s, auto cur_tag) {
    constexpr int cur = decltype(cur_tag)::value;
    const bool more = s + 1 < nstages;
    const bool has_next_chunk = chunk + 1 < nchunks;
    int nchunk = chunk, ntap = tap + 1;
    if (ntap == 9) { ntap = 0; ++nchunk; }
    // Halo traffic goes BEFORE the next tap's W fetch: re-using h_reg (tap 0) and consuming it (tap 4) both make the compiler
    // wait for everything in flight — here that is only the current tap's own W fragments, which the MFMAs below need anyway.
    if (tap == 4 && has_next_chunk) store_halo((chunk + 1) & 1);
    if (tap == 0 && has_next_chunk) load_halo(chunk + 1);
    if (tap == 8 && has_next_chunk) __syncthreads();          // halo(chunk+1) published; buffer of chunk-1 retired
    // UNCONDITIONAL prefetch (the last tap re-fetches itself): under `if (more)` the two paths merge in front of the MFMAs and
    // the merged wait-count state makes them wait for the loads just issued (vmcnt(4) .. vmcnt(0) instead of vmcnt(5))
    if (!more) { nchunk = chunk; ntap = tap; }
    load_b(nchunk, ntap, bf[cur ^ 1]);
    read_a(nchunk & 1, ntap, af[cur ^ 1]);
    __builtin_amdgcn_sched_barrier(0);                        // next tap's fetches stay ABOVE this tap's matrix work
    if (sizeof(T) == 4) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[cur][j][kk], af[cur][i][kk], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma_stage<T>(bf[cur][j], af[cur][i], acc[i][j]);
    }
    __builtin_amdgcn_sched_barrier(0);
    chunk = nchunk;
    tap = ntap;
  };
  for (int s = 0; s < nstages; s += 2) {
    step(s, std::integral_constant<int, 0>{});
    if (s + 1 < nstages) step(s + 1, std::integral_constant<int, 1>{});
  }

  const int epix = lane & 15;
  const int en = (lane >> 4) * 4;
  const int ox = x0 + epix;
  auto epilogue = [&](auto act_tag) {
  constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + (wn * NT + j) * 16 + en;
    if (n >= p.N) continue;
    float sc[4], sh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool ok = n + q < p.N;
      sc[q] = (ok && p.scale) ? p.scale[n + q] : 1.0f;
      sh[q] = (ok && p.shift) ? p.shift[n + q] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int oy = y0 + wm * MT + i;
      if (oy >= p.H || ox >= p.W) continue;
      const size_t m = (size_t)(b * p.H + oy) * p.W + ox;
      store4<T, ACT>(p, acc[i][j], n, m * p.ldd + n, m * p.ldres, sc, sh);
    }
  }
  };
  CCVPE_ACT_DISPATCH(p.act, epilogue);
}

#endif  // CCVPE_ABLATE

template <typename T, int MT, int NT, int WN, int NW>
static int launch3x3_nw(const IgemmParams& p0, int batch, hipStream_t stream) {
  constexpr int WM = NW / WN;
  constexpr int BM = 16 * MT * WM;
  constexpr int BN = 16 * NT * WN;
  constexpr int TH = BM / 16;
  IgemmParams p = p0;
  p.tiles_x = (p.W + 15) / 16;
  p.tiles_y = (p.H + TH - 1) / TH;
  p.tiles_n = (p.Npad + BN - 1) / BN;
  const long total = (long)p.tiles_x * p.tiles_y * batch * p.tiles_n;
  if (total > 0x7fffffffL) return fail(CCVPE_EINVAL, "conv3x3: grid too large");
  p.tiles_total = (int)total;
#ifdef CCVPE_ABLATE
  p.ablate = getenv("CCVPE_C3_ABLATE") ? atoi(getenv("CCVPE_C3_ABLATE")) : 0;
#endif
#ifdef CCVPE_ABLATE
  if constexpr (NW == 4 && sizeof(T) == 4) {
    if (g_conv3_wreg && (size_t)p.Npad * p.Kpad < (1u << 30)) {        // 32-bit W offsets
      hipLaunchKernelGGL((conv3x3_wreg_kernel<T, MT, NT, WN>), dim3(p.tiles_total), dim3(256), 0, stream, p);
      return check_launch("conv3x3_wreg_kernel");
    }
  }
#endif
  // W by LDS-DMA when the tile is fully inside the packed rows (no row guard possible), a row of taps per stage for the
  // 4-wave form (the 8-wave form lives under a 128-VGPR cap: no room for the second fragment set)
  static bool attr_set[3] = {false, false, false};      // per (T, tile) instantiation of this launcher: one flag per kernel variant
  auto go = [&](int variant, void (*kern)(const IgemmParams), int lds) -> int {
    if (!attr_set[variant] && lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return fail(CCVPE_ELAUNCH, "conv3x3: set smem attr: %s", hipGetErrorString(e));
      attr_set[variant] = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.tiles_total), dim3(64 * NW), lds, stream, p);
    return CCVPE_OK;
  };
  int rc;
  if (p.Npad % BN == 0) {
    if constexpr ((NW == 4 || sizeof(T) == 2) && !(NT == 5 && WN == 1)) {      // (256 x 80 tile: a row of taps per stage measured the same, 117.8 vs 117.0 TF)
      if (g_conv3_tps == 3) rc = go(0, conv3x3_kernel<T, MT, NT, WN, NW, true, 3>, Conv3Geom<T, MT, NT, WN, NW, true, 3>::LDS_BYTES);
      else rc = go(1, conv3x3_kernel<T, MT, NT, WN, NW, true, 1>, Conv3Geom<T, MT, NT, WN, NW, true, 1>::LDS_BYTES);
    } else {
      rc = go(1, conv3x3_kernel<T, MT, NT, WN, NW, true, 1>, Conv3Geom<T, MT, NT, WN, NW, true, 1>::LDS_BYTES);
    }
  } else {
    rc = go(2, conv3x3_kernel<T, MT, NT, WN, NW, false, 1>, Conv3Geom<T, MT, NT, WN, NW, false, 1>::LDS_BYTES);
  }
  if (rc) return rc;
  return check_launch("conv3x3_kernel");
}

template <typename T, int MT, int NT, int WN>
static int launch3x3(const IgemmParams& p0, int batch, hipStream_t stream) {
  // 8 waves (256-pixel tile) when the image is tall enough for the 2x taller tile and the grid still has
  // >= 2 workgroups per CU; only instantiated for the wide-N tiles that dominate the decoder
  // (NT = 5 needs > 128 VGPRs: with the 4-waves/SIMD cap it spills (121 -> 88 TF); uncapped at one
  // workgroup per CU it reaches 119 TF vs 122.5 for the 4-wave form, so only NT = 4 uses 8 waves)
  if constexpr (WN == 2 && NT == 4 && sizeof(T) == 4) {
    constexpr int TH8 = 16 * MT * (8 / WN) / 16;
    constexpr int BN = 16 * NT * WN;
    const long blocks8 = (long)((p0.W + 15) / 16) * ((p0.H + TH8 - 1) / TH8) * batch * ((p0.Npad + BN - 1) / BN);
    if (g_conv3_nw8 && p0.H % TH8 == 0 && blocks8 >= 512) return launch3x3_nw<T, MT, NT, WN, 8>(p0, batch, stream);
  }
  if constexpr (WN == 2 && (NT == 4 || NT == 5) && sizeof(T) == 2) {
    // bf16: 8 waves share one W panel and one halo (256-pixel tile, a row of taps per stage, one workgroup per CU, no VGPR cap).
    // Round 4, isolated layers at B = 64 (tools/gpu/ab_c3.sh): 640 -> 640 at 16 x 16 885 -> 916 TF, 320 -> 320 at 32 x 32 832 -> 870,
    // 1344 -> 640 875 -> 920; 160 -> 160 at 64 x 64 (ONE column tile) 757 -> 734: only layers with >= 2 column tiles take it.
    constexpr int TH8 = 16 * MT * (8 / WN) / 16;
    constexpr int BN = 16 * NT * WN;
    const long tiles_n = (p0.Npad + BN - 1) / BN;
    const long blocks8 = (long)((p0.W + 15) / 16) * ((p0.H + TH8 - 1) / TH8) * batch * tiles_n;
    if (p0.H % TH8 == 0 && blocks8 >= 256 && tiles_n >= 2) return launch3x3_nw<T, MT, NT, WN, 8>(p0, batch, stream);
  }
  return launch3x3_nw<T, MT, NT, WN, 4>(p0, batch, stream);
}

template <typename T>
int conv3x3_dispatch(const IgemmParams& p, int batch, int mt, int nt, int wn, hipStream_t stream) {
#define CCVPE_CASE(MT_, NT_, WN_) \
  if (mt == MT_ && nt == NT_ && wn == WN_) return launch3x3<T, MT_, NT_, WN_>(p, batch, stream);
  CCVPE_CASE(4, 5, 2) CCVPE_CASE(4, 4, 2) CCVPE_CASE(4, 3, 2) CCVPE_CASE(4, 2, 2) CCVPE_CASE(4, 1, 2)
  CCVPE_CASE(4, 5, 1) CCVPE_CASE(4, 3, 1) CCVPE_CASE(4, 1, 1) CCVPE_CASE(2, 7, 1)
#undef CCVPE_CASE
  return fail(CCVPE_EINVAL, "conv3x3: no tile <%d,%d,%d>", mt, nt, wn);
}

}  // namespace ccvpe
