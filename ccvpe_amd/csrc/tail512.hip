// tail512_kernel: the whole 512 x 512 level of a decoder in ONE launch.
//
//   loc: deconv1 (k2 s2) -> conv1.0 3x3 + bias -> ReLU -> conv1.2 3x3 (16 -> 1) + bias            models.py:124-127,319
//   ori: deconv1_ori     -> conv1_ori.0        -> ReLU -> conv1_ori.2 3x3 (16 -> 2) + bias -> F.normalize   models.py:145-148,341
//
// Before: upconv_halo_kernel<T,4,1,1> (folded deconv + conv.0, N = 16) wrote a [B,512,512,16] tensor (1.07 GB in fp32 at
// B = 64) that head_conv_kernel read straight back: 1.23 + 0.45 ms per decoder in the fp32 forward, bound by neither roof.
// Here the 16-channel tensor never leaves the chip — and never even reaches LDS:
//
//   stage 1  (MFMA, the FLOPs)  per output parity (py,px) the folded deconv + conv.0 is a GEMM over LOW-RES positions with
//            2 x 2 taps (upconv_impl.h): K = 4 * c0, N = 16.  A wave owns ONE parity and keeps that parity's whole W
//            (16 x 4*c0) in REGISTERS as MFMA A-fragments (48 VGPRs at c0 = 48): the K loop reads only pixel fragments from
//            an LDS halo tile of x ((TY+2) x (TX+2) low-res pixels, one 64-byte channel chunk at a time, double buffered).
//            No W traffic, no W barrier.  The accumulator of a 16-position tile ends as (lane = position, 4 consecutive
//            channels per lane group): C layout of v_mfma_f32_16x16x4_f32.
//   stage 2  (MFMA, +8 %)       conv.2 is applied in SCATTER form: s[t][pos] = sum_c w2[t][c] * mid[c][pos] for the 9 taps t is
//            one more 16x16 MFMA per tile whose B operand IS the stage-1 accumulator (bias + ReLU applied in place; the k
//            permutation "lane group q supplies channels 4q..4q+3" is exactly the C layout), A = w2 as [tap][channel].
//   stage 3  (LDS gather)       the 9 tap planes s[t] of the tile's (2TY+2) x (2TX+2) mid grid go to LDS (aliasing the dead x
//            buffers); out[Y][X] = b2 + sum_t s[t][Y+ky-1][X+kx-1] is 9 conflict-free ds_read_b32 per output, fixed order.
//
// The tile's mid grid has a one-pixel apron (conv.2 needs mid at Y-1 .. Y+1), so stage 1 computes (TY+1) x (TX+1) positions per
// parity for TY x TX useful ones; positions are numbered linearly and cut into 16-position MFMA tiles with PER-LANE LDS
// addresses, so only the last tile of a parity is ragged.  Mid pixels outside the image are zeros (conv.2's padding); x pixels
// outside the image are zeros of the deconv output (the 9 border classes of shift9 carry the deconv bias, upconv_impl.h).
//
// Element type T = storage AND matrix type of stage 1: float (v_mfma_f32_16x16x4_f32, exact fp32) or bf16
// (v_mfma_f32_16x16x32_bf16, fp32 accumulate); stage 2/3 are fp32 in both.  Same LDS byte geometry for both (a pixel row = one
// 64-byte chunk = 16 fp32 / 32 bf16 channels, pitch 80 bytes).
//
// SPLIT (T = float storage; the fp32 tail of the bf16 STORAGE path, never the fp32 path): stage 1 on the bf16 matrix cores at
// fp32-class accuracy.  Both operands are split into two bf16 planes, v = hi + lo with hi = bf16(v), lo = bf16(v - hi)
// (|v - hi - lo| <= 2^-17 |v|), and x.w is evaluated as hi.hi + lo.hi + hi.lo (the dropped lo.lo term is 2^-16 relative), fp32
// accumulate: a pixel row in LDS is [hi 0..15 | lo 0..15] (the same 64 bytes as 16 fp32), ONE ds_read_b128 per fragment,
// and per 16 channels two v_mfma_f32_16x16x32_bf16 — A = [w_hi | w_hi] then A = [w_lo | 0] against B = [x_hi | x_lo] — instead
// of four v_mfma_f32_16x16x4_f32: 32 instead of 128 matrix cycles.  Logit error ~1e-5 of scale (tests/test_ops_gpu.py) where
// bf16 storage of the same operands gives 4e-3: the arg-max of the bf16 path is decided exactly as with the fp32 tail.
#include "conv_common.h"

namespace ccvpe {

struct TailParams {
  const void* x;
  const void* w;
  const float* shift9;
  const float* w2;
  const float* b2;
  float* out;
  float* smx;        // cout == 1, optional: per (tile, wave) softmax partials (max, sum exp(l - max)) of the logits this wave wrote
  int H1, W1, c0, ld0, Kpad, normalize;
  int tiles_x, tiles_y, tiles_total, tiles_per_wg;
};

template <typename T, int COUT, int NCH, int TY, int TX, int WPP, bool SPLIT = false, bool PERSIST = false>
struct TailGeom {
  static constexpr int NW = 4 * WPP;
  static constexpr int NTHR = 64 * NW;
  static constexpr int PW = TX + 1;
  static constexpr int P = (TY + 1) * PW;            // stage-1 positions per parity
  static constexpr int NTILE = (P + 15) / 16;
  static constexpr int TPW = (NTILE + WPP - 1) / WPP; // 16-position tiles per wave
  static constexpr int HR = TY + 2, HC = TX + 2, HPX = HR * HC;
  // Staged pixel row: 64 bytes of data in a slot of LD floats, HCP slots per halo row.  The stage-1 fragment read is PER-LANE
  // addressed (position pt -> (pt / (TX+1), pt % (TX+1)): the 16 lanes of a tile usually straddle two halo rows), so the panel
  // swizzle of the GEMM kernels does not apply; in the bank model (tools/lds_layout.py, searched over pitch x slots per row x
  // swizzle) the 80-byte slot in rows of TX + 2 costs 7.8 cycles per ds_read_b128 (ideal 4) and ONE geometry is conflict-free:
  // 96-byte slots in rows of 25 (SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS of the bf16-path tails: 1.5-2.1 with the LDS pipe
  // 35 % busy, tools/gpu/lds_pmc.sh).  It costs 67 % more LDS per chunk buffer, so the latency-shaped 8 x 16 tiles of the bf16
  // storage path take it (48 KB per workgroup) and the matrix-bound 16 x 16 fp32 tile (which would need 86 KB) keeps the 80-byte slot.
  static constexpr bool WIDE_SLOT = TY == 8 && TX == 16 && (SPLIT || sizeof(T) == 2);
  static constexpr int LD = WIDE_SLOT ? 24 : 20;     // floats per staged pixel slot
  static constexpr int HCP = WIDE_SLOT ? 25 : HC;    // slots per halo row
  static constexpr int XBUF = HR * HCP * LD;         // floats per x chunk buffer
  static constexpr int NXB = NCH > 1 ? 2 : 1;
  static constexpr int SH = 2 * TY + 2, SW = 2 * TX + 2, SPL = SH * SW;   // one tap plane of the mid grid
  static constexpr int NPL = 10;                     // 9 tap planes + one junk plane (lanes whose tap index is >= 9 store there)
  static constexpr int MAIN = (NXB * XBUF > NPL * SPL) ? NXB * XBUF : NPL * SPL;
  static constexpr int LDS_BYTES = (MAIN + 9 * 16) * 4;
  static constexpr int NOUT = (4 * TY * TX) / NTHR;  // outputs per thread in stage 3
};

// hi / lo bf16 planes of 4 fp32 values (SPLIT mode)
__device__ __forceinline__ void split4(f32x4 v, bf16x4& hi, bf16x4& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hi[i] = (bf16_t)v[i];
    lo[i] = (bf16_t)(v[i] - (float)hi[i]);
  }
}

template <typename T, int COUT, int NCH, int TY, int TX, int WPP, bool SPLIT, bool PERSIST>
__global__ __launch_bounds__(256 * WPP, 2) void tail512_kernel(const TailParams p) {
  static_assert(!SPLIT || sizeof(T) == 4, "SPLIT reads fp32 operands");
  using G = TailGeom<T, COUT, NCH, TY, TX, WPP, SPLIT, PERSIST>;
  constexpr int E = ElemTraits<T>::E;
  constexpr int SK = 4 * E;                          // channels per 64-byte chunk
  constexpr int NTHR = G::NTHR, PW = G::PW, P = G::P, TPW = G::TPW;
  constexpr int HC = G::HC, HCP = G::HCP, HPX = G::HPX, LD = G::LD, XBUF = G::XBUF;
  constexpr int SW = G::SW, SPL = G::SPL;
  constexpr int H_IT = (HPX * 4 + NTHR - 1) / NTHR;
  static_assert((4 * TY * TX) % NTHR == 0, "stage 3 hands every thread the same number of outputs");

  extern __shared__ __attribute__((aligned(16))) float t5_sm[];
  float* Xs = t5_sm;                                 // [NXB][HR][HCP][LD]   (stage 1)
  float* Sp = t5_sm;                                 // [9][SH][SW]      (stage 2/3, aliases Xs)
  float* Sh9 = t5_sm + G::MAIN;                      // [9][16]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = sgpr(tid >> 6);
  const int par = wave & 3, wsub = wave >> 2;
  const int py = par >> 1, px = par & 1;
  const int pix = lane & 15, q = lane >> 4;

  // PERSIST: workgroup w walks the tiles [w * tpw, (w + 1) * tpw) of the XCD-aware order (neighbouring tiles share halo rows in
  // one L2).  W, w2 and the bias table are loaded ONCE per workgroup; the first halo chunk of tile t+1 is requested before
  // stage 2 of tile t, so its HBM / L2 latency sits under the previous tile's epilogue.  That is what the latency-shaped
  // instantiations want (bf16 and SPLIT: 8 x 16 tiles, a tile's matrix work is ~2 us): SPLIT 0.86 -> 0.62 ms, bf16 ori 0.44 ->
  // 0.40, bf16 loc 0.37 -> 0.35 at B = 64.  The MFMA-bound fp32 instantiations run ONE 16 x 16 tile per workgroup (!PERSIST): the
  // prefetch registers live across stage 2 / 3 pushed them into scratch (1.10 -> 1.58 ms), and with the smaller tile that fits
  // (1.26 ms) the larger apron costs more than the overlap gains.
  const int wg = xcd_tile(blockIdx.x, gridDim.x);
  const int t_begin = wg * p.tiles_per_wg;
  const int t_end = min(t_begin + p.tiles_per_wg, p.tiles_total);
  const int tiles_xy = p.tiles_x * p.tiles_y;
  int b, y0, x0;                                     // sample and low-res origin of the tile being COMPUTED
  auto tile_coords = [&](int t, int& tb, int& ty0, int& tx0) {
    tb = t / tiles_xy;
    const int r = t - tb * tiles_xy;
    const int ty = r / p.tiles_x;
    ty0 = ty * TY;
    tx0 = (r - ty * p.tiles_x) * TX;
  };
  const int H2 = 2 * p.H1, W2 = 2 * p.W1;
  const T* xg = reinterpret_cast<const T*>(p.x);

  for (int i = tid; i < 9 * 16; i += NTHR) Sh9[i] = p.shift9[i];

  // ---- W of this wave's parity: MFMA A-fragments in registers ------------------------------------------------------
  constexpr int NWR = SPLIT ? 2 : 1;                 // SPLIT: [0] = [w_hi | w_hi], [1] = [w_lo | 0]
  f32x4 wr[4][NCH][NWR];
  if constexpr (!SPLIT) {
    const T* wp = reinterpret_cast<const T*>(p.w) + ((size_t)par * 16 + pix) * p.Kpad;
#pragma unroll
    for (int tap = 0; tap < 4; ++tap)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = c * SK + q * E;                 // (all 4 * NCH loads in flight together: clamped address + mask, no branch)
        wr[tap][c][0] = *reinterpret_cast<const f32x4*>(wp + (ch < p.c0 ? tap * p.c0 + ch : 0));
      }
#pragma unroll
    for (int tap = 0; tap < 4; ++tap)
#pragma unroll
      for (int c = 0; c < NCH; ++c) wr[tap][c][0] = keep_if(wr[tap][c][0], c * SK + q * E < p.c0);
  } else {
    // lane (n = pix, q): channels 8*(q&1) .. +7 of the chunk; lane groups 2, 3 repeat the hi plane (they meet x_lo) and hold
    // zeros in the lo fragment
    const float* wp = reinterpret_cast<const float*>(p.w) + ((size_t)par * 16 + pix) * p.Kpad;
#pragma unroll
    for (int tap = 0; tap < 4; ++tap)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = c * 16 + (q & 1) * 8;
        const bool ok = ch < p.c0;                     // c0 % 8 == 0: the 8 channels are in range together
        const f32x4 v0 = keep_if(*reinterpret_cast<const f32x4*>(wp + (ok ? tap * p.c0 + ch : 0)), ok);
        const f32x4 v1 = keep_if(*reinterpret_cast<const f32x4*>(wp + (ok ? tap * p.c0 + ch + 4 : 0)), ok);
        bf16x4 h0, l0, h1, l1;
        split4(v0, h0, l0);
        split4(v1, h1, l1);
        bf16x8 hi, lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) { hi[i] = h0[i]; hi[i + 4] = h1[i]; lo[i] = l0[i]; lo[i + 4] = l1[i]; }
        wr[tap][c][0] = __builtin_bit_cast(f32x4, hi);
        wr[tap][c][1] = keep_if(__builtin_bit_cast(f32x4, lo), q < 2);
      }
  }
  // conv.2 as [tap][channel]: lane (tap = pix, q) holds channels 4q .. 4q+3 of output o
  f32x4 w2f[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) {
    w2f[o] = keep_if(*reinterpret_cast<const f32x4*>(p.w2 + (o * 9 + min(pix, 8)) * 16 + q * 4), pix < 9);
  }

  // ---- halo staging: thread -> (halo pixel, 16-byte piece); coordinates are recomputed per chunk (NCH <= 3 times) rather
  // than held in registers across the matrix loop ------------------------------------------------------------------------
  const int hsub = tid & 3;                          // NTHR % 4 == 0: the piece index is the same for every `it`
  // `tidv` is the thread index behind an opaque move that is re-executed at the top of every tile: all the address arithmetic
  // below (halo staging, fragment bases, tap planes, outputs) is tile-INVARIANT, and hoisted out of the tile loop it would
  // sit in ~80 registers across the matrix loop, which then spills (measured: 250-600 bytes of scratch per lane).
  int tidv = tid;
  f32x4 h_reg[H_IT];
  unsigned h_keep = 0;
  const int ld0s = sgpr(p.ld0);
  auto load_halo = [&](int c, int hb, int hy0, int hx0) {   // raw loads from clamped addresses, masked at the LDS store (STAGING RULE)
    const int ch = c * SK + hsub * E;
    h_keep = 0;
#pragma unroll
    for (int it = 0; it < H_IT; ++it) {
      const int pxl = (tidv + NTHR * it) >> 2;
      const int hy = pxl / HC, hx = pxl - hy * HC;
      const int iy = hy0 - 1 + hy, ix = hx0 - 1 + hx;
      const bool ok = pxl < HPX && (unsigned)iy < (unsigned)p.H1 && (unsigned)ix < (unsigned)p.W1 && ch < p.c0;
      h_reg[it] = *reinterpret_cast<const f32x4*>(xg + (ok ? (size_t)((hb * p.H1 + iy) * p.W1 + ix) * ld0s + ch : 0));
      h_keep |= ok ? (1u << it) : 0u;
    }
  };
  auto store_halo = [&](int buf) {
#pragma unroll
    for (int it = 0; it < H_IT; ++it) {
      const int pxl = (tidv + NTHR * it) >> 2;
      const int slot = HCP == HC ? pxl : (pxl / HC) * HCP + (pxl % HC);
      const f32x4 v = keep_if(h_reg[it], (h_keep >> it) & 1u);
      if constexpr (!SPLIT) {
        if (pxl < HPX) *reinterpret_cast<f32x4*>(Xs + buf * XBUF + slot * LD + hsub * 4) = v;
      } else {                                       // row = [hi 0..15 | lo 0..15] bf16: this piece's 4 channels -> 8 + 8 bytes
        bf16x4 hi, lo;
        split4(v, hi, lo);
        if (pxl < HPX) {
          float* row = Xs + buf * XBUF + slot * LD;
          *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(row) + hsub * 4) = hi;
          *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(row) + 16 + hsub * 4) = lo;
        }
      }
    }
  };

  if (t_begin >= t_end) return;
  tile_coords(t_begin, b, y0, x0);
  load_halo(0, b, y0, x0);
  for (int t = t_begin; t < t_end; ++t) {
  asm volatile("v_mov_b32 %0, %0" : "+v"(tidv));
  const int pixv = tidv & 15, qv = (tidv >> 4) & 3;
  // ---- stage 1 ------------------------------------------------------------------------------------------------------
  // position of (tile i, lane): pt = (wsub*TPW + i)*16 + pix -> (iy, ix) in the (TY+1) x (TX+1) grid of this parity; its four
  // taps are halo pixels (iy + du, ix + dv) for BOTH parities (the parity only moves the grid's origin)
  int fbase[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int pc = min((wsub * TPW + i) * 16 + pixv, P - 1);
    const int iy = pc / PW, ix = pc - iy * PW;
    fbase[i] = (iy * HCP + ix) * LD + qv * 4;
  }
  f32x4 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  store_halo(0);
  __syncthreads();
  // The matrix loop is software-pipelined BY HAND: left to itself the compiler (at the register limit) issues each fragment read
  // directly in front of the four MFMAs that consume it (ds_read, s_waitcnt lgkmcnt(0), 4 x v_mfma: an exposed LDS round trip
  // per 128 cycles of matrix work) and sinks the next chunk's global loads below the MFMAs.  Here: the next chunk's halo loads
  // go out first; a step = (pair of position tiles, tap); the fragments of step s+1 are requested before the MFMAs of step s;
  // the two tiles of a pair alternate so that no MFMA waits for the accumulator of its predecessor.
  constexpr int NPAIR = (TPW + 1) / 2, NS = NPAIR * 4;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (c + 1 < NCH) load_halo(c + 1, b, y0, x0);
    __builtin_amdgcn_sched_barrier(0);
    const float* xb = Xs + (c & 1) * XBUF;
    f32x4 fr[2][2];
    auto read_step = [&](int s, f32x4* dst) {
      const int g = s >> 2, tap = s & 3;
      const int toff = ((tap >> 1) * HCP + (tap & 1)) * LD;
      dst[0] = *reinterpret_cast<const f32x4*>(xb + fbase[2 * g] + toff);
      if (2 * g + 1 < TPW) dst[1] = *reinterpret_cast<const f32x4*>(xb + fbase[2 * g + 1] + toff);
    };
    read_step(0, fr[0]);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int g = s >> 2, tap = s & 3;
      if (s + 1 < NS) read_step(s + 1, fr[(s + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);             // the reads above stay above this step's MFMAs
      const f32x4 wv = wr[tap][c][0];
      const f32x4 f0 = fr[s & 1][0], f1 = fr[s & 1][1];
      if constexpr (SPLIT) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          acc[2 * g] = mfma_stage<bf16_t>(wr[tap][c][h], f0, acc[2 * g]);
          if (2 * g + 1 < TPW) acc[2 * g + 1] = mfma_stage<bf16_t>(wr[tap][c][h], f1, acc[2 * g + 1]);
        }
      } else if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          acc[2 * g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[kk], f0[kk], acc[2 * g], 0, 0, 0);
          if (2 * g + 1 < TPW) acc[2 * g + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[kk], f1[kk], acc[2 * g + 1], 0, 0, 0);
        }
      } else {
        acc[2 * g] = mfma_stage<T>(wv, f0, acc[2 * g]);
        if (2 * g + 1 < TPW) acc[2 * g + 1] = mfma_stage<T>(wv, f1, acc[2 * g + 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (c + 1 < NCH) {
      store_halo((c + 1) & 1);
      __syncthreads();
    }
  }
  // the next tile's first chunk: in flight under this tile's stage 2 / 3 (the last tile re-fetches its own: unconditional)
  if constexpr (PERSIST) {
    int nb, ny0, nx0;
    tile_coords(min(t + 1, t_end - 1), nb, ny0, nx0);
    load_halo(0, nb, ny0, nx0);
  }
  __syncthreads();                                   // every wave is done with the x buffers: the tap planes alias them
  __builtin_amdgcn_sched_barrier(0);                 // (keeps the epilogue's address arithmetic out of the matrix loop's registers)

  // ---- stage 2 + 3 ----------------------------------------------------------------------------------------------------
  // bias (border class of the mid pixel) + ReLU in place; mid pixels outside the image are conv.2's zero padding.
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int pc = min((wsub * TPW + i) * 16 + pixv, P - 1);
    const int iy = pc / PW, ix = pc - iy * PW;
    const int Y = 2 * (y0 + iy) - py, X = 2 * (x0 + ix) - px;
    const bool inside = (unsigned)Y < (unsigned)H2 && (unsigned)X < (unsigned)W2;
    const int rc = Y <= 0 ? 0 : (Y >= H2 - 1 ? 2 : 1);
    const int cc = X <= 0 ? 0 : (X >= W2 - 1 ? 2 : 1);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(Sh9 + (rc * 3 + cc) * 16 + qv * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[i][r] = inside ? fmaxf(acc[i][r] + sh[r], 0.f) : 0.f;
  }
  float res[COUT][G::NOUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) {
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) s = __builtin_amdgcn_mfma_f32_16x16x4f32(w2f[o][r], acc[i][r], s, 0, 0, 0);
      // lane (q, pix) now holds taps 4q .. 4q+3 of position pix
      const int pt = (wsub * TPW + i) * 16 + pixv;
      const int iy = pt / PW, ix = pt - iy * PW;
      float* dst = Sp + (2 * iy + 1 - py) * SW + (2 * ix + 1 - px);
      if (pt < P) {                                  // (only the last tile of a parity is ragged)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[min(4 * qv + r, 9) * SPL] = s[r];     // tap index >= 9: the junk plane
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < G::NOUT; ++k) {
      const int e = tidv + NTHR * k;
      const int Yo = e / (2 * TX), Xo = e - Yo * (2 * TX);
      float v = p.b2[o];
#pragma unroll
      for (int t = 0; t < 9; ++t) v += Sp[t * SPL + (Yo + t / 3) * SW + Xo + t % 3];
      res[o][k] = v;
    }
    if (o + 1 < COUT) __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < G::NOUT; ++k) {
    const int e = tidv + NTHR * k;
    const int Yo = e / (2 * TX), Xo = e - Yo * (2 * TX);
    if (COUT == 2 && p.normalize) {                  // F.normalize(p=2, dim=1, eps=1e-12)
      const float n = fmaxf(sqrtf(res[0][k] * res[0][k] + res[COUT - 1][k] * res[COUT - 1][k]), 1e-12f);
      res[0][k] /= n;
      res[COUT - 1][k] /= n;
    }
#pragma unroll
    for (int o = 0; o < COUT; ++o)
      p.out[((size_t)(b * COUT + o) * H2 + 2 * y0 + Yo) * W2 + 2 * x0 + Xo] = res[o][k];
  }
  if (COUT == 1 && p.smx) {
    // softmax partials of the heat-map head (models.py:319-320), one pair per (tile, wave): the row-wide max / sum sweeps of
    // softmax_rows_kernel (64 workgroups, one CU's bandwidth each: 108 us at B = 64) become a 2 KB merge in
    // softmax_apply_kernel.  Wave-local: no LDS, no barrier.
    float m = res[0][0];
#pragma unroll
    for (int k = 1; k < G::NOUT; ++k) m = fmaxf(m, res[0][k]);
    m = wave_max(m);
    float e = 0.f;
#pragma unroll
    for (int k = 0; k < G::NOUT; ++k) e += expf(res[0][k] - m);
    e = wave_sum(e);
    if (lane == 0) {
      float* dst = p.smx + ((size_t)t * G::NW + wave) * 2;       // tile index t is sample-major: [B][tiles][waves][2]
      dst[0] = m;
      dst[1] = e;
    }
  }
  if constexpr (!PERSIST) break;                     // one tile per workgroup
  tile_coords(min(t + 1, t_end - 1), b, y0, x0);
  __syncthreads();                                   // the tap planes are dead: the next tile's halo may overwrite them
  }  // tile loop
}

template <typename T, int COUT, int NCH, int TY, int TX, int WPP, bool SPLIT = false, bool PERSIST = false>
static int launch_tail(const TailParams& p0, int batch, hipStream_t stream) {
  using G = TailGeom<T, COUT, NCH, TY, TX, WPP, SPLIT, PERSIST>;
  TailParams p = p0;
  if (p.W1 % TX || p.H1 % TY) return fail(CCVPE_EINVAL, "tail512: %d x %d is not a multiple of the %d x %d tile", p.H1, p.W1, TY, TX);
  p.tiles_x = p.W1 / TX;
  p.tiles_y = p.H1 / TY;
  const long total = (long)p.tiles_x * p.tiles_y * batch;
  if (total <= 0 || total > 0x7fffffffL) return fail(CCVPE_EINVAL, "tail512: bad grid");
  p.tiles_total = (int)total;
  // persistent: about 8 workgroups per CU slot (2 resident per CU) so that the tail of the launch stays short
  const int max_wgs = 256 * 2 * 8;
  p.tiles_per_wg = PERSIST ? (int)((total + max_wgs - 1) / max_wgs) : 1;
  const int wgs = (int)((total + p.tiles_per_wg - 1) / p.tiles_per_wg);
  static bool attr_set = false;                      // one flag per instantiation
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)tail512_kernel<T, COUT, NCH, TY, TX, WPP, SPLIT, PERSIST>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    if (e != hipSuccess) return fail(CCVPE_ELAUNCH, "tail512: set smem attr: %s", hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL((tail512_kernel<T, COUT, NCH, TY, TX, WPP, SPLIT, PERSIST>), dim3(wgs), dim3(G::NTHR), G::LDS_BYTES, stream, p);
  return check_launch("tail512_kernel");
}

template <typename T>
static int tail_any(const ccvpe_tail_desc* d, void* stream) {
  constexpr int E = ElemTraits<T>::E;
  constexpr int SK = 4 * E;
  if (!d) return fail(CCVPE_EINVAL, "tail512: null desc");
  if (d->c0 <= 0 || d->c0 % 8 || d->ld0 % E || d->ld0 < d->c0) return fail(CCVPE_EINVAL, "tail512: c0 %% 8, ld0 %% %d, ld0 >= c0 required", E);
  if (d->kpad < 4 * d->c0) return fail(CCVPE_EINVAL, "tail512: kpad %d < 4 * c0", d->kpad);
  if (d->cout != 1 && d->cout != 2) return fail(CCVPE_EINVAL, "tail512: cout must be 1 or 2");
  if (!d->x || !d->w || !d->shift9 || !d->w2 || !d->b2 || !d->out) return fail(CCVPE_EINVAL, "tail512: null pointer");
  if (!aligned16(d->x) || !aligned16(d->w) || !aligned16(d->w2) || !aligned16(d->shift9))
    return fail(CCVPE_EINVAL, "tail512: pointers must be 16-byte aligned");
  if (d->batch <= 0 || d->h1 <= 0 || d->w1 <= 0 || (long)d->batch * d->h1 * d->w1 > 0x1fffffffL) return fail(CCVPE_EINVAL, "tail512: bad shape");
  if (d->h1 % 16 || d->w1 % 16) return fail(CCVPE_EINVAL, "tail512: h1 and w1 must be multiples of 16 (got %d x %d)", d->h1, d->w1);
  TailParams p;
  p.x = d->x; p.w = d->w; p.shift9 = d->shift9; p.w2 = d->w2; p.b2 = d->b2; p.out = d->out;
  p.smx = d->cout == 1 ? d->softmax_partial : nullptr;
  p.H1 = d->h1; p.W1 = d->w1; p.c0 = d->c0; p.ld0 = d->ld0; p.Kpad = d->kpad; p.normalize = d->normalize;
  p.tiles_x = p.tiles_y = p.tiles_total = p.tiles_per_wg = 0;
  const int nch = (d->c0 + SK - 1) / SK;
  hipStream_t st = (hipStream_t)stream;
  // Workgroup tile, fp32: 16 x 16 low-res pixels, one wave per parity, one tile per workgroup, two workgroups per CU (a workgroup's
  // stage 2/3 runs under the other's matrix loop).  Measured against 32 x 16 with two waves per parity and one workgroup per CU
  // (B = 64, tools/tail_probe.py): fp32 loc 1.11 vs 1.23 ms, fp32 ori 0.95 vs 1.01 — the smaller apron does not pay for the lost
  // overlap.  bf16 / SPLIT: 8 x 16 tiles walked by persistent workgroups (see the kernel's PERSIST comment).
#define CCVPE_TAIL(COUT_, NCH_)                                                                                \
  if (d->cout == COUT_ && nch == NCH_) {                                                                       \
    if constexpr (sizeof(T) == 4) return launch_tail<T, COUT_, NCH_, 16, 16, 1, false, false>(p, d->batch, st); \
    else return launch_tail<T, COUT_, NCH_, 8, 16, 1, false, true>(p, d->batch, st);                           \
  }
  if constexpr (sizeof(T) == 4) {
    if (d->split) {                                           // fp32 operands, bf16 hi/lo matrix arithmetic (bf16 storage path only)
      // (8 x 16 tile: the hi and lo W fragments are 96 registers at 41 channels, so the accumulators get half the tile)
      if (d->cout == 1 && nch == 2) return launch_tail<T, 1, 2, 8, 16, 1, true, true>(p, d->batch, st);
      if (d->cout == 1 && nch == 3) return launch_tail<T, 1, 3, 8, 16, 1, true, true>(p, d->batch, st);
      return fail(CCVPE_EINVAL, "tail512: split mode is instantiated for cout 1, 17..48 channels");
    }
    CCVPE_TAIL(1, 2) CCVPE_TAIL(1, 3) CCVPE_TAIL(2, 2)      // loc: 33 / 41 channels (ld 40 / 48); ori: 32
  } else {
    if (d->split) return fail(CCVPE_EINVAL, "tail512: split mode takes fp32 operands");
    CCVPE_TAIL(1, 1) CCVPE_TAIL(1, 2) CCVPE_TAIL(2, 1)      // bf16 chunks hold 32 channels
  }
#undef CCVPE_TAIL
  return fail(CCVPE_EINVAL, "tail512: (cout %d, c0 %d) not instantiated (fp32: cout 1 with c0 <= 48, cout 2 with c0 <= 32; bf16: cout 1 "
                            "with c0 <= 64, cout 2 with c0 <= 32)", d->cout, d->c0);
}

// (tile rows, tile columns, waves) of the instantiation a descriptor runs: the softmax-partial count depends on it
template <typename T>
static int tail_partials(const ccvpe_tail_desc* d) {
  if (!d || d->h1 <= 0 || d->w1 <= 0) return fail(CCVPE_EINVAL, "tail512_partials: bad desc");
  const bool small = sizeof(T) == 2 || d->split;                 // 8 x 16 tiles (bf16, split), 16 x 16 otherwise; 4 waves
  const int ty = small ? 8 : 16;
  if (d->h1 % ty || d->w1 % 16) return fail(CCVPE_EINVAL, "tail512_partials: shape not tiled");
  return (d->h1 / ty) * (d->w1 / 16) * 4;
}

}  // namespace ccvpe

using namespace ccvpe;

extern "C" int ccvpe_tail512_partials(const ccvpe_tail_desc* desc, int is_bf16) {
  return is_bf16 ? tail_partials<bf16_t>(desc) : tail_partials<float>(desc);
}
extern "C" int ccvpe_tail512_f32(const ccvpe_tail_desc* desc, void* stream) { return tail_any<float>(desc, stream); }
extern "C" int ccvpe_tail512_bf16(const ccvpe_tail_desc* desc, void* stream) { return tail_any<bf16_t>(desc, stream); }
