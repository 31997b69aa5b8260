// Narrow decoder levels in bf16 storage: kernels that keep their WEIGHTS IN REGISTERS and stream activation halo tiles through LDS.
// Included by narrow_bf16.hip.
//
// The last decoder levels of both branches (models.py:121-127, 142-148: 256 x 256 and 512 x 512 pixels, 16-40 output channels,
// K = 360-500) are long-M / short-K / narrow-N problems.  The tiled GEMM kernels (conv3x3_kernel, upconv_*_kernel) re-fetch the
// whole weight matrix per output tile — 55 KB of W panels against 26 KB of activations for the 40 -> 40 layer — pad every tap's
// channels to a 32-channel chunk (40 -> 64: 1.6 x the matrix work) and live between a prologue and an epilogue: round 4 traced
// them at 0.3-0.35 PF and 3-5 x off the HBM roof.  Here:
//   * K is FLAT over (tap, 8-channel octet): a 32-wide MFMA k-step takes four consecutive octets wherever they fall, so a layer
//     with 5 octets per tap runs 12 k-steps instead of 18 (the packed weights already have this K order);
//   * the weights of the layer — 12 k-steps x 3 column tiles x 4 registers = 144 VGPRs for 40 -> 40 — are loaded ONCE per
//     workgroup into registers as MFMA operands (one wave per SIMD, up to 512 registers per lane) and stay there: no W traffic,
//     no W panel barriers, LDS carries activations only;
//   * workgroups are PERSISTENT (one per CU) and walk the tiles; the halo tile of the tile after next is requested by LDS-DMA
//     (global_load_lds_dwordx4, lane-linear: the LDS image is [pixel][octet] in DMA piece order) right after the barrier that
//     frees its buffer, a whole tile of matrix work before it is needed; pieces outside the image are zeroed by the lane that
//     would have requested them;
//   * one barrier per tile, and the epilogue of a tile runs AFTER it: its stores are in flight under the next tile's matrix work
//     (waited for at the barrier, they were the largest single cost of the first version);
//   * the tile loop is BRANCH-FREE for interior tiles: output type and channel count are compile-time, tile coordinates come from
//     multiply-shift divisions, the epilogue pairs column tiles through v_permlane32_swap so that a lane stores 16 bytes.  (First
//     version, ablation build: with halo requests, stores, matrix instructions and fragment reads all removed the 40 -> 40 layer
//     still took 114 of 286 us — ~1 000 instructions of per-element branches and integer divisions per tile.)
#pragma once
#include "conv_common.h"

namespace ccvpe {

// ---- exact unsigned division by a runtime constant: q = (n * m) >> 32 with m = ceil(2^32 / d), exact for n * d < 2^32 ---------
struct FastDiv { unsigned d, m; };
static inline FastDiv make_fastdiv(unsigned d) { return FastDiv{d, d > 1 ? (unsigned)((0x100000000ull + d - 1) / d) : 0u}; }
__device__ __forceinline__ unsigned fd_div(unsigned n, FastDiv f) { return f.d > 1 ? __umulhi(n, f.m) : n; }

struct TileIndex {                     // tile t (XCD-aware order) -> (sample, tile row, tile column)
  FastDiv dx, dy;                      // tiles per row, tile rows per sample
  int total, q8, r8;                   // xcd_tile(): total / 8, total % 8
};
static inline TileIndex make_tile_index(int tiles_x, int tiles_y, int total) {
  return TileIndex{make_fastdiv((unsigned)tiles_x), make_fastdiv((unsigned)tiles_y), total, total / 8, total % 8};
}
__device__ __forceinline__ void tile_decode(const TileIndex& ti, int t, int& b, int& ty, int& tx) {
  const int xcd = t & 7, loc = t >> 3;
  const unsigned ts = (unsigned)((xcd < ti.r8 ? xcd * (ti.q8 + 1) : ti.r8 * (ti.q8 + 1) + (xcd - ti.r8) * ti.q8) + loc);
  const unsigned r = fd_div(ts, ti.dx);
  tx = (int)(ts - r * ti.dx.d);
  const unsigned bb = fd_div(r, ti.dy);
  ty = (int)(r - bb * ti.dy.d);
  b = (int)bb;
}

// one LDS-DMA request of 64 x 16 bytes: lane l's 16 bytes from sbase + voff land at lds + 16 l (lds wave-uniform)
__device__ __forceinline__ void dma16(unsigned lds, unsigned voff, const char* sbase) {
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(voff), "s"(sbase) : "memory", "m0");
}

// Pixel pitch of a halo image in 16-byte slots: ODD (an even channel-octet count gets one unused slot per pixel).  The 16 lanes of
// a fragment that read the same octet of 16 consecutive pixels then hit 16 different slots mod 16 for any window base, and the two
// lane sets a ds_read_b128 group mixes ({0-3, 12-15} of one octet, {4-11} of another) are complementary halves of the 16 (measured
// with the even pitch of 32 / 64 channels: 4- / 8-way conflicts, the kernel slower than the tiled one).
constexpr int odd_pitch(int cpt) { return cpt | 1; }

// Output channel of MFMA row rho = 4 q + r of column tile t.  NOT 16 t + rho: lane groups q and q ^ 2 (lanes 32 apart) hold
// ADJACENT channel quads, so that one v_permlane32_swap per register gives a lane 8 consecutive channels = one 16-byte bf16 store.
__device__ __forceinline__ int chan_of_row(int t, int rho) { return 16 * t + 8 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3); }
__device__ __forceinline__ int chan_of_quad(int t, int q) { return 16 * t + 8 * (q & 1) + 4 * (q >> 1); }

#ifdef CCVPE_ABLATE
#define CCVPE_NARROW_ABL(p) const int abl = (p).ablate
static int narrow_ablate_env() { const char* e = getenv("CCVPE_NARROW_ABLATE"); return e ? atoi(e) : 0; }
#else
#define CCVPE_NARROW_ABL(p) constexpr int abl = 0
static int narrow_ablate_env() { return 0; }
#endif

// Epilogue stores of one pixel row: acc[n] (this lane's quad of column tile n, bias / shift and activation already applied).
// bf16: column tiles are paired through v_permlane32_swap — lanes q < 2 store 8 channels of the even tile, lanes q >= 2 8 channels
// of the odd tile; an unpaired last tile is stored by the lanes q < 2.  fp32: one 16-byte store per quad.  N % 8 == 0.
// t0: first column tile of v[] (a wave that owns the column tiles t0 .. t0 + NT - 1 of a wider layer)
template <int NT, bool F32OUT>
__device__ __forceinline__ void store_row(void* dst, size_t o, const f32x4* v, int q, int N, int t0 = 0) {
  if constexpr (F32OUT) {
    float* dp = reinterpret_cast<float*>(dst) + o;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int ch = chan_of_quad(t0 + n, q);
      if (ch < N) *reinterpret_cast<f32x4*>(dp + ch) = v[n];
    }
  } else {
    bf16_t* dp = reinterpret_cast<bf16_t*>(dst) + o;
#pragma unroll
    for (int n = 0; n + 1 < NT; n += 2) {
      // after the swaps: 8 consecutive channels (lo, hi) of tile n (q < 2) / n + 1 (q >= 2).  Plain float scalars on purpose:
      // __builtin_bit_cast on an ELEMENT of an ext_vector_type compiled to code that used component 0 for every r
      // (tools/micro/store_row_check.hip caught it; the same compiler defect as the masked-select note in DESIGN section 4)
      float lo[4], hi[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = v[n][r], b = v[n + 1][r];
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        lo[r] = __uint_as_float(sw[0]);
        hi[r] = __uint_as_float(sw[1]);
      }
      bf16x8 ob;
#pragma unroll
      for (int r = 0; r < 4; ++r) { ob[r] = (bf16_t)lo[r]; ob[r + 4] = (bf16_t)hi[r]; }
      const int ch = 16 * (t0 + n + (q >> 1)) + 8 * (q & 1);
      if (ch < N) *reinterpret_cast<bf16x8*>(dp + ch) = ob;
    }
    if constexpr (NT & 1) {
      constexpr int n = NT - 1;
      float own[4], hi[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        own[r] = v[n][r];
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(own[r]), __float_as_uint(own[r]), false, false);
        hi[r] = __uint_as_float(sw[1]);               // lanes q < 2: the partner's (q + 2) quad = channels + 4 .. + 7
      }
      bf16x8 ob;
#pragma unroll
      for (int r = 0; r < 4; ++r) { ob[r] = (bf16_t)own[r]; ob[r + 4] = (bf16_t)hi[r]; }
      const int ch = 16 * (t0 + n) + 8 * (q & 1);
      if (q < 2 && ch < N) *reinterpret_cast<bf16x8*>(dp + ch) = ob;
    }
  }
}

// =====================================================================================================================
// c3n_kernel: 3x3 stride 1 pad 1 convolution, ONE source of 8 CPT channels, N <= 16 NT output channels (N % 8 == 0), bias +
// optional ReLU.  Tile = (4 MT) rows x 16 columns; wave w owns rows w MT .. w MT + MT - 1.  Needs H % (4 MT) == 0, W % 16 == 0.
// =====================================================================================================================
struct NarrowParams {
  const void* src;     // [B,H,W,ld] bf16, first 8*CPT channels used
  const void* w;       // packed [Npad][Kpad] bf16, k = tap * (8 CPT) + channel (models._pack_conv)
  const float* shift;  // bias [N] or nullptr
  void* dst;           // [B,H,W,ldd] bf16 or fp32
  int H, W, ld, N, Npad, Kpad, ldd;
  float act_floor;     // 0 for ReLU, -inf for none: v = max(v, act_floor)
  // MATCH (c3n_kernel<..., MATCH = true>): the NEXT level's rotational matching for ONE rotation hypothesis in this layer's epilogue
  // (models.py:186-205 pattern with a single shift: CVM_VIGOR_ori_prior(0)).  dst then receives the decoder input rows
  // [x / max(|x|, 1e-12) (N channels) | score | 0-pad] of pitch ldd, `scores` the [B,1,H,W] score volume.
  const float* g;      // [B][ldg] ground descriptor of the level (first L entries)
  float* scores;
  int ldg, L, off, batch;   // off = (-shift * stride - window_offset) mod N: window channel c of x meets g[(c + off) mod N]
  TileIndex ti;
  int ablate;          // diagnostics builds only (-DCCVPE_ABLATE, tools/gpu/ablate_narrow.sh): 1 = no halo requests after the first two tiles,
                       // 2 = no epilogue stores, 4 = no matrix instructions, 8 = no fragment reads, 16 = no barrier / DMA wait
};

// NS: the output channels are split over NS groups of waves (each NT column tiles wide); 4 / NS waves share the tile's rows
template <int CPT, int NT, int MT, int NS = 1>
struct C3nGeom {
  static constexpr int TH = (4 / NS) * MT, HR = TH + 2, HC = 18;
  static constexpr int PP = odd_pitch(CPT);                    // slots per pixel
  static constexpr int PIECES = HR * HC * PP;                  // 16-byte slots of a halo tile (DMA piece order)
  static constexpr int NDMA = (PIECES + 255) / 256;            // requests per thread per tile
  static constexpr int BUF_BYTES = NDMA * 256 * 16;
  static constexpr int LDS_BYTES = 2 * BUF_BYTES;
  static constexpr int NCH = (9 * CPT + 3) / 4;                // 32-wide k-steps
};

// LDS behind the two halo buffers of c3n_kernel<MATCH>: per sample, in this kernel's QUAD order, [16 NT] descriptor values
// g[(c + off) mod N] (0 outside the window), [16 NT] window indicators, then |g| per sample
template <int NT> constexpr int match_tab_floats(int batch) { return batch * (2 * 16 * NT + 1); }

// ILV: the previous tile's epilogue interleaved with this tile's matrix work (needs a second accumulator set; the 64-channel tile
// has no registers left for it)
template <int CPT, int NT, int MT, bool F32OUT, bool MATCH = false, bool ILV = true, int NS = 1>
__global__ __launch_bounds__(256, 1) void c3n_kernel(const NarrowParams p) {
  using G = C3nGeom<CPT, NT, MT, NS>;
  static_assert(!(MATCH && NS > 1), "the matching epilogue needs all of a pixel's channels in one wave");
  constexpr int HC = G::HC, NCH = G::NCH, NDMA = G::NDMA, PP = G::PP;
  constexpr int XP = PP * 16;                                   // pixel pitch in the halo image (bytes)
  extern __shared__ __attribute__((aligned(16))) char nsm[];
  CCVPE_NARROW_ABL(p);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = sgpr(tid >> 6);
  const int wr = NS > 1 ? wave / NS : wave;                     // row group of this wave
  const int t0 = NS > 1 ? (wave % NS) * NT : 0;                 // its first column tile
  const int f = lane & 15, q = lane >> 4;

  // ---- the layer's weights: MFMA "A" operands (rows = output channels, permuted: chan_of_row), resident for the whole kernel ---
  f32x4 wreg[NCH][NT];
  {
    const bf16_t* wp = reinterpret_cast<const bf16_t*>(p.w);
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int row = chan_of_row(t0 + t, f);                 // (a split layer's last column tile may lie beyond the packed rows)
        wreg[j][t] = keep_if(*reinterpret_cast<const f32x4*>(wp + (size_t)(row < p.Npad ? row : 0) * p.Kpad + 32 * j + 8 * q), row < p.Npad);
      }
  }
  f32x4 bias[NT];                                               // this lane's quads (no loads inside the tile loop)
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ch = chan_of_quad(t0 + n, q) + r;
      bias[n][r] = (p.shift && ch < p.N) ? p.shift[ch] : 0.f;
    }

  // ---- per-lane halo offsets of the pixel fragments: k-step j, lane group q -> octet 4 j + q = (tap, channel octet) ------
  int aoff[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int o = 4 * j + q;
    const int oo = o < 9 * CPT ? o : 0;
    const int tap = oo / CPT, c = oo - tap * CPT;
    const int ky = tap / 3, kx = tap - 3 * ky;
    aoff[j] = ((wr * MT + ky) * HC + f + kx) * XP + c * 16;
  }
  const bool tail_ok = 4 * (NCH - 1) + q < 9 * CPT;            // the last k-step may end in octets that do not exist (W is zero there)

  // ---- DMA pieces of this thread (tile-invariant): piece -> (halo pixel, octet).  The pad slot of an even octet count re-fetches
  // octet 0 (never read): every lane of a request takes part, no exec masking in the interior path -------------------------------
  unsigned voff[NDMA];
  int hyx[NDMA];
#pragma unroll
  for (int k = 0; k < NDMA; ++k) {
    const int pidx = min(k * 256 + tid, G::PIECES - 1);
    const int pix = pidx / PP, oct = pidx - pix * PP;
    const int hy = pix / HC, hx = pix - hy * HC;
    voff[k] = (unsigned)(((hy * p.W + hx) * p.ld + (oct < CPT ? oct : 0) * 8) * 2);
    hyx[k] = (hy << 8) | hx;
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)nsm;
  const int Hs = sgpr(p.H), Ws = sgpr(p.W), lds_ = sgpr(p.ld);
  float* mtab = reinterpret_cast<float*>(nsm + G::LDS_BYTES);  // MATCH: [batch][2][16 NT] tables, then [batch] |g|
  if constexpr (MATCH) {
    const int C = p.N;
    for (int e = tid; e < p.batch * 16 * NT; e += 256) {
      const int b = e / (16 * NT), rem = e - b * (16 * NT);      // rem = (n * 4 + q) * 4 + r
      const int ch = chan_of_quad(rem >> 4, (rem >> 2) & 3) + (rem & 3);
      int k = ch + p.off;
      k = k >= C ? k - C : k;
      const bool in = ch < C && k < p.L;
      mtab[(b * 2 + 0) * 16 * NT + rem] = in ? p.g[(size_t)b * p.ldg + k] : 0.f;
      mtab[(b * 2 + 1) * 16 * NT + rem] = in ? 1.f : 0.f;
    }
    for (int b = tid; b < p.batch; b += 256) {
      float s = 0.f;
      for (int k = 0; k < p.L; ++k) { const float v = p.g[(size_t)b * p.ldg + k]; s = fmaf(v, v, s); }
      mtab[p.batch * 2 * 16 * NT + b] = sqrtf(s);
    }
  }

  auto stage = [&](int t, int buf) {                         // request tile t's halo into buffer `buf`
    int b, ty, tx;
    tile_decode(p.ti, t, b, ty, tx);
    const int y0 = ty * G::TH, x0 = tx * 16;
    const char* sbase = reinterpret_cast<const char*>(p.src) + ((long)(b * Hs + y0 - 1) * Ws + (x0 - 1)) * (long)(lds_ * 2);
    const bool interior = y0 > 0 && y0 + G::TH < Hs && x0 > 0 && x0 + 16 < Ws;        // wave-uniform
    if (interior) {
#pragma unroll
      for (int k = 0; k < NDMA; ++k) {
        const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * G::BUF_BYTES + (k * 256 + wave * 64) * 16));
        if ((k + 1) * 256 <= G::PIECES || k * 256 + tid < G::PIECES) dma16(ldsw, voff[k], sbase);
      }
    } else {
#pragma unroll
      for (int k = 0; k < NDMA; ++k) {
        if ((k + 1) * 256 <= G::PIECES || k * 256 + tid < G::PIECES) {
          const int iy = y0 - 1 + (hyx[k] >> 8), ix = x0 - 1 + (hyx[k] & 255);
          const bool ok = (unsigned)iy < (unsigned)Hs && (unsigned)ix < (unsigned)Ws;
          const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * G::BUF_BYTES + (k * 256 + wave * 64) * 16));
          if (ok) dma16(ldsw, voff[k], sbase);
          else *reinterpret_cast<f32x4*>(nsm + buf * G::BUF_BYTES + (k * 256 + tid) * 16) = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
    }
  };

  // ---- epilogue of ONE pixel row (lane = pixel column f of row i of tile (b, ty, tx)); see run_tile for where it is issued ----------
  struct EpiCtx { size_t pix0; f32x4 gq[NT], wq[NT]; float gnorm; };
  auto epi_ctx = [&](int t, EpiCtx& c) {
    int b, ty, tx;
    tile_decode(p.ti, t, b, ty, tx);
    c.pix0 = ((size_t)(b * Hs + ty * G::TH + wr * MT) * Ws + tx * 16 + f);
    c.gnorm = 1.f;
    if constexpr (MATCH) {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        c.gq[n] = *reinterpret_cast<const f32x4*>(mtab + (b * 2 + 0) * 16 * NT + (n * 4 + q) * 4);
        c.wq[n] = *reinterpret_cast<const f32x4*>(mtab + (b * 2 + 1) * 16 * NT + (n * 4 + q) * 4);
      }
      c.gnorm = mtab[p.batch * 2 * 16 * NT + b];
    }
  };
  auto epi_row = [&](const EpiCtx& c, int i, const f32x4* accrow) {
    f32x4 v[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      v[n] = accrow[n] + bias[n];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[n][r] = fmaxf(v[n][r], p.act_floor);
    }
    if constexpr (MATCH) {
      // this lane holds 4 NT of the pixel's channels (the lanes f, f + 16, f + 32, f + 48 share a pixel; channels >= N are exact
      // zeros: zero weight rows, zero bias).  |x|^2, the window's |.|^2 and the dot product with the rolled descriptor: partial
      // sums here, combined over the four lanes (fixed order); then exactly match_kernel's formulas (no eps in the cosine).
      float s2 = 0.f, wn = 0.f, dt = 0.f;
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x2 = v[n][r] * v[n][r];
          s2 += x2;
          wn = fmaf(x2, c.wq[n][r], wn);
          dt = fmaf(v[n][r], c.gq[n][r], dt);
        }
      s2 += __shfl_xor(s2, 16, 64); wn += __shfl_xor(wn, 16, 64); dt += __shfl_xor(dt, 16, 64);
      s2 += __shfl_xor(s2, 32, 64); wn += __shfl_xor(wn, 32, 64); dt += __shfl_xor(dt, 32, 64);
      const float inv = 1.0f / fmaxf(sqrtf(s2), 1e-12f);
      const float score = dt / (sqrtf(wn) * c.gnorm);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int ch = chan_of_quad(n, q);
        if (ch < p.N) v[n] *= inv;
        else v[n] = (f32x4){ch == p.N ? score : 0.f, 0.f, 0.f, 0.f};          // [max score | 0-pad] (N % 4 == 0)
      }
      if (q == 0 && !(abl & 2)) p.scores[c.pix0 + (size_t)i * Ws] = score;       // [B,1,H,W]
    }
    if (abl & 2) {
#pragma unroll
      for (int n = 0; n < NT; ++n) asm volatile("" ::"v"(v[n]));
      return;
    }
    store_row<NT, F32OUT>(p.dst, (c.pix0 + (size_t)i * Ws) * p.ldd, v, q, MATCH ? p.ldd : p.N, t0);
  };

  // ---- one tile: the matrix work of tile s (halo buffer s & 1, accumulator set s & 1) with the EPILOGUE OF TILE s - 1 interleaved:
  // row i of the previous tile's results (other accumulator set) is converted and stored in the same scheduling region as the matrix
  // instructions of k-step i, one matrix instruction : a few vector instructions (sched_group_barrier).  At one wave per SIMD nothing
  // else can fill the issue slots beside a matrix instruction; run back to back, the ~300 instructions of an epilogue cost as much
  // as a third of the tile's matrix work (ablation: 83 of 216 us with every memory operation and matrix instruction removed). ---------
  f32x4 acc[ILV ? 2 : 1][MT][NT];
  auto run_tile = [&](auto buf_tag, auto prev_tag, int tprev) {
    constexpr int BUF = decltype(buf_tag)::value;
    constexpr bool PREV = decltype(prev_tag)::value;
    constexpr int SET = ILV ? BUF : 0, OTHER = ILV ? (BUF ^ 1) : 0;
    const char* hb = nsm + BUF * G::BUF_BYTES;
    EpiCtx ctx;
    if constexpr (PREV) epi_ctx(tprev, ctx);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[SET][i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 a[2][MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) a[0][i] = *reinterpret_cast<const f32x4*>(hb + aoff[0] + i * (HC * XP));
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int cur = j & 1;
      if (j + 1 < NCH && !(abl & 8)) {
#pragma unroll
        for (int i = 0; i < MT; ++i) a[cur ^ 1][i] = *reinterpret_cast<const f32x4*>(hb + aoff[j + 1] + i * (HC * XP));
      }
      __builtin_amdgcn_sched_barrier(0);                      // the next k-step's fragment reads stay above this k-step's MFMAs
      if (j == NCH - 1 && (9 * CPT) % 4 != 0) {
#pragma unroll
        for (int i = 0; i < MT; ++i) a[cur][i] = keep_if(a[cur][i], tail_ok);
      }
      if (!(abl & 4)) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[SET][i][n] = mfma_stage<bf16_t>(wreg[j][n], a[(abl & 8) ? 0 : cur][i], acc[SET][i][n]);
      }
      if constexpr (PREV) {
        if (j < MT) {
          epi_row(ctx, j, acc[OTHER][j]);
#pragma unroll
          for (int m = 0; m < MT * NT; ++m) {                 // 1 matrix instruction : 6 vector instructions, stores / LDS ops wherever they fall
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto epilogue = [&](int t, int set) {                       // the last tile of this workgroup: nothing left to hide it under
    EpiCtx ctx;
    epi_ctx(t, ctx);
#pragma unroll
    for (int i = 0; i < MT; ++i) epi_row(ctx, i, (ILV && set) ? acc[ILV ? 1 : 0][i] : acc[0][i]);
  };
  auto tile_end = [&]() {                                     // this wave's halo requests have landed + everyone is done reading
    if (abl & 16) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  // tile s of this workgroup's sequence computes from buffer s & 1; the halo of tile s + 2 is requested right after the barrier
  // that ends tile s's reads of that buffer, i.e. a whole tile of matrix work before it is needed
  int t = blockIdx.x;
  const int step = gridDim.x;
  const int total = p.ti.total;
  if (t >= total) return;
  stage(t, 0);
  if (t + step < total) stage(t + step, 1);
  tile_end();
  if constexpr (!ILV) {                                       // one accumulator set: the epilogue follows its tile's barrier
    while (true) {
      run_tile(std::integral_constant<int, 0>{}, std::false_type{}, 0);
      tile_end();
      if (t + 2 * step < total && !(abl & 1)) stage(t + 2 * step, 0);
      epilogue(t, 0);
      t += step;
      if (t >= total) return;
      run_tile(std::integral_constant<int, 1>{}, std::false_type{}, 0);
      tile_end();
      if (t + 2 * step < total && !(abl & 1)) stage(t + 2 * step, 1);
      epilogue(t, 0);
      t += step;
      if (t >= total) return;
    }
  }
  run_tile(std::integral_constant<int, 0>{}, std::false_type{}, 0);
  tile_end();
  if (t + 2 * step < total && !(abl & 1)) stage(t + 2 * step, 0);
  int tprev = t, last = 0;
  t += step;
  while (t < total) {
    run_tile(std::integral_constant<int, 1>{}, std::true_type{}, tprev);
    tile_end();
    if (t + 2 * step < total && !(abl & 1)) stage(t + 2 * step, 1);
    tprev = t; last = 1;
    t += step;
    if (t >= total) break;
    run_tile(std::integral_constant<int, 0>{}, std::true_type{}, tprev);
    tile_end();
    if (t + 2 * step < total && !(abl & 1)) stage(t + 2 * step, 0);
    tprev = t; last = 0;
    t += step;
  }
  epilogue(tprev, last);
}

int num_cus();   // narrow_bf16.hip

template <int CPT, int NT, int MT, bool WITH_MATCH = true, bool ILV = true, int NS = 1>
static int launch_c3n(NarrowParams p, int batch, bool f32out, bool match, hipStream_t stream) {
  using G = C3nGeom<CPT, NT, MT, NS>;
  const int tiles_x = p.W / 16, tiles_y = p.H / G::TH;
  const long total = (long)tiles_x * tiles_y * batch;
  if (total > 0x7fffffffL || total * (tiles_x > tiles_y ? tiles_x : tiles_y) >= (1L << 32)) return fail(CCVPE_EINVAL, "c3n: grid too large");
  p.ti = make_tile_index(tiles_x, tiles_y, (int)total);
  p.ablate = narrow_ablate_env();
  p.batch = batch;
  const int lds = G::LDS_BYTES + (match ? match_tab_floats<NT>(batch) * 4 : 0);
  if (lds > 160 * 1024) return fail(CCVPE_EINVAL, "c3n: batch %d needs %d B of LDS for the matching tables", batch, lds);
  void (*kern)(const NarrowParams) = f32out ? c3n_kernel<CPT, NT, MT, true, false, ILV, NS> : c3n_kernel<CPT, NT, MT, false, false, ILV, NS>;
  if (match) {
    if constexpr (WITH_MATCH) kern = f32out ? c3n_kernel<CPT, NT, MT, true, true, ILV, NS> : c3n_kernel<CPT, NT, MT, false, true, ILV, NS>;
    else return fail(CCVPE_EINVAL, "c3n: the matching epilogue is not built for this tile");
  }
  static int attr_lds[4] = {0, 0, 0, 0};
  const int slot = (match ? 2 : 0) + (f32out ? 1 : 0);
  if (attr_lds[slot] < lds) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return fail(CCVPE_ELAUNCH, "c3n_kernel: set smem attr: %s", hipGetErrorString(e));
    attr_lds[slot] = lds;
  }
  const int grid = (int)(total < num_cus() ? total : num_cus());
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);
  return check_launch("c3n_kernel");
}

// =====================================================================================================================
// up2_kernel: ConvTranspose2d(k2,s2) folded into the following 3x3 conv (the math of upconv_impl.h) for the NARROW levels.
// The tiled kernels give every output parity its own workgroup, so the low-res halo is fetched four times and the 9 skip
// taps are gathered per parity: at level 2 (81 -> 40 channels at 256 x 256) that is ~180 KB of L2 -> CU traffic per 20 KB of
// output, and the kernel ran at the L2's bandwidth (481 us at B = 64: 0.35 PF, 5 x off the HBM roof).  Here ONE workgroup
// computes all four parities of a MT x 16 low-res tile from ONE low-res halo and ONE skip halo in LDS: wave w owns parity
// (py, px) = (w >> 1, w & 1) and keeps THAT parity's folded weights in registers (the structure of tail512_kernel, with the flat
// (tap, octet) K order of c3n_kernel above).  Persistent workgroups, LDS-DMA halos, one barrier per tile, epilogue after it.
// LDS images (DMA piece order): low-res halo [row][18 columns][odd pitch of CPT0 octets]; skip halo [row][column parity][octet]
// [17 columns] (column parity planes: a fragment's 16 pixels are 16 CONSECUTIVE 16-byte slots).
// =====================================================================================================================
struct Up2Params {
  const void* src0;    // [B,H1,W1,ld0] bf16
  const void* src1;    // [B,2H1,2W1,ld1] bf16
  const void* w;       // [4][Npad][Kpad] bf16 (models._pack_upconv)
  const float* shift9; // [9][N]
  void* dst;           // [B,2H1,2W1,ldd] bf16
  int H1, W1, ld0, ld1, N, Npad, Kpad, ldd;
  float act_floor;
  TileIndex ti;
  int ablate;          // see NarrowParams
};

template <int CPT0, int CPT1, int NT, int MT>
struct Up2Geom {
  static constexpr int HR = MT + 2, HC = 18;
  static constexpr int PP0 = odd_pitch(CPT0);
  static constexpr int PX = HR * HC * PP0;
  static constexpr int NDX = (PX + 255) / 256;
  static constexpr int SR = 2 * MT + 2;
  static constexpr int SLINE = 17 * 16;                          // bytes of one (row, column parity, octet) line
  static constexpr int PS = SR * 2 * CPT1 * 17;
  static constexpr int NDS = (PS + 255) / 256;
  static constexpr int SKIP_BASE = NDX * 4096;
  static constexpr int BUF_BYTES = (NDX + NDS) * 4096;
  static constexpr int SHIFT_BASE = 2 * BUF_BYTES;               // shift9 [9][16 NT] floats in quad order (epilogue reads it by ds_read: no vmcnt traffic)
  static constexpr int LDS_BYTES = 2 * BUF_BYTES + 9 * 16 * NT * 4;
  static constexpr int NCHX = CPT0;                              // 4 taps x CPT0 octets = CPT0 k-steps exactly
  static constexpr int NCHS = (9 * CPT1 + 3) / 4;
  static constexpr int NCH = NCHX + NCHS;
};

template <int CPT0, int CPT1, int NT, int MT>
__global__ __launch_bounds__(256, 1) void up2_kernel(const Up2Params p) {
  using G = Up2Geom<CPT0, CPT1, NT, MT>;
  constexpr int HC = G::HC, NCH = G::NCH, NCHX = G::NCHX, NCHS = G::NCHS, NDX = G::NDX, NDS = G::NDS;
  constexpr int XP = G::PP0 * 16;
  constexpr int XROW = HC * XP;                                  // one low-res halo row
  constexpr int SROW2 = 2 * (2 * CPT1 * G::SLINE);               // two skip halo rows (one low-res row step)
  extern __shared__ __attribute__((aligned(16))) char nsm[];
  CCVPE_NARROW_ABL(p);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = sgpr(tid >> 6);
  const int py = wave >> 1, px = wave & 1;
  const int f = lane & 15, q = lane >> 4;

  // ---- this parity's folded weights, resident --------------------------------------------------------------------------
  f32x4 wreg[NCH][NT];
  {
    const bf16_t* wp = reinterpret_cast<const bf16_t*>(p.w) + (size_t)wave * p.Npad * p.Kpad;
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        wreg[j][t] = *reinterpret_cast<const f32x4*>(wp + (size_t)chan_of_row(t, f) * p.Kpad + 32 * j + 8 * q);
  }

  // ---- fragment offsets --------------------------------------------------------------------------------------------------
  int aoff[NCH];
#pragma unroll
  for (int j = 0; j < NCHX; ++j) {
    const int o = 4 * j + q;
    const int tap = o / CPT0, c = o - tap * CPT0;
    const int du = tap >> 1, dv = tap & 1;
    aoff[j] = ((du + py) * HC + f + dv + px) * XP + c * 16;
  }
#pragma unroll
  for (int j = 0; j < NCHS; ++j) {
    const int o = 4 * j + q;
    const int oo = o < 9 * CPT1 ? o : 0;
    const int tap = oo / CPT1, c = oo - tap * CPT1;
    const int ky = tap / 3, kx = tap - 3 * ky;
    const int cpar = (px + kx) & 1, chh = f + ((px + kx) >> 1);
    aoff[NCHX + j] = G::SKIP_BASE + (((py + ky) * 2 + cpar) * CPT1 + c) * G::SLINE + chh * 16;
  }
  const bool tail_ok = 4 * (NCHS - 1) + q < 9 * CPT1;

  // ---- DMA pieces (tile-invariant) ----------------------------------------------------------------------------------------
  const int W2 = 2 * p.W1, H2 = 2 * p.H1;
  unsigned xv[NDX], sv[NDS];
  int xh[NDX], shh[NDS];
#pragma unroll
  for (int k = 0; k < NDX; ++k) {
    const int pidx = min(k * 256 + tid, G::PX - 1);
    const int pix = pidx / G::PP0, oct = pidx - pix * G::PP0;
    const int hy = pix / HC, hx = pix - hy * HC;
    xv[k] = (unsigned)(((hy * p.W1 + hx) * p.ld0 + (oct < CPT0 ? oct : 0) * 8) * 2);
    xh[k] = (hy << 8) | hx;
  }
#pragma unroll
  for (int k = 0; k < NDS; ++k) {
    const int pidx = min(k * 256 + tid, G::PS - 1);
    const int line = pidx / 17, chh = pidx - line * 17;          // line = (row * 2 + cpar) * CPT1 + oct
    const int oct = line % CPT1, rc = line / CPT1;
    const int cpar = rc & 1, row = rc >> 1;
    const int hc = 2 * chh + cpar;
    sv[k] = (unsigned)(((row * W2 + hc) * p.ld1 + oct * 8) * 2);
    shh[k] = (row << 8) | hc;
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)nsm;
  const int H1s = sgpr(p.H1), W1s = sgpr(p.W1);
  {                                                              // the nine shift vectors -> LDS in QUAD order, zero-padded
    float* st = reinterpret_cast<float*>(nsm + G::SHIFT_BASE);
    for (int e = tid; e < 9 * 16 * NT; e += 256) {
      const int cls = e / (16 * NT), rem = e - cls * (16 * NT);  // rem = (n * 4 + q) * 4 + r
      const int ch = chan_of_quad(rem >> 4, (rem >> 2) & 3) + (rem & 3);
      st[e] = ch < p.N ? p.shift9[cls * p.N + ch] : 0.f;
    }
  }

  auto stage = [&](int t, int buf) {
    int b, ty, tx;
    tile_decode(p.ti, t, b, ty, tx);
    const int y0 = ty * MT, x0 = tx * 16;
    const char* sb0 = reinterpret_cast<const char*>(p.src0) + ((long)(b * H1s + y0 - 1) * W1s + (x0 - 1)) * (long)(p.ld0 * 2);
    const char* sb1 = reinterpret_cast<const char*>(p.src1) + ((long)(b * H2 + 2 * y0 - 1) * W2 + (2 * x0 - 1)) * (long)(p.ld1 * 2);
    const bool interior = y0 > 0 && y0 + MT < H1s && x0 > 0 && x0 + 16 < W1s;       // wave-uniform
    if (interior) {
#pragma unroll
      for (int k = 0; k < NDX; ++k) {
        const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * G::BUF_BYTES + (k * 256 + wave * 64) * 16));
        if ((k + 1) * 256 <= G::PX || k * 256 + tid < G::PX) dma16(ldsw, xv[k], sb0);
      }
#pragma unroll
      for (int k = 0; k < NDS; ++k) {
        const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * G::BUF_BYTES + G::SKIP_BASE + (k * 256 + wave * 64) * 16));
        if ((k + 1) * 256 <= G::PS || k * 256 + tid < G::PS) dma16(ldsw, sv[k], sb1);
      }
    } else {
#pragma unroll
      for (int k = 0; k < NDX; ++k) {
        if ((k + 1) * 256 <= G::PX || k * 256 + tid < G::PX) {
          const int iy = y0 - 1 + (xh[k] >> 8), ix = x0 - 1 + (xh[k] & 255);
          const bool ok = (unsigned)iy < (unsigned)H1s && (unsigned)ix < (unsigned)W1s;
          const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * G::BUF_BYTES + (k * 256 + wave * 64) * 16));
          if (ok) dma16(ldsw, xv[k], sb0);
          else *reinterpret_cast<f32x4*>(nsm + buf * G::BUF_BYTES + (k * 256 + tid) * 16) = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll
      for (int k = 0; k < NDS; ++k) {
        if ((k + 1) * 256 <= G::PS || k * 256 + tid < G::PS) {
          const int iy = 2 * y0 - 1 + (shh[k] >> 8), ix = 2 * x0 - 1 + (shh[k] & 255);
          const bool ok = (unsigned)iy < (unsigned)H2 && (unsigned)ix < (unsigned)W2;
          const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * G::BUF_BYTES + G::SKIP_BASE + (k * 256 + wave * 64) * 16));
          if (ok) dma16(ldsw, sv[k], sb1);
          else *reinterpret_cast<f32x4*>(nsm + buf * G::BUF_BYTES + G::SKIP_BASE + (k * 256 + tid) * 16) = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
    }
  };

  // ---- epilogue of one low-res row: pixel (2 (y0 + i) + py, 2 (x0 + f) + px); shift by the pixel's border class ----------------------
  struct EpiCtx { int b, y0, X, cc; };
  auto epi_ctx = [&](int t, EpiCtx& c) {
    int ty, tx;
    tile_decode(p.ti, t, c.b, ty, tx);
    c.y0 = ty * MT;
    c.X = 2 * (tx * 16 + f) + px;
    c.cc = c.X == 0 ? 0 : (c.X == W2 - 1 ? 2 : 1);
  };
  auto epi_row = [&](const EpiCtx& c, int i, const f32x4* accrow) {
    const int Y = 2 * (c.y0 + i) + py;
    const int rc = Y == 0 ? 0 : (Y == H2 - 1 ? 2 : 1);
    const char* shp = nsm + G::SHIFT_BASE + ((rc * 3 + c.cc) * 16 * NT + q * 4) * 4;
    f32x4 v[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      v[n] = accrow[n] + *reinterpret_cast<const f32x4*>(shp + 64 * n);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[n][r] = fmaxf(v[n][r], p.act_floor);
    }
    if (abl & 2) {
#pragma unroll
      for (int n = 0; n < NT; ++n) asm volatile("" ::"v"(v[n]));
      return;
    }
    store_row<NT, false>(p.dst, ((size_t)(c.b * H2 + Y) * W2 + c.X) * p.ldd, v, q, p.N);
  };

  // ---- the matrix work of HALF a tile (low-res rows HM h .. HM h + HM - 1, accumulator set h) with the epilogue of the PREVIOUS half
  // (the other accumulator set; of this tile for h = 1, of the previous tile for h = 0) interleaved: row i of it is converted and
  // stored in the scheduling region of k-step i's matrix instructions (see c3n_kernel's run_tile).  Two half-tile passes instead of
  // one whole-tile pass cost nothing here — the weights are registers, a k-step's W fragment is free to re-use — and the two
  // accumulator sets together are the registers one whole-tile set took. -------------------------------------------------------------
  constexpr int HM = MT / 2;
  static_assert(MT % 2 == 0, "up2_kernel: two half-tile passes");
  f32x4 acc[2][HM][NT];
  auto run_half = [&](auto buf_tag, auto half_tag, auto prev_tag, const EpiCtx& pc, int prow0) {
    constexpr int BUF = decltype(buf_tag)::value;
    constexpr int H = decltype(half_tag)::value;
    constexpr bool PREV = decltype(prev_tag)::value;
    const char* hb = nsm + BUF * G::BUF_BYTES;
#pragma unroll
    for (int i = 0; i < HM; ++i)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[H][i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 a[2][HM];
    auto rd = [&](int j, f32x4* dst) {
#pragma unroll
      for (int i = 0; i < HM; ++i)
        dst[i] = *reinterpret_cast<const f32x4*>(hb + aoff[j] + (H * HM + i) * (j < NCHX ? XROW : SROW2));
    };
    rd(0, a[0]);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int cur = j & 1;
      if (j + 1 < NCH && !(abl & 8)) rd(j + 1, a[cur ^ 1]);
      __builtin_amdgcn_sched_barrier(0);
      if (j == NCH - 1 && (9 * CPT1) % 4 != 0) {
#pragma unroll
        for (int i = 0; i < HM; ++i) a[cur][i] = keep_if(a[cur][i], tail_ok);
      }
      if (!(abl & 4)) {
#pragma unroll
        for (int i = 0; i < HM; ++i)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[H][i][n] = mfma_stage<bf16_t>(wreg[j][n], a[(abl & 8) ? 0 : cur][i], acc[H][i][n]);
      }
      if constexpr (PREV) {
        if (j < HM) {
          epi_row(pc, prow0 + j, acc[H ^ 1][j]);
#pragma unroll
          for (int m = 0; m < HM * NT; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto tile_end = [&]() {
    if (abl & 16) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  int t = blockIdx.x;
  const int step = gridDim.x;
  const int total = p.ti.total;
  if (t >= total) return;
  stage(t, 0);
  if (t + step < total) stage(t + step, 1);
  tile_end();
  EpiCtx cur_ctx, prev_ctx;
  epi_ctx(t, cur_ctx);
  prev_ctx = cur_ctx;
  // first tile: its first half has no predecessor
  run_half(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::false_type{}, prev_ctx, 0);
  run_half(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::true_type{}, cur_ctx, 0);
  tile_end();
  if (t + 2 * step < total && !(abl & 1)) stage(t + 2 * step, 0);
  prev_ctx = cur_ctx;
  t += step;
  int sbuf = 1;
  while (t < total) {
    epi_ctx(t, cur_ctx);
    if (sbuf) {
      run_half(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, std::true_type{}, prev_ctx, HM);
      run_half(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, std::true_type{}, cur_ctx, 0);
    } else {
      run_half(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::true_type{}, prev_ctx, HM);
      run_half(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::true_type{}, cur_ctx, 0);
    }
    tile_end();
    if (t + 2 * step < total && !(abl & 1)) stage(t + 2 * step, sbuf);
    prev_ctx = cur_ctx;
    sbuf ^= 1;
    t += step;
  }
  // the second half of the last tile
#pragma unroll
  for (int i = 0; i < HM; ++i) epi_row(prev_ctx, HM + i, acc[1][i]);
}

template <int CPT0, int CPT1, int NT, int MT>
static int launch_up2(Up2Params p, int batch, hipStream_t stream) {
  using G = Up2Geom<CPT0, CPT1, NT, MT>;
  static_assert(G::LDS_BYTES <= 160 * 1024, "up2_kernel: LDS");
  const int tiles_x = p.W1 / 16, tiles_y = p.H1 / MT;
  const long total = (long)tiles_x * tiles_y * batch;
  if (total > 0x7fffffffL || total * (tiles_x > tiles_y ? tiles_x : tiles_y) >= (1L << 32)) return fail(CCVPE_EINVAL, "up2: grid too large");
  p.ti = make_tile_index(tiles_x, tiles_y, (int)total);
  p.ablate = narrow_ablate_env();
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)up2_kernel<CPT0, CPT1, NT, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    if (e != hipSuccess) return fail(CCVPE_ELAUNCH, "up2_kernel: set smem attr: %s", hipGetErrorString(e));
    attr_set = true;
  }
  const int grid = (int)(total < num_cus() ? total : num_cus());
  hipLaunchKernelGGL((up2_kernel<CPT0, CPT1, NT, MT>), dim3(grid), dim3(256), G::LDS_BYTES, stream, p);
  return check_launch("up2_kernel");
}

}  // namespace ccvpe
