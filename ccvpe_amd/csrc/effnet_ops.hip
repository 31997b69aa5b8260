// EfficientNet-B0 memory-bound operators: stem conv, depthwise conv (+BN+swish+SE squeeze), SE gate.
// All HBM-bound (AI < 10 F/B): the design rules are coalesced 16-byte NHWC
// accesses, one pass over the tensor, and every elementwise op fused into the producing kernel.
#include "common.h"
#include "mbconv_plane.h"

namespace ccvpe {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// Stem: 3x3 stride-2 conv, NCHW image -> NHWC [.,32], folded BN + swish.
// efficientnet_pytorch/model.py:181-182,289 ; padding utils.py:265-277 (zero), :341-353 (circular W)
//
// Round 4: on the fp32 matrix cores.  The round-1..3 kernel (one thread per output pixel, 864 FMAs against LDS-broadcast
// weights, 27 strided dword gathers) took 213 us per aerial batch whatever the output type (469 MB bf16 / 738 MB fp32: neither
// HBM- nor FMA-bound — load-issue and LDS-broadcast bound).  Now a workgroup stages the (2 TH + 1) x (2 TW + 1) x 3 input patch of
// a TH x TW output tile in LDS with coalesced row loads and computes out[pixel][32] = patch[pixel][27] . W[27][32] as
// v_mfma_f32_16x16x4_f32: K = 27 padded to 28 = 7 k-steps, N = 32 = 2 column tiles, W (14 registers per lane) loaded once; a lane's
// B operand for k-step s is ONE ds_read_b32 at a per-lane constant offset (its k = 4s + q picks (ky, kx, ci)) plus the pixel's
// column.  Exact fp32 (the fp32 MFMA is a k-ordered fmaf chain), for both output types.  The C layout gives a lane 4 consecutive
// channels of one pixel: 16-byte (fp32) / 8-byte (bf16) stores, a pixel's 128 bytes complete within two store instructions.
// RAW: write the convolution result only (train mode: BatchNorm needs the batch statistics first)
// ---------------------------------------------------------------------------------------------
constexpr int ST_TH = 4, ST_TW = 64;                   // output tile: one wave per output row, 4 pixel tiles of 16 per wave
constexpr int ST_IR = 2 * ST_TH + 1, ST_IC = 2 * ST_TW + 1, ST_PITCH = ST_IC + 3;    // input patch rows / columns / LDS pitch

template <typename T, bool RAW>
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift, T* __restrict__ y,
                                                        int B, int H, int W, int Ho, int Wo, int circular, int tiles_x,
                                                        int tiles_y) {
  __shared__ float img[3 * ST_IR * ST_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pix = lane & 15, q = lane >> 4;
  int lb;
  {                                                    // XCD-aware order: neighbouring tiles of a sample share input rows in one L2
    const int total = gridDim.x;
    const int qq = total / 8, r = total % 8;
    const int xcd = blockIdx.x % 8, loc = blockIdx.x / 8;
    lb = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + loc;
  }
  const int tx = lb % tiles_x, ty = (lb / tiles_x) % tiles_y, b = lb / (tiles_x * tiles_y);
  const int oy0 = ty * ST_TH, ox0 = tx * ST_TW;
  // ---- the input patch: rows 2 oy0 .. 2 oy0 + 8, columns 2 ox0 .. 2 ox0 + 128 of the three planes (pad before = 0) --------
  const float* xb = x + (size_t)b * 3 * H * W;
  {
    // all loads of the patch in flight together, from clamped addresses, masked at the LDS store: `ok ? load : 0` compiles to
    // an exec-mask branch per load and serialises the 14 round trips of a thread
    constexpr int NIT = (3 * ST_IR * ST_PITCH + 255) / 256;
    float v[NIT];
    unsigned okm = 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = min(tid + 256 * it, 3 * ST_IR * ST_PITCH - 1);
      const int row = idx / ST_PITCH, c = idx - row * ST_PITCH;
      const int ci = row / ST_IR, r = row - ci * ST_IR;
      const int iy = 2 * oy0 + r;
      int ix = 2 * ox0 + c;
      ix = (circular && ix >= W) ? ix - W : ix;
      const bool ok = c < ST_IC && iy < H && ix < W;
      v[it] = xb[ok ? ((size_t)ci * H + iy) * W + ix : 0];
      okm |= ok ? (1u << it) : 0u;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + 256 * it;
      if (idx < 3 * ST_IR * ST_PITCH) img[idx] = ((okm >> it) & 1u) ? v[it] : 0.f;
    }
  }
  // ---- W fragments (A operand: lane (n = pix, q) supplies W[k = 4 s + q][n]) and the lane's patch offsets ----------------
  float wr[7][2];
  int off[7];
#pragma unroll
  for (int s = 0; s < 7; ++s) {
    const int k = 4 * s + q;
    const bool kv = k < 27;
    const int kc = kv ? k : 0;
    const float w0 = w[kc * 32 + pix], w1 = w[kc * 32 + 16 + pix];      // (unconditional loads: see the patch loads above)
    wr[s][0] = kv ? w0 : 0.f;
    wr[s][1] = kv ? w1 : 0.f;
    const int tap = kc / 3, ci = kc - tap * 3;
    const int ky = tap / 3, kx = tap - ky * 3;
    off[s] = (ci * ST_IR + 2 * wave + ky) * ST_PITCH + kx + 2 * pix;
  }
  f32x4 sc[2], sh[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    sc[j] = RAW ? (f32x4){1.f, 1.f, 1.f, 1.f} : *reinterpret_cast<const f32x4*>(scale + 16 * j + 4 * q);
    sh[j] = RAW ? (f32x4){0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(shift + 16 * j + 4 * q);
  }
  __syncthreads();
  const int oy = oy0 + wave;
#pragma unroll
  for (int i = 0; i < ST_TW / 16; ++i) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    float xv[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) xv[s] = img[off[s] + 32 * i];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[s][0], xv[s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[s][1], xv[s], acc1, 0, 0, 0);
    }
    const int ox = ox0 + 16 * i + pix;
    if (oy < Ho && ox < Wo) {
      T* dst = y + (((size_t)b * Ho + oy) * Wo + ox) * 32 + 4 * q;
      f32x4 v0 = acc0 * sc[0] + sh[0], v1 = acc1 * sc[1] + sh[1];
      if (!RAW) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { v0[r] = swishf(v0[r]); v1[r] = swishf(v1[r]); }
      }
      st4<T>(dst, v0);
      st4<T>(dst + 16, v1);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Depthwise KxK, stride S, NHWC, folded BN + swish, + SE squeeze partial sums.
// efficientnet_pytorch/model.py:70-73,108-110,114.
// A thread owns 4 channels x a strip of TW=4 output columns and slides the input window through
// registers (each input float4 is loaded once per strip row instead of once per tap).
// Block = P strips x CGX channel groups; the per-channel sum of the block's outputs is written
// to se_partial[b][blockIdx.x][c] (fixed order => bit-reproducible squeeze).
// ---------------------------------------------------------------------------------------------
constexpr int DW_TW = 4;

template <typename T, int K, int S, bool RAW>
__global__ __launch_bounds__(256) void dwconv_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ scale,
                                                     const float* __restrict__ shift, T* __restrict__ y,
                                                     float* __restrict__ se_partial, int H, int W, int C, int Ho,
                                                     int Wo, int cgx, int P, int nblk, int circular, int RB,
                                                     int ychunks, int total_blocks) {
  constexpr int PB = (S == 1) ? (K - 1) / 2 : (K - 2) / 2;  // pad before (224-schedule SAME)
  constexpr int NCOL = (DW_TW - 1) * S + K;
  constexpr int KYU_STRIP = K == 3 ? 3 : 1;    // k = 5: one kernel row (8-11 loads) in flight at a time, else 256 VGPRs
  extern __shared__ __attribute__((aligned(16))) float red[];  // [P][cgx] float4

  const int tid = threadIdx.x;
  const int cgl = tid % cgx;
  const int pl = tid / cgx;
  // XCD-aware order: workgroup ids round-robin over the 8 XCDs; give each XCD a contiguous run of
  // logical blocks (= neighbouring strips/rows of the same sample) so the k-row halo that adjacent
  // strips share is served by ONE L2 instead of being fetched from HBM by several.
  int lb;
  {
    const int q = total_blocks / 8, r = total_blocks % 8;
    const int xcd = blockIdx.x % 8, loc = blockIdx.x / 8;
    lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int bx = lb % nblk;
  const int by = (lb / nblk) % ychunks;
  const int b = lb / (nblk * ychunks);
  const int cg = by * cgx + cgl;
  const int c = cg * 4;
  const int sxn = (Wo + DW_TW - 1) / DW_TW;

  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  for (int rb = 0; rb < RB; ++rb) {
  const int strip = (bx * RB + rb) * P + pl;
  const bool active = pl < P && strip < Ho * sxn && c < C;
  if (active) {
    const int oy = strip / sxn;
    const int ox0 = (strip - oy * sxn) * DW_TW;
    f32x4 acc[DW_TW];
#pragma unroll
    for (int t = 0; t < DW_TW; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const T* xb = x + (size_t)b * H * W * C + c;
    // BRANCH-FREE window loads: rows / columns outside the image read a clamped (valid) address and are zeroed by AND-ing
    // the bits with a lane mask; the circular wrap is two selects.  (`ok ? load : 0` and `if (row outside) continue`
    // compiled to one exec-mask branch per load / per kernel row: each of the K * NCOL loads of a strip waited for the
    // previous one — the strip kernel ran at a fifth of the HBM rate on the 256^2 x 32 tensor of block 0.)
    typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll KYU_STRIP
    for (int ky = 0; ky < K; ++ky) {
      const int iy = oy * S - PB + ky;
      const bool rowok = (unsigned)iy < (unsigned)H;
      const size_t rbase = (size_t)(rowok ? iy : 0) * W;
      f32x4 col[NCOL];
#pragma unroll
      for (int j = 0; j < NCOL; ++j) {
        int ix = ox0 * S - PB + j;
        const int wrapped = ix < 0 ? ix + W : (ix >= W ? ix - W : ix);
        ix = circular ? wrapped : ix;
        const bool ok = rowok && (unsigned)ix < (unsigned)W;
        const int ixc = (unsigned)ix < (unsigned)W ? ix : 0;
        const int m = ok ? -1 : 0;
        const f32x4 v = ld4<T>(xb + (rbase + ixc) * C);
        col[j] = __builtin_bit_cast(f32x4, __builtin_bit_cast(i32x4, v) & (i32x4){m, m, m, m});
      }
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (size_t)(ky * K + kx) * C + c);
#pragma unroll
        for (int t = 0; t < DW_TW; ++t) acc[t] += col[t * S + kx] * wv;
      }
    }
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (!RAW) {
      sc = *reinterpret_cast<const f32x4*>(scale + c);
      sh = *reinterpret_cast<const f32x4*>(shift + c);
    }
    T* yb = y + ((size_t)(b * Ho + oy) * Wo) * C + c;
#pragma unroll
    for (int t = 0; t < DW_TW; ++t) {
      const int ox = ox0 + t;
      if (ox < Wo) {
        f32x4 v = acc[t] * sc + sh;
        if (!RAW) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = swishf(v[j]);
        }
        st4<T>(yb + (size_t)ox * C, v);
        sum += v;
      }
    }
  }
  }
  if (RAW) return;   // no squeeze partials in raw mode (uniform: whole workgroup)
  // block reduction over the P strips, fixed order
  f32x4* red4 = reinterpret_cast<f32x4*>(red);
  if (pl < P) red4[pl * cgx + cgl] = sum;
  __syncthreads();
  if (pl == 0 && c < C) {
    f32x4 tot = red4[cgl];
    for (int q = 1; q < P; ++q) tot += red4[q * cgx + cgl];
    *reinterpret_cast<f32x4*>(se_partial + ((size_t)b * nblk + bx) * C + c) = tot;
  }
}

// ---------------------------------------------------------------------------------------------
// Depthwise conv for SMALL planes (H*W <= 1024: MBConv blocks 6-15, 32x32 ... 10x20): one workgroup
// per (sample, channel chunk) loads the whole input plane into LDS once, computes every output pixel from
// LDS and writes ONE squeeze-partial row.  The strip kernel above is latency-bound there (~100 us per
// launch for a few MB: k dependent global-load rounds per thread, 2k+ tiny workgroups); this one has a
// single coalesced load phase.
//   * chunk = 32 bytes of channels per pixel (8 fp32 / 16 bf16): two 16-byte lanes per pixel;
//   * workgroup order is XCD-aware: consecutive workgroup ids round-robin over the 8 XCDs, so the chunks that
//     share a pixel's cache lines are given to ONE XCD (a contiguous run of logical blocks per XCD) — with
//     chunk = blockIdx % nch every 128-byte line was fetched into (and partially written from) 8 different L2s;
//   * a thread computes 4 adjacent output columns of one row for 4 channels: the (3S+K) input vectors of a kernel
//     row are read from LDS once for the 4 outputs (3x fewer LDS reads than one output per thread);
//   * the plane is kept in LDS in the storage type (bf16 stays bf16: 32 KB for 1024 pixels either way).
// ---------------------------------------------------------------------------------------------
template <typename T> struct DwpGeom { static constexpr int CH = 32 / (int)sizeof(T); };   // channels per chunk

template <typename T, int K, int S, bool RAW>
__global__ __launch_bounds__(256) void dwconv_plane_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, T* __restrict__ y,
                                                           float* __restrict__ se_partial, int H, int W, int C, int Ho,
                                                           int Wo, int circular, int total_blocks) {
  constexpr int PB = (S == 1) ? (K - 1) / 2 : (K - 2) / 2;
  constexpr int CH = DwpGeom<T>::CH;          // 8 (fp32) or 16 (bf16) channels = 32 bytes per pixel
  constexpr int CG = CH / 4;                  // 4-channel groups per chunk
  constexpr int NCOL = 3 * S + K;             // input columns feeding 4 adjacent outputs
  constexpr int KYU = K == 3 ? 3 : 1;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  T* plane = reinterpret_cast<T*>(sm);                                   // [H*W][CH] storage type
  float* wl = sm + (size_t)H * W * 8;                                    // [K*K][CH] (32 bytes per pixel = 8 floats)
  float* red = wl + K * K * CH;                                          // [256][4]
  const int tid = threadIdx.x;
  const int nch = C / CH;
  int lb;
  {
    const int q = total_blocks / 8, r = total_blocks % 8;
    const int xcd = blockIdx.x % 8, loc = blockIdx.x / 8;
    lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int chunk = lb % nch;
  const int b = lb / nch;
  const int c0 = chunk * CH;
  const T* xb = x + (size_t)b * H * W * C + c0;
  for (int i = tid; i < H * W * 2; i += 256) {                           // 2 x 16 bytes per pixel
    const int px = i >> 1, q = i & 1;
    *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(plane) + (size_t)px * 32 + q * 16) =
        *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(xb + (size_t)px * C) + q * 16);
  }
  for (int i = tid; i < K * K * CH; i += 256) wl[i] = w[(size_t)(i / CH) * C + c0 + (i % CH)];
  __syncthreads();
  const int cg = tid % CG;
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (!RAW) {
    sc = *reinterpret_cast<const f32x4*>(scale + c0 + cg * 4);
    sh = *reinterpret_cast<const f32x4*>(shift + c0 + cg * 4);
  }
  T* yb = y + (size_t)b * Ho * Wo * C + c0 + cg * 4;
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  const int xg = (Wo + 3) >> 2;                                          // 4-column groups per output row
  for (int it = tid / CG; it < Ho * xg; it += 256 / CG) {
    const int oy = it / xg, ox0 = (it - oy * xg) * 4;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // branch-free window reads (clamped LDS address + lane mask; see dwconv_kernel): one kernel row at a time for k = 5 so
    // that the hoisted reads do not blow up the register count
    typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll KYU
    for (int ky = 0; ky < K; ++ky) {
      const int iy = oy * S - PB + ky;
      const bool rowok = (unsigned)iy < (unsigned)H;
      const int rbase = (rowok ? iy : 0) * W;
      f32x4 col[NCOL];
#pragma unroll
      for (int j = 0; j < NCOL; ++j) {
        int ix = ox0 * S - PB + j;
        const int wrapped = ix < 0 ? ix + W : (ix >= W ? ix - W : ix);
        ix = circular ? wrapped : ix;
        const bool ok = rowok && (unsigned)ix < (unsigned)W;
        const int ixc = (unsigned)ix < (unsigned)W ? ix : 0;
        const int m = ok ? -1 : 0;
        const f32x4 v = ld4<T>(plane + (size_t)(rbase + ixc) * CH + cg * 4);
        col[j] = __builtin_bit_cast(f32x4, __builtin_bit_cast(i32x4, v) & (i32x4){m, m, m, m});
      }
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + (ky * K + kx) * CH + cg * 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] += col[t * S + kx] * wv;
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (ox0 + t >= Wo) continue;
      f32x4 v = acc[t] * sc + sh;
      if (!RAW) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = swishf(v[j]);
      }
      st4<T>(yb + (size_t)(oy * Wo + ox0 + t) * C, v);
      sum += v;
    }
  }
  if (RAW) return;
  *reinterpret_cast<f32x4*>(red + tid * 4) = sum;
  __syncthreads();
  // fixed-order reduction over the 256 / CG item lanes of each channel group: 16 partial sums, then one
  constexpr int LN = 256 / CG;                // threads per channel group (tid = lane * CG + cg)
  const bool lvl1 = tid < 16 * CG;            // (barriers stay outside the branches: 16 * CG can split a wave)
  f32x4 t1 = {0.f, 0.f, 0.f, 0.f};
  if (lvl1) {
    const int g = tid % CG, part = tid / CG;  // part 0..15 sums lanes part, part+16, ...
    for (int i = part; i < LN; i += 16) t1 += *reinterpret_cast<const f32x4*>(red + (i * CG + g) * 4);
  }
  __syncthreads();
  if (lvl1) *reinterpret_cast<f32x4*>(red + tid * 4) = t1;
  __syncthreads();
  if (tid < CG) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 16; ++i) t += *reinterpret_cast<const f32x4*>(red + (i * CG + tid) * 4);
    *reinterpret_cast<f32x4*>(se_partial + (size_t)b * C + c0 + tid * 4) = t;
  }
}

// one rule for both storage types (the SE-partial row count ccvpe_dwconv_nblk reports must not depend on the dtype)
static bool dw_use_plane(int H, int W, int C) { return H * W <= 1024 && C % 16 == 0; }

static void dw_geometry(int H, int W, int C, int stride, int* cgx, int* P, int* ychunks, int* nblk, int* Ho, int* Wo,
                        int k, int* RB) {
  const int total_pad = (stride == 1) ? (k - 1) : (k - 2);
  *Ho = (H + total_pad - k) / stride + 1;
  *Wo = (W + total_pad - k) / stride + 1;
  const int cg = C / 4;
  int yc = 1;
  while (cg / yc > 256 || cg % yc) ++yc;
  *ychunks = yc;
  *cgx = cg / yc;
  *P = 256 / *cgx;
  const int sxn = (*Wo + DW_TW - 1) / DW_TW;
  const int strips = *Ho * sxn;
  const int groups = (strips + *P - 1) / *P;
  // RB strip-groups per workgroup: fewer SE partial rows, still >= ~32 workgroups per sample
  int rb = groups / 32;
  if (rb < 1) rb = 1;
  if (rb > 8) rb = 8;
  *RB = rb;
  *nblk = (groups + rb - 1) / rb;
}

// ---------------------------------------------------------------------------------------------
// SE gate (model.py:113-118): gate = sigmoid(W2 . swish(W1 . mean + b1) + b2).
// A few 100 KFLOP per sample: pure LATENCY.  The round-1..3 kernel (256 threads, 4 outputs per wave and pass, 128-channel FC2
// slices) was a chain of dependent L2 round trips: the per-dispatch trace of round 4 (tools/gpu/se_trace.sh) shows 7-9 us for
// the early blocks but 16 / 24 / 51 us for the 480- / 672- / 1152-channel blocks — 20 of the 32 launches of a forward, 0.75 ms per
// step for microseconds of arithmetic (FC1: 3 passes x 3 round trips over W1; FC2: 24 dependent iterations per slice).
// Now: 1024 threads per workgroup and every phase is ONE round trip — the partial rows are summed by up to 1024 / (C/4) row lanes
// with four 16-byte loads in flight each; FC1 gives each of the 16 waves up to three outputs at once (every load of its three W1
// rows in flight together); FC2 gives a channel of the workgroup's range to 1024 / range threads that split the Cs sum.
// G workgroups per sample each recompute the cheap squeeze + FC1 and own C / G channels of FC2 (G = ceil(256 / B), <= C / 128).
// Fixed summation orders throughout (deterministic).
// ---------------------------------------------------------------------------------------------
constexpr int SE_THREADS = 1024;

__global__ __launch_bounds__(SE_THREADS) void se_gate_kernel(const float* __restrict__ part, int nblk, float inv_hw,
                                                             const float* __restrict__ w1, const float* __restrict__ b1,
                                                             const float* __restrict__ w2, const float* __restrict__ b2,
                                                             float* __restrict__ gate, int C, int Cs, int cpw) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // mean[C^4] | z[Cs^4] | red[1024][4]
  float* mean = sm;
  float* z = sm + ((C + 3) & ~3);
  float* red = z + ((Cs + 3) & ~3);       // 16-byte aligned: the row-lane sums are float4
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  // ---- squeeze: mean over the plane = inv_hw * sum of the nblk partial rows ------------------------------------------
  if ((C & 3) == 0) {
    const int C4 = C >> 2;
    const int cw = C4 < SE_THREADS ? C4 : SE_THREADS;
    const int R = SE_THREADS / cw;                 // row lanes (fixed assignment of rows to lanes => deterministic)
    f32x4* red4 = reinterpret_cast<f32x4*>(red);
    for (int c0 = 0; c0 < C4; c0 += cw) {
      const int cl = tid % cw, rr = tid / cw;
      const int c4 = c0 + cl;
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
      if (rr < R && c4 < C4) {
        const f32x4* p = reinterpret_cast<const f32x4*>(part + (size_t)b * nblk * C) + c4;
        int q = rr;
        for (; q + 3 * R < nblk; q += 4 * R) {     // 4 independent loads in flight
          s0 += p[(size_t)q * C4];
          s1 += p[(size_t)(q + R) * C4];
          s2 += p[(size_t)(q + 2 * R) * C4];
          s3 += p[(size_t)(q + 3 * R) * C4];
        }
        for (; q < nblk; q += R) s0 += p[(size_t)q * C4];
      }
      if (R == 1) {
        if (c4 < C4 && rr == 0) *reinterpret_cast<f32x4*>(mean + 4 * c4) = ((s0 + s1) + (s2 + s3)) * inv_hw;
      } else {
        red4[tid] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (rr == 0 && c4 < C4) {
          f32x4 t = red4[cl];
          for (int j = 1; j < R; ++j) t += red4[j * cw + cl];
          *reinterpret_cast<f32x4*>(mean + 4 * c4) = t * inv_hw;
        }
      }
      __syncthreads();
    }
  } else {
    const int cw = C < SE_THREADS ? C : SE_THREADS;
    const int R = SE_THREADS / cw;
    for (int c0 = 0; c0 < C; c0 += cw) {
      const int cl = tid % cw, rr = tid / cw;
      const int c = c0 + cl;
      float s0 = 0.f, s1 = 0.f;
      if (rr < R && c < C) {
        const float* p = part + (size_t)b * nblk * C + c;
        int q = rr;
        for (; q + R < nblk; q += 2 * R) {
          s0 += p[(size_t)q * C];
          s1 += p[(size_t)(q + R) * C];
        }
        if (q < nblk) s0 += p[(size_t)q * C];
      }
      red[tid] = s0 + s1;
      __syncthreads();
      if (rr == 0 && c < C) {
        float t = red[cl];
        for (int j = 1; j < R; ++j) t += red[j * cw + cl];
        mean[c] = t * inv_hw;
      }
      __syncthreads();
    }
  }
  // ---- FC1: wave w owns outputs w, w + 16, w + 32 (three at once: all of their W1 loads are in flight together) -------
  for (int j0 = wave; j0 < Cs; j0 += 48) {
    const int j1 = min(j0 + 16, Cs - 1), j2 = min(j0 + 32, Cs - 1);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if ((C & 3) == 0) {
      const int C4 = C >> 2;
      const f32x4* r0 = reinterpret_cast<const f32x4*>(w1 + (size_t)j0 * C);
      const f32x4* r1 = reinterpret_cast<const f32x4*>(w1 + (size_t)j1 * C);
      const f32x4* r2 = reinterpret_cast<const f32x4*>(w1 + (size_t)j2 * C);
#pragma unroll 5                             // C = 1152: 4.5 x 3 sixteen-byte loads per lane, one round trip
      for (int c4 = lane; c4 < C4; c4 += 64) {
        const f32x4 m = *reinterpret_cast<const f32x4*>(mean + 4 * c4);
        const f32x4 a = r0[c4], bb = r1[c4], cc = r2[c4];
        s0 += a[0] * m[0] + a[1] * m[1] + a[2] * m[2] + a[3] * m[3];
        s1 += bb[0] * m[0] + bb[1] * m[1] + bb[2] * m[2] + bb[3] * m[3];
        s2 += cc[0] * m[0] + cc[1] * m[1] + cc[2] * m[2] + cc[3] * m[3];
      }
    } else {
      for (int c = lane; c < C; c += 64) {
        const float m = mean[c];
        s0 = fmaf(w1[(size_t)j0 * C + c], m, s0);
        s1 = fmaf(w1[(size_t)j1 * C + c], m, s1);
        s2 = fmaf(w1[(size_t)j2 * C + c], m, s2);
      }
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) {
      z[j0] = swishf(s0 + b1[j0]);
      if (j0 + 16 < Cs) z[j0 + 16] = swishf(s1 + b1[j0 + 16]);
      if (j0 + 32 < Cs) z[j0 + 32] = swishf(s2 + b1[j0 + 32]);
    }
  }
  __syncthreads();
  // ---- FC2 for this workgroup's channels [c_lo, c_lo + cpw): JG threads per channel split the Cs sum (w2 is [Cs][C]:
  // coalesced over c), every load of a thread in flight together -------------------------------------------------------
  const int c_lo = blockIdx.x * cpw;
  const int cwid = cpw < SE_THREADS ? cpw : SE_THREADS;
  const int JG = SE_THREADS / cwid;
  for (int cb = 0; cb < cpw; cb += cwid) {
    const int cg = tid % cwid, jg = tid / cwid;
    const int c = c_lo + cb + cg;
    float s = 0.f;
    if (jg < JG && cb + cg < cpw && c < C) {
#pragma unroll 8
      for (int j = jg; j < Cs; j += JG) s = fmaf(w2[(size_t)j * C + c], z[j], s);
    }
    if (cb) __syncthreads();                 // red[] of the previous chunk has been consumed
    red[tid] = s;
    __syncthreads();
    if (jg == 0 && cb + cg < cpw && c < C) {
      float t = red[cg];
      for (int j = 1; j < JG; ++j) t += red[j * cwid + cg];
      gate[(size_t)b * C + c] = sigmoidf(t + b2[c]);
    }
  }
}

}  // namespace ccvpe

using namespace ccvpe;

template <typename T, bool RAW = false>
static int stem_any(const float* x, const float* w, const float* scale, const float* shift, T* y, int B, int H, int W,
                    int circular, void* stream) {
  if (B <= 0 || H < 3 || W < 3) return fail(CCVPE_EINVAL, "stem: bad shape");
  if (!aligned16(y)) return fail(CCVPE_EINVAL, "stem: y must be 16-byte aligned");
  const int Ho = (H + 1 - 3) / 2 + 1, Wo = (W + 1 - 3) / 2 + 1;
  const int tiles_x = (Wo + ST_TW - 1) / ST_TW, tiles_y = (Ho + ST_TH - 1) / ST_TH;
  const long total = (long)B * tiles_x * tiles_y;
  if (total > 0x7fffffffL) return fail(CCVPE_EINVAL, "stem: grid too large");
  hipLaunchKernelGGL((stem_conv_kernel<T, RAW>), dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, x, w, scale, shift, y,
                     B, H, W, Ho, Wo, circular, tiles_x, tiles_y);
  return check_launch("stem_conv_kernel");
}

extern "C" int ccvpe_stem_conv_f32(const float* x, const float* w, const float* scale, const float* shift, float* y,
                                   int B, int H, int W, int circular, void* stream) {
  return stem_any<float>(x, w, scale, shift, y, B, H, W, circular, stream);
}
extern "C" int ccvpe_stem_conv_raw_f32(const float* x, const float* w, float* y, int B, int H, int W, int circular,
                                       void* stream) {
  return stem_any<float, true>(x, w, nullptr, nullptr, y, B, H, W, circular, stream);
}
extern "C" int ccvpe_stem_conv_bf16(const float* x, const float* w, const float* scale, const float* shift, void* y,
                                    int B, int H, int W, int circular, void* stream) {
  return stem_any<cc_bf16>(x, w, scale, shift, reinterpret_cast<cc_bf16*>(y), B, H, W, circular, stream);
}

extern "C" int ccvpe_dwconv_nblk(int H, int W, int C, int stride) {
  // nblk does not depend on k for the SAME schedule (Ho = ceil-like of H/stride for both k)
  int cgx, P, yc, nblk, Ho, Wo, RB;
  if (C <= 0 || C % 4) return CCVPE_EINVAL;
  if (dw_use_plane(H, W, C)) {
    const int nb = (g_mbplane_mode & 2) ? mbplane_nblk(H, W, 0, C, 5, stride) : 0;   // row bands of csrc/mbconv_plane.hip
    return nb > 0 ? nb : 1;
  }
  dw_geometry(H, W, C, stride, &cgx, &P, &yc, &nblk, &Ho, &Wo, 3, &RB);
  return nblk;
}

template <typename T, bool RAW = false>
static int dwconv_any(const T* x, const float* w, const float* scale, const float* shift, T* y, float* se_partial, int B,
                      int H, int W, int C, int k, int stride, int circular, void* stream) {
  if (C <= 0 || C % 4) return fail(CCVPE_EINVAL, "dwconv: C %% 4 != 0");
  if (!(k == 3 || k == 5) || !(stride == 1 || stride == 2)) return fail(CCVPE_EINVAL, "dwconv: k/stride unsupported");
  if (!aligned16(x) || !aligned16(w) || !aligned16(y) || (!RAW && (!aligned16(se_partial) || !aligned16(scale) ||
      !aligned16(shift))))
    return fail(CCVPE_EINVAL, "dwconv: pointers must be 16-byte aligned");
  int cgx, P, yc, nblk, Ho, Wo, RB;
  dw_geometry(H, W, C, stride, &cgx, &P, &yc, &nblk, &Ho, &Wo, k, &RB);
  if (circular && (k / 2 + 1 > W)) return fail(CCVPE_EINVAL, "dwconv: W too small for circular wrap");
  if (!RAW && dw_use_plane(H, W, C) && (g_mbplane_mode & 2) && mbplane_nblk(H, W, 0, C, k, stride) > 0)
    return mbplane_launch(sizeof(T) == 2, 0, x, nullptr, 0, nullptr, nullptr, w, scale, shift, y, se_partial, B, H, W, 0, C, k,
                          stride, circular, stream);
  if (dw_use_plane(H, W, C)) {
    constexpr int CH = DwpGeom<T>::CH;
    const size_t psm = ((size_t)H * W * 8 + (size_t)k * k * CH + 1024) * sizeof(float);
    const int pblocks = (C / CH) * B;
    dim3 pgrid((unsigned)pblocks);
    hipStream_t pst = (hipStream_t)stream;
#define DWP_LAUNCH(K_, S_) \
  hipLaunchKernelGGL((dwconv_plane_kernel<T, K_, S_, RAW>), pgrid, dim3(256), psm, pst, x, w, scale, shift, y, se_partial, H, W, C, Ho, Wo, circular, pblocks)
    if (k == 3 && stride == 1) DWP_LAUNCH(3, 1);
    else if (k == 3 && stride == 2) DWP_LAUNCH(3, 2);
    else if (k == 5 && stride == 1) DWP_LAUNCH(5, 1);
    else DWP_LAUNCH(5, 2);
#undef DWP_LAUNCH
    return check_launch("dwconv_plane_kernel");
  }
  const long total = (long)nblk * yc * B;
  if (total > 0x7fffffffL) return fail(CCVPE_EINVAL, "dwconv: grid too large");
  dim3 grid((unsigned)total);
  const size_t smem = (size_t)P * cgx * 16;
  hipStream_t st = (hipStream_t)stream;
#define DW_LAUNCH(K_, S_)                                                                                         \
  hipLaunchKernelGGL((dwconv_kernel<T, K_, S_, RAW>), grid, dim3(256), smem, st, x, w, scale, shift, y, se_partial, H, W, \
                     C, Ho, Wo, cgx, P, nblk, circular, RB, yc, (int)total)
  if (k == 3 && stride == 1) DW_LAUNCH(3, 1);
  else if (k == 3 && stride == 2) DW_LAUNCH(3, 2);
  else if (k == 5 && stride == 1) DW_LAUNCH(5, 1);
  else DW_LAUNCH(5, 2);
#undef DW_LAUNCH
  return check_launch("dwconv_kernel");
}

extern "C" int ccvpe_dwconv_f32(const float* x, const float* w, const float* scale, const float* shift, float* y,
                                float* se_partial, int B, int H, int W, int C, int k, int stride, int circular,
                                void* stream) {
  return dwconv_any<float>(x, w, scale, shift, y, se_partial, B, H, W, C, k, stride, circular, stream);
}
extern "C" int ccvpe_dwconv_raw_f32(const float* x, const float* w, float* y, int B, int H, int W, int C, int k,
                                    int stride, int circular, void* stream) {
  return dwconv_any<float, true>(x, w, nullptr, nullptr, y, nullptr, B, H, W, C, k, stride, circular, stream);
}
extern "C" int ccvpe_dwconv_bf16(const void* x, const float* w, const float* scale, const float* shift, void* y,
                                 float* se_partial, int B, int H, int W, int C, int k, int stride, int circular,
                                 void* stream) {
  return dwconv_any<cc_bf16>(reinterpret_cast<const cc_bf16*>(x), w, scale, shift, reinterpret_cast<cc_bf16*>(y),
                             se_partial, B, H, W, C, k, stride, circular, stream);
}

extern "C" int ccvpe_se_gate_f32(const float* part, int nblk, float inv_hw, const float* w1, const float* b1,
                                 const float* w2, const float* b2, float* gate, int B, int C, int Cs, void* stream) {
  if (B <= 0 || C <= 0 || Cs <= 0 || nblk <= 0) return fail(CCVPE_EINVAL, "se_gate: bad shape");
  const size_t smem = (size_t)(((C + 3) & ~3) + ((Cs + 3) & ~3) + 4 * SE_THREADS) * sizeof(float);
  const int slices = (C + 127) / 128;
  int g = (256 + B - 1) / B;                 // workgroups per sample (see the kernel's comment)
  g = g < 1 ? 1 : (g > slices ? slices : g);
  const int cpw = (((C + g - 1) / g) + 3) & ~3;        // channels per workgroup
  static bool attr_set = false;
  if (!attr_set && smem > 48 * 1024) {
    if (hipFuncSetAttribute((const void*)se_gate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess)
      return fail(CCVPE_ELAUNCH, "se_gate: set smem attr");
    attr_set = true;
  }
  if (smem > 64 * 1024) return fail(CCVPE_EINVAL, "se_gate: C = %d is too wide", C);
  hipLaunchKernelGGL(se_gate_kernel, dim3((C + cpw - 1) / cpw, B), dim3(SE_THREADS), smem, (hipStream_t)stream, part,
                     nblk, inv_hw, w1, b1, w2, b2, gate, C, Cs, cpw);
  return check_launch("se_gate_kernel");
}
