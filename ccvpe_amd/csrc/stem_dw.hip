// Stem + MBConv block 0 front half in ONE kernel (eval): 3x3 stride-2 stem conv + BN + swish  ->  depthwise 3x3 stride 1 + BN +
// swish + SE squeeze partial sums.  efficientnet_pytorch/model.py:181-182,289 (stem), :108-110,114 (block 0 has expand_ratio 1:
// no expand conv, the depthwise conv reads the stem output directly).
//
// Why: the two unfused kernels write the [B, H/2, W/2, 32] stem tensor and read it back (2 x 268 MB per aerial batch of 64 in
// bf16, 2 x 537 MB in fp32) and the strip depthwise kernel reads each of its inputs 4.5 times through the vector cache in
// half-line pieces: 141 + 244 us (aerial) and 113 + 191 us (ground) of the B = 64 bf16 forward for ~75 us of HBM traffic.
// Here the stem tensor lives only in LDS.
//
// A workgroup owns (sample, 8 x 32 output pixels):
//   1. the 21 x 69 x 3 image patch under the 10 x 34 stem pixels of the tile + halo goes to LDS with coalesced row loads (all of a
//      thread's loads in flight together; circular padding is a column wrap of the source, zero padding a lane mask);
//   2. stem: out[pixel][32] = patch[pixel][27] . W[27][32] on the fp32 matrix cores exactly as stem_conv_kernel does it (K padded
//      to 28 = 7 k-steps, W resident in 14 registers, the B operand one ds_read_b32 per k-step) over the 340 halo pixels numbered
//      linearly (22 tiles of 16, no idle lanes whatever the row width), BN0 + swish, ds_write_b128 into the stem tile
//      [10 rows][40 px][36 floats].  Stem pixels outside the image are ZEROS (the depthwise conv zero-pads the STEM OUTPUT); with
//      circular padding the columns -1 and Wo are the wrapped stem columns, which the wrapped patch columns produce by themselves
//      when W is even (W = 2 Wo: checked by the launcher);
//   3. depthwise: lane = (4-channel group lane & 7, output row lane >> 3), wave = 8 adjacent output columns: the 3 x 10 input
//      vectors of a thread's 8 outputs are read once (ds_read_b128), 9 taps from registers, BN1 + swish, 8- or 16-byte stores, and
//      the per-channel sums of the tile for the squeeze (fixed butterfly + fixed order over the 4 waves: bit-reproducible).
// LDS geometry: pixel pitch 36 floats = 9 sixteen-byte slots makes the 8 lanes of a ds_write_b128 group (8 consecutive pixels, one
// channel group) land in 8 different slots; row pitch 40 pixels = 8 (mod 16) slots... x 9 makes a ds_read_b128 lane group
// {row r: channel groups 0-3, row r+1: 4-7, row r+2: 4-7, row r+3: 0-3} cover the 16 slots of a 256-byte bank row exactly once.
// The arithmetic is that of the unfused kernels in the same order (fp32 MFMA = k-ordered fmaf chain; taps ky-major), so the fp32
// output is bit-identical to stem_conv + dwconv; in bf16 storage the stem tensor is NOT rounded to bf16 on its way to the depthwise
// conv (one rounding fewer than the unfused pair).
#include "common.h"

namespace ccvpe {

typedef float sd_f32x4 __attribute__((ext_vector_type(4)));
typedef int sd_i32x4 __attribute__((ext_vector_type(4)));

constexpr int SD_TH = 8, SD_TW = 32;                         // output tile
constexpr int SD_SH = SD_TH + 2, SD_SW = SD_TW + 2;          // stem pixels with halo: 10 x 34
constexpr int SD_NPX = SD_SH * SD_SW;                        // 340
constexpr int SD_NT = (SD_NPX + 15) / 16;                    // 22 MFMA pixel tiles
constexpr int SD_IR = 2 * SD_SH + 1, SD_IC = 2 * SD_SW + 1;  // image patch 21 x 69
constexpr int SD_IPITCH = 72;
constexpr int SD_PP = 36;                                    // stem tile: floats per pixel
constexpr int SD_RP = 40;                                    // stem tile: pixels per row
constexpr int SD_IMG_FLOATS = 3 * SD_IR * SD_IPITCH;         // 4536
constexpr int SD_TILE_FLOATS = SD_SH * SD_RP * SD_PP;        // 14400
constexpr int SD_LDS_FLOATS = SD_IMG_FLOATS + SD_TILE_FLOATS + 4 * 8 * 4;

template <typename T>
__global__ __launch_bounds__(256, 2) void stem_dw_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ s0, const float* __restrict__ b0,
                                                         const float* __restrict__ wdw, const float* __restrict__ s1,
                                                         const float* __restrict__ b1, T* __restrict__ y,
                                                         float* __restrict__ se_partial, int H, int W, int Ho, int Wo,
                                                         int circular, int tiles_x, int tiles_y) {
  extern __shared__ __attribute__((aligned(16))) float sd_sm[];
  float* img = sd_sm;                                  // [3][21][72]
  float* tile = sd_sm + SD_IMG_FLOATS;                 // [10][40][36]
  float* red = tile + SD_TILE_FLOATS;                  // [4 waves][8 channel groups][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pix = lane & 15, q = lane >> 4;
  int lb;
  {                                                    // XCD-aware order: neighbouring tiles of a sample share patch rows in one L2
    const int total = gridDim.x;
    const int qq = total / 8, r = total % 8;
    const int xcd = blockIdx.x % 8, loc = blockIdx.x / 8;
    lb = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + loc;
  }
  const int tx = lb % tiles_x, ty = (lb / tiles_x) % tiles_y, b = lb / (tiles_x * tiles_y);
  const int oy0 = ty * SD_TH, ox0 = tx * SD_TW;
  // ---- 1. the image patch: rows 2 (oy0 - 1) .., columns 2 (ox0 - 1) .. of the three planes ---------------------------------
  const float* xb = x + (size_t)b * 3 * H * W;
  {
    constexpr int NIT = (SD_IMG_FLOATS + 255) / 256;   // 18
    float v[NIT];
    unsigned okm = 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = min(tid + 256 * it, SD_IMG_FLOATS - 1);
      const int row = idx / SD_IPITCH, c = idx - row * SD_IPITCH;
      const int ci = row / SD_IR, r = row - ci * SD_IR;
      const int iy = 2 * (oy0 - 1) + r;
      int ix = 2 * (ox0 - 1) + c;
      const int wrapped = ix < 0 ? ix + W : (ix >= W ? ix - W : ix);
      ix = circular ? wrapped : ix;
      const bool ok = c < SD_IC && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      v[it] = xb[ok ? ((size_t)ci * H + iy) * W + ix : 0];
      okm |= ok ? (1u << it) : 0u;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + 256 * it;
      if (idx < SD_IMG_FLOATS) img[idx] = ((okm >> it) & 1u) ? v[it] : 0.f;
    }
  }
  // ---- stem W fragments (A operand: lane (n = pix, q) supplies W[k = 4 s + q][n]) and the lane's patch offsets ----------------
  float wr[7][2];
  int koff[7];
#pragma unroll
  for (int s = 0; s < 7; ++s) {
    const int k = 4 * s + q;
    const bool kv = k < 27;
    const int kc = kv ? k : 0;
    const float w0 = w[kc * 32 + pix], w1 = w[kc * 32 + 16 + pix];
    wr[s][0] = kv ? w0 : 0.f;
    wr[s][1] = kv ? w1 : 0.f;
    const int tap = kc / 3, ci = kc - tap * 3;
    const int ky = tap / 3, kx = tap - ky * 3;
    koff[s] = (ci * SD_IR + ky) * SD_IPITCH + kx;
  }
  sd_f32x4 sc[2], sh[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    sc[j] = *reinterpret_cast<const sd_f32x4*>(s0 + 16 * j + 4 * q);
    sh[j] = *reinterpret_cast<const sd_f32x4*>(b0 + 16 * j + 4 * q);
  }
  // ---- depthwise weights and BN1 of this lane's 4 channels (in flight under the stem phase) ------------------------------------
  const int cg = lane & 7, orow = lane >> 3;           // channel group, output row of the tile
  sd_f32x4 wd[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wd[t] = *reinterpret_cast<const sd_f32x4*>(wdw + t * 32 + cg * 4);
  const sd_f32x4 sc1 = *reinterpret_cast<const sd_f32x4*>(s1 + cg * 4);
  const sd_f32x4 sh1 = *reinterpret_cast<const sd_f32x4*>(b1 + cg * 4);
  __syncthreads();
  // ---- 2. stem on the matrix cores over the linearly numbered halo pixels ------------------------------------------------------
  for (int t = wave; t < SD_NT; t += 4) {
    const int p = 16 * t + pix;
    const int pc = p < SD_NPX ? p : SD_NPX - 1;
    const int hy = pc / SD_SW, hx = pc - hy * SD_SW;
    const int base = 2 * hy * SD_IPITCH + 2 * hx;
    float xv[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) xv[s] = img[base + koff[s]];
    sd_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
    for (int s = 0; s < 7; ++s) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[s][0], xv[s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[s][1], xv[s], acc1, 0, 0, 0);
    }
    // D: row = channel 4 q + r (+16 for acc1), column = pixel `pix`
    const int sy = oy0 - 1 + hy, sx = ox0 - 1 + hx;
    const bool inside = (unsigned)sy < (unsigned)Ho && (circular || (unsigned)sx < (unsigned)Wo);
    sd_f32x4 v0 = acc0 * sc[0] + sh[0], v1 = acc1 * sc[1] + sh[1];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v0[r] = inside ? swishf(v0[r]) : 0.f;
      v1[r] = inside ? swishf(v1[r]) : 0.f;
    }
    if (p < SD_NPX) {
      float* dst = tile + (hy * SD_RP + hx) * SD_PP + 4 * q;
      *reinterpret_cast<sd_f32x4*>(dst) = v0;
      *reinterpret_cast<sd_f32x4*>(dst + 16) = v1;
    }
  }
  __syncthreads();
  // ---- 3. depthwise 3x3: 8 adjacent outputs of row `orow`, columns 8 wave .. 8 wave + 7, 4 channels -----------------------------
  sd_f32x4 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) acc[t] = (sd_f32x4){0.f, 0.f, 0.f, 0.f};
  const float* trow = tile + (orow * SD_RP + 8 * wave) * SD_PP + cg * 4;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    sd_f32x4 col[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) col[j] = *reinterpret_cast<const sd_f32x4*>(trow + (ky * SD_RP + j) * SD_PP);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] += col[t + kx] * wd[ky * 3 + kx];
    }
  }
  const int oy = oy0 + orow;
  const bool row_ok = oy < Ho;
  T* yrow = y + (((size_t)b * Ho + (row_ok ? oy : 0)) * Wo) * 32 + cg * 4;
  sd_f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int ox = ox0 + 8 * wave + t;
    sd_f32x4 o = acc[t] * sc1 + sh1;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = swishf(o[r]);
    if (row_ok && ox < Wo) {
      st4<T>(yrow + (size_t)ox * 32, o);
      sum += o;
    }
  }
  // squeeze partial of the tile: lanes that share a channel group differ in lane bits 3..5
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) sum[r] += __shfl_xor(sum[r], o, 64);
  }
  if (lane < 8) *reinterpret_cast<sd_f32x4*>(red + (wave * 8 + lane) * 4) = sum;
  __syncthreads();
  if (tid < 8) {
    sd_f32x4 t = *reinterpret_cast<const sd_f32x4*>(red + tid * 4);
#pragma unroll
    for (int wv = 1; wv < 4; ++wv) t += *reinterpret_cast<const sd_f32x4*>(red + (wv * 8 + tid) * 4);
    *reinterpret_cast<sd_f32x4*>(se_partial + ((size_t)b * (tiles_x * tiles_y) + ty * tiles_x + tx) * 32 + tid * 4) = t;
  }
}

static bool stem_dw_ok(int H, int W, int circular) {
  if (H < 3 || W < 3) return false;
  if (circular && (W & 1)) return false;               // the wrapped stem columns come from wrapped image columns only if W = 2 Wo
  return true;
}

}  // namespace ccvpe

using namespace ccvpe;

// squeeze-partial rows per sample (= output tiles per sample); 0: this shape runs the unfused kernels
extern "C" int ccvpe_stem_dw_nblk(int H, int W, int circular) {
  if (!stem_dw_ok(H, W, circular)) return 0;
  const int Ho = (H + 1 - 3) / 2 + 1, Wo = (W + 1 - 3) / 2 + 1;
  return ((Ho + SD_TH - 1) / SD_TH) * ((Wo + SD_TW - 1) / SD_TW);
}

template <typename T>
static int stem_dw_any(const float* x, const float* w, const float* s0, const float* b0, const float* wdw, const float* s1,
                       const float* b1, T* y, float* se_partial, int B, int H, int W, int circular, void* stream) {
  if (B <= 0 || !stem_dw_ok(H, W, circular)) return fail(CCVPE_EINVAL, "stem_dw: shape not supported (H=%d W=%d circular=%d)", H, W, circular);
  if (!aligned16(y) || !aligned16(se_partial) || !aligned16(s0) || !aligned16(b0) || !aligned16(wdw) || !aligned16(s1) ||
      !aligned16(b1))
    return fail(CCVPE_EINVAL, "stem_dw: pointers must be 16-byte aligned");
  const int Ho = (H + 1 - 3) / 2 + 1, Wo = (W + 1 - 3) / 2 + 1;
  const int tiles_x = (Wo + SD_TW - 1) / SD_TW, tiles_y = (Ho + SD_TH - 1) / SD_TH;
  const long total = (long)B * tiles_x * tiles_y;
  if (total > 0x7fffffffL) return fail(CCVPE_EINVAL, "stem_dw: grid too large");
  constexpr int lds = SD_LDS_FLOATS * 4;
  static_assert(lds <= 80 * 1024, "two workgroups per CU");
  hipError_t e = hipFuncSetAttribute((const void*)stem_dw_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return fail(CCVPE_ELAUNCH, "stem_dw: set smem attr: %s", hipGetErrorString(e));
  hipLaunchKernelGGL((stem_dw_kernel<T>), dim3((unsigned)total), dim3(256), lds, (hipStream_t)stream, x, w, s0, b0, wdw, s1, b1, y,
                     se_partial, H, W, Ho, Wo, circular, tiles_x, tiles_y);
  return check_launch("stem_dw_kernel");
}

extern "C" int ccvpe_stem_dw_f32(const float* x, const float* w, const float* s0, const float* b0, const float* wdw,
                                 const float* s1, const float* b1, float* y, float* se_partial, int B, int H, int W,
                                 int circular, void* stream) {
  return stem_dw_any<float>(x, w, s0, b0, wdw, s1, b1, y, se_partial, B, H, W, circular, stream);
}
extern "C" int ccvpe_stem_dw_bf16(const float* x, const float* w, const float* s0, const float* b0, const float* wdw,
                                  const float* s1, const float* b1, void* y, float* se_partial, int B, int H, int W,
                                  int circular, void* stream) {
  return stem_dw_any<cc_bf16>(x, w, s0, b0, wdw, s1, b1, reinterpret_cast<cc_bf16*>(y), se_partial, B, H, W, circular, stream);
}
