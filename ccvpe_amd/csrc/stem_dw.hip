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
// LDS geometry: pixel pitch 36 floats = 9 sixteen-byte slots puts the 8 lanes of a ds_write_b128 group (8 consecutive pixels, one
// channel group) into 8 different slots; row pitch 40 pixels = 360 slots = 8 (mod 16) makes a ds_read_b128 lane group
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
constexpr int SD_LDS_FLOATS = SD_IMG_FLOATS + SD_TILE_FLOATS + 4 * 8 * 4 + 9 * 32;

// Diagnostics builds only (-DCCVPE_ABLATE, tools/gpu/ablate_stem_dw.sh): CCVPE_STEM_DW_ABLATE bit 1 = no patch loads after the
// first tile, 2 = no stem MFMAs, 4 = no stem swish, 8 = no depthwise reads / FMAs, 16 = no output swish, 32 = no output stores.
#ifdef CCVPE_ABLATE
#define SD_ABL(a) const int abl = (a)
static int sd_ablate_env() { const char* e = getenv("CCVPE_STEM_DW_ABLATE"); return e ? atoi(e) : 0; }
#else
#define SD_ABL(a) constexpr int abl = 0
static int sd_ablate_env() { return 0; }
#endif

constexpr int SD_TPW = (SD_NT + 3) / 4;                      // 6 MFMA pixel tiles per wave (waves 2, 3: 5)

// PERSISTENT workgroups (grid = 2 per CU, the LDS limit): the per-workgroup set-up (stem weights, BN vectors, lane geometry) is
// paid once and the patch of a workgroup's NEXT tile is fetched between the stem and the depthwise phase of the current one.  Tiles are dealt so that each XCD walks a contiguous run of tiles (vertical neighbours share 5 of 21 patch
// rows through one L2).
// Measured (B = 64 aerial, bf16 output): 310 us against 177 + 258 us for the two unfused launches; fp32 output 327 against
// 210 + 320.  Ablation (tools/gpu/ablate_stem_dw.sh): no stem MFMAs -75 us, no stem swish -30, no depthwise reads/FMAs -43, no
// output swish -24, no stores -27, no patch loads -36, everything off 100 us: the parts ADD UP — with two waves per SIMD nothing
// overlaps, and the fp32 MFMA runs at the vector-fp32 rate (157 TF, MI355X_MICROARCH.md), so the 84 MFMAs of a wave-tile
// (2 688 cycles) cost what the ~750 vector instructions beside them cost.  Faster needs more waves per SIMD (a bf16 stem tile in
// LDS and < 170 registers for three workgroups per CU) — not built.
template <typename T>
__global__ __launch_bounds__(256, 2) void stem_dw_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ s0, const float* __restrict__ b0,
                                                         const float* __restrict__ wdw, const float* __restrict__ s1,
                                                         const float* __restrict__ b1, T* __restrict__ y,
                                                         float* __restrict__ se_partial, int H, int W, int Ho, int Wo,
                                                         int circular, int tiles_x, int tiles_y, int total_tiles, int abl_arg) {
  SD_ABL(abl_arg);
  extern __shared__ __attribute__((aligned(16))) float sd_sm[];
  float* img = sd_sm;                                  // [3][21][72]
  float* tile = sd_sm + SD_IMG_FLOATS;                 // [10][40][36]
  float* red = tile + SD_TILE_FLOATS;                  // [4 waves][8 channel groups][4]
  float* wdl = red + 4 * 8 * 4;                        // depthwise weights [9 taps][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pix = lane & 15, q = lane >> 4;
  // this workgroup's tiles: XCD x (= blockIdx % 8) owns tiles [T x / 8, T (x + 1) / 8), its workgroups take them round-robin
  const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
  const int t_end = (int)(((long)total_tiles * (xcd + 1)) >> 3);
  int t_cur = (int)(((long)total_tiles * xcd) >> 3) + wg_in_xcd;
  if (t_cur >= t_end) return;

  // ---- the patch: thread (plane pl = tid / 72, column pc = tid % 72) fetches its column of all 21 patch rows (216 of the 256
  // threads).  Row r of the patch is ONE scalar base (+ r W floats) for the whole workgroup and a thread's lane offset is the same
  // for its 21 loads, so a load costs no vector instruction besides itself; row validity is a scalar bit, column validity (and
  // the circular wrap) is computed once per tile per thread. ------------------------------------------------------------------
  const int pl = tid / SD_IPITCH, pc = tid - pl * SD_IPITCH;
  const bool loader = tid < 3 * SD_IPITCH;
  float pv[SD_IR];                                     // the patch column in flight
  unsigned rowmask = 0;                                // bit r: patch row r is inside the image (scalar)
  bool colok = false;
  auto request_patch = [&](int t) {
    const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
    const int iy0 = 2 * (ty * SD_TH - 1), ix0 = 2 * (tx * SD_TW - 1);
    int ix = ix0 + (pc < SD_IC ? pc : SD_IC - 1);      // pad columns 69..71 re-read column 68 (never used)
    const int wrapped = ix < 0 ? ix + W : (ix >= W ? ix - W : ix);
    ix = circular ? wrapped : ix;
    colok = loader && (unsigned)ix < (unsigned)W;
    // buffer loads: the sample's image is the buffer, a patch row is the SCALAR offset, the thread's column the lane offset
    // (a lane offset beyond the buffer reads as 0: columns outside the image, idle threads)
    const unsigned voff = colok ? (unsigned)((pl * H * W + ix) * 4) : 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (size_t)b * 3 * H * W), 0, 3 * H * W * 4, 0x00020000);
    rowmask = 0;
#pragma unroll
    for (int r = 0; r < SD_IR; ++r) {
      const int iy = iy0 + r;
      const bool rok = (unsigned)iy < (unsigned)H;
      rowmask |= rok ? (1u << r) : 0u;
      pv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (rok ? iy : 0) * W * 4, 0));
    }
  };
  auto park_patch = [&]() {
    if (loader) {
      float* dst = img + pl * SD_IR * SD_IPITCH + pc;
#pragma unroll
      for (int r = 0; r < SD_IR; ++r) dst[r * SD_IPITCH] = ((rowmask >> r) & 1u) ? pv[r] : 0.f;
    }
  };
  request_patch(t_cur);

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
  // this wave's MFMA pixel tiles i: halo pixel p = 16 (wave + 4 i) + pix -> patch read base, stem-tile write offset, (hy, hx)
  int rbase[SD_TPW], wofs[SD_TPW];
#pragma unroll
  for (int i = 0; i < SD_TPW; ++i) {
    const int p = 16 * (wave + 4 * i) + pix;
    const int pc = p < SD_NPX ? p : SD_NPX - 1;
    const int hy = pc / SD_SW, hx = pc - hy * SD_SW;
    rbase[i] = 2 * hy * SD_IPITCH + 2 * hx;
    wofs[i] = p < SD_NPX ? (hy * SD_RP + hx) * SD_PP + 4 * q : -1;
  }
  // ---- depthwise weights and BN1 of this lane's 4 channels -------------------------------------------------------------------
  const int cg = lane & 7, orow = lane >> 3;           // channel group, output row of the tile
  for (int i = tid; i < 9 * 32; i += 256) wdl[i] = wdw[i];
  // the nine depthwise weight vectors of this lane's channels live in registers for the whole (persistent) kernel: read from LDS
  // inside the tile loop they sat behind the window reads in the in-order LDS return queue (round 6, see csrc/mbconv_plane.hip)
  sd_f32x4 wreg[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) wreg[i] = *reinterpret_cast<const sd_f32x4*>(wdw + i * 32 + cg * 4);
  const sd_f32x4 sc1 = *reinterpret_cast<const sd_f32x4*>(s1 + cg * 4);
  const sd_f32x4 sh1 = *reinterpret_cast<const sd_f32x4*>(b1 + cg * 4);
  const float* trow = tile + (orow * SD_RP + 8 * wave) * SD_PP + cg * 4;

  park_patch();
  __syncthreads();

  for (; t_cur < t_end; t_cur += wgs_per_xcd) {
    const int tx = t_cur % tiles_x, ty = (t_cur / tiles_x) % tiles_y, b = t_cur / (tiles_x * tiles_y);
    const int oy0 = ty * SD_TH, ox0 = tx * SD_TW;
    const int t_next = t_cur + wgs_per_xcd;
    // every stem pixel of the halo inside the image (or wrapped): no masks (workgroup-uniform)
    const bool all_inside = oy0 >= 1 && oy0 + SD_TH + 1 <= Ho && (circular || (ox0 >= 1 && ox0 + SD_TW + 1 <= Wo));
    // ---- 2. stem on the matrix cores, two pixel tiles (four accumulator chains) at a time -------------------------------------
#pragma unroll
    for (int i = 0; i < SD_TPW; i += 2) {
      const bool second = wave + 4 * (i + 1) < SD_NT;  // wave-uniform: waves 2, 3 have 5 tiles
      // (opaque copies: otherwise the 6 x 7 patch read addresses rbase + koff, loop-invariant over the tiles, are hoisted out
      // of the persistent loop into 42 registers and the prefetched patch is spilled)
      int rb0 = rbase[i], rb1 = rbase[i + 1], wo0 = wofs[i], wo1 = wofs[i + 1];
      asm volatile("" : "+v"(rb0), "+v"(rb1), "+v"(wo0), "+v"(wo1));
      float xv[2][7];
#pragma unroll
      for (int s = 0; s < 7; ++s) {
        xv[0][s] = img[rb0 + koff[s]];
        xv[1][s] = img[rb1 + koff[s]];
      }
      sd_f32x4 acc[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u) acc[u][0] = acc[u][1] = (sd_f32x4){0.f, 0.f, 0.f, 0.f};
      if (!(abl & 2)) {
#pragma unroll
        for (int s = 0; s < 7; ++s) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[s][0], xv[0][s], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[s][1], xv[0][s], acc[0][1], 0, 0, 0);
          if (second) {
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[s][0], xv[1][s], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[s][1], xv[1][s], acc[1][1], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int s = 0; s < 7; ++s) { acc[0][0][s & 3] += xv[0][s]; acc[1][1][s & 3] += xv[1][s]; }
      }
      // D: row = channel 4 q + r (+16 for the second accumulator), column = pixel `pix`
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u == 1 && !second) break;
        sd_f32x4 v0 = acc[u][0] * sc[0] + sh[0], v1 = acc[u][1] * sc[1] + sh[1];
        if (!(abl & 4)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { v0[r] = swishf(v0[r]); v1[r] = swishf(v1[r]); }
        }
        if (!all_inside) {
          const int pc = min(16 * (wave + 4 * (i + u)) + pix, SD_NPX - 1);
          const int hy = pc / SD_SW, hx = pc - hy * SD_SW;
          const int sy = oy0 - 1 + hy, sx = ox0 - 1 + hx;
          const bool inside = (unsigned)sy < (unsigned)Ho && (circular || (unsigned)sx < (unsigned)Wo);
#pragma unroll
          for (int r = 0; r < 4; ++r) { v0[r] = inside ? v0[r] : 0.f; v1[r] = inside ? v1[r] : 0.f; }
        }
        const int wo = u ? wo1 : wo0;
        if (wo >= 0) {
          *reinterpret_cast<sd_f32x4*>(tile + wo) = v0;
          *reinterpret_cast<sd_f32x4*>(tile + wo + 16) = v1;
        }
      }
    }
    __syncthreads();
    // the patch buffer is free (every wave is past its stem phase): the next tile's patch.  (Requesting it before the stem
    // phase and parking it in 21 registers until here measured the same 310-330 us per aerial batch and spilled: the kernel is
    // issue-bound, not latency-bound — see the note at the top; the other workgroup of the CU covers the round trip.)
    if (t_next < t_end && !(abl & 1)) {
      request_patch(t_next);
      park_patch();
    }
    // ---- 3. depthwise 3x3: 8 adjacent outputs of row `orow`, columns 8 wave .. 8 wave + 7, 4 channels -------------------------
    sd_f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = (sd_f32x4){0.f, 0.f, 0.f, 0.f};
    if (!(abl & 8)) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        sd_f32x4 col[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) col[j] = *reinterpret_cast<const sd_f32x4*>(trow + (ky * SD_RP + j) * SD_PP);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const sd_f32x4 wv = wreg[ky * 3 + kx];
#pragma unroll
          for (int t = 0; t < 8; ++t) acc[t] += col[t + kx] * wv;
        }
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
      if (!(abl & 16)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = swishf(o[r]);
      }
      if (row_ok && ox < Wo) {
        if (!(abl & 32)) st4<T>(yrow + (size_t)ox * 32, o);
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
    __syncthreads();                                   // the stem tile is free, the next patch and `red` are published
    if (tid < 8) {
      sd_f32x4 t = *reinterpret_cast<const sd_f32x4*>(red + tid * 4);
#pragma unroll
      for (int wv = 1; wv < 4; ++wv) t += *reinterpret_cast<const sd_f32x4*>(red + (wv * 8 + tid) * 4);
      *reinterpret_cast<sd_f32x4*>(se_partial + ((size_t)b * (tiles_x * tiles_y) + ty * tiles_x + tx) * 32 + tid * 4) = t;
    }
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
  static bool attr_set = false;                       // per instantiation: a driver call, not one per launch
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)stem_dw_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return fail(CCVPE_ELAUNCH, "stem_dw: set smem attr: %s", hipGetErrorString(e));
    attr_set = true;
  }
  if ((double)3 * H * W * 4 >= 2147483648.0) return fail(CCVPE_EINVAL, "stem_dw: image of one sample larger than 2 GB");
  // persistent: two workgroups per CU (LDS), a multiple of the 8 XCDs, never more than the tiles of the smallest XCD share
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return fail(CCVPE_ELAUNCH, "stem_dw: device query failed");
    cus = pr.multiProcessorCount;
  }
  long grid = 2L * cus;
  grid -= grid % 8;
  if (grid < 8) grid = 8;
  if (grid > ((total + 7) / 8) * 8) grid = ((total + 7) / 8) * 8;
  hipLaunchKernelGGL((stem_dw_kernel<T>), dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, x, w, s0, b0, wdw, s1, b1, y,
                     se_partial, H, W, Ho, Wo, circular, tiles_x, tiles_y, (int)total, sd_ablate_env());
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
