// Fused MBConv front half: expand 1x1 + BN0 + swish -> depthwise KxK (stride S) + BN1 + swish, plus
// the SE squeeze partial sums.  efficientnet_pytorch/model.py:102-110,114.
//
// Why: the 6x-expanded tensor is the largest in the network (25 MB fp32 per aerial sample at block 1)
// and the unfused path writes it and reads it back (about half of all encoder HBM bytes).  Here it
// lives only in LDS.
//
// Round 3 mapping (the round-2 kernel owned 16 channels x a band of FULL-WIDTH rows and alternated a one-row expand phase
// with a one-row depthwise phase: two barriers per output row around a few hundred cycles of work each, and x re-fetched
// mid/16 times by different workgroups — 0.94 TB/s, 22 % MFMA-busy):
//   * a workgroup owns (sample, SPATIAL tile of TOH x 16 output pixels) and loops over ALL mid/16 channel chunks, so the
//     x halo tile ((TOH-1)*S+K) x (15*S+K) pixels is fetched by ONE workgroup (first chunk from HBM, the rest from L1/L2);
//   * per chunk ONE expand phase for the whole halo tile: its pixels are numbered linearly and cut into 16-pixel MFMA tiles
//     (no idle waves whatever the row width), x fragments are loaded straight from global memory in operand layout
//     (lane = pixel l&15, k group l>>4 -> 16 bytes; groups of tiles in flight together), v_mfma accumulates, BN0 + swish,
//     one ds_write_b128 per lane into the LDS tile [pixel][16 ch], pixel pitch 20 floats (conflict-free for the writes and
//     for the depthwise reads below).  Pixels outside the image are ZEROS (the depthwise conv pads the EXPANDED tensor);
//     circular padding is an index wrap of the source column;
//   * ONE depthwise phase per chunk: thread = (4 channels, 4/S adjacent output columns, row) slides the window through
//     registers — K + (4/S - 1)*S reads per kernel row instead of K per output (LDS bandwidth is the roof of this phase for
//     k = 5) — BN1 + swish, 16-byte stores, per-channel sums for the squeeze reduced by a fixed butterfly;
//   * two barriers per (tile, chunk) around ~10x the work of the old per-row phases; 3-4 workgroups per CU overlap each
//     other's phases.
// Halo recompute: ((TOH-1)*S+K)*(15*S+K) / (TOH*S*16*S) = 1.10 (k3 s2) ... 1.56 (k5 s1) of the expand work, which is the
// cheap part (MFMA).
#include "common.h"
#include "mbconv_plane.h"
#include <type_traits>

namespace ccvpe {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int mbf_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 keepv(f32x4 v, bool keep) {       // v or all-zero bits, without a branch
  return __builtin_bit_cast(f32x4, __builtin_bit_cast(mbf_i32x4, v) & (keep ? -1 : 0));
}

struct MbFrontParams {
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
  int tiles_x, tiles_y, nchunks, total_blocks;
  int ngroups, cpg;      // the chunk loop of a tile is cut into ngroups workgroups of cpg chunks when there are few tiles
};

constexpr int MBF_MAX_KK = 3;   // Cin <= 48 (fp32) / 96 (bf16)

// TOH = output rows of the tile (16 columns wide).  The x fragments of the whole halo tile stay in REGISTERS across the
// channel-chunk loop (TPW tiles x NKK pieces of 16 bytes per lane), so TOH is chosen per (k, stride, Cin) to keep that at
// <= ~14 registers-of-4 (mbf_toh below; the choice must not depend on the storage type: it fixes the squeeze-partial rows).
template <int K, int S, int TOH>
struct MbfGeom {
  static constexpr int NOUT = TOH / 4;               // adjacent output columns per depthwise thread (256 = 4 cg x 16/NOUT x TOH)
  static constexpr int IH = (TOH - 1) * S + K;
  static constexpr int IW = 15 * S + K;
  static constexpr int NPX = IH * IW;
  static constexpr int NT = (NPX + 15) / 16;         // 16-pixel MFMA tiles of the halo
  static constexpr int TPW = (NT + 3) / 4;           // tiles per wave
  // LDS pixel pitch (floats): neighbouring column groups of the depthwise phase are NOUT*S pixels apart; their 16-byte reads
  // are conflict-free when that distance is 16 or 48 floats modulo 64
  static constexpr int PITCH = (NOUT * S == 2) ? 24 : 20;
  static constexpr int LDS_FLOATS = NT * 16 * PITCH + 2 * K * K * 16 + 64;
};

// TE = storage type of x / w_exp / y (float or bf16).  NKK counts 64-byte K pieces: 16 fp32 or 32 bf16
// input channels each; the LDS tile (expanded tensor) and all depthwise math stay fp32 either way.
template <typename TE, int K, int S, int NKK, int TOH>
__global__ __launch_bounds__(256, 3) void mbconv_front_kernel(const MbFrontParams p) {
  using G = MbfGeom<K, S, TOH>;
  constexpr int E = 16 / sizeof(TE);
  constexpr int SK = 4 * E;
  constexpr int PB = (S == 1) ? (K - 1) / 2 : (K - 2) / 2;   // pad before (224-schedule SAME)
  constexpr int IW = G::IW, NPX = G::NPX, NT = G::NT, TPW = G::TPW, PITCH = G::PITCH;
  constexpr int KYU = 1;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* tile = sm;                                  // [NT*16][PITCH]
  float* wdw = sm + NT * 16 * PITCH;                 // [2][K*K][16]
  float* red = wdw + 2 * K * K * 16;                 // [4 waves][4 cg][4]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  int lb;
  {
    const int q = p.total_blocks / 8, r = p.total_blocks % 8;
    const int xcd = blockIdx.x % 8, loc = blockIdx.x / 8;
    lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int grp = lb % p.ngroups;                    // chunk groups of one tile are neighbours: they re-read the same x halo
  lb /= p.ngroups;
  const int tx = lb % p.tiles_x;                     // neighbouring tiles (shared halo) on one XCD
  const int ty = (lb / p.tiles_x) % p.tiles_y;
  const int b = lb / (p.tiles_x * p.tiles_y);
  const int oy0 = ty * TOH, ox0 = tx * 16;
  const int iy0 = oy0 * S - PB, ix0 = ox0 * S - PB;
  const int kq = (lane >> 4) * E;                    // first input channel of this lane's k group inside a 64-byte piece
  const int q4 = (lane >> 4) * 4;                    // D rows (channels) of this lane: q4 .. q4+3

  // ---- the x halo tile, ONCE: this wave's 16-pixel tiles t = wave + 4*i, lane -> pixel q = 16*t + (lane & 15), straight
  // from global memory in MFMA operand layout (16 bytes of k group lane>>4); all TPW*NKK loads in flight together.  Pixels
  // outside the image read a clamped address; their EXPANDED value is forced to zero below (the depthwise conv pads the
  // expanded tensor); k groups beyond Cin re-read the pixel's first channels and meet the zero padding of the packed weights.
  const TE* xb = reinterpret_cast<const TE*>(p.x) + (size_t)b * p.H * p.W * p.Cin;
  f32x4 xr[TPW][NKK];
  unsigned pvalid = 0;
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int q = 16 * (wave + 4 * i) + (lane & 15);
    const int hy = q / IW, hx = q - hy * IW;
    const int gy = iy0 + hy;
    int gx = ix0 + hx;
    if (p.circular) gx = gx < 0 ? gx + p.W : (gx >= p.W ? gx - p.W : gx);
    const bool ok = q < NPX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    pvalid |= ok ? (1u << i) : 0u;
    const unsigned off = ok ? (unsigned)((gy * p.W + gx) * p.Cin) * (unsigned)sizeof(TE) : 0u;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      const int ch = kk * SK + kq;
      xr[i][kk] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(xb) + off + (unsigned)(ch < p.Cin ? ch : 0) * (unsigned)sizeof(TE));
    }
  }

  // ---- depthwise-phase coordinates -----------------------------------------------------------------------------------------
  constexpr int NOUT = G::NOUT;
  constexpr int NCOLS = (NOUT - 1) * S + K;          // input columns read per kernel row
  constexpr int XG = 16 / NOUT;                      // column groups per tile row
  const int cg = tid & 3;
  const int oxg = (tid >> 2) % XG;
  const int oyl = (tid >> 2) / XG;                   // 0 .. TOH-1
  const int oy = oy0 + oyl;
  const int oxl = oxg * NOUT;
  const bool row_ok = oy < p.Ho;
  TE* yrow = reinterpret_cast<TE*>(p.y) + ((size_t)(b * p.Ho + (row_ok ? oy : 0)) * p.Wo) * p.mid;
  const float* trow = tile + ((oyl * S) * IW + oxl * S) * PITCH + cg * 4;

  // ---- per-chunk parameters: loaded one chunk AHEAD (under the previous chunk's depthwise phase) ------------------------------
  f32x4 wf[NKK], sc0, sh0, sc1, sh1, wdr;
  auto load_chunk_params = [&](int chunk) {
    const int c0 = chunk * 16;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk)   // expand weights: MFMA A operand, row n = lane&15, k group = lane>>4
      wf[kk] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const TE*>(p.w_exp) + (size_t)(c0 + (lane & 15)) * p.kpad + kk * SK + kq);
    sc0 = *reinterpret_cast<const f32x4*>(p.s0 + c0 + q4);
    sh0 = *reinterpret_cast<const f32x4*>(p.b0 + c0 + q4);
    sc1 = *reinterpret_cast<const f32x4*>(p.s1 + c0 + cg * 4);
    sh1 = *reinterpret_cast<const f32x4*>(p.b1 + c0 + cg * 4);
    const int wi = tid < K * K * 4 ? tid : 0;        // depthwise weights [tap][16]: one 16-byte piece per thread
    wdr = *reinterpret_cast<const f32x4*>(p.w_dw + (size_t)(wi >> 2) * p.mid + c0 + (wi & 3) * 4);
  };
  const int chunk_begin = grp * p.cpg, chunk_end = min(p.nchunks, chunk_begin + p.cpg);
  load_chunk_params(chunk_begin);

  for (int chunk = chunk_begin; chunk < chunk_end; ++chunk) {
    const int c0 = chunk * 16;
    float* wd = wdw + (chunk & 1) * K * K * 16;
    if (tid < K * K * 4) *reinterpret_cast<f32x4*>(wd + tid * 4) = wdr;
    // ---- expand: registers -> MFMA -> BN0 + swish -> LDS tile (no global loads in this phase) ------------------------------
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      if (wave + 4 * i < NT) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
          if (sizeof(TE) == 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kk][r], xr[i][kk][r], acc, 0, 0, 0);
          } else {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cc_bf16x8, wf[kk]), __builtin_bit_cast(cc_bf16x8, xr[i][kk]),
                                                          acc, 0, 0, 0);
          }
        }
        // D: row = channel q4 + reg, col = pixel lane & 15
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = swishf(acc[r] * sc0[r] + sh0[r]);
        // one mask word per tile, made from its bit when it is needed (a vector `keep ? -1 : 0` select was hoisted out of the
        // chunk loop as FOUR mask registers per tile: 36 VGPRs for 9 tiles)
        const bool inside = (pvalid >> i) & 1u;      // a lane mask in scalar registers, four selects
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = inside ? o[r] : 0.f;
        *reinterpret_cast<f32x4*>(tile + (16 * (wave + 4 * i) + (lane & 15)) * PITCH + q4) = o;
      }
      // two tiles at a time (two independent accumulator chains); without the fence the scheduler issues the MFMAs of all
      // TPW tiles first and keeps every accumulator + swish temporary live
      if (i & 1) __builtin_amdgcn_sched_barrier(0);
    }
    const f32x4 sc1c = sc1, sh1c = sh1;              // this chunk's BN1 (the registers are re-loaded below)
    __syncthreads();
    if (chunk + 1 < chunk_end) load_chunk_params(chunk + 1);   // in flight under the depthwise phase

    // ---- depthwise: NOUT adjacent outputs of one row for 4 channels, window slid through registers -------------------------
    f32x4 acc[NOUT];
#pragma unroll
    for (int t = 0; t < NOUT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // one kernel row of window reads in flight at a time (fully unrolled the scheduler hoists all of them: 200+ VGPRs)
#pragma unroll KYU
    for (int ky = 0; ky < K; ++ky) {
      f32x4 col[NCOLS];
#pragma unroll
      for (int j = 0; j < NCOLS; ++j) col[j] = *reinterpret_cast<const f32x4*>(trow + (ky * IW + j) * PITCH);
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wd + (ky * K + kx) * 16 + cg * 4);
#pragma unroll
        for (int t = 0; t < NOUT; ++t) acc[t] += col[t * S + kx] * wv;
      }
    }
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NOUT; ++t) {
      const int ox = ox0 + oxl + t;
      f32x4 o = acc[t] * sc1c + sh1c;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = swishf(o[r]);
      if (row_ok && ox < p.Wo) {
        st4<TE>(yrow + (size_t)ox * p.mid + c0 + cg * 4, o);
        sum += o;
      }
    }
    // squeeze partial of this (tile, chunk): fixed butterfly over the 16 lanes of a wave that share cg, then the 4 waves
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) sum[r] += __shfl_xor(sum[r], o, 64);
    }
    if (lane < 4) *reinterpret_cast<f32x4*>(red + (wave * 4 + lane) * 4) = sum;
    __syncthreads();                                 // the tile is free for the next chunk; red is published
    if (tid < 4) {
      f32x4 t = *reinterpret_cast<const f32x4*>(red + tid * 4);
#pragma unroll
      for (int w = 1; w < 4; ++w) t += *reinterpret_cast<const f32x4*>(red + (w * 4 + tid) * 4);
      *reinterpret_cast<f32x4*>(p.se_partial + ((size_t)b * (p.tiles_x * p.tiles_y) + ty * p.tiles_x + tx) * p.mid + c0 + tid * 4) = t;
    }
    // (red is rewritten only after the NEXT chunk's first barrier)
  }
}

// output rows per tile for (k, stride, Cin): keeps TPW * ceil(Cin / 16) register fragments per lane small (see MbfGeom);
// depends on Cin through the fp32 piece count for BOTH storage types
static int mbf_toh(int k, int stride, int cin) {
  const int n = (cin + 15) / 16;
  if (stride == 2) return n == 1 ? 8 : 4;
  return (n == 1 && k == 3) ? 16 : 8;
}

// (k = 5, stride 2 with more than 32 input channels is not instantiated: 11 tiles x 3 pieces of x do not fit the register
// budget of 3 waves per SIMD; EfficientNet-B0 has no such block.)  `stride` < 0: any.
static bool mbf_supported(int W, int cin, int mid, int k, int sk = 16, int stride = -1) {
  (void)W;
  if (k == 5 && stride == 2 && cin > 32) return false;
  return cin % 8 == 0 && cin <= 16 * MBF_MAX_KK && cin <= sk * MBF_MAX_KK && mid % 16 == 0;
}

}  // namespace ccvpe

using namespace ccvpe;

// number of squeeze-partial rows per sample = spatial tiles per sample (0 => use the unfused kernels)
extern "C" int ccvpe_mbconv_front_nblk(int in_h, int in_w, int cin, int mid, int k, int stride) {
  if (!(k == 3 || k == 5) || !(stride == 1 || stride == 2)) return CCVPE_EINVAL;
  if (!mbf_supported(in_w, cin, mid, k, 16, stride))      // late blocks (small planes, wide inputs): csrc/mbconv_plane.hip
    return (g_mbplane_mode & 1) ? mbplane_nblk(in_h, in_w, cin, mid, k, stride) : 0;
  const int total_pad = (stride == 1) ? (k - 1) : (k - 2);
  const int Ho = (in_h + total_pad - k) / stride + 1, Wo = (in_w + total_pad - k) / stride + 1;
  const int toh = mbf_toh(k, stride, cin);
  return ((Ho + toh - 1) / toh) * ((Wo + 15) / 16);
}

// which kernel ccvpe_mbconv_front_* runs for a shape (reporting: bench.py's launch recorder, tests): 0 = not served (use the
// unfused calls), 1 = mbconv_front_kernel (early blocks), 2 = mbconv_plane_kernel (late blocks, one channel slice per workgroup),
// 3 = mbconv_band_kernel (late blocks, bf16: producer / consumer waves)
extern "C" int ccvpe_mbconv_front_route(int in_h, int in_w, int cin, int mid, int k, int stride, int is_bf16, int batch) {
  if (!(k == 3 || k == 5) || !(stride == 1 || stride == 2)) return CCVPE_EINVAL;
  if (mbf_supported(in_w, cin, mid, k, 16, stride)) return 1;
  if (!(g_mbplane_mode & 1) || mbplane_nblk(in_h, in_w, cin, mid, k, stride) <= 0) return 0;
  return (is_bf16 && (g_mbplane_mode & 4) && mbband_takes(in_h, in_w, cin, mid, k, stride, batch)) ? 3 : 2;
}

template <typename TE>
static int mbconv_front_any(const void* x, const void* w_exp, int kpad, const float* s0, const float* b0,
                            const float* w_dw, const float* s1, const float* b1, void* y, float* se_partial, int B,
                            int H, int W, int cin, int mid, int k, int stride, int circular, void* stream) {
  constexpr int SK = 4 * (16 / (int)sizeof(TE));
  if (!(k == 3 || k == 5) || !(stride == 1 || stride == 2)) return fail(CCVPE_EINVAL, "mbconv_front: k/stride unsupported");
  if (!mbf_supported(W, cin, mid, k, 16, stride) && (g_mbplane_mode & 1) && mbplane_nblk(H, W, cin, mid, k, stride) > 0)
    return mbplane_launch(sizeof(TE) == 2, 1, x, w_exp, kpad, s0, b0, w_dw, s1, b1, y, se_partial, B, H, W, cin, mid, k, stride,
                          circular, stream);
  if (!mbf_supported(W, cin, mid, k, SK, stride)) return fail(CCVPE_EINVAL, "mbconv_front: shape not supported (W=%d cin=%d mid=%d)", W, cin, mid);
  if (kpad % SK || kpad < cin) return fail(CCVPE_EINVAL, "mbconv_front: bad kpad");
  if (!aligned16(x) || !aligned16(w_exp) || !aligned16(s0) || !aligned16(b0) || !aligned16(s1) || !aligned16(b1) ||
      !aligned16(y) || !aligned16(se_partial) || !aligned16(w_dw))
    return fail(CCVPE_EINVAL, "mbconv_front: pointers must be 16-byte aligned");
  if ((double)H * W * cin * sizeof(TE) >= 4294967296.0) return fail(CCVPE_EINVAL, "mbconv_front: sample larger than 4 GB");
  if (circular && W < k) return fail(CCVPE_EINVAL, "mbconv_front: W too small for circular wrap");
  MbFrontParams p;
  p.x = x; p.w_exp = w_exp; p.s0 = s0; p.b0 = b0; p.w_dw = w_dw; p.s1 = s1; p.b1 = b1; p.y = y; p.se_partial = se_partial;
  p.H = H; p.W = W; p.Cin = cin; p.kpad = kpad; p.mid = mid; p.circular = circular;
  const int total_pad = (stride == 1) ? (k - 1) : (k - 2);
  p.Ho = (H + total_pad - k) / stride + 1;
  p.Wo = (W + total_pad - k) / stride + 1;
  const int toh = mbf_toh(k, stride, cin);
  p.tiles_x = (p.Wo + 15) / 16;
  p.tiles_y = (p.Ho + toh - 1) / toh;
  p.nchunks = mid / 16;
  // few tiles (small planes): cut the chunk loop so that the launch still has ~8 workgroups per CU
  const long tiles = (long)p.tiles_x * p.tiles_y * B;
  int ng = (int)((2048 + tiles - 1) / tiles);
  if (ng > p.nchunks) ng = p.nchunks;
  if (ng < 1) ng = 1;
  p.cpg = (p.nchunks + ng - 1) / ng;
  p.ngroups = (p.nchunks + p.cpg - 1) / p.cpg;
  const long total = tiles * p.ngroups;
  if (total > 0x7fffffffL) return fail(CCVPE_EINVAL, "mbconv_front: grid too large");
  p.total_blocks = (int)total;
  hipStream_t st = (hipStream_t)stream;
  const int nkk = (cin + SK - 1) / SK;
  int rc = CCVPE_OK;
  auto go = [&](auto kern, int lds_floats) {
    const int lds = lds_floats * 4;
    static bool attr_set = false;                    // per instantiation (generic lambda)
    if (lds > 48 * 1024 && !attr_set) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) { rc = fail(CCVPE_ELAUNCH, "mbconv_front: set smem attr: %s", hipGetErrorString(e)); return; }
      attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.total_blocks), dim3(256), lds, st, p);
  };
  // instantiated: the big tile for Cin <= 16 (one K piece in either storage type), the small tile for 1-3 pieces
#define MBF_N(K_, S_, TOH_)                                                                                  \
  if (nkk == 1) go(mbconv_front_kernel<TE, K_, S_, 1, TOH_>, MbfGeom<K_, S_, TOH_>::LDS_FLOATS);                \
  else if (nkk == 2) go(mbconv_front_kernel<TE, K_, S_, 2, TOH_>, MbfGeom<K_, S_, TOH_>::LDS_FLOATS);           \
  else go(mbconv_front_kernel<TE, K_, S_, 3, TOH_>, MbfGeom<K_, S_, TOH_>::LDS_FLOATS)
#define MBF_T(K_, S_, BIG_, SMALL_)                                                                          \
  if (toh == BIG_) { go(mbconv_front_kernel<TE, K_, S_, 1, BIG_>, MbfGeom<K_, S_, BIG_>::LDS_FLOATS); } else { MBF_N(K_, S_, SMALL_); }
  if (k == 3 && stride == 1) { MBF_T(3, 1, 16, 8) }
  else if (k == 3 && stride == 2) { MBF_T(3, 2, 8, 4) }
  else if (k == 5 && stride == 1) { MBF_N(5, 1, 8); }
  else if (toh == 8) { go(mbconv_front_kernel<TE, 5, 2, 1, 8>, MbfGeom<5, 2, 8>::LDS_FLOATS); }
  else if (nkk == 1) { go(mbconv_front_kernel<TE, 5, 2, 1, 4>, MbfGeom<5, 2, 4>::LDS_FLOATS); }
  else { go(mbconv_front_kernel<TE, 5, 2, 2, 4>, MbfGeom<5, 2, 4>::LDS_FLOATS); }
#undef MBF_T
#undef MBF_N
  if (rc) return rc;
  return check_launch("mbconv_front_kernel");
}

extern "C" int ccvpe_mbconv_front_f32(const float* x, const float* w_exp, int kpad, const float* s0, const float* b0,
                                      const float* w_dw, const float* s1, const float* b1, float* y,
                                      float* se_partial, int B, int H, int W, int cin, int mid, int k, int stride,
                                      int circular, void* stream) {
  return mbconv_front_any<float>(x, w_exp, kpad, s0, b0, w_dw, s1, b1, y, se_partial, B, H, W, cin, mid, k, stride,
                                 circular, stream);
}
extern "C" int ccvpe_mbconv_front_bf16(const void* x, const void* w_exp, int kpad, const float* s0, const float* b0,
                                       const float* w_dw, const float* s1, const float* b1, void* y,
                                       float* se_partial, int B, int H, int W, int cin, int mid, int k, int stride,
                                       int circular, void* stream) {
  return mbconv_front_any<cc_bf16>(x, w_exp, kpad, s0, b0, w_dw, s1, b1, y, se_partial, B, H, W, cin, mid, k, stride,
                                   circular, stream);
}
