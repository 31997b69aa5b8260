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
};

constexpr int MBF_MAX_KK = 3;   // Cin <= 48 (fp32) / 96 (bf16)
constexpr int MBF_PITCH = 20;   // floats per LDS pixel (16 channels + 4 pad)

template <int K, int S>
struct MbfGeom {
  static constexpr int TOW = 16;
  static constexpr int TOH = S == 1 ? 16 : 8;
  static constexpr int IH = (TOH - 1) * S + K;
  static constexpr int IW = (TOW - 1) * S + K;
  static constexpr int NPX = IH * IW;
  static constexpr int NT = (NPX + 15) / 16;         // 16-pixel MFMA tiles of the halo
  static constexpr int TPW = (NT + 3) / 4;           // tiles per wave
  static constexpr int LDS_FLOATS = NT * 16 * MBF_PITCH + 2 * K * K * 16 + 64;
};

// TE = storage type of x / w_exp / y (float or bf16).  NKK counts 64-byte K pieces: 16 fp32 or 32 bf16
// input channels each; the LDS tile (expanded tensor) and all depthwise math stay fp32 either way.
template <typename TE, int K, int S, int NKK>
__global__ __launch_bounds__(256, 3) void mbconv_front_kernel(const MbFrontParams p) {
  using G = MbfGeom<K, S>;
  constexpr int E = 16 / sizeof(TE);
  constexpr int SK = 4 * E;
  constexpr int PB = (S == 1) ? (K - 1) / 2 : (K - 2) / 2;   // pad before (224-schedule SAME)
  constexpr int TOH = G::TOH, IW = G::IW, NPX = G::NPX, NT = G::NT, TPW = G::TPW;
  constexpr int GT = NKK == 1 ? 4 : 2;      // tiles whose loads are in flight together
  constexpr int KYU = 1;
  constexpr int NGR = (TPW + GT - 1) / GT;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* tile = sm;                                  // [NT*16][MBF_PITCH]
  float* wdw = sm + NT * 16 * MBF_PITCH;             // [2][K*K][16]
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
  const int tx = lb % p.tiles_x;                     // neighbouring tiles (shared halo) on one XCD
  const int ty = (lb / p.tiles_x) % p.tiles_y;
  const int b = lb / (p.tiles_x * p.tiles_y);
  const int oy0 = ty * TOH, ox0 = tx * 16;
  const int iy0 = oy0 * S - PB, ix0 = ox0 * S - PB;

  // ---- expand-phase coordinates: this wave's tiles t = wave + 4*i; lane -> pixel q = 16*t + (lane & 15) of the halo.
  // The address of a tile's pixel is recomputed where it is needed (a dozen integer instructions per 16-pixel tile) instead
  // of being kept in a register array: the first version held offsets + two fully unrolled register sets and needed 200-256
  // VGPRs (1-2 waves per SIMD).
  const TE* xb = reinterpret_cast<const TE*>(p.x) + (size_t)b * p.H * p.W * p.Cin;
  auto pixel_offset = [&](int i, bool& ok) -> unsigned {
    const int q = 16 * (wave + 4 * i) + (lane & 15);
    const int hy = q / IW, hx = q - hy * IW;
    const int gy = iy0 + hy;
    int gx = ix0 + hx;
    if (p.circular) gx = gx < 0 ? gx + p.W : (gx >= p.W ? gx - p.W : gx);
    ok = q < NPX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    return ok ? (unsigned)((gy * p.W + gx) * p.Cin) * (unsigned)sizeof(TE) : 0u;
  };
  const int kq = (lane >> 4) * E;                    // first input channel of this lane's k group inside a 64-byte piece
  const int q4 = (lane >> 4) * 4;                    // D rows (channels) of this lane: q4 .. q4+3

  // ---- depthwise-phase coordinates -----------------------------------------------------------------------------------------
  constexpr int NOUT = 4 / S;                        // adjacent output columns per thread: 4 (s1) / 2 (s2)
  constexpr int NCOLS = (NOUT - 1) * S + K;          // input columns read per kernel row
  constexpr int XG = 16 / NOUT;                      // column groups per tile row
  const int cg = tid & 3;
  const int oxg = (tid >> 2) % XG;
  const int oyl = (tid >> 2) / XG;                   // 0 .. TOH-1   (256 threads = 4 cg x XG x TOH)
  const int oy = oy0 + oyl;
  const int oxl = oxg * NOUT;
  const bool row_ok = oy < p.Ho;
  TE* yrow = reinterpret_cast<TE*>(p.y) + ((size_t)(b * p.Ho + (row_ok ? oy : 0)) * p.Wo) * p.mid;
  const float* trow = tile + ((oyl * S) * IW + oxl * S) * MBF_PITCH + cg * 4;

  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
    const int c0 = chunk * 16;
    // depthwise weights of this chunk -> LDS (double-buffered across chunks: no extra barrier)
    float* wd = wdw + (chunk & 1) * K * K * 16;
    if (tid < K * K * 4) *reinterpret_cast<f32x4*>(wd + tid * 4) =
        *reinterpret_cast<const f32x4*>(p.w_dw + (size_t)(tid >> 2) * p.mid + c0 + (tid & 3) * 4);
    // expand weights (MFMA A operand: row n = lane&15, k group = lane>>4) and BN0 of this chunk
    f32x4 wf[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk)
      wf[kk] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const TE*>(p.w_exp) + (size_t)(c0 + (lane & 15)) * p.kpad + kk * SK + kq);
    const f32x4 sc0 = *reinterpret_cast<const f32x4*>(p.s0 + c0 + q4);
    const f32x4 sh0 = *reinterpret_cast<const f32x4*>(p.b0 + c0 + q4);

    // ---- expand: groups of GT tiles, the loads of group g+1 in flight under the MFMAs / swish of group g ----------------
    f32x4 xv[2][GT][NKK];
    unsigned okm[2] = {0u, 0u};                      // per register set: bit u = the tile's pixel of this lane is inside the image
    auto load_group = [&](int g, auto set_tag) {
      constexpr int SET = decltype(set_tag)::value;
      okm[SET] = 0;
#pragma unroll
      for (int u = 0; u < GT; ++u) {
        const int i = g * GT + u;
        bool ok;
        const unsigned off = pixel_offset(i < TPW ? i : TPW - 1, ok);
        okm[SET] |= (ok && i < TPW) ? (1u << u) : 0u;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
          const int ch = kk * SK + kq;
          xv[SET][u][kk] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(xb) + off +
                                                           (unsigned)(ch < p.Cin ? ch : 0) * (unsigned)sizeof(TE));
        }
      }
    };
    auto compute_group = [&](int g, auto set_tag) {
      constexpr int SET = decltype(set_tag)::value;
#pragma unroll
      for (int u = 0; u < GT; ++u) {
        const int i = g * GT + u;
        if (i < TPW && wave + 4 * i < NT) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < NKK; ++kk) {
            // (k groups beyond Cin inside the last 64-byte piece re-read the pixel's first channels — the address is clamped —
            // and meet the zero padding of the packed weights)
            const f32x4 xq = xv[SET][u][kk];
            if (sizeof(TE) == 4) {
#pragma unroll
              for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kk][r], xq[r], acc, 0, 0, 0);
            } else {
              acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cc_bf16x8, wf[kk]), __builtin_bit_cast(cc_bf16x8, xq),
                                                            acc, 0, 0, 0);
            }
          }
          // D: row = channel q4 + reg, col = pixel lane & 15
          f32x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = swishf(acc[r] * sc0[r] + sh0[r]);
          o = keepv(o, (okm[SET] >> u) & 1u);
          *reinterpret_cast<f32x4*>(tile + (16 * (wave + 4 * i) + (lane & 15)) * MBF_PITCH + q4) = o;
        }
      }
    };
    load_group(0, std::integral_constant<int, 0>{});
#pragma unroll 1
    for (int g = 0; g < NGR; g += 2) {
      if (g + 1 < NGR) load_group(g + 1, std::integral_constant<int, 1>{});
      compute_group(g, std::integral_constant<int, 0>{});
      if (g + 2 < NGR) load_group(g + 2, std::integral_constant<int, 0>{});
      if (g + 1 < NGR) compute_group(g + 1, std::integral_constant<int, 1>{});
    }
    __syncthreads();

    // ---- depthwise: NOUT adjacent outputs of one row for 4 channels, window slid through registers -------------------------
    const f32x4 sc1 = *reinterpret_cast<const f32x4*>(p.s1 + c0 + cg * 4);
    const f32x4 sh1 = *reinterpret_cast<const f32x4*>(p.b1 + c0 + cg * 4);
    f32x4 acc[NOUT];
#pragma unroll
    for (int t = 0; t < NOUT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // k = 5: one kernel row of window reads in flight at a time (fully unrolled the scheduler hoists all 35-40 ds_read_b128)
#pragma unroll KYU
    for (int ky = 0; ky < K; ++ky) {
      f32x4 col[NCOLS];
#pragma unroll
      for (int j = 0; j < NCOLS; ++j) col[j] = *reinterpret_cast<const f32x4*>(trow + (ky * IW + j) * MBF_PITCH);
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
      f32x4 o = acc[t] * sc1 + sh1;
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

static bool mbf_supported(int W, int cin, int mid, int k, int sk = 16) {
  (void)W; (void)k;
  return cin % 8 == 0 && cin <= sk * MBF_MAX_KK && mid % 16 == 0;
}

}  // namespace ccvpe

using namespace ccvpe;

// number of squeeze-partial rows per sample = spatial tiles per sample (0 => use the unfused kernels)
extern "C" int ccvpe_mbconv_front_nblk(int in_h, int in_w, int cin, int mid, int k, int stride) {
  if (!(k == 3 || k == 5) || !(stride == 1 || stride == 2)) return CCVPE_EINVAL;
  if (!mbf_supported(in_w, cin, mid, k)) return 0;
  const int total_pad = (stride == 1) ? (k - 1) : (k - 2);
  const int Ho = (in_h + total_pad - k) / stride + 1, Wo = (in_w + total_pad - k) / stride + 1;
  const int toh = stride == 1 ? 16 : 8;
  return ((Ho + toh - 1) / toh) * ((Wo + 15) / 16);
}

template <typename TE>
static int mbconv_front_any(const void* x, const void* w_exp, int kpad, const float* s0, const float* b0,
                            const float* w_dw, const float* s1, const float* b1, void* y, float* se_partial, int B,
                            int H, int W, int cin, int mid, int k, int stride, int circular, void* stream) {
  constexpr int SK = 4 * (16 / (int)sizeof(TE));
  if (!(k == 3 || k == 5) || !(stride == 1 || stride == 2)) return fail(CCVPE_EINVAL, "mbconv_front: k/stride unsupported");
  if (!mbf_supported(W, cin, mid, k, SK)) return fail(CCVPE_EINVAL, "mbconv_front: shape not supported (W=%d cin=%d mid=%d)", W, cin, mid);
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
  const int toh = stride == 1 ? 16 : 8;
  p.tiles_x = (p.Wo + 15) / 16;
  p.tiles_y = (p.Ho + toh - 1) / toh;
  p.nchunks = mid / 16;
  const long total = (long)p.tiles_x * p.tiles_y * B;
  if (total > 0x7fffffffL) return fail(CCVPE_EINVAL, "mbconv_front: grid too large");
  p.total_blocks = (int)total;
  hipStream_t st = (hipStream_t)stream;
  const int nkk = (cin + SK - 1) / SK;
  int rc = CCVPE_OK;
  auto go = [&](auto kern, int lds_floats) {
    const int lds = lds_floats * 4;
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) { rc = fail(CCVPE_ELAUNCH, "mbconv_front: set smem attr: %s", hipGetErrorString(e)); return; }
    }
    hipLaunchKernelGGL(kern, dim3(p.total_blocks), dim3(256), lds, st, p);
  };
#define MBF_N(K_, S_)                                                                            \
  if (nkk == 1) go(mbconv_front_kernel<TE, K_, S_, 1>, MbfGeom<K_, S_>::LDS_FLOATS);              \
  else if (nkk == 2) go(mbconv_front_kernel<TE, K_, S_, 2>, MbfGeom<K_, S_>::LDS_FLOATS);         \
  else go(mbconv_front_kernel<TE, K_, S_, 3>, MbfGeom<K_, S_>::LDS_FLOATS)
  if (k == 3 && stride == 1) { MBF_N(3, 1); }
  else if (k == 3 && stride == 2) { MBF_N(3, 2); }
  else if (k == 5 && stride == 1) { MBF_N(5, 1); }
  else { MBF_N(5, 2); }
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
