// Implicit-GEMM convolution on the gfx950 fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
//   out[m, n] = act((sum_k A[m, k] * Wp[n, k]) * scale[n] + shift[n]) (+ residual[m, n])
//
// m = output pixel (b, oy, ox); k = (tap, concat channel).  Replaces the F.conv2d /
// F.conv_transpose2d / nn.Linear call sites of the reference:
// efficientnet_pytorch/model.py:104-106,121-130,299 ; models.py:42-47,57-97,102-148,173-184.
//
// Two kernels share the tile machinery:
//   igemm_f32_kernel   — generic gather (1x1, 2x2/s2, deconv-as-GEMM): A and W stages
//                        ([rows][16 k] fp32) go global -> VGPR -> LDS, double buffered.
//   conv3x3_f32_kernel — 3x3 stride 1 pad 1 (double_conv, models.py:42-47): the input HALO tile
//                        ((TH+2) x 18 pixels x 16 channels) is staged in LDS ONCE per channel chunk
//                        and the 9 taps read their A fragments from it at shifted addresses, so the
//                        activation is fetched from L2 once instead of 9 times (the per-tap re-read
//                        made the N<=40 layers L2-bound in the first profile); only the W stage
//                        changes per tap.  Reads cat[deconv_out, skip] as two sources.
//
// Tile: 256 threads = 4 waves laid out WM x WN; each wave owns (16*MT) pixels x (16*NT) channels as
// MT*NT accumulators.  Operand roles are SWAPPED (W fragment is the MFMA "A" operand, the pixel
// fragment the "B" operand) so that the C layout gives each lane 4 CONSECUTIVE OUTPUT CHANNELS of
// one pixel: the epilogue is one dwordx4 store (and dwordx4 residual load) per accumulator instead
// of four scalar ones — the HBM-bound 1x1 layers were store-issue bound before.
// K permutation: lane group q = lane>>4 consumes k in {4q..4q+3} over the 4 MFMAs of a stage, so
// each operand fragment is ONE ds_read_b128; both operands use the same map.
// LDS rows are padded 16 -> 20 floats (80 B: 16-byte aligned, breaks the 64 B power-of-two stride).
//
// The fp32 MFMA runs at the fp32 vector rate (157 TF peak) and is a bitwise k-ordered fmaf chain:
// the fp32 instantiations are exact fp32 (gfx950 has no TF32-like shortcut).
//
// Element type T: both kernels are instantiated for float and for bf16 storage (BASELINE configs
// C2/C4).  The LDS BYTE geometry is identical: a stage row is 64 bytes = 16 fp32 or 32 bf16 channels,
// a fragment is one ds_read_b128 = 4 fp32 (4 x v_mfma_f32_16x16x4_f32) or 8 bf16
// (1 x v_mfma_f32_16x16x32_bf16, fp32 accumulate).  bf16 stores round-to-nearest-even
// (v_cvt_pk_bf16_f32); scale/shift/gate and all accumulation stay fp32.
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace ccvpe {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> { static constexpr int E = 4; };    // elements per 16 bytes
template <> struct ElemTraits<bf16_t> { static constexpr int E = 8; };

// one K-stage (64 bytes of K per row) of MFMAs for one (W fragment, pixel fragment) pair
template <typename T>
__device__ __forceinline__ f32x4 mfma_stage(f32x4 wfrag, f32x4 afrag, f32x4 acc);
template <>
__device__ __forceinline__ f32x4 mfma_stage<float>(f32x4 wfrag, f32x4 afrag, f32x4 acc) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wfrag[kk], afrag[kk], acc, 0, 0, 0);
  return acc;
}
template <>
__device__ __forceinline__ f32x4 mfma_stage<bf16_t>(f32x4 wfrag, f32x4 afrag, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wfrag), __builtin_bit_cast(bf16x8, afrag),
                                                 acc, 0, 0, 0);
}

// SE gate on 16 raw bytes of activations (gate is fp32, per channel)
template <typename T>
__device__ __forceinline__ f32x4 apply_gate(f32x4 raw, const float* g);
template <>
__device__ __forceinline__ f32x4 apply_gate<float>(f32x4 raw, const float* g) {
  return raw * *reinterpret_cast<const f32x4*>(g);
}
template <>
__device__ __forceinline__ f32x4 apply_gate<bf16_t>(f32x4 raw, const float* g) {
  bf16x8 v = __builtin_bit_cast(bf16x8, raw);
  const f32x4 g0 = *reinterpret_cast<const f32x4*>(g), g1 = *reinterpret_cast<const f32x4*>(g + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i] = (bf16_t)((float)v[i] * g0[i]);
    v[i + 4] = (bf16_t)((float)v[i + 4] * g1[i]);
  }
  return __builtin_bit_cast(f32x4, v);
}

// ---------------------------------------------------------------------------------------------
// STAGING RULE for every kernel in this file: a stage's global loads are issued into RAW registers and nothing touches
// those registers until the stage's matrix instructions have been issued; validity masks, the SE gate and any other
// per-element work are applied when the registers are written to LDS (store_*).  Three ways the prefetch used to be
// waited for BEFORE the MFMAs it was meant to overlap (rocprof: 28-79 % MFMA-busy), all visible as `s_waitcnt vmcnt`
// between the loads and the first v_mfma of the loop body (tools/isa_loop.py):
//   * `v = ok ? load : 0` / `v = load * gate` at the load: the select / multiply needs the data;
//   * `f32x4 v = 0; if (ok) v = load;`: the zero-initialisation rewrites a register an earlier load may still own, so the
//     compiler waits vmcnt(0) — for the loads just issued as well;
//   * struct fields used only under a lane-dependent condition (`from0 ? p.ld0 : p.ld1`) were fetched from the kernarg
//     segment with a VECTOR load per use, a dependent load in front of every activation load.
// Out-of-range lanes read a clamped, valid address instead.
// ---------------------------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 keep_if(f32x4 v, bool keep) {       // v or all-zero bits, without a branch
  return __builtin_bit_cast(f32x4, __builtin_bit_cast(i32x4, v) & (keep ? -1 : 0));
}
// SE gate values of one 16-byte activation piece (4 fp32 / 8 bf16 channels), fetched with the stage's loads into two plain
// registers and applied at the LDS store
template <typename T>
__device__ __forceinline__ void gate_load(const float* g, f32x4& g0, f32x4& g1) {
  g0 = *reinterpret_cast<const f32x4*>(g);
  if (sizeof(T) == 2) g1 = *reinterpret_cast<const f32x4*>(g + 4);
}
template <typename T>
__device__ __forceinline__ f32x4 gate_apply(f32x4 raw, f32x4 g0, f32x4 g1) {
  if (sizeof(T) == 4) return raw * g0;
  bf16x8 v = __builtin_bit_cast(bf16x8, raw);
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[i] = (bf16_t)((float)v[i] * g0[i]);
    o[i + 4] = (bf16_t)((float)v[i + 4] * g1[i]);
  }
  return __builtin_bit_cast(f32x4, o);
}
// a kernel-argument field pinned in a scalar register (see the third bullet above)
__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

struct IgemmParams {
  const void* src0;
  const void* src1;
  const float* gate;
  const void* w;
  const float* scale;
  const float* shift;
  const void* residual;
  void* dst;
  int out_f32;       // bf16 kernels only: write fp32 instead of bf16 (final tensors handed to fp32 consumers)
  int c0, ld0, c1, ld1;
  int H, W, Ho, Wo;
  int kw, stride, pad;
  int N, Kpad, Npad;
  int cpt0, cpt, total_chunks, stages;
  int ldd, ldres, act, out_mode, cout;
  int M;
  long in_pixels;    // batch * H * W of the sources (igemm_kernel: 32-bit offsets when they are below 4 GB)
  int tiles_n, tiles_total;
  int tiles_x, tiles_y;  // conv3x3: spatial tiles per sample
  // split-K (igemm_kernel only; small-batch GEMMs with a handful of tiles and thousands of K stages): workgroup
  // (tile, blockIdx.y) walks stages [y*sps, (y+1)*sps) and writes its raw fp32 accumulators to
  // partial[y][M][Npad]; splitk_finish_kernel adds the slices in order and applies the epilogue.
  int ksplit, sps;
  float* partial;
  int ablate;        // diagnostics only (CCVPE_PW_ABLATE): 1 = no global stores, 2 = no global loads, 4 = no MFMAs
};

constexpr int LDS_LD = 20;  // floats per staged row (16 + 4 pad)

// XCD-aware tile order: consecutive workgroup ids round-robin over the 8 XCDs, so give each XCD a
// contiguous run of tiles (n fastest): the N-tiles that re-read one A panel, and spatially
// adjacent tiles that share a halo, hit the same L2.
__device__ __forceinline__ int xcd_tile(int bid, int total) {
  const int q = total / 8, r = total % 8;
  const int xcd = bid % 8, loc = bid / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// Epilogue for one accumulator: 4 consecutive channels n..n+3 of one pixel.
// ACT is a TEMPLATE parameter: the callers switch on p.act ONCE around their whole epilogue (CCVPE_ACT_DISPATCH).  With a
// runtime `if (p.act == ...)` here the compiler emitted a scalar compare + branch per VALUE (113-163 per kernel): ~8 000
// cycles per tile — nothing next to a K = 12 096 3x3 tile, a third of a bf16 tile's life.
template <typename T, int ACT>
__device__ __forceinline__ void store4(const IgemmParams& p, f32x4 v, int n, size_t obase, size_t rbase,
                                       const float* sc, const float* sh) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float t = v[r] * sc[r] + sh[r];
    if (ACT == CCVPE_ACT_RELU) t = fmaxf(t, 0.0f);
    else if (ACT == CCVPE_ACT_SWISH) t = swishf(t);
    v[r] = t;
  }
  const T* res = reinterpret_cast<const T*>(p.residual);
  const bool f32out = sizeof(T) == 4 || p.out_f32;
  if (n + 3 < p.N) {
    if (res) {
      f32x4 rf;
      if (sizeof(T) == 4) {
        rf = *reinterpret_cast<const f32x4*>(res + rbase + n);
      } else {
        const bf16x4 rv = *reinterpret_cast<const bf16x4*>(res + rbase + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) rf[r] = (float)rv[r];
      }
      if (ACT == CCVPE_ACT_RELU_MASK) {   // residual = a ReLU's output: pass the gradient where it was positive
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = rf[r] > 0.f ? v[r] : 0.f;
      } else {
        v += rf;
      }
    }
    if (f32out) {
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.dst) + obase) = v;
    } else {
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
      *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.dst) + obase) = o;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n + r < p.N) {
        float t = v[r];
        if (res) {
          const float rf = (float)res[rbase + n + r];
          t = ACT == CCVPE_ACT_RELU_MASK ? (rf > 0.f ? t : 0.f) : t + rf;
        }
        if (f32out) reinterpret_cast<float*>(p.dst)[obase + r] = t;
        else reinterpret_cast<bf16_t*>(p.dst)[obase + r] = (bf16_t)t;
      }
  }
}

// run `body(std::integral_constant<int, ACT>)` for the (workgroup-uniform) activation code `act`
#define CCVPE_ACT_DISPATCH(act, body)                                            \
  do {                                                                           \
    if ((act) == CCVPE_ACT_SWISH) body(std::integral_constant<int, CCVPE_ACT_SWISH>{});      \
    else if ((act) == CCVPE_ACT_RELU) body(std::integral_constant<int, CCVPE_ACT_RELU>{});   \
    else if ((act) == CCVPE_ACT_RELU_MASK) body(std::integral_constant<int, CCVPE_ACT_RELU_MASK>{}); \
    else body(std::integral_constant<int, CCVPE_ACT_NONE>{});                    \
  } while (0)

template <typename T, int MT, int NT, int WN>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams p) {
  constexpr int E = ElemTraits<T>::E;      // elements per 16-byte load
  constexpr int SK = 4 * E;                // K elements per stage row (64 bytes)
  constexpr int CPS = SK / 8;              // 8-channel chunks per stage: 2 (fp32) or 4 (bf16)
  constexpr int WM = 4 / WN;
  constexpr int BM = 16 * MT * WM;
  constexpr int BN = 16 * NT * WN;
  constexpr int A_IT = BM / 64;            // float4 loads per thread per stage (A)
  constexpr int B_IT = (BN + 63) / 64;     // float4 loads per thread per stage (W)

  __shared__ __attribute__((aligned(16))) float As[2][BM][LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN][LDS_LD];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave % WN;

  const int tile = xcd_tile(blockIdx.x, p.tiles_total);
  const int tm = tile / p.tiles_n;
  const int tn = tile % p.tiles_n;
  const int m0 = tm * BM;
  const int n0 = tn * BN;

  // ---- per-thread staging coordinates -------------------------------------------------------
  const int srow = tid >> 2;   // 0..63
  const int ssub = tid & 3;    // which 16-byte piece of the 64-byte stage row
  const int chunk_in_stage = (ssub * E) >> 3;      // fp32: ssub>>1, bf16: ssub
  const int half = (ssub * E) & 7;                 // channel offset inside the 8-chunk: fp32 0|4, bf16 0
  const T* src0 = reinterpret_cast<const T*>(p.src0);
  const T* src1 = reinterpret_cast<const T*>(p.src1);
  const T* wp = reinterpret_cast<const T*>(p.w);

  int a_b[A_IT], a_y[A_IT], a_x[A_IT];
  bool a_ok[A_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int m = m0 + srow + 64 * it;
    a_ok[it] = m < p.M;
    const int mm = a_ok[it] ? m : 0;
    const int hw = p.Ho * p.Wo;
    const int b = mm / hw;
    const int rem = mm - b * hw;
    const int oy = rem / p.Wo;
    const int ox = rem - oy * p.Wo;
    a_b[it] = b;
    a_y[it] = oy * p.stride - p.pad;
    a_x[it] = ox * p.stride - p.pad;
  }
  // chunk cursor of this thread: kc = CPS*stage + chunk_in_stage, decoded as (ky, kx, r)
  const int s_begin = p.ksplit > 1 ? blockIdx.y * p.sps : 0;
  const int s_end = p.ksplit > 1 ? min(p.stages, s_begin + p.sps) : p.stages;
  int kc = CPS * s_begin + chunk_in_stage;
  int r, ky, kx;
  {
    const int tap = kc / p.cpt;
    r = kc - tap * p.cpt;
    ky = tap / p.kw;
    kx = tap - ky * p.kw;
  }

  f32x4 a_reg[A_IT], b_reg[B_IT];
  // byte offset of this lane's 16-byte piece inside W, per staged row (rows past the tile / Npad: any valid row, never
  // stored): stage-invariant, so a W request is (scalar base + stage offset) + this register — no vector ALU work per stage
  unsigned wrow[B_IT];
#pragma unroll
  for (int it = 0; it < B_IT; ++it)
    wrow[it] = ((unsigned)min(n0 + srow + 64 * it, p.Npad - 1) * (unsigned)p.Kpad + (unsigned)(ssub * E)) * (unsigned)sizeof(T);
  unsigned a_keep = 0;                      // bit `it`: piece `it` of the stage in a_reg is inside the image and the K range
  f32x4 g_r0 = {0.f, 0.f, 0.f, 0.f}, g_r1 = g_r0;   // SE gate of the FIRST staged row's sample (rows of a tile nearly always share it)
  int g_ch = 0;                             // channel of the staged piece, for the rare rows of another sample
  const bool gated = p.gate != nullptr;     // (1x1, single-source convs only)
  const int ld0s = sgpr(p.ld0), ld1s = sgpr(p.ld1);
  // both sources below 4 GB (workgroup-uniform): 32-bit byte offsets, one v_mad per piece instead of 64-bit multiply-adds
  const bool small32 = (double)p.in_pixels * (double)(ld0s > ld1s ? ld0s : ld1s) * sizeof(T) < 4294967296.0;

  auto load_stage = [&](int s) {
    const bool kvalid = kc < p.total_chunks;
    const bool from0 = !kvalid || r < p.cpt0;
    const T* base = from0 ? src0 : src1;
    const int ld = from0 ? ld0s : ld1s;
    const int ch = kvalid ? (from0 ? r : r - p.cpt0) * 8 + half : 0;
    a_keep = 0;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int iy = a_y[it] + ky;
      const int ix = a_x[it] + kx;
      const bool ok = a_ok[it] && kvalid && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const int pix = ok ? (a_b[it] * p.H + iy) * p.W + ix : 0;
      if (small32) a_reg[it] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + ((unsigned)pix * (unsigned)ld + (unsigned)ch) * (unsigned)sizeof(T));
      else a_reg[it] = *reinterpret_cast<const f32x4*>(base + (size_t)pix * ld + ch);
      a_keep |= ok ? (1u << it) : 0u;
    }
    if (gated) {
      g_ch = ch;
      gate_load<T>(p.gate + (size_t)a_b[0] * p.c0 + ch, g_r0, g_r1);
    }
    {
      const char* wb = reinterpret_cast<const char*>(wp) + (size_t)s * (SK * sizeof(T));   // scalar base + per-lane row offset
#pragma unroll
      for (int it = 0; it < B_IT; ++it) b_reg[it] = *reinterpret_cast<const f32x4*>(wb + wrow[it]);
    }
    kc += CPS;
    r += CPS;
    while (r >= p.cpt) {
      r -= p.cpt;
      if (++kx == p.kw) { kx = 0; ++ky; }
    }
  };

  auto store_stage = [&](int buf) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      f32x4 v = a_reg[it];
      if (gated) {
        if (a_b[it] == a_b[0]) v = gate_apply<T>(v, g_r0, g_r1);
        else v = apply_gate<T>(v, p.gate + (size_t)a_b[it] * p.c0 + g_ch);
      }
      *reinterpret_cast<f32x4*>(&As[buf][srow + 64 * it][ssub * 4]) = keep_if(v, (a_keep >> it) & 1u);
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int nrow = srow + 64 * it;
      if (nrow < BN) *reinterpret_cast<f32x4*>(&Bs[buf][nrow][ssub * 4]) = b_reg[it];
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;
  const int fk = (lane >> 4) * 4;

  load_stage(s_begin);
  store_stage(0);
  __syncthreads();

  for (int s = s_begin; s < s_end; ++s) {
    const int buf = (s - s_begin) & 1;
    const bool more = s + 1 < s_end;
    if (more) load_stage(s + 1);

    f32x4 af[MT], bf[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
      af[i] = *reinterpret_cast<const f32x4*>(&As[buf][(wm * MT + i) * 16 + frow][fk]);
#pragma unroll
    for (int j = 0; j < NT; ++j)
      bf[j] = *reinterpret_cast<const f32x4*>(&Bs[buf][(wn * NT + j) * 16 + frow][fk]);
    if (sizeof(T) == 4) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][kk], af[i][kk], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma_stage<T>(bf[j], af[i], acc[i][j]);
    }

    if (more) store_stage(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: swapped roles => D rows = channels ((lane>>4)*4 + reg), D cols = pixels (lane&15)
  const int epix = lane & 15;
  const int en = (lane >> 4) * 4;
  if (p.ksplit > 1) {      // raw partial sums; the epilogue runs in splitk_finish_kernel
    float* part = p.partial + (size_t)blockIdx.y * p.M * p.Npad;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + (wm * MT + i) * 16 + epix;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = n0 + (wn * NT + j) * 16 + en;
        if (n < p.Npad) *reinterpret_cast<f32x4*>(part + (size_t)m * p.Npad + n) = acc[i][j];
      }
    }
    return;
  }
  float sc[NT][4], sh[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + (wn * NT + j) * 16 + en + q;
      const bool ok = n < p.N;
      sc[j][q] = (ok && p.scale) ? p.scale[n] : 1.0f;
      sh[j][q] = (ok && p.shift) ? p.shift[n] : 0.0f;
    }
  auto epilogue = [&](auto act_tag) {
  constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m0 + (wm * MT + i) * 16 + epix;
    if (m >= p.M) continue;
    size_t pbase = 0;
    if (p.out_mode == CCVPE_OUT_DECONV2X) {
      const int hw = p.Ho * p.Wo;
      const int b = m / hw;
      const int rem = m - b * hw;
      const int y = rem / p.Wo;
      const int x = rem - y * p.Wo;
      pbase = ((size_t)(b * 2 * p.Ho + 2 * y) * (2 * p.Wo) + 2 * x) * p.ldd;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + (wn * NT + j) * 16 + en;
      if (n >= p.N) continue;
      size_t obase;
      if (p.out_mode == CCVPE_OUT_NHWC) {
        obase = (size_t)m * p.ldd + n;
      } else {
        const int quad = n / p.cout;
        const int co = n - quad * p.cout;
        obase = pbase + ((size_t)(quad >> 1) * (2 * p.Wo) + (quad & 1)) * p.ldd + co;
      }
      store4<T, ACT>(p, acc[i][j], n, obase, (size_t)m * p.ldres, sc[j], sh[j]);
    }
  }
  };
  CCVPE_ACT_DISPATCH(p.act, epilogue);
}

// ---------------------------------------------------------------------------------------------
// Pointwise (1x1, stride 1, one source) GEMM: the EfficientNet expand / project / head convs and the fused ground
// descriptor conv (efficientnet_pytorch/model.py:62,86,209; models.py:57-97) and their 1x1 input gradients.
//
// These layers have SHORT K (16 ... 1152 channels) against millions of pixels: with 64-byte K stages the generic kernel
// above is a chain of dependent (global load -> LDS -> barrier) round trips per tile (4-72 of them) with ~15 KB in flight
// per workgroup, and its epilogue stores 8-16 bytes per lane at a pixel stride — the profile showed 0.5-1.2 TB/s
// algorithmic on layers whose roof is HBM.  Here:
//   * a K stage is KP x 64 bytes per row (KP = 4: 64 fp32 / 128 bf16 channels): one stage covers the whole K of most
//     expand convs in bf16; 4x fewer barriers and 4x the bytes in flight; 16 consecutive lanes read 256 contiguous bytes
//     of one pixel row (full cache lines instead of 64-byte fragments);
//   * register-staged, ONE LDS buffer (the loads of stage s+1 are in flight during the MFMAs of stage s; 2 workgroups per
//     CU cover the store/barrier bubbles) — register staging keeps the SE-gate multiply and the zero fill of the K / M tails;
//   * the epilogue goes through LDS: scale/shift/activation in registers, fp32 tile rows in LDS, then every thread stores
//     16 bytes with consecutive lanes along the channel axis — whole output rows (and whole residual rows) per wave.
// Same MFMA k-order as igemm_kernel (stage pieces in order, the q/kk permutation inside a 64-byte piece): fp32 results are
// bit-identical to the generic kernel's.
// ---------------------------------------------------------------------------------------------
template <typename T, int MT, int NT, int WN>
struct PwGeom {
  static constexpr int WM = 4 / WN;
  static constexpr int BM = 16 * MT * WM;
  static constexpr int BN = 16 * NT * WN;
  static constexpr int LDS_BUDGET = 80 * 1024;                                   // 2 workgroups per CU
  // 64-byte pieces per staged row.  2 (= 128 bytes of K per row: full cache lines, half the barriers of the generic
  // kernel): the staging registers of the NEXT tile's first stage stay live across the epilogue (persistent loop), and
  // with 4 pieces they push the 14-20 accumulator tiles over the 256-VGPR cap of 2 waves per SIMD (spills).
  static constexpr int KP = 2;
  static constexpr int LDF = 16 * KP + 4;                                        // floats per staged row
  static constexpr int OLD = BN + 4;                                             // floats per epilogue-tile row
  static constexpr int IC = (BM * OLD * 4 <= LDS_BUDGET) ? MT : MT / 2;          // MFMA row tiles per epilogue pass
  static constexpr int STAGE_BYTES = (BM + BN) * LDF * 4;
  static constexpr int OUT_BYTES = WM * IC * 16 * OLD * 4;
  static constexpr int TILE_BYTES = STAGE_BYTES > OUT_BYTES ? STAGE_BYTES : OUT_BYTES;
  static constexpr int LDS_BYTES = TILE_BYTES + 2 * BN * 4;                      // + the tile's scale / shift vectors
};

template <typename T, int MT, int NT, int WN, int ACT>
__global__ __launch_bounds__(256, 2) void pw_gemm_kernel(const IgemmParams p) {
  using G = PwGeom<T, MT, NT, WN>;
  constexpr int E = ElemTraits<T>::E;
  constexpr int WM = G::WM, BM = G::BM, BN = G::BN, KP = G::KP, LDF = G::LDF, OLD = G::OLD, IC = G::IC;
  constexpr int PPR = 4 * KP;                 // 16-byte pieces per staged row
  constexpr int KS = PPR * E;                 // K elements per stage
  constexpr int A_IT = BM * PPR / 256;
  constexpr int B_IT = (BN * PPR + 255) / 256;
  // next-tile prefetch across the epilogue keeps the staging registers live there: the 20-accumulator tile would spill
  constexpr bool PREFETCH = MT * NT < 20;

  extern __shared__ __attribute__((aligned(16))) float pw_sm[];
  float* As = pw_sm;                          // [BM][LDF]
  float* Bs = pw_sm + BM * LDF;               // [BN][LDF]
  float* Os = pw_sm;                          // epilogue tile [WM*IC*16][OLD] (aliases the dead stage buffers)
  float* Ss = pw_sm + G::TILE_BYTES / 4;      // [2][BN] scale, shift of the tile's channels (fetched with the first K
                                              // stage: the epilogue must not start with a chain of dependent global loads)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave % WN;
  const T* src0 = reinterpret_cast<const T*>(p.src0);
  const T* wp = reinterpret_cast<const T*>(p.w);
  const int hw = p.Ho * p.Wo;
  const int nstages = (p.c0 + KS - 1) / KS;
  const bool f32out = sizeof(T) == 4 || p.out_f32;
  const T* res = reinterpret_cast<const T*>(p.residual);

  const int prow = tid / PPR;                 // staged row of iteration 0; iteration `it` adds it * (256 / PPR)
  const int pc = tid % PPR;                   // 16-byte piece inside the staged row
  const int frow = lane & 15;
  const int fk = (lane >> 4) * 4;
  const int epix = lane & 15;
  const int en = (lane >> 4) * 4;
  f32x4 a_reg[A_IT], b_reg[B_IT];
  float sc_reg = 1.0f, sh_reg = 0.0f;

  auto load_ss = [&](int n0) {                // scale / shift of channel n0 + tid (threads < BN)
    const int n = n0 + tid;
    const bool ok = tid < BN && n < p.N;
    sc_reg = (ok && p.scale) ? p.scale[n] : 1.0f;
    sh_reg = (ok && p.shift) ? p.shift[n] : 0.0f;
  };
  // Staging (see STAGING RULE at the top of the file): raw loads from clamped addresses; row / K validity is kept as bits
  // and applied, with the SE gate, when the pieces are written to LDS.  W needs no mask: its rows beyond N and columns
  // beyond K are zero padding, rows beyond Npad are never stored, and a K piece beyond Kpad meets a zeroed A piece.
  unsigned row_ok = 0;                        // bit `it`: staged row `it` of the tile in a_reg is < M
  bool k_ok = false;                          // the staged K piece is < c0
  int st_m0 = 0, g_kc = 0, g_mend = 0;        // tile row base of a_reg; gate: piece channel, end row of the first row's sample
  f32x4 g_r0 = {0.f, 0.f, 0.f, 0.f}, g_r1 = g_r0;
  const bool gated = p.gate != nullptr;
  const int ld0s = sgpr(p.ld0);
  const unsigned ld0b = (unsigned)ld0s * (unsigned)sizeof(T);
  const bool small32 = (double)p.M * (double)ld0s * sizeof(T) < 4294967296.0;
  auto load_stage = [&](int m0, int n0, int s) {
    const int kcol = s * KS + pc * E;         // first K element of this thread's piece
    k_ok = kcol < p.c0;
    const int kc = k_ok ? kcol : 0;
    row_ok = 0;
    st_m0 = m0;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int m = m0 + prow + it * (256 / PPR);
      const bool rok = m < p.M;
      const int mc = rok ? m : p.M - 1;
      // 32-bit byte offset when the tensor allows it (workgroup-uniform switch): one v_mad instead of a 64-bit multiply-add
      if (small32) a_reg[it] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(src0) + ((unsigned)mc * ld0b + (unsigned)kc * (unsigned)sizeof(T)));
      else a_reg[it] = *reinterpret_cast<const f32x4*>(src0 + (size_t)mc * ld0s + kc);
      row_ok |= rok ? (1u << it) : 0u;
    }
    if (gated) {
      const int bf = min(m0 + prow, p.M - 1) / hw;
      gate_load<T>(p.gate + (size_t)bf * p.c0 + kc, g_r0, g_r1);
      g_kc = kc;
      g_mend = (bf + 1) * hw;
    }
    const unsigned wkb = (unsigned)(kcol < p.Kpad ? kcol : 0) * (unsigned)sizeof(T);
    const unsigned kpb = (unsigned)p.Kpad * (unsigned)sizeof(T);
    const char* wb = reinterpret_cast<const char*>(wp);
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const unsigned nr = (unsigned)min(n0 + prow + it * (256 / PPR), p.Npad - 1);
      b_reg[it] = *reinterpret_cast<const f32x4*>(wb + (nr * kpb + wkb));      // 32-bit offsets: W is far below 4 GB
    }
  };
  auto store_stage = [&]() {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      f32x4 v = a_reg[it];
      if (gated) {
        const int m = st_m0 + prow + it * (256 / PPR);
        if (m < g_mend) v = gate_apply<T>(v, g_r0, g_r1);                       // same sample as the thread's first row: the usual case
        else v = apply_gate<T>(v, p.gate + (size_t)(min(m, p.M - 1) / hw) * p.c0 + g_kc);
      }
      *reinterpret_cast<f32x4*>(&As[(prow + it * (256 / PPR)) * LDF + pc * 4]) = keep_if(v, k_ok && ((row_ok >> it) & 1u));
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int nrow = prow + it * (256 / PPR);
      if (nrow < BN) *reinterpret_cast<f32x4*>(&Bs[nrow * LDF + pc * 4]) = b_reg[it];
    }
  };

  // PERSISTENT workgroups: virtual block v = blockIdx.x + k * gridDim.x (gridDim.x is a multiple of 8, so every virtual
  // block of a workgroup maps to the same XCD and xcd_tile() keeps giving each XCD a contiguous run of tiles, n fastest).
  // The first K stage of the NEXT tile is loaded into the staging registers before the epilogue of the current one, so
  // the HBM latency of a tile's prologue hides behind the previous tile's epilogue.
  int v = blockIdx.x;
  if (v >= p.tiles_total) return;
  int tile = xcd_tile(v, p.tiles_total);
  int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
  load_stage(m0, n0, 0);
  load_ss(n0);
  while (true) {
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    store_stage();
    if (tid < BN) {
      Ss[tid] = sc_reg;
      Ss[BN + tid] = sh_reg;
    }
    __syncthreads();
    const int vn = v + gridDim.x;
    const bool has_next = vn < p.tiles_total;
    int m0n = 0, n0n = 0;
    if (has_next) {
      const int tn_ = xcd_tile(vn, p.tiles_total);
      m0n = (tn_ / p.tiles_n) * BM;
      n0n = (tn_ % p.tiles_n) * BN;
    }
    for (int s = 0; s < nstages; ++s) {
      const bool more = s + 1 < nstages;
      if (more) load_stage(m0, n0, s + 1);
      else if (PREFETCH && has_next) {                   // next tile's first stage: lands during the epilogue below
        load_stage(m0n, n0n, 0);
        load_ss(n0n);
      }
#pragma unroll
      for (int kp = 0; kp < KP; ++kp) {
        if (s * KS + kp * 4 * E >= p.c0 || (p.ablate & 4)) break;   // K tail: whole 64-byte pieces beyond K are zero (uniform branch)
        f32x4 af[MT], bf[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
          af[i] = *reinterpret_cast<const f32x4*>(&As[((wm * MT + i) * 16 + frow) * LDF + kp * 16 + fk]);
#pragma unroll
        for (int j = 0; j < NT; ++j)
          bf[j] = *reinterpret_cast<const f32x4*>(&Bs[((wn * NT + j) * 16 + frow) * LDF + kp * 16 + fk]);
        if (sizeof(T) == 4) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
              for (int j = 0; j < NT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][kk], af[i][kk], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = mfma_stage<T>(bf[j], af[i], acc[i][j]);
        }
      }
      __syncthreads();                        // every wave is done reading this stage
      if (more) {
        store_stage();
        __syncthreads();
      }
    }

    // ---- epilogue: registers -> (scale, shift, act) -> LDS rows -> 16-byte stores along the channel axis ----------
#pragma unroll
    for (int ic = 0; ic < MT / IC; ++ic) {
      // (the activation is a template parameter: a per-element `if (p.act == ...)` compiled to a scalar compare + branch
      // per value — 8 000 cycles per tile for 56 values per lane)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int nl = (wn * NT + j) * 16 + en;           // tile-local channel of acc[.][j][0]
        const f32x4 sc = *reinterpret_cast<const f32x4*>(&Ss[nl]);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(&Ss[BN + nl]);
#pragma unroll
        for (int ii = 0; ii < IC; ++ii) {
          f32x4 vv = acc[ic * IC + ii][j] * sc + sh;
          if (ACT == CCVPE_ACT_SWISH) {
#pragma unroll
            for (int q = 0; q < 4; ++q) vv[q] = swishf(vv[q]);
          }
          *reinterpret_cast<f32x4*>(&Os[((wm * IC + ii) * 16 + epix) * OLD + nl]) = vv;
        }
      }
      __syncthreads();
      // store phase without integer divisions: a thread owns one 16-byte column piece and walks down the rows.
      // Piece columns are padded to a power of two (PCP); row r of the tile <-> pixel via shifts (IC * 16 is a power of 2).
      auto store_rows = [&](auto eo_tag) {
        constexpr int EO = decltype(eo_tag)::value;      // output elements per 16 bytes: 4 (fp32) or 8 (bf16)
        constexpr int PPO = BN / EO;
        constexpr int PCP = PPO <= 2 ? 2 : PPO <= 4 ? 4 : PPO <= 8 ? 8 : PPO <= 16 ? 16 : PPO <= 32 ? 32 : 64;
        constexpr int RSTEP = 256 / PCP;
        const int pcol = tid % PCP;
        const int c = pcol * EO;
        const int n = n0 + c;
        if (pcol < PPO && n < p.N && !(p.ablate & 1)) {
          const bool full = n + EO <= p.N;
#pragma unroll 2
          for (int r = tid / PCP; r < WM * IC * 16; r += RSTEP) {
            const int m = m0 + ((r / (IC * 16)) * MT + ic * IC) * 16 + (r % (IC * 16));
            if (m >= p.M) continue;
            const float* o = &Os[r * OLD + c];
            f32x4 v0 = *reinterpret_cast<const f32x4*>(o);
            f32x4 v1 = {0.f, 0.f, 0.f, 0.f};
            if (EO == 8) v1 = *reinterpret_cast<const f32x4*>(o + 4);
            const size_t obase = (size_t)m * p.ldd + n;
            const size_t rbase = (size_t)m * p.ldres + n;
            if (full) {
              if (res) {
                if (sizeof(T) == 4) {
                  v0 += *reinterpret_cast<const f32x4*>(res + rbase);
                } else {          // bf16 residual: 8 channels = 16 bytes (4 when the output is fp32)
                  const bf16x4 r0 = *reinterpret_cast<const bf16x4*>(res + rbase);
#pragma unroll
                  for (int q = 0; q < 4; ++q) v0[q] += (float)r0[q];
                  if (EO == 8) {
                    const bf16x4 r1 = *reinterpret_cast<const bf16x4*>(res + rbase + 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) v1[q] += (float)r1[q];
                  }
                }
              }
              if (EO == 4) {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.dst) + obase) = v0;
              } else {
                bf16x8 ov;
#pragma unroll
                for (int q = 0; q < 4; ++q) { ov[q] = (bf16_t)v0[q]; ov[q + 4] = (bf16_t)v1[q]; }
                *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.dst) + obase) = ov;
              }
            } else {              // ragged N tail: element by element
              for (int q = 0; q < EO && n + q < p.N; ++q) {
                float t = q < 4 ? v0[q] : v1[q - 4];
                if (res) t += (float)res[rbase + q];
                if (EO == 4) reinterpret_cast<float*>(p.dst)[obase + q] = t;
                else reinterpret_cast<bf16_t*>(p.dst)[obase + q] = (bf16_t)t;
              }
            }
          }
        }
      };
      if (f32out) store_rows(std::integral_constant<int, 4>{});
      else store_rows(std::integral_constant<int, 8>{});
      __syncthreads();                        // the tile rows are dead: next epilogue pass / next tile's stage may overwrite
    }
    if (!has_next) break;
    v = vn;
    m0 = m0n;
    n0 = n0n;
    if (!PREFETCH) {
      load_stage(m0, n0, 0);
      load_ss(n0);
    }
  }
}

template <typename T, int MT, int NT, int WN, int ACT>
static int launch_pw_act(const IgemmParams& p0, hipStream_t stream) {
  using G = PwGeom<T, MT, NT, WN>;
  IgemmParams p = p0;
  const int tiles_m = (p.M + G::BM - 1) / G::BM;
  p.tiles_n = (p.Npad + G::BN - 1) / G::BN;
  p.tiles_total = tiles_m * p.tiles_n;
  p.ksplit = 1;
  p.sps = p.stages;
  p.partial = nullptr;
  static const int ablate = getenv("CCVPE_PW_ABLATE") ? atoi(getenv("CCVPE_PW_ABLATE")) : 0;
  p.ablate = ablate;
  static bool attr_set = false;               // one flag per instantiation
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)pw_gemm_kernel<T, MT, NT, WN, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       G::LDS_BYTES);
    if (e != hipSuccess) return fail(CCVPE_ELAUNCH, "pw_gemm: set smem attr: %s", hipGetErrorString(e));
    attr_set = true;
  }
  // persistent grid: 2 workgroups per CU (LDS / VGPR budget of the kernel), a multiple of 8 so that the virtual-block ->
  // XCD map is stable over a workgroup's iterations
  int grid = p.tiles_total < 512 ? (p.tiles_total + 7) / 8 * 8 : 512;
  hipLaunchKernelGGL((pw_gemm_kernel<T, MT, NT, WN, ACT>), dim3(grid), dim3(256), G::LDS_BYTES, stream, p);
  return check_launch("pw_gemm_kernel");
}

template <typename T, int MT, int NT, int WN>
static int launch_pw(const IgemmParams& p, hipStream_t stream) {
  if (p.act == CCVPE_ACT_SWISH) return launch_pw_act<T, MT, NT, WN, CCVPE_ACT_SWISH>(p, stream);
  return launch_pw_act<T, MT, NT, WN, CCVPE_ACT_NONE>(p, stream);
}

// ---------------------------------------------------------------------------------------------
// 3x3, stride 1, pad 1, two concatenated sources, halo tile in LDS.
// Pixel tile = TH rows x 16 columns (one MFMA tile = 16 consecutive x of one row), TH = BM/16.
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
__device__ __forceinline__ int w_swz(int r) { return (0x1320 >> (((r >> 2) & 3) * 4)) & 3; }
// taps per stage of the DMA 3x3 kernel: 3 (a row of taps, default) or 1 (CCVPE_CONV3_TPS=1, for A/B runs)
// CCVPE_CONV3_WREG=1: fp32 3x3 convolutions through conv3x3_wreg_kernel (W fragments straight from L2)
static const bool g_conv3_wreg = getenv("CCVPE_CONV3_WREG") && getenv("CCVPE_CONV3_WREG")[0] == '1';
// CCVPE_CONV3_NW8=0: never use the 8-wave form of the 128-column tile (A/B runs)
static const bool g_conv3_nw8 = !(getenv("CCVPE_CONV3_NW8") && getenv("CCVPE_CONV3_NW8")[0] == '0');
static const int g_conv3_tps = (getenv("CCVPE_CONV3_TPS") && getenv("CCVPE_CONV3_TPS")[0] == '1') ? 1 : 3;

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
  static constexpr int HS_FLOATS = HPX * LDS_LD;
  static constexpr int BS_FLOATS = 2 * TPS * BN * BLD;
  static constexpr int LDS_BYTES = (HS_FLOATS + BS_FLOATS) * 4;
};

template <typename T, int MT, int NT, int WN, int NW, bool DMA, int TPS>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 2) void conv3x3_kernel(const IgemmParams p) {
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
      h_off[it] = px * LDS_LD + sub * 4;
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
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int i = 0; i < MT; ++i)
        a[i] = *reinterpret_cast<const f32x4*>(Hs + (((wm * MT + i) + ky) * HC + frow + kx) * LDS_LD + fk);
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
      h_off[it] = px * LDS_LD + sub * 4;
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

// split-K second pass: thread = (pixel m, 4 channels); adds the K slices in index order (deterministic) and applies
// the same epilogue as the one-pass kernel (scale/shift, activation, residual, NHWC or pixel-shuffle addressing)
template <typename T>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const IgemmParams p) {
  const int n4 = p.Npad >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)p.M * n4) return;
  const int m = (int)(idx / n4);
  const int n = (int)(idx - (long)m * n4) * 4;
  if (n >= p.N) return;
  f32x4 v = *reinterpret_cast<const f32x4*>(p.partial + (size_t)m * p.Npad + n);
  for (int k = 1; k < p.ksplit; ++k) v += *reinterpret_cast<const f32x4*>(p.partial + ((size_t)k * p.M + m) * p.Npad + n);
  float sc[4], sh[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const bool ok = n + q < p.N;
    sc[q] = (ok && p.scale) ? p.scale[n + q] : 1.0f;
    sh[q] = (ok && p.shift) ? p.shift[n + q] : 0.0f;
  }
  size_t obase;
  if (p.out_mode == CCVPE_OUT_NHWC) {
    obase = (size_t)m * p.ldd + n;
  } else {
    const int hw = p.Ho * p.Wo;
    const int b = m / hw;
    const int rem = m - b * hw;
    const int y = rem / p.Wo;
    const int x = rem - y * p.Wo;
    const size_t pbase = ((size_t)(b * 2 * p.Ho + 2 * y) * (2 * p.Wo) + 2 * x) * p.ldd;
    const int quad = n / p.cout;
    const int co = n - quad * p.cout;
    obase = pbase + ((size_t)(quad >> 1) * (2 * p.Wo) + (quad & 1)) * p.ldd + co;
  }
  auto fin = [&](auto act_tag) { store4<T, decltype(act_tag)::value>(p, v, n, obase, (size_t)m * p.ldres, sc, sh); };
  CCVPE_ACT_DISPATCH(p.act, fin);
}

// K slices for a GEMM with `tiles` workgroups and `stages` K stages (1 = no split): only when the launch cannot fill
// the 256 CUs and the serial K walk is long; each slice keeps >= 12 stages
static int splitk_slices(int tiles, int stages) {
  if (tiles >= 320 || stages < 24) return 1;
  int s = (640 + tiles - 1) / tiles;
  if (s > stages / 12) s = stages / 12;
  if (s > 32) s = 32;
  return s < 2 ? 1 : s;
}

template <typename T, int MT, int NT, int WN>
static int launch(const IgemmParams& p0, hipStream_t stream, float* scratch = nullptr, long* want_floats = nullptr) {
  constexpr int WM = 4 / WN;
  constexpr int BM = 16 * MT * WM;
  constexpr int BN = 16 * NT * WN;
  IgemmParams p = p0;
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.Npad + BN - 1) / BN;
  p.tiles_total = tiles_m * p.tiles_n;
  p.ksplit = 1;
  p.sps = p.stages;
  p.partial = nullptr;
  const int S = splitk_slices(p.tiles_total, p.stages);
  if (want_floats) {                       // planning call: report the scratch a split-K run needs, launch nothing
    *want_floats = S > 1 ? (long)S * p.M * p.Npad : 0;
    return CCVPE_OK;
  }
  if (scratch && S > 1) {
    p.sps = (p.stages + S - 1) / S;
    p.ksplit = (p.stages + p.sps - 1) / p.sps;
    p.partial = scratch;
    hipLaunchKernelGGL((igemm_kernel<T, MT, NT, WN>), dim3(p.tiles_total, p.ksplit), dim3(256), 0, stream, p);
    const long n = (long)p.M * (p.Npad / 4);
    hipLaunchKernelGGL((splitk_finish_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p);
    return check_launch("igemm_kernel (split-K)");
  }
  hipLaunchKernelGGL((igemm_kernel<T, MT, NT, WN>), dim3(p.tiles_total), dim3(256), 0, stream, p);
  return check_launch("igemm_kernel");
}

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
  if constexpr (NW == 4 && sizeof(T) == 4) {
    if (g_conv3_wreg && (size_t)p.Npad * p.Kpad < (1u << 30)) {        // 32-bit W offsets
      hipLaunchKernelGGL((conv3x3_wreg_kernel<T, MT, NT, WN>), dim3(p.tiles_total), dim3(256), 0, stream, p);
      return check_launch("conv3x3_wreg_kernel");
    }
  }
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
    if constexpr (NW == 4 && !(NT == 5 && WN == 1)) {      // (256 x 80 tile: a row of taps per stage measured the same, 117.8 vs 117.0 TF)
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
  return launch3x3_nw<T, MT, NT, WN, 4>(p0, batch, stream);
}

// ---------------------------------------------------------------------------------------------
// ConvTranspose2d(k2,s2) folded into the following 3x3 conv (models.py:207-209: deconv -> cat skip ->
// conv.0).  For output parity (py,px) the pair is ONE implicit GEMM over LOW-RES pixels (y1,x1):
//   out[2y1+py, 2x1+px, n] = sum_{du,dv in {0,1}} Weff[py,px,du,dv][n,:] . x[y1+du-1+py, x1+dv-1+px, :]
//                          + sum_{ky,kx}          W3[n, Cd:, ky,kx]     . skip[2y1+py+ky-1, 2x1+px+kx-1, :]
//                          + shift9[border class of (Y,X)][n]
// with Weff = sum over the (ky,a)/(kx,b) pairs that land on that low-res pixel of W3[:, :Cd, ky,kx] . Wd[:,:,a,b]^T
// (packed by the host, ccvpe_amd/models.py:_pack_upconv).  K = 4*C' + 9*C1 instead of 9*(Cd + C1) plus the
// deconv GEMM, and the 2x-upsampled deconv tensor never exists.  The deconv bias only survives for the
// 3x3 taps that fall inside the image, hence the 9 (row class x column class) shift vectors.
// ---------------------------------------------------------------------------------------------
struct UpParams {
  const void* src0;
  const void* src1;
  const void* w;
  const float* shift9;
  void* dst;
  int out_f32;
  int c0, ld0, c1, ld1;
  int H1, W1;
  int N, Kpad, Npad;
  int cpt0, cpt1, total_chunks, stages;
  int ldd, act;
  int M;                 // batch * H1 * W1 (low-res pixels)
  int tiles_n, tiles_m, tiles_total;
};

template <typename T, int MT, int NT, int WN>
__global__ __launch_bounds__(256) void upconv_kernel(const UpParams p) {
  constexpr int E = ElemTraits<T>::E;
  constexpr int SK = 4 * E;
  constexpr int CPS = SK / 8;
  constexpr int WM = 4 / WN;
  constexpr int BM = 16 * MT * WM;
  constexpr int BN = 16 * NT * WN;
  constexpr int A_IT = BM / 64;
  constexpr int B_IT = (BN + 63) / 64;

  __shared__ __attribute__((aligned(16))) float As[2][BM][LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN][LDS_LD];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave % WN;

  // tile order: n fastest, then the 4 parities of one low-res tile (they share src0 / skip rows in L2)
  const int tile = xcd_tile(blockIdx.x, p.tiles_total);
  const int tn = tile % p.tiles_n;
  const int par = (tile / p.tiles_n) & 3;
  const int tm = tile / (p.tiles_n * 4);
  const int py = par >> 1, px = par & 1;
  const int m0 = tm * BM;
  const int n0 = tn * BN;
  const int H2 = 2 * p.H1, W2 = 2 * p.W1;

  const int srow = tid >> 2;
  const int ssub = tid & 3;
  const int chunk_in_stage = (ssub * E) >> 3;
  const int half = (ssub * E) & 7;
  const T* src0 = reinterpret_cast<const T*>(p.src0);
  const T* src1 = reinterpret_cast<const T*>(p.src1);
  const T* wp = reinterpret_cast<const T*>(p.w) + (size_t)par * p.Npad * p.Kpad;

  int a_b[A_IT], a_y[A_IT], a_x[A_IT];
  bool a_ok[A_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int m = m0 + srow + 64 * it;
    a_ok[it] = m < p.M;
    const int mm = a_ok[it] ? m : 0;
    const int hw = p.H1 * p.W1;
    const int b = mm / hw;
    const int rem = mm - b * hw;
    a_b[it] = b;
    a_y[it] = rem / p.W1;
    a_x[it] = rem - a_y[it] * p.W1;
  }
  const int k0end = 4 * p.cpt0;      // chunks belonging to the low-res source

  f32x4 a_reg[A_IT], b_reg[B_IT];
  unsigned wrow[B_IT];                      // stage-invariant byte offset of this lane's W piece per staged row (see igemm_kernel)
#pragma unroll
  for (int it = 0; it < B_IT; ++it)
    wrow[it] = ((unsigned)min(n0 + srow + 64 * it, p.Npad - 1) * (unsigned)p.Kpad + (unsigned)(ssub * E)) * (unsigned)sizeof(T);
  unsigned a_keep = 0;                      // bit `it`: piece `it` in a_reg is inside its image and the K range
  const int ld0s = sgpr(p.ld0), ld1s = sgpr(p.ld1);
  // both sources below 4 GB (workgroup-uniform): 32-bit byte offsets, one v_mad per piece instead of 64-bit multiply-adds
  const bool small32 = (double)p.M * 4.0 * (double)(ld0s > ld1s ? ld0s : ld1s) * sizeof(T) < 4294967296.0;

  // (an incrementally advanced cursor instead of the two divisions measured 5 % SLOWER: 96.8 vs 101.6 TF)
  // One unconditional load per piece (STAGING RULE): the source, its geometry and the tap offset are selected per thread
  // first; lanes outside the image read pixel 0 and are zeroed when the piece is written to LDS.
  auto load_stage = [&](int s) {
    const int kc = CPS * s + chunk_in_stage;
    const bool kvalid = kc < p.total_chunks;
    const bool from0 = !kvalid || kc < k0end;
    int dy, dx, ch;
    {
      const int tap0 = kc / p.cpt0;
      const int k2 = kc - k0end;
      const int tap1 = p.cpt1 > 0 ? k2 / p.cpt1 : 0;
      const int ky = tap1 / 3;
      ch = kvalid ? (from0 ? (kc - tap0 * p.cpt0) : (k2 - tap1 * p.cpt1)) * 8 + half : 0;
      dy = from0 ? (tap0 >> 1) - 1 + py : py + ky - 1;
      dx = from0 ? (tap0 & 1) - 1 + px : px + (tap1 - 3 * ky) - 1;
    }
    const T* base = from0 ? src0 : src1;
    const int ld = from0 ? ld0s : ld1s;
    const int mul = from0 ? 1 : 2;
    const int hh = from0 ? p.H1 : H2, ww = from0 ? p.W1 : W2;
    a_keep = 0;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int iy = mul * a_y[it] + dy, ix = mul * a_x[it] + dx;
      const bool ok = a_ok[it] && kvalid && (unsigned)iy < (unsigned)hh && (unsigned)ix < (unsigned)ww;
      const int pix = ok ? (a_b[it] * hh + iy) * ww + ix : 0;
      if (small32) a_reg[it] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + ((unsigned)pix * (unsigned)ld + (unsigned)ch) * (unsigned)sizeof(T));
      else a_reg[it] = *reinterpret_cast<const f32x4*>(base + (size_t)pix * ld + ch);
      a_keep |= ok ? (1u << it) : 0u;
    }
    {
      const char* wb = reinterpret_cast<const char*>(wp) + (size_t)s * (SK * sizeof(T));   // scalar base + per-lane row offset
#pragma unroll
      for (int it = 0; it < B_IT; ++it) b_reg[it] = *reinterpret_cast<const f32x4*>(wb + wrow[it]);
    }
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it)
      *reinterpret_cast<f32x4*>(&As[buf][srow + 64 * it][ssub * 4]) = keep_if(a_reg[it], (a_keep >> it) & 1u);
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int nrow = srow + 64 * it;
      if (nrow < BN) *reinterpret_cast<f32x4*>(&Bs[buf][nrow][ssub * 4]) = b_reg[it];
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;
  const int fk = (lane >> 4) * 4;

  load_stage(0);
  store_stage(0);
  __syncthreads();
  for (int s = 0; s < p.stages; ++s) {
    const int buf = s & 1;
    const bool more = s + 1 < p.stages;
    if (more) load_stage(s + 1);
    f32x4 af[MT], bf[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
      af[i] = *reinterpret_cast<const f32x4*>(&As[buf][(wm * MT + i) * 16 + frow][fk]);
#pragma unroll
    for (int j = 0; j < NT; ++j)
      bf[j] = *reinterpret_cast<const f32x4*>(&Bs[buf][(wn * NT + j) * 16 + frow][fk]);
    if (sizeof(T) == 4) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][kk], af[i][kk], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma_stage<T>(bf[j], af[i], acc[i][j]);
    }
    if (more) store_stage(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: pixel (2y1+py, 2x1+px); shift picked by the pixel's border class --------------
  const int epix = lane & 15;
  const int en = (lane >> 4) * 4;
  IgemmParams ep;   // reuse store4 (needs N, act, residual, dst, out_f32)
  ep.N = p.N; ep.act = p.act; ep.residual = nullptr; ep.dst = p.dst; ep.out_f32 = p.out_f32;
  const float one[4] = {1.f, 1.f, 1.f, 1.f};
  auto epilogue = [&](auto act_tag) {
  constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m0 + (wm * MT + i) * 16 + epix;
    if (m >= p.M) continue;
    const int hw = p.H1 * p.W1;
    const int b = m / hw;
    const int rem = m - b * hw;
    const int y1 = rem / p.W1;
    const int Y = 2 * y1 + py, X = 2 * (rem - y1 * p.W1) + px;
    const int rc = Y == 0 ? 0 : (Y == H2 - 1 ? 2 : 1);
    const int cc = X == 0 ? 0 : (X == W2 - 1 ? 2 : 1);
    const float* shp = p.shift9 + (size_t)(rc * 3 + cc) * p.N;
    const size_t pix = (size_t)(b * H2 + Y) * W2 + X;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + (wn * NT + j) * 16 + en;
      if (n >= p.N) continue;
      float sh[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) sh[q] = (n + q < p.N) ? shp[n + q] : 0.f;
      store4<T, ACT>(ep, acc[i][j], n, pix * p.ldd + n, 0, one, sh);
    }
  }
  };
  CCVPE_ACT_DISPATCH(p.act, epilogue);
}

// ---------------------------------------------------------------------------------------------
// upconv with the LOW-RES source staged as a halo tile (phase A) and the skip gathered (phase B).
// For parity (py,px) the four low-res taps are the (py..py+1) x (px..px+1) corner of the ordinary
// 3x3 halo neighbourhood, so phase A is the conv3x3 machinery with a 4-tap list: the low-res
// activation (64-72 % of K at levels 6-3) is fetched from L2 once per 16-channel chunk instead of
// once per tap.  Phase B walks the skip's 9 taps with the generic gather into an A stage that aliases
// the (now dead) halo buffer.  Same accumulators, same epilogue.
// ---------------------------------------------------------------------------------------------
template <typename T, int MT, int NT, int WN>
__global__ __launch_bounds__(256) void upconv_halo_kernel(const UpParams p) {
  constexpr int E = ElemTraits<T>::E;
  constexpr int SK = 4 * E;
  constexpr int CPS = SK / 8;
  constexpr int WM = 4 / WN;
  constexpr int BM = 16 * MT * WM;
  constexpr int BN = 16 * NT * WN;
  constexpr int TH = BM / 16;
  constexpr int HR = TH + 2, HC = 18, HPX = HR * HC;
  constexpr int H_IT = (HPX * 4 + 255) / 256;
  constexpr int A_IT = BM / 64;
  constexpr int B_IT = (BN + 63) / 64;
  constexpr int UROWS = (HPX > 2 * BM) ? HPX : 2 * BM;       // halo [HPX] rows  |  A stage [2][BM] rows

  __shared__ __attribute__((aligned(16))) float Us[UROWS][LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN][LDS_LD];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN;
  const int wn = wave % WN;

  const int tile = xcd_tile(blockIdx.x, p.tiles_total);
  const int tn = tile % p.tiles_n;
  const int par = (tile / p.tiles_n) & 3;
  const int ts = tile / (p.tiles_n * 4);                      // low-res spatial tile: x fastest, y, sample
  const int tiles_x = (p.W1 + 15) / 16;
  const int tiles_y = (p.H1 + TH - 1) / TH;
  const int tx = ts % tiles_x;
  const int ty = (ts / tiles_x) % tiles_y;
  const int b = ts / (tiles_x * tiles_y);
  const int py = par >> 1, px = par & 1;
  const int y0 = ty * TH, x0 = tx * 16;
  const int n0 = tn * BN;
  const int H2 = 2 * p.H1, W2 = 2 * p.W1;

  const T* src0 = reinterpret_cast<const T*>(p.src0);
  const T* src1 = reinterpret_cast<const T*>(p.src1);
  const T* wp = reinterpret_cast<const T*>(p.w) + (size_t)par * p.Npad * p.Kpad;
  const int srow = tid >> 2, ssub = tid & 3;

  // ---- phase A staging coordinates (halo of the low-res source) --------------------------------
  int h_off[H_IT], h_pix[H_IT], h_sub[H_IT];
#pragma unroll
  for (int it = 0; it < H_IT; ++it) {
    const int idx = tid + 256 * it;
    const int pxl = idx >> 2, sub = idx & 3;
    h_sub[it] = sub;
    if (pxl < HPX) {
      const int hy = pxl / HC, hx = pxl - hy * HC;
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      h_off[it] = pxl * LDS_LD + sub * 4;
      h_pix[it] = ((unsigned)iy < (unsigned)p.H1 && (unsigned)ix < (unsigned)p.W1) ? (b * p.H1 + iy) * p.W1 + ix : -1;
    } else {
      h_off[it] = -1;
      h_pix[it] = -1;
    }
  }
  f32x4 h_reg[H_IT], b_reg[B_IT], a_reg[A_IT];
  unsigned h_keep = 0, a_keep = 0;          // validity bits of the pieces in h_reg / a_reg, applied at the LDS store
  const int ld0s = sgpr(p.ld0), ld1s = sgpr(p.ld1);
  const bool small32 = (double)p.M * 4.0 * (double)(ld0s > ld1s ? ld0s : ld1s) * sizeof(T) < 4294967296.0;   // 32-bit byte offsets
  auto load_halo = [&](int chunk) {         // raw loads from clamped addresses (STAGING RULE)
    h_keep = 0;
#pragma unroll
    for (int it = 0; it < H_IT; ++it) {
      const int ch = chunk * SK + h_sub[it] * E;
      const bool ok = h_pix[it] >= 0 && ch < p.c0;
      if (small32) h_reg[it] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(src0) + (ok ? ((unsigned)h_pix[it] * (unsigned)ld0s + (unsigned)ch) * (unsigned)sizeof(T) : 0u));
      else h_reg[it] = *reinterpret_cast<const f32x4*>(src0 + (ok ? (size_t)h_pix[it] * ld0s + ch : 0));
      h_keep |= ok ? (1u << it) : 0u;
    }
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int it = 0; it < H_IT; ++it)
      if (h_off[it] >= 0) *reinterpret_cast<f32x4*>(&Us[0][0] + h_off[it]) = keep_if(h_reg[it], (h_keep >> it) & 1u);
  };
  unsigned wrow[B_IT];                          // stage-invariant byte offset of the staged W rows (see igemm_kernel)
#pragma unroll
  for (int it = 0; it < B_IT; ++it)
    wrow[it] = (unsigned)min(n0 + srow + 64 * it, p.Npad - 1) * (unsigned)p.Kpad * (unsigned)sizeof(T);
  auto load_w = [&](int kcol, bool ok) {        // kcol: first K column of this lane's 16-byte piece
    const unsigned kb = (unsigned)(ok ? kcol : 0) * (unsigned)sizeof(T);   // (a piece beyond the channel range meets a zeroed activation piece)
    const char* wb = reinterpret_cast<const char*>(wp);
#pragma unroll
    for (int it = 0; it < B_IT; ++it) b_reg[it] = *reinterpret_cast<const f32x4*>(wb + (wrow[it] + kb));
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int nrow = srow + 64 * it;
      if (nrow < BN) *reinterpret_cast<f32x4*>(&Bs[buf][nrow][ssub * 4]) = b_reg[it];
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15;
  const int fk = (lane >> 4) * 4;

  auto mfma_block = [&](const f32x4* af, int wbuf) {
    f32x4 bf[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
      bf[j] = *reinterpret_cast<const f32x4*>(&Bs[wbuf][(wn * NT + j) * 16 + frow][fk]);
    if (sizeof(T) == 4) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][kk], af[i][kk], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma_stage<T>(bf[j], af[i], acc[i][j]);
    }
  };

  // ================= phase A: low-res source, 4 taps, halo in LDS ================================
  const int nchunks0 = (p.c0 + SK - 1) / SK;
  const int nstA = nchunks0 * 4;
  int wbuf = 0;                                   // W double-buffer index carried across both phases
  {
    load_halo(0);
    { const int ch = ssub * E; load_w(ch, ch < p.c0); }            // (chunk 0, tap 0)
    store_halo();
    store_w(0);
    __syncthreads();
    int chunk = 0, tap = 0;
    for (int s = 0; s < nstA; ++s) {
      const bool more = s + 1 < nstA;
      int nchunk = chunk, ntap = tap + 1;
      if (ntap == 4) { ntap = 0; ++nchunk; }
      const bool next_halo = (tap == 0) && (chunk + 1 < nchunks0);
      if (next_halo) load_halo(chunk + 1);       // before the W loads: re-using h_reg costs a vmcnt(0), harmless while nothing is in flight
      if (more) {
        const int ch = nchunk * SK + ssub * E;
        load_w(ntap * p.c0 + ch, ch < p.c0);
      }
      const int du = tap >> 1, dv = tap & 1;
      f32x4 af[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i)
        af[i] = *reinterpret_cast<const f32x4*>(&Us[0][0] + (((wm * MT + i) + du + py) * HC + frow + dv + px) * LDS_LD + fk);
      mfma_block(af, wbuf);
      if (more) store_w(wbuf ^ 1);
      __syncthreads();
      if (tap == 3 && more) {
        store_halo();
        __syncthreads();
      }
      if (more) wbuf ^= 1;
      chunk = nchunk;
      tap = ntap;
    }
  }

  // ================= phase B: skip, 9 taps (stride 2, parity offset), gathered ===================
  const int chunksB = 9 * p.cpt1;
  if (chunksB > 0) {
    const int stagesB = (chunksB + CPS - 1) / CPS;
    const int chunk_in_stage = (ssub * E) >> 3;
    const int half = (ssub * E) & 7;
    const int kB0 = 4 * p.c0;                     // first K column of the skip part
    int a_pix[A_IT];                              // low-res pixel (for validity) per staged row
    int a_yy[A_IT], a_xx[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int ml = srow + 64 * it;              // tile-local pixel: row = ml/16, col = ml%16
      const int y1 = y0 + (ml >> 4), x1 = x0 + (ml & 15);
      a_pix[it] = (y1 < p.H1 && x1 < p.W1) ? 1 : 0;
      a_yy[it] = 2 * y1 + py - 1;
      a_xx[it] = 2 * x1 + px - 1;
    }
    auto load_a = [&](int s) {
      const int kc = CPS * s + chunk_in_stage;
      const bool kvalid = kc < chunksB;
      const int tapb = kc / p.cpt1;
      const int ch = kvalid ? (kc - tapb * p.cpt1) * 8 + half : 0;
      const int ky = tapb / 3, kx = tapb - 3 * ky;
      a_keep = 0;
#pragma unroll
      for (int it = 0; it < A_IT; ++it) {
        const int iy = a_yy[it] + ky, ix = a_xx[it] + kx;
        const bool ok = kvalid && a_pix[it] && (unsigned)iy < (unsigned)H2 && (unsigned)ix < (unsigned)W2;
        const int pix = ok ? (b * H2 + iy) * W2 + ix : 0;
        if (small32) a_reg[it] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(src1) + ((unsigned)pix * (unsigned)ld1s + (unsigned)ch) * (unsigned)sizeof(T));
        else a_reg[it] = *reinterpret_cast<const f32x4*>(src1 + (size_t)pix * ld1s + ch);
        a_keep |= ok ? (1u << it) : 0u;
      }
    };
    auto store_a = [&](int buf) {
#pragma unroll
      for (int it = 0; it < A_IT; ++it)
        *reinterpret_cast<f32x4*>(&Us[buf * BM + srow + 64 * it][ssub * 4]) = keep_if(a_reg[it], (a_keep >> it) & 1u);
    };
    // the halo is dead (phase A ended on a barrier); W buffer `wbuf` was the last one read
    load_a(0);
    load_w(kB0 + ssub * E, true);
    store_a(0);
    store_w(wbuf ^ 1);
    wbuf ^= 1;
    __syncthreads();
    for (int s = 0; s < stagesB; ++s) {
      const int abuf = s & 1;
      const bool more = s + 1 < stagesB;
      if (more) {
        load_a(s + 1);
        load_w(kB0 + (s + 1) * SK + ssub * E, true);
      }
      f32x4 af[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i)
        af[i] = *reinterpret_cast<const f32x4*>(&Us[abuf * BM + (wm * MT + i) * 16 + frow][fk]);
      mfma_block(af, wbuf);
      if (more) {
        store_a(abuf ^ 1);
        store_w(wbuf ^ 1);
      }
      __syncthreads();
      if (more) wbuf ^= 1;
    }
  }

  // ---- epilogue -----------------------------------------------------------------------------------
  const int epix = lane & 15;
  const int en = (lane >> 4) * 4;
  IgemmParams ep;
  ep.N = p.N; ep.act = p.act; ep.residual = nullptr; ep.dst = p.dst; ep.out_f32 = p.out_f32;
  const float one[4] = {1.f, 1.f, 1.f, 1.f};
  const int x1 = x0 + epix;
  auto epilogue = [&](auto act_tag) {
  constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int y1 = y0 + wm * MT + i;
    if (y1 >= p.H1 || x1 >= p.W1) continue;
    const int Y = 2 * y1 + py, X = 2 * x1 + px;
    const int rc = Y == 0 ? 0 : (Y == H2 - 1 ? 2 : 1);
    const int cc = X == 0 ? 0 : (X == W2 - 1 ? 2 : 1);
    const float* shp = p.shift9 + (size_t)(rc * 3 + cc) * p.N;
    const size_t pix = (size_t)(b * H2 + Y) * W2 + X;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + (wn * NT + j) * 16 + en;
      if (n >= p.N) continue;
      float sh[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) sh[q] = (n + q < p.N) ? shp[n + q] : 0.f;
      store4<T, ACT>(ep, acc[i][j], n, pix * p.ldd + n, 0, one, sh);
    }
  }
  };
  CCVPE_ACT_DISPATCH(p.act, epilogue);
}

template <typename T, int MT, int NT, int WN>
static int launch_up(const UpParams& p0, hipStream_t stream) {
  constexpr int WM = 4 / WN;
  constexpr int BM = 16 * MT * WM;
  constexpr int BN = 16 * NT * WN;
  constexpr int TH = BM / 16;
  UpParams p = p0;
  p.tiles_n = (p.Npad + BN - 1) / BN;
  const int batch = p.M / (p.H1 * p.W1);
  // The halo kernel tiles the low-res image in TH x 16 pixel tiles: with W1 < 16 (level 6: 8x8) half of
  // every MFMA pixel tile would be padding, so those shapes use the linear-M gather kernel.
  const bool halo = p.W1 >= 16;
  p.tiles_m = halo ? ((p.W1 + 15) / 16) * ((p.H1 + TH - 1) / TH) * batch : (p.M + BM - 1) / BM;
  const long total = (long)p.tiles_m * p.tiles_n * 4;
  if (total > 0x7fffffffL) return fail(CCVPE_EINVAL, "upconv: grid too large");
  p.tiles_total = (int)total;
  if (halo)
    hipLaunchKernelGGL((upconv_halo_kernel<T, MT, NT, WN>), dim3(p.tiles_total), dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL((upconv_kernel<T, MT, NT, WN>), dim3(p.tiles_total), dim3(256), 0, stream, p);
  return check_launch("upconv_kernel");
}

// Pick the N tile that wastes the fewest MFMA columns, then the widest.
struct TileCfg { int mt, nt, wn; };
static const TileCfg kCfgs[] = {
    {4, 5, 2}, {4, 4, 2}, {4, 3, 2}, {4, 2, 2}, {4, 1, 2},  // BN 160,128,96,64,32  BM 128
    {4, 5, 1}, {4, 3, 1}, {4, 1, 1},                        // BN 80,48,16          BM 256
    {2, 7, 1},                                              // BN 112               BM 128
};

static int pick_cfg(int npad16) {
  int best = 0;
  long best_cost = -1;
  for (int i = 0; i < (int)(sizeof(kCfgs) / sizeof(kCfgs[0])); ++i) {
    const int bn = 16 * kCfgs[i].nt * kCfgs[i].wn;
    const int tiles = (npad16 + bn - 1) / bn;
    const long cost = (long)tiles * bn * 1000 + (1000 - bn);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = i; }
  }
  return best;
}

}  // namespace ccvpe

using namespace ccvpe;

// CCVPE_PW_GEMM=0 routes the pointwise convs through the generic kernel again (A/B measurements)
static const bool g_use_pw = !(getenv("CCVPE_PW_GEMM") && getenv("CCVPE_PW_GEMM")[0] == '0');

template <typename T>
static int conv_igemm_any(const ccvpe_conv_desc* d, void* stream, int out_f32, float* scratch = nullptr,
                          long* want_floats = nullptr) {
  constexpr int E = ElemTraits<T>::E;
  constexpr int SK = 4 * E;
  if (!d) return fail(CCVPE_EINVAL, "conv_igemm: null desc");
  if (d->c0 <= 0 || d->c0 % 8 || d->c1 < 0 || d->c1 % 8)
    return fail(CCVPE_EINVAL, "conv_igemm: c0/c1 must be multiples of 8 (got %d,%d)", d->c0, d->c1);
  if (d->c1 > 0 && !d->src1) return fail(CCVPE_EINVAL, "conv_igemm: c1>0 but src1 null");
  if (d->ld0 % E || (d->c1 && d->ld1 % E) || d->kpad % SK || d->ldd % 4 || (d->residual && d->ldres % 4))
    return fail(CCVPE_EINVAL, "conv_igemm: ld0/ld1 %% %d, kpad %% %d, ldd/ldres %% 4 required", E, SK);
  if (!aligned16(d->src0) || (d->src1 && !aligned16(d->src1)) || !aligned16(d->w) ||
      (d->gate && !aligned16(d->gate)) || !aligned16(d->dst) || (d->residual && !aligned16(d->residual)))
    return fail(CCVPE_EINVAL, "conv_igemm: pointers must be 16-byte aligned");
  if (d->gate && (d->kh != 1 || d->c1 != 0)) return fail(CCVPE_EINVAL, "conv_igemm: gate only for 1x1 single-source");
  if (d->stride < 1 || d->kh < 1 || d->kw < 1) return fail(CCVPE_EINVAL, "conv_igemm: bad kernel/stride");
  if (d->act == CCVPE_ACT_RELU_MASK && (!d->residual || d->out_mode != CCVPE_OUT_NHWC))
    return fail(CCVPE_EINVAL, "conv_igemm: CCVPE_ACT_RELU_MASK needs the ReLU output in `residual` and an NHWC store");
  IgemmParams p;
  p.src0 = d->src0; p.src1 = d->src1; p.gate = d->gate; p.w = d->w;
  p.scale = d->scale; p.shift = d->shift; p.residual = d->residual; p.dst = d->dst;
  p.out_f32 = out_f32;
  p.c0 = d->c0; p.ld0 = d->ld0; p.c1 = d->c1; p.ld1 = d->ld1;
  p.H = d->in_h; p.W = d->in_w;
  p.in_pixels = (long)d->batch * d->in_h * d->in_w;
  p.Ho = (d->in_h + 2 * d->pad - d->kh) / d->stride + 1;
  p.Wo = (d->in_w + 2 * d->pad - d->kw) / d->stride + 1;
  p.kw = d->kw; p.stride = d->stride; p.pad = d->pad;
  p.N = d->n; p.Kpad = d->kpad; p.Npad = (d->n + 15) / 16 * 16;
  p.cpt0 = d->c0 / 8; p.cpt = (d->c0 + d->c1) / 8;
  p.total_chunks = p.cpt * d->kh * d->kw;
  if (p.total_chunks * 8 > p.Kpad) return fail(CCVPE_EINVAL, "conv_igemm: kpad %d < K %d", p.Kpad, p.total_chunks * 8);
  constexpr int CPS = SK / 8;
  p.stages = (p.total_chunks + CPS - 1) / CPS;
  p.ldd = d->ldd; p.ldres = d->ldres; p.act = d->act; p.out_mode = d->out_mode;
  p.cout = (d->out_mode == CCVPE_OUT_DECONV2X) ? d->n / 4 : d->n;
  if (d->out_mode == CCVPE_OUT_DECONV2X && (d->n % 16 || d->residual))
    return fail(CCVPE_EINVAL, "conv_igemm: deconv mode needs cout%%4==0 and no residual");
  const long M = (long)d->batch * p.Ho * p.Wo;
  if (M <= 0 || M > 0x7fffffffL) return fail(CCVPE_EINVAL, "conv_igemm: bad M");
  if ((long)d->batch * d->in_h * d->in_w > 0x7fffffffL) return fail(CCVPE_EINVAL, "conv_igemm: too many pixels");
  p.M = (int)M;
  p.tiles_n = p.tiles_total = p.tiles_x = p.tiles_y = 0;
  hipStream_t st = (hipStream_t)stream;
  const TileCfg c = kCfgs[pick_cfg(p.Npad)];
  const bool is3x3 = d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 &&
                     d->out_mode == CCVPE_OUT_NHWC && !d->gate;
  // split-K mode (planning or with scratch): everything, 3x3 included, goes through the generic gather kernel
  const bool sk = scratch != nullptr || want_floats != nullptr;
  if (scratch && !aligned16(scratch)) return fail(CCVPE_EINVAL, "conv_igemm: scratch must be 16-byte aligned");
  // pointwise convs (1x1, stride 1, one source, plain NHWC store) take the deep-stage kernel; the residual / bf16 rows
  // it touches with 16-byte accesses must be 16-byte aligned along the channel axis
  const int esz = (int)sizeof(T);
  const bool out32 = esz == 4 || out_f32;
  const bool is_pw = g_use_pw && d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0 && d->c1 == 0 && d->act != CCVPE_ACT_RELU &&
                     d->act != CCVPE_ACT_RELU_MASK &&
                     d->n > 48 &&      // narrow outputs (N <= 48: 16-48 column tiles) stay with the generic kernel
                     d->out_mode == CCVPE_OUT_NHWC && (d->ldd * (out32 ? 4 : 2)) % 16 == 0 &&
                     (!d->residual || (d->ldres * esz) % 16 == 0);
  // the 256 x 80 tile needs more than 256 VGPRs in the persistent pointwise kernel (staging registers live across the
  // epilogue): N = 65..80 pointwise layers take the 128 x 96 tile there
  if (is_pw && !sk && c.mt == 4 && c.nt == 5 && c.wn == 1) return launch_pw<T, 4, 3, 2>(p, st);
#define CCVPE_CASE(MT_, NT_, WN_)                                                  \
  if (c.mt == MT_ && c.nt == NT_ && c.wn == WN_) {                                 \
    if (sk) return launch<T, MT_, NT_, WN_>(p, st, scratch, want_floats);          \
    if (is3x3) return launch3x3<T, MT_, NT_, WN_>(p, d->batch, st);                \
    if constexpr (16 * NT_ * WN_ > 48 && !(MT_ == 4 && NT_ == 5 && WN_ == 1))     \
      if (is_pw) return launch_pw<T, MT_, NT_, WN_>(p, st);                        \
    return launch<T, MT_, NT_, WN_>(p, st);                                        \
  }
  CCVPE_CASE(4, 5, 2) CCVPE_CASE(4, 4, 2) CCVPE_CASE(4, 3, 2) CCVPE_CASE(4, 2, 2) CCVPE_CASE(4, 1, 2)
  CCVPE_CASE(4, 5, 1) CCVPE_CASE(4, 3, 1) CCVPE_CASE(4, 1, 1) CCVPE_CASE(2, 7, 1)
#undef CCVPE_CASE
  return fail(CCVPE_EINVAL, "conv_igemm: no tile config");
}

template <typename T>
static int upconv_any(const ccvpe_upconv_desc* d, void* stream) {
  constexpr int E = ElemTraits<T>::E;
  constexpr int SK = 4 * E;
  constexpr int CPS = SK / 8;
  if (!d) return fail(CCVPE_EINVAL, "upconv: null desc");
  if (d->c0 <= 0 || d->c0 % 8 || d->c1 < 0 || d->c1 % 8) return fail(CCVPE_EINVAL, "upconv: c0/c1 must be multiples of 8");
  if (d->c1 > 0 && !d->src1) return fail(CCVPE_EINVAL, "upconv: c1>0 but src1 null");
  if (d->ld0 % E || (d->c1 && d->ld1 % E) || d->kpad % SK || d->ldd % 4) return fail(CCVPE_EINVAL, "upconv: bad strides");
  if (!aligned16(d->src0) || (d->src1 && !aligned16(d->src1)) || !aligned16(d->w) || !aligned16(d->dst))
    return fail(CCVPE_EINVAL, "upconv: pointers must be 16-byte aligned");
  if (!d->shift9) return fail(CCVPE_EINVAL, "upconv: shift9 required");
  UpParams p;
  p.src0 = d->src0; p.src1 = d->src1; p.w = d->w; p.shift9 = d->shift9; p.dst = d->dst;
  p.out_f32 = sizeof(T) == 4;
  p.c0 = d->c0; p.ld0 = d->ld0; p.c1 = d->c1; p.ld1 = d->ld1;
  p.H1 = d->h1; p.W1 = d->w1;
  p.N = d->n; p.Kpad = d->kpad; p.Npad = (d->n + 15) / 16 * 16;
  p.cpt0 = d->c0 / 8; p.cpt1 = d->c1 / 8;
  p.total_chunks = 4 * p.cpt0 + 9 * p.cpt1;
  if (p.total_chunks * 8 > p.Kpad) return fail(CCVPE_EINVAL, "upconv: kpad %d < K %d", p.Kpad, p.total_chunks * 8);
  p.stages = (p.total_chunks + CPS - 1) / CPS;
  p.ldd = d->ldd; p.act = d->act;
  const long M = (long)d->batch * d->h1 * d->w1;
  if (M <= 0 || 4 * M > 0x7fffffffL) return fail(CCVPE_EINVAL, "upconv: bad M");
  p.M = (int)M;
  p.tiles_n = p.tiles_m = p.tiles_total = 0;
  hipStream_t st = (hipStream_t)stream;
  const TileCfg c = kCfgs[pick_cfg(p.Npad)];
#define CCVPE_CASE(MT_, NT_, WN_) \
  if (c.mt == MT_ && c.nt == NT_ && c.wn == WN_) return launch_up<T, MT_, NT_, WN_>(p, st);
  CCVPE_CASE(4, 5, 2) CCVPE_CASE(4, 4, 2) CCVPE_CASE(4, 3, 2) CCVPE_CASE(4, 2, 2) CCVPE_CASE(4, 1, 2)
  CCVPE_CASE(4, 5, 1) CCVPE_CASE(4, 3, 1) CCVPE_CASE(4, 1, 1) CCVPE_CASE(2, 7, 1)
#undef CCVPE_CASE
  return fail(CCVPE_EINVAL, "upconv: no tile config");
}

extern "C" int ccvpe_upconv3x3_f32(const ccvpe_upconv_desc* d, void* stream) { return upconv_any<float>(d, stream); }
extern "C" int ccvpe_upconv3x3_bf16(const ccvpe_upconv_desc* d, void* stream) { return upconv_any<bf16_t>(d, stream); }

extern "C" int ccvpe_conv_igemm_splitk_floats(const ccvpe_conv_desc* d, int is_bf16) {
  long want = 0;
  const int rc = is_bf16 ? conv_igemm_any<bf16_t>(d, nullptr, 0, nullptr, &want)
                         : conv_igemm_any<float>(d, nullptr, 0, nullptr, &want);
  if (want > 0x7fffffffL) want = 0;          // would not fit the int return: do not split
  return rc ? rc : (int)want;
}
extern "C" int ccvpe_conv_igemm_splitk_bf16(const ccvpe_conv_desc* d, int out_f32, float* scratch, void* stream) {
  if (!scratch) return fail(CCVPE_EINVAL, "conv_igemm_splitk: scratch is NULL");
  return conv_igemm_any<bf16_t>(d, stream, out_f32, scratch);
}
extern "C" int ccvpe_conv_igemm_splitk_f32(const ccvpe_conv_desc* d, float* scratch, void* stream) {
  if (!scratch) return fail(CCVPE_EINVAL, "conv_igemm_splitk: scratch is NULL");
  return conv_igemm_any<float>(d, stream, 0, scratch);
}

extern "C" int ccvpe_conv_igemm_f32(const ccvpe_conv_desc* d, void* stream) {
  return conv_igemm_any<float>(d, stream, 1);
}

extern "C" int ccvpe_conv_igemm_bf16(const ccvpe_conv_desc* d, int out_f32, void* stream) {
  return conv_igemm_any<bf16_t>(d, stream, out_f32 ? 1 : 0);
}
