// Shared machinery of the implicit-GEMM convolution kernels (conv_igemm.hip, conv_pw_*.hip, conv3x3_*.hip, upconv_*.hip):
// element traits, MFMA stage, staging helpers, IgemmParams, the 4-channel epilogue store, tile table.
//
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
#pragma once
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
  int ablate;        // diagnostics builds only (-DCCVPE_ABLATE): read nowhere in the product build
};

constexpr int LDS_LD = 20;  // floats per staged row (16 + 4 pad); see DESIGN section 4 "LDS bank conflicts" for the measured pitch-24 variant
// Staged panels WITHOUT padding (round 4) — pitch 16 floats = one 64-byte K chunk per row — and the 16-byte piece index XOR-ed by
// panel_swz(row): conflict-free ds_read_b128 fragments for any window base (tools/lds_layout.py, tests/test_lds_layout.py) and 20 % less
// LDS than pitch 20 (LDS_LD, still used by the folded-deconv halo), whose every fragment read is a 2-way conflict.  The bf16 matrix
// phases are 8x shorter than the fp32 ones for the same fragment bytes, so the LDS arrays are what the bf16 kernels wait for (3x3: +7-12 %
// on isolated layers); fp32 gains 0.7 % on the whole forward (33.66 -> 33.44 ms, tools/gpu/ab_f32.sh).  Fragment rows are
// 16j + (lane & 15) and staging rows (tid >> 2) + 64 it, so the swizzle term is a per-lane constant in both.
template <typename T> struct PanelLayout {
  static constexpr bool SWZ = true;
  static constexpr int LD = SWZ ? 16 : LDS_LD;
};
__device__ __forceinline__ int panel_swz(int row) { return ((row >> 2) & 1) << 1; }

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

// W-stage swizzle of the LDS-DMA kernels (conv3x3_impl.h): slot(r, c) = 4r + (c ^ perm[(r>>2)&3]), perm = (0,2,3,1)
__device__ __forceinline__ int w_swz(int r) { return (0x1320 >> (((r >> 2) & 3) * 4)) & 3; }

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


// Entry points of the per-kernel translation units (one per element type so that they compile in parallel).  `mt, nt, wn`
// is a row of kCfgs; the functions return CCVPE_EINVAL for a tile they do not instantiate.
template <typename T> int pw_dispatch(const IgemmParams& p, int mt, int nt, int wn, hipStream_t stream);        // conv_pw_*.hip
template <typename T> int conv3x3_dispatch(const IgemmParams& p, int batch, int mt, int nt, int wn, hipStream_t stream);   // conv3x3_*.hip
// conv_pw2_impl.h (conv_pw2_f32.hip / conv_pw2_bf16.hip): the pointwise GEMM with an LDS-DMA ring, for the shapes pw2_supported() accepts
extern bool g_use_pw2;                                    // ccvpe_set_pw_ring_kernels (conv_igemm.hip)
template <typename T> bool pw2_supported(const IgemmParams& p, int mt, int nt, int wn);
template <typename T> int pw2_dispatch(const IgemmParams& p, int mt, int nt, int wn, hipStream_t stream);
int num_cus();                                            // narrow_bf16.hip
// narrow_bf16.hip: bf16 3x3 layers with few channels on large images — weights resident in registers, persistent workgroups
extern bool g_use_narrow;                                 // ccvpe_set_narrow_kernels (conv_igemm.hip)
int c3n_supported(const IgemmParams& p, int batch);      // 0 = not served, else an instantiation id
int c3n_dispatch(const IgemmParams& p, int batch, hipStream_t stream);
int c3n_match_supported(const IgemmParams& p, int batch, int L);
int c3n_match_dispatch(const IgemmParams& p, int batch, const float* g, int ldg, int L, int off, float* scores, hipStream_t stream);
// pwn.hip: 1x1 projections with N <= 48 and K <= 256 (fp32: <= 144) on large planes — weights (x SE gate) resident in registers, waves stream tiles
extern bool g_use_pwn;                                    // ccvpe_set_pwn_kernels
bool pwn_supported(const IgemmParams& p, int batch, int esz);
int pwn_dispatch(const IgemmParams& p, int batch, int esz, hipStream_t stream);
int up2_supported(int c0, int c1, int n, int kpad, int h1, int w1, int batch);
int up2_dispatch(const void* src0, const void* src1, const void* w, const float* shift9, void* dst, int c0, int ld0, int c1, int ld1,
                 int h1, int w1, int n, int kpad, int ldd, int act, int batch, hipStream_t stream);

}  // namespace ccvpe
