// pw_gemm_kernel: see the comment block below.  Included by conv_pw_f32.hip / conv_pw_bf16.hip (one explicit instantiation each).
#pragma once
#include "conv_common.h"

namespace ccvpe {

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
  // floats per staged row: no padding, the 8 pieces of a row XOR-swizzled by (row >> 1) & 7 — conflict-free fragment
  // reads (tools/lds_layout.py; pitch 36 makes every ds_read_b128 a 2-way conflict), see PanelLayout in conv_common.h
  static constexpr bool SWZ = true;
  static constexpr int LDF = SWZ ? 16 * KP : 16 * KP + 4;
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
  constexpr bool SWZ = G::SWZ;
  const int pcs = SWZ ? (pc ^ ((prow >> 1) & 7)) : pc;                  // staged piece -> (swizzled) slot; rows advance by 32
  int fcol[KP];                                                          // fragment float column of 64-byte piece kp
#pragma unroll
  for (int kp = 0; kp < KP; ++kp) fcol[kp] = SWZ ? (((kp * 4 + (lane >> 4)) ^ ((frow >> 1) & 7)) * 4) : kp * 16 + fk;
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
      *reinterpret_cast<f32x4*>(&As[(prow + it * (256 / PPR)) * LDF + (SWZ ? pcs : pc) * 4]) = keep_if(v, k_ok && ((row_ok >> it) & 1u));
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int nrow = prow + it * (256 / PPR);
      if (nrow < BN) *reinterpret_cast<f32x4*>(&Bs[nrow * LDF + (SWZ ? pcs : pc) * 4]) = b_reg[it];
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
#ifdef CCVPE_ABLATE
  if ((p.ablate >> 8) > 0 && blockIdx.x >= (gridDim.x >> 1)) {          // experiment: de-phase the two workgroups of a CU
    for (int i = 0; i < (p.ablate >> 8); ++i) __builtin_amdgcn_s_sleep(16);   // ~1024 cycles each
  }
#endif
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
#ifdef CCVPE_ABLATE
      if (p.ablate & 2) {
      } else
#endif
      if (more) load_stage(m0, n0, s + 1);
      else if (PREFETCH && has_next) {                   // next tile's first stage: lands during the epilogue below
        load_stage(m0n, n0n, 0);
        load_ss(n0n);
      }
#pragma unroll
      for (int kp = 0; kp < KP; ++kp) {
#ifdef CCVPE_ABLATE
        if (p.ablate & 4) break;
#endif
        if (s * KS + kp * 4 * E >= p.c0) break;   // K tail: whole 64-byte pieces beyond K are zero (uniform branch)
        f32x4 af[MT], bf[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
          af[i] = *reinterpret_cast<const f32x4*>(&As[((wm * MT + i) * 16 + frow) * LDF + (SWZ ? fcol[kp] : kp * 16 + fk)]);
#pragma unroll
        for (int j = 0; j < NT; ++j)
          bf[j] = *reinterpret_cast<const f32x4*>(&Bs[((wn * NT + j) * 16 + frow) * LDF + (SWZ ? fcol[kp] : kp * 16 + fk)]);
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
#ifdef CCVPE_ABLATE
          if (!(p.ablate & 8))
#endif
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
#ifdef CCVPE_ABLATE
        const bool st_on = !(p.ablate & 1);
#else
        constexpr bool st_on = true;
#endif
        if (pcol < PPO && n < p.N && st_on) {
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
#ifdef CCVPE_ABLATE   // diagnostics build only: 1 = no global stores, 2 = no stage loads after the first, 4 = no MFMAs, 8 = no swish
  static const int ablate = getenv("CCVPE_PW_ABLATE") ? atoi(getenv("CCVPE_PW_ABLATE")) : 0;
  static const int stagger = getenv("CCVPE_PW_STAGGER") ? atoi(getenv("CCVPE_PW_STAGGER")) : 0;   // x 1024 cycles
  p.ablate = ablate | (stagger << 8);
#endif
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

template <typename T>
int pw_dispatch(const IgemmParams& p, int mt, int nt, int wn, hipStream_t stream) {
#define CCVPE_CASE(MT_, NT_, WN_) \
  if (mt == MT_ && nt == NT_ && wn == WN_) return launch_pw<T, MT_, NT_, WN_>(p, stream);
  // tiles wider than 48 columns only (conv_igemm_any routes N <= 48 to the generic kernel); the 256 x 80 tile is re-routed
  // to 128 x 96 by the caller (more than 256 VGPRs with the staging registers live across the epilogue)
  CCVPE_CASE(4, 5, 2) CCVPE_CASE(4, 4, 2) CCVPE_CASE(4, 3, 2) CCVPE_CASE(4, 2, 2) CCVPE_CASE(2, 7, 1)
#undef CCVPE_CASE
  return fail(CCVPE_EINVAL, "pw_gemm: no tile <%d,%d,%d>", mt, nt, wn);
}

}  // namespace ccvpe
