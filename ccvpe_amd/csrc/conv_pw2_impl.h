// pw2_kernel: the pointwise (1x1) GEMM of the EfficientNet expand / project / head convolutions with an LDS-DMA ring
// (efficientnet_pytorch/model.py:62,86,209: _expand_conv + _bn0 + swish, _project_conv + _bn2 (+ skip), _conv_head), fp32 and bf16
// storage (conv_pw2_f32.hip / conv_pw2_bf16.hip instantiate it).  Route CCVPE_ROUTE_PW_RING behind the unchanged
// ccvpe_conv_igemm_f32 / _bf16; ccvpe_set_pw_ring_kernels(0) brings pw_gemm_kernel back.  Designed on the fp32 layers:
//
// Why (round 5): pw_gemm_kernel (conv_pw_impl.h) stages ONE 32-float K stage ahead through registers into a single LDS buffer and
// turns every output tile through LDS behind two workgroup barriers.  On the encoder's fp32 layers (M = 16 384 / 65 536 pixels,
// K = 80 ... 1 152, N = 112 ... 1 280) it holds 44-49 % MFMA-busy (profiles/r05/mfma_busy_f32.json), and its time is the SUM of its
// matrix time and of everything else (tools/gpu/ablate_pw.sh, 112 -> 672 at 32 x 32 x 64: 135 us; no MFMAs 61 us; the MFMAs alone
// 63 us at the 157 TF fp32 matrix rate).  What the measurements of this file's development say about why, in the order found:
//   * the fp32 MFMA runs on the vector-fp32 pipe (MI355X_MICROARCH.md: "at the f32 VECTOR rate"; tools/micro/mfma_f32_valu_overlap.hip:
//     8 MFMAs + 32 v_fma per iteration take the SUM of the two alone, at one and at two waves per SIMD): every vector instruction of the
//     staging, the address arithmetic and the epilogue is matrix time lost — swish alone (two quarter-rate transcendentals per output)
//     is ~12 us of a 112 -> 672 layer;
//   * a bare MFMA loop of this shape (no LDS reads, no DMA, no epilogue) reaches 131 TF = 83 % of the peak, the conv3x3 kernels 86 %:
//     that is the practical ceiling, not 157;
//   * with 512 registers per lane on offer (__launch_bounds__(256, 1)) the compiler keeps the loop-carried accumulators in VGPRs and
//     the MFMA destinations in AGPRs and copies all of them both ways every 16 K (112 v_accvgpr moves per 56 MFMAs): the first version
//     of this kernel ran at 65 % of the matrix rate with nothing else in the loop.  __launch_bounds__(256, 2) removes the copies;
//   * a runtime `if (residual)` in the epilogue became unconditional adds behind a select with an s_waitcnt vmcnt in front of each —
//     in layers WITHOUT a skip every accumulator row waited for the DMA ring and the previous stores to drain (30 us of 131);
//   * 16-byte stores straight from the accumulators (64-byte fragments of 16 rows per instruction) cost 26 us of 128 against 8 us for
//     whole-row stores after a wave-private turn through LDS;
//   * ONE workgroup per CU (a four-stage ring in 150 KB) was SLOWER than pw_gemm_kernel whatever the ring depth: with one wave per SIMD
//     every barrier, LDS round trip and epilogue is dead matrix time.  De-phasing the persistent workgroups by a start delay changes
//     nothing (tried on both kernels).
// The kernel as it stands:
//   * two workgroups per CU (<= 80 KB each), a ring of three 16-float K stages per workgroup filled by global_load_lds_dwordx4 — no
//     staging registers, no ds_write pass, ONE barrier per stage, requests two stages ahead and across tile boundaries (persistent
//     workgroups: the next tile's first stages land under the current tile's last stages and epilogue);
//   * the LDS side of the DMA is lane-linear, so the panel swizzle (the four 16-byte pieces of a row at slots piece ^ (-(row >> 2) & 3):
//     conflict-free ds_read_b128 fragments) is produced by choosing WHICH global piece a lane requests;
//   * fragments of stage g + 1 are read into a second register set before the MFMAs of stage g are issued (stage loop unrolled by two);
//   * the SE gate (project convs) multiplies the A fragments (MT x 4 multiplies per 16 K against MT x NT x 4 MFMAs) from a per-tile copy
//     of the sample's gate vector in LDS; the BN vectors of the tile's channels reach LDS the same way — every per-tile vector is a DMA
//     request too, because a compiler-visible global load inside the loop makes the compiler wait for vmcnt(0), i.e. drain the ring;
//   * epilogue per 16-pixel row tile: BN (+ swish) in registers, a wave-private LDS patch (no workgroup barrier: a wave's LDS queue is
//     in order; the compiler is told with a wavefront-scope fence), 16-byte stores with consecutive lanes along a pixel row;
//   * same MFMA k-order and the same fp32 operations as pw_gemm_kernel / igemm_kernel: results are BIT-IDENTICAL (tests/test_ops_gpu.py).
//   * bf16 storage: a stage is 32 channels (ONE 16x16x32 MFMA per fragment pair), a last stage of 16 channels re-reads its lower
//     pieces on the activation side (finite data against the zero padding of the packed weights), the gate multiplies in fp32 and
//     rounds back as pw_gemm_kernel's staging does, 16-byte stores of 8 channels.
// Measured (isolated launches, B = 64, tools/pw_probe.py): 112 -> 672 @32^2 116 us against 138 (85 TF), 672 -> 112 100 / 109, 80 -> 480
// 63 / 79, 192 -> 1152 @16^2 80 / 88, 320 -> 1280 119 / 140; bf16: 39.5 / 47.1, 22.1 / 22.6, 26.0 / 32.1, 22.6 / 24.4, 28.2 / 31.7.  Inside the C1 forward, where the two encoders share the chip, the step time
// does not move measurably (32.68 -> 32.65 ms, tools/gpu/ab_pw_ring.sh): DESIGN.md section 4.
// Shapes: c0 % 16 == 0, M % BM == 0, N % BN == 0, gated tiles inside one sample, the three epilogue forms of the encoder; everything
// else stays with pw_gemm_kernel (pw2_supported).
#pragma once
#include "conv_common.h"
#include <type_traits>

namespace ccvpe {

__device__ __forceinline__ void pw2_dma(unsigned lds, unsigned voff, const char* sbase) {
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(voff), "s"(sbase) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void pw2_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// lanes of ONE wave exchange data through LDS: the hardware keeps a wave's LDS operations in order, the compiler must too (without
// the fence it may move a lane's reads above the same lane's writes — other addresses, as far as a single thread can tell)
__device__ __forceinline__ void pw2_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int PW2_LDS = 80 * 1024;                                    // two workgroups per CU
constexpr int PW2_NSTG = 3;                                          // ring stages (two in flight while one is consumed)
constexpr int PW2_RAW = -1;                                          // "activation" of the raw form: no BN vectors, plain store (train mode, 1x1 input gradients)
constexpr int PW2_SS = 256;                                          // floats reserved per scale / shift vector (BN <= 160)
// floats of one tile's vectors in LDS: the sample's gate vector (whole 256-float requests) + scale + shift
static inline __host__ __device__ int pw2_aux_stride(bool gated, int c0) { return (gated ? (c0 + 255) / 256 * 256 : 0) + 2 * PW2_SS; }

template <int MT, int NT, int WN, int NSTG_>
struct Pw2Geom {
  static constexpr int WM = 4 / WN;
  static constexpr int BM = 16 * MT * WM;
  static constexpr int BN = 16 * NT * WN;
  static constexpr int KS = 16;                                     // FLOATS (= 64 bytes) of K per row per stage: 16 fp32 / 32 bf16 channels
  static constexpr int STAGE_FLOATS = (BM + BN) * KS;
  static constexpr int NA = BM / 16, NB = BN / 16, ND = NA + NB;    // DMA requests (16 rows x 64 bytes each) per stage
  static constexpr int DPW = (ND + 3) / 4;                          // ... per wave (request j belongs to wave j % 4)
  static constexpr int DMIN = ND / 4;                               // the fewest any wave issues per stage: the unit of the waits
  static constexpr int NSTG = NSTG_;
  static constexpr int RING_BYTES = NSTG * STAGE_FLOATS * 4;
  // epilogue: per wave a private LDS patch of 16 pixel rows x (half of the wave's column tiles, 16 NH + 4 floats)
  static constexpr int NH = (NT + 1) / 2, EPP = 16 * NH + 4, EP_FLOATS = 16 * EPP;
  static int lds_bytes(bool gated, int c0) { return RING_BYTES + 4 * EP_FLOATS * 4 + 2 * pw2_aux_stride(gated, c0) * 4; }
  static_assert((NSTG - 2) * DMIN <= 63, "vmcnt immediate");
};

// (__launch_bounds__(256, 2) is not only the occupancy this kernel runs at: with 512 registers per lane on offer — (256, 1), tried with
// the one-workgroup-per-CU ring — the compiler keeps the loop-carried accumulators in VGPRs, the MFMA destinations in AGPRs and copies
// all of them both ways around every 16 K: 112 v_accvgpr moves + a drained matrix pipe per 56 MFMAs, 65 % of the matrix rate with
// nothing else in the loop)
template <typename T, int MT, int NT, int WN, int ACT, bool GATED, bool RES, int NSTG_>
__global__ __launch_bounds__(256, 2) void pw2_kernel(const IgemmParams p) {
  using G = Pw2Geom<MT, NT, WN, NSTG_>;
  constexpr bool RAW = ACT == PW2_RAW;
  constexpr int E = ElemTraits<T>::E;                    // elements per 16 bytes
  constexpr int KE = 4 * E;                              // K elements per stage
  constexpr int ES = (int)sizeof(T);
#ifdef CCVPE_ABLATE   // diagnostics build (tools/gpu/ablate_pw.sh): 1 no global stores, 2 no epilogue, 4 no MFMAs, 8 no DMA after the prologue, 16 no waits / ring barriers, 32 no fragment reads
  const int abl = p.ablate;
#else
  constexpr int abl = 0;
#endif
  constexpr int WM = G::WM, BM = G::BM, BN = G::BN, KS = G::KS, NSTG = G::NSTG;
  constexpr int NA = G::NA, ND = G::ND, DPW = G::DPW, DMIN = G::DMIN;
  extern __shared__ __attribute__((aligned(16))) float pw2_sm[];
  float* ring = pw2_sm;                                  // [NSTG][BM + BN rows][16 floats = 64 bytes, pieces swizzled]
  float* aux = pw2_sm + NSTG * G::STAGE_FLOATS;          // [2 tile parities][gate (K up to whole requests) | scale 256 | shift 256]
  float* ep = aux + 2 * pw2_aux_stride(GATED, p.c0) + ((int)threadIdx.x >> 6) * G::EP_FLOATS;   // this wave's epilogue patch
  const int AUX_STRIDE = pw2_aux_stride(GATED, p.c0), SS_OFF = AUX_STRIDE - 2 * PW2_SS;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)pw2_sm;
  const unsigned aux0 = lds0 + (unsigned)(NSTG * G::STAGE_FLOATS * 4);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int frow = lane & 15;
  // panel layout: row r holds its four 16-byte pieces at slots piece ^ swz(r), swz(r) = (-(r >> 2)) & 3: the 16 lanes of every
  // ds_read_b128 lane group ({0-3, 12-15, 20-27}, ...: rows 0-3 and 12-15 of one piece, rows 4-11 of the next) then cover the 16
  // slots of a 256-byte bank row exactly once
  const int fcol = ((lane >> 4) ^ ((0 - (frow >> 2)) & 3)) * 4;
  const int epix = lane & 15, en = (lane >> 4) * 4;
  const int hw = p.Ho * p.Wo;
  const int nst = (p.c0 + KE - 1) / KE;
  // bf16: the last stage may hold 16 of its 32 channels (c0 % 32 == 16): its upper two pieces re-read the lower two (finite data
  // against the zero padding of the packed weights; never an address beyond the tensors)
  const bool last_half = (p.c0 % KE) != 0;
  const T* res = reinterpret_cast<const T*>(p.residual);

  // ---- DMA lane geometry: request j (wave j % 4) covers rows 16 j' .. 16 j' + 15 of the A panel (j < NA) or of the W panel; lane ->
  // (row 16 j' + lane / 4, LDS slot lane % 4) fetches global piece slot ^ swz(row) of its row ---------------------------------------
  unsigned voff[DPW], voffh[ES == 2 ? DPW : 1];          // byte offsets from the stage's scalar base (whole stage / half stage)
#pragma unroll
  for (int i = 0; i < DPW; ++i) {
    const int j = wave + 4 * i;
    const bool isA = j < NA;
    const int r = 16 * (isA ? j : j - NA) + (lane >> 2);
    const int c = (lane & 3) ^ ((0 - (r >> 2)) & 3);
    const unsigned ld = (unsigned)(isA ? p.ld0 : p.Kpad);
    voff[i] = (unsigned)r * ld * (unsigned)ES + (unsigned)c * 16u;
    if (ES == 2) voffh[i] = (unsigned)r * ld * (unsigned)ES + (unsigned)(c & 1) * 16u;
  }
  const char* a0 = reinterpret_cast<const char*>(p.src0);
  const char* w0 = reinterpret_cast<const char*>(p.w);

  // ---- this workgroup's tiles: v = blockIdx.x + k gridDim.x (gridDim.x a multiple of 8: one XCD, a contiguous run of tiles) ------
  const int gsz = gridDim.x;
  const int ntile_wg = (p.tiles_total - (int)blockIdx.x + gsz - 1) / gsz;
  if (ntile_wg <= 0) return;
  const int total_stages = ntile_wg * nst;

  // issue side: stage counter gi -> (tile ti, stage si)
  int gi = 0, ti = 0, si = 0, mi0 = 0, ni0 = 0;
  auto issue_tile_coords = [&]() {
    const int t = xcd_tile((int)blockIdx.x + ti * gsz, p.tiles_total);
    mi0 = (t / p.tiles_n) * BM;
    ni0 = (t % p.tiles_n) * BN;
  };
  issue_tile_coords();
  auto issue = [&]() {                                   // DMA requests of stage gi into slot gi % NSTG
    const unsigned slot = lds0 + (unsigned)((gi % NSTG) * G::STAGE_FLOATS * 4);
    const char* ab = a0 + ((size_t)mi0 * p.ld0 + (size_t)si * KE) * ES;
    const char* wb = w0 + ((size_t)ni0 * p.Kpad + (size_t)si * KE) * ES;
    const bool half = ES == 2 && last_half && si == nst - 1;
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      const int j = wave + 4 * i;
      // (half stage: only the ACTIVATION rows re-read their lower pieces; the packed weights hold zeros beyond c0, inside Kpad)
      if (j < ND) pw2_dma(slot + (unsigned)(j * 1024), (ES == 2 && half && j < NA) ? voffh[ES == 2 ? i : 0] : voff[i], j < NA ? ab : wb);
    }
    if (si == 0 && wave == 3) {                          // the tile's vectors (wave 3 has the fewest panel requests)
      const unsigned ax = aux0 + (unsigned)((ti & 1) * AUX_STRIDE * 4);
      if (GATED) {
        const char* gb = reinterpret_cast<const char*>(p.gate + (size_t)(mi0 / hw) * p.c0);
        for (int c = 0; c < p.c0; c += 256) {            // 64 lanes x 4 floats per request; the last one re-reads the vector's tail
          const int k = min(c + lane * 4, p.c0 - 4);
          pw2_dma(ax + (unsigned)(c * 4), (unsigned)(k * 4), gb);
        }
      }
      if (!RAW && lane < BN / 4) {
        pw2_dma(ax + (unsigned)(SS_OFF * 4), (unsigned)(lane * 16), reinterpret_cast<const char*>(p.scale + ni0));
        pw2_dma(ax + (unsigned)((SS_OFF + PW2_SS) * 4), (unsigned)(lane * 16), reinterpret_cast<const char*>(p.shift + ni0));
      }
    }
    ++gi;
    if (++si == nst) {
      si = 0;
      ++ti;
      if (ti < ntile_wg) issue_tile_coords();
    }
  };
#pragma unroll 1
  for (int k = 0; k < NSTG && gi < total_stages; ++k) issue();

  // read side (one stage ahead of the MFMAs): stage gr -> (tile tr, stage sr).  The fragments of stage g + 1 are requested from LDS
  // BEFORE the MFMAs of stage g are issued (two register sets, the stage loop unrolled by two): every wave of the CU asks for its 9-12
  // fragments at the same moment, and with the reads right in front of their MFMAs all 8 waves then sat out the LDS queue together
  // (the bare read + MFMA loop ran at 77 % of the matrix rate).
  struct Frag { f32x4 a[MT], b[NT], g, g1; };              // g, g1: the SE gate of this lane's 4 (fp32) / 8 (bf16) channels of the stage
  Frag F0, F1;
  int gr = 0, tr = 0, sr = 0;
  auto read_frag = [&](Frag& F) {
    if (abl & 32) return;                                // (diagnostics: MFMAs on whatever the registers hold)
    const float* As = ring + (gr % NSTG) * G::STAGE_FLOATS;
    const float* Bs = As + BM * KS;
#pragma unroll
    for (int i = 0; i < MT; ++i) F.a[i] = *reinterpret_cast<const f32x4*>(&As[((wm * MT + i) * 16 + frow) * KS + fcol]);
#pragma unroll
    for (int j = 0; j < NT; ++j) F.b[j] = *reinterpret_cast<const f32x4*>(&Bs[((wn * NT + j) * 16 + frow) * KS + fcol]);
    if (GATED) {
      const float* gl = aux + (tr & 1) * AUX_STRIDE + sr * KE + (lane >> 4) * E;
      F.g = *reinterpret_cast<const f32x4*>(gl);
      if (ES == 2) F.g1 = *reinterpret_cast<const f32x4*>(gl + 4);
    }
    ++gr;
    if (++sr == nst) { sr = 0; ++tr; }
  };

  int m0 = 0, n0 = 0;
  f32x4 acc[MT][NT];
  int tc = 0, sc_ = 0;                                   // compute side: tile, stage
  // stage 0: at most the two later stages of this wave's requests may still be in flight (requests complete in order)
  if (total_stages >= 3) pw2_wait<2 * DMIN>();
  else if (total_stages == 2) pw2_wait<DMIN>();
  else pw2_wait<0>();
  __syncthreads();
  read_frag(F0);

  auto step = [&](Frag& Fc, Frag& Fn, int g) {
    if (sc_ == 0) {
      const int t = xcd_tile((int)blockIdx.x + tc * gsz, p.tiles_total);
      m0 = (t / p.tiles_n) * BM;
      n0 = (t % p.tiles_n) * BN;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (g + 1 < total_stages) {
      if (!(abl & 16)) {
        // this wave's LDS reads of stage g (requested a whole stage ago) have returned: its slot may be refilled after the barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (g + 2 < total_stages) pw2_wait<DMIN>();      // stage g + 1 has landed; stage g + 2 may be in flight
        else pw2_wait<0>();
        __syncthreads();                                 // stage g + 1 visible to every wave; every wave has read stage g
      }
      read_frag(Fn);
      if (gi < total_stages && !(abl & 8)) issue();      // stage g + 3 into the slot of stage g
    }
    if (!(abl & 4)) {
      if (GATED) {
#pragma unroll
        for (int i = 0; i < MT; ++i) Fc.a[i] = gate_apply<T>(Fc.a[i], Fc.g, Fc.g1);
      }
      if (ES == 4) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(Fc.b[j][kk], Fc.a[i][kk], acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = mfma_stage<T>(Fc.b[j], Fc.a[i], acc[i][j]);
      }
    }
    if (++sc_ < nst) return;
    sc_ = 0;
    // ---- epilogue: BN (+ swish) in registers, then each WAVE turns its 16-pixel x (16 NT)-channel row tiles through a private LDS
    // patch — no workgroup barrier, the LDS queue of a wave is in order — and stores whole row pieces: consecutive lanes write
    // consecutive 16 bytes of one pixel row (64 NT contiguous bytes per row).  Measured on 112 -> 672 at 32 x 32 x 64: pw_gemm_kernel's
    // workgroup-wide turn (two barriers per pass) 38 us of 141 with one wave per SIMD; 16-byte stores straight from the accumulators
    // (a lane holds 4 channels of one pixel: 64-byte fragments of 16 different rows per store) 26 us of 128 — all of it the stores,
    // the arithmetic costs nothing measurable.
    const float* Ss = aux + (tc & 1) * AUX_STRIDE + SS_OFF;
    ++tc;
    if (abl & 2) return;
    const float* Sl = Ss + wn * NT * 16 + en;             // this lane's channels of column tile 0 (BN vectors re-read per row tile:
                                                         // 2 NT registers of them held across the epilogue spill the fragment sets)
    // (wave-uniform 64-bit bases + 32-bit lane offsets: scalar-base stores / loads instead of a 64-bit address pair per access)
    char* dbase = reinterpret_cast<char*>(reinterpret_cast<T*>(p.dst) + (size_t)(m0 + wm * MT * 16) * p.ldd + n0 + wn * NT * 16);
    const char* rbase = RES ? reinterpret_cast<const char*>(res + (size_t)(m0 + wm * MT * 16) * p.ldres + n0 + wn * NT * 16) : nullptr;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {                      // the patch holds 16 pixel rows x half of the wave's column tiles
        constexpr int NH = G::NH;
        const int j0 = c * NH;
        const int nj = c == 0 ? NH : NT - NH;            // (compile-time after unrolling)
        if (nj <= 0) continue;
        pw2_wave_sync();                                 // the previous patch has been read back
#pragma unroll
        for (int j = 0; j < NH; ++j) {
          if (j >= nj) break;
          f32x4 vv = acc[i][j0 + j];
          if (!RAW) {
            const f32x4 scv = *reinterpret_cast<const f32x4*>(Sl + (j0 + j) * 16);
            const f32x4 shv = *reinterpret_cast<const f32x4*>(Sl + PW2_SS + (j0 + j) * 16);
            vv = vv * scv + shv;
          }
          if (ACT == CCVPE_ACT_SWISH) {
#pragma unroll
            for (int q = 0; q < 4; ++q) vv[q] = swishf(vv[q]);
          }
          *reinterpret_cast<f32x4*>(&ep[epix * G::EPP + j * 16 + en]) = vv;
        }
        pw2_wave_sync();
        // read back: 16 rows x (16 nj / E) output pieces of 16 bytes, consecutive lanes along a row
        // (RES is a template parameter: with a runtime `if (res)` the compiler made the adds unconditional behind a select and put an
        // s_waitcnt vmcnt before each of them — in the layers WITHOUT a skip that drained the DMA ring and the stores at every row tile)
#pragma unroll
        for (int k = 0; k < NH; ++k) {
          const int ppr = 16 * nj / E;                   // output pieces per row of this half
          if (64 * k >= 16 * ppr) break;
          const int q = lane + 64 * k;
          const int row = q / ppr, col = (q - row * ppr) * E;
          if (q < 16 * ppr) {
            f32x4 ov = *reinterpret_cast<const f32x4*>(&ep[row * G::EPP + col]);
            if (ES == 4) {
              if (RES) ov += *reinterpret_cast<const f32x4*>(rbase + (unsigned)(((i * 16 + row) * p.ldres + j0 * 16 + col) * 4));
              if (!(abl & 1)) *reinterpret_cast<f32x4*>(dbase + (unsigned)(((i * 16 + row) * p.ldd + j0 * 16 + col) * 4)) = ov;
            } else {
              f32x4 ov1 = *reinterpret_cast<const f32x4*>(&ep[row * G::EPP + col + 4]);
              if (RES) {                                 // bf16 residual: 8 channels = 16 bytes
                const bf16x8 rr = *reinterpret_cast<const bf16x8*>(rbase + (unsigned)(((i * 16 + row) * p.ldres + j0 * 16 + col) * 2));
#pragma unroll
                for (int e = 0; e < 4; ++e) { ov[e] += (float)rr[e]; ov1[e] += (float)rr[e + 4]; }
              }
              bf16x8 o8;
#pragma unroll
              for (int e = 0; e < 4; ++e) { o8[e] = (bf16_t)ov[e]; o8[e + 4] = (bf16_t)ov1[e]; }
              if (!(abl & 1)) *reinterpret_cast<bf16x8*>(dbase + (unsigned)(((i * 16 + row) * p.ldd + j0 * 16 + col) * 2)) = o8;
            }
          }
        }
      }
    }
  };
  for (int g = 0; g < total_stages; g += 2) {
    step(F0, F1, g);
    if (g + 1 < total_stages) step(F1, F0, g + 1);
  }
}

// tiles the pointwise router hands out; anything else stays with pw_gemm_kernel
static bool pw2_tile(int mt, int nt, int wn) {
  return (mt == 2 && nt == 7 && wn == 1) || (mt == 4 && nt == 5 && wn == 2) || (mt == 4 && nt == 4 && wn == 2) ||
         (mt == 4 && nt == 3 && wn == 2);
}

template <typename T>
bool pw2_supported(const IgemmParams& p, int mt, int nt, int wn) {
  constexpr int ES = (int)sizeof(T);
  if (!g_use_pw2 || !pw2_tile(mt, nt, wn)) return false;
  if (ES == 2 && p.out_f32) return false;                                        // bf16 storage writing fp32: pw_gemm_kernel
  const int bm = 16 * mt * (4 / wn), bn = 16 * nt * wn;
  const int hw = p.Ho * p.Wo;
  if (p.c0 % 16 || p.c0 < 64 || p.c0 > 1152 || p.c1 != 0) return false;          // whole 16-byte pieces; the gate vector's LDS copy
  // at least three K stages per tile: the requests run three stages ahead of the MFMAs, and the per-tile vectors (two LDS copies, by
  // tile parity) of tile t + 2 must not be requested before every wave is past the epilogue of tile t
  if ((p.c0 + 64 / ES - 1) / (64 / ES) < 3) return false;
  if (p.M % bm || p.N % bn || p.Npad % bn || p.N % 8 || p.Kpad % (64 / ES)) return false;
  if (p.gate && (hw % bm)) return false;                                          // a tile inside one sample: one gate vector
  const bool raw = !p.scale && !p.shift && p.act == CCVPE_ACT_NONE && !p.residual && ES == 4;      // train mode, 1x1 input gradients (fp32)
  if (!raw && (!p.scale || !p.shift)) return false;
  const bool expand = !p.gate && !p.residual && p.act == CCVPE_ACT_SWISH, project = p.gate && p.act == CCVPE_ACT_NONE;
  if (!expand && !project && !raw) return false;                                  // the instantiated epilogues (pw2_dispatch)
  if (project && p.residual && mt == 4 && nt == 5) return false;
  if (PW2_NSTG * (bm + bn) * 16 * 4 + 4 * 16 * (16 * ((nt + 1) / 2) + 4) * 4 + 2 * pw2_aux_stride(p.gate != nullptr, p.c0) * 4 > PW2_LDS) return false;
  if ((p.ldd * ES) % 16 || (p.residual && (p.ldres * ES) % 16) || (p.ld0 * ES) % 16) return false;
  if ((double)bm * p.ld0 * 4 >= 4294967296.0 || (double)bn * p.Kpad * 4 >= 4294967296.0) return false;
  if ((double)bm * p.ldd * 4 >= 2147483648.0 || (double)bm * p.ldres * 4 >= 2147483648.0) return false;
  return true;
}

template <typename T, int MT, int NT, int WN, int ACT, bool GATED, bool RES>
static int launch_pw2(const IgemmParams& p0, hipStream_t stream) {
  using G = Pw2Geom<MT, NT, WN, PW2_NSTG>;
  IgemmParams p = p0;
#ifdef CCVPE_ABLATE
  static const int ablate = getenv("CCVPE_PW2_ABLATE") ? atoi(getenv("CCVPE_PW2_ABLATE")) : 0;
  p.ablate = ablate;
#endif
  p.tiles_n = p.Npad / G::BN;
  p.tiles_total = (p.M / G::BM) * p.tiles_n;
  const int lds = G::lds_bytes(GATED, p.c0);
  if (lds > PW2_LDS) return fail(CCVPE_EINVAL, "pw2: tile <%d,%d,%d> does not fit the LDS", MT, NT, WN);
  static int attr_lds = 0;                                // per instantiation: the largest size asked for so far
  if (lds > attr_lds) {
    hipError_t e = hipFuncSetAttribute((const void*)pw2_kernel<T, MT, NT, WN, ACT, GATED, RES, PW2_NSTG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return fail(CCVPE_ELAUNCH, "pw2: set smem attr: %s", hipGetErrorString(e));
    attr_lds = lds;
  }
  const int cus = num_cus();
  const int grid = p.tiles_total < 2 * cus ? (p.tiles_total + 7) / 8 * 8 : 2 * cus / 8 * 8;
  hipLaunchKernelGGL((pw2_kernel<T, MT, NT, WN, ACT, GATED, RES, PW2_NSTG>), dim3(grid), dim3(256), lds, stream, p);
  return check_launch("pw2_kernel");
}

// the layer forms of the encoder: expand / head (BN + swish), project (SE gate, BN) without and with the skip; fp32 also the RAW
// forms of train mode (no BN vectors: the batch statistics come first) and of the 1x1 input gradients, plain or gated
template <typename T>
int pw2_dispatch(const IgemmParams& p, int mt, int nt, int wn, hipStream_t stream) {
#define CCVPE_PW2(MT_, NT_, WN_, SKIP_)                                                                           \
  if (mt == MT_ && nt == NT_ && wn == WN_) {                                                                      \
    if constexpr (sizeof(T) == 4) {                                                                               \
      if (!p.scale && !p.shift)                                                                                   \
        return p.gate ? launch_pw2<T, MT_, NT_, WN_, PW2_RAW, true, false>(p, stream) : launch_pw2<T, MT_, NT_, WN_, PW2_RAW, false, false>(p, stream); \
    }                                                                                                             \
    if (!p.gate && !p.residual && p.act == CCVPE_ACT_SWISH) return launch_pw2<T, MT_, NT_, WN_, CCVPE_ACT_SWISH, false, false>(p, stream); \
    if (p.gate && p.act == CCVPE_ACT_NONE && !p.residual) return launch_pw2<T, MT_, NT_, WN_, CCVPE_ACT_NONE, true, false>(p, stream);      \
    if constexpr (SKIP_) {                                                                                        \
      if (p.gate && p.act == CCVPE_ACT_NONE) return launch_pw2<T, MT_, NT_, WN_, CCVPE_ACT_NONE, true, true>(p, stream);                   \
    }                                                                                                             \
  }
  // (the 128 x 160 tile with gate AND skip needs more than 256 registers — no EfficientNet-B0 layer has that form: pw2_supported)
  CCVPE_PW2(2, 7, 1, true) CCVPE_PW2(4, 5, 2, false) CCVPE_PW2(4, 4, 2, true) CCVPE_PW2(4, 3, 2, true)
#undef CCVPE_PW2
  return fail(CCVPE_EINVAL, "pw2: no kernel for tile <%d,%d,%d> with this epilogue", mt, nt, wn);
}

}  // namespace ccvpe
