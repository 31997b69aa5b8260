// upconv_kernel / upconv_halo_kernel: see the comment blocks below.  Included by upconv_f32.hip / upconv_bf16.hip.
#pragma once
#include "conv_common.h"

namespace ccvpe {


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

  constexpr bool SWZ = PanelLayout<T>::SWZ;
  constexpr int PLD = PanelLayout<T>::LD;
  __shared__ __attribute__((aligned(16))) float As[2][BM][PLD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN][PLD];

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
  const int scol = (SWZ ? (ssub ^ panel_swz(srow)) : ssub) * 4;       // staged piece -> (swizzled) float column

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
      *reinterpret_cast<f32x4*>(&As[buf][srow + 64 * it][scol]) = keep_if(a_reg[it], (a_keep >> it) & 1u);
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int nrow = srow + 64 * it;
      if (nrow < BN) *reinterpret_cast<f32x4*>(&Bs[buf][nrow][scol]) = b_reg[it];
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;
  const int fk = (SWZ ? ((lane >> 4) ^ panel_swz(frow)) : (lane >> 4)) * 4;

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
  IgemmParams ep{};   // reuse store4 (needs N, act, residual, dst, out_f32)
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
  // Measured on the decoder shapes (tools/up_probe.py bf16, round 4): the swizzled W panel alone +3-5 % on the wide tiles (N >= 128),
  // -5 % on the 32-column tile; the swizzled HALO (as in conv3x3_kernel, where it is worth 7-12 %) -8-12 % here — this kernel reads
  // its fragments at the top of a one-tap stage and waits for them, it is not bound by LDS throughput.  So: W panel only, NT >= 2.
  constexpr bool SWZ = false;                                    // halo + phase-B stage (Us): padded pitch 20
  constexpr bool WSWZ = PanelLayout<T>::SWZ && NT >= 2;          // W panel (Bs): unpadded, XOR-swizzled
  constexpr int PLD = SWZ ? 16 : LDS_LD;
  constexpr int WLD = WSWZ ? 16 : LDS_LD;
  constexpr int HCP = SWZ ? 24 : HC;                         // halo slots per row (bf16: see conv3x3_kernel)
  constexpr int UROWS = (HR * HCP > 2 * BM) ? HR * HCP : 2 * BM;       // halo [HR][HCP] rows  |  A stage [2][BM] rows

  __shared__ __attribute__((aligned(16))) float Us[UROWS][PLD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN][WLD];

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
  const int scol = (SWZ ? (ssub ^ panel_swz(srow)) : ssub) * 4;        // staged piece -> (swizzled) float column
  const int wscol = (WSWZ ? (ssub ^ panel_swz(srow)) : ssub) * 4;

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
      h_off[it] = (hy * HCP + hx) * PLD + (SWZ ? (sub ^ panel_swz(hx)) : sub) * 4;
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
      if (nrow < BN) *reinterpret_cast<f32x4*>(&Bs[buf][nrow][wscol]) = b_reg[it];
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15;
  const int fk = (SWZ ? ((lane >> 4) ^ panel_swz(frow)) : (lane >> 4)) * 4;       // row-indexed panel (phase-B stage)
  const int wfk = (WSWZ ? ((lane >> 4) ^ panel_swz(frow)) : (lane >> 4)) * 4;     // W panel
  // halo fragment (phase A): per-lane float offset at column shift dv + px, dv = 0 / 1 (two plain registers: an indexed array
  // here became a scratch array)
  auto hcol_at = [&](int c) { return (wm * MT * HCP + frow + c) * PLD + (SWZ ? ((lane >> 4) ^ panel_swz(frow + c)) : (lane >> 4)) * 4; };
  const int hdv0 = hcol_at(px) + py * (HCP * PLD), hdv1 = hcol_at(px + 1) + py * (HCP * PLD);

  auto mfma_block = [&](const f32x4* af, int wbuf) {
    f32x4 bf[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
      bf[j] = *reinterpret_cast<const f32x4*>(&Bs[wbuf][(wn * NT + j) * 16 + frow][wfk]);
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
    // one (chunk, tap) stage; `ab` = this lane's halo fragment offset for the tap (swizzled layout only)
    auto stage_a = [&](int s, int chunk, int tap, int ab) {
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
      if constexpr (SWZ) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
          af[i] = *reinterpret_cast<const f32x4*>(&Us[0][0] + ab + i * (HCP * PLD));
      } else {
#pragma unroll
        for (int i = 0; i < MT; ++i)
          af[i] = *reinterpret_cast<const f32x4*>(&Us[0][0] + (((wm * MT + i) + du + py) * HC + frow + dv + px) * LDS_LD + fk);
      }
      mfma_block(af, wbuf);
      if (more) store_w(wbuf ^ 1);
      __syncthreads();
      if (tap == 3 && more) {
        store_halo();
        __syncthreads();
      }
      if (more) wbuf ^= 1;
    };
    // (swizzled layout) the fragment offset of a tap depends on the column shift dv + px through the XOR term: it is computed
    // for the NEXT stage under the current stage's matrix work — computed at the top of a stage it put a chain of dependent
    // selects in front of the stage's first ds_read (+7-20 % on these kernels)
    auto tap_off = [&](int tap) { return ((tap & 1) ? hdv1 : hdv0) + (tap >> 1) * (HCP * PLD); };
    int chunk = 0, tap = 0, ab = SWZ ? tap_off(0) : 0;
    for (int s = 0; s < nstA; ++s) {
      const int ntap = (tap + 1) & 3;
      const int abn = SWZ ? tap_off(ntap) : 0;
      stage_a(s, chunk, tap, ab);
      ab = abn;
      if (++tap == 4) { tap = 0; ++chunk; }
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
        *reinterpret_cast<f32x4*>(&Us[buf * BM + srow + 64 * it][scol]) = keep_if(a_reg[it], (a_keep >> it) & 1u);
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
  IgemmParams ep{};
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

// ---------------------------------------------------------------------------------------------
// upconv_dma_kernel (round 4; bf16 only): phase A of upconv_halo_kernel in the stage structure of conv3x3_kernel.
// In bf16 a one-tap stage of upconv_halo_kernel is 20 matrix instructions = 320 cycles, far less than the L2 round trip of the
// W panel it waits for at its end (register-staged, requested at the start of the same stage): the traced kernels run at 24 %
// of the matrix peak (5x the fp32 kernel where conv3x3_kernel gets 7x).  Here a stage is a PAIR of taps (du; dv = 0, 1) of a
// 32-channel chunk: its two W panels arrive by LDS-DMA (requested at the start of the PREVIOUS stage, no VGPR round trip, no
// ds_write), the fragments of the second tap are read while the first tap's matrix instructions run, and the halo uses the
// conflict-free swizzled layout of conv3x3_kernel.  Phase B (the skip's 9 taps, 28-36 % of K) is the register-staged gather of
// upconv_halo_kernel unchanged.  Needs Npad % BN == 0 (the DMA has no row guard) and W1 >= 16.
// ---------------------------------------------------------------------------------------------
template <typename T, int MT, int NT, int WN, bool PAIR = false>
struct UpDmaGeom {
  static constexpr int WM = 4 / WN;
  static constexpr int BM = 16 * MT * WM;
  static constexpr int BN = 16 * NT * WN;
  static constexpr int TH = BM / 16;
  static constexpr int HR = TH + 2, HCP = PAIR ? 32 : 24, HLD = 16;
  static constexpr int SUBS = (WN == 2 && NT >= 3) ? 2 : 1;   // phase B: K stages per barrier (256-pixel tiles: no LDS for two; narrow tiles measured slower with two)
  static constexpr int U_FLOATS = (HR * HCP * HLD > 2 * SUBS * BM * 16) ? HR * HCP * HLD : 2 * SUBS * BM * 16;   // halo | phase-B A stage [2][SUBS][BM][16]
  static constexpr int B_FLOATS = 2 * 2 * BN * 16;                                                           // [2][2 taps][BN][16] (phase B: [2][BN][16])
  static constexpr int LDS_BYTES = (U_FLOATS + B_FLOATS) * 4;
};

template <typename T, int MT, int NT, int WN, bool PAIR = false>
__global__ __launch_bounds__(256, 2) void upconv_dma_kernel(const UpParams p) {
  using G = UpDmaGeom<T, MT, NT, WN, PAIR>;
  constexpr int E = ElemTraits<T>::E;
  constexpr int SK = 4 * E;
  constexpr int CPS = SK / 8;
  constexpr int BM = G::BM, BN = G::BN, TH = G::TH, HR = G::HR, HCP = G::HCP, HLD = G::HLD;
  constexpr int HC = PAIR ? 20 : 18, HPX = HR * HC;   // staged halo columns (PAIR: two 10-column halos)
  constexpr int H_IT = (HPX * 4 + 255) / 256;
  constexpr int A_IT = BM / 64;
  constexpr int NSLOT = (BN / 16 + 3) / 4;
  constexpr int SUBS = G::SUBS;

  extern __shared__ __attribute__((aligned(16))) float up_sm[];
  float* Us = up_sm;                               // halo [HR][HCP][HLD]  |  phase-B A stage [2][SUBS][BM][16] (swizzled)
  float* Bs = up_sm + G::U_FLOATS;                 // [2][2][BN][16] by DMA (phase B uses [2][SUBS])

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = sgpr(tid >> 6);
  const int wm = wave / WN;
  const int wn = wave % WN;

  const int tile = xcd_tile(blockIdx.x, p.tiles_total);
  const int tn = tile % p.tiles_n;
  const int par = (tile / p.tiles_n) & 3;
  const int ts = tile / (p.tiles_n * 4);
  // PAIR (8 x 8 low-res images, level 6): a tile is TWO images side by side — pixel columns 0-7 = image 2 ts, 8-15 = image
  // 2 ts + 1; their 10-column halos sit at slots 0-9 and 16-25 of a 32-slot halo row (the gap of 8 keeps the swizzled
  // fragment reads conflict-free: tests/test_lds_layout.py).  Otherwise one TH x 16 tile of one image.
  const int tiles_x = PAIR ? 1 : (p.W1 + 15) / 16;
  const int tiles_y = PAIR ? 1 : (p.H1 + TH - 1) / TH;
  const int tx = ts % tiles_x;
  const int ty = (ts / tiles_x) % tiles_y;
  const int b = PAIR ? 2 * ts : ts / (tiles_x * tiles_y);
  const int nbatch = p.M / (p.H1 * p.W1);
  const int py = par >> 1, px = par & 1;
  const int y0 = ty * TH, x0 = tx * 16;
  const int n0 = tn * BN;
  const int H2 = 2 * p.H1, W2 = 2 * p.W1;

  const T* src0 = reinterpret_cast<const T*>(p.src0);
  const T* src1 = reinterpret_cast<const T*>(p.src1);
  const T* wp = reinterpret_cast<const T*>(p.w) + (size_t)par * p.Npad * p.Kpad;
  const int srow = tid >> 2, ssub = tid & 3;

  // ---- halo staging coordinates ----------------------------------------------------------------
  int h_off[H_IT], h_pix[H_IT], h_sub[H_IT];
#pragma unroll
  for (int it = 0; it < H_IT; ++it) {
    const int idx = tid + 256 * it;
    const int pxl = idx >> 2, sub = idx & 3;
    h_sub[it] = sub;
    if (pxl < HPX) {
      const int hy = pxl / HC, hc = pxl - hy * HC;
      const int img = PAIR ? hc / 10 : 0;
      const int lx = hc - 10 * img, hx = lx + 16 * img;              // column inside the image's halo, slot in the halo row
      const int iy = y0 - 1 + hy, ix = x0 - 1 + lx;
      h_off[it] = (hy * HCP + hx) * HLD + (sub ^ panel_swz(hx)) * 4;
      h_pix[it] = ((unsigned)iy < (unsigned)p.H1 && (unsigned)ix < (unsigned)p.W1 && b + img < nbatch) ? ((b + img) * p.H1 + iy) * p.W1 + ix : -1;
    } else {
      h_off[it] = -1;
      h_pix[it] = -1;
    }
  }
  f32x4 h_reg[H_IT];
  unsigned h_keep = 0;
  const int ld0s = sgpr(p.ld0), ld1s = sgpr(p.ld1);
  const bool small32 = (double)p.M * 4.0 * (double)(ld0s > ld1s ? ld0s : ld1s) * sizeof(T) < 4294967296.0;
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
      if (h_off[it] >= 0) *reinterpret_cast<f32x4*>(Us + h_off[it]) = keep_if(h_reg[it], (h_keep >> it) & 1u);
  };

  // ---- W by LDS-DMA (phase A): lane -> (row = lane >> 2 of a 16-row group, swizzled 16-byte piece), as conv3x3_kernel --------
  unsigned wvoff[NSLOT];
  {
    const int rl = lane >> 2;
    const int c = (lane & 3) ^ w_swz(rl);
#pragma unroll
    for (int q = 0; q < NSLOT; ++q) {
      const int g = min(wave + 4 * q, BN / 16 - 1);
      wvoff[q] = ((unsigned)(n0 + g * 16 + rl) * (unsigned)p.Kpad + (unsigned)(c * E)) * (unsigned)sizeof(T);
    }
  }
  const unsigned bs_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)Bs;
  const int kmax = p.Kpad - SK;                    // last K column a 64-byte panel row may start at (stay inside the W row)
  auto dma_w = [&](int chunk, int du, int dbuf) {  // taps (du, 0), (du, 1) of `chunk` -> Bs[dbuf][0..1]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      // (a panel beyond the channel range meets zeroed halo pieces; the launcher guarantees kcol <= kmax for every stage)
      const int kcol = min((du * 2 + t) * p.c0 + chunk * SK, kmax);
      const char* sbase = reinterpret_cast<const char*>(wp) + (size_t)kcol * sizeof(T);
#pragma unroll
      for (int q = 0; q < NSLOT; ++q) {
        const int g = wave + 4 * q;
        if (g < BN / 16) {
          const unsigned lds = __builtin_amdgcn_readfirstlane(bs_lds + (unsigned)(((dbuf * 2 + t) * BN + g * 16) * 16 * 4));
          asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(wvoff[q]), "s"(sbase) : "memory", "m0");
        }
      }
    }
  };
  auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15;
  const int bcol = ((lane >> 4) ^ w_swz(frow)) * 4;                   // DMA panel (phase A)
  const int fkb = ((lane >> 4) ^ panel_swz(frow)) * 4;                // phase-B A stage (row-indexed, panel_swz)
  auto hcol_at = [&](int c) {
    const int hx = frow + c + (PAIR ? 8 * (frow >> 3) : 0);
    return (wm * MT * HCP + hx) * HLD + ((lane >> 4) ^ panel_swz(hx)) * 4;
  };
  const int hdv0 = hcol_at(px) + py * (HCP * HLD), hdv1 = hcol_at(px + 1) + py * (HCP * HLD);

  // ================= phase A: low-res source, 2 stages (du = 0, 1) of 2 taps per 32-channel chunk =================
  const int nchunks0 = (p.c0 + SK - 1) / SK;
  const int nstA = nchunks0 * 2;
  load_halo(0);
  dma_w(0, 0, 0);
  store_halo();
  dma_wait();
  __syncthreads();
  auto stage_a = [&](int s, int chunk, auto du_tag) {
    constexpr int du = decltype(du_tag)::value;
    const bool more = s + 1 < nstA;
    const bool next_halo = (du == 0) && (chunk + 1 < nchunks0);
    f32x4 af[2][MT], bf[2][NT];
    auto read_frag = [&](int t, f32x4* a, f32x4* w_) {
      const int ab = (t ? hdv1 : hdv0) + du * (HCP * HLD);
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const f32x4*>(Us + ab + i * (HCP * HLD));
#pragma unroll
      for (int j = 0; j < NT; ++j)
        w_[j] = *reinterpret_cast<const f32x4*>(&Bs[((((s & 1) * 2 + t) * BN) + (wn * NT + j) * 16 + frow) * 16 + bcol]);
    };
    read_frag(0, af[0], bf[0]);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t == 0) {
        read_frag(1, af[1], bf[1]);                       // second tap's fragments: in flight under the first tap's MFMAs
        if (next_halo) load_halo(chunk + 1);
        if (more) dma_w(du == 0 ? chunk : chunk + 1, du ^ 1, (s + 1) & 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (sizeof(T) == 4) {                     // (fp32: the k-step-outer order of the other fp32 kernels)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[t][j][kk], af[t][i][kk], acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = mfma_stage<T>(bf[t][j], af[t][i], acc[i][j]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    dma_wait();
    __syncthreads();
    if (du == 1 && more) {                                // chunk boundary: every wave is done reading the halo -> overwrite it
      store_halo();
      __syncthreads();
    }
  };
  for (int chunk = 0; chunk < nchunks0; ++chunk) {
    stage_a(2 * chunk, chunk, std::integral_constant<int, 0>{});
    stage_a(2 * chunk + 1, chunk, std::integral_constant<int, 1>{});
  }

  // ================= phase B: skip, 9 taps (stride 2, parity offset), gathered =========================================
  // K stages of 32 (tap, channel) columns as in upconv_halo_kernel, but SUBS of them per barrier: the A pieces of the next
  // group are requested at the start of a group and written to LDS at its end, W arrives by DMA, the fragments of the second
  // K stage are read under the first one's matrix instructions.
  const int chunksB = 9 * p.cpt1;
  if (chunksB > 0) {
    const int stagesB = (chunksB + CPS - 1) / CPS;
    const int ngroups = (stagesB + SUBS - 1) / SUBS;
    const int chunk_in_stage = (ssub * E) >> 3;
    const int half = (ssub * E) & 7;
    const int kB0 = 4 * p.c0;
    const int scola = (ssub ^ panel_swz(srow)) * 4;
    f32x4 a_reg[SUBS][A_IT];
    unsigned a_keep[SUBS];
    int a_pix[A_IT], a_yy[A_IT], a_xx[A_IT], a_bb[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int ml = srow + 64 * it;
      const int col = ml & 15;
      const int y1 = y0 + (ml >> 4), x1 = PAIR ? (col & 7) : x0 + col;
      a_bb[it] = PAIR ? b + (col >> 3) : b;
      a_pix[it] = (y1 < p.H1 && x1 < p.W1 && a_bb[it] < nbatch) ? 1 : 0;
      a_yy[it] = 2 * y1 + py - 1;
      a_xx[it] = 2 * x1 + px - 1;
    }
    auto load_a = [&](int s2, int sub) {           // K stage s2 -> a_reg[sub] (sub: compile-time after unrolling)
      const int kc = CPS * s2 + chunk_in_stage;
      const bool kvalid = kc < chunksB;
      const int tapb = kc / p.cpt1;
      const int ch = kvalid ? (kc - tapb * p.cpt1) * 8 + half : 0;
      const int ky = tapb / 3, kx = tapb - 3 * ky;
      a_keep[sub] = 0;
#pragma unroll
      for (int it = 0; it < A_IT; ++it) {
        const int iy = a_yy[it] + ky, ix = a_xx[it] + kx;
        const bool ok = kvalid && a_pix[it] && (unsigned)iy < (unsigned)H2 && (unsigned)ix < (unsigned)W2;
        const int pix = ok ? (a_bb[it] * H2 + iy) * W2 + ix : 0;
        if (small32) a_reg[sub][it] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(src1) + ((unsigned)pix * (unsigned)ld1s + (unsigned)ch) * (unsigned)sizeof(T));
        else a_reg[sub][it] = *reinterpret_cast<const f32x4*>(src1 + (size_t)pix * ld1s + ch);
        a_keep[sub] |= ok ? (1u << it) : 0u;
      }
    };
    auto store_a = [&](int buf) {
#pragma unroll
      for (int sub = 0; sub < SUBS; ++sub)
#pragma unroll
        for (int it = 0; it < A_IT; ++it)
          *reinterpret_cast<f32x4*>(&Us[((buf * SUBS + sub) * BM + srow + 64 * it) * 16 + scola]) = keep_if(a_reg[sub][it], (a_keep[sub] >> it) & 1u);
    };
    auto dma_wb = [&](int grp, int dbuf) {          // the SUBS K stages of group `grp` -> Bs[dbuf][0..SUBS)
#pragma unroll
      for (int sub = 0; sub < SUBS; ++sub) {
        const int kcol = min(kB0 + (grp * SUBS + sub) * SK, kmax);      // (past the last stage: any panel, its A pieces are zero)
        const char* sbase = reinterpret_cast<const char*>(wp) + (size_t)kcol * sizeof(T);
#pragma unroll
        for (int q = 0; q < NSLOT; ++q) {
          const int g = wave + 4 * q;
          if (g < BN / 16) {
            const unsigned lds = __builtin_amdgcn_readfirstlane(bs_lds + (unsigned)(((dbuf * 2 + sub) * BN + g * 16) * 16 * 4));
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(wvoff[q]), "s"(sbase) : "memory", "m0");
          }
        }
      }
    };
    // the halo and phase A's panels are dead (phase A ended on a barrier)
#pragma unroll
    for (int sub = 0; sub < SUBS; ++sub) load_a(sub, sub);
    dma_wb(0, 0);
    store_a(0);
    dma_wait();
    __syncthreads();
    for (int grp = 0; grp < ngroups; ++grp) {
      const int buf = grp & 1;
      const bool more = grp + 1 < ngroups;
      f32x4 af[SUBS][MT], bf[SUBS][NT];
      auto read_frag = [&](int sub, f32x4* a, f32x4* w_) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
          a[i] = *reinterpret_cast<const f32x4*>(&Us[((buf * SUBS + sub) * BM + (wm * MT + i) * 16 + frow) * 16 + fkb]);
#pragma unroll
        for (int j = 0; j < NT; ++j)
          w_[j] = *reinterpret_cast<const f32x4*>(&Bs[((buf * 2 + sub) * BN + (wn * NT + j) * 16 + frow) * 16 + bcol]);
      };
      read_frag(0, af[0], bf[0]);
#pragma unroll
      for (int sub = 0; sub < SUBS; ++sub) {
        if (sub == 0) {
          if (SUBS > 1) read_frag(SUBS - 1, af[SUBS - 1], bf[SUBS - 1]);
          if (more) {
#pragma unroll
            for (int s2 = 0; s2 < SUBS; ++s2) load_a((grp + 1) * SUBS + s2, s2);
            dma_wb(grp + 1, buf ^ 1);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (sizeof(T) == 4) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
              for (int j = 0; j < NT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[sub][j][kk], af[sub][i][kk], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = mfma_stage<T>(bf[sub][j], af[sub][i], acc[i][j]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (more) store_a(buf ^ 1);
      dma_wait();
      __syncthreads();
    }
  }

  // ---- epilogue (as upconv_halo_kernel) -----------------------------------------------------------
  const int epix = lane & 15;
  const int en = (lane >> 4) * 4;
  IgemmParams ep{};
  ep.N = p.N; ep.act = p.act; ep.residual = nullptr; ep.dst = p.dst; ep.out_f32 = p.out_f32;
  const float one[4] = {1.f, 1.f, 1.f, 1.f};
  const int x1 = PAIR ? (epix & 7) : x0 + epix;
  const int eb = PAIR ? b + (epix >> 3) : b;
  auto epilogue = [&](auto act_tag) {
  constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int y1 = y0 + wm * MT + i;
    if (y1 >= p.H1 || x1 >= p.W1 || eb >= nbatch) continue;
    const int Y = 2 * y1 + py, X = 2 * x1 + px;
    const int rc = Y == 0 ? 0 : (Y == H2 - 1 ? 2 : 1);
    const int cc = X == 0 ? 0 : (X == W2 - 1 ? 2 : 1);
    const float* shp = p.shift9 + (size_t)(rc * 3 + cc) * p.N;
    const size_t pix = (size_t)(eb * H2 + Y) * W2 + X;
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

// route (optional): CCVPE_UPROUTE_* | MT << 8 | NT << 12 | WN << 16 of the kernel that WOULD run; nothing is launched
template <typename T, int MT, int NT, int WN>
static int launch_up(const UpParams& p0, hipStream_t stream, int* route = nullptr) {
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
  {
    // bf16: pair-of-taps stages with W by LDS-DMA (see upconv_dma_kernel) where the DMA's preconditions hold; fp32 (where the
    // kernel measured the same as upconv_halo_kernel, +-3 %) only for level 6, whose alternative is the linear-M gather kernel
    // (the 256 x 48 tile measured slower with it, 479 -> 531 us at N = 40, also with three workgroups per SIMD forced: 505)
    // Every 64-byte panel row the DMA reads must lie inside its W row: phase A reads K columns up to 3 c0 + ceil32(c0) (the last
    // tap's last, partly empty chunk: with c1 = 0 and c0 % 32 != 0 that crosses the row end — the reads of VALID K columns would
    // then have to be shifted, so such shapes stay on upconv_halo_kernel); phase B starts at 4 c0 = 0 mod 32 and ends inside Kpad.
    constexpr int SKL = 4 * ElemTraits<T>::E;
    const bool dma_ok = p.Npad % BN == 0 && 3 * p.c0 + (p.c0 + SKL - 1) / SKL * SKL <= p.Kpad && !(NT == 3 && WN == 1);
    auto go = [&](auto pair_tag) -> int {
      constexpr bool PAIR = decltype(pair_tag)::value;
      using G = UpDmaGeom<T, MT, NT, WN, PAIR>;
      static_assert(G::LDS_BYTES <= 80 * 1024, "upconv_dma_kernel: two workgroups per CU");
      static bool attr_set = false;                 // per (T, tile, PAIR) instantiation
      if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)upconv_dma_kernel<T, MT, NT, WN, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        if (e != hipSuccess) return fail(CCVPE_ELAUNCH, "upconv_dma_kernel: set smem attr: %s", hipGetErrorString(e));
        attr_set = true;
      }
      hipLaunchKernelGGL((upconv_dma_kernel<T, MT, NT, WN, PAIR>), dim3(p.tiles_total), dim3(256), G::LDS_BYTES, stream, p);
      return check_launch("upconv_dma_kernel");
    };
    const int tile_bits = (MT << 8) | (NT << 12) | (WN << 16);
    if constexpr (sizeof(T) == 2) {
      if (halo && dma_ok) {
        if (route) { *route = CCVPE_UPROUTE_DMA | tile_bits; return CCVPE_OK; }
        return go(std::false_type{});
      }
    }
    if constexpr (MT == 4 && NT == 5 && WN == 2) {
      // level 6 (8 x 8 low-res images): two images per 8 x 16 tile instead of the linear-M gather kernel
      if (dma_ok && p.W1 == 8 && p.H1 == TH) {
        if (route) { *route = CCVPE_UPROUTE_DMA_PAIR | tile_bits; return CCVPE_OK; }
        p.tiles_m = (batch + 1) / 2;
        p.tiles_total = p.tiles_m * p.tiles_n * 4;
        return go(std::true_type{});
      }
    }
    if (route) { *route = (halo ? CCVPE_UPROUTE_HALO : CCVPE_UPROUTE_GATHER) | tile_bits; return CCVPE_OK; }
  }
  if (halo)
    hipLaunchKernelGGL((upconv_halo_kernel<T, MT, NT, WN>), dim3(p.tiles_total), dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL((upconv_kernel<T, MT, NT, WN>), dim3(p.tiles_total), dim3(256), 0, stream, p);
  return check_launch("upconv_kernel");
}


int upconv_route_f32(const ccvpe_upconv_desc* d, int* route);     // upconv_f32.hip (the fp32 instantiations live there)

template <typename T>
static int upconv_any(const ccvpe_upconv_desc* d, void* stream, int* route = nullptr) {
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
  if constexpr (sizeof(T) == 2) {
    // the narrow levels: all four parities of a low-res tile in one workgroup, weights in registers (narrow_impl.h)
    if (g_use_narrow && d->c1 > 0 && d->ldd % 8 == 0 && d->n % 8 == 0 && (d->act == CCVPE_ACT_NONE || d->act == CCVPE_ACT_RELU) &&
        up2_supported(d->c0, d->c1, d->n, d->kpad, d->h1, d->w1, d->batch)) {
      if (route) { *route = CCVPE_UPROUTE_UP2; return CCVPE_OK; }
      return up2_dispatch(d->src0, d->src1, d->w, d->shift9, d->dst, d->c0, d->ld0, d->c1, d->ld1, d->h1, d->w1, d->n, d->kpad,
                          d->ldd, d->act, d->batch, st);
    }
  }
  const TileCfg c = kCfgs[pick_cfg(p.Npad)];
#define CCVPE_CASE(MT_, NT_, WN_) \
  if (c.mt == MT_ && c.nt == NT_ && c.wn == WN_) return launch_up<T, MT_, NT_, WN_>(p, st, route);
  CCVPE_CASE(4, 5, 2) CCVPE_CASE(4, 4, 2) CCVPE_CASE(4, 3, 2) CCVPE_CASE(4, 2, 2) CCVPE_CASE(4, 1, 2)
  CCVPE_CASE(4, 5, 1) CCVPE_CASE(4, 3, 1) CCVPE_CASE(4, 1, 1) CCVPE_CASE(2, 7, 1)
#undef CCVPE_CASE
  return fail(CCVPE_EINVAL, "upconv: no tile config");
}


}  // namespace ccvpe
