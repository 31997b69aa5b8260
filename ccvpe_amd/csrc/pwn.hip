// Narrow pointwise GEMM for the MBConv PROJECTION layers of the early blocks (bf16 and fp32 storage): 1x1 convolution + BN (+ SE gate on the
// input, + skip) with few output channels (N <= 48), short K (<= 256) and millions of pixels — efficientnet_pytorch/model.py:86-87,
// 118-131 for blocks 0-4 (32 -> 16 at 256 x 256 ... 240 -> 40 at 64 x 64).
//
// These layers are pure streaming (block 0: 268 MB in, 134 MB out, 0.1 FLOP per byte) and ran through the generic gather kernel
// (igemm_kernel<bf16,4,1,1> / <4,1,2> / <4,3,1>: register-staged operands -> LDS -> fragments, a 1-8 stage K loop between a prologue
// and an epilogue per 256-pixel tile) at ~2.6 TB/s: 0.64 ms of the bf16 forward.  Here (round 6):
//   * the layer's WHOLE weight matrix lives in registers as MFMA A operands (N <= 48 x K <= 256: at most 96 VGPRs), with the SE gate of
//     the current sample folded in (w'[n][k] = w[n][k] * gate[b][k], rebuilt when a wave's tiles move to the next sample — a wave
//     walks a contiguous run of 16-pixel tiles, i.e. one or two samples);
//   * a wave streams its tiles on its own: x fragments come straight from global memory in operand layout (16 bytes per lane), PF
//     tiles ahead (PF = 2 ... 6 by K), the skip rows with them; no LDS, no barrier, no workgroup-level phase;
//   * the MFMA C layout gives a lane 4 consecutive channels of one pixel: BN, skip, one 8-byte store; a tile's rows are contiguous
//     (16 pixels x N channels), so a wave's stores cover whole cache lines.
// fp32 storage (the fp32 PATH: exact fp32 on v_mfma_f32_16x16x4_f32, the gate folded as an fp32 product): the same kernel with 16-channel
// K pieces; 240 -> 40 does not fit the registers there (180 for W alone) and stays with the generic kernel.
#include "conv_common.h"
#include <type_traits>

namespace ccvpe {

bool g_use_pwn = true;

typedef float pwn_f32x4 __attribute__((ext_vector_type(4)));
typedef float pwn_f32x2 __attribute__((ext_vector_type(2)));

struct PwnParams {
  const void* x;
  const void* w;
  const float* scale;
  const float* shift;
  const float* gate;
  const void* res;
  void* dst;
  int M, c0, ld0, Kpad, N, ldd, ldres, hw, tiles, tpw;
};

// T = storage type; NKK = 64-byte K pieces per row (32 bf16 / 16 fp32 channels each)
template <typename T, int NT, int NKK, bool GATED, bool RES>
__global__ __launch_bounds__(256) void pwn_kernel(const PwnParams p) {
  constexpr int E = 16 / (int)sizeof(T);             // elements per 16-byte piece of a lane
  constexpr int SK = 4 * E;                          // channels per K piece
  constexpr int PF = NKK >= 6 ? 2 : (NKK >= 4 ? 3 : (NKK >= 2 ? 4 : 6));     // register sets of x fragments (tiles in flight)
  const T* xg = reinterpret_cast<const T*>(p.x);
  const T* wg = reinterpret_cast<const T*>(p.w);
  const T* rg = reinterpret_cast<const T*>(p.res);
  T* dg = reinterpret_cast<T*>(p.dst);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_g = blockIdx.x * 4 + (tid >> 6);
  const int px = lane & 15, kg = lane >> 4;
  const int t0 = wave_g * p.tpw;
  const int t1 = min(t0 + p.tpw, p.tiles);
  if (t0 >= t1) return;

  // ---- this lane's share of the weights (A operand: row n = 16 nt + px, k group kg of each 32-channel step) ------------------------
  pwn_f32x4 wf[NT][NKK];
  unsigned koff[NKK];                                  // channel offset of the lane's piece (pieces past c0 re-read channel 0..7: zero weights)
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) {
    const int ch = SK * kk + E * kg;
    koff[kk] = (unsigned)(ch < p.c0 ? ch : 0);
  }
  auto load_w = [&](int b) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        pwn_f32x4 raw = *reinterpret_cast<const pwn_f32x4*>(wg + (size_t)(16 * nt + px) * p.Kpad + SK * kk + E * kg);
        if (GATED) {
          const float* gp = p.gate + (size_t)b * p.c0 + koff[kk];
          if constexpr (sizeof(T) == 2) {
            const bf16x8 h = __builtin_bit_cast(bf16x8, raw);
            const pwn_f32x4 g0 = *reinterpret_cast<const pwn_f32x4*>(gp), g1 = *reinterpret_cast<const pwn_f32x4*>(gp + 4);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              o[j] = (bf16_t)((float)h[j] * g0[j]);
              o[j + 4] = (bf16_t)((float)h[j + 4] * g1[j]);
            }
            raw = __builtin_bit_cast(pwn_f32x4, o);
          } else {
            raw = raw * *reinterpret_cast<const pwn_f32x4*>(gp);
          }
        }
        wf[nt][kk] = raw;
      }
  };
  int cur_b = (int)(((long)t0 * 16) / p.hw);
  load_w(cur_b);
  // BN vectors of this lane's channels 16 nt + 4 kg .. + 3 (clamped where the tile's tail has no channel: never stored)
  pwn_f32x4 sc[NT], sh[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = min(16 * nt + 4 * kg, p.N - 4);
    sc[nt] = *reinterpret_cast<const pwn_f32x4*>(p.scale + n);
    sh[nt] = *reinterpret_cast<const pwn_f32x4*>(p.shift + n);
  }

  pwn_f32x4 xf[PF][NKK];
  typedef typename std::conditional<sizeof(T) == 2, pwn_f32x2, pwn_f32x4>::type res_t;   // 4 channels of the skip row
  res_t rf[PF][RES ? NT : 1];
  auto load_tile = [&](int s, int t) {                 // (s is a compile-time constant at every call site)
    const int pi = min(16 * min(t, t1 - 1) + px, p.M - 1);
    const T* src = xg + (size_t)pi * p.ld0;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) xf[s][kk] = *reinterpret_cast<const pwn_f32x4*>(src + koff[kk]);
    if (RES) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        rf[s][nt] = *reinterpret_cast<const res_t*>(rg + (size_t)pi * p.ldres + min(16 * nt + 4 * kg, p.N - 4));
    }
  };
#pragma unroll
  for (int s = 0; s < PF - 1; ++s) load_tile(s, t0 + s);

  for (int t = t0; t < t1; t += PF) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      const int tt = t + s;
      if (tt < t1) {                                     // wave-uniform
        load_tile((s + PF - 1) % PF, tt + PF - 1);       // PF - 1 tiles ahead (clamped re-reads past the wave's last tile)
        if (GATED) {
          const int b = (int)(((long)tt * 16) / p.hw);
          if (b != cur_b) {                              // wave-uniform: the wave's tiles moved to the next sample
            cur_b = b;
            load_w(b);
          }
        }
        pwn_f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[nt] = (pwn_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < NKK; ++kk) {
            if constexpr (sizeof(T) == 2) {
              acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[nt][kk]), __builtin_bit_cast(bf16x8, xf[s][kk]),
                                                                acc[nt], 0, 0, 0);
            } else {                                     // lane group kg supplies channels 16 kk + 4 kg + j to the j-th instruction on both sides
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt][kk][j], xf[s][kk][j], acc[nt], 0, 0, 0);
            }
          }
        }
        const int pi = 16 * tt + px;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int n = 16 * nt + 4 * kg;
          pwn_f32x4 v = acc[nt] * sc[nt] + sh[nt];
          if (RES) {
            if constexpr (sizeof(T) == 2) {
              const bf16x4 rv = __builtin_bit_cast(bf16x4, rf[s][nt]);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
            } else {
              v += rf[s][nt];
            }
          }
          if (pi < p.M && n < p.N) st4<T>(dg + (size_t)pi * p.ldd + n, v);
        }
      }
    }
  }
}

// which (NT, NKK) instantiation serves the layer: 0 = none.  Exact K-piece counts: the kernels read NKK x 64 bytes of every W row.
static int pwn_id(int n, int c0, int esz) {
  const int sk = 64 / esz;
  const int nt = (n + 15) / 16, nkk = (c0 + sk - 1) / sk;
  if (esz == 2) {
    if (nt == 1 && nkk == 1) return 1;      // 32 -> 16
    if (nt == 2 && nkk == 3) return 2;      // 96 -> 24
    if (nt == 2 && nkk == 5) return 3;      // 144 -> 24
    if (nt == 3 && nkk == 5) return 4;      // 144 -> 40
    if (nt == 3 && nkk == 8) return 5;      // 240 -> 40
  } else {
    if (nt == 1 && nkk == 2) return 1;
    if (nt == 2 && nkk == 6) return 2;
    if (nt == 2 && nkk == 9) return 3;
    if (nt == 3 && nkk == 9) return 4;      // (240 -> 40: 180 registers of W alone: the generic kernel)
  }
  return 0;
}

bool pwn_supported(const IgemmParams& p, int batch, int esz) {
  if (!g_use_pwn || (esz == 2 && p.out_f32) || p.c1 != 0 || p.kw != 1 || p.stride != 1 || p.pad != 0 || p.out_mode != CCVPE_OUT_NHWC) return false;
  if (p.H != p.Ho || p.W != p.Wo) return false;                        // 1x1
  if (p.act != CCVPE_ACT_NONE || !p.scale || !p.shift) return false;   // BN, no activation: the MBConv projection
  if (p.N > 48 || p.N % 8 || p.N < 8 || p.c0 > 256 || p.c0 % 8 || !pwn_id(p.N, p.c0, esz)) return false;
  if (p.Kpad != (p.c0 + 64 / esz - 1) / (64 / esz) * (64 / esz)) return false;
  if (p.ldd % 4 || (p.residual && p.ldres % 4) || p.ld0 % (16 / esz)) return false;
  const long hw = (long)p.Ho * p.Wo;
  if (p.gate && hw % 16) return false;                                 // a 16-pixel tile inside one sample: one gate vector
  if (p.M < 64 * 1024) return false;                                   // streaming form: large planes only (>= 32 tiles per wave slot)
  if ((double)p.M * p.ld0 * 2 >= 1e18) return false;
  (void)batch;
  return true;
}

template <typename T, int NT, int NKK>
static int pwn_launch(const PwnParams& q, bool gated, bool res, hipStream_t st) {
  // persistent: two workgroups per CU worth of waves, each wave a contiguous run of tiles
  const int waves = 2 * num_cus() * 4;
  PwnParams p = q;
  p.tpw = (p.tiles + waves - 1) / waves;
  if (p.tpw < 1) p.tpw = 1;
  const int wgs = ((p.tiles + p.tpw - 1) / p.tpw + 3) / 4;
#define PWN_GO(G_, R_) hipLaunchKernelGGL((pwn_kernel<T, NT, NKK, G_, R_>), dim3(wgs), dim3(256), 0, st, p)
  if (gated && res) PWN_GO(true, true);
  else if (gated) PWN_GO(true, false);
  else if (res) PWN_GO(false, true);
  else PWN_GO(false, false);
#undef PWN_GO
  return check_launch("pwn_kernel");
}

int pwn_dispatch(const IgemmParams& ip, int batch, int esz, hipStream_t st) {
  (void)batch;
  PwnParams p;
  p.x = ip.src0; p.w = ip.w;
  p.scale = ip.scale; p.shift = ip.shift; p.gate = ip.gate;
  p.res = ip.residual; p.dst = ip.dst;
  p.M = ip.M; p.c0 = ip.c0; p.ld0 = ip.ld0; p.Kpad = ip.Kpad; p.N = ip.N; p.ldd = ip.ldd; p.ldres = ip.ldres;
  p.hw = ip.Ho * ip.Wo;
  p.tiles = (ip.M + 15) / 16;
  p.tpw = 0;
  const bool gated = ip.gate != nullptr, res = ip.residual != nullptr;
  const int id = pwn_id(ip.N, ip.c0, esz);
  if (esz == 2) {
    switch (id) {
      case 1: return pwn_launch<bf16_t, 1, 1>(p, gated, res, st);
      case 2: return pwn_launch<bf16_t, 2, 3>(p, gated, res, st);
      case 3: return pwn_launch<bf16_t, 2, 5>(p, gated, res, st);
      case 4: return pwn_launch<bf16_t, 3, 5>(p, gated, res, st);
      case 5: return pwn_launch<bf16_t, 3, 8>(p, gated, res, st);
    }
  } else {
    switch (id) {
      case 1: return pwn_launch<float, 1, 2>(p, gated, res, st);
      case 2: return pwn_launch<float, 2, 6>(p, gated, res, st);
      case 3: return pwn_launch<float, 2, 9>(p, gated, res, st);
      case 4: return pwn_launch<float, 3, 9>(p, gated, res, st);
    }
  }
  return fail(CCVPE_EINVAL, "pwn: no kernel for N = %d, K = %d", ip.N, ip.c0);
}

}  // namespace ccvpe

extern "C" int ccvpe_set_pwn_kernels(int on) {
  const int prev = ccvpe::g_use_pwn ? 1 : 0;
  ccvpe::g_use_pwn = on != 0;
  return prev;
}
