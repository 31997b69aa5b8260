// Narrow decoder levels in bf16 storage: kernels that keep their WEIGHTS IN REGISTERS and stream activation halo tiles through LDS.
// Included by narrow_bf16.hip.
//
// The last decoder levels of both branches (models.py:121-127, 142-148: 256 x 256 and 512 x 512 pixels, 16-40 output channels,
// K = 360-500) are long-M / short-K / narrow-N problems.  The tiled GEMM kernels (conv3x3_kernel, upconv_*_kernel) re-fetch the
// whole weight matrix per output tile — 55 KB of W panels against 26 KB of activations for the 40 -> 40 layer — pad every tap's
// channels to a 32-channel chunk (40 -> 64: 1.6 x the matrix work) and live between a prologue and an epilogue: round 4 traced
// them at 0.3-0.35 PF and 3-5 x off the HBM roof.  Here:
//   * K is FLAT over (tap, 8-channel octet): a 32-wide MFMA k-step takes four consecutive octets wherever they fall, so a layer
//     with 5 octets per tap runs 12 k-steps instead of 18 (the packed weights already have this K order);
//   * the weights of the layer — 12 k-steps x 3 column tiles x 4 registers = 144 VGPRs for 40 -> 40 — are loaded ONCE per
//     workgroup into registers as MFMA operands (one wave per SIMD, up to 512 registers per lane) and stay there: no W traffic,
//     no W panel barriers, LDS carries activations only;
//   * workgroups are PERSISTENT (one per CU) and walk the tiles; the halo tile of the NEXT tile is requested by LDS-DMA
//     (global_load_lds_dwordx4, lane-linear: the LDS image is [pixel][octet] in DMA piece order) at the start of a tile into the
//     other of two buffers; pieces outside the image are zeroed by the lane that would have requested them;
//   * one barrier per tile.
#pragma once
#include "conv_common.h"

namespace ccvpe {

struct NarrowParams {
  const void* src;     // [B,H,W,ld] bf16, first 8*CPT channels used
  const void* w;       // packed [Npad][Kpad] bf16, k = tap * (8 CPT) + channel (models._pack_conv)
  const float* shift;  // bias [N] or nullptr
  void* dst;           // [B,H,W,ldd] bf16 or fp32
  int out_f32;
  int H, W, ld, N, Kpad, ldd, act;
  int tiles_x, tiles_y, tiles_total;
};

// one LDS-DMA request of 64 x 16 bytes: lane l's 16 bytes from sbase + voff land at lds + 16 l (lds wave-uniform)
__device__ __forceinline__ void dma16(unsigned lds, unsigned voff, const char* sbase) {
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(voff), "s"(sbase) : "memory", "m0");
}

template <int CPT, int NT, int MT>
struct C3nGeom {
  static constexpr int TH = 4 * MT, HR = TH + 2, HC = 18;
  static constexpr int PIECES = HR * HC * CPT;                 // 16-byte pieces of a halo tile
  static constexpr int NDMA = (PIECES + 255) / 256;            // requests per thread per tile
  static constexpr int BUF_BYTES = NDMA * 256 * 16;
  static constexpr int LDS_BYTES = 2 * BUF_BYTES;
  static constexpr int NCH = (9 * CPT + 3) / 4;                // 32-wide k-steps
};

// 3x3 stride 1 pad 1 convolution, ONE source of 8 CPT channels, N <= 16 NT output channels, bias + optional ReLU.
// Tile = (4 MT) rows x 16 columns; wave w owns rows w MT .. w MT + MT - 1.  Needs H % (4 MT) == 0, W % 16 == 0.
template <int CPT, int NT, int MT>
__global__ __launch_bounds__(256, 1) void c3n_kernel(const NarrowParams p) {
  using G = C3nGeom<CPT, NT, MT>;
  constexpr int HC = G::HC, NCH = G::NCH, NDMA = G::NDMA;
  constexpr int XP = CPT * 16;                                  // pixel pitch in the halo image (bytes)
  extern __shared__ __attribute__((aligned(16))) char nsm[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = sgpr(tid >> 6);
  const int f = lane & 15, q = lane >> 4;

  // ---- the layer's weights: MFMA "A" operands (rows = output channels), resident for the whole kernel -------------------
  f32x4 wreg[NCH][NT];
  {
    const bf16_t* wp = reinterpret_cast<const bf16_t*>(p.w);
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        wreg[j][t] = *reinterpret_cast<const f32x4*>(wp + (size_t)(16 * t + f) * p.Kpad + 32 * j + 8 * q);
  }

  // ---- per-lane halo offsets of the pixel fragments: k-step j, lane group q -> octet 4 j + q = (tap, channel octet) ------
  int aoff[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int o = 4 * j + q;
    const int oo = o < 9 * CPT ? o : 0;
    const int tap = oo / CPT, c = oo - tap * CPT;
    const int ky = tap / 3, kx = tap - 3 * ky;
    aoff[j] = ((wave * MT + ky) * HC + f + kx) * XP + c * 16;
  }
  const bool tail_ok = 4 * (NCH - 1) + q < 9 * CPT;            // the last k-step may end in octets that do not exist (W is zero there)

  // ---- DMA pieces of this thread (tile-invariant): piece -> (halo pixel, octet) ---------------------------------------------
  unsigned voff[NDMA];
  int hyx[NDMA];
#pragma unroll
  for (int k = 0; k < NDMA; ++k) {
    const int pidx = k * 256 + tid;
    if (pidx < G::PIECES) {
      const int pix = pidx / CPT, oct = pidx - pix * CPT;
      const int hy = pix / HC, hx = pix - hy * HC;
      voff[k] = (unsigned)(((hy * p.W + hx) * p.ld + oct * 8) * 2);
      hyx[k] = (hy << 8) | hx;
    } else {
      voff[k] = 0;
      hyx[k] = -1;
    }
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)nsm;
  const int Hs = sgpr(p.H), Ws = sgpr(p.W), lds_ = sgpr(p.ld);

  auto tile_xy = [&](int t, int& b, int& y0, int& x0) {
    const int ts = xcd_tile(t, p.tiles_total);
    const int tx = ts % p.tiles_x;
    const int r = ts / p.tiles_x;
    const int ty = r % p.tiles_y;
    b = r / p.tiles_y;
    y0 = ty * G::TH;
    x0 = tx * 16;
  };
  auto stage = [&](int t, int buf) {                         // request tile t's halo into buffer `buf`
    int b, y0, x0;
    tile_xy(t, b, y0, x0);
    const char* sbase = reinterpret_cast<const char*>(p.src) + ((long)(b * Hs + y0 - 1) * Ws + (x0 - 1)) * (long)(lds_ * 2);
#pragma unroll
    for (int k = 0; k < NDMA; ++k) {
      if (hyx[k] >= 0) {
        const int iy = y0 - 1 + (hyx[k] >> 8), ix = x0 - 1 + (hyx[k] & 255);
        const bool ok = (unsigned)iy < (unsigned)Hs && (unsigned)ix < (unsigned)Ws;
        const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * G::BUF_BYTES + (k * 256 + wave * 64) * 16));
        if (ok) dma16(ldsw, voff[k], sbase);
        else *reinterpret_cast<f32x4*>(nsm + buf * G::BUF_BYTES + (k * 256 + tid) * 16) = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  const int en = q * 4;
  auto compute = [&](auto buf_tag, int t) {
    constexpr int BUF = decltype(buf_tag)::value;
    const char* hb = nsm + BUF * G::BUF_BYTES;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 a[2][MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) a[0][i] = *reinterpret_cast<const f32x4*>(hb + aoff[0] + i * (HC * XP));
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int cur = j & 1;
      if (j + 1 < NCH) {
#pragma unroll
        for (int i = 0; i < MT; ++i) a[cur ^ 1][i] = *reinterpret_cast<const f32x4*>(hb + aoff[j + 1] + i * (HC * XP));
      }
      __builtin_amdgcn_sched_barrier(0);                      // the next k-step's fragment reads stay above this k-step's MFMAs
      if (j == NCH - 1 && (9 * CPT) % 4 != 0) {
#pragma unroll
        for (int i = 0; i < MT; ++i) a[cur][i] = keep_if(a[cur][i], tail_ok);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[i][n] = mfma_stage<bf16_t>(wreg[j][n], a[cur][i], acc[i][n]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- epilogue: lane = pixel column f of row i, channels 16 n + 4 q .. + 3 ---------------------------------------------
    int b, y0, x0;
    tile_xy(t, b, y0, x0);
    const size_t pix0 = ((size_t)(b * Hs + y0 + wave * MT) * Ws + x0 + f);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int ch = 16 * n + en;
      if (ch >= p.N) continue;
      f32x4 sh = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (p.shift) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sh[r] = ch + r < p.N ? p.shift[ch + r] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        f32x4 v = acc[i][n] + sh;
        if (p.act == CCVPE_ACT_RELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        const size_t o = (pix0 + (size_t)i * Ws) * p.ldd + ch;
        if (ch + 3 < p.N) {
          if (p.out_f32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.dst) + o) = v;
          else {
            bf16x4 ob;
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[r] = (bf16_t)v[r];
            *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.dst) + o) = ob;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (ch + r < p.N) {
              if (p.out_f32) reinterpret_cast<float*>(p.dst)[o + r] = v[r];
              else reinterpret_cast<bf16_t*>(p.dst)[o + r] = (bf16_t)v[r];
            }
        }
      }
    }
  };
  auto tile_end = [&]() {                                     // next tile's halo has landed (this wave's requests) + everyone is done reading
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  int t = blockIdx.x;
  const int step = gridDim.x;
  if (t >= p.tiles_total) return;
  stage(t, 0);
  tile_end();
  while (true) {
    int t1 = t + step;
    if (t1 < p.tiles_total) stage(t1, 1);
    compute(std::integral_constant<int, 0>{}, t);
    tile_end();
    t = t1;
    if (t >= p.tiles_total) break;
    t1 = t + step;
    if (t1 < p.tiles_total) stage(t1, 0);
    compute(std::integral_constant<int, 1>{}, t);
    tile_end();
    t = t1;
    if (t >= p.tiles_total) break;
  }
}

int num_cus();   // narrow_bf16.hip

template <int CPT, int NT, int MT>
static int launch_c3n(NarrowParams p, int batch, hipStream_t stream) {
  using G = C3nGeom<CPT, NT, MT>;
  p.tiles_x = p.W / 16;
  p.tiles_y = p.H / G::TH;
  const long total = (long)p.tiles_x * p.tiles_y * batch;
  if (total > 0x7fffffffL) return fail(CCVPE_EINVAL, "c3n: grid too large");
  p.tiles_total = (int)total;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)c3n_kernel<CPT, NT, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    if (e != hipSuccess) return fail(CCVPE_ELAUNCH, "c3n_kernel: set smem attr: %s", hipGetErrorString(e));
    attr_set = true;
  }
  const int grid = (int)(total < num_cus() ? total : num_cus());
  hipLaunchKernelGGL((c3n_kernel<CPT, NT, MT>), dim3(grid), dim3(256), G::LDS_BYTES, stream, p);
  return check_launch("c3n_kernel");
}

}  // namespace ccvpe
