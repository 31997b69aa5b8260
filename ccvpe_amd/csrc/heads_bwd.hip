// Backward of the small head / glue operators of the path (fp32):
//   softmax over the flattened heat-map (models.py:320), F.normalize of the (cos, sin) map (:341),
//   the 16 -> {1,2} 3x3 head convs (conv1.2 / conv1_ori.2), the ground-descriptor height collapse
//   (models.py:57-97), the stem conv's weight gradient (efficientnet_pytorch/model.py:181,289) and the
//   strided column accumulate used to route skip-connection gradients.
// All reductions write per-workgroup partials that are summed in a fixed order (deterministic).
#include "common.h"

namespace ccvpe {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float block_sum_b(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;
}

// dlogits = h * (dh - <h, dh>) (+ dlogits_direct)
__global__ __launch_bounds__(1024) void softmax_bwd_kernel(const float* __restrict__ h, const float* __restrict__ dh,
                                                           const float* __restrict__ dl_direct,
                                                           float* __restrict__ dlogits, int n) {
  __shared__ float sh[16];
  const size_t base = (size_t)blockIdx.x * n;
  const f32x4* h4 = reinterpret_cast<const f32x4*>(h + base);
  const f32x4* d4 = reinterpret_cast<const f32x4*>(dh + base);
  float acc = 0.f;
  for (int i = threadIdx.x; i < n / 4; i += 1024) {
    const f32x4 a = h4[i], b = d4[i];
    acc += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  }
  const float dot = block_sum_b(acc, sh);
  f32x4* o4 = reinterpret_cast<f32x4*>(dlogits + base);
  for (int i = threadIdx.x; i < n / 4; i += 1024) {
    f32x4 v = h4[i] * (d4[i] - dot);
    if (dl_direct) v += reinterpret_cast<const f32x4*>(dl_direct + base)[i];
    o4[i] = v;
  }
}

// r [B,2,HW] raw head output, d_o gradient w.r.t. r / max(|r|, 1e-12)  ->  dr
__global__ __launch_bounds__(256) void l2norm2_bwd_kernel(const float* __restrict__ r, const float* __restrict__ d_o,
                                                          float* __restrict__ dr, int B, int HW) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * HW) return;
  const int b = (int)(i / HW), p = (int)(i % HW);
  const size_t i0 = ((size_t)b * 2) * HW + p, i1 = i0 + HW;
  const float r0 = r[i0], r1 = r[i1], g0 = d_o[i0], g1 = d_o[i1];
  const float n = sqrtf(r0 * r0 + r1 * r1);
  if (n > 1e-12f) {
    const float o0 = r0 / n, o1 = r1 / n, dot = o0 * g0 + o1 * g1;
    dr[i0] = (g0 - o0 * dot) / n;
    dr[i1] = (g1 - o1 * dot) / n;
  } else {
    dr[i0] = g0 * 1e12f;
    dr[i1] = g1 * 1e12f;
  }
}

// dx[b,y,x,c] = sum_{o,ky,kx} dr[b,o,y+1-ky,x+1-kx] * w[o][ky][kx][c]
// Taps outside the image read a clamped address and are multiplied by 0 (no per-tap branch, so the nine loads of
// a plane issue back to back).  RELU: the convolution's input is a ReLU output, and the gradient is wanted in
// front of that ReLU: dx = x > 0 ? dx : 0 (what relu_bwd would do in a second pass over the tensor).
template <int COUT, bool RELU>
__global__ __launch_bounds__(256) void head_dgrad_kernel(const float* __restrict__ dr, const float* __restrict__ w,
                                                         const float* __restrict__ xin, float* __restrict__ dx, int B, int H,
                                                         int W) {
  __shared__ __attribute__((aligned(16))) float ws[COUT * 9 * 16];
  for (int i = threadIdx.x; i < COUT * 9 * 16; i += 256) ws[i] = w[i];
  __syncthreads();
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * H * W) return;
  const int x = (int)(idx % W), y = (int)((idx / W) % H), b = (int)(idx / ((long)W * H));
  f32x4 keep[4];
  if constexpr (RELU) {
#pragma unroll
    for (int q = 0; q < 4; ++q) keep[q] = *reinterpret_cast<const f32x4*>(xin + (size_t)idx * 16 + q * 4);
  }
  f32x4 acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float g[COUT][9];
#pragma unroll
  for (int o = 0; o < COUT; ++o) {
    const float* plane = dr + ((size_t)b * COUT + o) * H * W;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y + 1 - ky;
      const bool yok = (unsigned)yy < (unsigned)H;
      const int yc = yok ? yy : y;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = x + 1 - kx;
        const bool ok = yok && (unsigned)xx < (unsigned)W;
        const float v = plane[(size_t)yc * W + (ok ? xx : x)];
        g[o][ky * 3 + kx] = ok ? v : 0.f;
      }
    }
  }
#pragma unroll
  for (int o = 0; o < COUT; ++o)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] += g[o][t] * *reinterpret_cast<const f32x4*>(&ws[(o * 9 + t) * 16 + q * 4]);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 v = acc[q];
    if constexpr (RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = keep[q][e] > 0.f ? v[e] : 0.f;
    }
    *reinterpret_cast<f32x4*>(dx + (size_t)idx * 16 + q * 4) = v;
  }
}

// dw[o][ky][kx][c] = sum_px dr[b,o,px] * x[px + (ky-1,kx-1)][c] ; db[o] = sum_px dr[b,o,px]
constexpr int HW_TW = 64, HW_TH = 4, HW_LD = 17;
template <int COUT>
__global__ __launch_bounds__(256) void head_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dr,
                                                         float* __restrict__ part, int B, int H, int W, int ntiles) {
  __shared__ float tile[(HW_TH + 2) * (HW_TW + 2) * HW_LD];
  __shared__ float drs[COUT * HW_TH * HW_TW];
  const int tid = threadIdx.x;
  const int tap = tid >> 4, c = tid & 15;          // tid < 144: weight (tap, c); tid == 144: bias
  const int ky = tap / 3, kx = tap - ky * 3;
  const int nxb = (W + HW_TW - 1) / HW_TW, nyb = (H + HW_TH - 1) / HW_TH;
  float acc[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int x0 = (t % nxb) * HW_TW, y0 = ((t / nxb) % nyb) * HW_TH, b = t / (nxb * nyb);
    __syncthreads();
    constexpr int NPX = (HW_TH + 2) * (HW_TW + 2);
    for (int idx = tid; idx < NPX * 4; idx += 256) {
      const int pxl = idx >> 2, q = idx & 3;
      const int hy = pxl / (HW_TW + 2), hx = pxl - hy * (HW_TW + 2);
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
        v = *reinterpret_cast<const f32x4*>(x + (((size_t)b * H + iy) * W + ix) * 16 + q * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) tile[pxl * HW_LD + q * 4 + j] = v[j];
    }
    for (int idx = tid; idx < COUT * HW_TH * HW_TW; idx += 256) {
      const int o = idx / (HW_TH * HW_TW), p = idx % (HW_TH * HW_TW);
      const int oy = y0 + p / HW_TW, ox = x0 + p % HW_TW;
      drs[idx] = (oy < H && ox < W) ? dr[(((size_t)b * COUT + o) * H + oy) * W + ox] : 0.f;
    }
    __syncthreads();
    if (tid < 144) {
      for (int p = 0; p < HW_TH * HW_TW; ++p) {
        const int ty = p / HW_TW, tx = p % HW_TW;
        const float xv = tile[((ty + ky) * (HW_TW + 2) + tx + kx) * HW_LD + c];
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] = fmaf(drs[o * HW_TH * HW_TW + p], xv, acc[o]);
      }
    } else if (tid == 144) {
      for (int p = 0; p < HW_TH * HW_TW; ++p) {
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] += drs[o * HW_TH * HW_TW + p];
      }
    }
  }
  // partial layout per block: [COUT][145]  (144 weights in (ky,kx,c) order, then the bias)
  if (tid <= 144) {
#pragma unroll
    for (int o = 0; o < COUT; ++o) part[((size_t)blockIdx.x * COUT + o) * 145 + tid] = acc[o];
  }
}

struct GdescBwdCfg {
  int cd[6], off[6], obase[6];
};

// dy1[b,y,x,off_l+c] = wh_l[y] * dD[b, obase_l + x*cd_l + c]   (pad columns -> 0)
__global__ __launch_bounds__(256) void gdesc_bwd_data_kernel(const float* __restrict__ dD, const float* __restrict__ wh,
                                                             const GdescBwdCfg cfg, float* __restrict__ dy1, int B, int h,
                                                             int w, int ld, int ctot) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * h * w * ld) return;
  const int n = (int)(idx % ld);
  long r = idx / ld;
  const int xx = (int)(r % w);
  r /= w;
  const int yy = (int)(r % h);
  const int b = (int)(r / h);
  float v = 0.f;
  if (n < ctot) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < 6; ++i)
      if (n >= cfg.off[i]) l = i;
    v = wh[l * h + yy] * dD[(size_t)b * w * ctot + cfg.obase[l] + xx * cfg.cd[l] + (n - cfg.off[l])];
  }
  dy1[idx] = v;
}

// dwh[l][y] = sum_{b,x,c} y1[b,y,x,off_l+c] * dD[...] ; dbh[l] = sum_{b,x,c} dD[...]   (one workgroup per (l,y))
__global__ __launch_bounds__(256) void gdesc_bwd_w_kernel(const float* __restrict__ y1, const float* __restrict__ dD,
                                                          const GdescBwdCfg cfg, float* __restrict__ dwh,
                                                          float* __restrict__ dbh, int B, int h, int w, int ld, int ctot) {
  __shared__ float sh[4];
  const int l = blockIdx.x / h, yy = blockIdx.x % h;
  const int cd = cfg.cd[l];
  const int per = w * cd;
  float a = 0.f, s = 0.f;
  for (int i = threadIdx.x; i < B * per; i += 256) {
    const int b = i / per, rel = i % per;
    const int xx = rel / cd, c = rel % cd;
    const float g = dD[(size_t)b * w * ctot + cfg.obase[l] + rel];
    a = fmaf(y1[(((size_t)b * h + yy) * w + xx) * ld + cfg.off[l] + c], g, a);
    s += g;
  }
  a = block_sum_b(a, sh);
  s = block_sum_b(s, sh);
  if (threadIdx.x == 0) {
    dwh[l * h + yy] = a;
    if (yy == 0) dbh[l] = s;
  }
}

// dst[row, 0:C] (+)= src[row, off:off+C]
__global__ __launch_bounds__(256) void add_cols_kernel(const float* __restrict__ src, int lds_, int off,
                                                       float* __restrict__ dst, int ldd, int C, long rows,
                                                       int accumulate) {
  const int cg4 = C >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * cg4) return;
  const long row = idx / cg4;
  const int c = (int)(idx % cg4) * 4;
  f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)row * lds_ + off + c);
  f32x4* d = reinterpret_cast<f32x4*>(dst + (size_t)row * ldd + c);
  if (accumulate) v += *d;
  *d = v;
}

// stem weight gradient: dw[ky][kx][ci][co] = sum_{b,oy,ox} dy[b,oy,ox,co] * xpad[b,ci,2oy+ky,2ox+kx]
// A workgroup owns SW_ROWS output rows x SW_COLS output columns of one sample.  Per output row the three input rows it
// touches (3 channels, 2*SW_COLS + 1 columns, zero / wrapped past the right edge) are staged in LDS with coalesced loads;
// 256 threads = 8 groups of 4 output channels x 32 pixel lanes, 27 x 4 accumulators each: a pixel costs one 16-byte dY
// load, 18 LDS reads (broadcast over the channel groups, stride-2 over the pixel lanes: conflict-free) and 108 FMAs.
// (The first version read the 27 image values of every pixel from global memory, 2 distinct addresses per wave
// instruction: 1.2 ms per launch at B = 64, 2.4 ms of a training step.)
constexpr int SW_ROWS = 16, SW_COLS = 256, SW_XLD = 2 * SW_COLS + 4;
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ part, int B, int H, int W, int Ho, int Wo,
                                                         int circular, int row_groups, int col_groups) {
  __shared__ __attribute__((aligned(16))) float xs[9 * SW_XLD];      // [ci*3 + ky][column]
  __shared__ __attribute__((aligned(16))) float red[32 * 32];
  const int tid = threadIdx.x;
  const int cg = tid & 7, pl = tid >> 3;                              // channel group (4 output channels), pixel lane
  int blk = blockIdx.x;
  const int cgi = blk % col_groups; blk /= col_groups;
  const int rgi = blk % row_groups;
  const int b = blk / row_groups;
  const int x0 = cgi * SW_COLS;
  const int ncols = min(SW_COLS, Wo - x0);
  f32x4 acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const size_t plane = (size_t)H * W;
  for (int oy = rgi * SW_ROWS; oy < min((rgi + 1) * SW_ROWS, Ho); ++oy) {
    __syncthreads();                                                  // previous row's readers are done
    for (int idx = tid; idx < 9 * (2 * SW_COLS + 1); idx += 256) {
      const int r = idx / (2 * SW_COLS + 1), col = idx - r * (2 * SW_COLS + 1);
      const int ci = r / 3, ky = r - ci * 3;
      const int iy = 2 * oy + ky;
      int ix = 2 * x0 + col;
      bool ok = iy < H;
      if (ix >= W) {
        if (circular) ix -= W; else ok = false;
      }
      ok = ok && ix < W;
      const float v = x[((size_t)(b * 3 + ci)) * plane + (size_t)(ok ? iy : 0) * W + (ok ? ix : 0)];
      xs[r * SW_XLD + col] = ok ? v : 0.f;
    }
    __syncthreads();
    const float* dyrow = dy + ((size_t)(b * Ho + oy) * Wo + x0) * 32 + cg * 4;
#pragma unroll 2
    for (int ox = pl; ox < ncols; ox += 32) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(dyrow + (size_t)ox * 32);
#pragma unroll
      for (int r = 0; r < 9; ++r) {                                   // r = ci*3 + ky
        const float* xp = &xs[r * SW_XLD + 2 * ox];
        const float2 x01 = *reinterpret_cast<const float2*>(xp);
        const float x2 = xp[2];
        const int ci = r / 3, ky = r - ci * 3;
        acc[(ky * 3 + 0) * 3 + ci] += g * x01.x;
        acc[(ky * 3 + 1) * 3 + ci] += g * x01.y;
        acc[(ky * 3 + 2) * 3 + ci] += g * x2;
      }
    }
  }
  // fixed-order reduction over the 32 pixel lanes, one tap at a time (unrolled: acc[] must stay in registers)
#pragma unroll
  for (int t = 0; t < 27; ++t) {
    __syncthreads();
    *reinterpret_cast<f32x4*>(&red[pl * 32 + cg * 4]) = acc[t];
    __syncthreads();
    if (tid < 32) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int q = 0; q < 32; q += 2) {
        s0 += red[q * 32 + tid];
        s1 += red[(q + 1) * 32 + tid];
      }
      part[((size_t)blockIdx.x * 27 + t) * 32 + tid] = s0 + s1;
    }
  }
}

}  // namespace ccvpe

using namespace ccvpe;

extern "C" int ccvpe_softmax_bwd_f32(const float* heatmap, const float* dheatmap, const float* dlogits_direct,
                                     float* dlogits, int rows, int n, void* stream) {
  if (rows <= 0 || n <= 0 || n % 4) return fail(CCVPE_EINVAL, "softmax_bwd: bad shape");
  if (!aligned16(heatmap) || !aligned16(dheatmap) || !aligned16(dlogits) || (dlogits_direct && !aligned16(dlogits_direct)))
    return fail(CCVPE_EINVAL, "softmax_bwd: 16-byte aligned rows required");
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(rows), dim3(1024), 0, (hipStream_t)stream, heatmap, dheatmap, dlogits_direct,
                     dlogits, n);
  return check_launch("softmax_bwd_kernel");
}

extern "C" int ccvpe_l2norm2_bwd_f32(const float* raw, const float* dout, float* draw, int batch, int hw, void* stream) {
  if (batch <= 0 || hw <= 0) return fail(CCVPE_EINVAL, "l2norm2_bwd: bad shape");
  const long n = (long)batch * hw;
  hipLaunchKernelGGL(l2norm2_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, raw, dout, draw,
                     batch, hw);
  return check_launch("l2norm2_bwd_kernel");
}

extern "C" int ccvpe_head_conv3x3_bwd_f32(const float* x, const float* w, const float* dout, float* dx, float* dw, float* dbias,
                                          float* scratch, int batch, int h, int wd, int cout, int relu_mask_x, void* stream) {
  if (batch <= 0 || h <= 0 || wd <= 0 || (cout != 1 && cout != 2)) return fail(CCVPE_EINVAL, "head_conv_bwd: bad shape");
  if (!aligned16(x) || !aligned16(dx)) return fail(CCVPE_EINVAL, "head_conv_bwd: 16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  const long npx = (long)batch * h * wd;
  const int ntiles = ((wd + HW_TW - 1) / HW_TW) * ((h + HW_TH - 1) / HW_TH) * batch;
  const int nblk = ntiles < CCVPE_HEAD_WGRAD_BLOCKS ? ntiles : CCVPE_HEAD_WGRAD_BLOCKS;
  const dim3 dgrid((unsigned)((npx + 255) / 256));
#define CCVPE_HEAD_DGRAD(CO)                                                                                          \
  do {                                                                                                                \
    if (relu_mask_x)                                                                                                  \
      hipLaunchKernelGGL((head_dgrad_kernel<CO, true>), dgrid, dim3(256), 0, st, dout, w, x, dx, batch, h, wd);        \
    else                                                                                                              \
      hipLaunchKernelGGL((head_dgrad_kernel<CO, false>), dgrid, dim3(256), 0, st, dout, w, x, dx, batch, h, wd);       \
    hipLaunchKernelGGL((head_wgrad_kernel<CO>), dim3(nblk), dim3(256), 0, st, x, dout, scratch, batch, h, wd, ntiles); \
  } while (0)
  if (cout == 1)
    CCVPE_HEAD_DGRAD(1);
  else
    CCVPE_HEAD_DGRAD(2);
#undef CCVPE_HEAD_DGRAD
  // partials [nblk][cout][145] -> tmp [cout][145] (second half of scratch), then split into dw / dbias
  float* tmp = scratch + (size_t)CCVPE_HEAD_WGRAD_BLOCKS * 2 * 145;
  launch_sum_parts(scratch, nblk, cout * 145, cout * 145, tmp, st);
  for (int o = 0; o < cout; ++o) {
    if (hipMemcpyAsync(dw + o * 144, tmp + o * 145, 144 * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemcpyAsync(dbias + o, tmp + o * 145 + 144, sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
      return fail(CCVPE_ELAUNCH, "head_conv_bwd: copy failed");
  }
  return check_launch("head_conv_bwd");
}

extern "C" int ccvpe_ground_descriptor_bwd_f32(const float* y1, int ld, const float* wh, const int* cd, const float* dout,
                                               float* dy1, float* dwh, float* dbh, int batch, int h, int w, void* stream) {
  if (batch <= 0 || h <= 0 || w <= 0 || ld <= 0) return fail(CCVPE_EINVAL, "ground_descriptor_bwd: bad shape");
  GdescBwdCfg cfg;
  int off = 0;
  for (int l = 0; l < 6; ++l) {
    if (cd[l] <= 0) return fail(CCVPE_EINVAL, "ground_descriptor_bwd: cd must be positive");
    cfg.cd[l] = cd[l];
    cfg.off[l] = off;
    cfg.obase[l] = w * off;
    off += cd[l];
  }
  if (off > ld) return fail(CCVPE_EINVAL, "ground_descriptor_bwd: sum(cd) > ld");
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)batch * h * w * ld;
  hipLaunchKernelGGL(gdesc_bwd_data_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dout, wh, cfg, dy1, batch, h, w,
                     ld, off);
  hipLaunchKernelGGL(gdesc_bwd_w_kernel, dim3(6 * h), dim3(256), 0, st, y1, dout, cfg, dwh, dbh, batch, h, w, ld, off);
  return check_launch("ground_descriptor_bwd");
}

extern "C" int ccvpe_add_cols_f32(const float* src, int ld_src, int col_off, float* dst, int ld_dst, int channels, int rows,
                                  int accumulate, void* stream) {
  if (rows <= 0 || channels <= 0 || channels % 4 || col_off % 4 || ld_src % 4 || ld_dst % 4)
    return fail(CCVPE_EINVAL, "add_cols: channels, offsets and strides must be multiples of 4");
  if (!aligned16(src) || !aligned16(dst)) return fail(CCVPE_EINVAL, "add_cols: 16-byte alignment");
  const long n = (long)rows * (channels / 4);
  hipLaunchKernelGGL(add_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, ld_src, col_off,
                     dst, ld_dst, channels, (long)rows, accumulate);
  return check_launch("add_cols_kernel");
}

static void stem_wgrad_groups(int in_h, int in_w, int* rg, int* cgp) {
  const int Ho = (in_h - 2) / 2 + 1, Wo = (in_w - 2) / 2 + 1;     // (H + 1 - 3) / 2 + 1, also for odd sizes
  *rg = (Ho + SW_ROWS - 1) / SW_ROWS;
  *cgp = (Wo + SW_COLS - 1) / SW_COLS;
}
extern "C" int ccvpe_stem_wgrad_nblk(int batch, int in_h, int in_w) {
  int rg, cgp;
  stem_wgrad_groups(in_h, in_w, &rg, &cgp);
  return batch * rg * cgp;
}

extern "C" int ccvpe_stem_conv_wgrad_f32(const float* x_nchw, const float* dy, float* dw, float* scratch, int batch, int in_h,
                                         int in_w, int circular, void* stream) {
  if (batch <= 0 || in_h < 2 || in_w < 2) return fail(CCVPE_EINVAL, "stem_wgrad: bad shape");
  if (!aligned16(dy)) return fail(CCVPE_EINVAL, "stem_wgrad: dy must be 16-byte aligned");
  int rg, cgp;
  stem_wgrad_groups(in_h, in_w, &rg, &cgp);
  const int nblk = batch * rg * cgp;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3(nblk), dim3(256), 0, st, x_nchw, dy, scratch, batch, in_h, in_w, (in_h - 2) / 2 + 1,
                     (in_w - 2) / 2 + 1, circular, rg, cgp);
  launch_sum_parts(scratch, nblk, 27 * 32, 27 * 32, dw, st);
  return check_launch("stem_wgrad_kernel");
}

// ---------------------------------------------------------------------------------------------
// Loss backward (losses.py:4-29), gradient w.r.t. the prediction only (labels are data).  `dloss` is the
// upstream scalar gradient on the device.  One workgroup per sample, row statistics recomputed.
//   infoNCE : dL/ds_j   = (den_b * p_j - lab_j [lab_j > 1e-2]) / (D * T),  p = softmax(s / T) of the row,
//                         den_b = masked label sum of the row, D = sum_b den_b
//   CE      : dL/dz_j   = (softmax_j * sum_i lab_i - lab_j) / B
//   ori     : dL/do_c,i = -2 (gt_o - o) * gt_i / B
// ---------------------------------------------------------------------------------------------
namespace ccvpe {

// one pass: the row statistics (z_b, den_b) and the batch label mass D come from the forward (rows[b] = (num, den, z, 0),
// rows[4B] = D: csrc/heads.hip), workgroup (chunk, sample) writes its NCE_BCH gradients
constexpr int NCE_BCH = 8192;
__global__ __launch_bounds__(256) void infonce_bwd_kernel(const float* __restrict__ sc, const float* __restrict__ lab,
                                                          float inv_t, const float* __restrict__ dloss,
                                                          const float* __restrict__ rows, float* __restrict__ dsc, int n, int B) {
  const size_t row = (size_t)blockIdx.y * n;
  const float den = rows[4 * blockIdx.y + 1], rz = 1.0f / rows[4 * blockIdx.y + 2];
  // (an upstream gradient of exactly 0 — a rank without label mass under the global-ratio loss, harness._GlobalRatio —
  // gives 0, not 0 / 0)
  const float g = dloss[0];
  const float k = g == 0.f ? 0.f : g * inv_t / rows[4 * B];
  const int i0 = blockIdx.x * NCE_BCH, i1 = min(n, i0 + NCE_BCH);
  if ((n & 3) == 0) {
    for (int i = i0 + threadIdx.x * 4; i < i1; i += 1024) {
      const cc_f32x4 v = *reinterpret_cast<const cc_f32x4*>(sc + row + i);
      const cc_f32x4 l = *reinterpret_cast<const cc_f32x4*>(lab + row + i);
      cc_f32x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = k * (den * expf(v[q] * inv_t) * rz - (l[q] > 1e-2f ? l[q] : 0.f));
      *reinterpret_cast<cc_f32x4*>(dsc + row + i) = o;
    }
  } else {
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
      const float lv = lab[row + i];
      dsc[row + i] = k * (den * expf(sc[row + i] * inv_t) * rz - (lv > 1e-2f ? lv : 0.f));
    }
  }
}

__global__ __launch_bounds__(1024) void ce_bwd_kernel(const float* __restrict__ lg, const float* __restrict__ lab,
                                                      const float* __restrict__ dloss, float* __restrict__ dlg, int n,
                                                      int B) {
  __shared__ float sh[16];
  const float* r = lg + (size_t)blockIdx.x * n;
  const float* l = lab + (size_t)blockIdx.x * n;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, r[i]);
  m = wave_max(m);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, sh[i]);
  float z = 0.f, ls = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    z += expf(r[i] - m);
    ls += l[i];
  }
  z = block_sum_b(z, sh);
  ls = block_sum_b(ls, sh);
  const float k = dloss[0] / (float)B;
  float* o = dlg + (size_t)blockIdx.x * n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) o[i] = k * (ls * expf(r[i] - m) / z - l[i]);
}

__global__ __launch_bounds__(256) void ori_bwd_kernel(const float* __restrict__ ori, const float* __restrict__ gto,
                                                      const float* __restrict__ gt, const float* __restrict__ dloss,
                                                      float* __restrict__ dori, int hw, int B) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * hw) return;
  const int b = (int)(idx / hw), i = (int)(idx % hw);
  const float k = -2.0f * dloss[0] / (float)B * gt[(size_t)b * hw + i];
  const size_t i0 = (size_t)b * 2 * hw + i, i1 = i0 + hw;
  dori[i0] = k * (gto[i0] - ori[i0]);
  dori[i1] = k * (gto[i1] - ori[i1]);
}

}  // namespace ccvpe

extern "C" int ccvpe_infonce_loss_bwd_f32(const float* scores, const float* labels, float temperature, const float* dloss,
                                          const float* rows, float* dscores, int batch, int n, void* stream) {
  if (batch <= 0 || batch > 65535 || n <= 0 || temperature <= 0.f) return fail(CCVPE_EINVAL, "infonce_bwd: bad args");
  if ((n & 3) == 0 && (!aligned16(scores) || !aligned16(labels) || !aligned16(dscores)))
    return fail(CCVPE_EINVAL, "infonce_bwd: 16-byte alignment required");
  const int chunks = (n + NCE_BCH - 1) / NCE_BCH;
  hipLaunchKernelGGL(infonce_bwd_kernel, dim3(chunks, batch), dim3(256), 0, (hipStream_t)stream, scores, labels, 1.0f / temperature,
                     dloss, rows, dscores, n, batch);
  return check_launch("infonce_bwd_kernel");
}

extern "C" int ccvpe_cross_entropy_loss_bwd_f32(const float* logits, const float* labels, const float* dloss, float* dlogits,
                                                int batch, int n, void* stream) {
  if (batch <= 0 || n <= 0) return fail(CCVPE_EINVAL, "cross_entropy_bwd: bad args");
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(batch), dim3(1024), 0, (hipStream_t)stream, logits, labels, dloss, dlogits, n, batch);
  return check_launch("ce_bwd_kernel");
}

extern "C" int ccvpe_orientation_loss_bwd_f32(const float* ori, const float* gt_ori, const float* gt, const float* dloss,
                                              float* dori, int batch, int hw, void* stream) {
  if (batch <= 0 || hw <= 0) return fail(CCVPE_EINVAL, "orientation_loss_bwd: bad args");
  const long n = (long)batch * hw;
  hipLaunchKernelGGL(ori_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ori, gt_ori, gt, dloss,
                     dori, hw, batch);
  return check_launch("ori_bwd_kernel");
}
