// Small HBM/latency-bound ends of the path: ground-descriptor height collapse, the final 3x3 convs
// to 1 / 2 channels (NCHW out, fused L2-normalise), the 262144-way softmax, and the three losses.
#include "common.h"

namespace ccvpe {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// D_l[b, x*Cd_l + c] = sum_y wh_l[y] * y1[b,y,x,off_l + c] + bh_l          (models.py:57-97)
// ---------------------------------------------------------------------------------------------
struct GdescCfg {
  int cd[6];
  int off[6];    // channel offset of level l inside y1
  int obase[6];  // output offset of level l (= w * off[l])
};

__global__ __launch_bounds__(256) void gdesc_kernel(const float* __restrict__ y1, int ld, const float* __restrict__ wh,
                                                    const float* __restrict__ bh, const GdescCfg cfg,
                                                    float* __restrict__ out, int h, int w, int ctot) {
  const int b = blockIdx.y;
  const int o = blockIdx.x * 256 + threadIdx.x;
  const int per_sample = w * ctot;
  if (o >= per_sample) return;
  int l = 0;
#pragma unroll
  for (int i = 1; i < 6; ++i)
    if (o >= cfg.obase[i]) l = i;
  const int rel = o - cfg.obase[l];
  const int xx = rel / cfg.cd[l];
  const int c = rel - xx * cfg.cd[l];
  float s = 0.f;
  for (int yy = 0; yy < h; ++yy)
    s = fmaf(wh[l * h + yy], y1[((size_t)(b * h + yy) * w + xx) * ld + cfg.off[l] + c], s);
  out[(size_t)b * per_sample + o] = s + bh[l];
}

// ---------------------------------------------------------------------------------------------
// 3x3 conv 16 -> COUT (1 or 2), pad 1, NHWC in, NCHW out, optional 2-vector normalise.
// models.py:125-127 (conv1.2), :146-148 + :341 (conv1_ori.2 + F.normalize)
// HBM-bound: 64 B read, 4-8 B written per pixel.  One thread per pixel, lanes along W.
// ---------------------------------------------------------------------------------------------
// Workgroup = 4 rows x 64 columns of output; the (4+2) x (64+2) x 16-channel input halo is loaded
// coalesced into LDS ONCE (a first version issued 36 global loads per pixel and sat at 1.7 TB/s,
// TA-bound) and the 9 taps are read from LDS.  Pixel rows are padded 16 -> 20 floats.
constexpr int HC_TW = 64, HC_TH = 4, HC_LD = 20;

template <typename TX, int COUT>
__global__ __launch_bounds__(256) void head_conv_kernel(const TX* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        int H, int W, int normalize) {
  __shared__ __attribute__((aligned(16))) float ws[COUT * 9 * 16];
  __shared__ __attribute__((aligned(16))) float tile[(HC_TH + 2) * (HC_TW + 2) * HC_LD];
  for (int i = threadIdx.x; i < COUT * 9 * 16; i += 256) ws[i] = w[i];
  // XCD-aware order (see dwconv_kernel): neighbouring tiles of a sample go to the same XCD's L2.
  const int nxb = (W + HC_TW - 1) / HC_TW;
  const int nyb = (H + HC_TH - 1) / HC_TH;
  const int total = gridDim.x;
  int lb;
  {
    const int q = total / 8, r = total % 8;
    const int xcd = blockIdx.x % 8, loc = blockIdx.x / 8;
    lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int x0 = (lb % nxb) * HC_TW;
  const int y0 = ((lb / nxb) % nyb) * HC_TH;
  const int b = lb / (nxb * nyb);
  const TX* xb = x + (size_t)b * H * W * 16;
  // halo load: (HC_TH+2)*(HC_TW+2) pixels x 4 quads of 4 channels
  constexpr int NPX = (HC_TH + 2) * (HC_TW + 2);
  for (int idx = threadIdx.x; idx < NPX * 4; idx += 256) {
    const int pxl = idx >> 2, q = idx & 3;
    const int hy = pxl / (HC_TW + 2), hx = pxl - hy * (HC_TW + 2);
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = ld4<TX>(xb + ((size_t)iy * W + ix) * 16 + q * 4);
    *reinterpret_cast<f32x4*>(&tile[pxl * HC_LD + q * 4]) = v;
  }
  __syncthreads();
  const int tx = threadIdx.x & (HC_TW - 1), ty = threadIdx.x / HC_TW;
  const int ox = x0 + tx, oy = y0 + ty;
  if (ox >= W || oy >= H) return;
  float acc[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) acc[o] = bias[o];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const float* px = &tile[((ty + ky) * (HC_TW + 2) + tx + kx) * HC_LD];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(px + q * 4);
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(&ws[((o * 3 + ky) * 3 + kx) * 16 + q * 4]);
          acc[o] = fmaf(v[0], wv[0], acc[o]);
          acc[o] = fmaf(v[1], wv[1], acc[o]);
          acc[o] = fmaf(v[2], wv[2], acc[o]);
          acc[o] = fmaf(v[3], wv[3], acc[o]);
        }
      }
    }
  }
  if (COUT == 2 && normalize) {
    const float n = fmaxf(sqrtf(acc[0] * acc[0] + acc[COUT - 1] * acc[COUT - 1]), 1e-12f);
    acc[0] /= n;
    acc[COUT - 1] /= n;
  }
#pragma unroll
  for (int o = 0; o < COUT; ++o) out[((size_t)(b * COUT + o) * H + oy) * W + ox] = acc[o];
}

// ---------------------------------------------------------------------------------------------
// block-wide reductions (1024 threads = 16 waves)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = sh[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) t = fmaxf(t, sh[i]);
  return t;
}

// Softmax over a row (models.py:319-320).  One workgroup per row; the 1 MiB row stays L2-resident
// between the three sweeps.
__global__ __launch_bounds__(1024) void softmax_rows_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            int n) {
  __shared__ float sh[16];
  const float* r = in + (size_t)blockIdx.x * n;
  float* o = out + (size_t)blockIdx.x * n;
  const int n4 = n >> 2;
  const f32x4* r4 = reinterpret_cast<const f32x4*>(r);
  float m = -INFINITY;
  for (int i = threadIdx.x; i < n4; i += blockDim.x) {
    const f32x4 v = r4[i];
    m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
  }
  for (int i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, r[i]);
  m = block_max(m, sh);
  float s = 0.f;
  for (int i = threadIdx.x; i < n4; i += blockDim.x) {
    const f32x4 v = r4[i];
    s += expf(v[0] - m) + expf(v[1] - m) + expf(v[2] - m) + expf(v[3] - m);
  }
  for (int i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) s += expf(r[i] - m);
  s = block_sum(s, sh);
  const float inv = 1.0f / s;
  f32x4* o4 = reinterpret_cast<f32x4*>(o);
  for (int i = threadIdx.x; i < n4; i += blockDim.x) {
    const f32x4 v = r4[i];
    f32x4 e;
    e[0] = expf(v[0] - m) * inv; e[1] = expf(v[1] - m) * inv;
    e[2] = expf(v[2] - m) * inv; e[3] = expf(v[3] - m) * inv;
    o4[i] = e;
  }
  for (int i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) o[i] = expf(r[i] - m) * inv;
}

// Softmax from per-piece partials (ccvpe_tail512 writes (max, sum exp(l - max)) for every piece of a row it produced):
// workgroup (slice, row) merges the row's P partial pairs — M = max m_p, S = sum s_p exp(m_p - M), fixed order — and writes
// exp(l - M) / S for its slice of the row.  One read of the logits and one write of the heat-map by rows x slices workgroups.
__global__ __launch_bounds__(1024) void softmax_apply_kernel(const float* __restrict__ in, const float* __restrict__ part,
                                                             int P, float* __restrict__ out, int n, int slice_len) {
  __shared__ float sh[16];
  const int row = blockIdx.y;
  const float* pr = part + (size_t)row * P * 2;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < P; i += blockDim.x) m = fmaxf(m, pr[2 * i]);
  m = block_max(m, sh);
  float s = 0.f;
  for (int i = threadIdx.x; i < P; i += blockDim.x) s += pr[2 * i + 1] * expf(pr[2 * i] - m);
  s = block_sum(s, sh);
  const float inv = 1.0f / s;
  const int e0 = blockIdx.x * slice_len, e1 = min(e0 + slice_len, n);
  const float* r = in + (size_t)row * n;
  float* o = out + (size_t)row * n;
  for (int i = e0 + 4 * threadIdx.x; i < e1; i += 4 * blockDim.x) {      // slice_len % 4 == 0 and n % 4 == 0
    const f32x4 v = *reinterpret_cast<const f32x4*>(r + i);
    f32x4 e;
    e[0] = expf(v[0] - m) * inv; e[1] = expf(v[1] - m) * inv;
    e[2] = expf(v[2] - m) * inv; e[3] = expf(v[3] - m) * inv;
    *reinterpret_cast<f32x4*>(o + i) = e;
  }
}

// ---------------------------------------------------------------------------------------------
// Losses (losses.py).  Stage 1: one workgroup per sample writes partial (num, den); stage 2: one
// wave combines them in fixed order.  masked_select is rewritten as a masked sum (graph-capturable).
// ---------------------------------------------------------------------------------------------
// infoNCE, ONE pass over (scores, labels): workgroup (chunk, sample) reduces its NCE_CH elements to
//   z = sum exp(s/T)   a = sum_{label > 1e-2} (s/T) * label   den = sum_{label > 1e-2} label          (losses.py:12-17)
// and the finish kernel merges the chunks of a row in fixed order: num_b = a_b - log(z_b) * den_b (= sum (s/T - log z) * label,
// losses.py:16-17).  rows[b] = (num_b, den_b, z_b, 0) stays around for the backward, which therefore needs no statistics pass
// of its own.  (The first version ran ONE workgroup per sample — 64 workgroups on 256 CUs, each walking 1.3 M scores twice.)
constexpr int NCE_CH = 8192;
__global__ __launch_bounds__(256) void infonce_part_kernel(const float* __restrict__ sc, const float* __restrict__ lab,
                                                           float inv_t, float* __restrict__ part, int n, int chunks) {
  __shared__ float sh[16];
  const size_t row = (size_t)blockIdx.y * n;
  const int i0 = blockIdx.x * NCE_CH, i1 = min(n, i0 + NCE_CH);
  float z = 0.f, a = 0.f, den = 0.f;
  if ((n & 3) == 0) {
    for (int i = i0 + threadIdx.x * 4; i < i1; i += 1024) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(sc + row + i);
      const f32x4 l = *reinterpret_cast<const f32x4*>(lab + row + i);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float t = v[q] * inv_t;
        z += expf(t);
        if (l[q] > 1e-2f) { a = fmaf(t, l[q], a); den += l[q]; }               // losses.py:13
      }
    }
  } else {
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
      const float t = sc[row + i] * inv_t, lv = lab[row + i];
      z += expf(t);
      if (lv > 1e-2f) { a = fmaf(t, lv, a); den += lv; }
    }
  }
  z = block_sum(z, sh);
  a = block_sum(a, sh);
  den = block_sum(den, sh);
  if (threadIdx.x == 0) {
    float* o = part + ((size_t)blockIdx.y * chunks + blockIdx.x) * 4;
    o[0] = z; o[1] = a; o[2] = den;
  }
}

// one workgroup: wave w merges the chunk partials of rows w, w+4, ... (lane-strided, then a butterfly: fixed order), then
// wave 0 adds the rows: loss = -sum(num)/sum(den) (losses.py:18-20), rows[4B] = sum(den) for the backward
__global__ __launch_bounds__(256) void infonce_finish_kernel(const float* __restrict__ part, float* __restrict__ rows,
                                                             float* __restrict__ loss, int B, int chunks) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int b = wave; b < B; b += 4) {
    float z = 0.f, a = 0.f, den = 0.f;
    for (int c = lane; c < chunks; c += 64) {
      const float* q = part + ((size_t)b * chunks + c) * 4;
      z += q[0]; a += q[1]; den += q[2];
    }
    z = wave_sum(z); a = wave_sum(a); den = wave_sum(den);
    if (lane == 0) {
      rows[4 * b] = a - logf(z) * den;
      rows[4 * b + 1] = den;
      rows[4 * b + 2] = z;
      rows[4 * b + 3] = 0.f;
    }
  }
  __threadfence_block();
  __syncthreads();
  if (wave == 0) {
    float num = 0.f, den = 0.f;
    for (int b = lane; b < B; b += 64) { num += rows[4 * b]; den += rows[4 * b + 1]; }
    num = wave_sum(num); den = wave_sum(den);
    if (lane == 0) {
      loss[0] = -num / den;
      rows[4 * B] = den;
    }
  }
}

__global__ __launch_bounds__(1024) void ce_rows_kernel(const float* __restrict__ lg, const float* __restrict__ lab,
                                                       float* __restrict__ part, int n) {
  __shared__ float sh[16];
  const float* r = lg + (size_t)blockIdx.x * n;
  const float* l = lab + (size_t)blockIdx.x * n;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, r[i]);
  m = block_max(m, sh);
  float z = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) z += expf(r[i] - m);
  z = block_sum(z, sh);
  const float logz = m + logf(z);
  float num = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) num = fmaf(l[i], r[i] - logz, num);   // losses.py:24
  num = block_sum(num, sh);
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = num;
    part[2 * blockIdx.x + 1] = 0.f;
  }
}

__global__ __launch_bounds__(1024) void ori_rows_kernel(const float* __restrict__ ori, const float* __restrict__ gto,
                                                        const float* __restrict__ gt, float* __restrict__ part,
                                                        int hw) {
  __shared__ float sh[16];
  const size_t b = blockIdx.x;
  const float* o0 = ori + b * 2 * hw;
  const float* g0 = gto + b * 2 * hw;
  const float* w = gt + b * hw;
  float s = 0.f;
  for (int i = threadIdx.x; i < hw; i += blockDim.x) {
    const float d0 = g0[i] - o0[i], d1 = g0[hw + i] - o0[hw + i];
    s = fmaf(d0 * d0 + d1 * d1, w[i], s);                                         // losses.py:29
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = s;
    part[2 * blockIdx.x + 1] = 0.f;
  }
}

// ---------------------------------------------------------------------------------------------
// Eval post-processing (train_VIGOR.py:294-324): arg-max pixel (first maximum), orientation there,
// acos-based angle.  One workgroup per sample; ties resolve to the smallest index like numpy.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void eval_post_kernel(const float* __restrict__ heat, const float* __restrict__ ori,
                                                         float* __restrict__ out, int h, int w) {
  __shared__ float sv[16];
  __shared__ int si[16];
  const int n = h * w;
  const float* r = heat + (size_t)blockIdx.x * n;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float v = r[i];
    if (v > best) { best = v; bi = i; }          // strided scan keeps the smallest index per thread
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < (int)(blockDim.x >> 6); ++k)
      if (sv[k] > best || (sv[k] == best && si[k] < bi)) { best = sv[k]; bi = si[k]; }
    if (bi == 0x7fffffff) bi = 0;
    const float c = ori[((size_t)blockIdx.x * 2 + 0) * n + bi];
    const float s = ori[((size_t)blockIdx.x * 2 + 1) * n + bi];
    float ang = NAN;
    if (fabsf(c) <= 1.0f && fabsf(s) <= 1.0f) {
      const float a = acosf(c) * 57.29577951308232f;
      ang = (s < 0.0f) ? fmodf(360.0f - a, 360.0f) : a;   // degrees(-a) % 360 in Python
    }
    float* o = out + (size_t)blockIdx.x * 6;
    o[0] = (float)(bi / w); o[1] = (float)(bi % w); o[2] = c; o[3] = s; o[4] = ang; o[5] = best;
  }
}

// mode 0: -sum(num)/sum(den) ; mode 1: -sum(num)/B ; mode 2: +sum(num)/B
// bf16 -> fp32 widening of an activation tensor (the bf16 storage path hands its last decoder levels to the fp32
// kernels so that the heat-map arg-max keeps fp32 resolution: SURVEY.md section 7, "keep the last two levels in fp32")
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const cc_bf16* __restrict__ src, float* __restrict__ dst, long n8) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const cc_bf16x8 v = reinterpret_cast<const cc_bf16x8*>(src)[i];
  cc_f32x4 a, b;
#pragma unroll
  for (int j = 0; j < 4; ++j) { a[j] = (float)v[j]; b[j] = (float)v[j + 4]; }
  reinterpret_cast<cc_f32x4*>(dst)[2 * i] = a;
  reinterpret_cast<cc_f32x4*>(dst)[2 * i + 1] = b;
}

__global__ void loss_finish_kernel(const float* __restrict__ part, float* __restrict__ loss, int B, int mode) {
  float num = 0.f, den = 0.f;
  for (int i = threadIdx.x; i < B; i += 64) {
    num += part[2 * i];
    den += part[2 * i + 1];
  }
  num = wave_sum(num);
  den = wave_sum(den);
  if (threadIdx.x == 0) loss[0] = mode == 0 ? -num / den : (mode == 1 ? -num / (float)B : num / (float)B);
}

}  // namespace ccvpe

using namespace ccvpe;

thread_local char ccvpe::g_err[512] = "";

extern "C" const char* ccvpe_last_error(void) { return g_err; }
extern "C" int ccvpe_abi_version(void) { return 7; }

extern "C" int ccvpe_ground_descriptor_f32(const float* y1, int ld, const float* wh, const float* bh, const int* cd,
                                           float* out, int B, int h, int w, void* stream) {
  if (!cd || B <= 0 || h <= 0 || w <= 0) return fail(CCVPE_EINVAL, "ground_descriptor: bad args");
  GdescCfg cfg;
  int off = 0;
  for (int l = 0; l < 6; ++l) {
    if (cd[l] <= 0) return fail(CCVPE_EINVAL, "ground_descriptor: cd[%d] <= 0", l);
    cfg.cd[l] = cd[l];
    cfg.off[l] = off;
    cfg.obase[l] = w * off;
    off += cd[l];
  }
  if (off > ld) return fail(CCVPE_EINVAL, "ground_descriptor: sum(cd) > ld");
  dim3 grid((w * off + 255) / 256, B);
  hipLaunchKernelGGL(gdesc_kernel, grid, dim3(256), 0, (hipStream_t)stream, y1, ld, wh, bh, cfg, out, h, w, off);
  return check_launch("gdesc_kernel");
}

template <typename TX>
static int head_any(const TX* x, const float* w, const float* bias, float* out, int B, int H, int W, int cout,
                    int normalize, void* stream) {
  if (!aligned16(x)) return fail(CCVPE_EINVAL, "head_conv: x must be 16-byte aligned");
  dim3 grid(((W + HC_TW - 1) / HC_TW) * ((H + HC_TH - 1) / HC_TH) * B);
  if (cout == 1)
    hipLaunchKernelGGL((head_conv_kernel<TX, 1>), grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, out, H, W, 0);
  else if (cout == 2)
    hipLaunchKernelGGL((head_conv_kernel<TX, 2>), grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, out, H, W, normalize);
  else
    return fail(CCVPE_EINVAL, "head_conv: cout must be 1 or 2");
  return check_launch("head_conv_kernel");
}

extern "C" int ccvpe_head_conv3x3_f32(const float* x, const float* w, const float* bias, float* out, int B, int H,
                                      int W, int cout, int normalize, void* stream) {
  return head_any<float>(x, w, bias, out, B, H, W, cout, normalize, stream);
}
extern "C" int ccvpe_head_conv3x3_bf16(const void* x, const float* w, const float* bias, float* out, int B, int H,
                                       int W, int cout, int normalize, void* stream) {
  return head_any<cc_bf16>(reinterpret_cast<const cc_bf16*>(x), w, bias, out, B, H, W, cout, normalize, stream);
}

extern "C" int ccvpe_softmax_rows_f32(const float* in, float* out, int rows, int n, void* stream) {
  if (rows <= 0 || n <= 0) return fail(CCVPE_EINVAL, "softmax: bad shape");
  if (!aligned16(in) || !aligned16(out) || (n % 4)) return fail(CCVPE_EINVAL, "softmax: 16-byte aligned rows required");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(1024), 0, (hipStream_t)stream, in, out, n);
  return check_launch("softmax_rows_kernel");
}

extern "C" int ccvpe_softmax_apply_f32(const float* logits, const float* partials, int n_partials, float* out, int rows, int n,
                                       void* stream) {
  if (rows <= 0 || n <= 0 || n_partials <= 0 || rows > 65535) return fail(CCVPE_EINVAL, "softmax_apply: bad shape");
  if (!aligned16(logits) || !aligned16(out) || (n % 4)) return fail(CCVPE_EINVAL, "softmax_apply: 16-byte aligned rows required");
  int slices = (1024 + rows - 1) / rows;               // ~1024 workgroups per launch
  slices = slices < 1 ? 1 : (slices > 64 ? 64 : slices);
  const int slice_len = (((n + slices - 1) / slices) + 3) & ~3;
  hipLaunchKernelGGL(softmax_apply_kernel, dim3((n + slice_len - 1) / slice_len, rows), dim3(1024), 0, (hipStream_t)stream, logits,
                     partials, n_partials, out, n, slice_len);
  return check_launch("softmax_apply_kernel");
}

extern "C" int ccvpe_eval_postprocess_f32(const float* heatmap, const float* ori, float* out, int B, int h, int w,
                                          void* stream) {
  if (B <= 0 || h <= 0 || w <= 0) return fail(CCVPE_EINVAL, "eval_postprocess: bad shape");
  hipLaunchKernelGGL(eval_post_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, heatmap, ori, out, h, w);
  return check_launch("eval_post_kernel");
}

extern "C" int ccvpe_infonce_scratch_floats(int B, int n) {
  if (B <= 0 || n <= 0) return CCVPE_EINVAL;
  const long fl = (long)B * ((n + NCE_CH - 1) / NCE_CH) * 4;
  return fl > 0x7fffffffL ? CCVPE_EINVAL : (int)fl;
}

extern "C" int ccvpe_infonce_loss_f32(const float* scores, const float* labels, float temperature, float* loss,
                                      float* rows, float* scratch, int B, int n, void* stream) {
  if (B <= 0 || n <= 0 || temperature <= 0.f) return fail(CCVPE_EINVAL, "infonce: bad args");
  if (B > 65535) return fail(CCVPE_EINVAL, "infonce: batch > 65535");
  if ((n & 3) == 0 && (!aligned16(scores) || !aligned16(labels))) return fail(CCVPE_EINVAL, "infonce: 16-byte alignment required");
  hipStream_t st = (hipStream_t)stream;
  const int chunks = (n + NCE_CH - 1) / NCE_CH;
  hipLaunchKernelGGL(infonce_part_kernel, dim3(chunks, B), dim3(256), 0, st, scores, labels, 1.0f / temperature, scratch, n, chunks);
  hipLaunchKernelGGL(infonce_finish_kernel, dim3(1), dim3(256), 0, st, scratch, rows, loss, B, chunks);
  return check_launch("infonce");
}

extern "C" int ccvpe_cross_entropy_loss_f32(const float* logits, const float* labels, float* loss, float* scratch,
                                            int B, int n, void* stream) {
  if (B <= 0 || n <= 0) return fail(CCVPE_EINVAL, "cross_entropy: bad args");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_rows_kernel, dim3(B), dim3(1024), 0, st, logits, labels, scratch, n);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, st, scratch, loss, B, 1);
  return check_launch("cross_entropy");
}

extern "C" int ccvpe_orientation_loss_f32(const float* ori, const float* gt_ori, const float* gt, float* loss,
                                          float* scratch, int B, int hw, void* stream) {
  if (B <= 0 || hw <= 0) return fail(CCVPE_EINVAL, "orientation_loss: bad args");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ori_rows_kernel, dim3(B), dim3(1024), 0, st, ori, gt_ori, gt, scratch, hw);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, st, scratch, loss, B, 2);
  return check_launch("orientation_loss");
}

extern "C" int ccvpe_cast_bf16_f32(const void* src, float* dst, long n_elems, void* stream) {
  if (n_elems <= 0 || n_elems % 8) return fail(CCVPE_EINVAL, "cast_bf16_f32: n %% 8");
  if (!aligned16(src) || !aligned16(dst)) return fail(CCVPE_EINVAL, "cast_bf16_f32: 16-byte alignment required");
  const long n8 = n_elems / 8;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const cc_bf16*>(src), dst, n8);
  return check_launch("cast_bf16_f32_kernel");
}
