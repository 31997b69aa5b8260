// Training-step glue on the device: the ground-truth tensors the training scripts build on the
// host and copy over (24 MB per sample) are generated here from 3 scalars per sample, and Adam runs as ONE launch
// over all parameter tensors.
//
// Targets (datasets.py:145-166 VIGOR, :470-501 KITTI; train_VIGOR.py:120-128):
//   x_j = -W/2 + cx + j*W/(W-1),  y_i = -H/2 + cy + i*H/(H-1)         (np.linspace, both ends included)
//   gt[b,0,i,j]        = exp(-(x_j^2 + y_i^2) / (2 sigma^2))            (VIGOR: cx = col_offset, cy = -row_offset;
//                                                                        KITTI: cx = x_offset,  cy = y_offset)
//   gt_norm[b, i*W+j]  = gt / sum(gt)                                   (train_VIGOR.py:120-121)
//   gt_ori[b,{0,1},:,:] = cos / sin (angle)                             (datasets.py:163-164)
//   label_l[b,bin,Y,X] = w_bin * max_{k x k cell} gt,  k = 64 >> l      (MaxPool2d(k,k) of gt_with_ori, never materialised)
//     bins: index = floor(angle / width), ratio = (angle mod width)/width;
//           index == 0: w[0] = 1-ratio, w[n-1] = ratio;  else w[n-index] = 1-ratio, w[n-index-1] = ratio
//   The cell maximum is separable (exp and fl(x^2+y^2) are monotone): exp(-(min_j x_j^2 + min_i y_i^2)/(2 sigma^2)),
//   which equals the max-pool of the fp32 gt values exactly.
#include "common.h"

namespace ccvpe {

__device__ __forceinline__ float coord(int j, int n, float c) { return -0.5f * n + c + j * ((float)n / (float)(n - 1)); }

constexpr int TG_PIX = 1024;   // pixels per workgroup of the gt kernel

__global__ __launch_bounds__(256) void targets_gt_kernel(const float* __restrict__ center, const float* __restrict__ angle,
                                                         float inv2s2, float* __restrict__ gt, float* __restrict__ gt_ori,
                                                         float* __restrict__ part, int H, int W, int nblk) {
  __shared__ float sh[4];
  const int b = blockIdx.y;
  const float cx = center[2 * b], cy = center[2 * b + 1];
  const float a = angle[b] * 0.017453292519943295f;
  const float ca = cosf(a), sa = sinf(a);
  const int hw = H * W;
  float s = 0.f;
  for (int p = blockIdx.x * TG_PIX + threadIdx.x; p < min((blockIdx.x + 1) * TG_PIX, hw); p += 256) {
    const int i = p / W, j = p - i * W;
    const float x = coord(j, W, cx), y = coord(i, H, cy);
    const float v = expf(-(x * x + y * y) * inv2s2);
    gt[(size_t)b * hw + p] = v;
    gt_ori[(size_t)b * 2 * hw + p] = ca;
    gt_ori[(size_t)b * 2 * hw + hw + p] = sa;
    s += v;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[(size_t)b * nblk + blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void targets_norm_kernel(const float* __restrict__ gt, const float* __restrict__ part,
                                                           float* __restrict__ gt_norm, int hw, int nblk) {
  const int b = blockIdx.y;
  float tot = 0.f;
  for (int k = 0; k < nblk; ++k) tot += part[(size_t)b * nblk + k];
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p < hw) gt_norm[(size_t)b * hw + p] = gt[(size_t)b * hw + p] / tot;
}

struct PyramidPtrs {
  float* lab[6];
};

__global__ __launch_bounds__(256) void targets_pyramid_kernel(const float* __restrict__ center, const float* __restrict__ angle,
                                                              float inv2s2, int n_bins, float bin_width, int ascending,
                                                              const PyramidPtrs out, int H, int W) {
  const int b = blockIdx.y, l = blockIdx.z;
  const int k = 64 >> l;
  const int hl = H / k, wl = W / k;
  const int cell = blockIdx.x * 256 + threadIdx.x;
  if (cell >= hl * wl) return;
  const int Y = cell / wl, X = cell - Y * wl;
  const float cx = center[2 * b], cy = center[2 * b + 1];
  float mx = 3.0e38f, my = 3.0e38f;
  for (int t = 0; t < k; ++t) {
    const float x = coord(X * k + t, W, cx), y = coord(Y * k + t, H, cy);
    mx = fminf(mx, x * x);
    my = fminf(my, y * y);
  }
  const float v = expf(-(mx + my) * inv2s2);
  // datasets.py:154-155: index = int(angle // width), ratio = (angle % width) / width in float64.  The angle is wrapped
  // into [0, 360) first (the reference's callers do: datasets.py:483-487) so that a caller passing e.g. 90 - random_ori
  // directly cannot index outside the bins, and index is clamped against a wrap that rounds up to 360.
  double ang = (double)angle[b];
  ang -= 360.0 * floor(ang / 360.0);
  int index = (int)floor(ang / (double)bin_width);
  index = min(max(index, 0), n_bins - 1);
  const float ratio = (float)((ang - (double)index * (double)bin_width) / (double)bin_width);
  // VIGOR / KITTI count the bins DOWN from the angle (datasets.py:156-161, :489-494); the Oxford RobotCar loader counts them UP
  // (datasets.py:340-347: bins index and index + 1, wrapping n-1 -> 0)
  const int b0 = ascending ? index : (index == 0 ? 0 : n_bins - index);
  const int b1 = ascending ? (index == n_bins - 1 ? 0 : index + 1) : (index == 0 ? n_bins - 1 : n_bins - index - 1);
  float* o = out.lab[l] + (size_t)b * n_bins * hl * wl + cell;
  for (int q = 0; q < n_bins; ++q) {
    float w = 0.f;
    if (q == b0) w = 1.0f - ratio;
    if (q == b1) w = (b1 == b0) ? w : ratio;
    o[(size_t)q * hl * wl] = w * v;
  }
}

// ---------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam semantics, train_VIGOR.py:104: lr, betas=(0.9, 0.999), eps 1e-8, no weight decay,
// no amsgrad) over a table of tensors: one launch updates every parameter.
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// Every tensor carries its OWN hyper-parameter row (step size with its own bias correction, betas, eps), so tensors
// that started receiving gradients at different steps, or sit in different param groups, update in the same launch.
// grad_scale multiplies the gradient on load (data-parallel SUM all-reduce -> mean without a separate pass).
// ---------------------------------------------------------------------------------------------
constexpr int ADAM_CHUNK = 4096;   // elements per workgroup
constexpr int ADAM_HYPER = 8;      // floats per tensor: step_size, b1, b2, 1-b1, 1-b2, eps, sqrt(1-b2^t), reserved

__global__ __launch_bounds__(256) void adam_kernel(const long long* __restrict__ table /*[n][5]: p, g, m, v, numel*/,
                                                   const float* __restrict__ hyper /*[n][ADAM_HYPER]*/,
                                                   const int* __restrict__ chunk_tensor, const int* __restrict__ chunk_off,
                                                   float grad_scale) {
  const int t = chunk_tensor[blockIdx.x];
  const long long* row = table + (size_t)t * 5;
  float* p = reinterpret_cast<float*>(row[0]);
  const float* g = reinterpret_cast<const float*>(row[1]);
  float* m = reinterpret_cast<float*>(row[2]);
  float* v = reinterpret_cast<float*>(row[3]);
  const long long n = row[4];
  if (g == nullptr) return;
  const float* hy = hyper + (size_t)t * ADAM_HYPER;
  const float step = hy[0], b1 = hy[1], b2 = hy[2], omb1 = hy[3], omb2 = hy[4], eps = hy[5], bc2_sqrt = hy[6];
  const long long base = (long long)chunk_off[blockIdx.x] * ADAM_CHUNK;
  const long long end = min(base + ADAM_CHUNK, n);
  if (((n | base) & 3) == 0 && ((((size_t)p | (size_t)g | (size_t)m | (size_t)v) & 15) == 0)) {     // 16-byte path
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (long long i = base + 4 * threadIdx.x; i < end; i += 1024) {
      f4 gi = *reinterpret_cast<const f4*>(g + i) * grad_scale;
      f4 mi = *reinterpret_cast<const f4*>(m + i), vi = *reinterpret_cast<const f4*>(v + i), pi = *reinterpret_cast<const f4*>(p + i);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        mi[q] = b1 * mi[q] + omb1 * gi[q];
        vi[q] = b2 * vi[q] + omb2 * gi[q] * gi[q];
        pi[q] -= step * mi[q] / (sqrtf(vi[q]) / bc2_sqrt + eps);
      }
      *reinterpret_cast<f4*>(m + i) = mi;
      *reinterpret_cast<f4*>(v + i) = vi;
      *reinterpret_cast<f4*>(p + i) = pi;
    }
    return;
  }
  for (long long i = base + threadIdx.x; i < end; i += 256) {
    const float gi = g[i] * grad_scale;
    const float mi = b1 * m[i] + omb1 * gi;          // 1 - beta computed in double on the host, as torch does
    const float vi = b2 * v[i] + omb2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= step * mi / (sqrtf(vi) / bc2_sqrt + eps);
  }
}

}  // namespace ccvpe

using namespace ccvpe;

extern "C" int ccvpe_train_targets_nblk(int h, int w) { return (h * w + TG_PIX - 1) / TG_PIX; }

extern "C" int ccvpe_train_targets_ordered_f32(const float* center_xy, const float* angle_deg, int n_bins, int ascending, float sigma,
                                               float* gt, float* gt_norm, float* gt_ori, float* lab1, float* lab2, float* lab3,
                                               float* lab4, float* lab5, float* lab6, float* scratch, int batch, int h, int w,
                                               void* stream) {
  if (batch <= 0 || h < 64 || w < 64 || h % 64 || w % 64) return fail(CCVPE_EINVAL, "train_targets: h, w must be multiples of 64");
  if (n_bins < 2 || sigma <= 0.f) return fail(CCVPE_EINVAL, "train_targets: bad n_bins / sigma");
  hipStream_t st = (hipStream_t)stream;
  const float inv2s2 = 1.0f / (2.0f * sigma * sigma);
  const int nblk = ccvpe_train_targets_nblk(h, w);
  hipLaunchKernelGGL(targets_gt_kernel, dim3(nblk, batch), dim3(256), 0, st, center_xy, angle_deg, inv2s2, gt, gt_ori, scratch, h,
                     w, nblk);
  hipLaunchKernelGGL(targets_norm_kernel, dim3((h * w + 255) / 256, batch), dim3(256), 0, st, gt, scratch, gt_norm, h * w, nblk);
  PyramidPtrs pp;
  pp.lab[0] = lab1; pp.lab[1] = lab2; pp.lab[2] = lab3; pp.lab[3] = lab4; pp.lab[4] = lab5; pp.lab[5] = lab6;
  const int cells = (h / 2) * (w / 2);
  hipLaunchKernelGGL(targets_pyramid_kernel, dim3((cells + 255) / 256, batch, 6), dim3(256), 0, st, center_xy, angle_deg, inv2s2,
                     n_bins, 360.0f / n_bins, ascending ? 1 : 0, pp, h, w);
  return check_launch("train_targets");
}

extern "C" int ccvpe_train_targets_f32(const float* center_xy, const float* angle_deg, int n_bins, float sigma, float* gt,
                                       float* gt_norm, float* gt_ori, float* lab1, float* lab2, float* lab3, float* lab4,
                                       float* lab5, float* lab6, float* scratch, int batch, int h, int w, void* stream) {
  return ccvpe_train_targets_ordered_f32(center_xy, angle_deg, n_bins, 0, sigma, gt, gt_norm, gt_ori, lab1, lab2, lab3, lab4, lab5,
                                         lab6, scratch, batch, h, w, stream);
}

extern "C" int ccvpe_adam_chunk_elems(void) { return ADAM_CHUNK; }

extern "C" int ccvpe_adam_hyper_floats(void) { return ADAM_HYPER; }

extern "C" int ccvpe_adam_step_f32(const void* table, const float* hyper, const int* chunk_tensor, const int* chunk_off,
                                   int n_chunks, float grad_scale, void* stream) {
  if (n_chunks <= 0 || !table || !hyper || !chunk_tensor || !chunk_off) return fail(CCVPE_EINVAL, "adam_step: bad args");
  hipLaunchKernelGGL(adam_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const long long*>(table),
                     hyper, chunk_tensor, chunk_off, grad_scale);
  return check_launch("adam_kernel");
}

// ---------------------------------------------------------------------------------------------
// Many small device-to-device copies in one launch (gradients -> their slots of the flat all-reduce arena,
// ccvpe_amd/harness.py): ~420 tensors per training step, most of them BatchNorm / bias / SE vectors of 16-1280 floats; as
// individual copies they were 420 launches and 1.9 ms of kernel time per step.  Pointers travel BY VALUE in the kernel
// arguments (no table upload): up to CCVPE_MULTI_COPY_MAX tensors per launch, the host entry point loops.
// ---------------------------------------------------------------------------------------------
namespace ccvpe {
constexpr int MC_MAX = 96, MC_CHUNKS = 16;
struct MultiCopyArgs {
  const float* src[MC_MAX];
  float* dst[MC_MAX];
  int n[MC_MAX];
};
__global__ __launch_bounds__(256) void multi_copy_kernel(const MultiCopyArgs a) {
  const int t = blockIdx.y;
  const float* __restrict__ s = a.src[t];
  float* __restrict__ d = a.dst[t];
  const int n = a.n[t];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += MC_CHUNKS * 256) d[i] = s[i];
}

// One-launch weight re-pack (ccvpe_amd/repack.py).  Every packed weight of the training step is a pure re-layout of live
// parameters (permutes, flips, channel re-orders, zero padding, concatenations): element j of a packed tensor is either 0 or
// element idx[j] - 1 of ONE source tensor.  A chunk is <= RP_CHUNK consecutive packed elements that read the same source.
constexpr int RP_CHUNK = 4096;
__global__ __launch_bounds__(256) void gather_repack_kernel(float* const* __restrict__ dsts, const float* const* __restrict__ srcs,
                                                            const long long* __restrict__ idx_off, const int* __restrict__ counts,
                                                            const int* __restrict__ idx) {
  const int c = blockIdx.x;
  float* __restrict__ d = dsts[c];
  const float* __restrict__ s = srcs[c];
  const int* __restrict__ ix = idx + idx_off[c];
  const int n = counts[c];
  for (int i = threadIdx.x; i < n; i += 256) {
    const int k = ix[i];
    d[i] = k > 0 ? s[k - 1] : 0.f;
  }
}
}  // namespace ccvpe

extern "C" int ccvpe_gather_repack_f32(void* const* dsts, const void* const* srcs, const long long* idx_off, const int* counts,
                                       const int* idx, int n_chunks, void* stream) {
  if (n_chunks < 0 || (n_chunks > 0 && (!dsts || !srcs || !idx_off || !counts || !idx)))
    return fail(CCVPE_EINVAL, "gather_repack: bad arguments");
  if (n_chunks == 0) return CCVPE_OK;
  hipLaunchKernelGGL(ccvpe::gather_repack_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<float* const*>(dsts), reinterpret_cast<const float* const*>(srcs), idx_off, counts, idx);
  return check_launch("gather_repack_kernel");
}

extern "C" int ccvpe_gather_repack_chunk(void) { return ccvpe::RP_CHUNK; }

extern "C" int ccvpe_multi_copy_f32(const void* const* srcs, void* const* dsts, const int* counts, int n, void* stream) {
  if (n < 0 || (n > 0 && (!srcs || !dsts || !counts))) return fail(CCVPE_EINVAL, "multi_copy: bad arguments");
  for (int i0 = 0; i0 < n; i0 += ccvpe::MC_MAX) {
    ccvpe::MultiCopyArgs a;
    const int m = n - i0 < ccvpe::MC_MAX ? n - i0 : ccvpe::MC_MAX;
    for (int i = 0; i < m; ++i) {
      if (counts[i0 + i] < 0 || !srcs[i0 + i] || !dsts[i0 + i]) return fail(CCVPE_EINVAL, "multi_copy: bad entry %d", i0 + i);
      a.src[i] = static_cast<const float*>(srcs[i0 + i]);
      a.dst[i] = static_cast<float*>(dsts[i0 + i]);
      a.n[i] = counts[i0 + i];
    }
    hipLaunchKernelGGL(ccvpe::multi_copy_kernel, dim3(ccvpe::MC_CHUNKS, m), dim3(256), 0, (hipStream_t)stream, a);
  }
  return check_launch("multi_copy_kernel");
}
