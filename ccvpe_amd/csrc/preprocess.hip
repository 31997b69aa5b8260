// Input side of the path: what the reference does per image on the host after JPEG decoding —
//   transforms.Resize([h, w]) on a PIL image (= PIL.Image.resize(..., BILINEAR): antialiased, 8-bit fixed point),
//   ToTensor (uint8 HWC -> float CHW / 255), Normalize(mean, std)             train_VIGOR.py:57-70, train_KITTI.py:93-100
//   torch.roll along W by round(rotation * W)                                  datasets.py:112-121
//   FoV crop grd[:, :, :, :int(W * FoV / 360)]                                 train_VIGOR.py:177-178,272-273
// — as two kernels on the decoded uint8 image resident in HBM.
//
// Pillow's resampler (src/libImaging/Resample.c, the algorithm torchvision's Resize delegates to for PIL inputs) is
// reproduced bit for bit: horizontal pass then vertical pass, each output = clip8((2^21 + sum_k px_k * kk_k) >> 22)
// with the per-output windows / integer coefficients precomputed on the host in double precision exactly as
// precompute_coeffs() + normalize_coeffs_8bpc() do (ccvpe_amd/preprocess.py), the intermediate image in uint8.
// HBM-bound byte work: lanes along output columns (coalesced), coefficient rows are wave-uniform or per-lane L1 hits.
#include "common.h"

namespace ccvpe {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ unsigned char clip8(int v) {
  v >>= PRECISION_BITS;                       // arithmetic shift, as the C code's lookup index
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: src [H][W][3] u8 -> tmp [H][wo][3] u8
__global__ __launch_bounds__(256) void resample_h_kernel(const unsigned char* __restrict__ src, int H, int W,
                                                         const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                         unsigned char* __restrict__ tmp, int wo) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)H * wo) return;
  const int y = (int)(idx / wo), xx = (int)(idx - (long)y * wo);
  const int xmin = bounds[2 * xx], xn = bounds[2 * xx + 1];
  const int* k = kk + (size_t)xx * ksize;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  const unsigned char* row = src + ((size_t)y * W + xmin) * 3;
  for (int x = 0; x < xn; ++x) {
    const int c = k[x];
    s0 += row[3 * x] * c;
    s1 += row[3 * x + 1] * c;
    s2 += row[3 * x + 2] * c;
  }
  unsigned char* o = tmp + (size_t)idx * 3;
  o[0] = clip8(s0);
  o[1] = clip8(s1);
  o[2] = clip8(s2);
}

// vertical pass + ToTensor + Normalize + roll + crop: tmp [H][wo][3] u8 -> dst [3][ho][keep] f32
__global__ __launch_bounds__(256) void resample_v_norm_kernel(const unsigned char* __restrict__ tmp, int wo,
                                                              const int* __restrict__ bounds, const int* __restrict__ kk,
                                                              int ksize, float* __restrict__ dst, int ho, int keep, int roll,
                                                              float m0, float m1, float m2, float d0, float d1, float d2) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)ho * keep) return;
  const int yy = (int)(idx / keep), xo = (int)(idx - (long)yy * keep);
  int xs = xo - roll;                          // torch.roll(x, r, W): out[..., j] = in[..., (j - r) mod W]
  xs %= wo;
  if (xs < 0) xs += wo;
  const int ymin = bounds[2 * yy], yn = bounds[2 * yy + 1];
  const int* k = kk + (size_t)yy * ksize;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  const unsigned char* col = tmp + ((size_t)ymin * wo + xs) * 3;
  for (int y = 0; y < yn; ++y) {
    const int c = k[y];
    const unsigned char* p = col + (size_t)y * wo * 3;
    s0 += p[0] * c;
    s1 += p[1] * c;
    s2 += p[2] * c;
  }
  const size_t plane = (size_t)ho * keep;
  // ToTensor: float(v) / 255 ; Normalize: (t - mean) / std   (IEEE divisions, fp32, same order as torchvision)
  dst[idx] = ((float)clip8(s0) / 255.0f - m0) / d0;
  dst[plane + idx] = ((float)clip8(s1) / 255.0f - m1) / d1;
  dst[2 * plane + idx] = ((float)clip8(s2) / 255.0f - m2) / d2;
}

}  // namespace ccvpe

using namespace ccvpe;

extern "C" int ccvpe_preprocess_u8_f32(const unsigned char* src, int in_h, int in_w, const int* xbounds, const int* xcoef,
                                       int xksize, const int* ybounds, const int* ycoef, int yksize, unsigned char* tmp,
                                       float* dst, int out_h, int out_w, int keep_w, int roll, const float* mean,
                                       const float* stdv, void* stream) {
  if (in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0 || keep_w <= 0 || keep_w > out_w || xksize <= 0 || yksize <= 0)
    return fail(CCVPE_EINVAL, "preprocess: bad shape");
  if (!src || !tmp || !dst || !xbounds || !xcoef || !ybounds || !ycoef || !mean || !stdv) return fail(CCVPE_EINVAL, "preprocess: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const long n1 = (long)in_h * out_w;
  hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, st, src, in_h, in_w, xbounds, xcoef, xksize,
                     tmp, out_w);
  const long n2 = (long)out_h * keep_w;
  hipLaunchKernelGGL(resample_v_norm_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, tmp, out_w, ybounds, ycoef, yksize,
                     dst, out_h, keep_w, roll, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
  return check_launch("preprocess");
}
