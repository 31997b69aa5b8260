// Internal interface of csrc/mbconv_plane.hip (the late-block MBConv front: planes of <= 1024 pixels), used by the
// ccvpe_mbconv_front_* and ccvpe_dwconv_* entry points (mbconv_front.hip, effnet_ops.hip).
#pragma once

namespace ccvpe {

// SE partial rows per sample (= row bands per plane) of the plane kernel for this shape, 0 if it does not take it.
// cin == 0: the depthwise-only form (x is the expanded tensor).  Must not depend on the storage type or the batch.
int mbplane_nblk(int H, int W, int cin, int mid, int k, int stride);

// TE = float / cc_bf16 selected by is_bf16.  expand != 0: x [B,H,W,cin] + w_exp [mid^16][kpad]; else x [B,H,W,mid].
int mbplane_launch(int is_bf16, int expand, const void* x, const void* w_exp, int kpad, const float* s0, const float* b0,
                   const float* w_dw, const float* s1, const float* b1, void* y, float* se_partial, int B, int H, int W,
                   int cin, int mid, int k, int stride, int circular, void* stream);

// does the band-owner kernel (bf16, fused form) take this shape?  (the same test mbplane_launch applies)
bool mbband_takes(int H, int W, int cin, int mid, int k, int stride, int B);

// A/B switch (ccvpe_set_mbconv_plane_kernels): 0 = the round-5 chain (pointwise GEMM + dwconv_plane_kernel)
extern int g_mbplane_mode;   // bit 0: fused expand + depthwise, bit 1: depthwise-only form, bit 2: band-owner kernel (bf16, fused)

}  // namespace ccvpe
