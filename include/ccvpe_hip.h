/*
 * ccvpe_hip.h — C ABI of libccvpe_hip.so: the MI355X (gfx950) implementation of CCVPE's dense
 * cross-view matching forward path.
 *
 * The reference (tudelft-iv/CCVPE) has no FFI or plugin registry: the path sits behind a Python
 * nn.Module (`CVM_VIGOR.forward(grd, sat)`, /root/reference/models.py:150) that hands every
 * arithmetic step to a stock torch op.  Each entry point below therefore replaces one torch-op
 * group of that forward; the comment on each cites the reference lines it stands in for.  The
 * Python host (ccvpe_amd/models.py) binds them with ctypes — see INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller.
 *   - activations are fp32, NHWC ("pixel-major"): element (b,y,x,c) at ((b*H+y)*W+x)*ld + c with
 *     ld >= C the pixel stride in floats (lets a kernel write into a wider concat buffer).
 *   - the reference's NCHW fp32 tensors cross the boundary only at the two ends: the input images
 *     (stem kernel reads NCHW) and the returned heat-map / orientation / score tensors (written
 *     NCHW by the kernels that produce them).
 *   - `stream` is a hipStream_t passed as void*; every call only ENQUEUES work on it (graph-capture
 *     safe: no allocation, no synchronisation, no host-side state).
 *   - return 0 on success, a negative CCVPE_E* code otherwise; ccvpe_last_error() gives the text.
 *     No C++ exception crosses the ABI.
 */
#ifndef CCVPE_HIP_H
#define CCVPE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define CCVPE_OK 0
#define CCVPE_EINVAL (-1)   /* bad argument (shape/alignment not supported) */
#define CCVPE_ELAUNCH (-2)  /* hipLaunchKernel / hipGetLastError failed     */

#define CCVPE_ACT_NONE 0
#define CCVPE_ACT_RELU 1
#define CCVPE_ACT_SWISH 2
#define CCVPE_ACT_RELU_MASK 3  /* out = residual > 0 ? value : 0 — the backward of a ReLU fused into the conv that produces its
                                 output gradient (residual = the ReLU's forward output; it is a mask here, not added) */

#define CCVPE_OUT_NHWC 0      /* dst[(pixel)*ldd + n]                                            */
#define CCVPE_OUT_DECONV2X 1  /* n = (dy*2+dx)*cout + co -> dst[(b,2y+dy,2x+dx)*ldd + co]        */

#define CCVPE_MAX_SHIFTS 48

const char* ccvpe_last_error(void);
int ccvpe_abi_version(void);

/* -------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_16x16x4_f32).
 *
 *   out[m, n] = act( (sum_{tap, c} in[pixel(m) + tap, c] * w[n, tap, c]) * scale[n] + shift[n] )
 *               (+ residual[m, n])
 *
 * One kernel covers every dense contraction of the path:
 *   1x1  : MBConv expand / project (+SE gate on the input, +skip) and the 320->1280 head
 *          (efficientnet_pytorch/model.py:62,86,104-106,121-130,209,299), the fused ground
 *          descriptor 1x1 convs (models.py:57-97)
 *   2x2/2: the aerial descriptor Linear over 2x2 patches (models.py:102-104,173-184)
 *   3x3  : double_conv, reading cat[deconv_out, skip] as two sources without materialising the
 *          concat (models.py:42-47,208-209,...)
 *   deconv: ConvTranspose2d(k=2,s=2) as one GEMM with N = 4*cout and a pixel-shuffle store
 *          (models.py:109-145)
 *
 * Weights are pre-packed by the host: w[n][k] row-major, k = tap*(c0+c1) + concat-channel,
 * K zero-padded to `kpad` (multiple of 16), N zero-padded to a multiple of 16 rows.
 * c0, c1 must be multiples of 8; ld0/ld1/ldd/ldres multiples of 4; pointers 16-byte aligned.
 * ----------------------------------------------------------------------------------------- */
typedef struct ccvpe_conv_desc {
  const float* src0; /* [B,H,W,ld0] first source                                   */
  const float* src1; /* [B,H,W,ld1] second (concatenated) source, or NULL           */
  const float* gate; /* [B,c0] per-sample per-channel multiplier on src0 (SE), NULL  */
  const float* w;    /* packed weights [npad16][kpad]                                */
  const float* scale;    /* [n] or NULL (=1)   folded BN scale                       */
  const float* shift;    /* [n] or NULL (=0)   folded BN shift / conv bias           */
  const float* residual; /* [M,ldres] or NULL  MBConv identity skip                  */
  float* dst;
  int c0, ld0, c1, ld1;
  int batch, in_h, in_w;
  int kh, kw, stride, pad; /* (1,1,1,0) (3,3,1,1) (2,2,2,0)                          */
  int n;                   /* real GEMM N (for deconv: 4*cout)                        */
  int kpad;
  int ldd, ldres;
  int act;      /* CCVPE_ACT_*  */
  int out_mode; /* CCVPE_OUT_*  */
} ccvpe_conv_desc;

int ccvpe_conv_igemm_f32(const ccvpe_conv_desc* desc, void* stream);

/* Split-K form for small-batch GEMMs (a handful of output tiles, thousands of K stages: decoder levels 6-5 and the late
 * encoder 1x1s at batch <= 8 leave most of the 256 CUs idle and walk K serially).  K is cut into slices, each slice
 * is its own workgroup row writing raw fp32 partial sums to `scratch`, and a second kernel adds the slices in index
 * order (deterministic) and applies the epilogue.  3x3 convs take the generic gather kernel in this mode.
 *   ccvpe_conv_igemm_splitk_floats: floats of scratch a split-K run of `desc` wants; 0 = splitting would not help
 *       (call ccvpe_conv_igemm_* instead); negative = error.
 *   ccvpe_conv_igemm_splitk_f32 / _bf16: same contract as ccvpe_conv_igemm_f32 / _bf16, with that scratch. */
int ccvpe_conv_igemm_splitk_floats(const ccvpe_conv_desc* desc, int is_bf16);
/* Which kernel the one-pass entry points (ccvpe_conv_igemm_f32 / _bf16) run for `desc`; launches nothing.  Returns
 * CCVPE_ROUTE_* | MT << 8 | NT << 12 | WN << 16 (workgroup tile = 16*MT*(4/WN) pixels x 16*NT*WN channels), or a negative
 * error code.  For tests ("did this shape reach the pointwise kernel?") and for bench.py's per-kernel accounting. */
#define CCVPE_ROUTE_IGEMM 0   /* generic gather kernel */
#define CCVPE_ROUTE_PW_GEMM 1 /* deep-stage pointwise kernel (1x1, stride 1, one source) */
#define CCVPE_ROUTE_CONV3X3 2 /* LDS-halo 3x3 kernel */
#define CCVPE_ROUTE_C3N 3     /* bf16 narrow 3x3 kernel: weights in registers, persistent workgroups (csrc/narrow_impl.h) */
#define CCVPE_ROUTE_PW_RING 4 /* pointwise kernel with a three-stage LDS-DMA ring, two workgroups per CU, fp32 and bf16 (csrc/conv_pw2_impl.h) */
#define CCVPE_ROUTE_PWN 5     /* (ABI 7) narrow projection kernel: N <= 48, K <= 256 (fp32: <= 144), weights x SE gate in registers, waves stream 16-pixel tiles (csrc/pwn.hip); the route's MT / NT fields hold N / 16 and K bytes / 64 */
int ccvpe_conv_igemm_route(const ccvpe_conv_desc* desc, int is_bf16, int out_f32);
/* A/B switch for measurements (process-wide, default on): 0 sends the narrow bf16 decoder layers (CCVPE_ROUTE_C3N, and the
 * narrow form of ccvpe_upconv3x3_bf16) back to the tiled kernels.  Returns the previous setting. */
int ccvpe_set_narrow_kernels(int on);
/* The same kind of switch for CCVPE_ROUTE_PW_RING: 0 sends those fp32 pointwise layers back to CCVPE_ROUTE_PW_GEMM (bit-identical
 * results either way).  Returns the previous setting. */
int ccvpe_set_pw_ring_kernels(int on);
/* (ABI 7) The same kind of switch for CCVPE_ROUTE_PWN: 0 sends the narrow bf16 projections back to the generic kernel. */
int ccvpe_set_pwn_kernels(int on);
/* (ABI 7) The same kind of switch for the late-block MBConv front (csrc/mbconv_plane.hip: planes of <= 1024 pixels, Cin in
 * {80, 112, 192}), which runs behind the UNCHANGED ccvpe_mbconv_front_* / ccvpe_dwconv_* entry points and their *_nblk queries:
 * bit 0 = the fused expand + depthwise form (ccvpe_mbconv_front_nblk reports it), bit 1 = the depthwise-only form
 * (ccvpe_dwconv_*), bit 2 = the band-owner kernel for the fused form in bf16 storage (producer / consumer waves).  0 restores the round-5 chain (pointwise GEMM + dwconv_plane_kernel).  Returns the previous mode; change
 * it only between forwards (the *_nblk queries size the caller's squeeze-partial buffers). */
int ccvpe_set_mbconv_plane_kernels(int mode);
/* A 3x3 layer (desc: bf16 storage, one source, bias, no activation — convK.2 of double_conv, models.py:42-47) WITH the next
 * level's rotational matching in its epilogue, for ONE rotation hypothesis (the localisation branch of CVM_VIGOR_ori_prior(0):
 * models.py:489-500 with k = 0): replaces  x = convK.2(y);  scores, cat = match(x)  (models.py:211-228 pattern) — x never reaches
 * HBM.  desc->n = channels of x, desc->dst / ldd = the decoder input rows [x / max(|x|, 1e-12) | score | 0-pad] (ldd >= n + 1;
 * bf16, or fp32 with out_f32), scores [B,1,H,W] fp32; g [B][ldg] fp32 ground descriptor (first L entries), shift / stride /
 * window_offset as ccvpe_match_level.  Same formulas as ccvpe_match_level (no eps in the cosine).  _ok: 1 if the library serves
 * this layer / shape (narrow-level kernel, csrc/narrow_impl.h), else 0 — then run the two calls separately. */
int ccvpe_conv3x3_match1_ok(const ccvpe_conv_desc* desc, int out_f32, int L);
int ccvpe_conv3x3_match1_bf16(const ccvpe_conv_desc* desc, int out_f32, const float* g, int ldg, int L, int shift, int stride,
                              int window_offset, float* scores, void* stream);
int ccvpe_conv_igemm_splitk_f32(const ccvpe_conv_desc* desc, float* scratch, void* stream);

/* -------------------------------------------------------------------------------------------
 * ConvTranspose2d(k=2,s=2) folded into the 3x3 conv that follows it (models.py:207-209 and the same
 * pair at every decoder level of both branches: x = deconvK(x); x = cat[x, skip]; x = convK.0(x)).
 * Per output parity (py,px) the pair is one implicit GEMM over the LOW-RES pixels with 2x2 taps on
 * src0 and the ordinary 3x3 taps (stride 2, parity offset) on the skip:
 *   K = 4*c0 + 9*c1   instead of   9*(Cd + c1) + the deconv GEMM, and no upsampled intermediate.
 * w: [4][npad16][kpad] effective weights, parity-major ((py*2+px)), k = (du*2+dv)*c0 + ch for the
 * low-res source then 4*c0 + (ky*3+kx)*c1 + ch for the skip.  shift9: [9][n] = conv bias + the
 * deconv-bias terms of the 3x3 taps that fall inside the image, indexed by
 * (row class*3 + col class), class 0 = first row/col, 1 = interior, 2 = last.
 * ----------------------------------------------------------------------------------------- */
typedef struct ccvpe_upconv_desc {
  const void* src0;   /* [B,h1,w1,ld0] low-res input of the deconv (fp32 or bf16)   */
  const void* src1;   /* [B,2h1,2w1,ld1] skip, or NULL                              */
  const void* w;      /* packed effective weights                                    */
  const float* shift9;
  void* dst;          /* [B,2h1,2w1,ldd]                                             */
  int c0, ld0, c1, ld1;
  int batch, h1, w1;
  int n, kpad, ldd;
  int act;
} ccvpe_upconv_desc;

int ccvpe_upconv3x3_f32(const ccvpe_upconv_desc* desc, void* stream);
int ccvpe_upconv3x3_bf16(const ccvpe_upconv_desc* desc, void* stream);
/* Which kernel ccvpe_upconv3x3_f32 / _bf16 run for `desc`; launches nothing.  Returns CCVPE_UPROUTE_* | MT << 8 | NT << 12 |
 * WN << 16 (tile bits zero for CCVPE_UPROUTE_UP2), or a negative error code.  For tests and bench.py's per-kernel accounting. */
#define CCVPE_UPROUTE_GATHER 0   /* upconv_kernel: linear-M gather (low-res images narrower than 16 pixels) */
#define CCVPE_UPROUTE_HALO 1     /* upconv_halo_kernel: low-res halo in LDS, skip gathered, W through registers */
#define CCVPE_UPROUTE_DMA 2      /* upconv_dma_kernel (bf16): pair-of-taps stages, W by LDS-DMA */
#define CCVPE_UPROUTE_DMA_PAIR 3 /* upconv_dma_kernel, two 8 x 8 images per tile (level 6) */
#define CCVPE_UPROUTE_UP2 4      /* up2_kernel (bf16, narrow levels): four parities per workgroup, weights in registers */
int ccvpe_upconv3x3_route(const ccvpe_upconv_desc* desc, int is_bf16);

/* -------------------------------------------------------------------------------------------
 * The whole 512 x 512 level of a decoder in one launch (csrc/tail512.hip):
 *   loc: x = deconv1(x); x = conv1(x)  [3x3 + ReLU + 3x3, 16 -> 16 -> 1]      models.py:124-127 (logits, :319)
 *   ori: x = deconv1_ori(x); x = conv1_ori(x) [16 -> 16 -> 2]; F.normalize    models.py:145-148,341
 * x [B,h1,w1,ld0] low-res input (c0 channels incl. zero padding; fp32 or bf16), w / shift9 as for ccvpe_upconv3x3
 * with n = 16 and c1 = 0 ([4][16][kpad], [9][16]), w2 [cout][3][3][16] fp32, b2 [cout], out [B,cout,2h1,2w1] fp32
 * (NCHW).  The 16-channel intermediate is never written.  h1, w1 multiples of 16.  Same results as ccvpe_upconv3x3 +
 * ccvpe_head_conv3x3 (fp32: to rounding; bf16: better — the intermediate is not rounded to bf16).
 * split (ccvpe_tail512_f32 only, cout = 1): 0 = exact fp32 matrix arithmetic; 1 = the fp32 operands are split into bf16 hi + lo
 * planes and multiplied on the bf16 matrix cores (hi.hi + lo.hi + hi.lo, fp32 accumulate: ~1e-5 of scale instead of bit-level
 * fp32) — the fp32 tail of the bf16 STORAGE path (set_precision("bf16")), never used by the fp32 path.
 * ----------------------------------------------------------------------------------------- */
typedef struct ccvpe_tail_desc {
  const void* x;
  const void* w;
  const float* shift9;
  const float* w2;
  const float* b2;
  float* out;
  int batch, h1, w1;
  int c0, ld0, kpad;
  int cout, normalize, split;
  float* softmax_partial;   /* cout = 1, optional (NULL: not written): [B][ccvpe_tail512_partials()][2] = (max, sum exp(l - max)) of the
                               logits of every (tile, wave) — the heat-map softmax (models.py:319-320) then needs no row-wide max /
                               sum sweep: ccvpe_softmax_apply_f32 */
} ccvpe_tail_desc;

int ccvpe_tail512_f32(const ccvpe_tail_desc* desc, void* stream);
int ccvpe_tail512_bf16(const ccvpe_tail_desc* desc, void* stream);
int ccvpe_tail512_partials(const ccvpe_tail_desc* desc, int is_bf16);   /* partial pairs per sample for this descriptor (launches nothing) */
/* Softmax(dim=-1) over rows of n logits from n_partials (max, sum exp) pairs per row (models.py:320): out = exp(l - M) / S. */
int ccvpe_softmax_apply_f32(const float* logits, const float* partials, int n_partials, float* out, int rows, int n,
                            void* stream);

/* -------------------------------------------------------------------------------------------
 * The WHOLE eval forward behind one entry point (csrc/plan.hip) — replaces the call sites
 *   logits, heatmap, ori, score1..6 = model(grd, sat)      train_VIGOR.py:252-254,282 / train_KITTI.py:285-287,302 /
 *                                                           visualize_qualitative_results_VIGOR.py:68-71
 * for a model in eval mode (models.py:150-343, :448-652, :752-950, :954-1246).
 * A ctx is created from a PLAN: the launch list of one forward for fixed (model, weights, batch, image sizes, precision),
 * recorded and serialised by ccvpe_amd/plan.py (or read from a file the Python side wrote: tools/plan_run.cpp is a caller
 * without Python).  plan: the serialised bytes (header, calls, descriptor blobs, packed weights).
 * weights_dev / workspace_dev: device buffers of at least the plan's weights / workspace size (ccvpe_ctx_info on a ctx, or the
 * Python Plan object) owned by the CALLER — or NULL: the library allocates and frees them.  The weights are uploaded here.
 * ccvpe_forward: grd [B,3,h,w], sat [B,3,H,W] NCHW fp32 device tensors of the recorded shapes; every kernel is enqueued on
 * `stream` (capturable in a hipGraph); out (may be NULL) receives the device pointers of the nine outputs — fp32, NCHW-
 * contiguous, inside the ctx's workspace, valid until the next ccvpe_forward on this ctx.  One ctx serves one stream at a time.
 * Errors: int status, ccvpe_last_error() names the failing call.
 * ----------------------------------------------------------------------------------------- */
typedef struct ccvpe_ctx ccvpe_ctx;
int ccvpe_ctx_create(const void* plan, long long n_bytes, void* weights_dev, void* workspace_dev, ccvpe_ctx** ctx);
int ccvpe_ctx_destroy(ccvpe_ctx* ctx);
int ccvpe_ctx_info(const ccvpe_ctx* ctx, long long* workspace_bytes, long long* weights_bytes, long long* grd_bytes,
                   long long* sat_bytes, int* n_outputs, int* n_calls);
int ccvpe_ctx_output(const ccvpe_ctx* ctx, int i, void** dev_ptr, long long* bytes, int* ndim, long long* dims4,
                     long long* strides4);   /* strides in ELEMENTS: an output may be a view (ori_prior's score1 is a channel slice) */
int ccvpe_forward(ccvpe_ctx* ctx, const void* grd, const void* sat, void** out, void* stream);

/* -------------------------------------------------------------------------------------------
 * EfficientNet stem: 3x3 stride-2 conv on the NCHW image + folded BN + swish, NHWC out.
 * TF-"SAME" padding (0 before, 1 after) from the 224 schedule; `circular` wraps along W and
 * zero-pads along H.  efficientnet_pytorch/model.py:181-182,289; utils.py:254-282,318-358.
 *   x [B,3,H,W] NCHW, w [3][3][3][32] (ky,kx,ci,co), y [B,H/2,W/2,32]
 * ----------------------------------------------------------------------------------------- */
int ccvpe_stem_conv_f32(const float* x_nchw, const float* w, const float* scale, const float* shift,
                        float* y, int batch, int in_h, int in_w, int circular, void* stream);

/* -------------------------------------------------------------------------------------------
 * Depthwise kxk conv (k in {3,5}, stride in {1,2}) + folded BN + swish, and the squeeze half of
 * squeeze-excite: per-(sample, channel) partial sums of the activated output, one row per
 * workgroup (fixed order => deterministic), reduced later by ccvpe_se_gate_f32.
 * efficientnet_pytorch/model.py:70-73,108-110,114; padding as for the stem
 * (stride 1: (k-1)/2 both sides; stride 2: k=3 (0,1), k=5 (1,2)).
 *   x [B,H,W,C] (ld=C), w [k][k][C], y [B,Ho,Wo,C], se_partial [B][nblk][C]
 * ccvpe_dwconv_nblk() returns nblk for a given shape (so the caller can size se_partial).
 * ----------------------------------------------------------------------------------------- */
int ccvpe_dwconv_nblk(int in_h, int in_w, int channels, int stride);
int ccvpe_dwconv_f32(const float* x, const float* w, const float* scale, const float* shift, float* y,
                     float* se_partial, int batch, int in_h, int in_w, int channels, int k, int stride,
                     int circular, void* stream);

/* -------------------------------------------------------------------------------------------
 * Stem + MBConv block 0 front half in one launch (eval): ccvpe_stem_conv + ccvpe_dwconv(k = 3, stride 1, 32
 * channels) + the SE squeeze partials, with the [B, H/2, W/2, 32] stem tensor kept in LDS
 * (efficientnet_pytorch/model.py:289 then :108-110,114 of block 0, which has no expand conv).  fp32: the
 * same bits as the two unfused calls.
 * ccvpe_stem_dw_nblk() returns the number of SE partial rows per sample (8 x 32 output tiles), 0 if the shape
 * is not supported (circular padding with an odd width): then use the two unfused calls.
 *   x_nchw [B,3,H,W] fp32, w [27][32], w_dw [3][3][32], y [B,Ho,Wo,32], se_partial [B][nblk][32]
 * ----------------------------------------------------------------------------------------- */
int ccvpe_stem_dw_nblk(int in_h, int in_w, int circular);
int ccvpe_stem_dw_f32(const float* x_nchw, const float* w, const float* s0, const float* b0, const float* w_dw,
                      const float* s1, const float* b1, float* y, float* se_partial, int batch, int in_h,
                      int in_w, int circular, void* stream);
int ccvpe_stem_dw_bf16(const float* x_nchw, const float* w, const float* s0, const float* b0, const float* w_dw,
                       const float* s1, const float* b1, void* y, float* se_partial, int batch, int in_h,
                       int in_w, int circular, void* stream);

/* -------------------------------------------------------------------------------------------
 * Fused MBConv front half for the early (large) blocks: expand 1x1 + BN0 + swish -> depthwise kxk +
 * BN1 + swish + SE squeeze partials, with the 6x-expanded tensor kept in LDS
 * (efficientnet_pytorch/model.py:102-110,114).  Same results as ccvpe_conv_igemm_f32 followed by
 * ccvpe_dwconv_f32.  w_exp is the igemm-packed expand weight [mid^16][kpad].
 * ccvpe_mbconv_front_nblk() returns the number of SE partial rows per sample, 0 if the shape is
 * not supported (then use the two unfused calls), <0 on bad arguments.
 *   x [B,H,W,cin], y [B,Ho,Wo,mid], se_partial [B][nblk][mid]
 * ----------------------------------------------------------------------------------------- */
int ccvpe_mbconv_front_nblk(int in_h, int in_w, int cin, int mid, int k, int stride);
/* (ABI 7) which kernel the fused call runs for a shape — 0 none (use the unfused calls), 1 mbconv_front_kernel (early blocks),
 * 2 mbconv_plane_kernel, 3 mbconv_band_kernel (late blocks: planes of <= 1024 pixels; 3 = bf16 storage only).  Reporting only. */
int ccvpe_mbconv_front_route(int in_h, int in_w, int cin, int mid, int k, int stride, int is_bf16, int batch);
int ccvpe_mbconv_front_f32(const float* x, const float* w_exp, int kpad, const float* s0, const float* b0,
                           const float* w_dw, const float* s1, const float* b1, float* y, float* se_partial,
                           int batch, int in_h, int in_w, int cin, int mid, int k, int stride, int circular,
                           void* stream);

/* Squeeze-excite gate: mean -> 1x1 (C->Cs) + swish -> 1x1 (Cs->C) -> sigmoid.
 * efficientnet_pytorch/model.py:113-118.  w1 [Cs][C], w2 TRANSPOSED [Cs][C], gate [B][C]. */
int ccvpe_se_gate_f32(const float* se_partial, int nblk, float inv_hw, const float* w1, const float* b1,
                      const float* w2, const float* b2, float* gate, int batch, int channels,
                      int squeezed, void* stream);

/* -------------------------------------------------------------------------------------------
 * Ground descriptors: height collapse of the fused 1x1-conv output.
 *   D_l[b, x*Cd_l + c] = sum_y wh_l[y] * y1[b,y,x,off_l + c] + bh_l      (models.py:57-97)
 * y1 [B,h,w,ld] holds all six heads' channels side by side (conv bias already added);
 * out [B][sum_l w*Cd_l], level l at offset w*off_l.  wh [6][h], bh [6], cd [6].
 * ----------------------------------------------------------------------------------------- */
int ccvpe_ground_descriptor_f32(const float* y1, int ld, const float* wh, const float* bh, const int* cd,
                                float* out, int batch, int h, int w, void* stream);

/* (ABI 7) A/B switch of the matrix-core form of ccvpe_match_level_* (9 <= n_shifts <= 32, even table offsets: the circulant
 * [n_shifts x C] of the ground descriptor times the [C x pixels] tile as v_mfma_f32_16x16x4_f32, exact fp32).  0 = the vector-ALU
 * form for every configuration, 1 = the tiled matrix-core form, 2 (default) = additionally the streaming form of the narrow levels
 * (C <= 80, no tail scores, >= 65 536 pixels: every wave streams 16-pixel tiles, no LDS).  Returns the previous setting. */
int ccvpe_set_match_mfma(int on);
/* -------------------------------------------------------------------------------------------
 * Rotational matching + LMU concat, fused (models.py:186-205 and the five blocks after it;
 * ori_prior :484-514; KITTI :788-806):
 *   score[b,i,p] = sum_{c<L} g[b,c] * X[b,p,(c + shift_i*stride) mod C]
 *                  / ( ||X[b,p,window_i]||_2 * ||g[b]||_2 )               (no eps, as upstream)
 *   dstx[b,p,0:C]        = X[b,p,:] / max(||X[b,p,:]||_2, 1e-12)
 *   dstx[b,p,C]          = max_{i < n_max} score[b,i,p]
 *   dstx[b,p,C+1+j]      = score[b, n_max_first + j, p]   j < n_tail   (level-1 orientation input)
 *   dstx[b,p,C+1+n_tail : ldo] = 0
 * X [B,H*W,ldx]; g [B,ldg] (first L entries); scores [B,n_shifts,H*W] (NCHW, returned to caller).
 * The first n_max shifts feed the max; the last n_tail shifts are also copied into dstx.
 * window_offset: first channel of the L-wide window inside the rolled volume — 0 for VIGOR / KITTI, int(C/2 - L/2)
 * for CVM_OxfordRobotCar's centred window (models.py:1094): window_i[c] = X[(c + window_offset + shift_i*stride) mod C].
 * ----------------------------------------------------------------------------------------- */
int ccvpe_match_level_f32(const float* x, int ldx, const float* g, int ldg, int L, const int* shifts,
                          int n_shifts, int n_max, int n_tail, int stride, int window_offset, float* scores,
                          float* dstx, int ldo, int batch, int hw, int channels, void* stream);

/* -------------------------------------------------------------------------------------------
 * Final 3x3 conv to 1 or 2 channels, NCHW out (models.py:125-127 conv1.2, :146-148 conv1_ori.2);
 * normalize!=0 divides the 2-vector by max(norm,1e-12) (models.py:341).
 * x [B,H,W,16] NHWC, w [cout][3][3][16], out [B,cout,H,W].
 * ----------------------------------------------------------------------------------------- */
int ccvpe_head_conv3x3_f32(const float* x, const float* w, const float* bias, float* out_nchw, int batch,
                           int h, int w_, int cout, int normalize, void* stream);

/* Row softmax over n columns (models.py:319-320: Softmax over the 262144 flattened logits). */
int ccvpe_softmax_rows_f32(const float* logits, float* out, int rows, int n, void* stream);
/* bf16 -> fp32 widening of an activation tensor (n_elems % 8 == 0, 16-byte aligned): the bf16 storage path runs its last
 * localisation-decoder levels through the fp32 kernels (models.py:299-320 decide the heat-map arg-max). */
/* n device-to-device copies of fp32 vectors (counts[i] floats from srcs[i] to dsts[i]) in ceil(n / 96) launches; the three
 * arrays are HOST arrays, read before the call returns.  Gradients -> slots of the flat all-reduce arena (harness.py). */
int ccvpe_multi_copy_f32(const void* const* srcs, void* const* dsts, const int* counts, int n, void* stream);
/* Train-mode weight re-pack in ONE launch (train_VIGOR.py:146-150: optimizer.step() rewrites every weight, so every step
 * re-lays all of them out for the forward and backward GEMMs).  n_chunks chunks; chunk c writes counts[c] (<=
 * ccvpe_gather_repack_chunk()) floats to dsts[c]: element i = idx[idx_off[c] + i] > 0 ? srcs[c][idx[idx_off[c] + i] - 1] : 0.
 * All five arrays are DEVICE arrays (built once per parameter placement by ccvpe_amd/repack.py). */
int ccvpe_gather_repack_f32(void* const* dsts, const void* const* srcs, const long long* idx_off, const int* counts,
                            const int* idx, int n_chunks, void* stream);
int ccvpe_gather_repack_chunk(void);
int ccvpe_cast_bf16_f32(const void* src, float* dst, long n_elems, void* stream);

/* -------------------------------------------------------------------------------------------
 * Evaluation post-processing on the device (train_VIGOR.py:294-324, train_KITTI.py:304-343): per
 * sample the arg-max pixel of the heat-map (first maximum, as numpy.argmax), the orientation
 * vector there and the acos-based angle in degrees — 6 floats per sample instead of a 3 MB D2H
 * copy of the full maps:   out[b] = { y, x, cos, sin, angle_deg (NaN if |cos|>1 or |sin|>1), prob }.
 *   heatmap [B,h*w], ori [B,2,h*w] (NCHW), out [B,6]
 * ----------------------------------------------------------------------------------------- */
int ccvpe_eval_postprocess_f32(const float* heatmap, const float* ori, float* out, int batch, int h, int w,
                               void* stream);

/* -------------------------------------------------------------------------------------------
 * Losses (losses.py:4-29), device scalars out.  `acc` is caller-provided scratch.
 *   infoNCE: accumulates num = sum_{label>1e-2} log(softmax(s/T))*label and den = sum label over
 *            the whole batch; loss = -num/den.            scores/labels [B,n]
 *            ONE pass over the data with (chunk, sample) workgroups.  rows [4*B + 4] floats out: per sample
 *            (num_b, den_b, z_b = sum exp(s/T), 0), then the batch label mass — the backward reads them instead of
 *            recomputing row statistics.  scratch: ccvpe_infonce_scratch_floats(batch, n) floats.
 *   CE     : -sum(labels*log_softmax(logits))/B           [B,n]
 *   ori    : sum(||gt_ori-ori||^2 * gt)/B                 ori/gt_ori [B,2,hw], gt [B,hw]
 * ----------------------------------------------------------------------------------------- */
int ccvpe_infonce_scratch_floats(int batch, int n);
int ccvpe_infonce_loss_f32(const float* scores, const float* labels, float temperature, float* loss,
                           float* rows, float* scratch, int batch, int n, void* stream);
int ccvpe_cross_entropy_loss_f32(const float* logits, const float* labels, float* loss, float* scratch,
                                 int batch, int n, void* stream);
int ccvpe_orientation_loss_f32(const float* ori, const float* gt_ori, const float* gt, float* loss,
                               float* scratch, int batch, int hw, void* stream);

/* -------------------------------------------------------------------------------------------
 * Train-mode pieces (efficientnet_pytorch/model.py:63,73,87,182,210: BatchNorm2d with batch statistics,
 * momentum 0.01, eps 1e-3; utils.py:129-154 drop_connect).  In train mode the conv kernels write their
 * raw output (ccvpe_conv_igemm_f32 with scale = shift = NULL, ccvpe_stem_conv_raw_f32,
 * ccvpe_dwconv_raw_f32), then:
 *   ccvpe_bn_stats_f32 : per-channel mean and BIASED variance of x [rows, channels] (shifted-sum partials per
 *       workgroup merged in two levels with Chan's formula: deterministic); if run_mean/run_var are given they
 *       are updated in place: run = (1-momentum)*run + momentum*(mean | UNBIASED variance).
 *       scratch: 3 * channels * ccvpe_bn_stats_nblk(rows) floats.
 *   ccvpe_bn_act_f32   : y = act((x-mean)/sqrt(var+eps)*gamma+beta) [* dc_scale[b]] [+ residual]; with
 *       se_partial != NULL also writes per-(sample, workgroup) channel sums of y
 *       [batch][ccvpe_bn_act_nblk(rows_per_sample)][channels] for ccvpe_se_gate_f32.
 * ----------------------------------------------------------------------------------------- */
int ccvpe_stem_conv_raw_f32(const float* x_nchw, const float* w, float* y, int batch, int in_h, int in_w,
                            int circular, void* stream);
int ccvpe_dwconv_raw_f32(const float* x, const float* w, float* y, int batch, int in_h, int in_w, int channels,
                         int k, int stride, int circular, void* stream);
int ccvpe_bn_stats_nblk(int rows);
int ccvpe_bn_stats_f32(const float* x, int rows, int channels, float* mean, float* var, float* run_mean,
                       float* run_var, float momentum, float* scratch, void* stream);
int ccvpe_bn_act_nblk(int rows_per_sample);
int ccvpe_bn_act_f32(const float* x, const float* mean, const float* var, const float* gamma, const float* beta,
                     float eps, int act, const float* residual, const float* dc_scale, float* y,
                     float* se_partial, int batch, int rows_per_sample, int channels, void* stream);

/* -------------------------------------------------------------------------------------------
 * Backward building blocks (wired into one torch.autograd.Function for the whole model by ccvpe_amd/train.py).
 *   ccvpe_conv_wgrad_f32: weight gradient of any dense conv of the path,
 *       dw[n][tap][c] = sum_pixels dy[pixel][n] * x[pixel*stride - pad + tap][c]     (layout O,kh,kw,I)
 *     x given as one or two concatenated NHWC sources (like the forward), dy [batch,Ho,Wo,ldy].
 *     scratch: ccvpe_conv_wgrad_scratch_floats(...) floats.  ConvTranspose2d(k2,s2): call it with
 *     x := the (high-res) output gradient, dy := the (low-res) forward input, k=2, stride=2.
 *   input gradients need no new kernel: they are ccvpe_conv_igemm_f32 calls with re-packed weights
 *     (1x1: W^T; 3x3: taps flipped, in/out swapped; deconv <-> conv 2x2 stride 2) — ccvpe_amd/backward.py.
 *   ccvpe_colsum_f32: per-channel column sums (bias gradients); scratch ceil(rows/256)*channels floats.
 * ----------------------------------------------------------------------------------------- */
int ccvpe_conv_wgrad_tile(int n, int ncols);   /* (tile rows << 16) | tile columns the launch will use (reporting) */
int ccvpe_conv_wgrad_scratch_floats(int batch, int in_h, int in_w, int kh, int kw, int stride, int pad, int ctot,
                                    int n);
int ccvpe_conv_wgrad_f32(const float* src0, int c0, int ld0, const float* src1, int c1, int ld1, const float* dy,
                         int ldy, float* dw, float* scratch, int batch, int in_h, int in_w, int kh, int kw,
                         int stride, int pad, int n, void* stream);
/* the same launch also returns the conv's BIAS gradient dbias[n] = sum_pixels dy[pixel][n]: the first column tile of every row
 * tile adds up the dY pieces it stages anyway (one pass over dY instead of a ccvpe_colsum_f32 pass). */
int ccvpe_conv_wgrad_bias_f32(const float* src0, int c0, int ld0, const float* src1, int c1, int ld1, const float* dy,
                              int ldy, float* dw, float* dbias, float* scratch, int batch, int in_h, int in_w, int kh,
                              int kw, int stride, int pad, int n, void* stream);
/* 1x1 conv whose input is the squeeze-excite product x[b, px, c] * gate[b, c] (MBConv projection, model.py:118-121;
 * fewer than 128 output channels): the gate is applied while x is staged, the gated tensor is never materialised. */
int ccvpe_conv_wgrad_gated_f32(const float* x, int c, int ld, const float* gate, const float* dy, int ldy, float* dw,
                               float* scratch, int batch, int in_h, int in_w, int n, void* stream);
int ccvpe_colsum_f32(const float* x, int rows, int channels, int ld, float* out, float* scratch, void* stream);

/* -------------------------------------------------------------------------------------------
 * Backward of the train-mode EfficientNet pieces (csrc/train_bwd.hip).
 *   ccvpe_bn_act_bwd_f32 : backward of ccvpe_bn_act_f32 (batch-statistics BN + activation [* dc_scale]).
 *       dv is the gradient w.r.t. the consumer's input; with gate/dmean ([batch][channels]) the SE product
 *       and average-pool branches are folded in: du = dv*gate + dmean.  Writes dx (w.r.t. the raw conv
 *       output), dgamma, dbeta.  scratch: 2*channels*batch*ccvpe_bn_bwd_nblk(rows_per_sample) floats.
 *       The residual branch's gradient is dv itself.
 *   ccvpe_se_dgate_f32   : partial sums [batch][ccvpe_bn_bwd_nblk()][channels] of dv * u, u = act(bn(x)).
 *   ccvpe_se_bwd_f32     : the two SE 1x1 convs; consumes the forward squeeze partials and the dgate partials,
 *       writes dmean [batch][channels] (already / HW), dw1 [Cs][C], db1, dw2 [C][Cs], db2.
 *       scratch: 2*batch*(channels+squeezed) floats.
 *   ccvpe_dwconv_dgrad_f32 / ccvpe_dwconv_wgrad_f32 : depthwise conv input / weight gradients (dw [k][k][C]);
 *       wgrad scratch: batch*ccvpe_dwconv_wgrad_nblk()*k*k*channels floats.
 *   ccvpe_relu_bwd_f32   : dx = dy * (y > 0).
 * ----------------------------------------------------------------------------------------- */
int ccvpe_bn_bwd_nblk(int rows_per_sample);
int ccvpe_bn_act_bwd_f32(const float* x, const float* dv, const float* mean, const float* var, const float* gamma,
                         const float* beta, const float* gate, const float* dmean, const float* dc_scale, float eps,
                         int act, float* dx, float* dgamma, float* dbeta, float* scratch, int batch,
                         int rows_per_sample, int channels, void* stream);
/* BatchNorm + squeeze-excite backward in two passes over (x, dv) instead of three (csrc/train_bwd.hip): the reduce pass
 * returns A [5][batch][channels] = per-(sample, channel) sums of (dv*u, dv*a', a', dv*a'*xhat, a'*xhat); A[0] is the SE gate
 * gradient (feed it to ccvpe_se_bwd_f32 as a one-row partial), and the apply call finishes dbeta / dgamma from A, gate and
 * dmean before writing dx.  scratch: batch * ccvpe_bn_bwd_nblk(rows_per_sample) * 5 * channels floats. */
int ccvpe_se_bn_bwd_reduce_f32(const float* x, const float* dv, const float* mean, const float* var, const float* gamma,
                               const float* beta, float eps, int act, float* A, float* scratch, int batch,
                               int rows_per_sample, int channels, void* stream);
int ccvpe_se_bn_bwd_apply_f32(const float* x, const float* dv, const float* mean, const float* var, const float* gamma,
                              const float* beta, const float* gate, const float* dmean, float eps, int act, const float* A,
                              float* dx, float* dgamma, float* dbeta, int batch, int rows_per_sample, int channels,
                              void* stream);
int ccvpe_se_dgate_f32(const float* x, const float* dv, const float* mean, const float* var, const float* gamma,
                       const float* beta, float eps, int act, float* part, int batch, int rows_per_sample,
                       int channels, void* stream);
int ccvpe_se_bwd_f32(const float* se_partial, int nblk, float inv_hw, const float* dgate_partial, int ndg,
                     const float* w1, const float* b1, const float* w2t, const float* b2, float* dmean, float* dw1,
                     float* db1, float* dw2, float* db2, float* scratch, int batch, int channels, int squeezed,
                     void* stream);
int ccvpe_dwconv_dgrad_f32(const float* dy, const float* w, float* dx, int batch, int in_h, int in_w, int channels,
                           int k, int stride, int circular, void* stream);
int ccvpe_dwconv_wgrad_nblk(int in_h, int in_w, int k, int stride);
int ccvpe_dwconv_wgrad_f32(const float* x, const float* dy, float* dw, float* scratch, int batch, int in_h, int in_w,
                           int channels, int k, int stride, int circular, void* stream);
int ccvpe_relu_bwd_f32(const float* y, const float* dy, float* dx, int n_elems, void* stream);
/* v = u * gate[b,c] materialised (the projection conv's weight gradient reads the gated tensor). */
int ccvpe_gate_mul_f32(const float* u, const float* gate, float* v, int batch, int rows_per_sample, int channels,
                       void* stream);

/* -------------------------------------------------------------------------------------------
 * Backward of the head / glue operators (csrc/heads_bwd.hip, csrc/matching_bwd.hip).
 *   ccvpe_softmax_bwd_f32   : dlogits = heatmap * (dheatmap - <heatmap, dheatmap>) [+ dlogits_direct]  (models.py:320)
 *   ccvpe_l2norm2_bwd_f32   : backward of F.normalize over the (cos, sin) pair (models.py:341); raw [B,2,HW]
 *   ccvpe_head_conv3x3_bwd_f32 : backward of ccvpe_head_conv3x3_f32 with normalize = 0: x [B,H,W,16] NHWC,
 *       dout [B,cout,H,W]; writes dx [B,H,W,16], dw [cout][3][3][16], dbias [cout].
 *       scratch: CCVPE_HEAD_BWD_SCRATCH floats.
 *   ccvpe_ground_descriptor_bwd_f32 : backward of ccvpe_ground_descriptor_f32: dout [B][w*sum(cd)] ->
 *       dy1 [B,h,w,ld] (pad columns zeroed), dwh [6][h], dbh [6].
 *   ccvpe_add_cols_f32      : dst[row, 0:channels] (+)= src[row, col_off : col_off+channels]  (skip-connection
 *       gradients are column slices of the decoder convs' input gradients).
 *   ccvpe_stem_conv_wgrad_f32 : stem weight gradient dw [3][3][3][32]; scratch: 864 * ccvpe_stem_wgrad_nblk() floats.
 *   ccvpe_match_level_bwd_f32 : backward of ccvpe_match_level_f32.  scores = the forward's scores; dscores
 *       [B,n_shifts,HW] (may be NULL) is the gradient arriving at the returned score volume, ddst [B,HW,ldo]
 *       the gradient arriving at dstx (normalised features, max column, tail columns).  Writes dx [B,HW,lddx]
 *       (first `channels` columns) and dg [B,ldg_out] (first L entries).  The max routes its gradient to the
 *       first maximal shift, as torch.max does.  scratch: batch * ccvpe_match_bwd_nblk(hw, batch, channels) * (L+1) floats.
 * ----------------------------------------------------------------------------------------- */
#define CCVPE_HEAD_WGRAD_BLOCKS 1024
#define CCVPE_HEAD_BWD_SCRATCH ((CCVPE_HEAD_WGRAD_BLOCKS + 1) * 2 * 145)
int ccvpe_softmax_bwd_f32(const float* heatmap, const float* dheatmap, const float* dlogits_direct, float* dlogits,
                          int rows, int n, void* stream);
int ccvpe_l2norm2_bwd_f32(const float* raw, const float* dout, float* draw, int batch, int hw, void* stream);
int ccvpe_head_conv3x3_bwd_f32(const float* x, const float* w, const float* dout, float* dx, float* dw, float* dbias,
                               float* scratch, int batch, int h, int wd, int cout, int relu_mask_x, void* stream);
int ccvpe_ground_descriptor_bwd_f32(const float* y1, int ld, const float* wh, const int* cd, const float* dout,
                                    float* dy1, float* dwh, float* dbh, int batch, int h, int w, void* stream);
int ccvpe_add_cols_f32(const float* src, int ld_src, int col_off, float* dst, int ld_dst, int channels, int rows,
                       int accumulate, void* stream);
int ccvpe_stem_wgrad_nblk(int batch, int in_h, int in_w);
int ccvpe_stem_conv_wgrad_f32(const float* x_nchw, const float* dy, float* dw, float* scratch, int batch, int in_h,
                              int in_w, int circular, void* stream);
int ccvpe_match_bwd_nblk(int hw, int batch, int channels);
int ccvpe_match_level_bwd_f32(const float* x, int ldx, const float* g, int ldg, int L, const int* shifts, int n_shifts,
                              int n_max, int n_tail, int stride, int window_offset, const float* scores, const float* dscores,
                              const float* ddst, int ldo, float* dx, int lddx, float* dg, int ldg_out, float* scratch,
                              int batch, int hw, int channels, void* stream);

/* Loss gradients w.r.t. the prediction (losses.py:4-29); dloss = upstream scalar gradient (device pointer).
 * infoNCE: `rows` is what ccvpe_infonce_loss_f32 wrote for the same (scores, labels); one pass, no statistics recomputed. */
int ccvpe_infonce_loss_bwd_f32(const float* scores, const float* labels, float temperature, const float* dloss,
                               const float* rows, float* dscores, int batch, int n, void* stream);
int ccvpe_cross_entropy_loss_bwd_f32(const float* logits, const float* labels, const float* dloss, float* dlogits,
                                     int batch, int n, void* stream);
int ccvpe_orientation_loss_bwd_f32(const float* ori, const float* gt_ori, const float* gt, const float* dloss,
                                   float* dori, int batch, int hw, void* stream);

/* -------------------------------------------------------------------------------------------
 * Training-step glue on the device (csrc/train_glue.hip).
 *   ccvpe_train_targets_f32: the ground-truth tensors of datasets.py:145-166 (VIGOR) / :470-501 (KITTI) and
 *     train_VIGOR.py:120-128 from 3 scalars per sample instead of 24 MB of host tensors:
 *       center_xy [B][2] = (cx, cy) with x_j = -W/2 + cx + j*W/(W-1), y_i = -H/2 + cy + i*H/(H-1)
 *                          (VIGOR: cx = col_offset, cy = -row_offset; KITTI: cx = x_offset, cy = y_offset)
 *       angle_deg [B] (wrapped into [0, 360) by the kernel);  n_bins 20 (VIGOR) / 16 (KITTI);  sigma 4
 *     writes gt [B,1,H,W], gt_norm [B,H*W] (= gt / sum gt), gt_ori [B,2,H,W] (cos, sin) and the six max-pooled
 *     orientation-binned label maps lab_l [B,n_bins,H/k,W/k], k = 64,32,16,8,4,2 (gt_with_ori is never materialised).
 *     scratch: batch * ccvpe_train_targets_nblk(h, w) floats.
 *   ccvpe_adam_step_f32: torch.optim.Adam (no weight decay / amsgrad; train_VIGOR.py:104) for ALL parameter tensors in
 *     one launch.  table [n_tensors][5] int64 device array of (param, grad, exp_avg, exp_avg_sq, numel) — a NULL grad
 *     skips the tensor; hyper [n_tensors][ccvpe_adam_hyper_floats()] fp32 device array, per tensor
 *     (lr / (1 - beta1^t), beta1, beta2, 1 - beta1, 1 - beta2, eps, sqrt(1 - beta2^t), 0) with t that tensor's own
 *     1-based step count (bias corrections and 1 - beta formed in double on the host, as torch does);
 *     chunk_tensor / chunk_off [n_chunks] map each workgroup to (tensor, chunk of ccvpe_adam_chunk_elems() elements);
 *     grad_scale multiplies every gradient on load (1 / world after a SUM all-reduce; 1 otherwise).
 * ----------------------------------------------------------------------------------------- */
int ccvpe_train_targets_nblk(int h, int w);
int ccvpe_train_targets_f32(const float* center_xy, const float* angle_deg, int n_bins, float sigma, float* gt,
                            float* gt_norm, float* gt_ori, float* lab1, float* lab2, float* lab3, float* lab4,
                            float* lab5, float* lab6, float* scratch, int batch, int h, int w, void* stream);
/* The same with the order of the orientation bins as a parameter: ascending = 0 is ccvpe_train_targets_f32 (VIGOR / KITTI: the
 * two bins are n - index and n - index - 1, datasets.py:156-161 / :489-494), ascending = 1 the Oxford RobotCar loader's
 * (datasets.py:340-347: bins index and index + 1, n - 1 wrapping to 0). */
int ccvpe_train_targets_ordered_f32(const float* center_xy, const float* angle_deg, int n_bins, int ascending, float sigma,
                                    float* gt, float* gt_norm, float* gt_ori, float* lab1, float* lab2, float* lab3,
                                    float* lab4, float* lab5, float* lab6, float* scratch, int batch, int h, int w,
                                    void* stream);
int ccvpe_adam_chunk_elems(void);
int ccvpe_adam_hyper_floats(void);
int ccvpe_adam_step_f32(const void* table, const float* hyper, const int* chunk_tensor, const int* chunk_off, int n_chunks,
                        float grad_scale, void* stream);

/* -------------------------------------------------------------------------------------------
 * Input pipeline after JPEG decoding (csrc/preprocess.hip): transforms.Resize on a PIL image
 * (= Pillow's antialiased 8-bit BILINEAR resampler, reproduced bit for bit), ToTensor, Normalize
 * (train_VIGOR.py:57-70), torch.roll along W (datasets.py:112-121) and the FoV crop (train_VIGOR.py:177-178).
 *   src [in_h][in_w][3] uint8 (device), tmp [in_h][out_w][3] uint8 scratch, dst [3][out_h][keep_w] fp32 (one sample
 *   of the NCHW batch).  xbounds/ybounds [out][2] = (first source index, count), xcoef/ycoef [out][ksize] = Pillow's
 *   22-bit fixed-point coefficients (ccvpe_amd/preprocess.py builds them as precompute_coeffs() does).
 *   mean / std: HOST pointers to 3 floats.  roll: columns, as torch.roll's shift; keep_w <= out_w.
 * ----------------------------------------------------------------------------------------- */
int ccvpe_preprocess_u8_f32(const unsigned char* src, int in_h, int in_w, const int* xbounds, const int* xcoef,
                            int xksize, const int* ybounds, const int* ycoef, int yksize, unsigned char* tmp, float* dst,
                            int out_h, int out_w, int keep_w, int roll, const float* mean, const float* stdv,
                            void* stream);

/* -------------------------------------------------------------------------------------------
 * bf16 storage variants (BASELINE configs C2 / C4).  Same kernels instantiated for bf16 NHWC
 * activations and bf16 packed weights (kpad a multiple of 32), fp32 accumulation on
 * v_mfma_f32_16x16x32_bf16, fp32 scale/shift/gate/bias, round-to-nearest-even on store.  Pointers
 * typed `void*` address bf16 data; everything else is as in the _f32 entry point of the same name.
 * In ccvpe_conv_desc src0/src1/w/residual/dst then address bf16 data; `out_f32` != 0 makes the GEMM
 * write fp32 instead (tensors handed to fp32 consumers: the ground-descriptor collapse).
 * Matching scores, heat-map logits and the orientation field are always fp32.
 * ----------------------------------------------------------------------------------------- */
int ccvpe_conv_igemm_bf16(const ccvpe_conv_desc* desc, int out_f32, void* stream);
int ccvpe_conv_igemm_splitk_bf16(const ccvpe_conv_desc* desc, int out_f32, float* scratch, void* stream);
int ccvpe_stem_conv_bf16(const float* x_nchw, const float* w, const float* scale, const float* shift, void* y,
                         int batch, int in_h, int in_w, int circular, void* stream);
int ccvpe_dwconv_bf16(const void* x, const float* w, const float* scale, const float* shift, void* y,
                      float* se_partial, int batch, int in_h, int in_w, int channels, int k, int stride,
                      int circular, void* stream);
int ccvpe_mbconv_front_bf16(const void* x, const void* w_exp, int kpad, const float* s0, const float* b0,
                            const float* w_dw, const float* s1, const float* b1, void* y, float* se_partial,
                            int batch, int in_h, int in_w, int cin, int mid, int k, int stride, int circular,
                            void* stream);
int ccvpe_match_level_bf16(const void* x, int ldx, const float* g, int ldg, int L, const int* shifts,
                           int n_shifts, int n_max, int n_tail, int stride, int window_offset, float* scores, void* dstx, int ldo,
                           int batch, int hw, int channels, void* stream);
int ccvpe_head_conv3x3_bf16(const void* x, const float* w, const float* bias, float* out_nchw, int batch, int h,
                            int w_, int cout, int normalize, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CCVPE_HIP_H */
