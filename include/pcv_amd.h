/*
 * pcv_amd.h - C ABI of the MI355X (gfx950) conv-net inference hot path for the pytorchcv model zoo.
 *
 * The reference (osmr/pytorchcv 0.0.73) is pure Python: every op on its inference path reaches PyTorch ATen
 * through `torch.nn` modules. This library replaces those ATen ops, for the path only, with hand-written
 * HIP kernels. Each entry point below names the reference call site whose ATen op(s) it replaces
 * (paths relative to the reference root); `INTEGRATION.md` shows the ctypes stub a maintainer adds.
 *
 * Conventions
 *   - extern "C", plain C types, no C++/torch types; every function returns 0 on success or a negative
 *     pcv_status; `pcv_last_error(ctx)` gives the text. The library never throws and never calls back.
 *   - All data pointers are DEVICE pointers borrowed for the duration of the call. Work is enqueued on
 *     `stream` (a hipStream_t passed as void*; NULL = the null stream) and the call returns without
 *     synchronising. `pcv_conv_pack`/`pcv_dwconv_pack` additionally upload a small host-built table with a
 *     synchronous copy: they are weight-load-time functions, not hot-path ones.
 *   - Activations are NHWC ("pixels x channels", channels contiguous) in `dtype`; the ABI never sees NCHW
 *     except in the two layout-conversion entry points. fp32 is only used for scale/shift/bias/gates/logits.
 *   - A context belongs to one device; the library is re-entrant per context.
 */
#ifndef PCV_AMD_H
#define PCV_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: pcv_conv_desc starts with struct_size and ends with y_cpitch (version 1 had neither check; a binding built against
 * another layout is refused with PCV_ERR_INVALID instead of being read past its end)
 * 3: + pcv_fp16_guard_begin / pcv_fp16_guard_end / pcv_fp16_overflow_count, + pcv_rccl_* (no layout change)
 * 4: no signature or struct change; the packed-weight blobs of 16-bit depthwise 3x3 layers and of the dense layers listed at
 *    pcv_conv_pack carry a second table, so a blob packed by a version-3 library is too short (size blobs with *_packed_bytes) */
#define PCV_ABI_VERSION 4

typedef struct pcv_ctx pcv_ctx;

typedef enum pcv_status {
    PCV_OK = 0,
    PCV_ERR_INVALID = -1,     /* bad argument / unsupported configuration (never silently wrong) */
    PCV_ERR_HIP = -2,         /* a HIP runtime call failed; text in pcv_last_error */
    PCV_ERR_NO_DEVICE = -3,   /* no gfx950 device visible */
    PCV_ERR_TOO_LARGE = -4    /* a tensor exceeds the 2 GiB addressing window of one launch; split the batch */
} pcv_status;

typedef enum pcv_dtype { PCV_F32 = 0, PCV_BF16 = 1, PCV_F16 = 2 } pcv_dtype;

/* activ.py:50-81,117-132 (relu, relu6, sigmoid) and activ.py:16-47 (swish, hsigmoid, hswish) */
typedef enum pcv_act {
    PCV_ACT_NONE = 0, PCV_ACT_RELU = 1, PCV_ACT_RELU6 = 2, PCV_ACT_SIGMOID = 3,
    PCV_ACT_SWISH = 4, PCV_ACT_HSIGMOID = 5, PCV_ACT_HSWISH = 6
} pcv_act;

/*
 * One fused ConvBlock launch: y = post_act( act( conv(x) * scale + shift ) + residual ).
 * Replaces Conv2d + BatchNorm2d(eval) + activation of ConvBlock.forward (pytorchcv/models/common/conv.py:278-286,
 * incl. the explicit ZeroPad2d of conv.py:245-249 through the four pad fields) and, when has_residual, the
 * `x + identity` + activation that follows the block in the unit (resnet.py:227-228, resnext.py:114-115,
 * mobilenetv2.py:69-70).
 */
typedef struct pcv_conv_desc {
    int32_t struct_size;           /* sizeof(pcv_conv_desc) as the CALLER's binding sees it (= pcv_conv_desc_size() of the library it
                                      was written against); every entry point refuses a descriptor whose size differs */
    int32_t N, H, W;               /* input batch and logical spatial size */
    int32_t Cin, Cout;             /* logical channels (all groups) */
    int32_t kh, kw;
    int32_t stride_h, stride_w;
    int32_t pad_t, pad_l, pad_b, pad_r;
    int32_t dil_h, dil_w;
    int32_t groups;
    int32_t act;                   /* pcv_act after scale/shift */
    int32_t post_act;              /* pcv_act after the residual add */
    int32_t has_residual;
    int32_t dtype;                 /* pcv_dtype of x, packed weights, residual */
    int32_t out_dtype;             /* pcv_dtype of y: dtype, or PCV_F32 (classifier logits) */
    int32_t x_cpitch;              /* channel pitch of x in elements: Cin, or 4 / round-up-8 for a padded stem input */
    int32_t x_wpitch;              /* row pitch of x in pixels: W, or W rounded up to even for a padded stem input */
    int32_t y_cpitch;              /* channel pitch of y in elements: 0 or Cout = dense; wider when `y` points at a channel
                                      slice of a concatenation buffer (torch.cat((identity, x), dim=1), densenet.py:58) -
                                      honoured by pcv_conv2d_fused only, every other entry point requires a dense y */
} pcv_conv_desc;

/* ---- context ------------------------------------------------------------------------------------------ */
int pcv_abi_version(void);
size_t pcv_conv_desc_size(void);                    /* sizeof(pcv_conv_desc) in this build: a binding asserts it once at load time */
int pcv_create(pcv_ctx** out, int device);
int pcv_destroy(pcv_ctx* ctx);
const char* pcv_last_error(const pcv_ctx* ctx);     /* ctx may be NULL: returns the last creation error */
/* Tuning/debug switches ("persist", "tile", "d3x3", "max_blocks", ...; some also settable as PCV_AMD_* environment variables
 * before pcv_create). They select among kernels / grid sizes that compute the same result; nothing in the reference
 * corresponds to them. "max_blocks" (test only) caps every persistent grid so that small inputs walk several tiles per block. */
int pcv_set_tuning(pcv_ctx* ctx, const char* key, int value);

/* ---- fp16 range guard ----------------------------------------------------------------------------------- */
/* With dtype PCV_F16 every kernel that rounds a result to fp16 also checks its magnitude: fp16 turns |v| >= 65520 into infinity,
 * which the next ReLU6 / sigmoid would clamp back into range (a wrong but plausible result; bf16 and fp32 storage share fp32's
 * exponent range and need no check). A context counts such roundings in a device word; these three calls read it IN STREAM ORDER:
 *   pcv_fp16_guard_begin  slot (device, 4 bytes) <- the counter, before the first kernel of a forward;
 *   pcv_fp16_guard_end    after the last one: if the counter moved since `slot`, the `count` fp32 values at y (the logits) are
 *                         overwritten with NaN - no host synchronisation, captured into a hipGraph like any launch. A forward that
 *                         runs concurrently with an overflowing one on the same context is poisoned too (conservative);
 *   pcv_fp16_overflow_count  host read of the counter (synchronises `stream`): diagnostics, tests, bench.py's post-run check.
 * Nothing in the reference corresponds to this (it computes in fp32: `net(x)`, resnet.py:333-337); it is what makes fp16 a safe
 * default storage type for the depthwise families, where bf16 misses the 1e-2 bound. */
int pcv_fp16_guard_begin(pcv_ctx* ctx, unsigned* slot, void* stream);
int pcv_fp16_guard_end(pcv_ctx* ctx, const unsigned* slot, float* y, long count, void* stream);
int pcv_fp16_overflow_count(pcv_ctx* ctx, unsigned* count, void* stream);

/* ---- layout: the only NCHW-facing calls ---------------------------------------------------------------- */
/* x: fp32 NCHW [N,C,H,W] (what callers hand to `net(x)`, resnet.py:333) -> y: NHWC [N,H,wpitch,cpitch] in dtype,
 * pad channels / pad columns zero-filled. */
int pcv_nchw_to_nhwc(pcv_ctx* ctx, const float* x, void* y, int N, int C, int H, int W,
                     int cpitch, int wpitch, int dtype, void* stream);
/* x: NHWC [N,H,W,cpitch] in dtype (cpitch >= C: channels padded to a multiple of 8; 0 = C) -> y: fp32 NCHW [N,C,H,W]
 * (block-level drop-in use; not on the whole-net hot path). */
int pcv_nhwc_to_nchw(pcv_ctx* ctx, const void* x, float* y, int N, int C, int H, int W, int cpitch, int dtype,
                     void* stream);

/* Decoded uint8 frames [N,Hs,Ws,C] (C <= 4) -> the network's input: crop [top, top+H) x [left, left+W), (u8/255 - mean[c]) *
 * inv_std[c] (the "ordinary normalization" every pretrained model expects, README.md:12-13), NHWC [N,H,wpitch,4] in dtype with
 * zero pad channels/columns - exactly what pcv_nchw_to_nhwc would produce from the fp32 NCHW tensor of the host pipeline. */
int pcv_preprocess_u8(pcv_ctx* ctx, const unsigned char* x, void* y, int N, int Hs, int Ws, int C, int top, int left,
                      int H, int W, int wpitch, const float* mean, const float* inv_std, int dtype, void* stream);

/* ---- weights (load time) ------------------------------------------------------------------------------- */
/* Size of the packed-weight blob of a dense or grouped conv (groups < Cin). */
int pcv_conv_packed_bytes(const pcv_conv_desc* d, size_t* bytes);
/* w: fp32 OIHW [Cout, Cin/groups, kh, kw] exactly as `conv.weight` in the reference state_dict (conv.py:250-258)
 * -> packed: K-major, MFMA-row-ordered blob in d->dtype (layout in DESIGN.md). N/H/W of d are ignored. For the 16-bit layers that the
 * image-resident kernels can take (dense 3x3 / s1 / p1 with 256 or 512 input channels, 1x1 / s1 with 1024 or 2048, Cout % 64 == 0) the
 * blob continues with the same weights in MFMA-fragment load order (csrc/d3i_conv.hpp, csrc/d1i_conv.hpp): about twice the bytes for
 * those layers - always size the buffer with pcv_conv_packed_bytes. */
int pcv_conv_pack(pcv_ctx* ctx, const pcv_conv_desc* d, const float* w, void* packed, void* stream);
/* Depthwise (groups == Cin == Cout, conv.py:437-473): w fp32 [C,1,kh,kw] -> packed [kh*kw][C] in dtype. For 16-bit 3x3 filters the
 * blob continues (from the next 16-byte boundary) with the same taps as the compressed diagonal fragments of the sparse matrix
 * instruction, 6 KB per 32 channels - what pcv_mbconv_fused's register-resident kernel reads (csrc/mbr.hpp); always size the buffer
 * with pcv_dwconv_packed_bytes. The depthwise kernels themselves read only the first part. */
int pcv_dwconv_packed_bytes(const pcv_conv_desc* d, size_t* bytes);
int pcv_dwconv_pack(pcv_ctx* ctx, const pcv_conv_desc* d, const float* w, void* packed, void* stream);
/* Eval-mode BatchNorm2d (common/norm.py:34-50) folded to fp32 scale/shift; any of gamma..var NULL means "no BN"
 * (scale 1, shift 0); conv_bias (conv.bias, may be NULL) is folded in: shift += bias * scale. */
int pcv_bn_fold(pcv_ctx* ctx, int C, const float* gamma, const float* beta, const float* mean, const float* var,
                float eps, const float* conv_bias, float* scale, float* shift, void* stream);

/* ---- hot path ------------------------------------------------------------------------------------------ */
/* Dense / grouped implicit-GEMM convolution on MFMA, fused epilogue. x NHWC [N,H,x_wpitch,x_cpitch];
 * residual (or NULL) and y NHWC [N,Ho,Wo,Cout]. */
int pcv_conv2d_fused(pcv_ctx* ctx, const pcv_conv_desc* d, const void* x, const void* packed,
                     const float* scale, const float* shift, const void* residual, void* y, void* stream);

/* The stem convolution (Cin <= 4, stride 2, Cout <= 64, 16 bit) and the MaxPool2d(3, stride 2, pad 1) that follows it in the
 * ResNet-style init blocks (reference resnet.py:250-258: `conv` then `pool`) as ONE launch: y is the POOLED map
 * [N, Hq, Wq, Cout] (Hq = (Ho + 2 - 3) / 2 + 1), the full-resolution convolution output is never written. Results are
 * bit-identical to pcv_conv2d_fused followed by pcv_maxpool2d. `pcv_conv2d_maxpool_supported` tells whether a descriptor and
 * pooling geometry are covered (the caller otherwise issues the two launches). */
/* pcv_conv2d_fused with a per-image channel gate: y = post_act(act(scale * conv + shift) * gate[n, c] + residual), gate fp32
 * [N][Cout]. This is how an SE block behind a LINEAR convolution runs in one pass (seresnet.py:60-71: body.conv3 has no
 * activation, so mean_hw(BN(conv3(z))) = BN(conv3(mean_hw(z))): the squeeze is taken on the 4x narrower input z, the excitation
 * runs before the convolution, and the scale + skip add + ReLU ride in its epilogue - the SE block's own two passes over the
 * wide tensor disappear). `pcv_fc_f32` is the small fp32 dense layer out[N,J] = act(b + in[N,K] . w[J,K]^T) that maps the
 * squeezed input through the (BN-folded) convolution. */
int pcv_conv2d_gated_fused(pcv_ctx* ctx, const pcv_conv_desc* d, const void* x, const void* packed, const float* scale,
                           const float* shift, const float* gate, const void* residual, void* y, void* stream);
int pcv_fc_f32(pcv_ctx* ctx, const float* in, const float* w, const float* b, float* out, int N, int K, int J, int act,
               void* stream);
int pcv_conv2d_maxpool_supported(const pcv_conv_desc* d, int k, int s, int p, int ceil_mode);
int pcv_conv2d_maxpool_fused(pcv_ctx* ctx, const pcv_conv_desc* d, const void* x, const void* packed, const float* scale,
                             const float* shift, void* y, int k, int s, int p, int ceil_mode, void* stream);

/* The stem convolution (+ BN + activation, + the MaxPool2d(3, 2, 1) behind it when `pool`) reading the caller's fp32 NCHW image
 * itself: what `ResNet.forward` hands to `features.init_block` (reference resnet.py:333-334, mobilenetv2.py:152-153) without the
 * pcv_nchw_to_nhwc pass in front. `d` describes the convolution exactly as for pcv_conv2d_fused on the padded NHWC4 view
 * (Cin <= 3, x_cpitch 4, even x_wpitch); x_nchw is [N, Cin, H, W] fp32 with W a multiple of 4. Results are bit-identical to
 * pcv_nchw_to_nhwc + pcv_conv2d_fused / pcv_conv2d_maxpool_fused. */
int pcv_conv2d_nchw_stem_supported(const pcv_conv_desc* d, int pool);
int pcv_conv2d_nchw_stem_fused(pcv_ctx* ctx, const pcv_conv_desc* d, const float* x_nchw, const void* packed, const float* scale,
                               const float* shift, void* y, int pool, void* stream);
/* Depthwise direct convolution, fused epilogue (same contract). */
int pcv_dwconv2d_fused(pcv_ctx* ctx, const pcv_conv_desc* d, const void* x, const void* packed,
                       const float* scale, const float* shift, const void* residual, void* y, void* stream);
/* nn.MaxPool2d(k, s, p, ceil_mode) of ResInitBlock (resnet.py:255-258, floor) and ShuffleInitBlock (shufflenetv2.py:111-115,
 * ceil_mode=True): -inf padding; with ceil_mode the last window may hang over the bottom / right edge (PyTorch's rule). */
int pcv_maxpool2d(pcv_ctx* ctx, const void* x, void* y, int N, int H, int W, int C, int k, int s, int p,
                  int ceil_mode, int dtype, void* stream);
/* torch.chunk / channel slicing at ANY channel offset (shufflenetv2.py:80): y[rows, y_cpitch] gets x[.., offset .. offset+C),
 * pad channels zero. */
int pcv_channel_slice(pcv_ctx* ctx, const void* x, void* y, long rows, int C, int offset, int x_cpitch, int y_cpitch,
                      int dtype, void* stream);
/* torch.cat((a, b), dim=1) + ChannelShuffle(groups=2) (shufflenetv2.py:89-90, common/tutti.py:267-291) in one pass:
 * y[.., 2i] = a[.., i], y[.., 2i+1] = b[.., i], i < Ch. */
int pcv_channel_interleave2(pcv_ctx* ctx, const void* a, const void* b, void* y, long rows, int Ch, int a_cpitch,
                            int b_cpitch, int y_cpitch, int dtype, void* stream);
/* torch.cat(..., dim=1) of `Concurrent` / `SequentialConcurrent` (common/arch.py:58-131) in its copy form: y[rows, y_offset + c] =
 * x[rows, c], c < C; channel counts, offset and pitches are multiples of 8. (A branch that ends in a convolution writes its
 * slice itself: pcv_conv2d_fused with y pointing at the slice and d->y_cpitch set - no copy at all.) */
int pcv_channel_concat(pcv_ctx* ctx, const void* x, void* y, long rows, int C, int x_cpitch, int y_cpitch, int y_offset,
                       int dtype, void* stream);
/* F.interpolate of `InterpolationBlock` (common/tutti.py:194-264): x NHWC [N,H,W,C] -> y NHWC [N,Ho,Wo,C]; bilinear != 0:
 * mode "bilinear" with the given align_corners, else mode "nearest". fp32 arithmetic, ATen's source-coordinate rules. */
int pcv_interpolate(pcv_ctx* ctx, const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, int bilinear,
                    int align_corners, int dtype, void* stream);
/* nn.AvgPool2d(k, stride=s), no padding (resnet.py:316-318, mobilenetv2.py:134-136); k == H == W is the
 * global-average-pool of the classifier tail. fp32 accumulation. */
int pcv_avgpool2d(pcv_ctx* ctx, const void* x, void* y, int N, int H, int W, int C, int k, int s,
                  int dtype, int out_dtype, void* stream);
/* nn.AdaptiveAvgPool2d(output_size=1) `final_pool` (efficientnet.py:339): mean over the HW positions of any map shape. */
int pcv_global_avgpool(pcv_ctx* ctx, const void* x, void* y, int N, int HW, int C, int dtype, int out_dtype,
                       void* stream);
/* nn.Linear (resnet.py:320-322,336) / bias-free 1x1 classifier conv (mobilenetv2.py:138-141,154):
 * y[N,Cout] (out_dtype) = x[N,Cin] (dtype) . W^T + bias. `packed` comes from pcv_conv_pack with kh=kw=1. */
int pcv_gemm_bias(pcv_ctx* ctx, const void* x, const void* packed, const float* bias, void* y,
                  int N, int Cin, int Cout, int dtype, int out_dtype, void* stream);
/* SEBlock (common/att.py:94-105): squeeze = AdaptiveAvgPool2d(1) -> mean fp32 [N,C]. */
int pcv_se_squeeze(pcv_ctx* ctx, const void* x, float* mean, int N, int HW, int C, int dtype, void* stream);
/* mid[N,M] = mid_act(W1 . mean + b1); gate[N,C] = out_act(W2 . mid + b2); W1 fp32 [M,C], W2 fp32 [C,M]
 * (att.py:76-92). `mid` is caller-provided fp32 storage for the hidden activations (two launches). */
int pcv_se_excite(pcv_ctx* ctx, const float* mean, const float* w1, const float* b1, const float* w2,
                  const float* b2, float* mid, float* gate, int N, int C, int M, int mid_act, int out_act,
                  void* stream);
/* y = post_act(x * gate[n,c] + residual) (att.py:104 + seresnet.py:69-71); residual may be NULL. */
int pcv_se_scale(pcv_ctx* ctx, const void* x, const float* gate, const void* residual, void* y,
                 int N, int HW, int C, int post_act, int dtype, void* stream);

/* Two chained 1x1 ConvBlocks in one launch: y1 = post_act1(act1(conv1(x)*s1+b1) + residual) - the last convolution of a
 * bottleneck unit with its skip add (resnet.py:227-228) - and y2 = act2(conv2(y1)*s2+b2), the first convolution of the NEXT
 * unit (resnet.py:106-109), without re-reading y1 from HBM. `pcv_conv1x1_pair_supported` tells whether a pair of descriptors is
 * covered (currently 64 -> 256 -> 64 channels, stride 1, 16-bit); unsupported pairs are run as two pcv_conv2d_fused calls. */
int pcv_conv1x1_pair_supported(const pcv_conv_desc* d1, const pcv_conv_desc* d2);
int pcv_conv1x1_pair_fused(pcv_ctx* ctx, const pcv_conv_desc* d1, const pcv_conv_desc* d2, const void* x,
                           const void* packed1, const float* scale1, const float* shift1, const void* residual, void* y1,
                           const void* packed2, const float* scale2, const float* shift2, void* y2, void* stream);

/* The pair with pcv_conv2d_gated_fused's per-image channel gate on the first convolution (an SE block run inside it:
 * y1 = post_act(act(BN(conv1(x))) * gate[n, c] + residual)); same shapes as pcv_conv1x1_pair_fused. */
int pcv_conv1x1_pair_gated_supported(const pcv_conv_desc* d1, const pcv_conv_desc* d2);
int pcv_conv1x1_pair_gated_fused(pcv_ctx* ctx, const pcv_conv_desc* d1, const pcv_conv_desc* d2, const void* x,
                                 const void* packed1, const float* scale1, const float* shift1, const float* gate,
                                 const void* residual, void* y1, const void* packed2, const float* scale2, const float* shift2,
                                 void* y2, void* stream);

/* The same pair when the unit's skip tensor is itself a 1x1 convolution + BN of the unit's input x0 (first unit of a stage,
 * `identity_conv`, resnet.py:214-216,225-226): the kernel recomputes the skip tile from x0 instead of reading it, so the
 * identity convolution's own launch and both passes over its 4x wider output disappear. `d_id` describes the identity
 * convolution (x0 -> skip, no activation, stride 1), d1/d2 as above with d1->has_residual = 1. Covered today: 64 -> 256 pairs. */
int pcv_conv1x1_pair_idconv_supported(const pcv_conv_desc* d_id, const pcv_conv_desc* d1, const pcv_conv_desc* d2);
int pcv_conv1x1_pair_idconv_fused(pcv_ctx* ctx, const pcv_conv_desc* d_id, const pcv_conv_desc* d1, const pcv_conv_desc* d2,
                                  const void* x0, const void* packed_id, const float* scale_id, const float* shift_id,
                                  const void* x, const void* packed1, const float* scale1, const float* shift1, void* y1,
                                  const void* packed2, const float* scale2, const float* shift2, void* y2, void* stream);

/* A whole inverted-residual unit in one launch: [1x1 expand ConvBlock] -> depthwise 3x3 ConvBlock -> 1x1 project ConvBlock
 * (+ skip add): LinearBottleneck.forward (mobilenetv2.py:62-71), MobileNetV3Unit without SE (mobilenetv3.py:82-93),
 * DwsConvBlock (conv.py:612-615; d_exp == NULL). The expanded tensor stays on the CU. The descriptors and packed blobs are
 * those of the three separate calls (pcv_conv_pack / pcv_dwconv_pack); `residual` belongs to d_proj (has_residual, post_act).
 * `pcv_mbconv_supported` tells whether a triple is covered AND pays (16-bit, depthwise 3x3 pad 1 stride 1/2, <= 96 input and
 * <= 32 output channels, output maps at least 24 wide - the large early maps where the expanded tensor dominates);
 * anything else runs as the separate calls. */
int pcv_mbconv_supported(const pcv_conv_desc* d_exp, const pcv_conv_desc* d_dw, const pcv_conv_desc* d_proj);
int pcv_mbconv_fused(pcv_ctx* ctx, const pcv_conv_desc* d_exp, const pcv_conv_desc* d_dw, const pcv_conv_desc* d_proj,
                     const void* x, const void* packed_exp, const float* scale_e, const float* shift_e,
                     const void* packed_dw, const float* scale_d, const float* shift_d, const void* packed_proj,
                     const float* scale_p, const float* shift_p, const void* residual, void* y, void* stream);

/* y[rows,C] = act(x * scale[c] + shift[c]): the BatchNorm2d + activation a PreConvBlock applies BEFORE its convolution
 * (conv.py:776-779) and PreResActivation (preresnet.py:199-222), for the places where it cannot ride in the producing
 * convolution's epilogue (the unit input, which the skip path needs un-activated). rows = N*H*W; x_cpitch (0 = C) is the
 * channel pitch of x when its C channels are the leading slice of a wider concatenation buffer (densenet.py:55-59);
 * y is dense. */
int pcv_bn_act(pcv_ctx* ctx, const void* x, const float* scale, const float* shift, void* y, long rows, int C,
               int x_cpitch, int act, int dtype, void* stream);

/* ---- multi-GPU: RCCL helpers for a C caller -------------------------------------------------------------------------------------
 * The reference has no distributed code; BASELINE's north star shards the batch over the 8 GPUs of a node - one
 * process per GPU, the packed weights broadcast once over xGMI, the logits gathered per batch, no collective between layers. The
 * Python package does this through torch.distributed (pytorchcv_amd/parallel.py); these three entry points are the same two
 * collectives for a caller that only has this ABI. `comm` is the caller's communicator (an ncclComm_t from ncclCommInitRank, passed
 * as void*); the RCCL entry points are resolved at run time from the librccl the process already has (PyTorch's or the caller's; else
 * dlopen("librccl.so.1")) - the library has no link-time dependency on RCCL. All calls are asynchronous on `stream`. */
int pcv_rccl_available(void);                       /* 1 when ncclBroadcast / ncclAllGather / ncclGroupStart / ncclGroupEnd resolve */
/* bufs[i] (device, bytes[i] long, same sizes on every rank) becomes rank `root`'s: ONE grouped RCCL launch for all `count` buffers -
 * xGMI is point-to-point, a ring broadcast is per-link bound, so few large messages (pcv_conv_pack's blobs, the folded scale/shift). */
int pcv_rccl_broadcast(pcv_ctx* ctx, void* comm, void* const* bufs, const size_t* bytes, int count, int root, void* stream);
/* recv[r * bytes_per_rank ...] = rank r's `send`: the fp32 logits of every rank's image shard, in rank order (4 KB per image). */
int pcv_rccl_allgather(pcv_ctx* ctx, void* comm, const void* send, void* recv, size_t bytes_per_rank, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PCV_AMD_H */
