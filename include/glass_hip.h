/*
 * glass_hip.h — C ABI of libglass_hip.so, the MI355X (gfx950) kernel library behind the
 * GLASS inference hot path.
 *
 * The reference (amazon-science/glass-text-spotting) has no FFI of its own: its hot path
 * reaches native code through PyTorch / detectron2==0.6 operators.  Each entry point below
 * names the reference call site (file:line relative to the reference root) whose native
 * work it replaces.  INTEGRATION.md shows the ctypes binding a reference maintainer adds.
 *
 * Conventions
 *  - plain C, no torch types: device pointers are raw `float*`/`int*` (e.g. from
 *    `torch.Tensor.data_ptr()`), sizes are ints, `stream` is a `hipStream_t` cast to void*.
 *  - activations are fp32 NHWC; a "pixel stride" (ld*) is the distance in floats between
 *    consecutive pixels, so a tensor may be a channel slice of a wider one.
 *  - every function returns 0 on success, a negative GLASS_E* code on failure;
 *    glass_last_error() returns a thread-local message.  Nothing throws across the ABI.
 *  - all work is enqueued on the caller's stream; no function synchronises unless stated.
 */
#ifndef GLASS_HIP_H
#define GLASS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLASS_OK 0
#define GLASS_EINVAL (-1)   /* bad argument */
#define GLASS_EHIP (-2)     /* HIP runtime error (message has hipGetErrorString) */
#define GLASS_ENOMEM (-3)   /* workspace too small */

typedef void* glass_stream_t;

const char* glass_last_error(void);
int glass_abi_version(void);
/* number of visible HIP devices (0 if none / runtime unusable) */
int glass_device_count(void);

/* ------------------------------------------------------------------ dense contraction
 * Implicit-GEMM convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32), NHWC, with the
 * epilogue y = act(conv(x,w) + bias [+ residual]).
 * Replaces the cuDNN/MKL-DNN conv + BatchNorm(eval, folded into w/bias at load) + ReLU +
 * residual-add sequences of: d2 ResNet/FPN/RPN head (glass/modeling/meta_arch/
 * glass_rcnn.py:83,87), P2P3Fusion (glass/modeling/fusion/fusion_modules.py:281-286),
 * the local extractor (glass/modeling/fusion/local_feature_extraction.py:153-188,308-323),
 * the fusion out-conv (fusion_modules.py:156), CNN_V1_1 (glass/modeling/recognition/
 * recognizer_backbone.py:77-81), and every nn.Linear on the path (H=W=KH=KW=1).
 * w layout: [Cout][KH][KW][Cin] (K-contiguous per output channel), Cin % 4 == 0.       */
typedef struct glass_conv_desc {
  int N, H, W, Cin;              /* input: N images of H x W pixels, Cin channels used */
  int Cout, KH, KW;
  int stride_h, stride_w, pad_h, pad_w;
  int Ho, Wo;                    /* output spatial size */
  int ldx;                       /* input pixel stride (floats), >= Cin */
  int ldy, y_coff, y_cstride;    /* output pixel stride, first channel, channel step */
  int relu;                      /* 0 none, 1 ReLU after everything, 2 ReLU before the residual add */
  int res_mode;                  /* 0 none, 1 same-shape residual, 2 nearest-x2-upsampled residual
                                    (residual tensor is [N, Ho/2, Wo/2, *]) */
  int ldr;                       /* residual pixel stride */
} glass_conv_desc;

int glass_conv2d_nhwc(const glass_conv_desc* d, const float* x, const float* w, const float* bias,
                      const float* residual, float* y, glass_stream_t stream);

/* Same operator with the operands rounded to fp16 (round to nearest even) as they are staged and multiplied on the
 * fp16 matrix cores with fp32 accumulation (v_mfma_f32_32x32x16_f16, 16x the fp32 matrix rate); x, w, bias, residual
 * and y stay fp32 in memory.  An opt-in precision mode for BASELINE.json configs[4] ("fp16"): results equal
 * conv(fp16(x), fp16(w)) accumulated in fp32, NOT the fp32 reference path's 1e-3 bar.  Needs tensors < 2 GiB.   */
int glass_conv2d_nhwc_f16(const glass_conv_desc* d, const float* x, const float* w, const float* bias,
                          const float* residual, float* y, glass_stream_t stream);

/* fp16 STORAGE form of glass_conv2d_nhwc_f16 (BASELINE configs[4]: "fp16"): the same kernel, but the activation tensors
 * themselves may live in HBM as IEEE fp16 NHWC - flags bit 0: x is fp16, bit 1: y is written as fp16 (round to nearest
 * even from the fp32 accumulator after bias / ReLU / residual), bit 2: the residual is fp16.  Weights and bias stay fp32
 * in memory (weights are rounded to fp16 as they are staged, like the operands of the _f16 entry); accumulation is fp32.
 * ldx / ldy / ldr / y_coff are in ELEMENTS of the respective tensor.  The op equals, bit for bit up to fp32 summation
 * order, fp16?(act(conv(fp16(x), fp16(w)) + bias [+ residual])) - the arithmetic oracle/glass_cpu.py emulates for the
 * fp16-storage parity tests.                                                                                         */
int glass_conv2d_nhwc_h16(const glass_conv_desc* d, const void* x, const float* w, const float* bias, const void* residual,
                          void* y, int flags, glass_stream_t stream);

/* The fp16-storage convolution built for the fp16 matrix cores (csrc/conv_h16.hip): same operator, descriptor, flags and
 * arithmetic as glass_conv2d_nhwc_h16 - fp16?(act(conv(fp16(x), fp16(w)) + bias [+ residual])), fp32 accumulation on
 * v_mfma_f32_16x16x32_f16 - for the layers whose INPUT is an fp16 tensor (flags bit 0 set) with Cin % 64 == 0 and
 * Cout % 64 == 0: any kernel size up to 32 taps, any stride / zero padding, same-size or x2-upsampled residual,
 * channel-offset output.  `u_packed` replaces `w`: glass_conv_h16_pack_weights rounds W [Cout][KH][KW][Cin] to fp16
 * (round to nearest even - the rounding glass_conv2d_nhwc_h16 applies while staging) and lays it out in MFMA fragment
 * order (Cout*KH*KW*Cin halves) once per layer.  Results equal glass_conv2d_nhwc_h16 up to fp32 summation order.
 * glass_conv_h16_supported(d, flags) == 0 (fp32 input, other channel counts, rows not 16-byte aligned, operands
 * >= 2 GiB) -> callers use glass_conv2d_nhwc_h16.                                                                  */
int glass_conv_h16_supported(const glass_conv_desc* d, int flags);
size_t glass_conv_h16_weight_halves(int Cout, int KH, int KW, int Cin);
int glass_conv_h16_pack_weights(const float* w, int Cout, int KH, int KW, int Cin, void* u_packed, glass_stream_t stream);
int glass_conv2d_nhwc_h16_packed(const glass_conv_desc* d, const void* x, const void* u_packed, const float* bias,
                                 const void* residual, void* y, int flags, glass_stream_t stream);

/* Winograd F(2x2,3x3) form of the same operator for the 3x3 / stride 1 / pad 1 layers (FPN output convs,
 * RPN head conv, every 3x3 of the ResNet trunk and of the local extractor's BasicBlocks, fusion output
 * conv): 2.25x fewer fp32 MFMA multiplies, results equal to glass_conv2d_nhwc to fp32 rounding
 * (|diff| <~ 1e-5 of the output scale; tests/test_gpu_f_ops.py).  Same descriptor, epilogue semantics and
 * error behaviour as glass_conv2d_nhwc; `u_packed` replaces `w`:
 *   glass_winograd_supported(d)            1 if the descriptor can take this path (3x3 s1 p1, Cin % 16 == 0,
 *                                          Cout % 64 == 0, y_cstride == 1, ldy/y_coff % 4 == 0, res_mode 0/1,
 *                                          input, output and residual spans < 2 GiB each - the kernels use 32-bit
 *                                          bounds-checked buffer addressing), else 0 - callers fall back to
 *                                          glass_conv2d_nhwc.
 *   glass_winograd_weight_floats(Cout,Cin) floats in the packed buffer (16 * Cout * Cin).
 *   glass_winograd_pack_weights            w [Cout][3][3][Cin] (BN-folded) -> U = G w G^t in the kernel's
 *                                          MFMA fragment order; run once per layer at checkpoint load.
 *   glass_winograd_block_channels(Cout,Cin) output channels per workgroup of the kernel this layer gets: 128
 *                                          (conv3x3_wino128_f32: 32 tiles x 128 channels, when Cout % 128 == 0 and
 *                                          Cin % 32 == 0) or 64 (conv3x3_wino_f32: 64 tiles x 64 channels).  The
 *                                          packed layout follows the same rule; informational for callers
 *                                          (profiling / launch-count heuristics).                              */
int glass_winograd_supported(const glass_conv_desc* d);
int glass_winograd_block_channels(int Cout, int Cin);
size_t glass_winograd_weight_floats(int Cout, int Cin);
int glass_winograd_pack_weights(const float* w, int Cout, int Cin, float* u_packed, glass_stream_t stream);
int glass_conv3x3_winograd_nhwc(const glass_conv_desc* d, const float* x, const float* u_packed, const float* bias,
                                const float* residual, float* y, glass_stream_t stream);

/* Winograd F(4x4,3x3) form (csrc/winograd43.hip): 36 multiplies per 4x4 output tile = 1.78x fewer than F(2x2,3x3),
 * 4x fewer than the direct convolution, for the layers with Cout % 128 == 0 and Cin % 32 == 0 (the 128/256/512
 * channel 3x3 layers: FPN outputs, RPN head, trunk, local extractor layer2..4, mask head).  Transform points
 * (0, 1, -1, 1/2, -2, inf): results equal glass_conv2d_nhwc to <= 2e-5 of the output range (fp64 reference;
 * tests/test_gpu_f_ops.py), still two orders of magnitude inside the path's 1e-3 bar.  Same descriptor, epilogue
 * semantics and error behaviour as glass_conv3x3_winograd_nhwc; its own packed weight layout (36 * Cout * Cin
 * floats).  glass_winograd43_supported additionally wants input, output and residual spans < 1 GiB each (split
 * 32-bit offsets) - callers fall back to the F(2x2) entry.                                                        */
int glass_winograd43_supported(const glass_conv_desc* d);
size_t glass_winograd43_weight_floats(int Cout, int Cin);
int glass_winograd43_pack_weights(const float* w, int Cout, int Cin, float* u_packed, glass_stream_t stream);
int glass_conv3x3_winograd43_nhwc(const glass_conv_desc* d, const float* x, const float* u_packed, const float* bias,
                                  const float* residual, float* y, glass_stream_t stream);
/* the same kernel restricted to the FULL tile columns: writes output columns [0, 4 * (W / 4)) only (W >= 4).  For maps of
 * width 4 k + 1 - the local extractor's 16 x 33 maps (reference glass/modeling/fusion/local_feature_extraction.py:123,
 * MaxPool2d(2, (2, 1), (0, 1)); 17 launches per step) - the ragged tile column would cost a full column of tiles for one
 * pixel column; the caller computes that column with glass_conv2d_nhwc on the last two input columns (KH 3, KW 1 over
 * channels = (kw, cin), see glass_amd/ops/native.py).  Same descriptor (the TRUE H, W), epilogue and errors.          */
int glass_conv3x3_winograd43_body_nhwc(const glass_conv_desc* d, const float* x, const float* u_packed, const float* bias,
                                       const float* residual, float* y, glass_stream_t stream);
/* Split-K form of the F(4x4,3x3) entry (ABI 7; wide shape only: Cout % 128 == 0, Cin % 32 == 0): `splits` (2..32, dividing Cin / 32)
 * k-slices of the layer run as independent workgroups, each writes the raw partial output of its slice to
 * workspace[slice][N H W][Cout] (>= glass_winograd43_splitk_workspace_bytes, 16-byte aligned), and an ordered reduction adds the
 * slices (deterministic), applies bias / ReLU / residual and writes y with the descriptor's strides.  For the 3x3 layers whose
 * 16-tile x 128-channel grid leaves most of the chip idle when ONE image is in flight - the reference predictor's batch,
 * glass/inference/glass_runner.py:93-96: res4 / res5 3x3 layers, the fusion conv, FPN / RPN on the small levels.  `body_only` != 0:
 * full tile columns only, as glass_conv3x3_winograd43_body_nhwc (the caller's strip convolution must follow: the reduction
 * writes the last pixel column from unwritten workspace).                                                                 */
int glass_winograd43_splitk_supported(const glass_conv_desc* d, int splits);
int64_t glass_winograd43_splitk_workspace_bytes(const glass_conv_desc* d, int splits);
int glass_conv3x3_winograd43_splitk_nhwc(const glass_conv_desc* d, const float* x, const float* u_packed, const float* bias,
                                         const float* residual, float* y, int splits, int body_only, void* workspace,
                                         int64_t workspace_bytes, glass_stream_t stream);
/* ... and the F(2x2,3x3) kernel restricted to its full tile columns: output columns [0, 2 * (W / 2)) (W >= 2), packed
 * weights of glass_winograd_pack_weights.  With one image in flight the 16 x 33 maps of 32 RoIs are 272 workgroups on the full
 * grid (two rounds on 256 CUs) and exactly 256 on the body grid; the last column is the same strip convolution.           */
int glass_conv3x3_winograd_body_nhwc(const glass_conv_desc* d, const float* x, const float* u_packed, const float* bias,
                                     const float* residual, float* y, glass_stream_t stream);

/* 1x1 convolution as a weight-streaming GEMM (csrc/pointwise.hip): the bottleneck / lateral / shortcut 1x1 layers with
 * Cin % 32 == 0 and Cout % 128 == 0, any square stride, pad 0.  Same descriptor, epilogue semantics (bias, ReLU before /
 * after the residual, same-size or x2-upsampled residual, channel-offset output) and fp32 arithmetic as
 * glass_conv2d_nhwc - an exact-fp32 fma chain, results equal to it up to summation order; `u_packed` replaces `w`:
 * glass_pointwise_pack_weights lays W [Cout][Cin] out in MFMA fragment order (Cout * Cin floats) once per layer.
 * glass_pointwise_supported(d) == 0 -> callers use glass_conv2d_nhwc.                                               */
int glass_pointwise_supported(const glass_conv_desc* d);
size_t glass_pointwise_weight_floats(int Cout, int Cin);
int glass_pointwise_pack_weights(const float* w, int Cout, int Cin, float* u_packed, glass_stream_t stream);
int glass_conv1x1_pointwise_nhwc(const glass_conv_desc* d, const float* x, const float* u_packed, const float* bias,
                                 const float* residual, float* y, glass_stream_t stream);

/* The same 1x1 convolution with its fp32 products computed EXACTLY on the bf16 matrix cores (csrc/pointwise_split.hip):
 * every fp32 operand is the exact sum of three bf16 pieces (8 + 8 + 8 significant bits), the product of two pieces is exact
 * in fp32, and the nine piece products accumulate in fp32 - the same sum of exact products an fp32 fma chain accumulates,
 * in 9 x 16 instead of 8 x 32 matrix cycles per 32 input channels.  Same descriptor / epilogue as glass_conv1x1_pointwise_nhwc,
 * results equal to glass_conv2d_nhwc up to summation order.  `u_packed`: glass_pointwise_split_pack_weights lays W [Cout][Cin]
 * out as three bf16 planes in MFMA fragment order (glass_pointwise_split_weight_bytes = 6 Cout Cin bytes), once per layer.
 * `products`: 9 (the exact product; the model path) or 6 (without the three piece pairs that together stay below 2^-23 relative: measurement
 * only).  Inputs must be finite (inf - inf in the split).                                                               */
int glass_pointwise_split_supported(const glass_conv_desc* d);
size_t glass_pointwise_split_weight_bytes(int Cout, int Cin);
int glass_pointwise_split_pack_weights(const float* w, int Cout, int Cin, void* u_packed, glass_stream_t stream);
int glass_conv1x1_pointwise_split_nhwc(const glass_conv_desc* d, const float* x, const void* u_packed, const float* bias,
                                       const float* residual, float* y, int products, glass_stream_t stream);
/* A bottleneck block's shortcut folded into its conv3 (detectron2 BottleneckBlock.forward behind reference
 * glass/modeling/meta_arch/glass_rcnn.py:83: `out = conv3(out); out += shortcut(x); relu` [d2-recall]) - ONE launch, one
 * accumulator:  y = act([x1 strided | x2] . [W1 | W2]^T + bias).  `d` describes source 1 and the output (x1 [N,H,W,ldx], Cin,
 * square stride, relu; res_mode must be 0); x2 [N,Ho,Wo,ldx2] carries Cin2 channels on the output grid; `u_packed` is
 * glass_pointwise_split_pack_weights(Cout, Cin + Cin2) of the two weights concatenated along Cin (W1 first), `bias` the sum of the
 * two folded biases.  Nine exact piece products per element as above; the [N,Ho,Wo,Cout] shortcut map never reaches HBM.   */
int glass_pointwise_split_dual_supported(const glass_conv_desc* d, int Cin2, int ldx2);
int glass_conv1x1_pointwise_split_dual_nhwc(const glass_conv_desc* d, const float* x1, const float* x2, int Cin2, int ldx2,
                                            const void* u_packed, const float* bias, float* y, glass_stream_t stream);

/* Fused head of the local-crop feature extractor (reference glass/modeling/fusion/local_feature_extraction.py:103-112):
 * conv0_1 (3x3, 3->16) + BN + ReLU, conv0_2 (3x3, 16->32) + BN + ReLU, maxpool1 2x2 in ONE kernel - the two intermediate
 * maps stay in LDS.  x [R,H,W,4] NHWC4 crops, w1 [16][3][3][4] / b1 [16], w2 [32][3][3][16] / b2 [32] (BatchNorm folded),
 * y [R,H/2,W/2,32]; fp32.  Same results as the three separate entries up to fp32 summation order
 * (tests/test_gpu_f_ops.py).  glass_local_stem_supported: H and W positive multiples of 32 (128 x 128 crops in every
 * reference config) - otherwise callers use the separate entries.                                                   */
int glass_local_stem_supported(int H, int W);
int glass_local_stem_fused(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* y,
                           int R, int H, int W, glass_stream_t stream);
/* fp16-storage form (conv precision "fp16s"): fp32 crops and weights in, operands rounded to fp16 (round to nearest even),
 * fp32 accumulation, the conv0_1 map rounded to fp16 where the separate entries store it, y [R,H/2,W/2,32] written as
 * fp16 - the results of glass_conv2d_nhwc_h16 x 2 + glass_maxpool2d_nhwc_h16 up to fp32 summation order.            */
int glass_local_stem_fused_h16(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, void* y,
                               int R, int H, int W, glass_stream_t stream);

/* Split-K form of glass_conv2d_nhwc for FEW output pixels and a LONG K - what one image per call (the reference predictor,
 * glass/inference/glass_runner.py:93-96) leaves of the box head (reference recognizers_hybrid_head.py:320-322 -> d2
 * FastRCNNConvFCHead [d2-recall]: 100 x 12544 -> 2048), of the res4 / res5 3x3 layers and of the 11-row box predictors:
 * same descriptor, operands, epilogue (bias, relu 0/1/2, res_mode 0/1, strided output) and results up to fp32 summation
 * order; the k-tiles are cut into `splits` slices that run as independent workgroups of the implicit-GEMM kernel (partial
 * sums in `workspace`, >= glass_conv2d_splitk_workspace_bytes), added in slice order by a second kernel - deterministic.
 * glass_conv2d_splitk_supported: Cin % 32 == 0, (KH KW Cin / 32) % splits == 0, 2 <= splits <= 32, res_mode 0/1, fp32.   */
int glass_conv2d_splitk_supported(const glass_conv_desc* d, int splits);
int64_t glass_conv2d_splitk_workspace_bytes(const glass_conv_desc* d, int splits);
int glass_conv2d_nhwc_splitk(const glass_conv_desc* d, const float* x, const float* w, const float* bias, const float* residual,
                             float* y, int splits, void* workspace, int64_t workspace_bytes, glass_stream_t stream);

/* Fused ResNet stem (detectron2 BasicStem, [d2-recall], as restated in oracle/glass_cpu.py resnet50_fpn):
 * conv 7x7 stride 2 pad 3 (3 -> 64, BatchNorm folded into w / bias) + ReLU + max_pool2d(3, 2, 1) in ONE kernel
 * (csrc/backbone_stem.hip) - the [N,H/2,W/2,64] map between them stays on the CU.  x [N,H,W,4] NHWC4 (4th channel ignored),
 * w [64][7][7][4], bias [64], y [N,H/4,W/4,64]; fp32.  The results of glass_conv2d_nhwc + glass_maxpool2d_nhwc up to fp32
 * summation order.  glass_backbone_stem_supported: H and W positive multiples of 4 - otherwise callers use the two entries. */
int glass_backbone_stem_supported(int H, int W);
int glass_backbone_stem_fused(const float* x, const float* w, const float* bias, float* y, int N, int H, int W,
                              glass_stream_t stream);

/* max pooling NHWC (d2 stem max_pool2d k3 s2 p1; local extractor maxpool1..3,
 * glass/modeling/fusion/local_feature_extraction.py:112,118,124). Padding acts as -inf. */
int glass_maxpool2d_nhwc(const float* x, float* y, int N, int H, int W, int C, int KH, int KW, int sh, int sw,
                         int ph, int pw, int Ho, int Wo, glass_stream_t stream);
/* the same on fp16 tensors (fp16 storage mode): exact - the maximum of fp16 values is one of them */
int glass_maxpool2d_nhwc_h16(const void* x, void* y, int N, int H, int W, int C, int KH, int KW, int sh, int sw,
                             int ph, int pw, int Ho, int Wo, glass_stream_t stream);

/* ------------------------------------------------------------------ image preprocess
 * (x - mean) / std per channel, CHW float [3,H,W] -> NHWC4 slot `n` of a zero-padded
 * batch [N,Hp,Wp,4] (4th channel 0).  d2 GeneralizedRCNN.preprocess_image +
 * ImageList.from_tensors at glass/modeling/meta_arch/glass_rcnn.py:82.                   */
int glass_preprocess_image(const float* chw, int H, int W, const float* mean3, const float* std3, float* batch_nhwc4,
                           int n, int Hp, int Wp, glass_stream_t stream);
/* uint8 HWC (3 channels) -> float CHW with optional bilinear resize (align_corners=False):
 * GlassRunner._image_to_tensor, glass/inference/glass_runner.py:123-148.                 */
int glass_image_u8hwc_to_chw_resized(const uint8_t* hwc, int H, int W, float* chw, int Ho, int Wo, int flip_channels,
                                     glass_stream_t stream);

/* ------------------------------------------------------------------ rotated RoIAlign
 * detectron2 ROIPooler(ROIAlignRotated) over up to 5 pyramid levels
 * (glass/modeling/fusion/recognizers_hybrid_head.py:320 box pooler 7x7 sr2 p2..p6;
 *  :550 recognizer pooler 8x32 sr0 on the P2P3 map; :556 image pooler 128x128 sr2).
 * boxes [R,5] = (cx,cy,w,h,angle_deg), batch_idx [R]; level l has tensor feat[l] of
 * [N,Hl,Wl,*] with pixel stride ld[l] and spatial scale scale[l]; C channels pooled.
 * out is [R,PH,PW,*] with pixel stride ldy, first channel y_coff, channel step y_cstride. */
typedef struct glass_roialign_desc {
  int num_levels;
  const float* feat[5];
  int H[5], W[5], ld[5];
  float scale[5];
  int min_level;                 /* log2(1/scale[0]); level map of d2 assign_boxes_to_levels */
  int C, PH, PW, sampling_ratio;
  int ldy, y_coff, y_cstride;
} glass_roialign_desc;

int glass_roi_align_rotated(const glass_roialign_desc* d, const float* boxes, const int* batch_idx, int R, float* out,
                            glass_stream_t stream);
/* the same with fp16 pyramid levels (d->feat[] point to IEEE fp16 NHWC tensors, d->ld[] in elements): taps are widened to
 * fp32, interpolation and output are fp32 as above */
int glass_roi_align_rotated_h16(const glass_roialign_desc* d, const float* boxes, const int* batch_idx, int R,
                                float* out, glass_stream_t stream);

/* the same on nearest_up2(level): d->feat[] are HALF-resolution fp32 tensors [N,H/2,W/2,*], d->H / d->W / d->scale describe the
 * x2-upsampled map (even H, W): sampling grid, clamps and bilinear weights are those of the materialised upsampled map, tap
 * (y, x) reads (y >> 1, x >> 1).  Lets the recognizer pooler (glass/modeling/fusion/recognizers_hybrid_head.py:550) read
 * P2P3Fusion's second operand without the upsampled map existing (glass/modeling/fusion/fusion_modules.py:281-286).        */
int glass_roi_align_rotated_up2(const glass_roialign_desc* d, const float* boxes, const int* batch_idx, int R, float* out,
                                glass_stream_t stream);

/* ------------------------------------------------------------------ rotated-box proposals
 * RRPN proposal selection for a whole batch and all pyramid levels (d2 RRPN.predict_proposals +
 * find_top_rrpn_proposals, reached from glass/modeling/meta_arch/glass_rcnn.py:87): per image and
 * level, the `topk` highest objectness logits in descending order (ties -> lower flat index
 * (h,w,a), i.e. a stable descending sort) are selected, their anchors generated analytically,
 * deltas applied (Box2BoxTransformRotated with `weights5_host`), and written to
 * out_boxes[n][slot_off + i] / out_scores / out_level (= level index).
 * logits: [N,H,W,A] with pixel stride ldl; deltas: [N,H,W,A*5] with pixel stride ldd (both may be
 * channel slices of one head tensor); cell_anchors: [A,5] on device ((.,.,w,h,angle) per anchor).
 * topk <= 2048, num_levels <= 8.  `workspace` is device scratch of >= glass_rpn_workspace_bytes().  */
typedef struct glass_rpn_level {
  const float* logits;
  const float* deltas;
  const float* cell_anchors;
  int ldl, ldd, H, W, stride, topk, slot_off;
} glass_rpn_level;
int64_t glass_rpn_workspace_bytes(int N, int num_levels);
int glass_rpn_topk_decode(const glass_rpn_level* levels, int num_levels, int N, int A, float anchor_offset,
                          const float* weights5_host, int slots_per_image, float* out_boxes, float* out_scores,
                          int* out_level, void* workspace, int64_t workspace_bytes, glass_stream_t stream);

/* Per image: drop non-finite rows, drop rows with score <= score_thresh, clip (|angle| <= 1
 * deg only; flag GLASS_NMS_CLIP), drop empty boxes (flag GLASS_NMS_DROP_EMPTY), sort by
 * descending score (stable), class/level-offset greedy rotated NMS (iou >= thr suppresses:
 * CPU semantics of d2 nms_rotated), keep the first `post_topk`.
 * Used for RPN (thr 0.7, post 100, CLIP|DROP_EMPTY; d2 find_top_rrpn_proposals) and for the
 * box head (glass/modeling/roi_heads/rotated_fast_rcnn.py:88-148: CLIP, score > 0.05,
 * thr 0.35, top 100).  boxes [N,S,5], scores [N,S], cat [N,S] (level / class id; NULL = 0),
 * valid_count [N] device ints (slots used per image; NULL = S), image_hw [N,2] device ints
 * (h,w).  S <= 8192.  Outputs: out_boxes [N,post_topk,5] (clipped), out_scores
 * [N,post_topk], out_index [N,post_topk] (slot index into the input), out_count [N].     */
#define GLASS_NMS_CLIP 1
#define GLASS_NMS_DROP_EMPTY 2
int glass_rotated_nms_select(const float* boxes, const float* scores, const int* cat, const int* valid_count, int N, int S,
                             const int* image_hw, float score_thresh, float nms_thresh, int post_topk, int flags,
                             float* out_boxes, float* out_scores, int* out_index, int* out_count, glass_stream_t stream);

/* Box-head prediction decode (glass/modeling/roi_heads/rotated_fast_rcnn.py:335-342,
 * 480-491): softmax over (K+1=2) class logits, apply_deltas with weights, orientation
 * softmax -> (argmax, prob).  One class (K=1).  Outputs per proposal row.                */
int glass_box_decode(const float* cls_logits, const float* deltas, const float* orient_logits, const float* proposals,
                     int R, const float* weights5_host, float* out_boxes, float* out_fg_prob, float* out_orient2,
                     glass_stream_t stream);

/* Batched GlassRCNN._postprocess (glass/modeling/meta_arch/glass_rcnn.py:103-128) for all N images of a
 * step: filter_small_boxes (min(w,h) >= min_box_dim, if do_filter_small; post_processor_rotated_boxes.py:
 * 89-94) then detector_postprocess (post_processor_academic.py:118-178): RotatedBoxes.scale(sx,sy),
 * clip to the output size, drop empty boxes [d2-recall for scale/clip] — with ordered compaction of every
 * per-detection field.  boxes [N,K,5], scores [N,K], orient [N,K,2] or NULL, counts [N] device ints
 * (slots used), text [sum R, TC] with image n's rows starting at roi_start[n] (NULL: no text),
 * scale_xy [N,2] device floats (sx,sy), out_hw [N,2] device ints (h,w).  Outputs are padded [N,K,...]
 * tensors with out_count[n] valid leading rows.  K <= 1024.                                      */
int glass_detections_finalize(const float* boxes, const float* scores, const float* orient, const float* text,
                              const int* counts, const int* roi_start, const float* scale_xy, const int* out_hw, int N,
                              int K, int TC, float min_box_dim, int do_filter_small, float* out_boxes, float* out_scores,
                              float* out_orient, float* out_text, int* out_count, glass_stream_t stream);

/* Word post-processing for all N images of a step in one launch (one workgroup per image): the
 * reference's PostProcessorAcademic = PostProcessorRotatedBoxes.__call__ (glass/postprocess/
 * post_processor_rotated_boxes.py:66-87: small-box filter, score >= VALID_CONFIDENCE, iterative pair merge
 * with min-area-rect + NMS 0.99, score >= DETECT_THRESHOLD, polygons) followed by the text decode and
 * text-score filter (post_processor_academic.py:26-35, text_evaluator.py:323-348, text_encoder.py:81-151).
 * boxes [N,K,5], scores [N,K], counts [N] (device); text_arg / text_max [N,K,T] = argmax character and its probability
 * per decoding step (glass_text_argmax below) or both NULL (then no text filter); scale_xy [N,2] device floats or NULL
 * (GlassRunner's un-scaling, glass_runner.py:100-101; a (1,1) entry is skipped exactly like the reference's
 * `if scale_ratio != 1`).
 * thresholds8_host = {MIN_BOX_DIMENSION, VALID_CONFIDENCE, DETECT_THRESHOLD, MERGE_IOA_THRESH,
 * PAIRS_HEIGHT_RATIO_THRESH, MAX_ANGLE_DIFF, minimal_ioa (0.01), TEXT_THRESHOLD}; stop_index = index of '[s]'.
 * Outputs (padded to K, out_count[n] valid rows, in the reference's output order): boxes, scores,
 * polygons [N,K,4,2], out_src [N,K] source slot of each survivor (to gather other fields), out_char
 * [N,K,T] argmax character index per step, out_text_score [N,K], out_text_len [N,K] (characters before the
 * stop symbol).  K <= 128, T <= 64.                                                               */
int glass_postprocess_words(const float* boxes, const float* scores, const int* counts, const int* text_arg,
                            const float* text_max, const float* scale_xy, int N, int K, int T,
                            const float* thresholds8_host, int stop_index, float* out_boxes, float* out_scores,
                            float* out_polygons, int* out_src, int* out_char, float* out_text_score, int* out_text_len,
                            int* out_count, glass_stream_t stream);

/* The decode's argmax (reference glass/modeling/recognition/text_encoder.py:81-151 `preds_prob.max(dim=2)`, consumed by
 * text_evaluator.py:323-348): text [N,K,T,C] probability rows -> out_arg [N,K,T] first index of the row maximum,
 * out_max [N,K,T] the maximum; rows of boxes k >= counts[n] are skipped (outputs untouched).  One wavefront per row.  */
int glass_text_argmax(const float* text, const int* counts, int N, int K, int T, int C, int* out_arg, float* out_max,
                      glass_stream_t stream);
/* The fixed-size per-image WORD record the ranks exchange with one all_gather per step (glass_amd/distributed.py; replaces
 * the reference's pickled comm.gather of per-image results, glass/evaluation/text_evaluator.py:246-249), packed in one launch
 * from the padded outputs of glass_postprocess_words: records [N, 1 + max_det (16 + steps)] float32 =
 * [count | boxes 5D | score D | text score D | polygon 8D | text length D | character index D x steps]; rows beyond
 * min(count, K, max_det) and steps beyond min(Tw, steps) are zero.  chars [N,K,Tw], polygons [N,K,4,2].                    */
int glass_pack_word_records(const float* boxes, const float* scores, const float* text_score, const float* polygons,
                            const int* text_len, const int* chars, const int* count, int N, int K, int Tw, int max_det, int steps,
                            float* records, glass_stream_t stream);

/* pairwise rotated IoU matrix out[n1][n2] (d2 pairwise_iou_rotated; glass/structures/boxes.py:33,
 * used by the post-processor's pairwise_ioa_rotated).                                     */
int glass_pairwise_iou_rotated(const float* boxes1, int n1, const float* boxes2, int n2, float* out,
                               glass_stream_t stream);

/* ------------------------------------------------------------------ fusion attention
 * MultiAspectGCAttention minus its out-conv (glass/modeling/fusion/fusion_modules.py:
 * 91-154): x is the channel-INTERLEAVED cat(local,global) [R,HW,C] (C=512, channel 2i =
 * local i, 2i+1 = global i, i.e. already x[:, order]); per head softmax(conv_mask) pooling,
 * channel_add MLP (conv1x1 -> LayerNorm -> ReLU -> conv1x1) and the broadcast add are
 * applied IN PLACE on x.  w_mask [C/heads], b_mask [1], w1 [P][C], b1 [P], ln_g/ln_b [P],
 * w2 [C][P], b2 [C].  Requires C == 512, heads == 8, P == 256, HW == 256 (the GLASS shapes). */
int glass_gc_attention_inplace(float* x, int R, int HW, int C, int heads, int P, const float* w_mask, const float* b_mask,
                               const float* w1, const float* b1, const float* ln_g, const float* ln_b, const float* w2,
                               const float* b2, glass_stream_t stream);

/* mean over H of an NHWC map: [R,H,W,C] -> [R,W,C]
 * (BiLSTMBlockV2.forward, glass/modeling/recognition/recognizer_encoder.py:119).          */
int glass_mean_over_h(const float* x, float* y, int R, int H, int W, int C, glass_stream_t stream);

/* "k-blocked" weight packing used by the recurrent kernels so that a thread owning output row j
 * streams 16-byte pieces that are contiguous across neighbouring threads:
 *   packed[k/4][j][k%4] = W[j][k]   for W [rows][K], K % 4 == 0.
 * The host packs once at checkpoint load (any tensor library can do it:
 *   W.view(rows, K/4, 4).permute(1, 0, 2).contiguous()).                                   */

/* a[i] *= b[i] (the gate of the `SimpleAttention` fusion variant, glass/modeling/fusion/fusion_modules.py:181-186) */
int glass_mul_inplace(float* a, const float* b, int64_t n, glass_stream_t stream);

/* y[i] = fp16(x[i]), round to nearest even: the operand rounding of the fp16 conv modes applied once to an fp32 activation
 * tensor so that it can feed glass_conv2d_nhwc_h16_packed (same results as glass_conv2d_nhwc_h16 rounding it on the fly) */
int glass_cast_f32_to_f16(const float* x, void* y, int64_t n, glass_stream_t stream);

/* ------------------------------------------------------------------ rotated mask branch (inference)
 * MaskRotatedRecognizerHybridHead._forward_mask + d2 MaskRCNNConvUpsampleHead + mask_rcnn_inference
 * (glass/modeling/fusion/recognizers_hybrid_head.py:378-442,595-606; rotated_mask_head.py:409-442): the pooler
 * is glass_roi_align_rotated (14x14 over p2..p6), the four 3x3 convs and the 1x1 predictor are
 * glass_conv2d_nhwc / glass_conv3x3_winograd_nhwc, ConvTranspose2d(k=2,s=2)+ReLU is a 1x1 conv to 4*C channels
 * (weight rows ordered (a*2+b)*C + c for output offset (a,b)) followed by glass_pixel_shuffle2x_nhwc:
 *   y[n, 2h+a, 2w+b, c] = x[n, h, w, (a*2+b)*C + c],  x [N,H,W,4C] -> y [N,2H,2W,C].                      */
int glass_pixel_shuffle2x_nhwc(const float* x, float* y, int N, int H, int W, int C, glass_stream_t stream);
int glass_sigmoid_inplace(float* x, int64_t n, glass_stream_t stream);
/* paste_masks_in_image / _do_paste_mask for rotated boxes (glass/postprocess/post_processor_academic.py:187-335,
 * called from detector_postprocess :167-173): masks [R,M,M] (probabilities), boxes [R,5] (cx,cy,w,h,angle deg,
 * already scaled to the output resolution) -> out uint8 [R,H,W]; threshold >= 0: 1 where the bilinearly sampled
 * mask >= threshold else 0 (torch.bool layout); threshold < 0: (value * 255) truncated (the reference's
 * visualisation mode).                                                                                      */
int glass_paste_rotated_masks(const float* masks, const float* boxes, int R, int M, int H, int W, float threshold,
                              uint8_t* out, glass_stream_t stream);

/* ------------------------------------------------------------------ recurrent encoder
 * One bidirectional LSTM layer's recurrence (nn.LSTM gate order i,f,g,o; zero initial
 * state; glass/modeling/recognition/recognizer_encoder.py:123-144).  The input projection
 * xg = x @ W_ih^T + b_ih + b_hh for both directions is computed beforehand with
 * glass_conv2d_nhwc: xg [R,T,2,4*Hd] (direction-major, then gate-major i,f,g,o).
 * w_hh [2][4*Hd][Hd] (nn.LSTM layout, forward then reverse direction).  out [R,T,2*Hd] (fwd | bwd).
 * `workspace`: device scratch (hidden/cell state) of >= glass_bilstm_workspace_bytes().  Requires Hd == 256.
 * Implementation note: T step launches are issued by this one call (each step is spread over the chip).                                                                    */
int64_t glass_bilstm_workspace_bytes(int R, int Hd);
int glass_bilstm_recurrence(const float* xg, const float* w_hh, float* out, int R, int T, int Hd, void* workspace,
                            int64_t workspace_bytes, glass_stream_t stream);
/* The same recurrence as ONE launch per layer (csrc/recurrent_persistent.hip; same reference lines, same arguments, outputs
 * bit-identical to glass_bilstm_recurrence): W_hh resident in registers, h_t handed between the 8 workgroups of a
 * (16-RoI group, direction) chain inside the launch as 8-byte {step tag, value} device-scope granules.
 * dirs_per_workgroup / groups_per_workgroup: how many independent chains a workgroup interleaves - (2,1) (0,0 = default:
 * both directions of one RoI group, 8 workgroups per 16 RoIs), (2,2) or (1,1).  `workspace` (16-byte aligned, >=
 * glass_bilstm_persistent_workspace_bytes()) is zeroed by the call.  Every in-kernel wait is bounded; a wait that gave up
 * (the outputs of that call are then undefined) raises bit 0 (bit 1: glass_attention_decode_persistent) of
 *   - `call_status`, a caller-owned device int of THIS call (may be null; the caller zeroes it and reads it back with whatever
 *     it reads back next on the stream - the product path does, and re-runs the call on glass_bilstm_recurrence when it is set:
 *     glass_amd/modeling/fusion/recognizers_hybrid_head.py), and
 *   - the device's sticky status word, read (after a device synchronise) with glass_recurrence_status - a diagnostic.       */
int64_t glass_bilstm_persistent_workspace_bytes(int R, int Hd);
int glass_bilstm_recurrence_persistent(const float* xg, const float* w_hh, float* out, int R, int T, int Hd,
                                       int dirs_per_workgroup, int groups_per_workgroup, int* call_status, void* workspace,
                                       int64_t workspace_bytes, glass_stream_t stream);
int glass_recurrence_status(int* status_out, int reset);
/* Test hook for the bounded waits of the two persistent recurrent kernels (process-wide, applies to launches issued after the
 * call): `spin_limit` sweeps before a wavefront gives up (<= 0: the built-in bound, ~2-4 s), `withhold_ticket` >= 0: the
 * workgroup that draws that start ticket never publishes its slice, so its peers' waits give up (-1: off).  The product never
 * calls it; tests/test_gpu_g_persistent_rnn.py uses it to force the fall-back of the product path.                        */
int glass_recurrence_test_hook(int64_t spin_limit, int withhold_ticket);

/* ------------------------------------------------------------------ attention decoder
 * Greedy additive-attention GRU decoder (AttentionRecognitionHead.sample,
 * glass/modeling/recognition/prediction_aster.py:63-99,247-266,291-302), all `max_len`
 * steps on device with no host sync; the reference's batch-global early break (rows of
 * steps after every RoI OF THE SAME IMAGE has emitted `eos` stay zero) is applied as a
 * mask using roi_image [R] (image id per RoI in [0,num_images), non-decreasing).
 * x [R,T,D]; xproj [R,T,D] = xEmbed(x) precomputed with glass_conv2d_nhwc (it is
 * step-invariant).  Weights: sW [D/4][D][4] and fcW [D/4][C][4] in the k-blocked packing above; sB [D],
 * wW [D], wB [1], emb [C][D], w_ih [3D][2D] and w_hh [3D][D] row-major (nn.GRU layout; input =
 * [embedding | context]), b_ih [3D], b_hh [3D], fcB [C], temperature (host).
 * out [R,max_len,C] softmax probabilities; pred_scratch [R,max_len] ints; `workspace`: device scratch of
 * >= glass_decode_workspace_bytes().  Requires D == 256, T <= 64, C <= 256.  Two step launches per
 * decoding step are issued by this one call.                                                */
typedef struct glass_decoder_weights {
  const float *sW, *sB, *wW, *wB, *emb, *w_ih, *w_hh, *b_ih, *b_hh, *fcW, *fcB;
  float temperature;
} glass_decoder_weights;
int64_t glass_decode_workspace_bytes(int R, int D);
int glass_attention_decode(const float* x, const float* xproj, const glass_decoder_weights* w, const int* roi_image, int R,
                           int num_images, int T, int D, int C, int max_len, int eos, float* out, int* pred_scratch,
                           void* workspace, int64_t workspace_bytes, glass_stream_t stream);

/* The same decoder as ONE launch for all max_len steps (csrc/recurrent_persistent.hip; same reference lines and outputs up to
 * fp32 summation order): a group of 16 RoIs is served by 16 workgroups that keep the GRU rows of 16 hidden units each (W_hh
 * and the context half of W_ih as MFMA fragments) and sEmbed in registers, one RoI's x / xproj rows in LDS, and hand h_i and
 * (context_i, arg-max) to each other inside the launch as 8-byte {step tag, value} device-scope granules.  Extra operands,
 * built once at load: sW_rowmajor [D][D] (sEmbed.weight as stored by torch; `w->sW` is not read) and emb_gi [C][3D] =
 * tgt_embedding.weight @ W_ih[:, :D]^T + b_ih (the embedding half of the GRU input needs no arithmetic per step; `w->emb`,
 * `w->b_ih` are not read).  Needs D == 256, T <= 32, C <= 128 (glass_decode_persistent_supported); `workspace` (16-byte
 * aligned, >= glass_decode_persistent_workspace_bytes) is zeroed by the call.  Bounded waits: bit 1 of `call_status` (may be
 * null) and of glass_recurrence_status, as for glass_bilstm_recurrence_persistent.  Two launches (decoder, early-break mask). */
int glass_decode_persistent_supported(int T, int D, int C, int max_len);
int64_t glass_decode_persistent_workspace_bytes(int R);
int glass_attention_decode_persistent(const float* x, const float* xproj, const glass_decoder_weights* w, const float* sW_rowmajor,
                                      const float* emb_gi, const int* roi_image, int R, int num_images, int T, int D, int C,
                                      int max_len, int eos, float* out, int* pred_scratch, int* call_status, void* workspace,
                                      int64_t workspace_bytes, glass_stream_t stream);

/* ONE step of the same decoder for a search driven by the caller - `output, state, alpha = self.decoder(x, state, y_prev)`
 * inside AttentionRecognitionHead.beam_search (glass/modeling/recognition/prediction_aster.py:133-134, DecoderUnit.forward
 * :291-302): additive attention with the state h_in [R,D] -> context, embedding of y_prev [R] (ints), GRU cell -> h_out
 * [R,D] (must not alias h_in), fc(h_out) * temperature -> logits_out [R,C] and their softmax probs_out [R,C].  Same
 * weights, packing and limits as glass_attention_decode; R here is batch x beam width (rows may repeat the same x /
 * xproj sequence: pass the inflated tensors).  `workspace`: >= glass_decode_step_workspace_bytes().  Three launches. */
int64_t glass_decode_step_workspace_bytes(int R, int D);
int glass_attention_decode_step(const float* x, const float* xproj, const glass_decoder_weights* w, int R, int T, int D, int C,
                                const float* h_in, const int* y_prev, float* h_out, float* logits_out, float* probs_out,
                                void* workspace, int64_t workspace_bytes, glass_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GLASS_HIP_H */
