/* hoig_kernels.h -- C ABI of libhoig_hip.so: the hand-written gfx950 (CDNA4)
 * kernels behind the HOGAN generator/discriminator hot path.
 *
 * Boundary conventions (they mirror what the reference's own native layer does,
 * thirdparty/block_extractor/block_extractor_cuda.cc:5-33 and
 * thirdparty/local_attn_reshape/local_attn_reshape_cuda.cc:5-29, wrapped by
 * block_extractor.py:5-54 / local_attn_reshape.py:5-46):
 *   - plain device pointers + sizes, no torch types;
 *   - the CALLER owns and allocates every buffer (outputs, workspaces);
 *   - work is enqueued on the given hipStream_t, never synchronises, no
 *     allocation inside (graph-capturable);
 *   - return 0 on success, a negative HOIG_E* code on a rejected argument or a
 *     failed launch (the reference returns `int 1` and never checks).
 * Differences from the reference ops, by design: activations are NHWC fp32
 * ("channels-last"), conv weights are packed [Cout][kh][kw][Cin] for Conv2d AND
 * ConvTranspose2d (the host keeps the reference's logical NCHW shapes as views
 * of that storage, hoig_amd/nn.py), gradients of weights ACCUMULATE into the
 * caller's (zeroed) flat gradient buffer.
 *
 * Reference call sites each family replaces are cited per function
 * (paths relative to /root/reference/HOIG_HOv3).
 */
#ifndef HOIG_KERNELS_H
#define HOIG_KERNELS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *hoig_stream_t; /* hipStream_t */

enum { HOIG_OK = 0, HOIG_EINVAL = -1, HOIG_ELAUNCH = -2, HOIG_EUNSUPPORTED = -3 };

/* epilogue activations */
enum { HOIG_ACT_NONE = 0, HOIG_ACT_RELU = 1, HOIG_ACT_LRELU = 2, HOIG_ACT_TANH = 3, HOIG_ACT_SIGMOID = 4 };

/* arithmetic of the MFMA contraction (accumulation is always fp32):
 *   F32    v_mfma_f32_32x32x2_f32, exact fp32 products (parity mode)
 *   BF16X3 both operands split hi+lo into 16-bit halves, 3 MFMAs per k-step (hi*hi + lo*hi + hi*lo): forward launches split
 *          on fp16 (v_mfma_f32_32x32x16_f16, products to ~2^-21), backward launches on bf16 (~2^-16)
 *   F16X2  the gathered operand (activations / dy) split hi+lo, the other one (weights; x in the weight gradient) rounded
 *          to ONE 16-bit value: 2 MFMAs per k-step (hi*hi + lo*hi), ~2^-12 (fp16 forward) / ~2^-9 (bf16 backward) per product
 *   BF16   one MFMA per k-step, both operands rounded to 16 bits (fp16 forward, bf16 backward)
 *   F16F6  FORWARD only (hoig_conv2d_fwd_f6): hi*hi on fp16, the two cross terms of the split on block-scaled e2m3 (fp6)
 *          v_mfma_scale_f32_32x32x64_f8f6f4 (one E8M0 scale per 32 channels): BF16X3's three terms at 1.6 instead of 3 MFMA
 *          units per product; shapes outside that kernel run as BF16X3 */
enum { HOIG_PREC_F32 = 0, HOIG_PREC_BF16X3 = 1, HOIG_PREC_BF16 = 2, HOIG_PREC_F16X2 = 3, HOIG_PREC_F16F6 = 4 };

/* One convolution problem.  NHWC activations, weights [Co][R][S][Ci].
 * transposed=0: y = conv2d(x, w, stride, pad)            x:[B,Hi,Wi,Ci] y:[B,Ho,Wo,Co]
 * transposed=1: y = conv_transpose2d(x, w, stride, pad)  (output_padding implied by Ho,Wo)
 * Replaces the cuDNN convolutions reached from generator.py:15-235,
 * spade.py:18-22, discriminator.py:29-49, extract_attn.py:18-20, vgg19.py:56. */
typedef struct hoig_conv_desc {
    int32_t B, Hi, Wi, Ci;
    int32_t Ho, Wo, Co;
    int32_t R, S;
    int32_t stride, pad;
    int32_t transposed;
    int32_t act;       /* forward epilogue: HOIG_ACT_* applied after +bias */
    float slope;       /* LeakyReLU slope */
    int32_t precision; /* HOIG_PREC_* */
} hoig_conv_desc;

int hoig_conv2d_fwd(const hoig_conv_desc *d, const float *x, const float *w, const float *bias /*nullable*/,
                    float *y, hoig_stream_t stream);
/* dx = d(loss)/dx given dy (dy already carries the activation derivative) */
int hoig_conv2d_bwd_data(const hoig_conv_desc *d, const float *dy, const float *w, float *dx, hoig_stream_t stream);
/* dw += ..., dbias += ... (atomic accumulation into caller-zeroed buffers; dbias nullable) */
int hoig_conv2d_bwd_weight(const hoig_conv_desc *d, const float *x, const float *dy, float *dw, float *dbias,
                           hoig_stream_t stream);

/* PRE-SPLIT gradients (round 5).  The two-term backward arithmetic (HOIG_PREC_F16X2) multiplies bf16(dy) and bf16(dy - bf16(dy)) with
 * one bf16 plane of the other operand; the kernels above make that split of every dy tile in every workgroup that loads it.  A "split
 * tensor" holds it once: for an fp32 NHWC tensor [.., C] the same 4 bytes per element laid out per pixel as [hi: C x bf16][lo: C x bf16].
 * hoig_split_planes_bf16 converts (C % 4 == 0); producers of gradients (hoig_inorm_tile_bwd_split ...) write it directly.  The
 * consumers below compute exactly what their fp32-dy counterparts compute (same two MFMA terms, same accumulation order).
 * hoig_conv2d_bwd_weight_split: stride-1 "same" 3x3 layers with Wo % 32 == 0, Ho % 4 == 0, Co % 128 == 0, Ci % 32 == 0 in
 * HOIG_PREC_F16X2 / HOIG_PREC_BF16 (the lo plane is not read then), no bias gradient; HOIG_EUNSUPPORTED otherwise. */
int hoig_split_planes_bf16(const float *x, uint16_t *out, int64_t npix, int C, hoig_stream_t stream);
/* split -> fp32 (hi + lo; out may alias nothing): for a consumer that has no pre-split form */
int hoig_unsplit_planes_bf16(const uint16_t *in, float *out, int64_t npix, int C, hoig_stream_t stream);
/* hoig_inorm_bwd_add_ld / hoig_inorm_bwd_fused_add with dx written as a split tensor (same arguments otherwise; addend stays fp32) */
int hoig_inorm_bwd_add_ld_split(const float *x, const float *mean, const float *rstd, int mode, const float *p0, const float *p1,
                                int ld_p, const float *y, const float *dy, int act, float slope, const float *addend /*nullable*/,
                                uint16_t *dx_split, float *dp0, float *dp1, int B, int HW, int C, void *workspace,
                                hoig_stream_t stream);
int hoig_inorm_bwd_fused_add_split(const float *x, const float *mean, const float *rstd, int mode, const float *p0, const float *p1,
                                   int ld_p, const float *y, const float *dy, int act, float slope, const float *addend /*nullable*/,
                                   uint16_t *dx_split, float *dp0, float *dp1, int B, int HW, int C, hoig_stream_t stream);
/* ---- Winograd F(2x2,3x3) forward for stride-1 "same" 3x3 layers (round 6; hoig_amd/csrc/conv_wino.hip): 16
 *      element-wise products per 2 x 2 outputs instead of 36, on three fp16 terms like the direct forward (HOIG_PREC_BF16X3): U = G (2^8 w) G^T
 *      is made once per weight version by hoig_pack_conv_weight_wino -- u_hi / u_lo: hoig_wino_plane_halfs(Co, Ci) = 16 Co Ci fp16 values
 *      each, in MFMA A-fragment order [position][Co / 16][Ci / 32][64 lanes][8] -- V = B^T d B is made in the kernel from the fp32 input
 *      and split after the transform, the accumulators are fp32, Y = A^T M A, then 2^-8, bias (nullable), activation.  Hi % 16 == 0,
 *      Wi % 16 == 0, Ci % 32 == 0, Co % 64 == 0; HOIG_EUNSUPPORTED otherwise (the caller then uses hoig_conv2d_fwd_packed).  Replaces, for
 *      these layers, the same F.conv2d calls as hoig_conv2d_fwd_packed (models/networks/generator.py:9-32,35-71, spade.py:18-22). ---- */
int64_t hoig_wino_plane_halfs(int Co, int Ci);
int hoig_pack_conv_weight_wino(const float *w /*[Co][3][3][Ci]*/, int Co, int Ci, uint16_t *u_hi, uint16_t *u_lo, hoig_stream_t stream);
int hoig_conv2d_fwd_wino(const hoig_conv_desc *d, const float *x, const uint16_t *u_hi, const uint16_t *u_lo, const float *bias /*nullable*/,
                         float *y, hoig_stream_t stream);
/* ---- INFERENCE: the norm between two convolutions applied by the second one's loader (VERDICT r1-r4 "consumer half of the norm
 *      fusion"; the reference chains generator.py:16-22 (ResidualBlock: conv - IN - ReLU - conv) and :298-309 (decoder level: ConvTranspose
 *      - IN - ReLU - cat - conv)).  hoig_inorm_fold turns the statistics of the RAW tensor (hoig_inorm_stats / _stats_from_sums) and the
 *      affine parameters (nullable) into one FMA per element: scale[b][c] = rstd * gamma, shift[b][c] = beta - mean * scale, rows of
 *      ld_out floats.  hoig_conv2d_fwd_packed_normin is hoig_conv2d_fwd_packed / hoig_conv2d_cat_fwd_packed(_stats) (x2 nullable: the
 *      gathered tensor is [x | x2] with C1 channels in x) for stride-1 "same" 3x3 layers on the 8-row tilings of the 16x16x32 kernel:
 *      every in-image element of the gathered tensor becomes x * in_scale[b][c] + in_shift[b][c] (c over all Ci gathered channels), then
 *      max(., 0) for c >= in_relu_c0, before it is split for the MFMAs; the zero padding stays zero.  No backward: forward-only launches
 *      (torch.no_grad()).  HOIG_EUNSUPPORTED where that kernel cannot run the layer (the caller then normalises in a pass of its own). */
int hoig_inorm_fold(const float *mean, const float *rstd, const float *gamma /*nullable*/, const float *beta /*nullable*/, int B, int C,
                    float *scale, float *shift, int ld_out, hoig_stream_t stream);
int hoig_conv2d_fwd_packed_normin(const hoig_conv_desc *d, const float *x, int C1, const float *x2 /*nullable*/, const uint16_t *w_hi,
                                  const uint16_t *w_lo, const float *bias /*nullable*/, const float *in_scale, const float *in_shift,
                                  int in_relu_c0, float *y, float *stats /*nullable*/, hoig_stream_t stream);
int hoig_conv2d_bwd_weight_split(const hoig_conv_desc *d, const float *x, const uint16_t *dy_split, float *dw, hoig_stream_t stream);
/* dx = data gradient (+ addend when non-null: hoig_conv2d_bwd_data_packed_add) of a stride-1 "same" 3x3 Conv2d from pre-split dy, on the
 * 8-row tilings of the v_mfma_f32_16x16x32 kernel (Hi % 8 == 0, Wi % 32 == 0 and enough tiles: HOIG_EUNSUPPORTED otherwise -- the
 * caller then converts nothing and uses the fp32 entry points). */
int hoig_conv2d_bwd_data_packed_split(const hoig_conv_desc *d, const uint16_t *dy_split, const uint16_t *wt_hi, const uint16_t *wt_lo,
                                      const float *addend /*nullable*/, float *dx, hoig_stream_t stream);

/* GROUPED launches (round 5): two convolutions of ONE descriptor d (d->B images each) over different tensors with different weights --
 * src_model's and tsf_model's layer (generator.py:379-464: same architecture, separate parameters) -- as ONE grid: the tiles of the
 * second problem ride behind the first's, so the launch fills the chip where the two half-size launches each covered half of it.
 * Results are those of the two single launches.  Stride-1 "same" 3x3 layers on the 8-row tilings of conv_halo16.hip / on
 * wgrad_dma.hip; HOIG_EUNSUPPORTED otherwise (the caller launches the two problems one after the other).  The backward forms read
 * pre-split dy ('PRE-SPLIT gradients' above); addend_a / addend_b: both or neither. */
int hoig_conv2d_fwd_packed_pair(const hoig_conv_desc *d, const float *xa, const float *xb, const uint16_t *wa_hi, const uint16_t *wa_lo,
                                const uint16_t *wb_hi, const uint16_t *wb_lo, const float *bias_a /*nullable*/,
                                const float *bias_b /*nullable*/, float *ya, float *yb, hoig_stream_t stream);
int hoig_conv2d_bwd_data_packed_split_pair(const hoig_conv_desc *d, const uint16_t *dys_a, const uint16_t *dys_b, const uint16_t *wta_hi,
                                           const uint16_t *wta_lo, const uint16_t *wtb_hi, const uint16_t *wtb_lo,
                                           const float *addend_a /*nullable*/, const float *addend_b /*nullable*/, float *dxa, float *dxb,
                                           hoig_stream_t stream);
int hoig_conv2d_bwd_weight_split_pair(const hoig_conv_desc *d, const float *xa, const float *xb, const uint16_t *dys_a,
                                      const uint16_t *dys_b, float *dwa, float *dwb, hoig_stream_t stream);

/* Forward of a stride-1 'same' convolution with <= 16 output channels over Ci % 64 == 0 inputs and ONE activation PER OUTPUT
 * CHANNEL (acts: the HOIG_ACT_* code of channel f in bits [4f, 4f+4)): the generator's image / mask heads (generator.py:219-235,
 * 311-315: tanh image, sigmoid masks) evaluated as one convolution over the decoder's last feature map.  7x7 with <= 5
 * outputs: exact fp32 (VALU kernel) in every precision mode; 3x3 with <= 16 outputs: 16-bit MFMA arithmetic of d->precision;
 * HOIG_EUNSUPPORTED otherwise. */
int hoig_conv2d_fwd_heads(const hoig_conv_desc *d, const float *x, const float *w, const float *bias /*nullable*/, float *y,
                          uint64_t acts, hoig_stream_t stream);

/* 16-bit-operand fast path (HOIG_PREC_BF16X3 / HOIG_PREC_F16X2 / HOIG_PREC_BF16): weights are pre-split once per optimiser step into
 * 16-bit planes hi and lo of n rows x K reduction indices: the FORWARD planes (for_dgrad=0) hold fp16(256*w) and
 * fp16(256*w - hi) -- forward launches split their operands on fp16 and scale the accumulator by 1/256 -- the
 * DATA-GRADIENT planes (for_dgrad=1) hold bf16(w) and bf16(w - hi).  for_dgrad=0: n = co, k = (r*S+s)*Ci + ci
 * (forward GEMM); for_dgrad=1: n = ci, k = (r*S+s)*Co + co (data-gradient GEMM).  A plane is stored in 32(n) x 32(k)
 * blocks of 2 KB: element (n, k) at ((n/32)*(K/32) + k/32)*1024 + (n%32)*32 + k%32, so that the weight tile of one
 * k-step is a few fully used contiguous runs.  Co and Ci must be multiples of 32 (else HOIG_EUNSUPPORTED).  The *_packed
 * entry points return HOIG_EUNSUPPORTED for shapes outside the fast path (channels % 32 != 0, <= 32 output channels);
 * use the fp32 entry points then. */
int hoig_pack_conv_weight_bf16(const float *w, int Co, int RS, int Ci, int for_dgrad, uint16_t *hi, uint16_t *lo /*nullable*/,
                               hoig_stream_t stream);
/* The same split for every conv weight of a network's flat parameter buffer in one launch.  segs (device memory) holds
 * nseg rows of 6 int64: {element offset in flat, Co, R*S, Ci, flags (1: forward planes, 2: data-gradient planes), index
 * of the weight's first 32x32 (co, ci) tile}; ntiles = total tile count (the grid).  The planes of a weight are written
 * at the weight's own element offset of hi_f/lo_f (forward) and hi_d/lo_d (dgrad). */
int hoig_pack_conv_weights_bf16_all(const float *flat, const int64_t *segs, int nseg, int64_t ntiles, uint16_t *hi_f,
                                    uint16_t *lo_f, uint16_t *hi_d, uint16_t *lo_d, hoig_stream_t stream);
int hoig_conv2d_fwd_packed(const hoig_conv_desc *d, const float *x, const uint16_t *w_hi, const uint16_t *w_lo,
                           const float *bias /*nullable*/, float *y, hoig_stream_t stream);
int hoig_conv2d_bwd_data_packed(const hoig_conv_desc *d, const float *dy, const uint16_t *wt_hi, const uint16_t *wt_lo,
                                float *dx, hoig_stream_t stream);
/* y = conv(x) AND, from the same epilogue, the statistics of the instance norm that reads y next (generator.py:16-22 conv -> IN ->
 * ReLU): stats[b][0][co] += sum of y over image b, stats[b][1][co] += sum of y*y (fp32 atomics, one per workgroup and
 * channel; stats = the accumulators of an instance-norm workspace, zero on entry), consumed by hoig_inorm_stats_from_sums instead of
 * a pass over y.  HOIG_EUNSUPPORTED where the layer's kernel has no such epilogue (it exists on the 3x3 stride-1 "same" and the 3x3
 * stride-2 halo kernels, Conv2d and ConvTranspose2d): the caller then runs the plain convolution and hoig_inorm_stats. */
int hoig_conv2d_fwd_packed_stats(const hoig_conv_desc *d, const float *x, const uint16_t *w_hi, const uint16_t *w_lo,
                                 const float *bias /*nullable*/, float *y, float *stats, hoig_stream_t stream);
int hoig_conv2d_cat_fwd_packed_stats(const hoig_conv_desc *d, const float *x1, int C1, const float *x2, const uint16_t *w_hi,
                                     const uint16_t *w_lo, const float *bias /*nullable*/, float *y, float *stats,
                                     hoig_stream_t stream);
/* The same for the layers the packed kernels do not take: hoig_conv2d_fwd (unpacked weights) with the sums, HOIG_EUNSUPPORTED unless
 * the layer is a thin-INPUT convolution on the MFMA path (Ci <= 8 (12 for 3x3), Co % 64 == 0, odd square stride-1 "same" kernel,
 * Hi % 4 == 0, Wi % 32 == 0, a 16-bit precision): the generator's 7x7 stems (generator.py:100,262), whose output is the largest
 * tensor an instance norm's statistics pass would re-read. */
int hoig_conv2d_fwd_stats(const hoig_conv_desc *d, const float *x, const float *w, const float *bias /*nullable*/, float *y, float *stats,
                          hoig_stream_t stream);
/* dx = data gradient + addend (addend: the gradient that reaches the same tensor through its OTHER consumer, e.g. the skip path of
 * a residual block, generator.py:29-32 `x + self.main(x)`; torch's autograd engine sums the two in a separate pass).  Returns
 * HOIG_EUNSUPPORTED for layers whose kernel has no such epilogue (everything but stride-1 "same" 1x1/3x3/5x5 and stride-2 3x3 on the halo kernels): the
 * caller then adds separately. */
int hoig_conv2d_bwd_data_packed_add(const hoig_conv_desc *d, const float *dy, const uint16_t *wt_hi, const uint16_t *wt_lo,
                                    const float *addend, float *dx, hoig_stream_t stream);

/* fp16 + block-scaled fp6 forward of 3x3 stride-1 pad-1 convolutions (hoig_amd/csrc/conv_f6.hip).  q_hi / q_lo: fp6 records of
 * the hi / lo fp16 halves of 256*w, hoig_f6_plane_bytes(Co, 9, Ci) bytes each, made by hoig_pack_conv_weight_f6 once per
 * optimiser step: [(tap * Ci/64 + ci/64)][co] records of 56 B = 2 x 24 B of e2m3 elements (32 channels each, element j in bits
 * [6j, 6j+6)) + the two E8M0 scale bytes.  w_hi: the forward hi plane of hoig_pack_conv_weight_bf16.  HOIG_EUNSUPPORTED unless
 * Ci % 64 == 0, Co % 64 == 0, H % 8 == 0, W % 32 == 0 and the launch has enough 8x32-pixel tiles to fill the chip.
 * hoig_conv2d_cat_fwd_f6: the same over [x1 | x2] along channels (C1 % 32 == 0) without the concatenated tensor. */
int64_t hoig_f6_plane_bytes(int Co, int RS, int Ci);
int hoig_pack_conv_weight_f6(const float *w, int Co, int RS, int Ci, uint8_t *q_hi, uint8_t *q_lo, hoig_stream_t stream);
int hoig_conv2d_fwd_f6(const hoig_conv_desc *d, const float *x, const uint16_t *w_hi, const uint8_t *q_hi, const uint8_t *q_lo,
                       const float *bias /*nullable*/, float *y, hoig_stream_t stream);
int hoig_conv2d_cat_fwd_f6(const hoig_conv_desc *d, const float *x1, int C1, const float *x2, const uint16_t *w_hi,
                           const uint8_t *q_hi, const uint8_t *q_lo, const float *bias /*nullable*/, float *y,
                           hoig_stream_t stream);
/* The same kernel with the loader and epilogue options of the three-term path (round 6: eval.py's forward runs on this arithmetic
 * by default, and the inference chain conv - IN - ReLU - conv, generator.py:16-22 / :298-309, keeps its two fusions):
 *   x2 (nullable, C1 % 32 == 0) -- the second tensor of a channel concatenation, as hoig_conv2d_cat_fwd_f6;
 *   in_scale / in_shift (nullable, together; B x Ci floats from hoig_inorm_fold) and in_relu_c0 (% 32 == 0) -- the gathered tensor is
 *     RAW: x * in_scale + in_shift per (image, gathered channel) and ReLU on channels >= in_relu_c0 are applied when the halo is
 *     converted, as hoig_conv2d_fwd_packed_normin does (Ci <= 1024: the image's two rows live in LDS);
 *   stats (nullable) -- the accumulators of an instance-norm workspace, as hoig_conv2d_fwd_packed_stats.
 * Same HOIG_EUNSUPPORTED conditions as hoig_conv2d_fwd_f6. */
int hoig_conv2d_fwd_f6_ex(const hoig_conv_desc *d, const float *x, int C1, const float *x2 /*nullable*/, const uint16_t *w_hi,
                          const uint8_t *q_hi, const uint8_t *q_lo, const float *bias /*nullable*/, const float *in_scale /*nullable*/,
                          const float *in_shift /*nullable*/, int in_relu_c0, float *y, float *stats /*nullable*/, hoig_stream_t stream);
/* all eligible weights of a flat parameter buffer in one launch: rows = int64[nrows][6] = (offset of the weight in `flat`, Co,
 * RS, Ci, byte offset of its records in q_hi / q_lo (a multiple of 4), index of its first task); a task = one (output channel,
 * tap, 32 input channels) half record: Co*RS*Ci/32 per weight */
int hoig_pack_conv_weights_f6_all(const float *flat, const int64_t *rows, int nrows, int64_t ntasks, uint8_t *q_hi, uint8_t *q_lo,
                                  hoig_stream_t stream);
/* launches with fewer workgroups than this run as three fp16 terms (default 192; returns the previous value; n <= 0: query).
 * Parity tests set 1 so that the fp6 kernel is exercised at their small sizes. */
int hoig_set_f6_min_tiles(int n);

/* conv(cat[x1, x2] along channels) without materialising the concatenation (the decoder's skip convolutions,
 * generator.py:305-306): 3x3 stride-1 "same" convolutions on the fast path only -- every other shape returns
 * HOIG_EUNSUPPORTED and the caller concatenates.  d->Ci = C1 + C2; x1 holds the first C1 (multiple of 32) channels.
 * The data gradient writes [dx1 | dx2] (C1 a multiple of 64); the weight gradient accumulates like hoig_conv2d_bwd_weight. */
int hoig_conv2d_cat_fwd_packed(const hoig_conv_desc *d, const float *x1, int C1, const float *x2, const uint16_t *w_hi,
                               const uint16_t *w_lo, const float *bias /*nullable*/, float *y, hoig_stream_t stream);
int hoig_conv2d_cat_bwd_data_packed(const hoig_conv_desc *d, const float *dy, const uint16_t *wt_hi, const uint16_t *wt_lo,
                                    float *dx1, int C1, float *dx2, hoig_stream_t stream);
int hoig_conv2d_cat_bwd_weight(const hoig_conv_desc *d, const float *x1, int C1, const float *x2, const float *dy, float *dw,
                               float *dbias /*nullable*/, hoig_stream_t stream);

/* ---- instance norm (generator.py:16-22,101-120,154-208; spade.py:13; discriminator.py:37,45 via
 *      base_network.py:31): per-(b,c) mean / biased variance over H*W, eps 1e-5, no running stats. ---- */
/* stats: mean[b*C+c], rstd[b*C+c].  workspace: >= hoig_inorm_workspace_bytes(B,HW,C) bytes whose first 2^18 floats (the pool
 * of atomic accumulators, B*2*C of them used) must be ZERO on entry; hoig_inorm_stats / hoig_inorm_bwd* leave them zero again on return, so one
 * zero-initialised workspace per stream serves every call without memset launches. */
int64_t hoig_inorm_workspace_bytes(int B, int HW, int C);
int hoig_inorm_stats(const float *x, int B, int HW, int C, float eps, float *mean, float *rstd, void *workspace,
                     hoig_stream_t stream);
/* y = act( (x-mean)*rstd * scale + shift ) + residual
 *   mode 0: scale=1, shift=0 (param-free)           mode 1: scale=weight[c], shift=bias[c] (affine)
 *   mode 2: scale=1+gamma[b,hw,c], shift=beta[b,hw,c] (SPADE, spade.py:36) */
int hoig_inorm_apply(const float *x, const float *mean, const float *rstd, int mode, const float *p0, const float *p1,
                     int act, float slope, const float *residual /*nullable*/, float *y, int B, int HW, int C,
                     hoig_stream_t stream);
/* same with gamma/beta rows `ld_p` floats apart (mode 2 only): lets gamma and beta live side by side in ONE [.,2C] tensor
 * (the output of the fused gamma|beta convolution), p0 = gb, p1 = gb + C, ld_p = 2C */
/* mean / rstd from sums that a convolution's epilogue left in the workspace's accumulators (hoig_conv2d_fwd_packed_stats; plain
 * sums: var = E[y^2] - E[y]^2 in fp32, adequate for the outputs of convolutions over normalised activations this is used for);
 * leaves the accumulators zero like hoig_inorm_stats */
int hoig_inorm_stats_from_sums(int B, int HW, int C, float eps, float *mean, float *rstd, void *workspace, hoig_stream_t stream);
int hoig_inorm_apply_ld(const float *x, const float *mean, const float *rstd, int mode, const float *p0, const float *p1,
                        int ld_p, int act, float slope, const float *residual /*nullable*/, float *y, int B, int HW, int C,
                        hoig_stream_t stream);
int hoig_inorm_bwd_ld(const float *x, const float *mean, const float *rstd, int mode, const float *p0, const float *p1,
                      int ld_p, const float *y, const float *dy, int act, float slope, float *dx, float *dp0, float *dp1, int B,
                      int HW, int C, void *workspace, hoig_stream_t stream);
/* Single-launch instance norm for maps of at most 1024 pixels with C % 32 == 0 (the 32x32 bottleneck and below): a workgroup
 * keeps a (sample, 32-channel) slab in registers -- statistics (two-pass), normalise / modulate / activate (+ residual) from
 * one read of x; the backward likewise from one read of x and dy.  Same argument meaning as hoig_inorm_stats +
 * hoig_inorm_apply_ld / hoig_inorm_bwd_ld; mode-1 parameter gradients accumulate atomically, no workspace.
 * HOIG_EUNSUPPORTED for other shapes: the caller then uses the streaming kernels above. */
int hoig_inorm_fwd_fused(const float *x, int mode, const float *p0, const float *p1, int ld_p, int act, float slope,
                         const float *residual /*nullable*/, float eps, float *y, float *mean, float *rstd, int B, int HW,
                         int C, hoig_stream_t stream);
int hoig_inorm_bwd_fused(const float *x, const float *mean, const float *rstd, int mode, const float *p0, const float *p1,
                         int ld_p, const float *y /*nullable, see hoig_inorm_bwd_ld*/, const float *dy, int act, float slope,
                         float *dx, float *dp0, float *dp1, int B, int HW, int C, hoig_stream_t stream);
/* ... + addend (nullable; same shape as x): dx = norm backward + addend.  The SPADE residual block reads its input twice -- the
 * first norm and the skip `x + dx` (generator.py:63-71) -- and the skip's gradient is added here instead of by a pass of its own. */
int hoig_inorm_bwd_add_ld(const float *x, const float *mean, const float *rstd, int mode, const float *p0, const float *p1,
                          int ld_p, const float *y, const float *dy, int act, float slope, const float *addend, float *dx,
                          float *dp0, float *dp1, int B, int HW, int C, void *workspace, hoig_stream_t stream);
int hoig_inorm_bwd_fused_add(const float *x, const float *mean, const float *rstd, int mode, const float *p0, const float *p1,
                             int ld_p, const float *y, const float *dy, int act, float slope, const float *addend, float *dx,
                             float *dp0, float *dp1, int B, int HW, int C, hoig_stream_t stream);

/* backward of hoig_inorm_apply(+stats).  dy is d/d(y) ; y is the forward output (for the activation mask; pass
 * the pre-residual activation output, or NULL when act==NONE).  For ReLU / LeakyReLU after a plain (mode 0) or affine
 * (mode 1, p1 = its bias) norm, y may also be NULL: the mask is recomputed from x, one tensor less to read per pass.
 * p1 is only read in that case (nullable otherwise).
 * Outputs: dx; mode 1: dweight[c] += , dbias[c] += ; mode 2: dgamma, dbeta (same shape as x, overwritten).
 * workspace >= hoig_inorm_workspace_bytes. */
int hoig_inorm_bwd(const float *x, const float *mean, const float *rstd, int mode, const float *p0, const float *p1,
                   const float *y, const float *dy, int act, float slope, float *dx, float *dp0, float *dp1, int B, int HW,
                   int C, void *workspace, hoig_stream_t stream);

/* ---- local attention warping: ExtractorAttn (extract_attn.py:23-29) = K1 block extraction of source (with flow)
 *      and target (zero flow) + conv k5/s5 + LeakyReLU(0.01) + conv1x1 + softmax(25) + K3 reshape + weighted 5x5 average.
 *      Composition (host: hoig_amd/ops.py::_LocalAttn), no 25x-sized tensor anywhere: the taps of a pixel share their
 *      bilinear fractions and each corner is border-clamped on its own, so the sampling commutes with the linear map over
 *      the taps -- the target half of the k5/s5 conv is a 5x5 valid convolution Gt of replicate_pad(target, 2), the source
 *      half a BILINEAR READ (at pixel + flow) of the 5x5 valid convolution Gs of replicate_pad(source, 4), both on
 *      hoig_conv2d_*; the weighted average reads the source's 6x6 footprint.  These entry points are the pieces around the
 *      two convolutions.  flow: [B,2,H,W] (ch0 = x, ch1 = y, PIXEL units as K1 reads them,
 *      block_extractor_kernel.cu:62-67); it carries no gradient on the path (it is data: generator.py:481-488). ---- */
/* y[b, py, px] = x[b, clamp(py-pad), clamp(px-pad)]  ([B,H,W,C] -> [B,H+2pad,W+2pad,C]) and its adjoint */
int hoig_replicate_pad_fwd(const float *x, float *y, int B, int H, int W, int C, int pad, hoig_stream_t stream);
int hoig_replicate_pad_bwd(const float *dy, float *dx, int B, int H, int W, int C, int pad, hoig_stream_t stream);
/* ... + addend (nullable, [B,H,W,C]): the gradient that reached x through another consumer (tx = tx + attention(.., target = tx),
 * generator.py:391-392,413-414), added here instead of by a pass of its own */
int hoig_replicate_pad_bwd_add(const float *dy, const float *addend, float *dx, int B, int H, int W, int C, int pad,
                               hoig_stream_t stream);
/* hidden[m] = gt[m] + bilinear(gs; m + flow(m))            gt: [B,H,W,128] (bias included), gs: [B,H+4,W+4,128] = Gs on the
 *                                                          grid [-2,H+1] x [-2,W+1], hidden: [M,128] pre-activation (out)
 * attn[m]   = softmax_25(w2 . leaky_0.01(hidden[m]) + b2)  w2: [25][128]
 * out[m][c] = (1/25) sum_q attn[m][q] * S[m][q][c]         S = K1's border-clamped bilinear samples of `source` [B,H,W,C],
 *                                                          evaluated over the pixel's 6x6 footprint, never stored */
int hoig_attn_pixel_fwd(const float *gt, const float *gs, const float *flow, const float *w2, const float *b2,
                        const float *source, float *hidden, float *attn, float *out, float *kf /*nullable*/, int B, int H,
                        int W, int C, hoig_stream_t stream);
/* kf (training only): [M][36] weights of the pixel's 6x6 source footprint in `out`, k[i][j] = (1/25) sum_ab w_ab attn[i-a][j-b],
 * kept for hoig_attn_src_gather */
/* given dout: dhidden (overwritten; it is also dGt), dw2 / db2 (accumulated); e_ws: M*36 floats of scratch (the dot products
 * of dout[m] with the 36 source cells of pixel m's footprint) */
int hoig_attn_pixel_bwd(const float *hidden, const float *attn, const float *w2, const float *source, const float *flow,
                        const float *dout, float *dhidden, float *dw2, float *db2, float *e_ws, int B, int H, int W, int C,
                        hoig_stream_t stream);
/* The two source-side gradients are transposes of gathers through the pixels' sampling frames P(m) = floor(m + flow(m)).
 * Instead of scattering with atomics (K2, block_extractor_kernel.cu:158-161) the pixels are bucketed by their frame cell
 * (counting sort: `index`, hoig_attn_index_ints(B,H,W) ints of caller memory; depends on the flow only, so one index serves
 * every attention layer of a resolution) and each output cell gathers from the buckets whose footprints reach it. */
int64_t hoig_attn_index_ints(int B, int H, int W);
int hoig_attn_build_index(const float *flow, int32_t *index, int B, int H, int W, hoig_stream_t stream);
/* dsource[cell] = init[cell] + sum_m kf[m][cell's position in m's 6x6 footprint] * dout[m]   (the weighted average's source
 * gradient; every element written; init (nullable = zeros): the gradient the source receives from its other consumers) */
int hoig_attn_src_gather(const int32_t *index, const float *kf, const float *dout, const float *init, float *dsource, int B,
                         int H, int W, int C, hoig_stream_t stream);
/* dgs = bilinear^T(dhidden)  (dgs: [B,H+4,W+4,128], every element written) */
int hoig_attn_gs_gather(const int32_t *index, const float *flow, const float *dhidden, float *dgs, int B, int H, int W,
                        hoig_stream_t stream);

/* Stand-alone drop-ins for the reference's two pybind ops, same argument meaning, contiguous NCHW fp32,
 * caller zero-fills outputs: block_extractor_cuda.forward/backward (block_extractor_cuda.cc:5-33) and
 * local_attn_reshape_cuda.forward/backward (local_attn_reshape_cuda.cc:5-29). */
int hoig_block_extractor_forward(const float *source, const float *flow, float *output, int B, int C, int Hs, int Ws,
                                 int Hf, int Wf, int kernel_size, hoig_stream_t stream);
int hoig_block_extractor_backward(const float *source, const float *flow, const float *grad_output, float *grad_source,
                                  float *grad_flow, int B, int C, int Hs, int Ws, int Hf, int Wf, int kernel_size,
                                  hoig_stream_t stream);
int hoig_local_attn_reshape_forward(const float *inputs, float *output, int B, int Hs, int Ws, int kernel_size,
                                    hoig_stream_t stream);
int hoig_local_attn_reshape_backward(const float *grad_output, float *grad_inputs, int B, int Hs, int Ws,
                                     int kernel_size, hoig_stream_t stream);

/* ---- sampling (generator.py:466-478; spade.py:30) ---- */
/* F.grid_sample(x, grid) bilinear / zeros / align_corners=False. x:[B,H,W,C] NHWC, grid:[B,Ho,Wo,2], y:[B,Ho,Wo,C] */
int hoig_grid_sample_fwd(const float *x, const float *grid, float *y, int B, int H, int W, int C, int Ho, int Wo,
                         hoig_stream_t stream);
/* dx accumulates (caller zero-fills) */
int hoig_grid_sample_bwd(const float *grid, const float *dy, float *dx, int B, int H, int W, int C, int Ho, int Wo,
                         hoig_stream_t stream);
/* F.interpolate(bilinear, align_corners=True) on NHWC [B,Hi,Wi,C] -> [B,Ho,Wo,C] */
int hoig_resize_bilinear_ac(const float *x, float *y, int B, int Hi, int Wi, int C, int Ho, int Wo,
                            hoig_stream_t stream);
/* F.interpolate(nearest) NHWC */
int hoig_resize_nearest(const float *x, float *y, int B, int Hi, int Wi, int C, int Ho, int Wo, hoig_stream_t stream);
/* attention flow of generator.py:484-488: flow[b,ch,y,x] = Tscale[b,y,x,ch] - idt ; idt ch0 = -1+2*y/h (the ROW
 * coordinate, 'ij' meshgrid quirk), ch1 = -1+2*x/h.  tscale:[B,h,h,2] -> flow:[B,2,h,h] */
int hoig_attn_flow(const float *tscale, float *flow, int B, int h, hoig_stream_t stream);

/* ---- pooling for the VGG19 feature path (vgg19.py:56; MaxPool2d(2,2)) NHWC ---- */
int hoig_maxpool2_fwd(const float *x, float *y, int B, int H, int W, int C, hoig_stream_t stream);
int hoig_maxpool2_bwd(const float *x, const float *y, const float *dy, float *dx, int B, int H, int W, int C,
                      hoig_stream_t stream);

/* ---- pointwise / layout ---- */
int hoig_nchw_to_nhwc(const float *x, float *y, int B, int C, int H, int W, hoig_stream_t stream);
int hoig_nhwc_to_nchw(const float *x, float *y, int B, int C, int H, int W, hoig_stream_t stream);
/* y[:, c_off : c_off+Cx] = x  for NHWC tensors with Cy channels (channel concat building block) */
int hoig_copy_channels(const float *x, float *y, int64_t npix, int Cx, int x_off, int Cy, int y_off, int Ccopy,
                       int accumulate, hoig_stream_t stream);
/* torch.cat([x1, x2], channel axis) of two contiguous [npix][C] tensors in one pass over y (coalesced for any C1 + C2) */
int hoig_cat2_channels(const float *x1, int C1, const float *x2, int C2, float *y, int64_t npix, hoig_stream_t stream);
/* y = a + b (n elements) ; y = act_bwd: dx = dy * act'(y) */
int hoig_add(const float *a, const float *b, float *y, int64_t n, hoig_stream_t stream);
/* y = act(a + b): the activation of a convolution that was split over two input tensors (conv(cat[x1, x2]) = conv(x1) + conv(x2)) */
int hoig_add_act(const float *a, const float *b, float *y, int act, float slope, int64_t n, hoig_stream_t stream);
int hoig_act_bwd(const float *y, const float *dy, float *dx, int act, float slope, int64_t n, hoig_stream_t stream);
/* the same, fused with the bias gradient of the convolution that produced y [rows][C]: g = dy * act'(y), dbias[c] += sum_rows g */
int hoig_act_bwd_colsum(const float *y, const float *dy, float *g, float *dbias, int act, float slope, int64_t rows, int C,
                        hoig_stream_t stream);
/* column sums: out[c] += sum_rows x[row][c]  (bias gradients) */
int hoig_colsum_accum(const float *x, float *out, int64_t rows, int C, hoig_stream_t stream);

/* alpha compositing of trainer.py:400-401 : img = mbg*bg + (1-mbg)*(obj*mh + hand*(1-mh)); NHWC, masks 1 channel */
int hoig_compose_fwd(const float *bg, const float *obj, const float *hand, const float *mbg, const float *mh, float *img,
                     int64_t npix, int C, hoig_stream_t stream);
int hoig_compose_bwd(const float *bg, const float *obj, const float *hand, const float *mbg, const float *mh,
                     const float *dimg, float *dbg, float *dobj, float *dhand, float *dmbg, float *dmh, int64_t npix,
                     int C, hoig_stream_t stream);

/* ---- losses (trainer.py:436-481): each writes sum-reductions into out[] (caller-zeroed, fp32 atomics of
 *      per-block partials) and, when dpred != NULL, the gradient of (scale * mean-loss) w.r.t. pred. ---- */
enum { HOIG_LOSS_L1 = 0, HOIG_LOSS_MSE = 1, HOIG_LOSS_BCE = 2 };
/* out[0] += sum loss(pred, target) ; dpred = gscale * dloss/dpred (gscale already includes 1/n and lambda).
 * target_const used when target == NULL (LSGAN targets 0/+1/-1, trainer.py:439,467-468). */
int hoig_loss_fwd_bwd(int kind, const float *pred, const float *target, float target_const, float gscale, float *out,
                      float *dpred, int64_t n, hoig_stream_t stream);
/* TV-L1 smoothness trainer.py:479-481 on [B,H,W] single-channel maps: out[0] += sum|dx|, out[1] += sum|dy| ;
 * dm (overwritten) = gx * d(sum|dx|) + gy * d(sum|dy|) */
int hoig_tv_fwd_bwd(const float *m, float gx, float gy, float *out, float *dm, int B, int H, int W, hoig_stream_t stream);
/* The same two as TERMS OF AN OBJECTIVE (trainer.py:448-457: loss_G = g_adv + g_rec + g_tsf + g_mask + g_mask_smooth): *term +=
 * gscale * sum loss(pred, target)  resp.  gx * sum|dx| + gy * sum|dy|  -- the scaled value goes straight into a caller-owned
 * fp32 slot (several calls may share one: the five VGG levels of g_tsf), so that the host composes the objective without
 * per-term scalar launches. */
int hoig_loss_accumulate(int kind, const float *pred, const float *target, float target_const, float gscale, float *term,
                         float *dpred, int64_t n, hoig_stream_t stream);
int hoig_tv_accumulate(const float *m, float gx, float gy, float *term, float *dm, int B, int H, int W, hoig_stream_t stream);
/* out[0] += sum x */
int hoig_sum(const float *x, float *out, int64_t n, hoig_stream_t stream);
/* out[0] += scale * sum x   (a mean written straight into a report slot: d_real / d_fake, trainer.py:470-471) */
int hoig_sum_scaled(const float *x, float scale, float *out, int64_t n, hoig_stream_t stream);

/* ---- fused Adam over one flat parameter buffer (torch.optim.Adam defaults of trainer.py:275-278:
 *      no weight decay, no amsgrad): step is the 1-based step count after increment ---- */
int hoig_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, double lr,
                   double beta1, double beta2, double eps, int step, float grad_scale, hoig_stream_t stream);

/* The same update with the schedule in DEVICE memory (an optimiser step that can be captured in a hipGraph):
 * `state` = {lr, beta1, beta2, eps, step} as doubles (the host only ever rewrites lr: Trainer.update_learning_rate,
 * trainer.py:574-591); hoig_adam_tick advances state[4] by one and writes the six fp32 scalars of this step into `derived`
 * (bias corrections evaluated in double, as hoig_adam_step does on the host); hoig_adam_step_dev applies them to a slice. */
int hoig_adam_tick(double *state, float *derived, hoig_stream_t stream);
int hoig_adam_step_dev(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, const float *derived,
                       float grad_scale, hoig_stream_t stream);
/* hoig_adam_step_dev over a WHOLE flat buffer and hoig_pack_conv_weights_bf16_all of the updated weights in one launch: the
 * weights of the table (`segs`, `nseg`, `ntiles`: as for hoig_pack_conv_weights_bf16_all; every table weight has Co % 32 == Ci % 32
 * == 0 and the rows must not overlap) are updated tile by tile and their planes written from the registers that hold the new values;
 * `plain` (device memory) holds nplain rows of 2 int64 {first element, count <= 1024, both multiples of 4}: the rest of the buffer,
 * updated without planes.  Every element of the buffer must be covered exactly once by the table or by `plain`.  Same arithmetic as
 * hoig_adam_step_dev, element for element. */
int hoig_adam_pack_step(float *flat, const float *grad, float *exp_avg, float *exp_avg_sq, const float *derived, float grad_scale,
                        const int64_t *segs, int nseg, int64_t ntiles, const int64_t *plain, int64_t nplain, uint16_t *hi_f,
                        uint16_t *lo_f, uint16_t *hi_d, uint16_t *lo_d, hoig_stream_t stream);

/* A non-blocking HIP stream owned by the library (the step's side streams: see hoig_amd/ops.py new_stream for why they are not
 * taken from PyTorch's round-robin stream pool). */
int hoig_stream_create(hoig_stream_t *out);
int hoig_stream_destroy(hoig_stream_t stream);   /* also forgets the stream's scratch block */
/* Per-stream scratch, CALLER-owned like every other buffer of this interface (nothing behind it allocates device memory): kernels
 * that reduce per-workgroup partial sums through memory -- the thin-channel weight gradients behind hoig_conv2d_bwd_weight -- use the
 * block registered for the stream they are launched on (launches of one stream are ordered, so they share it).  Register >=
 * hoig_stream_scratch_bytes() bytes per stream that launches weight gradients; ptr = NULL forgets the stream.  Without a block those
 * kernels take a (slower, same-result-up-to-summation-order) atomic path and report it once on stderr.  The block must stay valid
 * until the stream's last launch that may use it has run. */
int64_t hoig_stream_scratch_bytes(void);
/* what hoig_conv2d_bwd_weight would use of it for layer d: hoig_stream_scratch_bytes() for the thin-channel layers, 0 for all others
 * (a caller may register scratch only on the streams that launch such layers) */
int64_t hoig_conv2d_bwd_weight_scratch_bytes(const hoig_conv_desc *d);
int hoig_stream_scratch_set(hoig_stream_t stream, void *ptr, int64_t bytes);

/* eval.py output stage (utils/util.py:249-264): uint8 = (x+1)/2*255 truncated, NHWC fp32 -> CHW uint8 grid tile */
int hoig_tensor2im_u8(const float *x, uint8_t *out, int B, int H, int W, int C, int nrow, int unnormalize,
                      hoig_stream_t stream);

/* ---- input preparation AFTER the rasteriser (tensor stage): HandRecoveryFlow.forward
 *      (models/trainer.py:46-145) + the MANORenderer helpers it calls (utils/nmr.py:567-595, 874-968, 973-1100) +
 *      util.morph (utils/util.py:142-158).  Planar NCHW fp32 (the reference's layout at this boundary), int32 face index
 *      maps (-1 = no face; ids < 1538 are hand faces), 256 x 256 images and a 256 x 640 texture atlas as in the reference.
 *      Replaces, per sample: get_texture_backward_warp -> hoig_prep_texture; encode_fim / encode_sem /
 *      sample_from_texture_dense + F.grid_sample(align_corners=True) / cal_bc_transform -> hoig_prep_lookup (once per view);
 *      and, per batch, trainer.py:73,79,82,110-141 -> hoig_prep_assemble. ---- */
/* src_img [3,256,256]; src_faces [F,3,3] (x, y, z of the projected face vertices as render_fim_wim returns them: y is
 * negated here, trainer.py:67-68); fim_uv [256,640], wim_uv [256,640,3], obj_tex_img [256,256,3] (HWC): the object's
 * buffers; occ_ws: 256*640 bytes of workspace; tex [3,256,640] out */
int hoig_prep_texture(const float *src_img, const float *src_faces, const int32_t *src_fim, const int32_t *fim_uv,
                      const float *wim_uv, const float *obj_tex_img, unsigned char *occ_ws, float *tex,
                      hoig_stream_t stream);
/* one view of one sample: fim [256,256], wim [256,256,3]; map_fn [(F+1),3], sem_full [F+1], faces_uv_coord [F,3,2] with
 * F = n_faces (row F = background, what index -1 addresses in the reference); tex from hoig_prep_texture.
 * out: cond [3,256,256], seg [256,256] (label), hand_region [256,256] (1 - hand faces, BEFORE the erosion),
 * rend [3,256,256]; T [256,256,2] (nullable; needs src_faces): cal_bc_transform of THIS view's fim / wim */
int hoig_prep_lookup(const int32_t *fim, const float *wim, const float *map_fn, const float *sem_full,
                     const float *faces_uv_coord, int n_faces, const float *tex, const float *src_faces, float *cond,
                     float *seg, float *hand_region, float *rend, float *T, hoig_stream_t stream);
/* The same two stages for a whole BATCH in one call each (round 6: two + one launches instead of B x (two + one); the chain of small
 * launches is what the raw-batch stage costs a loader-fed step).  Tensors are the per-sample ones stacked over B; src_faces is
 * [B][.][3][3] with face_stride floats between samples; the per-sample object buffers -- every sample may hold another object -- are
 * passed as HOST arrays of B device pointers (n_faces: host array of B ints) and travel to the kernels by value, so B <=
 * HOIG_PREP_MAX_BATCH (HOIG_EUNSUPPORTED beyond: the caller then loops over the per-sample entry points).  occ_ws: B * 256 * 640 bytes;
 * tex [B,3,256,640]. */
#define HOIG_PREP_MAX_BATCH 32
int hoig_prep_texture_batched(int B, const float *src_img, const float *src_faces, int64_t face_stride, const int32_t *src_fim,
                              const int32_t *const *fim_uv, const float *const *wim_uv, const float *const *obj_tex_img,
                              unsigned char *occ_ws, float *tex, hoig_stream_t stream);
int hoig_prep_lookup_batched(int B, const int32_t *fim, const float *wim, const float *const *map_fn, const float *const *sem_full,
                             const float *const *faces_uv_coord, const int *n_faces, const float *tex, const float *src_faces,
                             int64_t face_stride, float *cond, float *seg, float *hand_region, float *rend, float *T /*nullable*/,
                             hoig_stream_t stream);
/* batch: images [B,3,256,256]; the hoig_prep_lookup outputs of both views stacked over the batch; T_raw [B,256,256,2].
 * out: src_bg [B,4,..], tsf_bg (nullable: bg_both) [B,4,..], src_obj / tsf_obj [B,15,..], src_hand / ref_hand
 * [B,hand_channels,..] with hand_channels = 6 (HOv3) or 12 (DexYCB: + the six hand-part one-hots, HOIG_DexYCB/models/trainer.py:131,135),
 * T_hand [B,256,256,2], the four crop masks [B,1,..] (bg src, bg ref, hand src, hand ref) */
int hoig_prep_assemble(int B, const float *src_img, const float *ref_img, const float *cond_s, const float *cond_r,
                       const float *seg_s, const float *seg_r, const float *hr_s, const float *hr_r, const float *rend_s,
                       const float *rend_r, const float *T_raw, float *src_bg, float *tsf_bg, float *src_obj,
                       float *tsf_obj, float *src_hand, float *ref_hand, int hand_channels, float *T_hand, float *smb,
                       float *rmb, float *smh, float *rmh, hoig_stream_t stream);

/* ---- rasteriser of MANORenderer.render_fim_wim (utils/nmr.py:496-513): replaces
 *      nr.rasterize_face_index_map_and_weight_map(faces, image_size, anti_aliasing=False), i.e. neural_renderer's
 *      forward_face_index_map kernels 1+2 (thirdparty/neural_renderer/neural_renderer/cuda/rasterize_cuda_kernel.cu:40-186)
 *      with the wrapper's -1 / 0 fill and vertical flip (neural_renderer/rasterize.py:50-52,334-338).
 *      faces [B,F,3,3] (vertices_to_faces output: x, y in [-1,1], y up, z depth); fim [B,S,S] int32 (-1 = no face),
 *      wim [B,S,S,3]; workspace: hoig_rasterize_workspace_bytes(B, F) bytes.  Tile-binned (16x16-pixel tiles). ---- */
size_t hoig_rasterize_workspace_bytes(int B, int F);
/* The vertex stage in front of it for a whole batch in ONE launch: MANORenderer's projection (orthographic_proj_withz_idrot,
 * utils/nmr.py:109-140: cam_dim 15 = 3x3 camera matrix | 2x3 crop transform; HOIG_DexYCB/utils/nmr.py:146-163: cam_dim 10 = fx, fy, cx,
 * cy | crop transform), the y flip (:506), nr.look_at with the renderer's eye (z - eye_z) and nr.vertices_to_faces (:508-511).
 * verts [B][V][3]; face_lists: HOST array of B device pointers to [n_faces[b]][3] int64 vertex indices (a sample's object decides its
 * list), n_faces: host array; faces_out [B][Fmax][3][3], rows >= n_faces[b] filled with pad_value (a point no pixel can see).
 * B <= HOIG_PREP_MAX_BATCH. */
int hoig_project_faces(const float *cam, int cam_dim, const float *verts, int V, const int64_t *const *face_lists, const int *n_faces,
                       int B, int Fmax, float eye_z, float pad_value, float *faces_out, hoig_stream_t stream);
int hoig_rasterize_fim_wim(const float *faces, int B, int F, int image_size, float near, float far, int32_t *fim,
                           float *wim, void *workspace, hoig_stream_t stream);

/* ---- data loader, device side: the image work of HOv3Dataset._get_sample (HOIG_HOv3/data/hov3_dataset.py:215-223,
 *      63-91) and of __getitem__'s transform (:208-212,267) for a batch of decoded 8-bit frames.  Both replace calls into
 *      opencv-python 4.5.1.48 (requirements.txt:99) and follow its fixed-point algorithms integer for integer (hoig_amd/csrc/data_prep.hip).
 * hoig_resize_linear_u8: cv2.resize(src, (Wd, Hd)) with the default INTER_LINEAR (:219): uint8 [B][Hs][Ws][C] -> [B][Hd][Wd][C], C <= 4.
 * hoig_warp_affine_u8: cv2.warpAffine(src, M, (Wd, Hd), flags=cv2.INTER_LINEAR) (:78; BORDER_CONSTANT, value 0) with M the FORWARD
 *   2x3 transform of each sample as the reference passes it (float32 values widened to double: M [B][6]); mode 0: dst uint8
 *   [B][Hd][Wd][C]; mode 1 (C == 3): dst float [B][3][Hd][Wd], channel c = ((float(v[2-c]) / 255) - 0.5) / 0.5 -- astype(float32),
 *   / 255.0, [:, :, ::-1], ToTensor, Normalize(0.5, 0.5) (:79,222,209,267); mode 2: dst float [B][1][Hd][Wd] = float(v[C-1]) / 128
 *   (:223). ---- */
int hoig_resize_linear_u8(const uint8_t *src, int B, int Hs, int Ws, int C, uint8_t *dst, int Hd, int Wd, hoig_stream_t stream);
int hoig_warp_affine_u8(const uint8_t *src, int B, int Hs, int Ws, int C, const double *M, int Hd, int Wd, int mode, void *dst,
                        hoig_stream_t stream);

const char *hoig_version(void);

/* Kernel-variant choices that are tuning, not semantics (every value computes the same result up to summation order): one table
 * instead of per-variant environment switches.  key -> value; returns the previous value, or -1 for an unknown key; value < 0 only
 * queries.  Keys and defaults (each adopted on an interleaved A/B committed under profiles/r04_*):
 *   "s2_16"   1  the stride-2 3x3 layers on v_mfma_f32_16x16x32 (conv_halo16.hip; scatter launches with 64-channel tiles stay on 32x32x16)
 *   "igemm16" 1  the generic implicit GEMM on it (conv_igemm16.hip): 1 = forward launches, 2 = data gradients too
 *   "flat5"   2  the flattened-axis halo kernel (conv_flat16.hip): 1 = the attention's valid 5x5 convolutions and their data
 *                gradients, 2 = also the 3x3 "same" layers with too few tiles for the halo kernels
 *   "wflat5"  1  the weight gradient of the attention's valid 5x5 convolutions on the flattened pixel axis (wgrad_flat.hip):
 *                1 = where the output width is not a multiple of 32 (the 2 x 32-pixel halo kernel cannot run), 2 = always
 *   "head16"  1  the forward of the 7x7 image / mask heads (64 -> 3..5 channels) on 16x16x32 with the horizontal taps as MFMA
 *                columns (conv_head16.hip), three-term forward arithmetic only; 0: the exact-fp32 VALU kernel (conv_small.hip)
 *   "d_early" 1  (read by the host side, hoig_amd/models/trainer.py) without a gradient exchange the D step is issued on its stream BEFORE
 *                G's backward instead of after it; 0: after (both orders compute the same step: D's weights change only in D's own
 *                update, which stays last).  With an exchange (world > 1) it always follows G's backward: G's all-reduce hides behind it
 *   "pair"    2  (read by the host side, hoig_amd/models/networks/generator.py) the 3x3 512 -> 512 convolutions of src_model's and
 *                tsf_model's residual blocks as grouped launches (hoig_conv2d_*_pair): 1 = always, 2 = in CAPTURED steps only, 0 = never
 *                (one launch per sub-network, on two streams).  Measured (profiles/r05_pair_ab.txt): the eager step loses 0.6 ms to the
 *                lock-step of the two chains (the norms between the convolutions no longer overlap), the replayed graph gains 0.7 ms
 *   "wdma16"  2  the weight tiles of the 16x16x32 3x3 stride-1 kernel (1) and of the flattened-axis kernel (2) by LDS-DMA (conv_halo16.hip /
 *                conv_flat16.hip WDMA: inline-asm global_load_lds, one step ahead, pieces spread over the MFMA groups): +3..7 % on every
 *                layer of the first, +2..5 % on the attention's 5x5 layers, step 63.85 -> 63.03 -> 62.8 ms (profiles/r05_wdma16_ab.txt);
 *                0: through registers, two steps ahead
 *   "s2_pipe"  1  the stride-2 3x3 kernel on the 16x16 MFMA in its statically walked form (conv_s2_16.hip: loads in flight behind the
 *                MFMAs, weight tiles by LDS-DMA) where it measured faster: 8 x 32 tiles with eight waves wherever that leaves every CU
 *                a workgroup and N % 128 == 0 (12-24 %: 132 -> 100 us on 16 x 64x64 256 -> 512), 4 x 32 tiles for grids of at most one
 *                workgroup per CU (15-30 %); 2 / 3: the 4-row / 8-row form wherever it can run; 0: never.  profiles/r05_s2_dma_ab.txt
 *   "norm_in"  1  (read by the host side, hoig_amd/ops.py conv2d_after_norm) INFERENCE forwards apply the norm + ReLU of a single-reader
 *                conv - IN - ReLU - conv3x3 chain in the second convolution's loader (hoig_conv2d_fwd_packed_normin); 0: every norm is a
 *                pass of its own.  Batch-32 generator forward 2.511 -> 2.494 ms per image (the flagship's SPADE norms modulate per
 *                pixel and stay passes)
 *   "split_grads" 1  (read by the host side, hoig_amd/ops.py) the backward of a norm that follows an eligible 3x3 convolution writes its
 *                dx as bf16 hi | lo planes and that convolution's weight / data gradient read them without splitting ('PRE-SPLIT
 *                gradients' above); 0: fp32 gradients everywhere, split in every consuming workgroup
 *   "wino8"    0  (read by the host side, hoig_amd/ops.py _conv_fwd_raw; round 6) 1: three-term FORWARD launches of 3x3 stride-1 layers that
 *                the direct kernel would run on at most half the chip (<= 128 workgroups) and that ONE round of the Winograd F(2x2,3x3)
 *                kernel covers (<= 256 workgroups: 8 images of 512 -> 512 at 32 x 32) go to hoig_conv2d_fwd_wino.  Alone such a launch
 *                is 37 % faster; in the step they are the src / tsf twins, which already share the chip
 *                (profiles/r06_winograd_ab.txt: the in-step A/B); off
 * Round 6 removed the keys whose losing side had lost two rounds running, and with them that side's code: "wgrad16" (the 3x3 weight
 * gradient on 16x16x32: 5-20 % slower, wgrad_halo16.hip deleted), "mfma16" (the 8-row 3x3 stride-1 tilings on 16x16x32: always; their
 * 32x32x16 instantiations are gone), "wgrad_ko" (knock-out instantiations of the LDS-DMA weight gradient:
 * a diagnostic), "few128" / "wgrad_few" (launch sizing for the 8-image launches: always on), "adam_pack" / "pad_in" (always on),
 * "split_grads" 2 / 3 (a split pass of its own, SPADE's [dgamma | dbeta] as planes: both +0.25 ms per step, profiles/r05_split_*_ab.txt).
 * Process-wide, not synchronised: set before launching. */
int hoig_set_tuning(const char *key, int value);

/* ---- MANO hand layer: pose / shape parameters -> skinned hand vertices, the step in front of the rasteriser.
 *      Replaces, for this path, smplx 0.1.28's MANO layer (HOIG_HOv3/models/networks/hmr.py:55,84-85: `mano_layer_right(global_orient,
 *      hand_pose, betas, transl).vertices`, use_pca=False, flat_hand_mean=True) and manopth's ManoLayer (HOIG_DexYCB/models/networks/
 *      hmr.py:55-60,85-86: 45 PCA coefficients + hand mean, `+ trans`, x 1000 / 1000): linear blend skinning (smplx.lbs.lbs) in ONE
 *      launch, one workgroup per sample.
 * model (device, fp32): v_template [V][3], shapedirs [V][3][10], posedirs [135][V*3] (smplx's buffer layout), lbs_weights [V][16],
 *   parents [16] (int32, parents[0] ignored), j_template [16][3] = J_regressor . v_template and j_shapedirs [16][3][10] =
 *   J_regressor . shapedirs (the joint regression is linear in betas: folded once on the host).
 * pose: root [B][3] axis-angle; hand [B][45] axis-angle (hands_components == NULL) or [B][ncomps] PCA coefficients expanded with
 *   hands_components [ncomps][45]; hands_mean [45] (nullable = flat hand) is added either way.  betas [B][10], transl [B][3] (nullable).
 * out: verts rows 0..V-1 of a [B][ld_v][3] tensor (ld_v >= V: the caller's [hand | object] vertex buffer, hmr.py:89, is written
 *   in place), joints [B][16][3] posed joints + transl (nullable). */
int hoig_mano_lbs(const float *v_template, const float *shapedirs, const float *posedirs, const float *j_template,
                  const float *j_shapedirs, const float *lbs_weights, const int32_t *parents, const float *hands_mean /*nullable*/,
                  const float *hands_components /*nullable*/, int ncomps, int V, const float *root, const float *hand,
                  const float *betas, const float *transl /*nullable*/, float *verts, int ld_v, float *joints /*nullable*/, int B,
                  hoig_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
