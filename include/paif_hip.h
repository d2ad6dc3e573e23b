/* paif_hip.h -- C ABI of libpaif_hip.so: hand-written gfx950 (MI355X, CDNA4) kernels for the PAIF
 * hot path (fusion-net forward -> colour glue -> SegFormer -> PGD inner loop).
 *
 * The reference (LiuZhu-CV/PAIF) has NO native interface: every op below replaces an implicit
 * ATen/cuDNN call made by a torch nn.Module.  Each entry point cites the reference site it
 * replaces (file:line under the reference root).  Contract:
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless noted;
 *   - asynchronous on `stream` (a hipStream_t passed as void*); never allocates / frees / syncs;
 *   - returns 0 on success, a negative PAIF_E* code on bad arguments, or the positive hipError_t of
 *     a failed launch; paif_last_error() gives a thread-local message;
 *   - re-entrant, no global mutable state (autograd calls backward bodies from its own thread).
 * Activations are float32 NHWC ("channels-last": [B,H,W,C]) unless a parameter says otherwise.
 */
#ifndef PAIF_HIP_H
#define PAIF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PAIF_ABI_VERSION 1
#define PAIF_EINVAL (-1)   /* bad argument (null pointer, unsupported size) */
#define PAIF_ENOSUP (-2)   /* combination not built (kernel size / dilation / channels) */

typedef void* paif_stream_t; /* hipStream_t */

int paif_version(void);
const char* paif_last_error(void);
/* number of CUs of the current device (grid sizing on the host side); <0 on error */
int paif_device_cus(void);

/* ---------------------------------------------------------------------------------------------
 * Colour transforms and the fusion->segmentation glue
 * ------------------------------------------------------------------------------------------- */

/* RGB2YCrCb, core/model_fusion_auto.py:69-92.  rgb, ycc: NCHW [B,3,H,W]. */
int paif_rgb2ycrcb_fwd(const float* rgb, float* ycc, int B, int H, int W, paif_stream_t stream);

/* YCrCb2RGB stand-alone, core/model_fusion_auto.py:94-111: (x + [0,-.5,-.5]) @ [[1,1,1],[1.403,-.714,0],[0,-.344,1.773]],
 * no clamp.  ycc, rgb: NCHW [B,3,H,W]. */
int paif_ycrcb2rgb_fwd(const float* ycc, float* rgb, int B, int H, int W, paif_stream_t stream);

/* cat(fused, Cr, Cb) -> YCrCb2RGB -> clamp[0,1]  (core/model_fusion_auto.py:715-720) with per-block
 * min/max partials for the batch-global min-max that follows (:721-723).
 *   fused NCHW [B,1,H,W]; ycc NCHW [B,3,H,W] (Cr,Cb read); rgb_out NCHW [B,3,H,W];
 *   minmax_partial: [2 * paif_minmax_blocks(B,H,W)] floats (mins then maxs). */
int paif_minmax_blocks(int B, int H, int W);
int paif_recompose_clamp_fwd(const float* fused, const float* ycc, float* rgb_out, float* minmax_partial,
                             int B, int H, int W, paif_stream_t stream);
/* Fused-image writer post-processing, test_original.py:186-197, on the output of paif_recompose_clamp_fwd (rgb + partials):
 * q = uint8(255*rgb) (truncation), batch-global min/max of q, uint8(255 * (q - mn)/(mx - mn)) with the ratio in float64.
 * out: NHWC uint8 [B,H,W,3] (device). */
int paif_fused_uint8_fwd(const float* rgb, const float* minmax_partial, int npartial, unsigned char* out, int B, int H, int W,
                         paif_stream_t stream);
/* (x - min)/(max - min) * 255, then per channel (x - mean[c]) / std[c]   (:721-727), in place allowed.
 * Reduces the partials itself (every block re-reduces them in a fixed order: deterministic).
 * minmax_out (optional, 2 floats): the global min and max. */
int paif_minmax_normalize_fwd(const float* rgb, const float* minmax_partial, int npartial, float* out,
                              float* minmax_out, int B, int H, int W, paif_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fusion network (Network_Fusion_Searched, core/model_fusion_auto.py:599-635)
 * ------------------------------------------------------------------------------------------- */

/* stem_1 / stem_2: Conv2d(1,32,3,pad 1,no bias) + PReLU (:607-614) fused with Cell_Decom.get_residue
 * (max_c - min_c, :517-521).  img: 1-channel planes [H,W], image b at img + b*img_bstride floats
 * (so a channel-0 view of an NCHW [B,3,H,W] tensor needs no copy); w [32,1,3,3]; prelu: 1 float on
 * device; feat NHWC [B,H,W,32]; guide [B,H,W] (may be NULL). */
int paif_stem_fwd(const float* img, size_t img_bstride, const float* w, const float* prelu, float* feat, float* guide,
                  int B, int H, int W, paif_stream_t stream);
/* Same, and the map once more as bf16 (feat_bf16: [B,H,W,32] `unsigned short` data, round-to-nearest-even): in the bf16 storage mode
 * the fp32 map feeds the fp32 guided-filter block, its bf16 twin the residual inputs of the bf16-stored layers. */
int paif_stem_fwd_twin(const float* img, size_t img_bstride, const float* w, const float* prelu, float* feat, float* feat_bf16,
                       float* guide, int B, int H, int W, paif_stream_t stream);
/* the same with the twin as IEEE fp16 (the fp16 storage mode, round 5).  Round 6: `feat` may be NULL -- the fp16 forward then keeps the
 * 16-bit map only (the guided filter reads it through paif_guided_filter_fused_fwd_hf16_y16); the guide is formed from the fp32 values
 * either way. */
int paif_stem_fwd_twin_f16(const float* img, size_t img_bstride, const float* w, const float* prelu, float* feat, float* feat_f16,
                           float* guide, int B, int H, int W, paif_stream_t stream);
/* Cell_Decom.get_residue on an existing NHWC [B,H,W,32] map (:517-521): guide = max_c - min_c. */
int paif_channel_residue_fwd(const float* x, float* guide, int B, int H, int W, paif_stream_t stream);

/* Guided filter r=4, 1-channel guide, 32-channel target, for BOTH eps of Cell_Decom.decomposition
 * (:522-535; third-party guided_filter_pytorch.GuidedFilter, see oracle/shims).
 * Stage 1: per-channel linear coefficients.  ab: [4][B,H,W,32] = A(eps0), b(eps0), A(eps1), b(eps1).
 * Stage 2: LF_e = box(A_e)/N * guide + box(b_e)/N.  lf: [2][B,H,W,32].
 * Box sums are direct 9x9 border-clipped window sums (fp32), not cumsum differences.
 * Returns PAIF_EINVAL when H <= 9 or W <= 9 (the package asserts h,w > 2r+1). */
int paif_guided_filter_ab_fwd(const float* guide, const float* y, float* ab, float eps0, float eps1,
                              int B, int H, int W, paif_stream_t stream);
int paif_guided_filter_lf_fwd(const float* guide, const float* ab, float* lf, int B, int H, int W,
                              paif_stream_t stream);
/* Round 6, the taped forward the attack loops and the training step run (csrc/gf_taped.hip; replaces the same reference site,
 * core/model_fusion_auto.py:522-535 Cell_Decom.decomposition's two GuidedFilter(4, eps) calls, under autograd): the same two filters with the
 * TWO-map tape mc [2][B,H,W,32] = (mean_y, cov) instead of ab's four -- A_e, b_e are re-formed where they are used from mc and
 * the per-pixel guide statistics in `workspace` (paif_guided_filter_fused_workspace_floats(B,H,W) floats, filled here; the reverse
 * pass paif_guided_filter_bwd_input_mc reads it again).  lf [2][B,H,W,32].  paif_guided_filter_taped_fits: 1 if the streaming
 * kernels take the size (their row offsets are 32-bit), else the caller keeps paif_guided_filter_ab_fwd + _lf_fwd. */
int paif_guided_filter_taped_fits(int B, int H, int W);
int paif_guided_filter_taped_fwd(const float* guide, const float* y, float* mc, float* lf, float eps0, float eps1,
                                 float* workspace, int B, int H, int W, paif_stream_t stream);
/* Both stages fused (inference: the coefficient maps never reach HBM); same result as ab_fwd + lf_fwd.
 * workspace: paif_guided_filter_fused_workspace_floats(B,H,W) floats (per-pixel guide statistics for both eps). */
size_t paif_guided_filter_fused_workspace_floats(int B, int H, int W);
int paif_guided_filter_fused_fwd(const float* guide, const float* y, float* lf, float eps0, float eps1, float* workspace,
                                 int B, int H, int W, paif_stream_t stream);
/* Same, the two low-frequency maps written as bf16 (`lf`: [2][B,H,W,32] `unsigned short` data, round to nearest even): the bf16
 * configuration holds the maps behind the guided-filter block as bf16; statistics, A, b and every sum stay fp32. */
int paif_guided_filter_fused_fwd_bf16(const float* guide, const float* y, float* lf, float eps0, float eps1, float* workspace, int B,
                                      int H, int W, paif_stream_t stream);
/* The fp16 configuration (round 5): the two HIGH-frequency maps HF_e = y - LF_e (core/model_fusion_auto.py:531-532 forms them from the
 * filter's output anyway) written as IEEE fp16 (`hf`: [2][B,H,W,32] `unsigned short` data, round to nearest even).  |HF| << |LF| ~ |y|:
 * the fp16 rounding of what the folded 1x1 behind this block reads is ~8x smaller than with LF stored (paif_pack_decomp1x1_hf_weight_f16x2
 * folds that conv over [x, HF1, HF2]).  Statistics, A, b, every sum and y - LF itself are fp32. */
int paif_guided_filter_fused_fwd_hf16(const float* guide, const float* y, float* hf, float eps0, float eps1, float* workspace, int B,
                                      int H, int W, paif_stream_t stream);

/* Round 6: the same with y READ as IEEE fp16 (`y16`: [B,H,W,32] `unsigned short` data -- the 16-bit twin the stem writes, the same map the
 * folded 1x1 takes as its first source): HF_e = y16 - LF_e(y16); the fp16 forward then has no fp32 stem map at all (replaces the same
 * reference lines, core/model_fusion_auto.py:522-535). */
int paif_guided_filter_fused_fwd_hf16_y16(const float* guide, const float* y16, float* hf, float eps0, float eps1, float* workspace,
                                          int B, int H, int W, paif_stream_t stream);

/* Dense k x k convolution, stride 1, "same" zero padding (pad = dil*(k-1)/2), Cout <= 32, inputs =
 * virtual concat of up to 3 NHWC sources of `cin` channels each, fp32 MFMA (v_mfma_f32_32x32x2_f32)
 * implicit GEMM with an LDS-staged halo tile.  Replaces BasicConv / nn.Conv2d inside
 * ResidualDenseBlock (operations_m.py:435-449), ResidualModule (:451-464), ECABasicBlock (:368-393),
 * DilConv's 1x1 (:501), Cell_Decom.conv1x1_* (core/model_fusion_auto.py:501-502), stem_out.0 (:616).
 *
 *   y   = act( conv(in_act(src...)) * scale[c] + shift[c] )        scale/shift NULL -> 1 / 0
 *   out = y * alpha + res[0] + res[1] + res[2]                      res NULL -> skipped
 *   pool_partial (optional): [paif_conv2d_blocks(B,H,W)][32] per-block channel sums of `out`
 *                            (ECA's AdaptiveAvgPool2d, operations_m.py:358)
 */
typedef struct {
  const float* src[3];   /* NHWC [B,H,W,cin] each */
  int nsrc;              /* 1..3 */
  int cin;               /* channels per source: 32 or 16 */
  const float* wpk;      /* packed by paif_pack_conv_weight */
  int kh;                /* 1,3,5,7 */
  int dil;               /* 1,2 */
  int in_act;            /* 0 none, 1 PReLU(in_prelu) on every source, 2 ReLU */
  const float* in_prelu; /* 1 float on device */
  const float* scale;    /* [cout] or NULL */
  const float* shift;    /* [cout] or NULL */
  int act;               /* 0 none, 1 PReLU(prelu), 2 ReLU */
  const float* prelu;    /* 1 float on device */
  float alpha;
  const float* res[3];   /* NHWC [B,H,W,cout] or NULL */
  float* out;            /* NHWC [B,H,W,cout] */
  int cout;              /* 32 or 16 */
  float* pool_partial;   /* optional */
  int precision;         /* PAIF_CONV_F32 (exact fp32 MFMA), PAIF_CONV_BF16X3 (split-bf16, ~1e-5 rel.) or PAIF_CONV_BF16 (plain bf16
                            MFMA, fp32 accumulate: bf16-stored maps only, weights rounded to bf16 = the hi half of the BF16X3 pack);
                            wpk must have been packed for the same precision (BF16 takes the BF16X3 pack) */
  /* ---- backward-pass support (all optional / zero for a plain forward) ---- */
  float* aux_out;        /* forward: also store the pre-activation z = conv*scale+shift (saved for dgrad) */
  const float* in_aux;   /* dgrad staging, in_act 3/4: the saved pre-activation of the layer being back-propagated */
  const float* in_scale; /* in_act 3/4/5: per-input-channel factor (that layer's folded BN scale) or NULL */
  float in_alpha;        /* in_act 3/4/5: scalar factor (that layer's alpha) */
  const float* epi_aux;  /* epi_dact: tensor gating the output */
  int epi_dact;          /* 0 none; 1: out *= (epi_aux >= 0 ? 1 : *prelu); 2: out *= (epi_aux > 0), before residuals */
  int reverse_tiles;     /* 1: walk the tiles from the end of each XCD's range.  Results do not depend on it; a caller that alternates it
                            between consecutive layers ("serpentine") lets a layer start with the inputs its producer wrote last, which
                            are still in L2 / MALL (+0.7 % on the fusion forward).  Not honoured by the wave-specialised kernel */
  int storage;           /* PAIF_ST_*: storage of the 32-channel activation maps (src / res / out are then `unsigned short` bf16 data behind
                            the float* fields).  BASELINE configs[1] "bf16": fp32 accumulate, bf16 maps, round-to-nearest-even on store.
                            Built for the split-bf16 kernels of the inference forward (no gradient hooks) */
  float* cpool;          /* optional (round 5; cout = 32): ChannelPool of the OUTPUT map fused into the epilogue (core/model_fusion_auto.py:1352-1355):
                            cpool[4 * pixel + 0] = max_c out, cpool[4 * pixel + 1] = mean_c out (fp32, from the un-rounded fp32 values).  The pointer is
                            pre-offset by 0 (infrared map) or 2 (visible map) floats into the [B,H,W,4] plane spatial_attn_layer_M reads.  Only the
                            kernels paif_conv2d_can_cpool() reports; paif_channel_pool1_fwd* is the stand-alone pass otherwise.  LAST field */
} paif_conv_desc;
#define PAIF_ST_F32 0          /* fp32 maps in, fp32 maps out (default) */
#define PAIF_ST_BF16 1         /* bf16 in, bf16 out */
#define PAIF_ST_F32_BF16 2     /* fp32 in, bf16 out: the 1x1 behind the fp32 guided-filter block */
#define PAIF_ST_F16 3          /* round 5: IEEE fp16 in, fp16 out (`unsigned short` data behind the float* fields, round-to-nearest-even on
                                  store, fp32 accumulate): 8x finer than bf16 at the same bytes; precision PAIF_CONV_F16 / PAIF_CONV_F16X2 */
#define PAIF_ST_F16_F32 4      /* fp16 in (sources and residual maps), fp32 out: the conv that writes the forward's last 32-channel map
                                  (core/model_fusion_auto.py:634, feature2), which feeds the fp32-input stem_out kernel; 3x3 dilation 2 */
/* in_act: 0 none, 1 PReLU, 2 ReLU, 3 src*in_alpha*in_scale[c]*(in_aux>=0?1:*in_prelu), 4 ...*(in_aux>0), 5 src*in_alpha*in_scale[c] */
#define PAIF_CONV_F32 0
#define PAIF_CONV_BF16X3 1
#define PAIF_CONV_BF16 2        /* storage != PAIF_ST_F32 only: one bf16 MFMA per product (BASELINE configs[1] "bf16") */
#define PAIF_CONV_F16 4         /* PAIF_ST_F16 / PAIF_ST_F16_F32 only: one fp16 MFMA per product; wpk = the F16X2 pack (its hi pieces are read) */
#define PAIF_CONV_F16X2 5       /* PAIF_ST_F16 only, 1x1: weights as fp16 hi + lo (two MFMAs per product, 22-bit weights): the folded
                                   decomposition 1x1, whose weight rounding is the one that moves the segmentation argmax */
#define PAIF_CONV_F16X3 6       /* fp32 storage, cin = 32 (paif_conv2d_fwd: wpk = paif_pack_conv_weight_f16x2 of 2^8 * w) and paif_gemm_fwd /
                                   paif_gemm_masked_fwd / paif_gemm_splitk_fwd_p / paif_gemm_conv_fwd: each operand as TWO IEEE fp16 pieces (22
                                   significant bits), hi*hi + hi*lo + lo*hi on the fp16 MFMA, the weight side scaled by 2^8 (undone on the
                                   accumulator): ~2^-21.5 per product for O(1) data at half the MFMAs of BF16X6 -- the attack loops' arithmetic.
                                   Operands must lie inside fp16's exponent range: a caller that sends gradients through it scales them first */
#define PAIF_CONV_BF16X6 3      /* fp32 storage, cin = 32: three bf16 pieces per operand, six MFMAs per product: fp32-level parity (2^-25 per
                                   product) at 6/16 of the exact fp32 MFMA's matrix-pipe time; the arithmetic of the attack loops */

/* ---- bf16-stored activation maps (PAIF_ST_BF16; BASELINE configs[1] "...bf16..."): the elementwise kernels of the fusion
 * network's inference forward on bf16 32-channel maps (bf16 data behind the float* of x / out / ir / vis / agg / o / r / a / b),
 * fp32 arithmetic, round-to-nearest-even on store.  Same reference sites as their fp32 twins (DilConv depthwise operations_m.py:499;
 * ChannelPool + spatial_attn_layer_M blend core/model_fusion_auto.py:1352-1368,631-632; eca_layer operations_m.py:353-367;
 * stem_out.1/.2 + tanh core/model_fusion_auto.py:616-635; Cell_Chain's `inp + ops(inp)` :445).
 * paif_cast_storage_fwd converts n elements between fp32 and a 16-bit format; n % 4 == 0. ---- */
int paif_dwconv_fwd_bf16(const float* x, const float* w, float* out, int k, int dil, int in_relu, int B, int H, int W,
                         paif_stream_t stream);
int paif_channel_pool2_fwd_bf16(const float* ir, const float* vis, float* comp, int B, int H, int W, paif_stream_t stream);
int paif_spa_blend_fwd_bf16(const float* comp, const float* w, const float* ir, const float* vis, float* agg, int B, int H, int W,
                            paif_stream_t stream);
int paif_eca_finish_fwd_bf16(const float* o, const float* r, const float* pool_partial, const float* w1d, int k,
                             const float* prelu, float* gate, float* out, int B, int H, int W, paif_stream_t stream);
int paif_tail_fwd_bf16(const float* x, const float* w, const float* prelu, float* fused, int B, int H, int W, paif_stream_t stream);
int paif_add_fwd_bf16(const float* a, const float* b, float* out, size_t n, paif_stream_t stream);
/* mode: 0 bf16 -> fp32, 1 fp32 -> bf16, 2 fp32 -> fp16, 3 fp16 -> fp32 */
int paif_cast_storage_fwd(const float* src, float* dst, size_t n, int mode, paif_stream_t stream);
/* ---- the same kernels on IEEE-fp16-stored maps (PAIF_ST_F16, round 5): fp16 data behind the float*, fp32 arithmetic ---- */
int paif_dwconv_fwd_f16(const float* x, const float* w, float* out, int k, int dil, int in_relu, int B, int H, int W,
                        paif_stream_t stream);
int paif_channel_pool2_fwd_f16(const float* ir, const float* vis, float* comp, int B, int H, int W, paif_stream_t stream);
int paif_spa_blend_fwd_f16(const float* comp, const float* w, const float* ir, const float* vis, float* agg, int B, int H, int W,
                           paif_stream_t stream);
int paif_eca_finish_fwd_f16(const float* o, const float* r, const float* pool_partial, const float* w1d, int k,
                            const float* prelu, float* gate, float* out, int B, int H, int W, paif_stream_t stream);
int paif_tail_fwd_f16(const float* x, const float* w, const float* prelu, float* fused, int B, int H, int W, paif_stream_t stream);
int paif_add_fwd_f16(const float* a, const float* b, float* out, size_t n, paif_stream_t stream);

int paif_conv2d_blocks(int B, int H, int W);
int paif_conv2d_fwd(const paif_conv_desc* d, int B, int H, int W, paif_stream_t stream);
/* 1 if the kernel paif_conv2d_fwd runs for this descriptor / shape writes d->cpool (the fused ChannelPool of its output), else 0. */
int paif_conv2d_can_cpool(const paif_conv_desc* d, int B, int H, int W);
/* ChannelPool of ONE NHWC-32 map into the interleaved [B,H,W,4] plane (comp_off = comp + 0 or + 2): comp_off[4 p] = max_c x, comp_off[4 p + 1] =
 * mean_c x -- the stand-alone pass for producers whose epilogue does not pool (core/model_fusion_auto.py:1352-1355).  fp32 / bf16 / fp16 maps. */
int paif_channel_pool1_fwd(const float* x, float* comp_off, int B, int H, int W, paif_stream_t stream);
int paif_channel_pool1_fwd_bf16(const float* x, float* comp_off, int B, int H, int W, paif_stream_t stream);
int paif_channel_pool1_fwd_f16(const float* x, float* comp_off, int B, int H, int W, paif_stream_t stream);
/* 1 if paif_conv2d_fwd would run this descriptor on the persistent wave-specialised kernel (conv_bf16x3_ws),
 * 0 for the tile-per-workgroup kernel (conv_mfma_*): lets a profiler-side caller name the kernel it times. */
int paif_conv2d_is_persistent(const paif_conv_desc* d, int B, int H, int W);
/* The kernel paif_conv2d_fwd runs for this descriptor and shape, named as rocprofv3 lists it (without the anonymous
 * namespace and argument list), e.g. "conv_bf16x3_ms<3, 1, 2>": tile-per-workgroup (conv_mfma_bf16x3<k, d, hooks>),
 * multi-source 3x3 (conv_bf16x3_ms), resident-B persistent 3x3 (conv_bf16x3_res), wave-specialised persistent
 * (conv_bf16x3_ws), exact fp32 (conv_mfma_f32).  For timing tags and dispatch tests; writes a NUL-terminated string. */
int paif_conv2d_kernel_name(const paif_conv_desc* d, int B, int H, int W, char* buf, int buflen);
/* w: torch layout [cout, nsrc*cin, kh, kh]; wpk: paif_conv_wpk_floats(...) floats.
 * Layout wpk[src][tap][cin/8][64 lanes][4]: lane (h = lane>>5, n = lane&31) holds
 * w[n][src*cin + 8*o + 4*h + i][tap], i = 0..3 -- the B operand of four consecutive MFMAs. */
size_t paif_conv_wpk_floats(int nsrc, int cin, int kh);
int paif_pack_conv_weight(const float* w, float* wpk, int cout, int nsrc, int cin, int kh, paif_stream_t stream);
/* Same weights for precision = PAIF_CONV_BF16X3 (cin = 32 only): each value split into bf16 hi + bf16 lo,
 * wpk[src][tap][k16][hi|lo][64 lanes][8 bf16] (same size in bytes as the fp32 packing). */
int paif_pack_conv_weight_bf16x3(const float* w, float* wpk, int cout, int nsrc, int kh, paif_stream_t stream);
/* fp16 pieces (hi = rn(w), lo = rn(w - hi)) in the BF16X3 stream layout, for PAIF_CONV_F16 / PAIF_CONV_F16X2: paif_conv_wpk_floats(...) floats. */
int paif_pack_conv_weight_f16x2(const float* w, float* wpk, int cout, int nsrc, int kh, paif_stream_t stream);
/* Three-piece packs for PAIF_CONV_BF16X6: 1.5 x paif_conv_wpk_floats(...) floats. */
int paif_pack_conv_weight_bf16x6(const float* w, float* wpk, int cout, int nsrc, int kh, paif_stream_t stream);
/* dw [cin][1][kh][kh] (depthwise) and pw [cout][cin][1][1] (the 1x1 behind it) -> out [cout][cin][kh][kh] = pw * dw: the dense
 * kernel of  conv1x1(dwconv(x))  (operations_m.py:494-506 DilConv: ReLU -> depthwise -> conv1x1 -> BN), packed afterwards with
 * paif_pack_conv_weight*.  Used by the bf16 inference forward: one paif_conv2d_fwd (in_act = ReLU) instead of
 * paif_dwconv_fwd_bf16 + a 1x1, the depthwise map never goes to HBM. */
int paif_compose_dw_pw_weight(const float* dw, const float* pw, float* out, int cout, int cin, int kh, paif_stream_t stream);
/* w [cmid][cin][kh][kh] (a dense conv) and pw [cout][cmid][1][1] (the 1x1 behind it, nothing in between) -> out [cout][cin][kh][kh] =
 * sum_m pw[.][m] w[m][.][.]: the dense kernel of  conv1x1(conv_kxk(x))  (operations_m.py:451-464 ResidualModule: conv3x3 dil 2 -> conv1x1
 * -> BN -> PReLU; the 1x1 has no padding, so the composition is exact at the image border too).  Inference forward: one
 * paif_conv2d_fwd with the BN / PReLU / residual epilogue instead of two launches; the 32-channel map between them never goes to HBM. */
int paif_compose_pw_conv_weight(const float* pw, const float* w, float* out, int cout, int cmid, int cin, int kh, paif_stream_t stream);

/* ResidualDenseBlock (operations_m.py:435-449: x1 = P(c1(x)), x2 = P(c2([x, x1])), out = P(c3([x, x1, x2])) * alpha + x, one shared PReLU
 * slope, k = 3, dilation 1, 32 channels) of the 16-bit inference forward as ONE kernel (round 5, csrc/rdb_fused.hip): x read once, out
 * written once, x1 / x2 never leave LDS.  x / res0 / res1 / out: NHWC-32 16-bit maps (bf16, or IEEE fp16 when f16 != 0) behind the
 * float*; res0, res1 (optional, packed from res0): further residual maps added to the output (a block closing a Cell_Chain,
 * core/model_fusion_auto.py:445); cpool (optional): fused ChannelPool of the output as in paif_conv_desc.cpool.  wpk: the three convs'
 * weights as 16x16x32-MFMA operands (paif_rdb_fused_pack from w1 [32,32,3,3], w2 [32,64,3,3], w3 [32,96,3,3];
 * paif_rdb_fused_wpk_floats() floats).  Intermediate maps are rounded to the storage format like the three-launch form's. */
size_t paif_rdb_fused_wpk_floats(void);
int paif_rdb_fused_pack(const float* w1, const float* w2, const float* w3, float* wpk, int f16, paif_stream_t stream);
int paif_rdb_fused_fwd(const float* x, const float* wpk, const float* prelu, float alpha, const float* res0, const float* res1, float* out,
                       float* cpool, int f16, int reverse_tiles, int B, int H, int W, paif_stream_t stream);

/* stem_out of the fusion network (core/model_fusion_auto.py:616-620, :640: conv3x3 32->16, conv3x3 16->1, PReLU, tanh) on a bf16-stored
 * NHWC-32 map as one launch pair: the two linear convs composed into a 5x5 32->1 conv evaluated on the matrix cores with the TAPS as the
 * M dimension (three-piece bf16 weights: fp32-level products), and the outermost pixel ring -- where the reference's zero padding of the
 * 16-channel map breaks the composition -- corrected by its closed-form surplus (5-tap edge convs).  paif_stem_out_pack builds the weights
 * (paif_stem_out_pack_floats() floats) from w1 [16][32][3][3] and w2 [1][16][3][3].  x: bf16 data behind the float*; fused [B][1][H][W] fp32. */
int paif_stem_out_pack_floats(void);
int paif_stem_out_pack(const float* w1, const float* w2, float* wpk, paif_stream_t stream);
int paif_stem_out_fwd_bf16(const float* x, const float* wpk, const float* prelu, float* fused, int B, int H, int W, paif_stream_t stream);
/* the same on an fp32 NHWC-32 map (the fp32-storage forward, round 4): x as bf16 hi + lo (2^-17), five MFMAs per k-step -- the operand
 * split of every dense conv of that path; same packed weights. */
int paif_stem_out_fwd_f32(const float* x, const float* wpk, const float* prelu, float* fused, int B, int H, int W, paif_stream_t stream);
/* Round 6, range guard of the fp16 storage mode: the same launches (f32in != 0: paif_stem_out_fwd_f32, else _bf16) and `*flag |= 1` when a
 * pre-tanh value is inf / NaN -- an fp16 map of the forward overflowed (|v| >= 65520 stores inf; convs and residual adds carry it to this
 * kernel's input; tanh alone would turn it into a finite, wrong +-1).  flag: one 32-bit device word the caller clears and reads. */
int paif_stem_out_fwd_guard(const float* x, int f32in, const float* wpk, const float* prelu, float* fused, unsigned* flag, int B, int H, int W,
                            paif_stream_t stream);
int paif_pack_decomp1x1_weight_bf16x6(const float* w, float* wpk, paif_stream_t stream);
int paif_pack_decomp1x1_weight_bf16x3(const float* w, float* wpk, paif_stream_t stream);
/* fp16 forward (round 5): the guided filter writes HF_i = x - LF_i (small magnitudes: 8x less fp16 rounding than LF_i), and the same 1x1
 * folds over [x, HF1, HF2]: (Wl1+Wl2) x + (Wh1-Wl1) HF1 + (Wh2-Wl2) HF2; fp16 hi | lo pieces for PAIF_CONV_F16X2 (nsrc = 3). */
int paif_pack_decomp1x1_hf_weight_f16x2(const float* w, float* wpk, paif_stream_t stream);
/* Cell_Decom's 1x1 over cat[LF1,LF2,x-LF1,x-LF2] (core/model_fusion_auto.py:512-513) folded to a 1x1
 * over [x, LF1, LF2]: (Wh1+Wh2) x + (Wl1-Wh1) LF1 + (Wl2-Wh2) LF2.  w [32,128,1,1] -> wpk for nsrc=3. */
int paif_pack_decomp1x1_weight(const float* w, float* wpk, paif_stream_t stream);
/* eval-mode BatchNorm folded to scale/shift: scale = g/sqrt(var+eps), shift = b - mean*scale. */
int paif_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                 float* scale, float* shift, int C, paif_stream_t stream);

/* Depthwise k x k conv (groups = C = 32), optional ReLU on the input: DilConv / SepConv,
 * operations_m.py:499-500, 514-515.  x, out NHWC [B,H,W,32]; w [32,1,k,k]. */
int paif_dwconv_fwd(const float* x, const float* w, float* out, int k, int dil, int in_relu,
                    int B, int H, int W, paif_stream_t stream);

/* ChannelPool(ir, vis) + spatial_attn_layer_M + blend (core/model_fusion_auto.py:1352-1368, 631-632):
 * pool: comp NHWC [B,H,W,4] = (max_c ir, mean_c ir, max_c vis, mean_c vis);
 * blend: scale = sigmoid(conv5x5(comp)); agg = scale*ir + (1-scale)*vis.  w [1,4,5,5];
 * scale_out [B,H,W] optional. */
int paif_channel_pool2_fwd(const float* ir, const float* vis, float* comp, int B, int H, int W, paif_stream_t stream);
int paif_spa_blend_fwd(const float* comp, const float* w, const float* ir, const float* vis, float* agg,
                       float* scale_out, int B, int H, int W, paif_stream_t stream);

/* eca_layer tail + ECABasicBlock residual (operations_m.py:353-367, 390-392):
 * mean[b][c] = sum over that image's conv blocks of pool_partial / (H*W);
 * s = sigmoid(conv1d_k(mean)) over channels (zero padded); out = PReLU(o*s + r).
 * w1d [1,1,k]. */
int paif_eca_finish_fwd(const float* o, const float* r, const float* pool_partial, const float* w1d, int k,
                        const float* prelu, float* gate /* [B,32] out: the sigmoid gates */, float* out,
                        float* u_out /* optional: pre-activation o*gate + r, saved for the backward pass */,
                        int B, int H, int W, paif_stream_t stream);

/* stem_out.1 + .2 + tanh (core/model_fusion_auto.py:617-619,634): Conv2d(16,1,3,pad 1) -> PReLU -> tanh.
 * x NHWC [B,H,W,16]; w [1,16,3,3]; fused [B,H,W]. */
int paif_tail_fwd(const float* x, const float* w, const float* prelu, float* fused,
                  float* z_out /* optional: conv output before PReLU/tanh, saved for the backward pass */,
                  int B, int H, int W, paif_stream_t stream);

/* plain elementwise a + b (Cell_Chain's residual when it cannot be fused, :445); n floats. */
int paif_add_fwd(const float* a, const float* b, float* out, size_t n, paif_stream_t stream);


/* ---------------------------------------------------------------------------------------------
 * Segmentation network: MixTransformer encoder + SegFormer head (core/mix_transformer.py,
 * core/segformer_head.py).  Token tensors [B,N,C] (N = H*W row-major) are NHWC images.
 * ------------------------------------------------------------------------------------------- */

/* C[M,N] = act( (A[M,K] . W[N,K]^T) * scale[n] + shift[n] ) + res[M,N]  -- fp32 MFMA GEMM.
 * nn.Linear (core/mix_transformer.py:22-25,66-69; core/segformer_head.py:19), the 1x1 convs of the head
 * (core/segformer_head.py:50-57; scale/shift = folded BatchNorm or bias) and every strided conv after
 * paif_im2col_fwd (OverlapPatchEmbed.proj core/mix_transformer.py:168, Attention.sr :74).
 * lda/ldc/ldres: row strides in floats; K % 32 == 0; scale/shift/res may be NULL;
 * act: 0 none, 1 GELU (erf), 2 ReLU.
 * precision: PAIF_CONV_F32 (exact fp32 MFMA) or PAIF_CONV_BF16X3 (operands split into bf16 hi + lo at staging,
 * hi*hi + hi*lo + lo*hi on the bf16 MFMA, fp32 accumulate: ~1e-5 relative, 5.3x less matrix-pipe time). */
int paif_gemm_fwd(const float* A, int lda, const float* W, const float* scale, const float* shift, int act,
                  const float* res, int ldres, float* C, int ldc, int M, int N, int K, int precision,
                  paif_stream_t stream);
/* Split-K form of paif_gemm_fwd (exact fp32) for small output grids with a long k loop: paif_gemm_splitk_plan returns
 * the number of k splits (1 = do not split); the caller provides workspace[splits * M * N] floats.  Partial sums are
 * added in split order by a second pass (deterministic), which also applies scale / shift / act / res. */
int paif_gemm_splitk_plan(int M, int N, int K);
int paif_gemm_splitk_fwd(const float* A, int lda, const float* W, const float* scale, const float* shift, int act,
                         const float* res, int ldres, float* C, int ldc, int M, int N, int K, int splits,
                         float* workspace, paif_stream_t stream);
/* Same with the arithmetic of the partial products selectable (0 = exact fp32 MFMA, 1 = split-bf16); the reduction pass is fp32. */
int paif_gemm_splitk_fwd_p(const float* A, int lda, const float* W, const float* scale, const float* shift, int act,
                         const float* res, int ldres, float* C, int ldc, int M, int N, int K, int splits,
                         float* workspace, int precision, paif_stream_t stream);

/* Strided conv as a GEMM whose A operand is GATHERED from the NHWC map: paif_im2col_fwd + paif_gemm_fwd (OverlapPatchEmbed.proj
 * core/mix_transformer.py:168-169, Attention.sr :74) without materialising the im2col matrix; bit-identical to that pair.
 * x [B,H,W,Cin]; Wt [N, Kpad], Kpad = k*k*Cin rounded up to a multiple of 32 (paif_pack_conv_gemm_weight); out [B*OH*OW, >= N], row
 * stride ldc.  precision PAIF_CONV_BF16X3 / PAIF_CONV_BF16X6: Cin % 32 == 0 (a 32-wide k tile = 128 contiguous bytes of one input pixel);
 * splits = paif_gemm_splitk_plan(B*OH*OW, N, Kpad), splits > 1 needs workspace[splits * M * N] floats.  precision PAIF_CONV_F32 (exact
 * fp32 MFMA): any Cin, element-wise gather through a column table, Kpad <= 160 (the 3-channel 7x7 stride-4 patch embed), splits = 1.
 * PAIF_ENOSUP for other combinations: callers keep the im2col pair. */
int paif_gemm_conv_fwd(const float* x, int B, int H, int W, int Cin, int k, int stride, int pad, const float* Wt,
                       const float* scale, const float* shift, int act, const float* res, int ldres, float* out, int ldc,
                       int N, int precision, int splits, float* workspace, paif_stream_t stream);

/* The input gradient of a NON-OVERLAPPING strided conv (Attention.sr, core/mix_transformer.py:74: kernel = stride = sr_ratio) as ONE GEMM:
 * paif_gemm_fwd(dY, Wt) + paif_col2im_fwd with the col2im done by the GEMM epilogue's addresses (every column of the [M, sr*sr*C]
 * product lands on exactly one input pixel); bit-identical to that pair.  dY [B*(H/sr)*(W/sr), >= K] row stride lda; Wt [sr*sr*C, K]
 * (the transposed packed conv weight); dx [B,H,W,C]; H, W multiples of sr (PAIF_ENOSUP otherwise).  precision as paif_gemm_fwd. */
int paif_gemm_col2im_fwd(const float* dY, int lda, const float* Wt, float* dx, int B, int H, int W, int C, int sr, int K,
                         int precision, paif_stream_t stream);

/* Measurement helpers (bench.py's live roofline, BASELINE.json metric / SURVEY 8(d); not on the product path): HIP events created with
 * hipEventDisableSystemFence -- timing-only events: no system-scope fence between the kernels they bracket -- recorded on the launch
 * stream.  paif_timing_event_elapsed_ms needs both events completed (synchronise the stream first). */
int paif_timing_event_create(void** ev);
int paif_timing_event_record(void* ev, paif_stream_t stream);
int paif_timing_event_elapsed_ms(void* start, void* stop, float* ms);
int paif_timing_event_destroy(void* ev);

/* Wide-tile form of the split-bf16 paif_gemm_fwd (csrc/gemm_split2.hip) for the same nn.Linear sites (core/mix_transformer.py:22-25,
 * 66-69,74; core/segformer_head.py:19): workgroup tile 128 x 64*nt, wave tile 64 x 32*nt.  paif_gemm2_plan returns nt (1, 2, 4 or 5
 * column tiles per wave) for a shape it is built for, 0 otherwise (the caller then takes paif_gemm_fwd).  precision: PAIF_CONV_BF16X3 or
 * PAIF_CONV_BF16X6.  N % (64*nt) == 0; ldc, ldres % 4 == 0; 16-byte aligned A, C, res, scale, shift.  Same arithmetic per product and
 * the same k order as paif_gemm_fwd: results are bit-identical to it. */
int paif_gemm2_plan(int M, int N, int K, int precision);
/* W of paif_gemm2_fwd is the PRE-SPLIT image of the nn.Linear weight [N, K]: [K/32][N][pieces][32] bf16 (pieces = 2 for BF16X3, 3 for
 * BF16X6; paif_gemm2_packed_bytes bytes), written by paif_gemm2_pack_weight -- once per weight version, not per call. */
size_t paif_gemm2_packed_bytes(int N, int K, int precision);
int paif_gemm2_pack_weight(const float* W, void* out, int N, int K, int precision, paif_stream_t stream);
int paif_gemm2_fwd(const float* A, int lda, const void* W, const float* scale, const float* shift, int act,
                   const float* res, int ldres, float* C, int ldc, int M, int N, int K, int precision, int nt,
                   paif_stream_t stream);

/* nn.LayerNorm over the last dim (core/mix_transformer.py:75,122,127,172,232-253). x,y [M,C]; C % 4 == 0. */
int paif_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, int M, int C, float eps,
                       paif_stream_t stream);

/* im2col of an NHWC [B,H,W,Cin] map for a k x k conv with stride/pad: col [B*OH*OW, Kpad],
 * column index (ky*k + kx)*Cin + c, zero padded up to Kpad; OH = (H + 2*pad - k)/stride + 1. */
int paif_im2col_fwd(const float* x, float* col, int B, int H, int W, int Cin, int k, int stride, int pad, int Kpad,
                    paif_stream_t stream);
/* conv weight [Cout,Cin,k,k] -> GEMM weight [Cout,Kpad] in the im2col column order. */
int paif_pack_conv_gemm_weight(const float* w, float* out, int Cout, int Cin, int k, int Kpad, paif_stream_t stream);

/* Mlp: DWConv (3x3 depthwise, bias) + GELU on NHWC tokens (core/mix_transformer.py:48-49,376-387).
 * x,y [B,H,W,C]; w [C,1,3,3]; bias [C]; C % 4 == 0. */
int paif_dwconv3_bias_gelu_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int C,
                               paif_stream_t stream);

/* Attention (core/mix_transformer.py:93-115) after the q / kv linears: q [B,N,C], kv [B,Nk,2C]
 * (k = channels [0,C), v = [C,2C), head hd at offset hd*(C/heads)), out [B,N,C];
 * softmax(q k^T * (C/heads)^-0.5) v per head, fused, K and V resident in LDS.
 * lse (optional, [B,heads,N]): log-sum-exp of the scaled scores per query, saved for the backward pass.
 * PAIF_ENOSUP when Nk*(C/heads)*8 B > 160 KiB or the head dim is not 32/64. */
int paif_sr_attention_fwd(const float* q, const float* kv, float* out, float* lse, int B, int N, int Nk, int C, int heads,
                          paif_stream_t stream);
/* Same contract, split-bf16 products on the bf16 MFMA (fp32 accumulate and softmax).  precision PAIF_CONV_BF16X3: operands as two bf16
 * pieces, hi*hi + hi*lo + lo*hi (~1e-5 relative on the products, 5.3x less matrix-pipe time than the exact-fp32 kernel);
 * PAIF_CONV_BF16X6: three pieces, six products (2^-25: fp32-level -- the attack loops' arithmetic, 2.7x less matrix-pipe time).  K and
 * V^T are staged in LDS in chunks of keys (the online softmax carries over chunks): two pieces x 300 keys x 64 dims fit at once, three
 * pieces take chunks of 160 keys; any Nk.  paif_sr_attention_bf16x3_fwd = precision PAIF_CONV_BF16X3. */
int paif_sr_attention_split_fwd(const float* q, const float* kv, float* out, float* lse, int B, int N, int Nk, int C, int heads,
                                int precision, paif_stream_t stream);
int paif_sr_attention_bf16x3_fwd(const float* q, const float* kv, float* out, float* lse, int B, int N, int Nk, int C, int heads,
                          paif_stream_t stream);

/* F.interpolate(mode='bilinear', align_corners=False) of NHWC x [B,IH,IW,C] written into channels
 * [coff, coff+C) of out [B,OH,OW,ldo] -- the head's upsample + torch.cat (core/segformer_head.py:66-77). */
int paif_resize_bilinear_into_fwd(const float* x, float* out, int B, int IH, int IW, int C, int OH, int OW, int ldo,
                                  int coff, paif_stream_t stream);

/* layout changes at the module boundary (3-channel input, 9-channel logits): [B,HW,C] <-> [B,C,HW]. */
int paif_nhwc_to_nchw_fwd(const float* x, float* y, int B, int HW, int C, paif_stream_t stream);
int paif_nchw_to_nhwc_fwd(const float* x, float* y, int B, int HW, int C, paif_stream_t stream);
/* NCHW [B,C,HW] -> NHWC [B,HW,CP], channels [C,CP) zero (pads dlogits to the GEMM's K % 32 == 0). */
int paif_nchw_to_nhwc_pad_fwd(const float* x, float* y, int B, int HW, int C, int CP, paif_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Input-gradient (dgrad) entry points for the PGD inner loop (attack/attack.py:443-512: loss.backward()
 * then sign of delta.grad).  Only d(loss)/d(input) is produced; parameter gradients are not.
 * GEMM / dense-conv dgrads reuse paif_gemm_fwd / paif_conv2d_fwd with transposed weights.
 * ------------------------------------------------------------------------------------------- */

/* paif_gemm_fwd with a prologue on A: A'[m,k] = A[m,k] * (a_mask[m,k] > 0) * a_scale[k]  (a_mask / a_scale
 * may be NULL).  Backward of ReLU(BN(.)) in front of a dgrad GEMM (core/segformer_head.py:50-55). */
int paif_gemm_masked_fwd(const float* A, int lda, const float* a_mask, const float* a_scale, const float* W,
                         const float* scale, const float* shift, int act, const float* res, int ldres, float* C,
                         int ldc, int M, int N, int K, int precision, paif_stream_t stream);
/* w [N,K] -> wt [K,Npad] (zero padded): weight operand of the dgrad GEMM dA = dC . W. */
int paif_transpose_pad_fwd(const float* w, float* wt, int N, int K, int Npad, paif_stream_t stream);

/* LayerNorm backward w.r.t. x: dx = rstd*(g - mean(g) - xhat*mean(g*xhat)), g = dy*gamma; + add (optional). */
int paif_layernorm_bwd_input(const float* x, const float* gamma, const float* dy, const float* add, float* dx,
                             int M, int C, float eps, paif_stream_t stream);
/* backward of paif_dwconv3_bias_gelu_fwd w.r.t. x (tmp: scratch of x's size). */
int paif_dwconv3_bias_gelu_bwd_input(const float* x, const float* w, const float* bias, const float* dy, float* tmp,
                                     float* dx, int B, int H, int W, int C, paif_stream_t stream);
/* adjoint of paif_im2col_fwd: dx [B,H,W,Cin] from dcol [B*OH*OW,Kpad]. */
int paif_col2im_fwd(const float* dcol, float* dx, int B, int H, int W, int Cin, int k, int stride, int pad, int Kpad,
                    paif_stream_t stream);
/* adjoint of paif_resize_bilinear_into_fwd: dx [B,IH,IW,C] from channels [coff,coff+C) of dout [B,OH,OW,ldo]. */
int paif_resize_bilinear_adjoint_fwd(const float* dout, float* dx, int B, int IH, int IW, int C, int OH, int OW,
                                     int ldo, int coff, paif_stream_t stream);
/* backward of paif_sr_attention_fwd: dq [B,N,C], dkv [B,Nk,2C] from dout, the saved output o and lse.
 * delta [B,heads,N] scratch/out; dkv_partial: paif_sr_attention_bwd_chunks(B,N,heads) * B*Nk*2C floats. */
int paif_sr_attention_bwd_chunks(int B, int N, int heads);
int paif_sr_attention_bwd_input(const float* q, const float* kv, const float* o, const float* dout, const float* lse,
                                float* delta, float* dq, float* dkv, float* dkv_partial, int B, int N, int Nk, int C,
                                int heads, paif_stream_t stream);
/* Same, with the arithmetic of the matrix products selectable: precision 0 = exact fp32 MFMA (= paif_sr_attention_bwd_input),
 * 1 = split-bf16 (hi*hi + hi*lo + lo*hi on the bf16 MFMA, fp32 accumulate; softmax statistics, delta and the slab reduction stay fp32),
 * 3 = three bf16 pieces per operand, six products (fp32-level; keys in LDS chunks of 128 at head dim 64). */
int paif_sr_attention_bwd_input_p(const float* q, const float* kv, const float* o, const float* dout, const float* lse,
                                float* delta, float* dq, float* dkv, float* dkv_partial, int B, int N, int Nk, int C,
                                int heads, int precision, paif_stream_t stream);

/* Seg_loss on bilinearly upsampled logits (attack/attack.py:103-114,446-448; F.interpolate align_corners=False
 * + CrossEntropyLoss(ignore_index), mean over valid pixels).  logits NHWC [B,IH,IW,C]; label int64 [B,OH,OW].
 * fwd: partial = 2*paif_upsample_ce_blocks floats of scratch; loss_count[0] = loss, [1] = #valid pixels.
 * bwd: dfull [B,OH,OW,CP] = gscale[0] * (softmax - onehot) on valid pixels (CP >= C, CP % 4 == 0, zero padded);
 *      gscale = dloss/count on device; follow with paif_resize_bilinear_adjoint_fwd to get dlogits. */
int paif_upsample_ce_blocks(int B, int OH, int OW);
int paif_upsample_ce_fwd(const float* logits, const long long* label, float* partial, float* loss_count, int B, int IH,
                         int IW, int C, int OH, int OW, int ignore_index, paif_stream_t stream);
int paif_upsample_ce_bwd(const float* logits, const long long* label, const float* gscale, float* dfull, int B, int IH,
                         int IW, int C, int OH, int OW, int ignore_index, int CP, paif_stream_t stream);

/* The loss glue of the attack variants (attack/attack.py:447-499, the `attack_way` branches of attack_both :447-499, seg_pgd
 * :335-350, cos_pgd :393-399) on bilinearly upsampled logits, value and gradient w.r.t. the full-resolution logits:
 *   way 0  PGD     loss = CE(o, label)                                  (= paif_upsample_ce_*; also serves 'newPGD', whose factor
 *                                                                         cos/cos is exactly 1 with an identically zero gradient)
 *   way 1  segPGD  loss = w_true * CE(t*o, label) + w_false * CE((1-t)*o, label),  t = (max_c o == label)  [float vs integer, as the
 *                  reference compares them];  w_true = 1 - lambda, w_false = lambda, lambda = (i-1)/(2*iters)
 *   way 2  cosPGD  loss = cosine_similarity(max_c o, label over ALL pixels) * CE(o, label)
 * fwd: partial = 6 * paif_attack_loss_blocks floats of scratch; coef[8] (device) = {loss, #valid, CE, cos, a, bl, bp, nbad}; nbad = the
 *      number of labels outside [0, C) that are not ignore_index: such a pixel is dropped like an ignored one and never used as an
 *      index (torch's CrossEntropyLoss raises on it; the attack entry points raise when nbad != 0).
 * bwd: dfull [B,OH,OW,CP] = upstream * d loss / d o (CP >= C, CP % 4 == 0, zero padded), reads coef written by fwd; follow with
 *      paif_resize_bilinear_adjoint_fwd to get dlogits.  Deterministic (fixed-order reductions, no float atomics). */
int paif_attack_loss_blocks(int B, int OH, int OW);
int paif_attack_loss_fwd(const float* logits, const long long* label, float* partial, float* coef, int way, float w_true,
                         float w_false, int B, int IH, int IW, int C, int OH, int OW, int ignore_index, paif_stream_t stream);
int paif_attack_loss_bwd(const float* logits, const long long* label, const float* coef, float* dfull, int way, float w_true,
                         float w_false, float upstream, int B, int IH, int IW, int C, int OH, int OW, int ignore_index, int CP,
                         paif_stream_t stream);

/* ---- fusion-network dgrad helpers (dense-conv dgrads go through paif_conv2d_fwd with these weights and the
 * in_act 3/4/5, epi_dact hooks of paif_conv_desc) ---- */
/* forward w [Co,Ctot,k,k] -> dgrad weight w.r.t. source channels [coff,coff+cs): wt [cs,Co,k,k],
 * wt[n][c][ky][kx] = w[c][coff+n][k-1-ky][k-1-kx]  (pack it with paif_pack_conv_weight*, nsrc = 1). */
int paif_conv_weight_dgrad(const float* w, float* wt, int Co, int Ctot, int k, int coff, int cs, paif_stream_t stream);
/* Cell_Decom 1x1 [32,128,1,1] -> folded ordinary weight [32,96,1,1] over [x, LF1, LF2]. */
int paif_fold_decomp1x1_weight(const float* w, float* wf, paif_stream_t stream);
/* backward of paif_tail_fwd w.r.t. x (NHWC16). */
int paif_tail_bwd_input(const float* dfused, const float* fused, const float* z, const float* w, const float* prelu,
                        float* dt16, int B, int H, int W, paif_stream_t stream);
/* backward of paif_stem_fwd w.r.t. the 1-channel image (PReLU slope must be >= 0: sign(feat) = sign(pre-act)). */
int paif_stem_bwd_input(const float* dfeat, const float* feat, const float* w, const float* prelu, float* dimg,
                        int B, int H, int W, paif_stream_t stream);
/* depthwise dgrad: out = dwconv(dt; rotated w) * (aux > 0) + add   (aux, add optional). */
int paif_dwconv_bwd_input(const float* dt, const float* w, const float* aux, const float* add, float* out, int k, int dil,
                          int B, int H, int W, paif_stream_t stream);
/* backward of paif_eca_finish_fwd: d_o, d_r from dout; partial: B*paif_eca_bwd_blocks(H,W)*32 floats, coef: B*32. */
int paif_eca_bwd_blocks(int H, int W);
int paif_eca_bwd_input(const float* dout, const float* u, const float* o, const float* gate, const float* w1d, int k,
                       const float* prelu, float* partial, float* coef, float* d_o, float* d_r, int B, int H, int W,
                       paif_stream_t stream);
/* backward of paif_channel_pool2_fwd + paif_spa_blend_fwd: d_ir, d_vis (+ optional add_*); dpre: [B,H,W] scratch. */
int paif_spa_blend_bwd_input(const float* dagg, const float* w, const float* ir, const float* vis, const float* s,
                             const float* add_ir, const float* add_vis, float* dpre, float* d_ir, float* d_vis,
                             int B, int H, int W, paif_stream_t stream);
/* backward of the guided-filter pair (+ get_residue): dlf [2][B,H,W,32] -> dy [B,H,W,32] (guide gradient routed to
 * the arg-max/arg-min channels; + optional add).  Scratch: gstat = paif_guided_filter_fused_workspace_floats(B,H,W) floats
 * (the forward's per-pixel guide statistics), t_my/t_mgy [B,H,W,32], t_g [B,H,W,4].  Round 6: both eps in one stage-1
 * launch, 48-column strips walked as runs of rows (csrc/gf_backward.hip); environment PAIF_GF_BWD=v1 selects the round-1
 * kernels (csrc/fusion_backward.hip), which also take the sizes whose row offsets do not fit 32 bits. */
int paif_guided_filter_bwd_input(const float* guide, const float* y, const float* ab, const float* dlf, float eps0,
                                 float eps1, const float* add, float* gstat, float* t_my, float* t_mgy, float* t_g,
                                 float* dy, int B, int H, int W, paif_stream_t stream);
/* the same reverse pass (autograd of core/model_fusion_auto.py:517-535) over paif_guided_filter_taped_fwd's tape (mc, and the
 * statistics workspace it filled) */
int paif_guided_filter_bwd_input_mc(const float* guide, const float* y, const float* mc, const float* stats, const float* dlf,
                                    const float* add, float* t_my, float* t_mgy, float* t_g, float* dy, int B, int H, int W,
                                    paif_stream_t stream);

/* ---- glue backward + PGD update ---- */
/* backward of paif_recompose_clamp_fwd + paif_minmax_normalize_fwd: dseg NCHW [B,3,H,W] -> dfused [B,H,W]
 * (+ optional dfused_direct), dcrcb [B,2,H,W].  minmax: the 2 floats paif_minmax_normalize_fwd wrote.
 * torch semantics: clamp passes gradient where 0 <= r <= 1; min/max gradients are spread evenly over equal elements.
 * partial: 4 * paif_glue_bwd_blocks floats. */
int paif_glue_bwd_blocks(int B, int H, int W);
int paif_glue_bwd_input(const float* dseg, const float* fused, const float* ycc, const float* minmax,
                        const float* dfused_direct, float* partial, float* dfused, float* dcrcb, int B, int H, int W,
                        paif_stream_t stream);
/* forward_object's extra step on the fused plane, core/model_fusion_auto.py:743-751 (Network_MM_CompModel) / :1074-1082
 * (Network_MM_Searched): clamp to [0,1] (two torch.where), then (f - min f) / (max f - min f) with BATCH-GLOBAL min / max.
 * x, out: n = B*H*W floats (in place allowed); partial: 2 * paif_plane_minmax_blocks(n) floats of scratch for the forward,
 * 4 * paif_plane_minmax_blocks(n) for the backward; minmax_out: the 2 floats (min, max of the clamped plane) the backward needs.
 * Backward = torch's: the clamp passes gradient where 0 <= x <= 1; the min / max gradients are spread evenly over equal elements. */
int paif_plane_minmax_blocks(size_t n);
int paif_plane_clamp_minmax_fwd(const float* x, float* out, float* partial, float* minmax_out, size_t n, paif_stream_t stream);
int paif_plane_clamp_minmax_bwd_input(const float* dout, const float* x, const float* minmax, float* partial, float* dx, size_t n,
                                      paif_stream_t stream);
/* Stand-alone eca_layer.forward (operations_m.py:353-367: AdaptiveAvgPool2d(1) of the map itself): per-image channel sums of
 * an NHWC [B,H,W,32] map as `chunks` partial sums per image, partial [B][chunks][32] -- with chunks = paif_conv2d_blocks(1,H,W) the
 * layout paif_eca_finish_fwd reduces (inside ECABasicBlock the conv epilogue produces them). */
int paif_channel_sum_chunks_fwd(const float* x, float* partial, int chunks, int B, int H, int W, paif_stream_t stream);
/* backward of paif_rgb2ycrcb_fwd: (dY [B,H,W], dcrcb [B,2,H,W]) -> dvis NCHW [B,3,H,W]. */
int paif_rgb2ycrcb_bwd_input(const float* dY, const float* dcrcb, float* dvis, int B, int H, int W, paif_stream_t stream);
/* attack/attack.py:504-512: delta <- clamp(clamp(delta + alpha*sign(grad_sum), -eps, eps), 0 - X, 1 - X), in place. */
int paif_pgd_step(float* delta, const float* grad_sum, const float* X, float alpha, float eps, size_t n,
                  paif_stream_t stream);
/* y += a*x (the never-zeroed delta.grad of the reference's attack loop accumulates, attack/attack.py:501). */
int paif_axpy(float* y, const float* x, float a, size_t n, paif_stream_t stream);

/* ---- evaluation harness (test_original.py:180,206-211; robust_test.py:181-212) ---- */
/* F.interpolate(bilinear, align_corners=False) to [OH,OW] + argmax over classes: logits NHWC [B,IH,IW,C] -> pred int64 [B,OH,OW]. */
int paif_upsample_argmax_fwd(const float* logits, long long* pred, int B, int IH, int IW, int C, int OH, int OW,
                             paif_stream_t stream);
/* conf[l*ncls + p] += #pixels with label l and prediction p (uint64, pairs outside [0,ncls) dropped) -- accumulates. */
int paif_confusion_matrix_accum(const long long* label, const long long* pred, unsigned long long* conf, size_t n, int ncls,
                                paif_stream_t stream);

/* ---- SPAattention (operations_m.py:148-204; in the search space, not in the shipped genotype) ---- */
/* comp = (max_c o, mean_c o); s = sigmoid(conv kxk 2->1 (comp)); out = PReLU(o*s + r).  w [1,2,k,k]; comp [B,H,W,2] scratch;
 * s_out [B,H,W], u_out (pre-activation) optional, saved for the backward pass. */
int paif_spa1_fwd(const float* o, const float* r, const float* w, int k, const float* prelu, float* comp, float* s_out,
                  float* u_out, float* out, int B, int H, int W, paif_stream_t stream);
int paif_spa1_bwd_input(const float* dout, const float* u, const float* o, const float* s, const float* w, int k,
                        const float* prelu, float* dpre, float* d_o, float* d_r, int B, int H, int W, paif_stream_t stream);

/* ---- training step, first kernel: dense-conv weight gradient (exact fp32 MFMA) --------------------------------------------
 * dW[cout=32][nsrc*32][kh][kh] (PyTorch layout) of the forward conv of paif_conv2d_fwd over the virtual concat of `src`:
 *   dAcc = dout * alpha * act'(z) * scale[cout]   (act 0 none | 1 PReLU(slope) | 2 ReLU; z = the saved pre-activation)
 *   dW[co][s*32+ci][ky][kx] = sum_px dAcc[px][co] * src_s[px + tap][ci]   (zero padding, dilation 1 or 2)
 * workspace: paif_conv2d_wgrad_workspace_floats(nsrc, kh, B, H) floats (per-workgroup slabs, summed in block order). */
size_t paif_conv2d_wgrad_workspace_floats(int nsrc, int kh, int B, int H);
/* dout always has 32 channels; only the first `cout` rows of dW are produced (stem_out.0 is 32 -> 16: its dout is zero padded).
 * accumulate != 0: dW += (gradients accumulate like loss.backward()); 0: dW = . */
int paif_conv2d_wgrad(const float* const* src, int nsrc, const float* dout, const float* z, const float* scale,
                      const float* prelu, int act, float alpha, int kh, int dil, float* workspace, float* dw, int cout,
                      int accumulate, int B, int H, int W, paif_stream_t stream);

/* Linear-layer weight / bias gradient (exact fp32 MFMA): dw[N][K] = dy^T x, db[N] = column sums of dy (db may be NULL).
 * dy [M, >=N] row stride lddy, x [M, >=K] row stride ldx.  splits = paif_gemm_wgrad_splits(M,N,K) token slices;
 * workspace: splits * (N*K + N) floats; slices are summed in order by a second pass (deterministic). */
int paif_gemm_wgrad_splits(int M, int N, int K);
int paif_gemm_wgrad(const float* dy, int lddy, const float* x, int ldx, float* dw, float* db, int M, int N, int K, int splits,
                    float* workspace, int accumulate, paif_stream_t stream);
/* Same with the arithmetic selectable: precision PAIF_CONV_F32 = paif_gemm_wgrad (exact fp32 MFMA); PAIF_CONV_F16X3 = fp16 pairs (dY and x
 * as two IEEE fp16 pieces each, three fp16 MFMAs per 16 tokens instead of eight fp32 ones), dY multiplied by the exact power of two gscale
 * before the split and the result by its inverse: |dY| * gscale and |x| must stay below 65504. */
int paif_gemm_wgrad_p(const float* dy, int lddy, const float* x, int ldx, float* dw, float* db, int M, int N, int K, int splits,
                      float* workspace, int accumulate, int precision, float gscale, paif_stream_t stream);

/* LayerNorm affine gradients: dgamma[c] = sum_rows dy * xhat, dbeta[c] = sum_rows dy (x, dy [M, C] dense).
 * workspace: paif_layernorm_wgrad_blocks(M) * 2 * C floats (per-workgroup partials, summed in block order). */
int paif_layernorm_wgrad_blocks(int M);
int paif_layernorm_wgrad(const float* x, const float* dy, float* dgamma, float* dbeta, float* workspace, int M, int C,
                         float eps, int accumulate, paif_stream_t stream);

/* ---- training-API losses (forward values; core/loss.py:490-502, pytorch_ssim/__init__.py:8-43) --------------------------
 * x, y: [B,1,H,W] planes.  window1d: the 11 fp32 Gaussian weights (sigma 1.5, normalised) as the reference builds them.
 * partial[2*blocks]: per-workgroup scratch; means[2] = (mean of the SSIM map, mean of |y - x|), reduced on the device in
 * block order (double). */
/* ---- image-space attack losses (attack/attack.py:75-100, 132-133, 216-218) ---- */
/* out[b][c] = x[b][c] * scale[c] + shift[c] on NCHW planes (shift may be NULL): trans_format = this on the SegFormer-normalised image
 * ((s * sd + mean) / 255 = the min-max normalised RGB), and its adjoint. */
int paif_channel_affine_nchw_fwd(const float* x, const float* scale, const float* shift, float* out, int B, int C, int H, int W,
                                 paif_stream_t stream);
/* nn.MSELoss (kind 0) / nn.L1Loss (kind 1), "mean" reduction, of a [B,C,H,W] against target [B,Ct,H,W], Ct = C or 1 (broadcast over the
 * channels, as torch does).  partial: paif_image_loss_blocks() floats; loss: 1 float.  bwd: da = g * d loss / d a. */
int paif_image_loss_blocks(void);
int paif_image_loss_fwd(const float* a, const float* target, int kind, int B, int C, int Ct, int H, int W, float* partial, float* loss,
                        paif_stream_t stream);
int paif_image_loss_bwd(const float* a, const float* target, int kind, int B, int C, int Ct, int H, int W, float g, float* da,
                        paif_stream_t stream);
int paif_ssim_l1_blocks(int B, int H, int W);
int paif_ssim_l1_fwd(const float* x, const float* y, const float* window1d, float* partial, float* means, int B, int H, int W,
                     paif_stream_t stream);
/* d/dx of  k[0] * sum|y - x| + k[1] * sum(1 - SSIM_map)  (k: 2 device floats = upstream gradient times the loss weights
 * over B*H*W);  abc_scratch[3*B*H*W] floats (per-pixel partials of the SSIM map).  dx [B,1,H,W]. */
int paif_ssim_l1_bwd_input(const float* x, const float* y, const float* window1d, const float* k, float* abc_scratch,
                           float* dx, int B, int H, int W, paif_stream_t stream);

/* ---- SegFormerHead with linear_fuse folded in front of the upsampling (core/segformer_head.py:63-80) ------------------------------
 * conv1x1(cat[up y4, up y3, up y2, y1]) = up(W4 y4) + up(W3 y3) + up(W2 y2) + W1 y1 (both maps are linear): z_i = W_i y_i are
 * GEMMs at each stage's own resolution; this kernel forms relu((z1 + up z2 + up z3 + up z4) * scale + shift) at 1/4 resolution.
 * z1..z4 NHWC [B,h_i,w_i,C]; hw = HOST array {h1,w1,h2,w2,h3,w3,h4,w4}; scale/shift = the folded eval-mode BatchNorm. */
int paif_head_sum_fwd(const float* z1, const float* z2, const float* z3, const float* z4, const int* hw, const float* scale,
                      const float* shift, float* out, int B, int C, paif_stream_t stream);
/* its backward through BatchNorm + ReLU: out = dx * (x > 0) * scale[c]   (x = the forward's output). */
int paif_relu_mask_scale_fwd(const float* dx, const float* x, const float* scale, float* out, size_t M, int C, paif_stream_t stream);

/* ---- input pipeline, device side (TaskFusion_dataset2.py:57-70,85-98; csrc/io_kernels.hip) ----------------------------------
 * src uint8 [B][HW][C] (decoded image bytes, HWC) -> dst float32 [B][C][HW] = src / 255 (fp32 division, as numpy does). */
int paif_u8_to_planes_fwd(const unsigned char* src, float* dst, int B, int HW, int C, paif_stream_t stream);
/* label bytes -> int64 */
int paif_u8_to_i64_fwd(const unsigned char* src, long long* dst, size_t n, paif_stream_t stream);

/* =============================================================================================
 * Training step (BASELINE configs[4]; csrc/train_kernels.hip).  Every gradient entry point ACCUMULATES
 * into its destination (like loss.backward() into .grad); reductions are two-pass in a fixed order (deterministic).
 * "rows" = NHWC pixels or tokens, M of them, C channels (C % 4 == 0).
 * ============================================================================================= */

/* NHWC [B,HW,ldx] channels [0,C) -> NCHW [B,C,HW] (gradient of the 9-class logits leaving the fused CE kernel). */
int paif_nhwc_slice_to_nchw_fwd(const float* x, float* y, int B, int HW, int C, int ldx, paif_stream_t stream);
/* y[M,Cd] = x[M,Cs] zero padded on the channel axis (the 16-channel gradient of stem_out.0 entering the dense-conv wgrad). */
int paif_pad_channels_fwd(const float* x, float* y, size_t M, int Cs, int Cd, paif_stream_t stream);
/* Cell_Decom.decomposition's return value (core/model_fusion_auto.py:522-535): x [M,32], lf [2][M,32] (both eps) ->
 * lfcat = cat(LF0, LF1) [M,64], hfcat = cat(x-LF0, x-LF1) [M,64]. */
int paif_decomp_cat_fwd(const float* x, const float* lf, float* lfcat, float* hfcat, size_t M, paif_stream_t stream);
/* floats of per-workgroup partials a row reduction with `nacc` float4-accumulator planes needs (0 = unsupported C). */
size_t paif_row_reduce_workspace_floats(int M, int C, int nacc);

/* Train-mode nn.BatchNorm2d statistics (operations_m.py:458-459,503; mmcv ConvModule core/segformer_head.py:50-55) over x [M,C]:
 * mean, invstd = 1/sqrt(biased var + eps), scale = gamma*invstd, shift = beta - mean*scale (gamma/beta NULL -> 1/0);
 * running_mean/var (optional) updated in place with `momentum` (unbiased variance), as torch does.
 * workspace: 2 * paif_row_reduce_workspace_floats(M, C, 2) floats (double partials). */
int paif_bn_stats_fwd(const float* x, int M, int C, const float* gamma, const float* beta, float eps, float momentum,
                      float* running_mean, float* running_var, float* mean, float* invstd, float* scale, float* shift,
                      float* workspace, paif_stream_t stream);
/* out = act(x*scale[c] + shift[c]) + res0 + res1 (act 0 none | 1 PReLU | 2 ReLU; z_out: the pre-activation, optional). */
int paif_affine_act_res_fwd(const float* x, const float* scale, const float* shift, int act, const float* prelu, const float* res0,
                            const float* res1, float* out, float* z_out, size_t M, int C, paif_stream_t stream);
/* Backward of y = act(BN_train(x)):  z = x*scale+shift, dz = g*act'(z), xhat = (x-mean)*invstd,
 *   dbeta += sum dz, dgamma += sum dz*xhat, dslope += sum g*z over z<0 (PReLU), dx = scale*(dz - mean(dz) - xhat*mean(dz*xhat)).
 * training == 0 (eval-mode BN, running statistics from paif_bn_eval_stats): dx = scale*dz, no batch-mean terms.
 * sums: 2*C floats scratch; workspace: paif_row_reduce_workspace_floats(M, C, 3). dgamma/dbeta/dslope may be NULL. */
int paif_bn_act_bwd(const float* g, const float* x, const float* scale, const float* shift, const float* mean, const float* invstd,
                    int act, const float* prelu, float* dx, float* dgamma, float* dbeta, float* dslope, float* sums, float* workspace,
                    int training, int M, int C, paif_stream_t stream);
/* eval-mode nn.BatchNorm2d as the same statistics: mean = running_mean, invstd = 1/sqrt(running_var + eps), scale, shift. */
int paif_bn_eval_stats(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps, int C,
                       float* mean, float* invstd, float* scale, float* shift, paif_stream_t stream);
/* PReLU on an element stream: dx = t*P'(r) + add (dx, add optional), dslope[0] += factor * sum t*r over r<0
 * (nn.PReLU weight gradient, operations_m.py:441,460,392; workspace: 2048 floats). */
int paif_prelu_bwd(const float* t, const float* r, const float* add, const float* prelu, float factor, float* dx, float* dslope,
                   float* workspace, size_t n, paif_stream_t stream);
/* tail (stem_out.2 + tanh, core/model_fusion_auto.py:618-619,634): dz = dfused*(1-fused^2)*P'(z); dslope += sum dfused*(1-fused^2)*z, z<0. */
int paif_tail_dz(const float* dfused, const float* fused, const float* z, const float* prelu, float* dz, float* dslope, float* workspace,
                 size_t n, paif_stream_t stream);
/* out[c] += sum_rows x[row*ld + c], c < C (conv bias gradient, Cell_Decom.conv1x1_*.bias :501-502). workspace: row_reduce(M,C,1). */
int paif_colsum(const float* x, int ld, float* out, float* workspace, int M, int C, paif_stream_t stream);
/* Depthwise conv weight/bias gradient: dw[C][k][k] += sum dy[px][c]*in(x)[px+tap][c], db[c] += sum dy (db optional);
 * in = ReLU when in_relu (DilConv, operations_m.py:496-498); MiT Mlp.dwconv (core/mix_transformer.py:376-387).
 * k in {3,5}, dil in {1,2}; workspace: row_reduce(B*H*W, C, k*k+1). */
int paif_dwconv_wgrad(const float* x, const float* dy, float* dw, float* db, float* workspace, int k, int dil, int in_relu, int B, int H,
                      int W, int C, paif_stream_t stream);
/* stem_1/stem_2 (Conv2d(1,32,3)+PReLU, core/model_fusion_auto.py:607-614): dw[32][9] +=, dslope +=; the pre-activation is
 * recomputed from the image.  img as in paif_stem_fwd.  workspace: row_reduce(B*H*W, 32, 10). */
int paif_stem_wgrad(const float* img, size_t img_bstride, const float* dfeat, const float* w, const float* prelu, float* dw, float* dslope,
                    float* workspace, int B, int H, int W, paif_stream_t stream);
/* Weight gradient of a Cm -> 1 "same" conv from its 1-channel output gradient s [B,H,W] and input m [B,H,W,Cm]:
 * dw[Cm][k][k] += sum s[px]*m[px+tap][c] (stem_out.1: Cm 16, k 3, :617; spatial_attn_layer_M: Cm 4, k 5, :1361).
 * workspace: row_reduce(B*H*W, Cm, k*k). */
int paif_corr1_wgrad(const float* s, const float* m, float* dw, float* workspace, int Cm, int k, int B, int H, int W, paif_stream_t stream);
/* eca_layer's Conv1d weight gradient (operations_m.py:353-367): dw[k] += sum_b sum_c dpre[b][c]*mean[b][c+j-pad].
 * pool_partial: the forward conv's per-tile channel sums; dgate_partial [B][blocks][32]: paif_eca_bwd_input's partials;
 * workspace: 9*B floats. */
int paif_eca_wgrad(const float* pool_partial, const float* dgate_partial, int dgate_blocks_per_img, const float* gate, int k, float* dw,
                   float* workspace, int B, int H, int W, paif_stream_t stream);
/* Cell_Decom 1x1: G [32][96] = gradient of the folded weight over [x, LF1, LF2] -> dw [32][128] += [G1, G2, Gx-G1, Gx-G2]. */
int paif_unfold_decomp1x1_wgrad(const float* G, float* dw, paif_stream_t stream);
/* conv-as-GEMM weight gradient back to the PyTorch layout: dw[Cout][Cin][k][k] += dwp[Cout][tap*Cin + c] (row stride Kpad). */
int paif_unpack_conv_gemm_wgrad(const float* dwp, float* dw, int Cout, int Cin, int k, int Kpad, paif_stream_t stream);
/* Keep masks of DropPath (timm; core/mix_transformer.py:126) / Dropout2d (core/segformer_head.py:47,79):
 * out[i] = u_i >= p ? 1/(1-p) : 0 with the counter-based uniform u_i = (splitmix64(seed*0x100000001B3 + offset + i) >> 11) * 2^-53. */
int paif_keep_mask(float* out, int n, unsigned long long seed, unsigned long long offset, float p, paif_stream_t stream);
/* out[b][r][c] = x[b][r][c] * s[b] (per_channel: s[b][c]) + res (res optional). */
int paif_rowscale_add_fwd(const float* x, const float* s, const float* res, float* out, int B, size_t rows_per_b, int C, int per_channel,
                          paif_stream_t stream);
/* Multi-tensor AdamW over a flat arena (utils/optimizer.py:3-33 = torch.optim.AdamW, eps 1e-8, no amsgrad).  chunk_group[i]
 * (device bytes): parameter group of arena chunk i (1024 floats), >= 8 = skip.  group_decay[g] = 1 - lr_g*wd_g and
 * group_step_size[g] = lr_g / (1 - beta1^t) are HOST arrays (the schedule changes them every step); bc2_sqrt = sqrt(1 - beta2^t). */
int paif_adamw_step(float* p, const float* g, float* m, float* v, const unsigned char* chunk_group, size_t nchunks, int ngroups,
                    const float* group_decay, const float* group_step_size, float one_minus_beta1, float beta2, float one_minus_beta2,
                    float bc2_sqrt, float eps, paif_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PAIF_HIP_H */
