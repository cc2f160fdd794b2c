/*
 * sc2_bottleneck.h -- C-ABI of libsc2amd.so, the MI355X (gfx950) implementation of the
 * supervised-compression bottleneck hot path of sc2bench.
 *
 * The reference has no FFI of its own: the path sits behind Python nn.Modules
 * (sc2bench/models/layer.py) that call CompressAI (Python + two pybind11
 * extensions) and torch's conv kernels.  Each entry point below names the
 * reference interface it replaces (file:line under the sc2bench tree, or the
 * CompressAI symbol the reference calls at that line).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch types.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  All device
 *    work is enqueued on it; nothing synchronises, nothing allocates.
 *  - every buffer is caller-allocated device memory unless marked HOST.
 *  - return value: 0 = success, negative = error (SC2_ERR_*).  No exceptions cross
 *    the boundary.  sc2_last_error() returns a thread-local message.
 *  - stateless and re-entrant.
 */
#ifndef SC2_BOTTLENECK_H
#define SC2_BOTTLENECK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SC2_OK 0
#define SC2_ERR_INVALID_ARG (-1)
#define SC2_ERR_UNSUPPORTED (-2)
#define SC2_ERR_DOMAIN (-3)      /* pmf has a negative / non-finite entry (std::domain_error upstream) */
#define SC2_ERR_ZERO_PMF (-4)    /* pmf sums to zero */
#define SC2_ERR_LAUNCH (-5)      /* hip launch error */
#define SC2_ERR_NO_DEVICE (-6)
#define SC2_ERR_INTERNAL (-7)

/* ABI version: bumped on any signature change. */
#define SC2_ABI_VERSION 51
int sc2_abi_version(void);

/* ------------------------------------------------------------------------------------------ */
/* Dispatch policy.  Which kernel serves a launch is a function of the call's arguments and of  */
/* THIS struct -- the library reads no environment variable (round 5; rounds 1 - 4 steered ~27    */
/* choices through getenv).  The defaults are the measured choices; every   */
/* field exists for an A/B measurement or a diagnostic, and `tools/env_policy.py` is the only     */
/* place that maps SC2_* environment variables onto it.  The policy is process-wide             */
/* configuration: set it before issuing launches (sc2_policy_set copies the struct; concurrent    */
/* launches from other threads see either the old or the new value of each field).  The entry     */
/* points themselves stay re-entrant: no call reads or writes anything else that outlives it,     */
/* except per-device resource caches (unit counters, function attributes).                        */
/* ------------------------------------------------------------------------------------------ */
typedef struct sc2_policy {
    int32_t struct_bytes;        /* sizeof(sc2_policy) of the caller's header (sc2_policy_default sets it; _set checks it) */
    /* sc2_conv2d_fwd */
    int32_t conv_patch3;         /* window staging of stride-1 slab-major layers: 0 off, 1 = 3x3 on the 128-wide tile (default), 256 = also the 256-wide tile, 2 = also the 2x2 decoder layers */
    int32_t conv_s2;             /* static 3x3 stride-2 tile: 0 off, 1 (default) 256-wide where it wins, 128 / 256 force */
    int32_t conv_persist;        /* persistent 8-wave decoder tile (conv_dec_persist): 0 off, 1, 2, 3 (default: one phase per slab) */
    int32_t conv_half;           /* 1: half-width static decoder tiles (A/B; default 0) */
    int32_t conv_big4;           /* 1: the 4-wave register-tile kernel for the 256-channel layers (A/B; default 0) */
    int32_t conv_no_big;         /* 1: never the 8-wave 256-row tile */
    int32_t conv_force_big;      /* 1: the 8-wave tile wherever Cout allows (tests) */
    int32_t conv_no_epx;         /* 1: no epilogue-operand prefetch instantiations */
    int32_t conv_debug;          /* development: bit 0 skips the store epilogue, bit 1 the K loop (results garbage) */
    int32_t conv_chunk;          /* conv_dec_persist: tiles per claim (0 = default 2) */
    /* window-plane kernels */
    int32_t w2_run;              /* conv2x2_win: tiles per workgroup run (0 = default: min(share, 2)) */
    int32_t win_half;            /* conv3x3_win: half tiles, two workgroups per CU (default 1) */
    int32_t win_dbg;             /* conv3x3_win 14 x 14: timing experiments (results garbage; -DSC2_EXPERIMENTS builds only; default 0) */
    int32_t win_stamps;          /* conv3x3_win: 1 = per-workgroup wall-clock summary on stderr (diagnostic) */
    int32_t p1_half;             /* conv1x1_win: 112-pixel tiles (default 0) */
    int32_t p1_nbuf;             /* conv1x1_win: LDS ring depth 2 / 4 (0 = by shape) */
    int32_t pair_alt;            /* conv1x1_pair: alternate the traversal direction between launches (default 1) */
    /* reference-precision encoder, decoder head, training */
    int32_t f32_persist0;        /* conv_f32: the persistent first stage (default 1) */
    int32_t dec_stagger;         /* conv2x2_gdn512: stagger workgroup starts (A/B; default 0) */
    int32_t wgrad_wgs;           /* conv_wgrad: target workgroup count (0 = default 1024) */
    /* range coder */
    int32_t rans_lds_pad_kb;     /* small coder launches ask for this much LDS so that nothing shares their CU (default 159; 0 off) */
    int32_t rans_pad_waves;      /* ... launches of up to this many serial waves (default 16) */
    int32_t rans_ragged2;        /* four-lanes-per-stream decoder for per-symbol CDF rows (default 1) */
    int32_t rans_ragged2_waves;  /* ... waves per workgroup sharing one table copy: 0 (default) = 1 below 1 024 streams per launch, 2 from there; 1, 2, 4, 8 force */
    int32_t rans_lut8;           /* 1: one-lookup bucketed decode tables for implicit CDF rows (default 0: measured 6 % SLOWER than the two-lookup decoder) */
    int32_t rans_dq_lds;         /* 1: the dequantising last pass of a decode launch through LDS for every channel count (default 0: <= 32 channels transpose in registers, no LDS: fits beside the persistent kernels) */
    int32_t wgrad_ct;            /* conv_wgrad: 0 (default) = 256-channel tiles for layers with more than 128 output channels, 64-channel tiles for 64 or fewer; 128 = always the 128 x 128 tile (A/B) */
    int32_t reserved[6];
} sc2_policy;
void sc2_policy_default(sc2_policy *p);
int sc2_policy_set(const sc2_policy *p);      /* SC2_ERR_INVALID_ARG if p is NULL or struct_bytes != sizeof(sc2_policy) */
void sc2_policy_get(sc2_policy *p);
const char *sc2_last_error(void);
/* number of visible HIP devices (0 on a CPU-only box); never throws. */
int sc2_device_count(void);

/* ------------------------------------------------------------------------------------------ */
/* Layout conversion                                                                          */
/* ------------------------------------------------------------------------------------------ */
/* x: f32 NCHW [N,C,H,W]  ->  y: bf16 NHWC [N,H,W,Cpad] (channels c>=C zero-filled).
 * Replaces the implicit layout of the tensor handed to nn.Conv2d at layer.py:475. */
int sc2_nchw_f32_to_nhwc_bf16(const float *x, void *y, int N, int C, int H, int W, int Cpad, void *stream);
/* x: bf16 NHWC [N,H,W,C] -> y: f32 NCHW [N,C,H,W] (what a reference caller sees). */
int sc2_nhwc_bf16_to_nchw_f32(const void *x, float *y, int N, int C, int H, int W, void *stream);

/* AdaptiveAvgPool2d((1,1)) + flatten of a bf16 NHWC feature map [N, HW, C] (torchvision ResNet.avgpool,
 * sc2bench/models/backbone.py:250-252): mean over HW in f32 -> y_f32 [N, C] and / or y_bf16 [N, C] (either may be NULL). */
int sc2_avgpool_nhwc(const void *x, float *y_f32, void *y_bf16, int N, int HW, int C, void *stream);

/* nn.BatchNorm2d in TRAINING mode (batch statistics) on a bf16 NHWC map x [M = N H W][C], with the ReLU and the residual add of a
 * torchvision Bottleneck block folded in (stage 2 of the Entropic-Student recipe fine-tunes layer2 .. layer4 with their norm layers
 * training: configs/.../splitable_resnet50-fp-beta0.08_from_resnet50.yaml:231-295; sc2bench/models/backbone.py:235-254 runs the blocks).
 *   forward : y = relu?((x - mean) rstd gamma + beta (+ residual)); mean / biased variance over the M rows; running_mean / running_var
 *             (may both be NULL) take the momentum update with the UNBIASED variance, as torch; save_mean / save_rstd [C] for backward
 *   backward: dz = y ? dy * (y > 0) : dy (y = the forward's output when it applied the ReLU, else NULL); dgamma = sum dz xhat, dbeta =
 *             sum dz; dx = gamma rstd (dz - dbeta / M - xhat dgamma / M); dz (NULL, or bf16 like dx) = the residual operand's gradient
 *   gamma, beta, running_*, save_*, dgamma, dbeta : f32 [C];  ws : f32 scratch of sc2_bn_ws_floats(M, C) floats (per-workgroup partial
 *   sums + coefficient rows; either direction);  C % 8 == 0, C <= 2048 */
long long sc2_bn_ws_floats(long long M, int C);
int sc2_bn_train_fwd(const void *x, const void *residual, const float *gamma, const float *beta, float *running_mean, float *running_var,
                     float momentum, float eps, int relu, void *y, float *save_mean, float *save_rstd, float *ws, long long M, int C,
                     void *stream);
int sc2_bn_train_bwd(const void *dy, const void *x, const void *y, const float *gamma, const float *save_mean, const float *save_rstd,
                     void *dx, void *dz, float *dgamma, float *dbeta, float *ws, long long M, int C, void *stream);

/* nn.MaxPool2d (floor mode, no dilation) on a bf16 NHWC map: x [N,H,W,C] -> y [N,OH,OW,C], C % 8 == 0 (torchvision ResNet.maxpool
 * behind the stem, sc2bench/models/backbone.py:235-254 via torchvision's resnet forward; the teacher of the training step and the
 * input-compression classifier run it).  Bit-identical to torch's kernel: same update rule, NaN propagates. */
int sc2_maxpool_nhwc(const void *x, void *y, int N, int H, int W, int C, int KH, int KW, int stride_h, int stride_w, int pad_h, int pad_w,
                     void *stream);

/* Classifier on the pooled features: out[m][n] = sum_k a[m][k] w[n][k] + bias[n] (torchvision ResNet.fc behind the pool,
 * sc2bench/models/backbone.py:247-253).  K split over the four waves of a workgroup, operands straight from L2.
 *   a : bf16 [M,K] (sc2_avgpool_nhwc's y_bf16), K % 128 == 0;   w_frag : bf16 fragment blocks [Npad/16][K/32][64][8] of the
 *   [Npad,K] weight (rows >= N zero; hip.pack_weight_fragments);   bias : f32 [Npad];   out : f32 [M,Npad], Npad % 16 == 0. */
int sc2_fc_fwd(const void *a, const void *w_frag, const float *bias, float *out, int M, int K, int Npad, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* Implicit-GEMM convolution on the matrix cores (bf16 in, f32 accumulate)                    */
/* Replaces nn.Conv2d(bias=False) at layer.py:475-476,479-480,482-483 (encoder) and            */
/* 485-486,489-490,492-493 (decoder); with a_op/epilogue set, CompressAI GDN1.forward at       */
/* layer.py:478,481,488,491 (norm = conv2d(|x|, gamma 1x1, beta); y = x/norm or x*norm).       */
/* ------------------------------------------------------------------------------------------ */
enum sc2_conv_aop { SC2_AOP_NONE = 0, SC2_AOP_ABS = 1,
                    SC2_AOP_SQUARE = 2 /* x^2: the operand of CompressAI GDN (squared form), compressai.layers.GDN.forward */ };
enum sc2_conv_epilogue {
    SC2_EPI_NONE = 0,
    SC2_EPI_GDN = 1,  /* y = ep_x / (ep_beta[c] + acc)   (GDN1, inverse=False) */
    SC2_EPI_IGDN = 2, /* y = ep_x * (ep_beta[c] + acc)   (GDN1, inverse=True)  */
    SC2_EPI_BIAS = 3, /* y = acc + ep_beta[c]            (conv + bias / folded BN) */
    SC2_EPI_BIAS_RELU = 4,     /* y = relu(acc + ep_beta[c]) */
    SC2_EPI_BIAS_ADD_RELU = 5, /* y = relu(acc + ep_beta[c] + ep_x) (residual add) */
    /* conv FOLLOWED BY GDN1 in one launch where one tile holds every channel of a pixel: Cout in {32,48,64,96}, or
     * Cout == 256 on the 256-wide big-tile path (sc2_conv_fused_gdn_supported tells):
     * x = acc; y = x / (ep_beta + gamma |x|)  resp.  x * (ep_beta + gamma |x|).  `ep_x` carries the packed bf16
     * gamma matrix instead of an activation, in the layout sc2_conv_fused_gdn_supported reports. */
    SC2_EPI_FUSED_GDN = 6,
    SC2_EPI_FUSED_IGDN = 7,
    SC2_EPI_BIAS_LEAKY_RELU = 8, /* y = leaky_relu(acc + ep_beta[c], 0.01) (nn.LeakyReLU() default slope; h_a / h_s) */
    /* CompressAI GDN (squared form; bmshj2018_factorized g_a / g_s, reached from sc2bench/models/registry.py:73-80): with
     * a_op = SC2_AOP_SQUARE and the 1x1 gamma GEMM, acc = gamma x^2 */
    SC2_EPI_GDN2 = 9,  /* y = ep_x * rsqrt(ep_beta[c] + acc)   (GDN, inverse=False) */
    SC2_EPI_IGDN2 = 10, /* y = ep_x * sqrt(ep_beta[c] + acc)    (GDN, inverse=True)  */
    /* sc2_gdn1_bwd_gemm only (round 5): the two GEMMs of the GDN1 backward with its element-wise halves in their epilogues.
     * n = ep_beta[c] + acc (acc = gamma |x|, a_op ABS), g = the gradient at the GDN's output, x = its input: */
    SC2_EPI_GDN1_BWD_PRE = 11,   /* ep_x = g, ep_x2 = x:  y2 = g / n (direct term),  y = -y2 * x / n  (d_norm) */
    SC2_EPI_IGDN1_BWD_PRE = 12,  /* ep_x = g, ep_x2 = x:  y2 = g * n,                y = g * x         (d_norm) */
    SC2_EPI_GDN1_BWD_POST = 13   /* acc = gamma^T d_norm, ep_x = direct term, ep_x2 = x:  y = ep_x + sign(x) * acc  (= dL/dx) */
};
enum sc2_conv_out { SC2_OUT_BF16_NHWC = 0, SC2_OUT_F32_NCHW = 1, SC2_OUT_F32_NHWC = 2,
                    /* int32 NCHW symbols round_half_even(acc - ep_beta[c]) (ep_beta = the entropy bottleneck's medians,
                     * epilogue NONE): the last encoder conv followed by EntropyModel.quantize(.., 'symbols') in one
                     * launch (layer.py:482-483 + :506), bit-identical to sc2_conv2d_fwd(F32_NCHW) + sc2_eb_symbols */
                    SC2_OUT_I32_NCHW_SYM = 3 };
/* Order of the K axis of the packed weights (and of the kernel's walk over the input):
 *   TAP_MAJOR  k = (kh*KW + kw)*Cin + ci                       (default)
 *   SLAB_MAJOR k = ((ci/32)*KH*KW + kh*KW + kw)*32 + ci%32     (Cin % 32 == 0): the taps of one 32-channel slab are
 *              consecutive, so the overlapping pixels they re-read stay in L1/L2 (measured 4x fewer fabric reads on
 *              the 2x2 decoder convs). */
enum sc2_conv_k_order { SC2_K_TAP_MAJOR = 0, SC2_K_SLAB_MAJOR = 1,
                        /* flag, OR-ed in: w_packed is [Kpad/32][Cout_pad][32] (the B tile of one 32-deep k-slab is
                         * contiguous) instead of [Cout_pad][Kpad] */
                        SC2_K_B_TILE_MAJOR = 2,
                        /* flag, with SLAB_MAJOR only: w_packed is MFMA-fragment-major [Kpad/32][Cout_pad/16][64][8], entry
                         * (kt, jt, lane = fq*16 + frow, e) = W[jt*16 + frow][kt*32 + fq*8 + e]; taken by the LDS-patch
                         * kernel of the 5x5 stride-2 geometry (sc2_conv_patch_supported) */
                        SC2_K_B_FRAG_MAJOR = 4 };

typedef struct sc2_conv_desc {
    int32_t N, H, W, Cin;          /* input  : bf16 NHWC [N,H,W,Cin], Cin % 8 == 0              */
    int32_t Cout;                  /* output channels, Cout % 8 == 0                             */
    int32_t KH, KW;                /* filter taps                                                */
    int32_t stride_h, stride_w;
    int32_t pad_h, pad_w;
    int32_t OH, OW;                /* output spatial size (caller computes, library validates)   */
    int32_t a_op;                  /* enum sc2_conv_aop: transform applied to the input on load  */
    int32_t epilogue;              /* enum sc2_conv_epilogue                                     */
    int32_t out_format;            /* enum sc2_conv_out                                          */
    int32_t Kpad;                  /* row pitch (elements) of the packed weights, % 64 == 0      */
    int32_t Cout_pad;              /* rows of the packed weight matrix (>= Cout, % 128 == 0 or
                                      == tile width; see sc2_conv_weight_rows)                   */
    /* Output scatter, for the data gradient of a strided conv (one launch per stride-parity class).  out_H == 0:
     * dense output [N,OH,OW,Cout].  Otherwise OH/OW are taken as given (rows past the symmetric-padding formula
     * see implicit zeros) and output pixel (oh, ow) is written to (oh*out_stride_h + out_off_h,
     * ow*out_stride_w + out_off_w) of an NHWC tensor [N,out_H,out_W,Cout]; pixels outside it are dropped. */
    int32_t out_H, out_W, out_stride_h, out_stride_w, out_off_h, out_off_w;
    int32_t k_order;               /* enum sc2_conv_k_order */
    /* filter dilation (0 or 1 = none), sc2_conv2d_fwd only: tap (kh, kw) reads input pixel (oh*stride_h - pad_h + kh*dil_h, ..);
     * OH = (H + 2 pad_h - dil_h (KH - 1) - 1) / stride_h + 1.  The atrous layers of the dense-prediction models
     * (sc2bench/models/segmentation/deeplabv3.py ASPP rates 12 / 24 / 36; torchvision's `replace_stride_with_dilation` layer3 /
     * layer4): Cout > 96, plain epilogues (NONE / BIAS / BIAS_RELU / BIAS_ADD_RELU / GDN / IGDN), dense output. */
    int32_t dil_h, dil_w;
} sc2_conv_desc;

/* Rows the packed weight buffer must have for a given Cout (zero rows beyond Cout). */
int sc2_conv_weight_rows(int Cout);
/* Row pitch (elements) the packed weight buffer must have for K = KH*KW*Cin. */
int sc2_conv_weight_pitch(int K);

/* w_packed: bf16 [Cout_pad][Kpad], element (co, (kh*KW+kw)*Cin + ci); zero padded.
 * ep_x   : bf16 NHWC [N,OH,OW,Cout] for GDN/IGDN/ADD epilogues (NULL otherwise)
 * ep_beta: f32 [Cout] for GDN/IGDN/BIAS epilogues (NULL otherwise)
 * y      : per out_format. */
/* Non-zero if SC2_EPI_FUSED_GDN / _IGDN is available for this geometry (all of `d` filled in as for sc2_conv2d_fwd):
 * 1 = `ep_x` carries gamma as packed rows [sc2_conv_weight_rows(Cout)][sc2_conv_weight_pitch(Cout)] (tiles of
 * 32..96 channels); 2 = `ep_x` carries gamma as MFMA-fragment blocks [Cout/16][Cout/32][64][8], entry (jt, ks,
 * lane = fq*16 + frow, e) = gamma[jt*16 + frow][ks*32 + fq*8 + e] (the 256-wide 8-wave tile). */
int sc2_conv_fused_gdn_supported(const sc2_conv_desc *d);
/* 1 if this geometry runs on the LDS-resident-patch kernel (5x5 stride 2 pad 2, Cin 96 -> Cout 48, OW <= 64, bf16 NHWC
 * output: the second encoder conv, layer.py:479-480), which takes SC2_K_SLAB_MAJOR | SC2_K_B_FRAG_MAJOR weights. */
int sc2_conv_patch_supported(const sc2_conv_desc *d);
int sc2_conv2d_fwd(const sc2_conv_desc *d, const void *x, const void *w_packed, void *y,
                   const void *ep_x, const float *ep_beta, void *stream);

/* The two channel-mixing GEMMs of the GDN1 / inverse-GDN1 BACKWARD (CompressAI GDN1 under autograd, reached by loss.backward() for
 * sc2bench/models/layer.py:478,481,488,491) with the element-wise halves of that backward fused into their epilogues
 * (SC2_EPI_GDN1_BWD_PRE / SC2_EPI_IGDN1_BWD_PRE / SC2_EPI_GDN1_BWD_POST): a 1x1 "conv" (d: KH = KW = 1, stride 1, no padding,
 * Cin == Cout == C in {96, 256, 512, ...: packed rows 96 or a multiple of 128}, bf16 NHWC output) that reads TWO per-element
 * operands and, for the PRE forms, writes TWO outputs.  Replaces, per GDN layer, the norm GEMM + sc2_gdn_bwd_pre and the
 * gamma^T GEMM + sc2_gdn_bwd_post (nine passes over [pixels x C] tensors instead of fifteen).
 *   x : bf16 [M, C] the GEMM's operand (PRE: the GDN input, a_op ABS; POST: d_norm);  w_packed : gamma resp. gamma^T, packed as for
 *   sc2_conv2d_fwd;  ep_x, ep_x2 : bf16 [M, C];  ep_beta : f32 [C] (PRE; ignored by POST);  y, y2 : bf16 [M, C] (y2: PRE only). */
int sc2_gdn1_bwd_gemm(const sc2_conv_desc *d, const void *x, const void *w_packed, void *y, void *y2, const void *ep_x,
                      const void *ep_x2, const float *ep_beta, void *stream);

/* GDN1 / inverse GDN1 over C = 96, 256 or 512 channels with the whole channel row of a pixel tile resident in LDS (gdn512_rows.hip: 128-pixel
 * tiles, C = 256 / 512; gdn96_strips.hip: 32-pixel strips per wave, C = 96) -- the training-time forms of the first encoder normalisation
 * and the decoder's two (sc2bench/models/layer.py:476-477, 486-491; forward in train mode keeps its input
 * for the backward, which loss.backward() of script/task/image_classification.py:79 reaches):
 *   sc2_gdn1_rows_fwd : y = x * (beta + gamma |x|)  (inverse != 0)  or  x / (...);  x, y bf16 [M, C]
 *   sc2_gdn1_rows_bwd : given x and the gradient gy of y, BOTH GEMMs of the backward and its element-wise halves in one launch:
 *                       d_norm (bf16 [M, C]: its column sums are d_beta, d_norm^T |x| is d_gamma) and dx (bf16 [M, C]).
 *                       Same quantities as SC2_EPI_(I)GDN1_BWD_PRE + SC2_EPI_GDN1_BWD_POST of sc2_gdn1_bwd_gemm, the direct term kept
 *                       in f32 in the accumulators instead of rounded to bf16 in between; sign(0) = 0 as torch.abs's gradient.
 *                       d_beta (f32 [C] or NULL): the column sums of d_norm -- accumulated inside the launch for C = 96 / 256 (from
 *                       the f32 values, before their rounding to bf16), by sc2_colsum_bf16 behind it for C = 512.
 *   gamma_frag / gamma_t_frag : the effective gamma [C, C] and its transpose as MFMA fragments, bf16 [C/16][C/32][64][8]: entry
 *                       (jt, ks, lane = fq*16 + frow, e) = W[jt*16 + frow][ks*32 + fq*8 + e];  beta f32 [C];  M * C * 2 < 2 GB. */
int sc2_gdn1_rows_supported(int C);
int sc2_gdn1_rows_fwd(const void *x, const void *gamma_frag, const float *beta, void *y, long long M, int C, int inverse,
                      void *stream);
int sc2_gdn1_rows_bwd(const void *x, const void *gy, const void *gamma_frag, const void *gamma_t_frag, const float *beta,
                      void *d_norm, void *dx, float *d_beta, long long M, int C, int inverse, void *stream);
/* column sums of a bf16 [M, C] tensor in f32 (d_beta = sum over pixels of d_norm): out f32 [C], C % 8 == 0, C <= 2048. */
int sc2_colsum_bf16(const void *x, long long M, int C, float *out, void *stream);

/* Two consecutive 1x1 layers of the ResNet tail across a block boundary in ONE launch (torchvision Bottleneck blocks b and
 * b + 1 of layer2, eval mode, BatchNorm folded; sc2bench/models/backbone.py:235-254 runs them one after the other):
 *     h = relu(W3 o + b3 + identity)      conv3 + bn3 + residual + ReLU of block b        (K1 -> C)
 *     u = relu(W1 h + b1)                 conv1 + bn1 + ReLU of block b + 1               (C -> N2)
 * h is written once (it is the next block's identity) and feeds the second GEMM from LDS instead of being read again.
 *   o : bf16 [M][K1];  identity, h : bf16 [M][C];  u : bf16 [M][N2];  w3_frag : bf16 fragment-major [C/16][K1/32][64][8];
 *   w1_frag : bf16 fragment-major [N2/16][C/32][64][8] (hip.pack_weight_fragments);  b3 : f32 [C];  b1 : f32 [N2].
 * Supported: K1 = 128, C = 512, N2 = 128 (the block boundaries inside layer2 of ResNet-50) or 256 (layer2.3 -> layer3.0).  Bit-identical to sc2_conv1x1_stream_fwd (residual, relu)
 * followed by the 1x1 kernel of the next block. */
int sc2_conv1x1_pair_supported(int K1, int C, int N2);
int sc2_conv1x1_pair_fwd(const void *o, const void *w3_frag, const float *b3, const void *identity, void *h,
                         const void *w1_frag, const float *b1, void *u, long long M, int K1, int C, int N2, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* Reference-precision analysis transform: the same convolution / GDN1 with f32 OPERANDS on the   */
/* f32 matrix cores (v_mfma_f32_16x16x4_f32: bit for bit a k-ordered f32 fma chain), for callers   */
/* that need the symbols -- hence the byte streams -- the reference's f32 CPU path produces        */
/* (nn.Conv2d + compressai GDN1 in f32, sc2bench/models/layer.py:475-483, quantised at :506).      */
/* 1/16 of the bf16 matrix rate; `FPBasedResNetBottleneck.set_encoder_precision('f32')`.           */
/* ------------------------------------------------------------------------------------------ */
/* x: f32 NCHW [N,C,H,W] -> y: f32 NHWC [N,H,W,Cpad], channels >= C zero, Cpad % 4 == 0. */
int sc2_nchw_f32_to_nhwc_f32(const float *x, float *y, int N, int C, int H, int W, int Cpad, void *stream);
/* Output channels per weight chunk for a given Cout (32, 48 or 96): the packing unit of w_frag below. */
int sc2_conv_f32_chunk_channels(int Cout);
/* d      : as for sc2_conv2d_fwd; Cin % 4 == 0 (the padded channel count of x), square stride / padding, no output scatter;
 *          Cout_pad / k_order are ignored.  Kpad: 0, or the number of REAL input channels: Cin == 4 with Kpad == 3 says that
 *          channel 3 of x is zero and zero-weighted (the layout sc2_nchw_f32_to_nhwc_f32 and the packer produce for RGB), and
 *          the kernel skips those products (same sums: they are exact zeros).  With that, k_order == 1 says that x is the f32 NCHW image
 *          [N, 3, H, W] itself (the reference's input layout): the three channel planes are read in place, no NHWC copy.  a_op: NONE / ABS / SQUARE.  epilogue: NONE, BIAS, GDN (y = ep_x * (1 /
 *          (ep_beta[c] + acc))), IGDN (y = ep_x * (ep_beta[c] + acc)) -- the operation order of compressai GDN1.forward;
 *          FUSED_GDN / FUSED_IGDN (Cout <= 96): the conv followed by GDN1 over its own output in one launch, `ep_x` = the
 *          effective gamma as the w_frag of a 1x1 conv Cout -> Cout (same packing), ep_beta = the effective beta; bit-identical
 *          to the two launches.
 * x      : f32 NHWC [N,H,W,Cin]; addressed through a 32-bit buffer descriptor: N*H*W*Cin*4 (N*H*W*12 for the NCHW image) must be
 *          below 0x7FF00000 bytes (SC2_ERR_UNSUPPORTED otherwise -- about 445 images of the 96-channel 112 x 112 map; the host side
 *          runs larger batches as slices, `FPBasedResNetBottleneck._analysis_f32`)
 * w_frag : f32, [chunks][steps][NT][64 lanes][4] with cc = sc2_conv_f32_chunk_channels(Cout), NT = cc / 16, chunks =
 *          ceil(Cout / cc), steps = ceil(KH*KW*Cin / 16); entry (ch, s, nt, lane = q*16 + r, j) =
 *          W[ch*cc + nt*16 + r][k = 16 s + 4 q + j], k = (kh*KW + kw)*Cin + ci, zero beyond Cout / K.
 * ep_x   : f32 NHWC [N,OH,OW,Cout] for GDN / IGDN;  ep_beta : f32 [Cout] (beta, bias, or the medians for symbols)
 * y      : SC2_OUT_F32_NHWC [N,OH,OW,Cout] / SC2_OUT_F32_NCHW / SC2_OUT_I32_NCHW_SYM (round_half_even(acc - ep_beta[c])). */
int sc2_conv2d_f32_fwd(const sc2_conv_desc *d, const float *x, const float *w_frag, void *y, const float *ep_x,
                       const float *ep_beta, void *stream);

/* First decoder stage in ONE launch: y = GDN1_512(Conv2d(Cin -> 512, k2, s1, p1, bias=False)(x)) with the inverse
 * (multiplicative) or forward (divisive) normalisation: t = conv(x); y = t * (beta + gamma |t|) resp. t / (...).
 * Replaces decoder[0] + decoder[1] of FPBasedResNetBottleneck (sc2bench/models/layer.py:486-488); the 512-channel
 * intermediate never reaches HBM.
 *   x : bf16 NHWC [N,H,W,Cin], Cin in {8,16,24};   w_packed : bf16 [512][Kpad], Kpad = sc2_conv_weight_pitch(4*Cin)
 *   gamma_frag : effective gamma as bf16 MFMA-fragment blocks [32 channel tiles][16 k steps][64 lanes][8]: entry
 *                (jt, ks, lane = fq*16 + frow, e) = gamma[jt*16 + frow][ks*32 + fq*8 + e] (one operand fragment =
 *                1 KB contiguous);   beta : f32 [512] (effective beta)
 *   y : bf16 NHWC [N,H+1,W+1,512]
 *   t_out : NULL, or bf16 laid out like y (round 5, training): the conv output in front of the GDN, the tensor the GDN's backward
 *           needs (`_forward2train`, layer.py:529-533: decoder[0] + decoder[1] as this one launch); needs N*(H+1)*(W+1)*1024 B < 2^31 */
int sc2_conv2x2_gdn512_supported(int Cin, int Cout, int KH, int KW, int stride, int pad);
int sc2_conv2x2_gdn512_fwd(const void *x, const void *w_packed, int Kpad, const void *gamma_frag, const float *beta,
                           void *y, void *t_out, int N, int H, int W, int Cin, int inverse, void *stream);

/* First encoder stage in ONE persistent launch: y = GDN1_96(Conv2d(3 -> 96, k5, s2, p2, bias=False)(x)) on the
 * pixel-pair view of the image (replaces encoder[0] + encoder[1], sc2bench/models/layer.py:476-478).
 *   x_pairs : bf16 [N, H, W/2, 8] (two pixels x four channels, channel 3 zero: sc2_nchw_f32_to_nhwc_bf16 with c_pad 4)
 *   w_frag  : bf16 MFMA-fragment blocks [6][4][64][8] of the pair-packed weights W'[96][128], k = (kh*3 + t)*8 + dw*4 + c
 *   gamma_frag : bf16 fragment blocks [6][3][64][8] of the effective gamma;  beta : f32 [96]
 *   y : bf16 NHWC [N, OH, W/2, 96], OH = (H - 1)/2 + 1.   Any width: rows are cut into 112-pixel output segments.
 *   t_out : NULL, or bf16 laid out like y (round 5, training): the conv output in front of the GDN, the tensor the GDN's backward
 *           needs (`_forward2train`, layer.py:529-533: encoder[0] + encoder[1] as this one launch) */
int sc2_conv0_gdn96_supported(int Cin_pairs, int Cout, int W_pairs);
int sc2_conv0_gdn96_fwd(const void *x_pairs, const void *w_frag, const void *gamma_frag, const float *beta, void *y, void *t_out,
                        int N, int H, int W_pairs, int inverse, void *stream);
/* The same launch on the reference's own input: x_nchw = the f32 NCHW image batch [N, 3, H, W] that
 * `FPBasedResNetBottleneck.encoder` receives (sc2bench/models/layer.py:496-498), read where it lies -- the three colour planes
 * of a pixel pair are rounded to bf16 (round-to-nearest-even, as sc2_nchw_f32_to_nhwc_bf16 rounds) as the kernel stages them:
 * bit-identical to sc2_nchw_f32_to_nhwc_bf16(c_pad 4) + sc2_conv0_gdn96_fwd without the layout pass.  W even (an odd width
 * goes through the pair view, zero-padded by the caller); y : bf16 NHWC [N, OH, W/2, 96]. */
int sc2_conv0_gdn96_nchw_fwd(const float *x_nchw, const void *w_frag, const void *gamma_frag, const float *beta, void *y,
                             int N, int H, int W, int inverse, void *stream);

/* 1x1 convolution with a long K and the weights resident in registers (conv1 + bn1 + ReLU and the stride-2 downsample of
 * the ResNet tail's layer3 / layer4 in eval mode, sc2bench/models/backbone.py:235-254): y = act(x W^T + bias).
 * Cin = 1024: Cout % 128 == 0, Cout <= 2048;  Cin = 2048: Cout % 64 == 0, Cout <= 1024;  stride 1 or 2 (no padding).
 *   x : bf16 NHWC [N,H,W,Cin];  w_frag : bf16 fragment blocks [Cout/16][Cin/32][64][8] (BN folded);  bias : f32 [Cout];
 *   y : bf16 NHWC [N,OH,OW,Cout], OH = (H-1)/stride + 1;  relu != 0: ReLU. */
int sc2_conv1x1_kres_supported(int Cin, int Cout, int stride);
int sc2_conv1x1_kres_fwd(const void *x, const void *w_frag, const float *bias, void *y, int N, int H, int W, int Cin, int Cout,
                         int stride, int relu, void *stream);

/* The last encoder convolution of the FP bottleneck, Conv2d(48 -> Cout <= 32, k2, s1, p0, bias=False) (sc2bench/models/layer.py:482
 * `encoder[4]`; 48 -> 24 in every released configuration), as a streaming kernel without LDS (conv2x2_c48.hip): the two column
 * taps of a pixel are 96 contiguous NHWC channels, so the MFMA operand fragments are plain 16-byte loads; weights resident in
 * registers.  Results bit-identical to sc2_conv2d_fwd's.
 *   x : bf16 NHWC [N,H,W,48];   w_frag : bf16 fragment blocks [2][6][64][8] of the [32][192] matrix W[co][kh*96 + kw*48 + ci]
 *       (rows >= Cout zero; hip.pack_conv2x2_c48);   y : NCHW [N,Cout,H-1,W-1], f32 latent (symbols == 0) or int32 symbols
 *       round-half-even(acc - medians[co]) (symbols != 0: EntropyModel.quantize(..., 'symbols') fused, as SC2_OUT_I32_NCHW_SYM). */
int sc2_conv2x2_c48_supported(int H, int W, int Cin, int Cout);
int sc2_conv2x2_c48_fwd(const void *x, const void *w_frag, const float *medians, void *y, int N, int H, int W, int Cin, int Cout,
                        int symbols, void *stream);

/* 1x1 convolution + bias (+ residual) (+ ReLU) in the window-plane structure (conv1x1_win.hip): conv1 + bn1 + ReLU, conv3 + bn3 +
 * identity + ReLU and the downsample layers of the torchvision Bottleneck blocks behind the bottleneck
 * (sc2bench/models/backbone.py:235-254) where K is long and the layer is not HBM-bound.  Tiles of 208 output pixels x 128
 * channels, 64-channel slabs of the pixel operand as chunk planes in LDS, weights straight into registers.
 *   x : bf16 NHWC [N,H,W,Cin], Cin % 128 == 0;   y : bf16 NHWC [N,OH,OW,Cout], OH = (H-1)/stride + 1, Cout % 128 == 0, stride 1 | 2
 *   w_frag : bf16 [Cin/32][Cout/16][64][8] (BN folded), the k-step stream of sc2_conv3x3_win_fwd for a 1 x 1 kernel (same row
 *            permutation);   bias : f32 [Cout];   residual : bf16 [N,OH,OW,Cout] or NULL (added before the ReLU);   relu != 0: ReLU. */
int sc2_conv1x1_win_supported(int Cin, int Cout, int stride);
int sc2_conv1x1_win_fwd(const void *x, const void *w_frag, const float *bias, const void *residual, const void *mask, void *y, int N,
                        int H, int W, int Cin, int Cout, int stride, int relu, void *stream);

/* 3x3 stride-1 pad-1 convolution + bias (+ ReLU) on 28 x 28 / 14 x 14 / 7 x 7 maps: conv2 + bn2 + ReLU of the torchvision
 * Bottleneck blocks of layer2 / layer3 / layer4 behind the bottleneck (sc2bench/models/backbone.py:235-254 runs them) at the
 * 224 x 224 operating point.  Tiles of 196 output pixels x 128 channels, zero-padded window planes in LDS, weights straight
 * into registers (conv3x3_win.hip).
 *   x : bf16 NHWC [N,H,W,Cin], H == W in {28, 14, 7}, Cin % 64 == 0;   y : bf16 NHWC [N,H,W,Cout], Cout % 128 == 0
 *   w_frag : bf16 [Cin/32 * 9][Cout/16][64][8] (BN folded): entry (kt = slab*9 + kh*3 + kw, tile t = 2 g + j, lane = fq*16 +
 *            frow, e) = W[32 g + 8 (frow / 4) + 4 j + frow % 4][slab*32 + fq*8 + e][kh][kw]   (the row permutation leaves
 *            every lane with eight consecutive output channels of a pixel);   bias : f32 [Cout];   relu != 0: ReLU. */
int sc2_conv3x3_win_supported(int H, int W, int Cin, int Cout);
int sc2_conv3x3_win_fwd(const void *x, const void *w_frag, const float *bias, const void *mask, void *y, int N, int H, int W, int Cin,
                        int Cout, int relu, void *stream);

/* The stride-2 form of the same kernel: conv2 + bn2 + ReLU of layer2.0 / layer3.0 / layer4.0 (3x3, stride 2, pad 1; torchvision
 * puts the stride on conv2), 56 -> 28, 28 -> 14, 14 -> 7 at the 224 x 224 operating point (sc2bench/models/backbone.py:235-254).
 * The window is stored in four parity classes so that the nine taps stay immediate offsets (conv3x3_win.hip, GeoS2).
 *   x : bf16 NHWC [N,H,W,Cin], H == W in {56, 28, 14}, Cin % 32 == 0;   y : bf16 NHWC [N,H/2,W/2,Cout], Cout % 128 == 0
 *   w_frag, bias, relu : as sc2_conv3x3_win_fwd. */
int sc2_conv3x3s2_win_supported(int H, int W, int Cin, int Cout);
int sc2_conv3x3s2_win_fwd(const void *x, const void *w_frag, const float *bias, void *y, int N, int H, int W, int Cin, int Cout,
                          int relu, void *stream);

/* The two MFMA-bound decoder convolutions of the FP / SHP / MSHP bottlenecks on the window-plane structure, with the
 * inverse GDN1 that follows the first one fused in (sc2bench/models/layer.py:489-493: Conv2d(512 -> 256, k2, p0) + GDN1(256,
 * inverse) and Conv2d(256 -> 256, k2, p1)); conv2x2_win.hip.  Results bit-identical to sc2_conv2d_fwd's.
 *   x : bf16 NHWC [N,H,W,Cin], Cin % 64 == 0, pad 0 (H, W >= 2) or 1;   y : bf16 NHWC [N,H+2pad-1,W+2pad-1,256]
 *       (W == 56 with pad 0 and W == 55 with pad 1 -- the 224 x 224 operating point -- run the static-geometry instantiation;
 *        every other width runs the same kernel over column segments of 55 output pixels: BASELINE configs 4 / 5)
 *   w_frag : bf16 [Cin/32 * 4 (+ 8 when fused)][16][64][8]: conv k-step kt = slab*4 + kh*2 + kw, then (fused) the 8 k-steps of
 *            the effective gamma [256][256] as a 1x1 layer; entry (kt, tile t = 2 g + j, lane = fq*16 + frow, e) =
 *            W[32 g + 8 (frow / 4) + 4 j + frow % 4][slab*32 + fq*8 + e][kh][kw]  (row permutation as sc2_conv3x3_win_fwd)
 *   fused != 0: y = GDN1(conv(x)) with beta f32 [256]; inverse != 0: x * (beta + gamma |x|), else x / (...).
 *   y_channels 256, y_channel0 0: y as above.  y_channels 512 (plain pad-1 convs only): y is bf16 NHWC [N,H+1,W+1,512] and this
 *   launch writes its channels [y_channel0, y_channel0 + 256), y_channel0 in {0, 256} -- the data gradient of the 512 -> 256 layer
 *   (loss.backward() of script/task/image_classification.py:79 through layer.py:489) is two such launches. */
int sc2_conv2x2_win_supported(int H, int W, int Cin, int Cout, int pad);
int sc2_conv2x2_win_fwd(const void *x, const void *w_frag, const float *beta, void *y, int N, int H, int W, int Cin, int pad,
                        int fused, int inverse, int y_channels, int y_channel0, void *stream);

/* The last decoder convolution (256 -> 256, k2, p1: 55 -> 56) with the two 1x1 layers of the caller that consume its output
 * fused behind it: conv1 + bn1 + ReLU (256 -> 128) and downsample conv + bn (256 -> 512, stride 2) of layer2.0 of the ResNet
 * tail (sc2bench/models/backbone.py:235-254 runs decoder then layer2; torchvision Bottleneck.forward).  The 411 MB
 * feature map between the bottleneck and the head is then never written (y == NULL) or written once and not read back.
 *   x : bf16 NHWC [N,55,55,Cin];   w_stream : bf16 [Cin/32*4 + 24][16][64][8] = the conv's k-steps, 8 k-steps of W1 (BN folded,
 *   rows 128..255 zero), 8 k-steps of Wds rows 0..255, 8 k-steps of Wds rows 256..511 (packing as sc2_conv2x2_win_fwd);
 *   bias1 f32 [128], bias_ds f32 [512] (folded BN);   y : bf16 NHWC [N,56,56,256] or NULL;
 *   o1 : bf16 NHWC [N,56,56,128] = relu(W1 y + bias1);   ods : bf16 NHWC [N,28,28,512] = Wds y[::2, ::2] + bias_ds.
 * Same operation order per element as the separate launches: bit-identical results. */
int sc2_conv2x2_win_tail_supported(int H, int W, int Cin);
int sc2_conv2x2_win_tail_fwd(const void *x, const void *w_stream, const float *bias1, const float *bias_ds, void *y, void *o1,
                             void *ods, int N, int H, int W, int Cin, void *stream);

/* Second encoder stage in ONE persistent launch: y = GDN1_48(Conv2d(96 -> 48, k5, s2, p2, bias=False)(x)) (replaces
 * encoder[2] + encoder[3], sc2bench/models/layer.py:479-481; inverse != 0: inverse GDN1).  W == 112 (the 224 x 224 operating
 * point) runs the static geometry, any other width the same kernel over 56-column output segments.
 *   x : bf16 NHWC [N, H, W, 96];   y : bf16 NHWC [N, (H - 1)/2 + 1, (W - 1)/2 + 1, 48]
 *   w_frag : the conv weights packed SC2_K_SLAB_MAJOR | SC2_K_B_FRAG_MAJOR ([k-step = slab*25 + tap][3][64][8] bf16)
 *   gamma_frag : bf16 fragment blocks [3][2][64][8] of the effective gamma [48][48 -> 64 zero-padded];  beta : f32 [48]
 *   t_out : NULL, or bf16 laid out like y (round 5, training): the conv output in front of the GDN, which the GDN's backward needs --
 *           `_forward2train` (layer.py:529-533) then runs encoder[2] + encoder[3] as this one launch. */
int sc2_conv2_gdn48_supported(int Cin, int Cout, int W);
int sc2_conv2_gdn48_fwd(const void *x, const void *w_frag, const void *gamma_frag, const float *beta, void *y, void *t_out, int N, int H,
                        int W, int inverse, void *stream);

/* Streaming 1x1 convolution with a short K and a wide N: y = act(x W^T + bias [+ residual]) in one persistent launch
 * (the HBM-bound 1x1 layers of the ResNet-50 tail behind the bottleneck, backbone.py:235-254: third conv of a
 * Bottleneck block with its residual add + ReLU, stride-2 downsample).
 *   x : bf16 NHWC [N,H,W,Cin], Cin in {128, 256};  w_frag : bf16 MFMA-fragment blocks [Cout/16][Cin/32][64][8], entry
 *   (jt, ks, lane = fq*16 + frow, e) = W[jt*16 + frow][ks*32 + fq*8 + e];  bias : f32 [Cout];  residual : bf16 NHWC
 *   [N,OH,OW,Cout] or NULL;  y : bf16 NHWC [N,OH,OW,Cout], OH = (H-1)/stride + 1;  Cout % 256 == 0; stride 1 or 2.
 *   (round 5: Cin also 64 and 512, Cout % 128 == 0.)  mask : NULL, or bf16 like y (sc2_conv1x1_stream_mask_supported: Cin 128 / 256,
 *   Cout % 256 == 0, stride 1, relu 0): y = mask > 0 ? x W^T + bias [+ residual] : 0 -- the layer is the data gradient of a Bottleneck
 *   block's conv1 and `mask` the previous block's output, whose ReLU gradient then needs no pass of its own (as sc2_conv1x1_win_fwd /
 *   sc2_conv3x3_win_fwd take it). */
int sc2_conv1x1_stream_supported(int Cin, int Cout, int stride);
int sc2_conv1x1_stream_mask_supported(int Cin, int Cout, int stride);
int sc2_conv1x1_stream_fwd(const void *x, const void *w_frag, const float *bias, const void *residual, const void *mask, void *y, int N,
                           int H, int W, int Cin, int Cout, int stride, int relu, void *stream);

/* Weight gradient of sc2_conv2d_fwd: dw[co][(kh*KW+kw)*Cin+ci] = sum over output pixels of gy * im2col(x).
 * Replaces the weight half of nn.Conv2d's backward (reached through loss.backward(), image_classification.py:79).
 *   x  : bf16 NHWC [N,H,W,Cin]      gy : bf16 NHWC [N,OH,OW,Cout]
 *   dw : f32 [Cout][KH*KW*Cin], overwritten (zeroed on the stream, then accumulated with f32 atomics)
 * Uses N,H,W,Cin,Cout,KH,KW,stride,pad,OH,OW and a_op (SC2_AOP_ABS: |x| operand, for GDN1's gamma) of the
 * descriptor; the other fields are ignored. */
int sc2_conv2d_wgrad(const sc2_conv_desc *d, const void *x, const void *gy, float *dw, void *stream);

/* Element-wise halves of the GDN1 backward (norm = beta + gamma|x|, y = x/norm or x*norm; all tensors bf16 NHWC
 * [M pixels][C], C % 8 == 0).  The channel-mixing halves are sc2_conv2d_fwd (gamma^T d_norm) and sc2_conv2d_wgrad.
 *   pre : d_norm, dx_direct out; d_beta f32[C] out (zeroed on the stream, then atomically accumulated)
 *   post: dx = dx_direct + sign(x) * t */
int sc2_gdn_bwd_pre(const void *gy, const void *x, const void *norm, long long M, int C, int inverse, void *d_norm,
                    void *dx_direct, float *d_beta, void *stream);
int sc2_gdn_bwd_post(const void *dx_direct, const void *x, const void *t, long long n_elements, void *dx, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* Entropy bottleneck (factorised prior), filters fixed to (3,3,3,3) as sc2bench uses them     */
/* Replaces CompressAI EntropyBottleneck.forward called at layer.py:531 (and wrapper.py:238),  */
/* EntropyModel.quantize/dequantize at layer.py:545-547.                                       */
/* ------------------------------------------------------------------------------------------ */
#define SC2_EB_PARAM_STRIDE 64
/* Per-channel parameter block (f32[64]); "sp" = softplus(matrix), "th" = tanh(factor):
 *  [0..2]   sp(M0)[3x1]  [3..5]   b0[3]  [6..8]   th(f0)[3]
 *  [9..17]  sp(M1)[3x3]  [18..20] b1[3]  [21..23] th(f1)[3]
 *  [24..32] sp(M2)[3x3]  [33..35] b2[3]  [36..38] th(f2)[3]
 *  [39..47] sp(M3)[3x3]  [48..50] b3[3]  [51..53] th(f3)[3]
 *  [54..56] sp(M4)[1x3]  [57]     b4     [58]     median (quantiles[c,0,1])   [59..63] 0 */
enum sc2_eb_mode { SC2_EB_NOISE = 0, SC2_EB_DEQUANTIZE = 1 };

/* y        : f32 NCHW [N,C,HW]
 * noise    : f32 NCHW, same shape (mode NOISE: y_hat = y + noise; ignored otherwise; the caller
 *            draws U(-1/2,1/2) as the reference does with torch.empty_like().uniform_())
 * y_hat    : f32 NCHW out (nullable)
 * y_hat_bf16_nhwc : bf16 NHWC out, the decoder's input (nullable)
 * lik      : f32 NCHW out = max(sigmoid(L(y_hat+.5)) - sigmoid(L(y_hat-.5)), lik_bound) (nullable)
 * bits_partial : f32 [N*C*gridDim.y] partial sums of -log2(lik) (nullable; n_partial receives count)
 */
int sc2_eb_forward(const float *y, const float *noise, const float *params, int N, int C, int HW,
                   int mode, float lik_bound, float *y_hat, void *y_hat_bf16_nhwc, float *lik,
                   float *bits_partial, int bits_partial_len, void *stream);
/* number of partial sums sc2_eb_forward writes for this problem size */
int sc2_eb_bits_partial_len(int N, int C, int HW);

/* Backward of sc2_eb_forward (CompressAI EntropyBottleneck.forward under autograd, reached from the training loop
 * through layer.py:531) w.r.t. y and the effective parameter block.  Re-evaluates the forward from (y, noise).
 *   g_yhat, g_lik : upstream gradients, f32 NCHW (nullable = zero)
 *   g_y           : f32 NCHW out
 *   g_params_partial : f32 [n_partial][64] out, n_partial = sc2_eb_bits_partial_len(N, C, HW) rows ordered
 *                   (n, c, tile); the caller sums the rows of a channel.  Slot 58 is the median's gradient. */
int sc2_eb_backward(const float *y, const float *noise, const float *params, int N, int C, int HW, int mode,
                    float lik_bound, const float *g_yhat, const float *g_lik, float *g_y, float *g_params_partial,
                    int n_partial, void *stream);

/* symbols = int32(round_half_even(y - median[c])) in NCHW order, one row of C*HW per image.
 * Replaces EntropyModel.quantize(x, "symbols", means) reached from layer.py:506. */
int sc2_eb_symbols(const float *y, const float *medians, int N, int C, int HW, int32_t *symbols, void *stream);
/* y_hat = float(symbols) + median[c]   (EntropyModel.dequantize reached from layer.py:520). */
int sc2_eb_dequantize(const int32_t *symbols, const float *medians, int N, int C, int HW, float *y_hat_f32_nchw,
                      void *y_hat_bf16_nhwc, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* GaussianConditional (hyperprior bottlenecks, sc2bench/models/layer.py:553-817)              */
/* Replaces CompressAI GaussianConditional.forward / quantize / dequantize / build_indexes     */
/* reached from layer.py:646-647,665,679,691-693,776,785,794,811-813.                          */
/* All tensors f32 NCHW with `chw` elements per image; `scales` / `means` may be channel       */
/* slices of a wider tensor and carry their own per-image element stride.                      */
/* ------------------------------------------------------------------------------------------ */
/* mode SC2_EB_NOISE: y_hat = y + noise (means ignored, as upstream); SC2_EB_DEQUANTIZE: y_hat = round(y - means) +
 * means.  lik = max(Phi((.5 - |v|)/s) - Phi((-.5 - |v|)/s), lik_bound), v = y_hat - means, s = max(scales,
 * scale_bound), Phi(t) = .5 erfc(-t / sqrt 2).  means, noise, y_hat, lik nullable. */
int sc2_gc_forward(const float *y, const float *scales, int64_t scales_img_stride, const float *means,
                   int64_t means_img_stride, const float *noise, int64_t n_img, int64_t chw, int mode,
                   float scale_bound, float lik_bound, float *y_hat, float *lik, void *stream);
/* Backward of sc2_gc_forward in SC2_EB_NOISE mode (GaussianConditional.forward under autograd in training, reached
 * from layer.py:679,794 through loss.backward()): g_yhat / g_lik upstream gradients (nullable = zero); g_y, g_scales,
 * g_means dense f32 [n_img][chw] outputs (each nullable).  noise nullable = zero. */
int sc2_gc_backward(const float *y, const float *scales, int64_t scales_img_stride, const float *means,
                    int64_t means_img_stride, const float *noise, int64_t n_img, int64_t chw, float scale_bound,
                    float lik_bound, const float *g_yhat, const float *g_lik, float *g_y, float *g_scales,
                    float *g_means, void *stream);
/* symbols = int32(round_half_even(y - means)); indexes = (n_table - 1) - #{t < n_table - 1 : max(scales,
 * scale_bound) <= scale_table[t]} (GaussianConditional.build_indexes).  Either output may be NULL. */
int sc2_gc_symbols_indexes(const float *y, const float *scales, int64_t scales_img_stride, const float *means,
                           int64_t means_img_stride, int64_t n_img, int64_t chw, const float *scale_table, int n_table,
                           float scale_bound, int32_t *symbols, int32_t *indexes, void *stream);
/* y_hat = float(symbols) + means (means nullable), f32 NCHW and / or bf16 NHWC [n_img, HW, C]. */
int sc2_gc_dequantize(const int32_t *symbols, const float *means, int64_t means_img_stride, int64_t n_img, int C,
                      int HW, float *y_hat_f32_nchw, void *y_hat_bf16_nhwc, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* CDF quantisation (HOST function, bit-exact integer result)                                  */
/* Replaces compressai._CXX.pmf_to_quantized_cdf reached from layer.py:431-441 (update()).     */
/* pmf: HOST f32[n]; cdf: HOST u32[n+1].                                                       */
/* ------------------------------------------------------------------------------------------ */
int sc2_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf);

/* ------------------------------------------------------------------------------------------ */
/* rANS (64-bit state, 32-bit renormalisation, 16-bit precision, 4-bit bypass), one stream per */
/* image, all streams of a batch in one launch.  Bit-exact to CompressAI's                     */
/* RansEncoder.encode_with_indexes / RansDecoder.decode_with_indexes (layer.py:506,520).       */
/* ------------------------------------------------------------------------------------------ */
/* symbols : i32 [n_streams][n_sym]
 * indexes : i32 [n_streams][n_sym] CDF row per symbol, or NULL -> row = position / index_div
 *           (the entropy bottleneck's indexes[n,c,h,w] = c with index_div = H*W)
 * cdfs    : i32 [n_cdfs][cdf_stride];  cdf_sizes, offsets : i32 [n_cdfs]
 * out     : u8  [n_streams][out_stride]; stream i occupies
 *           out[i*out_stride + out_offset[i] .. + out_nbytes[i])  (streams are END-aligned in
 *           their rows because rANS emits its words back to front)
 * out_stride must be >= sc2_rans_max_bytes(n_sym); out_stride % 4 == 0.
 * status  : i32 [n_streams] 0 ok; bit 0 = row overflow (cannot happen with sc2_rans_max_bytes); bit 1 = a symbol with
 *           |symbol - offset| >= 2^30 was clamped (non-finite / diverged latent); decoders only: bit 3 = corrupt, truncated or
 *           hostile stream (an escape that announces more than the eight nibbles of a 32-bit value -- upstream's decoder
 *           loops on such a count for ever -- or words read past the end of the stream, where zeros are supplied): the
 *           decode always terminates, the symbols of a flagged stream are undefined.
 * workspace : device scratch of sc2_rans_workspace_bytes(n_streams, n_sym, n_cdfs, cdf_stride) bytes (the
 *             [position][lane] transposed intermediate of the multi-pass coder and the per-entry reciprocal
 *             table); contents undefined afterwards.
 */
int64_t sc2_rans_max_bytes(int64_t n_sym);
int64_t sc2_rans_workspace_bytes(int n_streams, int64_t n_sym, int n_cdfs, int cdf_stride);
int sc2_rans_encode_batch(const int32_t *symbols, const int32_t *indexes, int64_t index_div, int n_streams,
                          int64_t n_sym, const int32_t *cdfs, int n_cdfs, int cdf_stride,
                          const int32_t *cdf_sizes, const int32_t *offsets, uint8_t *out, int64_t out_stride,
                          int32_t *out_offset, int32_t *out_nbytes, int32_t *status, void *workspace,
                          int64_t workspace_bytes, void *stream);
/* in : u8 [n_streams][in_stride], stream i at in[i*in_stride + in_offset[i] ..+in_nbytes[i]),
 *      in_offset[i] % 4 == 0.  symbols_out : i32 [n_streams][n_sym]. */
int sc2_rans_decode_batch(const uint8_t *in, int64_t in_stride, const int32_t *in_offset, const int32_t *in_nbytes,
                          const int32_t *indexes, int64_t index_div, int n_streams, int64_t n_sym,
                          const int32_t *cdfs, int n_cdfs, int cdf_stride, const int32_t *cdf_sizes,
                          const int32_t *offsets, int32_t *symbols_out, int32_t *status, void *workspace,
                          int64_t workspace_bytes, void *stream);
/* sc2_rans_decode_batch for implicit indexes (row = position / index_div, n_sym == n_cdfs * index_div) with
 * EntropyModel.dequantize fused into its last pass (sc2bench/models/layer.py:520 `entropy_bottleneck.decompress` ->
 * dequantize(values, medians)): y_hat_bf16_nhwc[s][pix][c] = bf16(symbol[s][c * index_div + pix] + medians[c]), the layout the
 * synthesis kernels read (as sc2_eb_dequantize's y_hat_bf16_nhwc, same rounding).  symbols_out may be NULL (the int32 symbols
 * are then never written).  n_cdfs % 8 == 0, n_cdfs <= 64. */
int sc2_rans_decode_dequantize_batch(const uint8_t *in, int64_t in_stride, const int32_t *in_offset, const int32_t *in_nbytes,
                                     int64_t index_div, int n_streams, int64_t n_sym, const int32_t *cdfs, int n_cdfs,
                                     int cdf_stride, const int32_t *cdf_sizes, const int32_t *offsets, const float *medians,
                                     int32_t *symbols_out, void *y_hat_bf16_nhwc, int32_t *status, void *workspace,
                                     int64_t workspace_bytes, void *stream);

/* The same call; additionally records the caller's two hipEvent_t (may be NULL) on `stream` directly in front of and behind its
 * LAST pass (dequantise + transposition to NHWC = EntropyModel.dequantize, layer.py:520), so that a caller can time that pass
 * as part of the bottleneck forward while the serial passes count as the range coder's. */
int sc2_rans_decode_dequantize_batch_ev(const uint8_t *in, int64_t in_stride, const int32_t *in_offset, const int32_t *in_nbytes,
                                        int64_t index_div, int n_streams, int64_t n_sym, const int32_t *cdfs, int n_cdfs,
                                        int cdf_stride, const int32_t *cdf_sizes, const int32_t *offsets, const float *medians,
                                        int32_t *symbols_out, void *y_hat_bf16_nhwc, int32_t *status, void *workspace,
                                        int64_t workspace_bytes, void *stream, void *ev_dequantize_begin, void *ev_dequantize_end);

/* ------------------------------------------------------------------------------------------ */
/* Element-wise pieces of the distillation step (stage 1 of the Entropic-Student recipe:          */
/* nn.MSELoss(reduction='sum') between student and teacher feature maps, yaml:155-200; ReLU       */
/* gradients of the frozen ResNet tail, yaml:135), bf16 tensors of n elements, n % 8 == 0.         */
/* ------------------------------------------------------------------------------------------ */
/* partial : f32 [sc2_mse_partial_len(n)] block sums of (x - y)^2 (f32 accumulation, fixed order); the caller adds them. */
int sc2_mse_partial_len(long long n);
int sc2_mse_sum_bf16(const void *x, const void *y, long long n, float *partial, void *stream);
/* gx = bf16(2 * scale[0] * (x - y)); scale: DEVICE f32 scalar (the upstream gradient of the loss). */
int sc2_mse_grad_bf16(const void *x, const void *y, long long n, const float *scale, void *gx, void *stream);
/* gi = (g [+ add]) * (out > 0): gradient through a ReLU whose output was saved; `add` (nullable) = a second gradient that
 * reaches the same tensor, summed in f32 before the mask. */
int sc2_relu_bwd_bf16(const void *g, const void *out, const void *add, long long n, void *gi, void *stream);
/* ... with the feature-matching MSE term that sits on the same tensor folded in: g_in = ([g] + 2 * scale[0] * (out - t)) * (out > 0),
 * g may be NULL (no other gradient reaches the tensor).  `out` = the student's stack output the criterion compares with the teacher's
 * `t` (yaml:155-200: MSELoss(reduction='sum') per layer), scale = loss weight x upstream gradient [/ numel for 'mean'] as an f32 device
 * scalar; relu = 0: no activation behind the tensor (the bottleneck's own output: [g] + 2 scale (out - t)).  Replaces sc2_mse_grad_bf16 +
 * the add + sc2_relu_bwd_bf16 at the output of a frozen stack. */
int sc2_relu_bwd_mse_bf16(const void *g, const void *out, const void *t, const float *scale, long long n, int relu, void *gi, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* HOST range coder (same bit-exact format; every pointer is HOST memory, no HIP call is made).  */
/* For the reference's evaluation mode -- test batch size 1, ONE stream per forward               */
/* (script/task/image_classification.py:106-145 -> layer.py:506,520): a single rANS stream is a   */
/* serial chain that one CPU core steps ~30x faster than one GPU lane, so below a stream-count    */
/* threshold EntropyBottleneck / GaussianConditional .compress / .decompress call these instead   */
/* of the batched device coder.  Product code (csrc/rans_host.cpp), not the test oracle.          */
/* ------------------------------------------------------------------------------------------ */
/* Opaque prepared tables: per CDF row the cumulative frequencies and a 256-bucket index for the decoder's symbol
 * search.  cdfs : i32 [n_cdfs][cdf_stride], cdf_sizes / offsets : i32 [n_cdfs] as above.  SC2_ERR_INVALID_ARG if a row is
 * not a strictly increasing table from 0 to 65 536 with at least two entries. */
typedef struct sc2_rans_host_tables sc2_rans_host_tables;
int sc2_rans_host_tables_create(const int32_t *cdfs, int n_cdfs, int cdf_stride, const int32_t *cdf_sizes,
                                const int32_t *offsets, sc2_rans_host_tables **out);
void sc2_rans_host_tables_destroy(sc2_rans_host_tables *tables);
/* Both passes in one call: every stream is encoded into its row of `out` and that row decoded into symbols_out by the same host
 * thread (the first coder groups of a pipelined run: sc2bench_amd/pipeline.py `host_steps`); status = encode | decode bits. */
int sc2_rans_code_host(const sc2_rans_host_tables *tables, const int32_t *symbols, const int32_t *indexes, int64_t index_div,
                       int n_streams, int64_t n_sym, uint8_t *out, int64_t out_stride, int32_t *out_offset, int32_t *out_nbytes,
                       int32_t *symbols_out, int32_t *status, int n_threads);
/* floor(x / freq) as the host ENCODER forms it (a multiplication by freq's precomputed reciprocal, exact for x < 2^63: the
 * coder's state never leaves [2^31, 2^63)); exported so that a test can sweep it against the division (1 <= freq <= 65 536). */
uint64_t sc2_rans_host_rcp_div(uint64_t x, uint32_t freq);
/* Same arguments, row layout (END-aligned streams, out_offset / out_nbytes) and status bits as sc2_rans_encode_batch /
 * sc2_rans_decode_batch (additionally bit 2 = a CDF-row index outside the table); `out` / `in` 4-byte aligned;
 * streams are spread over n_threads host threads (<= 1: the calling thread). */
int sc2_rans_encode_host(const sc2_rans_host_tables *tables, const int32_t *symbols, const int32_t *indexes,
                         int64_t index_div, int n_streams, int64_t n_sym, uint8_t *out, int64_t out_stride,
                         int32_t *out_offset, int32_t *out_nbytes, int32_t *status, int n_threads);
int sc2_rans_decode_host(const sc2_rans_host_tables *tables, const uint8_t *in, int64_t in_stride,
                         const int32_t *in_offset, const int32_t *in_nbytes, const int32_t *indexes, int64_t index_div,
                         int n_streams, int64_t n_sym, int32_t *symbols_out, int32_t *status, int n_threads);

/* ------------------------------------------------------------------------------------------ */
/* Diagnostics (csrc/diag.hip; no product path calls these).                                    */
/* ------------------------------------------------------------------------------------------ */
/* The shader clock the chip holds while OTHER kernels run: `n_workgroups` probe waves (one per workgroup; launch >= 8 so that
 * every XCD gets one) each write n_samples triples (s_memtime, s_memrealtime, XCC id) as u64 into samples[wg][sample][3], one
 * every period_ticks ticks of the constant 100 MHz counter.  clock = delta s_memtime / delta s_memrealtime x 100 MHz between two
 * samples of one workgroup (MI355X_MICROARCH.md, DVFS item 6).  Launch it on a stream of its own BEFORE the kernels under study
 * so that it is resident while they run (tools/clock_probe.py). */
int sc2_clock_probe(unsigned long long *samples, int n_workgroups, int n_samples, unsigned period_ticks, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SC2_BOTTLENECK_H */
