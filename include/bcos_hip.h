/*
 * bcos_hip.h -- C ABI of the MI355X (gfx950) B-cos forward / explanation hot path.
 *
 * The reference (shrebox/B-cosification) has no FFI for this path: the boundary is the
 * Python nn.Module API of bcos/modules.  Every entry point below
 * therefore names the reference *function* whose ATen launch sequence it replaces
 * (paths relative to the reference root), and the Python host layer in
 * b-cosification_amd/bcos/ binds them with ctypes.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - every pointer is a DEVICE pointer to fp32 data unless the name says otherwise;
 *     the caller owns every buffer; the library never allocates or frees device memory.
 *   - all work is enqueued on the hipStream_t passed as `void* stream` (NULL = default
 *     stream); calls are asynchronous and re-entrant; the only global mutable state is the
 *     thread-local last-error string, the process-wide DEFAULT contraction mode (which a call
 *     overrides through bcos_operands.contraction) and the option table of bcos_set_option.
 *     The library never reads the process environment.
 *   - return value: 0 = ok, negative = error (BCOS_E_*); never throws.
 *   - activations are NHWC ("channels-last": pixel-major, channels contiguous).  The
 *     logical NCHW shape of the reference is kept by the Python layer through
 *     torch.channels_last strides, so no copies are involved.
 *   - weights are [Cout][taps][Cin] (K-contiguous, "KRSC"); Cin must be a multiple of 4.
 */
#ifndef BCOS_HIP_H
#define BCOS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BCOS_ABI_VERSION 9

enum {
    BCOS_OK = 0,
    BCOS_E_INVAL = -22,   /* bad shape / alignment / NULL where required */
    BCOS_E_NOSUP = -95,   /* combination not supported by the kernels   */
    BCOS_E_LAUNCH = -5    /* hip launch error (see bcos_last_error_string) */
};

/* B-cos scaling flavour applied to the contraction result `lin` of one output row.
 *   BCOS_NONE        y = lin                                (b == 1: bcosconv2d.py:172-174)
 *   BCOS_CONV_EPS    norm = sqrt(sum_patch x^2 + 1e-6)       (bcosconv2d.py:212-221)
 *   BCOS_LINEAR_EPS  norm = sqrt(sum x^2) + 1e-12            (bcoslinear.py:113)
 * with  b == 2 : s = |lin| / norm                            (bcosconv2d.py:186-187)
 *       b != 2 : s = (|lin / norm| + 1e-6)^(b-1)             (bcosconv2d.py:188-190)
 * and   y = s * lin.                                                                  */
enum { BCOS_NONE = 0, BCOS_CONV_EPS = 1, BCOS_LINEAR_EPS = 2 };

/*
 * Generic "tap convolution" descriptor: one implicit GEMM
 *     acc[m, co] = sum_{t < TH*TW} sum_{c < C} A[n, i*in_sh + dh(t), j*in_sw + dw(t), c] * Wt[co, t, c]
 * over rows m = (n, i, j), n < N, i < P, j < Q, with dh(t) = dh0 + (t / TW) * dstep_h,
 * dw(t) = dw0 + (t % TW) * dstep_w, taps outside [0,H)x[0,W) contributing zero
 * (zero padding).  Row m is written to pixel (n, i*out_sh + out_h0, j*out_sw + out_w0)
 * of an NHWC tensor [N, OH, OW, Cout].
 *
 * A forward convolution (stride s, padding p, dilation d) is
 *     in_s = s, dh0 = -p, dstep = d, TH x TW = kernel, P x Q = OH x OW, out_s = 1, out_0 = 0.
 * The input-gradient ("dgrad") of a strided convolution is one launch per output parity
 * class with the matching sub-kernel, in_s = 1 and out_s = stride (bcos_hip/ops.py: DgradPlan),
 * or -- narrow outputs -- ONE launch whose columns are (parity class, channel), see out_cgroup.
 */
typedef struct bcos_tapconv_geom {
    int32_t N, H, W, C;          /* A operand: NHWC, C % 4 == 0                      */
    int32_t P, Q;                /* row grid per image                              */
    int32_t in_sh, in_sw;        /* input coordinate step per row-grid step         */
    int32_t dh0, dw0;            /* first tap offset                                */
    int32_t dstep_h, dstep_w;    /* tap-to-tap offset (dilation)                    */
    int32_t TH, TW;              /* tap grid                                        */
    int32_t OH, OW;              /* output tensor spatial size                      */
    int32_t out_sh, out_sw;      /* output pixel step per row-grid step             */
    int32_t out_h0, out_w0;      /* output pixel offset                             */
    int32_t Cout;                /* GEMM N; output channel count                    */
    int32_t a_pitch;             /* floats between consecutive pixels of A (0 = C): lets a launch
                                    read a channel slice (grouped convolution)      */
    int32_t out_pitch;           /* floats between consecutive output pixels (0 = Cout); every
                                    per-element epilogue tensor uses the same pitch  */
    int32_t norm_pitch;          /* floats between consecutive pixels of norm_out (0 = 1) */
    int32_t out_cgroup;          /* 0: off.  G > 0 ("depth to space"): Cout = out_sh * out_sw * G and column
                                    (dh * out_sw + dw) * G + c of row (n, i, j) is channel c of output pixel
                                    (n, i*out_sh + dh, j*out_sw + dw): ALL parity classes of a strided input
                                    gradient in one launch over the union of their taps (weights of taps a class
                                    does not use are zero; bcos_hip/ops.py: DgradPlan).  Needs G % 4 == 0,
                                    out_h0 == out_w0 == 0, 16-byte addressable tensors, no max_out, no *_absmax */
    int32_t groups;              /* 0 / 1: off.  G > 1 (grouped convolution, bcosconv2d.py:84-140 `groups`): one launch for all
                                    groups.  C and Cout are PER GROUP: group g reads channels [g C, (g+1) C) of every pixel of A
                                    (a_pitch 0 = G C), contracts them with weight rows [g Cout, (g+1) Cout) of wt [G Cout][taps][C]
                                    and writes output columns g Cout + co (out_pitch 0 = G Cout; bias / ch_scale / ch_shift are
                                    indexed by that global column); norm_out holds one patch norm per pixel AND group
                                    ([pixels][norm_pitch], norm_pitch 0 = G).  Runs on the fp32 / bf16x3 loops (no f16x2, no
                                    *_absmax, no out_cgroup, no fused MaxOut); the pre-split bf16x3 image is used when
                                    Cout % 32 == 0. */
} bcos_tapconv_geom;

/*
 * Fused epilogue, applied per output element (row m -> output pixel index `pix`,
 * channel c, idx = pix*out_pitch + c).  NULL pointers / zero flags switch a stage off.
 * Stages run in this order:
 *     v = acc
 *     v *= col_scale[c]                              unit-norm projection of NormedConv2d / NormedLinear folded into the contraction:
 *                                                    conv(x, w / ||w||) = conv(x, w) / ||w_c|| per output channel (bcosconv2d.py:26-35,
 *                                                    bcoslinear.py:25-27); col_scale from bcos_weight_row_invnorm (ABI v5)
 *     v += bias[c]                                   nn.Conv2d bias (bcosifyconv2d.py:18-31)
 *     s = bcos_scale(v, norm[m]); v *= s             B-cos transform (bcos_mode, b)
 *     v *= ch_scale[c]; s *= ch_scale[c]             BatchNormUncentered2d eval: weight/sqrt(var+eps)
 *                                                    (batchnorm_uncentered.py:46-60)
 *     v += ch_shift[c]                               ... its bias
 *     v += addend[idx]                               residual add (fwd) / gradient accumulation (dgrad)
 *     relu == 2: gate = 0.5 (1 + erf(v / sqrt 2)); s *= gate; v *= gate   (MyGELU, bcosify_vit.py:27-32)
 *     relu == 1: s = v > 0 ? s : 0; v = max(v, 0)    (relu_gate != NULL: the decision is relu_gate[idx] > 0 instead of
 *                                                    v > 0 -- replay of recorded gates, see tests/ gate-pinned parity)
 *     out [idx] = mul  ? v * mul [idx] : v           dgrad: multiply by the stored scale of the layer below
 *     out2[idx] = v [* mul2[idx]] [* (gate2[idx] > 0)]   second product of the same v (shortcut gradient)
 *     scale_out[idx] = s                             d out / d lin with the dynamic scale detached
 *                                                    (explanation mode: bcosconv2d.py:181-184)
 *     norm_out[pix] = norm[m]                        (only written by blocks of the first Cout tile)
 *     out_absmax[pix]  = max(out_absmax[pix],  bits(max_c |out [idx]|))   atomically; the caller zeroes the buffer.
 *     out2_absmax[pix] = ... of out2                  "bits" = the fp32 bit pattern (monotonic for non-negative values).
 *                                                    These per-pixel maxima are the `a_absmax` of the launch that reads
 *                                                    the tensor as its A operand (f16x2 contraction, see bcos_operands).
 */
typedef struct bcos_epilogue {
    const float* bias;
    const float* ch_scale;
    const float* ch_shift;
    const float* addend;
    const float* mul;
    const float* mul2;
    const float* gate2;
    const float* relu_gate;
    float* out;
    float* out2;
    float* scale_out;
    float* norm_out;
    uint32_t* out_absmax;   /* NULL or [N*OH*OW] */
    uint32_t* out2_absmax;  /* NULL or [N*OH*OW] */
    const float* mul_norm;  /* BCOS_EPI_MUL_FROM_ACT: patch norms [N*OH*OW] of the layer whose activation `mul` holds */
    const float* mul_csc;   /* ... its ch_scale [Cout] (NULL = 1) */
    const float* mul_csh;   /* ... its ch_shift [Cout] (NULL = 0) */
    const float* col_scale; /* NULL or [Cout] (grouped launches: [G Cout]): factor of every accumulator column, applied first (ABI v5) */
    const float* row_scale; /* NULL or [N*OH*OW]: factor of every accumulator ROW, applied with col_scale ahead of the bias (ABI v7)         */
    const float* a_sumsq;   /* NULL or [N*OH*OW] (B-cos launches): sum of squares of the row's EFFECTIVE operand, taken in place of the sum
                               the loop collects from A.  The two fields fold a LayerNorm into the contraction that reads it
                               (centered_norms.py:197-224 in front of a linear layer): with z = gamma (x - mean(x)) r + beta,
                                   W z = r (W' x) + W beta,   W'[j,k] = gamma[k] W[j,k] - mean_k'(gamma[k'] W[j,k'])
                               -- A = x, wt = W' (built once per weight), row_scale = r = 1 / sqrt(var + eps), bias = W beta,
                               a_sumsq = |z|^2 (both per row from bcos_layernorm_stats) -- and the gradient of the detached-variance
                               form, gx = r (gamma gz - mean(gamma gz)), is the input-gradient launch over W'^T with the same row_scale. */
    uint32_t* out_imgmax;   /* NULL or [N] (ABI v9), ZERO-FILLED by the caller before the first launch that writes the tensor: the launch folds
                               every row maximum it emits through out_absmax into max over the pixels of image n -- what
                               bcos_image_absrange computes from out_absmax in a pass of its own -- so that a 3 x 3 launch reading `out`
                               (bcos_operands.a_imgmax) needs no such pass.  Several launches filling disjoint pixels of one tensor
                               accumulate.  Only for launches bcos_tapconv_fuses_image_range() answers 1 for; with out_imgmin_c.      */
    uint32_t* out_imgmin_c; /* ... and a LOWER BOUND of the min over the NONZERO pixels of image n, stored complemented (~v; 0 = no nonzero
                               pixel): exact when a tile owns its pixels (one column tile), the minimum over the column tiles' own maxima
                               otherwise.  Handed on as bcos_operands.a_imgmin_c.                                                     */
    const float* rowadd;    /* NULL or a tensor indexed like `out` (ABI v9), with rowadd_scale [output pixels]: a plain gradient launch
                               (bcos_mode BCOS_NONE, no mul / out2, addend_sub <= 1) writes out = acc + (rowadd_scale[pixel] * rowadd + addend)
                               -- the patch-norm term of a 1 x 1 / stride-1 B-cos layer's input gradient, x * (sum over the patches that
                               contain the pixel of r) = x[pixel] * r[pixel], which bcos_patch_norm_bwd_add otherwise writes as a tensor of
                               its own for this launch to read back as `addend`.  Specialised epilogues only: BCOS_E_NOSUP otherwise
                               (the caller falls back to bcos_patch_norm_bwd_add).                                                    */
    const float* rowadd_scale;
    int32_t bcos_mode;      /* BCOS_NONE / BCOS_CONV_EPS / BCOS_LINEAR_EPS */
    int32_t relu;           /* 0 none, 1 ReLU, 2 GELU with constant gate  */
    float b;                /* the B-cos exponent B (2 = fast path)       */
    int32_t flags;          /* BCOS_EPI_* bits                            */
    int32_t max_out;        /* 0 / 1: off.  2 or 4 (Cout % 4 == 0): MaxOut fused (bcosconv2d.py:166-170): the Cout accumulator
                               columns are the M adjacent filters of Cout / M units; v = max over each unit before the B-cos
                               scale; out is [pixels, Cout / M] (out_pitch 0 = Cout / M); scale_out keeps width Cout and holds s at
                               the winning filter (first maximum) and 0 elsewhere, i.e. d out / d lin.  Only with bias, the scale,
                               out, scale_out, norm_out.                                                          */
    int32_t addend_sub;     /* 0 / 1: `addend` is indexed like `out`.  s > 1 (ABI v4): `addend` is the dense tensor
                               [N, ceil(OH / s), ceil(OW / s), out_pitch] of the output pixels (h % s == 0, w % s == 0); nothing is
                               added at the other pixels.  This is the input gradient of a 1x1 / stride-s shortcut convolution
                               (zero off its s-grid) handed to the main branch's gradient launch without being scattered into a
                               zero-filled full-size tensor first.  Gradient launches only (bcos_mode BCOS_NONE, no out_cgroup, no groups). */
} bcos_epilogue;

/* compute the patch norms (norm_out) but leave v unscaled: used by the MaxOut / grouped
 * general path, where bcos_maxout_scale applies the scaling afterwards. */
#define BCOS_EPI_NORM_ONLY 1
/* use the general (|lin/norm| + 1e-6)^(b-1) form even when b == 2: the reference takes that branch for
 * its learnable-B variants (bcosifyconv2d.py:91-98 with b_loss). */
#define BCOS_EPI_FORCE_POW 2
/* ReLU launches with scale_out: store the gate decision in the least significant mantissa bit of the stored
 * multiplier t = s * ch_scale * gate (1 = open; a closed gate stores exactly 0).  The one-ulp change of t (6e-8
 * relative) is far below the fp32 rounding of the products it enters, and the explanation pass then needs no
 * separate gate tensor: */
#define BCOS_EPI_SCALE_GATE_LSB 4
/* out2 = v [* mul2] gated by that bit of `mul` (instead of gate2 > 0): saves one output-sized read per launch. */
#define BCOS_EPI_GATE2_FROM_MUL 8

/* `mul` holds the kept forward ACTIVATION a = relu(lin s ch_scale + ch_shift) of the layer below (B = 2, s = |lin| / norm, no
 * residual addend) instead of its stored multiplier t = s ch_scale gate; the epilogue rebuilds
 *     t = ch_scale sqrt(|a - ch_shift| / (|ch_scale| norm))  where a > 0, else 0
 * from mul_norm / mul_csc / mul_csh.  The forward launch of such a layer then writes out + norm_out only: one
 * output-sized HBM write less (the explanation pass reads `a` where it would have read t). */
#define BCOS_EPI_MUL_FROM_ACT 16
/* Unit-norm weight projection fused into the contraction (NormedConv2d / NormedLinear, bcos/modules/bcosconv2d.py:26-35,
 * bcoslinear.py:25-27: w_hat = w / ||w||_2 per output unit, recomputed on every call): `wt` holds the RAW weights; the launch
 * gathers sum_k w[c,k]^2 of its tile's weight rows from the staging registers they pass through and multiplies every
 * accumulator column by 1 / ||w_c|| (times col_scale[c], the optional trainable `scale`) ahead of the other stages --
 * conv(x, w / ||w||) = conv(x, w) / ||w_c||.  No projected weight tensor is written or read.  Runs on the fp32 / bf16x3 loops
 * with on-the-fly operand splits (weights that change every step have no pre-split image); inference keeps a cached
 * col_scale from bcos_weight_row_invnorm and the pre-split image of the raw weights instead.  (ABI v5) */
#define BCOS_EPI_UNIT_NORM_W 32

/* -- library -------------------------------------------------------------------------- */

/* ABI version (BCOS_ABI_VERSION of the build).  A library compiled with -DBCOS_DEV_BUILD -- the only builds in which the kernels'
 * timing knock-outs and unvalidated code paths can be switched on (csrc/bcos_internal.h: BCOS_DEV_SWITCH) -- returns the version
 * with BCOS_VERSION_DEV_FLAG set: such a library may compute wrong results and must not be used as the product. */
#define BCOS_VERSION_DEV_FLAG 0x40000000
int bcos_version(void);

/* Human-readable description of the last error on this thread ("" if none). */
const char* bcos_last_error_string(void);

/* Arithmetic of the contraction inside bcos_tapconv.  Inputs, outputs, accumulation and every stored tensor are fp32 in
 * all modes; the mode only chooses how an fp32 product is evaluated on the matrix pipe:
 *   f32     v_mfma_f32_32x32x2_f32: exact fp32 FMA chain (64 cycles per 32x32x2).
 *   bf16x3  every fp32 operand is split exactly into three bf16 slices (x = h + m + l, 3 x 8 significand bits, fp32
 *           exponent range) and a*b is evaluated with the 6 leading products on v_mfma_f32_32x32x16_bf16 (products of
 *           bf16 numbers are exact in fp32; dropped terms <= 2^-21 |a b|).  Needs nothing from the caller.
 *   f16x2   every operand is scaled by a power of two (exact) and split into two fp16 numbers, x 2^e = h + l
 *           (|x 2^e - h - l| <= 2^-22 |x|), a*b = l_a h_b + h_a l_b + h_a h_b on v_mfma_f32_32x32x16_f16: 3 matrix
 *           instructions per 16 k.  The scales are per GEMM row (taken from `a_absmax`, the per-pixel max |A| side tensor
 *           that the producer of A emitted through bcos_epilogue.out_absmax or bcos_rows_absmax) and per weight row
 *           (stored in the image made by bcos_split_weights_f16x2), and are undone exactly in the epilogue.  A call that
 *           lacks either falls back to bf16x3.  All three agree with an fp64 reference to fp32-rounding level
 *           (relL2 ~8e-7 at K = 2304); tests/test_gpu_parity.py runs every parity case in every mode.
 * The mode is chosen PER CALL by bcos_operands.contraction; BCOS_CONTRACT_DEFAULT defers to a process-wide default
 * (the only process-wide setting of the library; initial value f16x2) that bcos_set_contraction_mode changes:
 * 0 = f32, 1 = bf16x3, 2 = f16x2. */
int bcos_set_contraction_mode(int mode);
int bcos_get_contraction_mode(void);

enum { BCOS_CONTRACT_DEFAULT = 0, BCOS_CONTRACT_F32 = 1, BCOS_CONTRACT_BF16X3 = 2, BCOS_CONTRACT_F16X2 = 3 };

/* Process-wide development / test switches (ABI v7; rounds 1-3 read them from the environment on every launch).  Every
 * option selects between code paths that compute the SAME operator -- none changes what an entry point means -- and each
 * has the default a deployment wants; tests and the A/B scripts flip them through bcos_set_option.  Values are plain
 * integers held in atomics: a launch reads the table once, without locks and without touching the environment. */
enum bcos_option {
    BCOS_OPT_TAIL_SPLIT = 0,      /* 1 (default): half-height tiles fill the last round of a launch of 2-8 rounds; 0: off          */
    BCOS_OPT_D_ONE_WG = 1,        /* 0 (default): off; n > 0: LDS request of the LDS-DMA kernels that leaves room for n workgroups per CU */
    BCOS_OPT_EPI_GENERIC = 2,     /* 0 (default): specialised epilogues where a launch's feature set has one; 1: general epilogue  */
    BCOS_OPT_H2_LOOP = 3,         /* 0 (default): LDS-DMA staged split-f16 loop (tile_body_d); 1: register-staged loop (same bits) */
    BCOS_OPT_PATCH = 4,           /* 1 (default): 3 x 3 / 4 x 4-union launches with per-image maxima run over an LDS-resident input
                                     patch; 0: per-tap loops with per-row operand scales everywhere                               */
    BCOS_OPT_PATCH_WIDE = 5,      /* 1 (default): 128 x 256 patch tiles for 129..256 output columns; 0: 128 x 128                 */
    BCOS_OPT_H2_TILE = 6,         /* 0 (default): tile width by cost model; 1: force 128 x 128; 2: force 128 x 256                */
    BCOS_OPT_H2_TALL = 7,         /* 1 (default): 256-row tiles for <= 64-column launches of >= H2_TALL_MIN rows; 0: 128-row tiles */
    BCOS_OPT_H2_TALL_MIN = 8,     /* row count from which the 256 x 64 tiles are used (default 2 * 256 * 512)                      */
    BCOS_OPT_ATTENTION_F32 = 9,   /* 0 (default): attention on the f16 matrix pipe (exact 2-way splits); 1: fp32 MFMA kernel       */
    BCOS_OPT_SPLIT_LIMIT = 10,    /* bytes of A from which a split-operand launch is cut into batch chunks (default and maximum
                                     2^31: the 32-bit buffer offsets); tests lower it to exercise the chunked path                */
    BCOS_OPT_LDS_MIN_KB = 11,     /* 0 (default): off; n (<= 160): the split-f16 LDS-DMA / input-patch launches request at least n KiB of
                                     LDS -- > 80 keeps two workgroups of such a launch off one CU while a launch with a smaller request
                                     can still sit beside it (round 5 probe of matrix-bound beside bandwidth-bound launches:
                                     scripts/probe/corun2_probe.py, profiles/r05_corun_probe.txt).  (Number of an option measured in
                                     round 4 and not adopted.)                                                                            */
    BCOS_OPT_RESERVED_12 = 12,    /* reserved (value fixed at 0; any other value is BCOS_E_INVAL): number of the 2-way split of K,
                                     measured in round 4 and not adopted                                                  */
    BCOS_OPT_PATCH_LEVELS = 13,   /* 1 (default): the input-patch loop runs one pass per operand-scale level present in a tile
                                     (bcos_operands.a_imgmax); 0: level 0 only, the single per-image scale of ABI v6 -- kept so that
                                     tests can show what the ladder is for (rows far darker than their image lose accuracy)      */
    BCOS_OPT_H2_WIDE_COST = 14,   /* cost of a 128 x 256 tile in quarters of a 128 x 128 tile in the tile-width choice of the f16x2 loop
                                     (rounds of tiles on 512 slots are compared): 8 = two narrow tiles (rounds 2-3), default 7            */
    BCOS_OPT_WGRAD_WGS = 15,      /* workgroups the ordered weight gradient (bcos_conv2d_wgrad_ordered) aims for when it cuts the pixels into
                                     chunks (default 768 = 3 per CU; 64 .. 4096): more chunks = more parallel work, a larger workspace and
                                     a longer combine (ABI v9)                                                                        */
    BCOS_OPT_COUNT = 16
};
/* 0, or BCOS_E_INVAL for an unknown option or a value outside its range. */
int bcos_set_option(int option, int64_t value);
int bcos_get_option(int option, int64_t* value);

/* Operands of one bcos_tapconv launch. */
typedef struct bcos_operands {
    const float* a;             /* A: activations (forward) or gradients (dgrad), NHWC fp32                          */
    const uint32_t* a_absmax;   /* NULL or [N*H*W]: fp32 bit pattern of max_c |a[pixel, c]| (any value >= the true max
                                   and < 2^8 times it keeps full precision)                                        */
    const float* wt;            /* [Cout][taps][C] fp32; always required (fallback paths read it)                  */
    const void* wt_bf16x3;      /* NULL or the image made by bcos_split_weights                                    */
    const void* wt_f16x2;       /* NULL or the image made by bcos_split_weights_f16x2                              */
    int32_t contraction;        /* BCOS_CONTRACT_*                                                                  */
    const uint32_t* a_imgmax;   /* NULL or [N]: max over the pixels of image n of a_absmax (bcos_image_absrange; ABI v6).  With it,
                                   stride-1 3 x 3 launches (and the 4 x 4 tap union of a depth-to-space gradient) of the f16x2
                                   contraction run over an LDS-resident input patch: every input element is loaded and split once
                                   per 16 channels instead of once per tap.  A pixel then serves rows with different tap windows,
                                   so the operand scales are per IMAGE, on a ladder (ABI v7): row r of image n is contracted with
                                   the scale 2^(16 l) above the image's, l = floor((E_n - E_r) / 16) for the exponents E of the image
                                   maximum and of the row's maximum over its taps (from a_absmax; exact maxima as the epilogues and
                                   bcos_rows_absmax emit them); a tile whose rows span several levels is contracted once per level,
                                   each pass writing its own rows.  Every row is therefore computed with a scale within 2^16 of its
                                   own window's maximum: no element of it is off by more than 2^-23 of that maximum, and
                                   |lin - exact| <= 2e-6 ||patch|| ||w|| per output row -- as with the per-row scales of the other
                                   f16x2 loops -- whatever the dynamic range inside the image
                                   (tests/test_gpu_parity.py::test_patch_loop_dynamic_range_inside_an_image: measured ~2e-7).  Level,
                                   scale and result of a row are functions of its image alone: an image's bits do not depend on its
                                   batch neighbours or position.                                                              */
    const uint32_t* a_imgmin;   /* NULL or [N]: min over the NONZERO pixels of image n of a_absmax (0xffffffff: all zero;
                                   bcos_image_absrange; ABI v7).  Images whose [a_imgmin, a_imgmax] span at most 2^16 have level 0
                                   everywhere and skip the level bookkeeping; launches with >= 25 taps (7 x 7 stem) take the image
                                   maximum as the scale of such an image's rows instead of scanning every row's taps.  NULL: the
                                   range is unknown -- every tile derives its rows' levels, >= 25-tap launches scan.          */
    const uint32_t* a_imgmin_c; /* NULL or [N] (ABI v9; read only when a_imgmin is NULL): a lower bound of a_imgmin stored complemented, as
                                   the launch that produced A left it in bcos_epilogue.out_imgmin_c.  It can switch the level
                                   bookkeeping of the input-patch loop on for an image that would not have needed it, never off for one
                                   that does, and decides no bit of any result; >= 25-tap launches ignore it (they scan).              */
} bcos_operands;

/* -- contraction kernels (LDS-tiled implicit GEMM on the matrix cores) ---------------------------------------- */

/* The generic fused implicit GEMM every entry point below lowers to (bcos_tapconv / bcos_tapconv_presplit are the
 * same call with only `a`, `wt` [, `wt_bf16x3`] set). */
int bcos_tapconv_ops(const bcos_operands* ops, const bcos_tapconv_geom* geom, const bcos_epilogue* epi, void* stream);

/* Would bcos_tapconv_ops(ops, geom, epi, ...) fold the per-image range of its out_absmax into epi->out_imgmax / out_imgmin_c (1), or
 * not (0: the call then rejects the two fields), or are the arguments invalid (< 0)?  Launches nothing.  (ABI v9) */
int bcos_tapconv_fuses_image_range(const bcos_operands* ops, const bcos_tapconv_geom* geom, const bcos_epilogue* epi);
int bcos_tapconv(const float* a, const float* wt, const bcos_tapconv_geom* geom,
                 const bcos_epilogue* epi, void* stream);

/* Pre-split, pre-scaled weights for the f16x2 contraction: wt [rows][Ktot] fp32 -> image
 * [32-row tile][16-k step][plane h|l][lane][8 f16] (B fragments of v_mfma_f32_32x32x16_f16; rows padded to a multiple of
 * 128, k to a multiple of 16) followed by float[padded rows] inverse row scales.  Copied verbatim into LDS by the kernel. */
int bcos_split_weights_f16x2_bytes(int rows, int Ktot, int64_t* bytes);
int bcos_split_weights_f16x2(const float* wt, void* image, int rows, int Ktot, void* stream);
/* ... of convolution weights [rows][taps][C] (Ktot = taps * C).  For 1 < taps <= 16 and C % 16 == 0 the image stores K
 * channel-chunk-major (16-k step ks = tap ks % taps of channel chunk ks / taps), the order in which the kernel then walks
 * the taps so that a workgroup's input pixels stay cache-resident across the taps; a launch must be given the image
 * made with ITS tap count and C (bcos_tapconv_geom.TH * TW, C). */
int bcos_split_weights_f16x2_conv(const float* wt, void* image, int rows, int taps, int C, void* stream);

/* The weight banks and f16x2 images of MANY layers from ONE call -- two launches (ABI v9; a training step rebuilds every layer's forward
 * bank, its transposed / tap-reversed input-gradient banks and their images from the updated weights: ~350 small launches per ResNet-50 step).
 * Job j gathers bank[r][t][c] = c < channels ? src[r * row_stride + c * ch_stride + tap_offset[t]] : 0 (r < rows, t < taps, c < Cp) into
 * `bank` ([rows][taps][Cp] fp32) and writes `image` = what bcos_split_weights_f16x2_conv(bank, image, rows, taps, Cp) would make of it, bit
 * for bit (bcos_split_weights_f16x2_bytes(rows, taps * Cp) bytes, 16-byte aligned; NULL: the bank only).  `jobs` is a DEVICE array of
 * `njobs` descriptors -- the caller uploads it once and reuses it while the pointers in it stay valid; `max_rows` / `max_ktot` >= every
 * job's rows / taps * Cp (they size the grids); `row_max`: DEVICE scratch, ZERO-FILLED by the caller before every call, in which job j
 * owns the words [row_offset, row_offset + rows).
 *   forward bank of OIHW weights [Cout][Cin][kh][kw]:   rows = Cout, channels = Cin, row_stride = Cin kh kw, ch_stride = kh kw, tap_offset[t] = t
 *   a parity class of the input gradient:               rows = Cin, channels = Cout, row_stride = kh kw, ch_stride = Cin kh kw,
 *                                                        tap_offset[a TW + b] = rs_h[a] kw + rs_w[b] (the class's tap lists, bcos_hip/ops.py: DgradPlan)
 *   a linear layer [Cout][Cin] / its transpose:         rows = Cout / Cin, channels = Cin / Cout, strides (Cin, 1) / (1, Cin), one tap at offset 0 */
#define BCOS_PREP_MAX_TAPS 49
typedef struct bcos_weight_prep_job {
    const float* src;
    float* bank;
    void* image;
    int32_t rows, channels, Cp, taps;
    int32_t row_stride, ch_stride;
    int32_t tap_offset[BCOS_PREP_MAX_TAPS];
    int32_t row_offset;          /* the job's first word in row_max */
} bcos_weight_prep_job;
int bcos_weight_prep_batch(const bcos_weight_prep_job* jobs, int njobs, int max_rows, int max_ktot, uint32_t* row_max, void* stream);

/* out[r] = fp32 bit pattern of max_c |x[r*pitch + c]|, c < C (C % 4 == 0; pitch 0 = C): the `a_absmax` of a tensor whose
 * producer is not a bcos_tapconv epilogue (network input, pooling, attention ...).  One pass over x. */
int bcos_rows_absmax(const float* x, uint32_t* out, int64_t rows, int C, int pitch, void* stream);
/* out[n] = max over p < pixels_per_image of absmax[n * pixels_per_image + p]: bcos_operands.a_imgmax from a_absmax (ABI v6). */
int bcos_image_absmax(const uint32_t* absmax, uint32_t* out, int n_images, int pixels_per_image, void* stream);
/* ... and out_min[n] = the smallest NONZERO absmax of image n, 0xffffffff if every pixel is zero (bcos_operands.a_imgmin; ABI v7;
 * out_min may be NULL). */
int bcos_image_absrange(const uint32_t* absmax, uint32_t* out_max, uint32_t* out_min, int n_images, int pixels_per_image, void* stream);
/* The same range by several workgroups per image (ABI v9): out_max [N] and out_min_c [N] are ZERO-FILLED by the caller; out_min_c[n] receives
 * the COMPLEMENT of the minimum over the nonzero pixels (~v; 0 = no nonzero pixel) -- the form of bcos_epilogue.out_imgmin_c, handed to the
 * reading launch as bcos_operands.a_imgmin_c. */
int bcos_image_absrange_c(const uint32_t* absmax, uint32_t* out_max, uint32_t* out_min_c, int n_images, int pixels_per_image, void* stream);

/* Pre-split weights for the bf16x3 contraction.  Weights are constant at inference (NormedConv2d / BcosifyConv2d
 * weights only change in training, bcosconv2d.py:26-35), so their exact 3-way bf16 split is done once:
 *   wt [rows][Ktot] fp32 -> wt3, laid out [32-row tile][16-k step][plane h|m|l][lane][8 bf16], which is exactly the B
 *   fragment a wavefront feeds to v_mfma_f32_32x32x16_bf16 (rows padded to a multiple of 128, k to a multiple of 16).
 * bcos_tapconv_presplit then loads B straight into registers (six coalesced 16-byte loads per lane and 16-k step): no
 * conversion, no LDS traffic and no barrier participation for the weights.  Results are bit-identical to bcos_tapconv.
 * wt3 == NULL, contraction mode 0 or a narrow-output launch fall back to wt (which must always be passed). */
int bcos_split_weights_bytes(int rows, int Ktot, int64_t* bytes);
int bcos_split_weights(const float* wt, void* wt3, int rows, int Ktot, void* stream);
int bcos_tapconv_presplit(const float* a, const float* wt, const void* wt3, const bcos_tapconv_geom* geom,
                          const bcos_epilogue* epi, void* stream);

/* `count` tapconv launches over the SAME input `a` (e.g. the parity classes of a strided input gradient, one tap set,
 * weight tensor and output offset each): same result as calling bcos_tapconv once per entry.  Narrow outputs
 * (Cout <= 8, count <= 4, identical shapes, plain / addend / mul epilogue on one output tensor) run as ONE launch in
 * which every workgroup stages its input patch once for all tap sets (the ResNet stem gradient: 4 launches -> 1). */
int bcos_tapconv_group(const float* a, const float* const* wts, const bcos_tapconv_geom* geoms,
                       const bcos_epilogue* epis, int count, void* stream);

/*
 * Replaces BcosConv2d.forward_impl (bcos/modules/bcosconv2d.py:153-194) and
 * BcosifyConv2d.forward_impl (bcos/modules/bcosifyconv2d.py:50-102) for groups == 1,
 * max_out == 1: conv2d + calc_patch_norms (bcosconv2d.py:196-231) + |cos|^(B-1) scaling in
 * one pass.  x [N,H,W,Cin] NHWC, w [Cout,kh,kw,Cin], y [N,Ho,Wo,Cout] NHWC.
 * `bias` may be NULL.  `scale_out` (NULL or [N,Ho,Wo,Cout]) receives the detached dynamic
 * scale s = dy/dlin used by the explanation pass; `norm_out` (NULL or [N,Ho,Wo]) the patch
 * norms.  Unit-norm weights (NormedConv2d, bcosconv2d.py:26-35) are obtained by running
 * bcos_weight_rownorm_scale on `w` first (the projection is input independent).
 */
int bcos_conv2d_fwd(const float* x, const float* w, const float* bias, float* y,
                    float* scale_out, float* norm_out,
                    int N, int Cin, int H, int W, int Cout, int kh, int kw,
                    int sh, int sw, int ph, int pw, int dh, int dw,
                    float b, void* stream);

/*
 * Replaces BcosLinear.forward (bcos/modules/bcoslinear.py:88-130) and BcosifyLinear.forward
 * (bcos/modules/bcosifylinear.py:42-95), max_out == 1.  x [rows,Cin], w [Cout,Cin],
 * y [rows,Cout].
 */
int bcos_linear_fwd(const float* x, const float* w, const float* bias, float* y,
                    float* scale_out, float* norm_out,
                    int64_t rows, int Cin, int Cout, float b, void* stream);

/*
 * Input gradient of bcos_conv2d_fwd in explanation mode (scale detached): replaces the
 * autograd convolution_backward + mul of bcos/common.py:177.
 *     gx = conv_transpose(gy * s, w)
 * gy, s: [N,Ho,Wo,Cout]; wT: [Cin,kh,kw,Cout] = w with the spatial taps flipped and
 * Cout/Cin swapped (stride 1) -- produced by bcos_hip/ops.py: DgradPlan; for
 * stride > 1 the Python layer issues one bcos_tapconv per parity class instead.
 * `gylin` is gy*s precomputed by the caller (bcos_mul) or fused into the producer.
 */
int bcos_conv2d_dgrad_s1(const float* gylin, const float* wT, float* gx,
                         int N, int Cin, int H, int W, int Cout, int kh, int kw,
                         int ph, int pw, void* stream);

/* Input gradient of bcos_linear_fwd in explanation mode: gx[rows,Cin] = gylin[rows,Cout] @ w. */
int bcos_linear_dgrad(const float* gylin, const float* wT, float* gx,
                      int64_t rows, int Cin, int Cout, void* stream);

/* -- HBM-bound helper kernels ---------------------------------------------------------- */

/* w[r,:] *= gain[r] / ||w[r,:]||_2  (gain NULL = 1): NormedConv2d / NormedLinear unit-norm
 * projection (bcosconv2d.py:28-35, bcoslinear.py:25-27).  One wavefront per row. */
int bcos_weight_rownorm_scale(const float* w, const float* gain, float* w_out,
                              int rows, int64_t cols, void* stream);

/* inv[r] = gain[r] / ||w[r,:]||_2  (gain NULL = 1): the per-filter factor bcos_epilogue.col_scale takes when the unit-norm
 * projection is folded into the contraction instead of being written out as a projected weight tensor (one read of w, `rows`
 * floats written).  (ABI v5) */
int bcos_weight_row_invnorm(const float* w, const float* gain, float* inv, int rows, int64_t cols, void* stream);

/* out[i] = a[i] * b[i] */
int bcos_mul(const float* a, const float* b, float* out, int64_t n, void* stream);

/* dst[i] = src[i], n floats, 16-byte aligned, as the plainest streaming kernel this library can issue: every thread keeps eight
 * global_load_dwordx4 in flight and stores them with global_store_dwordx4 (non-temporal both ways), 2048 workgroups walking the
 * buffers with a grid stride.  It exists as the REFERENCE the bandwidth-bound launches are priced against (bench.py:
 * roofline.by_bound.hbm.stream_copy_gbps -- /opt/skills/guides/MI355X_MICROARCH.md measures 6.29 TB/s for such a copy), and as the
 * device-to-device copy of the engines where one is needed on a given stream.  (ABI v9) */
int bcos_stream_copy(const float* src, float* dst, int64_t n, void* stream);

/* Row-wise L2 normalisation y[r,:] = x[r,:] / ||x[r,:]||_2 and / or its inverse norms inv_norm[r] (either may be NULL): the
 * `attn_unpool` head of BcosAttentionPool2d (bcos/modules/bcosattnpool.py:23-32: x / x.norm(dim=-1), norm detached in
 * explanation mode) and `outa / outa.norm(dim=-1)` of the zero-shot attribution
 * (interpretability/analyses/text_localisation.py:76).  One wavefront per row.  (ABI v5) */
int bcos_rows_normalize(const float* x, float* y, float* inv_norm, int64_t rows, int C, void* stream);

/* Gradient of a cosine logit w.r.t. the un-normalised feature row f (text_localisation.py:76-78, 101:
 * `img_features = outa / outa.norm(...)`, `logits = img_features @ zeroshot_weight`, `logits.max(1).values.backward`):
 *   out[r,:] = coef[r] * inv_norm[r] * (w[r,:] - l[r] * u[r,:])
 * with u = f / ||f|| (given), w[r,:] = the text embedding of the explained class of row r, l[r] = u[r,:] . w[r,:], coef[r] the
 * (detached) pooling weight of the row (NULL = 1).  (ABI v5) */
int bcos_cosine_grad(const float* u, const float* w, const float* l, const float* inv_norm, const float* coef, float* out,
                     int64_t rows, int C, void* stream);

/* Backward of the fused MaxOut: glin[r, c] = gy[r, c / M] * t[r, c] with t = the scale_out of a max_out launch
 * (gy [rows, Cout / M], t and glin [rows, Cout], Cout % 4 == 0): routes the gradient to the winning filter. */
int bcos_maxout_expand(const float* gy, const float* t, float* glin, int64_t rows, int Cout, int max_out, void* stream);

/* MaxOut + B-cos scaling for the non-fused general path (max_out > 1, groups > 1):
 * lin [rows, Cout*max_out] -> y [rows, Cout] with per-row norms given (bcosconv2d.py:166-194). */
int bcos_maxout_scale(const float* lin, const float* norm, float* y, float* scale_out,
                      int32_t* argmax_out, int64_t rows, int Cout, int max_out,
                      int norm_stride, float b, void* stream);

/*
 * Network input: AddInverse (bcos/data/transforms.py:54-55, if add_inverse) + the 6-channel
 * Normalize of BcosifyNetwork (bcosify.py:38-43) + NCHW -> NHWC with channel padding.
 * x: [N,Cx,H,W] NCHW with Cx = 3 (add_inverse) or 6; out: [N,H,W,Cpad], channels >= 6 zero.
 * absmax_out (NULL or [N*H*W]): per-pixel max |out| bit patterns, the operand-scale side tensor of the f16x2
 * contraction (see bcos_operands.a_absmax) -- saves the separate bcos_rows_absmax pass over the tensor.
 */
int bcos_prep_input(const float* x, float* out, const float* mean6, const float* std6, uint32_t* absmax_out,
                    int N, int Cx, int H, int W, int Cpad, int add_inverse, void* stream);

/*
 * End of the explanation pass (bcos/common.py:180-181): from the gradient w.r.t. the
 * normalised NHWC input gxn [N,H,W,Cpad] produce the dynamic linear weights
 * W(x) = d logit / d x  [N,6,H,W] NCHW (chain rule through Normalize: / std) and the
 * contribution map sum_c x_c * W_c  [N,H,W].  x6 is the 6-channel network input
 * ([N,6,H,W], or [N,3,H,W] with add_inverse).
 */
int bcos_finalize_explanation(const float* gxn, const float* x, const float* std6,
                              float* weights_out, float* contrib_out,
                              int N, int Cx, int H, int W, int Cpad, int add_inverse,
                              void* stream);

/* (x * gx).sum(channel) for NCHW tensors: bcos/common.py:181. */
int bcos_contrib_map(const float* x, const float* gx, float* out,
                     int N, int C, int H, int W, void* stream);

/* AvgPool2d(k, s, p), count_include_pad = True (torchvision maxpool swapped for
 * nn.AvgPool2d(3,2,1): bcosification/experiment_parameters.py:99), NHWC. */
int bcos_avgpool2d_fwd(const float* x, float* y, int N, int H, int W, int C,
                       int k, int s, int p, int OH, int OW, void* stream);
/* ... also writing the per-pixel max |y| bit patterns of the pooled tensor (absmax_out [N*OH*OW], may be NULL) for the f16x2 contraction
 * that reads it: k in {2, 3}, C / 4 a power of two <= 64.  (ABI v9) */
int bcos_avgpool2d_fwd_absmax(const float* x, float* y, uint32_t* absmax_out, int N, int H, int W, int C, int k, int s, int p,
                              int OH, int OW, void* stream);
/* ... its input gradient, optionally multiplied elementwise by `mul` ([N,H,W,C]); absmax_out (NULL or [N*H*W], needs
 * C / 4 a power of two <= 64): per-pixel max |gx| bit patterns for the f16x2 contraction that reads gx. */
int bcos_avgpool2d_bwd(const float* gy, const float* mul, float* gx, uint32_t* absmax_out, int N, int H, int W, int C,
                       int k, int s, int p, int OH, int OW, void* stream);

/* AdaptiveAvgPool2d(1) + flatten + LogitLayer (bcos/modules/logitlayer.py:22-27):
 * y[n,c] = mean_hw x[n,hw,c] / T + bias. */
int bcos_global_avgpool_logits(const float* x, float* y, int N, int HW, int C,
                               float inv_temperature, float logit_bias, void* stream);

/* Start of the explanation backward for a GAP head: for every image n with explained class
 * cls[n],  glin[n,hw,c] = (c == cls[n]) * scale[n,hw,c] * inv_temperature / HW
 * (the gradient of logit cls[n] w.r.t. the head's `lin`, bcos/common.py:166-177). */
int bcos_head_onehot_grad(const int64_t* cls, const float* scale, float* glin,
                          int N, int HW, int C, float inv_temperature, void* stream);
/* The same gradient carried THROUGH the head layer in one launch: it is rank one per image (only column cls[n] of the one-hot tensor
 * is non-zero), so the input gradient of the head's linear map is  v[n, r, :] = inv_temperature / R * scale[n, r, cls[n]] *
 * row_scale[n r] * w[cls[n], :]  -- bcos_head_onehot_grad followed by the K-long input-gradient contraction, without the [N, R, K]
 * tensor.  scale [N, R, K] (the head's stored multiplier), w [K, D] (its weight rows, D % 4 == 0), row_scale [N R] or NULL (the rstd of
 * a LayerNorm folded into the head), mul [N R, D] or NULL.  out = v * mul (mul NULL: v), out2 = v (may be NULL), out_absmax [N R] or
 * NULL: per-row max |out| bit patterns.  16-byte aligned tensors.  An image whose cls[n] is outside [0, K) gets a ZERO gradient
 * (what the one-hot tensor gave; nothing is read out of range).  (ABI v8; gap-reordered SimpleViT head, vit.py:197-199) */
int bcos_head_rank1_grad(const int64_t* cls, const float* scale, const float* w, const float* row_scale, const float* mul, float* out,
                         float* out2, uint32_t* out_absmax, int N, int R, int K, int D, float inv_temperature, void* stream);
/* ... with the whole second output of the gradient epilogue (bcos_epilogue.out2, the shortcut's share behind a residual block):
 * out2 = v [* mul2], zeroed where the producing block's ReLU was closed -- gate2 [N R, D] > 0, or (gate2_from_mul) the low mantissa
 * bit of mul, as BCOS_EPI_GATE2_FROM_MUL reads it -- and its row maxima.  The GAP + fc head of the ResNets (N x 7 x 7 x 1000 one-hot
 * tensor and a K = 1000 contraction in rounds 1-4).  (ABI v8) */
int bcos_head_rank1_grad_ex(const int64_t* cls, const float* scale, const float* w, const float* row_scale, const float* mul,
                            const float* mul2, const float* gate2, int gate2_from_mul, float* out, float* out2, uint32_t* out_absmax,
                            uint32_t* out2_absmax, int N, int R, int K, int D, float inv_temperature, void* stream);

/* Row-wise arg-max over logits [N,C] -> idx [N] (int64), val [N]; ties -> lowest index
 * (torch.max semantics used at bcos/common.py:166). */
int bcos_argmax_rows(const float* x, int64_t* idx, float* val, int N, int C, void* stream);

/* BatchNormUncentered2d eval as a standalone op (module API path), NHWC:
 * y = x * scale[c] + shift[c] (shift may be NULL) (batchnorm_uncentered.py:46-60). */
int bcos_channel_affine(const float* x, const float* scale, const float* shift, float* y,
                        int64_t pixels, int C, int relu, void* stream);
/* y = [relu](x * scale[c] + shift[c] + addend): normalisation + affine + residual add + ReLU of a training-mode unit in one pass
 * (BatchNormUncentered2d with batch statistics followed by `out += identity; relu`, torchvision BasicBlock / Bottleneck.forward);
 * out = act > 0 ? g : 0, the gate of that ReLU on the way back.  (ABI v7; bcos_hip/train_plan.py) */
int bcos_channel_affine_add(const float* x, const float* scale, const float* shift, const float* addend, float* y, int64_t pixels,
                            int C, int relu, void* stream);
int bcos_relu_bwd(const float* g, const float* act, float* out, int64_t n, void* stream);

/* The same map row by row with the per-row max |y| (fp32 bit pattern, like bcos_epilogue.out_absmax) written to y_absmax [rows]:
 * a training-mode unit's output carries the operand scale of the contraction that reads it, so that the forward and input-gradient
 * contractions of a training step run the 3-product split-f16 loop.  addend / shift may be NULL; tensors 16-byte aligned.  (ABI v8) */
int bcos_channel_affine_rows(const float* x, const float* scale, const float* shift, const float* addend, float* y,
                             uint32_t* y_absmax, int64_t rows, int C, int relu, void* stream);

/* -- training-mode backward (bcos_train.hip) ------------------------------------------ */
/* Outside explanation mode the dynamic scale is not detached (bcosconv2d.py:176-194), so with lin = conv(x, W) (+ bias),
 * y = s(lin, norm) * lin:   gx = dgrad(gy * dy/dlin, W) + x (.) PatchSum^T(dL/dnorm / norm),   gW = wgrad(gy * dy/dlin, x).
 *
 * bcos_train_scale_bwd: per output pixel m (rows) and channel c, from the forward's y, s (scale_out) and norm (norm_out):
 *     glin[m,c] = gy[m,c] * dy/dlin            B == 2: 2 s                  else: s (1 + (B-1) q / (q + 1e-6)), q = |lin| / norm
 *     rnorm[m]  = sum_c gy[m,c] * dy/dnorm / (d norm / d x denominator)
 *                                              B == 2: dy/dnorm = -y / norm   else: -(B-1) y q / ((q + 1e-6) norm)
 *   with the denominator norm (BCOS_CONV_EPS: sqrt(S + 1e-6)) or norm - 1e-12 (BCOS_LINEAR_EPS: ||x|| + 1e-12).
 *   force_pow selects the general form at B == 2 (the b_loss variants, bcosifyconv2d.py:91-98).
 *   bgrad (NULL = off; general form only): *bgrad += sum_{m,c} gy y ln(q + 1e-6) = dL/dB_eff, the gradient of a learnable
 *   exponent (the reference makes `b` an nn.Parameter, bcos/training/trainer.py:451-463; bcosifyconv2d.py:60-65,91-98); the
 *   caller zeroes it and applies d B_eff / d b (clamping: [b >= 1 + 1e-6]; b_loss: 1). */
int bcos_train_scale_bwd(const float* gy, const float* y, const float* s, const float* norm, float* glin, float* rnorm,
                         float* bgrad, int64_t rows, int C, int bcos_mode, float b, int force_pow, void* stream);
/* The same with glin_absmax [rows] (may be NULL): the per-row max |glin| for the input-gradient launch that reads glin.  (ABI v8) */
int bcos_train_scale_bwd_absmax(const float* gy, const float* y, const float* s, const float* norm, float* glin, float* rnorm,
                                float* bgrad, uint32_t* glin_absmax, int64_t rows, int C, int bcos_mode, float b, int force_pow,
                                void* stream);
/* The same behind a BatchNormUncentered2d (batchnorm_uncentered.py:36-44): g_out is the gradient w.r.t. the NORM's output and the launch
 * forms the gradient w.r.t. y itself on the way -- gy = g_out * bn_g[c] + (y - bn_mean[c]) * bn_coef[c] with bn_g = weight / std and
 * bn_coef the variance-term coefficient of bcos_relu_bwd_colsums (NULL: the variance is a constant -- explanation mode or a layer in
 * eval(); bn_mean NULL: 0) -- what bcos_channel_axpby would write and this launch read back.  Channel vectors 16-byte aligned.  (ABI v8) */
int bcos_train_scale_bwd_bn(const float* g_out, const float* y, const float* s, const float* norm, const float* bn_g,
                            const float* bn_mean, const float* bn_coef, float* glin, float* rnorm, uint32_t* glin_absmax, int64_t rows,
                            int C, int bcos_mode, float b, int force_pow, void* stream);

/* Backward of bcos_weight_rownorm_scale (NormedConv2d / NormedLinear in training mode, bcosconv2d.py:26-35,
 * bcoslinear.py:25-27): with w_hat = w / ||w|| per row and w_eff = gain * w_hat,
 *     gw[r,:] = gain[r] / ||w[r]|| * (g_eff[r,:] - w_hat[r,:] <w_hat[r], g_eff[r]>),   ggain[r] = <w_hat[r], g_eff[r]>.
 * gain NULL = 1; gw or ggain may be NULL. */
int bcos_weight_rownorm_bwd(const float* w, const float* g_eff, const float* gain, float* gw, float* ggain, int rows,
                            int64_t cols, void* stream);

/* MaxOut backward by index (training mode): full[r, c * M + argmax[r, c]] = g[r, c], zero at the other filters of the unit
 * (g [rows, Cout], argmax from bcos_maxout_scale, full [rows, Cout * M]; bcosconv2d.py:166-170). */
int bcos_maxout_scatter(const float* g, const int32_t* argmax, float* full, int64_t rows, int Cout, int max_out, void* stream);

/* out[n,h,w,c] = x[n,h,w,c] * sum of rnorm[n,i,j] over the output pixels (i,j) whose patch (kernel kh x kw, stride,
 * padding, dilation) contains (h,w): the input gradient through calc_patch_norms (bcosconv2d.py:196-231).  x [N,H,W,x_pitch]
 * (C channels used, x_pitch 0 = C), rnorm [N,P,Q], out [N,H,W,C]; handed to the dgrad launch as its epilogue addend. */
int bcos_patch_norm_bwd(const float* x, const float* rnorm, float* out, int N, int H, int W, int C, int x_pitch, int P, int Q,
                        int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, void* stream);
/* The same plus a gradient that reaches the same tensor by another path: out = x * PatchSum^T(rnorm) + addend (addend [N,H,W,C] dense,
 * may be NULL) -- the shortcut gradient of a residual block folded into the patch-norm term of the block's first convolution, one
 * elementwise pass less per block and training step.  (ABI v8) */
int bcos_patch_norm_bwd_add(const float* x, const float* rnorm, const float* addend, float* out, int N, int H, int W, int C, int x_pitch,
                            int P, int Q, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, void* stream);

/* Weight gradient of a convolution / linear layer: gw[co][th][tw][ci] += sum over output pixels (n,i,j) of
 * glin[n,i,j,co] * x[n, i*sh - ph + th*dh, j*sw - pw + tw*dw, ci]   (the autograd convolution_backward weight branch).
 * glin [N,P,Q,g_pitch] (Cout used), x [N,H,W,x_pitch] (C used), gw [Cout][kh][kw][gw_cin] fp32, ZEROED BY THE CALLER
 * (partial sums of pixel chunks are combined with atomics).  Pitches 0 = dense.  Exact fp32 products on
 * v_mfma_f32_32x32x2_f32; a linear layer is the 1x1 case with N = 1, H = 1, W = rows. */
int bcos_conv2d_wgrad(const float* glin, const float* x, float* gw, int N, int H, int W, int C, int x_pitch, int P, int Q,
                      int Cout, int g_pitch, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int gw_cin,
                      void* stream);

/* The same weight gradient with a FIXED summation order (ABI v9): reproducible bit for bit from call to call.  Both operands of a
 * 16-pixel stage are split into their three bf16 planes once, at staging (pixel-contiguous LDS planes, one ds_read_b128 per fragment),
 * six v_mfma_f32_32x32x16_bf16 per product as in bcos_conv2d_wgrad; every (tile, tap, pixel chunk) workgroup stores its partial tile into
 * slab `chunk` of the workspace `ws`, and a second launch adds the slabs of every element in chunk order.  gw need NOT be zeroed (it is
 * written, not accumulated into); gw_cin must equal C (0 = C) and gw hold a multiple of 4 floats; contraction modes bf16x3 / f16x2 (mode
 * f32: BCOS_E_NOSUP -- use bcos_conv2d_wgrad).  ws: bcos_conv2d_wgrad_ws_floats() floats, 16-byte aligned (0 floats: NULL is fine). */
int bcos_conv2d_wgrad_ws_floats(int N, int H, int W, int C, int x_pitch, int P, int Q, int Cout, int g_pitch, int kh, int kw, int sh,
                                int sw, int ph, int pw, int dh, int dw, int gw_cin, int64_t* floats);
int bcos_conv2d_wgrad_ordered(const float* glin, const float* x, float* gw, float* ws, int N, int H, int W, int C, int x_pitch, int P,
                              int Q, int Cout, int g_pitch, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int gw_cin,
                              void* stream);

/* out[c] += sum_r (a[r,c] - shift_a[c]) * (b ? b[r,c] - shift_b[c] : 1), C % 4 == 0, `out` zeroed by the caller: bias
 * gradients (sum of glin) and the batch statistics / parameter gradients of BatchNormUncentered2d in training mode
 * (batchnorm_uncentered.py:36-44: var = x.var((0,2,3), unbiased=False), two passes: mean, then centred squares). */
int bcos_colsum(const float* a, const float* b, const float* shift_a, const float* shift_b, float* out, int64_t rows, int C,
                void* stream);
/* The same sums at full bandwidth AND in a fixed order, given a caller-owned workspace of bcos_colsum_ws_floats(rows, C) floats
 * (16-byte aligned): every workgroup writes its partial sums there, a second launch adds them in workgroup order and WRITES `out`
 * (no zeroing by the caller, no atomics).  The order depends on (rows, C) alone: results are bit-identical from run to run and from
 * process to process.  What the training path (BatchNormUncentered2d batch statistics, bias gradients) and synth.calibrate use.
 * (ABI v7) */
int bcos_colsum_ws_floats(int64_t rows, int C, int64_t* floats);
int bcos_colsum_ws(const float* a, const float* b, const float* shift_a, const float* shift_b, float* out, float* workspace,
                   int64_t workspace_floats, int64_t rows, int C, void* stream);
/* The same sums in a FIXED order (no atomics: one workgroup owns 64 channels over all rows, every thread walks its rows in order,
 * the partial sums meet in a fixed tree) written -- not added -- to `out`: bit-identical from run to run and from process to
 * process, at a fraction of bcos_colsum's bandwidth.  What replicas that must agree bit for bit derive their statistics with
 * (bcos_hip/synth.py: calibrate).  (ABI v7) */
int bcos_colsum_ordered(const float* a, const float* b, const float* shift_a, const float* shift_b, float* out, int64_t rows, int C,
                        void* stream);

/* BatchNormUncentered2d in training mode without separate statistics passes (ABI v8; reference batchnorm_uncentered.py:36-44,
 * trainer step bcos/training/trainer.py:666-784).  Both entry points take a caller-owned workspace of bcos_bn_train_ws_floats(rows, C)
 * floats (16-byte aligned) and sum in one fixed order per (rows, C): bit-identical from run to run.
 *   bcos_bn_batch_stats: ONE pass over y [rows, C]: mean[c], var[c] = the centred variance y.var(unbiased=False) (per-workgroup shifted
 *     sums combined as (n, mean, M2) triples), rstd = 1 / sqrt(var + eps), g = weight * rstd (weight NULL: rstd), and -- running_var not
 *     NULL -- running_var = (1 - momentum) running_var + momentum var in place.
 *   bcos_relu_bwd_colsums: ga = act > 0 ? g : 0 (act NULL: ga = g and nothing is written) together with sgx[c] = sum_r ga y and
 *     (sg not NULL) sg[c] = sum_r ga; with the forward's rstd / gvec = g also gw = sgx rstd (the weight gradient) and
 *     coef = -(gvec sgx) rstd^2 / rows (the coefficient bcos_channel_axpby takes for the variance term of the input gradient). */
int bcos_bn_train_ws_floats(int64_t rows, int C, int64_t* floats);
int bcos_bn_batch_stats(const float* y, const float* weight, float* running_var, float* mean, float* var, float* rstd, float* g,
                        float* workspace, int64_t workspace_floats, int64_t rows, int C, float eps, float momentum, void* stream);
int bcos_relu_bwd_colsums(const float* g, const float* act, const float* y, float* ga, const float* rstd, const float* gvec,
                          float* sgx, float* sg, float* gw, float* coef, float* workspace, int64_t workspace_floats, int64_t rows,
                          int C, void* stream);

/* out[r,c] = a[r,c] * sa[c] + (b[r,c] - mb[c]) * sb[c]   (b / mb / sb may be NULL: out = a * sa): the input gradient of the
 * training-mode uncentered batch norm, gx = gy * w / std + (x - mean) * coef. */
int bcos_channel_axpby(const float* a, const float* sa, const float* b, const float* mb, const float* sb, float* out,
                       int64_t rows, int C, void* stream);

/* -- transformer pieces (bcos_vit.hip) ---------------------------------------------------- */

/* LayerNorm over the last dimension D of x [rows, D] (weight / bias may be NULL); rstd_out (NULL or [rows]) keeps
 * 1/sqrt(var+eps) for the backward.  Forward of DetachableLayerNorm (bcos/modules/norms/centered_norms.py:197-224;
 * the value does not depend on explanation mode).  y_absmax (NULL or [rows], ABI v5): the fp32 bit pattern of max_c |y[r,c]|,
 * the operand scale source of the split-f16 contraction that reads y (bcos_operands.a_absmax). */
int bcos_layernorm_fwd(const float* x, const float* weight, const float* bias, float* y, float* rstd_out, uint32_t* y_absmax,
                       int64_t rows, int D, float eps, void* stream);

/* Row statistics of the same LayerNorm WITHOUT writing y: what a contraction needs to read x in place of y (bcos_epilogue.row_scale /
 * a_sumsq).  rstd_out [rows] = 1 / sqrt(var + eps); zsumsq_out (NULL or [rows]) = sum_c (weight[c] (x[r,c] - mean) rstd + bias[c])^2,
 * the squared norm of the LayerNorm output the B-cos scale of the reading layer divides by; x_absmax (NULL or [rows]): bit
 * pattern of max_c |x[r,c]| (the operand maxima of x for the split-f16 contraction).  One read of x, nothing output-sized
 * written.  (ABI v7) */
int bcos_layernorm_stats(const float* x, const float* weight, const float* bias, float* rstd_out, float* zsumsq_out,
                         uint32_t* x_absmax, int64_t rows, int D, float eps, void* stream);

/* Input gradient of DetachableLayerNorm in explanation mode (variance constant, mean differentiable,
 * centered_norms.py:204-215):  g = gy * weight * rstd - mean_D(gy * weight * rstd)  (+ addend);
 * out = g, out2 = g * mul2 (either may be NULL; mul2 NULL -> out2 = g); out2_absmax (NULL or [rows], ABI v5): row maxima of
 * out2 as in bcos_layernorm_fwd. */
int bcos_layernorm_bwd_detached(const float* gy, const float* weight, const float* rstd, const float* addend,
                                const float* mul2, float* out, float* out2, uint32_t* out2_absmax, int64_t rows, int D,
                                void* stream);

/* DetachableGroupNorm2d (bcos/modules/norms/centered_norms.py:93-160; the conv stems of the ViT-C models) on NHWC tensors
 * x [N, HW, C]: group g owns channels [g C/G, (g+1) C/G) of every pixel; per (image, group): biased variance, eps inside the
 * square root, then the per-channel affine (weight / bias may be NULL).  rstd_out (NULL or [N*G]): 1 / std per (image, group)
 * for the gradient. */
int bcos_groupnorm_fwd(const float* x, const float* weight, const float* bias, float* y, float* rstd_out,
                       int N, int HW, int C, int G, float eps, void* stream);
/* ... its explanation-mode input gradient (variance detached, mean not: centered_norms.py:118-124):
 * gx = h - mean_group(h), h = gy * weight / std. */
int bcos_groupnorm_bwd_detached(const float* gy, const float* weight, const float* rstd, float* gx,
                                int N, int HW, int C, int G, void* stream);

/* MyGELU (bcosify_vit.py:27-32): y = gate * x with gate = 0.5 (1 + erf(x / sqrt 2)); gate_out may be NULL. */
int bcos_gelu_gate(const float* x, float* y, float* gate_out, int64_t n, void* stream);

/* x[i] += pe[i % period]: sin-cos positional embedding added to the tokens (bcos/models/vit.py:324-326). */
int bcos_add_rows_bcast(float* x, const float* pe, int64_t total, int64_t period, void* stream);

/* softmax(q k^T * scale) v per (batch, head), head dim 64.  qkv [B, T, 3*H*64] ordered (q | k | v) x (h d) like
 * vit.py:145-146; out [B, T, H*64]; stats (NULL or [B, H, T, 2]) = (row max, 1 / row sum) for the backward.
 * out_absmax (NULL or [B*T], ZERO-FILLED by the caller, ABI v5): row maxima of out (one atomic max per head and row). */
int bcos_attention_fwd(const float* qkv, float* out, float* stats, uint32_t* out_absmax, int B, int T, int H, int Dh, float scale,
                       void* stream);

/* Gradient w.r.t. v with q, k detached (vit.py:148-151, bcosattnpool.py:37-39): gv = attn^T gout, attn recomputed from
 * qkv and stats. gout, gv: [B, T, H*64]; gv_absmax as out_absmax of bcos_attention_fwd. */
int bcos_attention_bwd_v(const float* qkv, const float* stats, const float* gout, float* gv, uint32_t* gv_absmax, int B, int T,
                         int H, int Dh, float scale, void* stream);

/* -- training-mode backward of the token path (nothing detached; SURVEY.md section 8(f) N4 for the ViT family) ---------- */
/* LayerNorm over the last dimension (centered_norms.py:187-245 outside explanation mode = F.layer_norm's gradient):
 * gx = rstd (h - mean(h) - x_hat mean(h x_hat)), h = gy * weight, x_hat = (x - mean) rstd; xhat_out (NULL or [rows, D])
 * receives x_hat for the weight gradient sum_rows gy x_hat (bcos_colsum). */
int bcos_layernorm_bwd(const float* gy, const float* x, const float* weight, const float* rstd, float* gx, float* xhat_out,
                       int64_t rows, int D, void* stream);
/* The same plus the gradient that reaches the LayerNorm's input by the residual connection around it: gx = (above) + addend
 * (addend [rows, D], may be NULL) -- one elementwise pass less per LayerNorm and training step.  (ABI v8) */
int bcos_layernorm_bwd_add(const float* gy, const float* x, const float* weight, const float* rstd, const float* addend, float* gx,
                           float* xhat_out, int64_t rows, int D, void* stream);
/* DetachableGroupNorm2d outside explanation mode (= F.group_norm's gradient, centered_norms.py:109-113): per (image, group)
 * gx = rstd (h - mean(h) - x_hat mean(h x_hat)), h = gy * weight; xhat_out (NULL or like x) for the affine gradients. */
int bcos_groupnorm_bwd(const float* gy, const float* x, const float* weight, const float* rstd, float* gx, float* xhat_out,
                       int N, int HW, int C, int G, void* stream);
/* MyGELU with the gate differentiated (bcosify_vit.py:27-32 with detach off): gx = gy (Phi(x) + x phi(x)). */
int bcos_gelu_bwd(const float* gy, const float* x, float* gx, int64_t n, void* stream);
/* Softmax attention with q, k, v all differentiated (vit.py:143-158 outside explanation mode): gqkv [B, T, 3*H*64] from qkv,
 * the forward's stats and output, and gout [B, T, H*64].  T <= 256.  Up to 207 tokens (16-byte aligned tensors) the five T x T x 64
 * products of a head run on the fp32 matrix pipe, one workgroup per (image, head), fixed summation order (round 5); longer sequences
 * on scalar FMA chains. */
int bcos_attention_bwd(const float* qkv, const float* stats, const float* out, const float* gout, float* gqkv,
                       int B, int T, int H, int Dh, float scale, void* stream);

/* End of the ViT explanation pass: gp [N, H/p, W/p, p, p, Cpad] (input gradient of the patch embedding, one row per
 * patch in the "(p1 p2 c)" order of vit.py:291) -> W(x) [N,6,H,W] (/ std) and contribution map [N,H,W]. */
int bcos_finalize_explanation_patches(const float* gp, const float* x, const float* std6, float* weights_out,
                                      float* contrib_out, int N, int Cx, int H, int W, int patch, int Cpad,
                                      int add_inverse, void* stream);

/* Batched RGBA rendering of explanations on the device: gradient_to_image (bcos/common.py:387-436; duplicate in
 * interpretability/analyses/text_localisation.py:106-119) for N images at once.
 *   x [N,Cx,H,W] network input (Cx = 3 with add_inverse, else 6), weights [N,6,H,W] = W(x)  ->  rgba [N,H,W,4]:
 *   rgb = pair-normalised positive weight direction, alpha = ||W||_2 (1e-12 where the contribution is negative),
 *   box-smoothed (smooth x smooth, zero padded, count_include_pad; smooth odd, 0/1 = off) and divided by its
 *   q-quantile per image (torch.quantile linear interpolation: exact order statistics by radix select), clipped to [0,1].
 * scratch: 2*N*H*W floats; quantiles: NULL or [N] (the per-image divisor). */
int bcos_render_explanations(const float* x, const float* weights, float* rgba, float* scratch, float* quantiles,
                             int N, int Cx, int H, int W, int smooth, float q, int add_inverse, void* stream);

/* -- localisation (grid pointing game) harness, SURVEY.md section 8(f) N2 ----------------------------------------- */
/* out = avg_pool2d(in, k, stride 1, padding (k-1)/2) of N maps [N,H,W] (zero padded, divisor k*k; k odd): the
 * attribution smoothing of interpretability/analyses/localisation.py:313-317.  in != out. */
int bcos_box_filter(const float* in, float* out, int N, int H, int W, int k, void* stream);

/* Per-cell share of the positive attribution (localisation.py:319-321,387-401): attr [T,H,W] -> frac [T, cells]:
 * a = clamp(neg ? -attr : attr, 0); cell means over the (H/cell_h) x (W/cell_w) grid; mean / sum of means where that is
 * positive, else 0; cell order as the reference's permute(0,1,3,2).reshape: index = col * rows + row. */
int bcos_localisation_fractions(const float* attr, float* frac, int T, int H, int W, int cell_h, int cell_w, int neg,
                                void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BCOS_HIP_H */
