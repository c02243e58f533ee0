/* Host-side argument validation of the C ABI under AddressSanitizer (SURVEY.md section 5 "sanitizers"; CPU build only --
 * GPU ASan is not available on the pool).  Linked against a host-sanitised build of libbcos_hip (device code
 * unsanitised, -fno-gpu-sanitize); every call below must be REJECTED by the library's own checks before any HIP call,
 * so it runs without a GPU.  Exit code 0 = every call returned the expected error and ASan saw no bad access. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bcos_hip.h"

static int failures = 0;
#define EXPECT(call, code)                                                                              \
    do {                                                                                                \
        int rc_ = (call);                                                                               \
        if (rc_ != (code)) { printf("FAIL %s -> %d (expected %d)\n", #call, rc_, (code)); ++failures; } \
        else if ((code) != BCOS_OK && strlen(bcos_last_error_string()) == 0) {                          \
            printf("FAIL %s: no error message\n", #call); ++failures; }                                 \
    } while (0)

int main(void) {
    if (bcos_version() != BCOS_ABI_VERSION) { printf("ABI version mismatch\n"); return 2; }
    /* 16-byte aligned host buffers stand in for device pointers: validation never dereferences them */
    float* buf = (float*)aligned_alloc(64, 4096);
    uint32_t* am = (uint32_t*)aligned_alloc(64, 4096);
    bcos_tapconv_geom g;
    bcos_epilogue e;
    bcos_operands o;
    memset(&g, 0, sizeof g); memset(&e, 0, sizeof e); memset(&o, 0, sizeof o);
    g.N = 1; g.H = 4; g.W = 4; g.C = 8; g.P = 4; g.Q = 4; g.in_sh = g.in_sw = 1; g.dstep_h = g.dstep_w = 1; g.TH = g.TW = 1;
    g.OH = 4; g.OW = 4; g.out_sh = g.out_sw = 1; g.Cout = 8;
    e.out = buf; e.b = 2.0f;
    o.a = buf; o.wt = buf;
    EXPECT(bcos_tapconv_ops(NULL, &g, &e, NULL), BCOS_E_INVAL);
    EXPECT(bcos_tapconv_ops(&o, NULL, &e, NULL), BCOS_E_INVAL);
    EXPECT(bcos_tapconv_ops(&o, &g, NULL, NULL), BCOS_E_INVAL);
    o.a = NULL;        EXPECT(bcos_tapconv_ops(&o, &g, &e, NULL), BCOS_E_INVAL); o.a = buf;
    o.contraction = 7; EXPECT(bcos_tapconv_ops(&o, &g, &e, NULL), BCOS_E_INVAL); o.contraction = 0;
    g.C = 6;           EXPECT(bcos_tapconv_ops(&o, &g, &e, NULL), BCOS_E_INVAL); g.C = 8;       /* C % 4 */
    g.Cout = 0;        EXPECT(bcos_tapconv_ops(&o, &g, &e, NULL), BCOS_E_INVAL); g.Cout = 8;
    o.a = buf + 1;     EXPECT(bcos_tapconv_ops(&o, &g, &e, NULL), BCOS_E_INVAL); o.a = buf;    /* misaligned operand */
    e.out = NULL;      EXPECT(bcos_tapconv_ops(&o, &g, &e, NULL), BCOS_E_INVAL); e.out = buf;  /* no output */
    g.out_h0 = 9;      EXPECT(bcos_tapconv_ops(&o, &g, &e, NULL), BCOS_E_INVAL); g.out_h0 = 0; /* mapping outside [OH,OW] */
    g.a_pitch = 6;     EXPECT(bcos_tapconv_ops(&o, &g, &e, NULL), BCOS_E_INVAL); g.a_pitch = 0;
    g.out_pitch = 4;   EXPECT(bcos_tapconv_ops(&o, &g, &e, NULL), BCOS_E_INVAL); g.out_pitch = 0;
    e.addend_sub = 2;  EXPECT(bcos_tapconv_ops(&o, &g, &e, NULL), BCOS_E_INVAL); e.addend_sub = 0;   /* subsampled addend without addend */
    e.addend_sub = -1; EXPECT(bcos_tapconv_ops(&o, &g, &e, NULL), BCOS_E_INVAL); e.addend_sub = 0;
    EXPECT(bcos_tapconv(NULL, buf, &g, &e, NULL), BCOS_E_INVAL);
    EXPECT(bcos_tapconv_presplit(buf, NULL, NULL, &g, &e, NULL), BCOS_E_INVAL);
    EXPECT(bcos_tapconv_group(buf, NULL, &g, &e, 1, NULL), BCOS_E_INVAL);
    EXPECT(bcos_set_contraction_mode(5), BCOS_E_INVAL);
    /* option table (ABI v7): unknown option, value outside its range, round trip, no environment involved */
    { int64_t v = -1;
      EXPECT(bcos_set_option(-1, 0), BCOS_E_INVAL);
      EXPECT(bcos_set_option(BCOS_OPT_COUNT, 0), BCOS_E_INVAL);
      EXPECT(bcos_set_option(BCOS_OPT_PATCH, 2), BCOS_E_INVAL);
      EXPECT(bcos_set_option(BCOS_OPT_SPLIT_LIMIT, 16), BCOS_E_INVAL);
      EXPECT(bcos_set_option(BCOS_OPT_RESERVED_12, 1), BCOS_E_INVAL);
      EXPECT(bcos_set_option(BCOS_OPT_LDS_MIN_KB, 161), BCOS_E_INVAL);
      EXPECT(bcos_set_option(BCOS_OPT_H2_WIDE_COST, 3), BCOS_E_INVAL);
      EXPECT(bcos_set_option(BCOS_OPT_WGRAD_WGS, 32), BCOS_E_INVAL);
      EXPECT(bcos_get_option(BCOS_OPT_PATCH, NULL), BCOS_E_INVAL);
      EXPECT(bcos_get_option(BCOS_OPT_COUNT, &v), BCOS_E_INVAL);
      setenv("BCOS_PATCH", "0", 1);                         /* rounds 1-3 read this on every launch */
      EXPECT(bcos_get_option(BCOS_OPT_PATCH, &v), BCOS_OK);
      if (v != 1) { printf("FAIL default of BCOS_OPT_PATCH %lld\n", (long long)v); ++failures; }
      EXPECT(bcos_set_option(BCOS_OPT_PATCH, 0), BCOS_OK);
      EXPECT(bcos_get_option(BCOS_OPT_PATCH, &v), BCOS_OK);
      if (v != 0) { printf("FAIL BCOS_OPT_PATCH round trip %lld\n", (long long)v); ++failures; }
      EXPECT(bcos_set_option(BCOS_OPT_PATCH, 1), BCOS_OK); }
    EXPECT(bcos_image_absrange(NULL, am, am, 1, 4, NULL), BCOS_E_INVAL);
    EXPECT(bcos_channel_affine_add(buf, buf, NULL, NULL, buf, 4, 8, 1, NULL), BCOS_E_INVAL);      /* no addend */
    EXPECT(bcos_channel_affine_add(buf, buf, NULL, buf, buf, 4, 6, 1, NULL), BCOS_E_INVAL);       /* C % 4 */
    EXPECT(bcos_relu_bwd(buf, NULL, buf, 16, NULL), BCOS_E_INVAL);
    EXPECT(bcos_relu_bwd(buf, buf, buf, 6, NULL), BCOS_E_INVAL);
    EXPECT(bcos_colsum_ordered(NULL, NULL, NULL, NULL, buf, 4, 8, NULL), BCOS_E_INVAL);
    { int64_t nf = 0;
      EXPECT(bcos_colsum_ws_floats(0, 8, &nf), BCOS_E_INVAL);
      EXPECT(bcos_colsum_ws_floats(1000, 8, NULL), BCOS_E_INVAL);
      EXPECT(bcos_colsum_ws_floats(1000, 8, &nf), BCOS_OK);
      if (nf != 4 * 8) { printf("FAIL colsum workspace %lld\n", (long long)nf); ++failures; }
      EXPECT(bcos_colsum_ws(buf, NULL, NULL, NULL, buf, NULL, nf, 1000, 8, NULL), BCOS_E_INVAL);          /* no workspace */
      EXPECT(bcos_colsum_ws(buf, NULL, NULL, NULL, buf, buf, nf - 1, 1000, 8, NULL), BCOS_E_INVAL);       /* workspace too small */
      EXPECT(bcos_colsum_ws(buf, NULL, NULL, NULL, buf, buf + 1, nf, 1000, 8, NULL), BCOS_E_INVAL); }     /* misaligned workspace */
    EXPECT(bcos_colsum_ordered(buf, NULL, NULL, NULL, buf, 4, 6, NULL), BCOS_E_INVAL);            /* C % 4 */
    EXPECT(bcos_colsum_ordered(buf + 1, NULL, NULL, NULL, buf, 4, 8, NULL), BCOS_E_INVAL);        /* misaligned */
    EXPECT(bcos_image_absrange(am, NULL, am, 1, 4, NULL), BCOS_E_INVAL);
    EXPECT(bcos_image_absrange(am, am, NULL, 0, 4, NULL), BCOS_E_INVAL);
    { int64_t nb = 0;
      EXPECT(bcos_split_weights_bytes(0, 16, &nb), BCOS_E_INVAL);
      EXPECT(bcos_split_weights_bytes(64, 64, &nb), BCOS_OK);
      if (nb != 4 * 4 * 3 * 1024) { printf("FAIL split bytes %lld\n", (long long)nb); ++failures; }
      EXPECT(bcos_split_weights_f16x2_bytes(64, 64, NULL), BCOS_E_INVAL);
      EXPECT(bcos_split_weights_f16x2_bytes(200, 72, &nb), BCOS_OK);
      if (nb != 8 * 5 * 2 * 1024 + 8 * 32 * 4) { printf("FAIL f16x2 bytes %lld\n", (long long)nb); ++failures; } }
    EXPECT(bcos_split_weights(buf, (char*)buf + 4, 8, 16, NULL), BCOS_E_INVAL);                   /* misaligned image */
    EXPECT(bcos_split_weights_f16x2_conv(buf, buf, 8, 0, 16, NULL), BCOS_E_INVAL);
    EXPECT(bcos_rows_absmax(buf, am, 4, 6, 0, NULL), BCOS_E_INVAL);                               /* C % 4 */
    EXPECT(bcos_rows_absmax(NULL, am, 4, 8, 0, NULL), BCOS_E_INVAL);
    EXPECT(bcos_image_absmax(NULL, am, 2, 4, NULL), BCOS_E_INVAL);                                /* ABI v6 */
    EXPECT(bcos_image_absmax(am, am, 0, 4, NULL), BCOS_E_INVAL);
    EXPECT(bcos_image_absmax(am, am, 2, 0, NULL), BCOS_E_INVAL);
    EXPECT(bcos_conv2d_fwd(buf, buf, NULL, buf, NULL, NULL, 1, 8, 4, 4, 8, 3, 3, 0, 1, 1, 1, 1, 1, 2.0f, NULL), BCOS_E_INVAL);
    EXPECT(bcos_conv2d_fwd(NULL, buf, NULL, buf, NULL, NULL, 1, 8, 4, 4, 8, 3, 3, 1, 1, 1, 1, 1, 1, 2.0f, NULL), BCOS_E_INVAL);
    EXPECT(bcos_linear_fwd(buf, NULL, NULL, buf, NULL, NULL, 4, 8, 8, 2.0f, NULL), BCOS_E_INVAL);
    EXPECT(bcos_train_scale_bwd(buf, buf, buf, buf, buf, buf, NULL, 4, 6, BCOS_CONV_EPS, 2.0f, 0, NULL), BCOS_E_INVAL);
    EXPECT(bcos_train_scale_bwd(buf, buf, buf, buf, buf, buf, NULL, 4, 8, BCOS_NONE, 2.0f, 0, NULL), BCOS_E_INVAL);
    EXPECT(bcos_train_scale_bwd(buf, buf, buf, buf, buf, buf, NULL, 4, 8, BCOS_CONV_EPS, 1.0f, 0, NULL), BCOS_E_INVAL);
    EXPECT(bcos_train_scale_bwd(buf, buf, buf, buf, buf, buf, buf, 4, 8, BCOS_CONV_EPS, 2.0f, 0, NULL), BCOS_E_INVAL);   /* bgrad without the pow form */
    EXPECT(bcos_weight_rownorm_bwd(NULL, buf, NULL, buf, NULL, 4, 8, NULL), BCOS_E_INVAL);
    EXPECT(bcos_weight_rownorm_bwd(buf, buf, NULL, NULL, NULL, 4, 8, NULL), BCOS_E_INVAL);                                /* no output */
    EXPECT(bcos_maxout_scatter(buf, NULL, buf, 4, 8, 2, NULL), BCOS_E_INVAL);
    e.a_sumsq = buf; EXPECT(bcos_tapconv_ops(&o, &g, &e, NULL), BCOS_E_INVAL); e.a_sumsq = NULL;                          /* a_sumsq without a B-cos mode */
    g.out_cgroup = 4; EXPECT(bcos_tapconv_ops(&o, &g, &e, NULL), BCOS_E_INVAL); g.out_cgroup = 0;                         /* Cout != out_sh*out_sw*G */
    EXPECT(bcos_patch_norm_bwd(buf, buf, NULL, 1, 4, 4, 8, 0, 4, 4, 1, 1, 1, 1, 0, 0, 1, 1, NULL), BCOS_E_INVAL);
    EXPECT(bcos_conv2d_wgrad(buf, buf, buf, 1, 4, 4, 8, 6, 4, 4, 8, 0, 1, 1, 1, 1, 0, 0, 1, 1, 0, NULL), BCOS_E_INVAL);
    EXPECT(bcos_colsum(buf, NULL, NULL, NULL, buf, 4, 6, NULL), BCOS_E_INVAL);
    EXPECT(bcos_channel_axpby(buf, buf, buf, NULL, NULL, buf, 4, 8, NULL), BCOS_E_INVAL);
    /* ABI v8 entry points (training plan) */
    { int64_t n = 0;
      EXPECT(bcos_bn_train_ws_floats(0, 8, &n), BCOS_E_INVAL);
      EXPECT(bcos_bn_train_ws_floats(64, 6, &n), BCOS_E_INVAL);
      EXPECT(bcos_bn_train_ws_floats(64, 8, NULL), BCOS_E_INVAL); }
    EXPECT(bcos_bn_batch_stats(NULL, NULL, NULL, buf, buf, buf, buf, buf, 1 << 20, 64, 8, 1e-5f, 0.1f, NULL), BCOS_E_INVAL);
    EXPECT(bcos_bn_batch_stats(buf, NULL, NULL, buf, buf, buf, buf, buf, 1 << 20, 64, 6, 1e-5f, 0.1f, NULL), BCOS_E_INVAL);     /* C % 4 */
    EXPECT(bcos_bn_batch_stats(buf, NULL, NULL, buf, buf, buf, buf, buf, 3, 64, 8, 1e-5f, 0.1f, NULL), BCOS_E_INVAL);           /* workspace too small */
    EXPECT(bcos_relu_bwd_colsums(buf, buf, buf, NULL, NULL, NULL, buf, NULL, NULL, NULL, buf, 1 << 20, 64, 8, NULL), BCOS_E_INVAL);   /* act without ga */
    EXPECT(bcos_relu_bwd_colsums(buf, NULL, buf, NULL, NULL, NULL, buf, NULL, buf, NULL, buf, 1 << 20, 64, 8, NULL), BCOS_E_INVAL);   /* gw without rstd */
    EXPECT(bcos_relu_bwd_colsums(buf, NULL, buf, NULL, buf, NULL, buf, NULL, NULL, buf, buf, 1 << 20, 64, 8, NULL), BCOS_E_INVAL);    /* coef without gvec */
    EXPECT(bcos_channel_affine_rows(buf, buf, NULL, NULL, buf, NULL, 4, 8, 1, NULL), BCOS_E_INVAL);                                  /* maxima are the point */
    EXPECT(bcos_channel_affine_rows(buf, buf, NULL, NULL, buf, am, 4, 6, 1, NULL), BCOS_E_INVAL);
    EXPECT(bcos_train_scale_bwd_absmax(buf, buf, buf, buf, buf, buf, NULL, am, 4, 6, BCOS_CONV_EPS, 2.0f, 0, NULL), BCOS_E_INVAL);
    EXPECT(bcos_train_scale_bwd_bn(buf, buf, buf, buf, NULL, NULL, NULL, buf, buf, am, 4, 8, BCOS_CONV_EPS, 2.0f, 0, NULL), BCOS_E_INVAL);      /* no bn_g */
    EXPECT(bcos_train_scale_bwd_bn(buf, buf, buf, buf, buf, buf, NULL, buf, buf, am, 4, 8, BCOS_CONV_EPS, 2.0f, 0, NULL), BCOS_E_INVAL);        /* mean without coef */
    EXPECT(bcos_train_scale_bwd_bn(buf, buf, buf, buf, buf, NULL, buf + 1, buf, buf, am, 4, 8, BCOS_CONV_EPS, 2.0f, 0, NULL), BCOS_E_INVAL);   /* misaligned vector */
    EXPECT(bcos_train_scale_bwd_bn(buf, buf, buf, buf, buf, NULL, NULL, buf, buf, am, 4, 8, BCOS_CONV_EPS, 1.0f, 0, NULL), BCOS_E_INVAL);       /* B == 1 */
    EXPECT(bcos_patch_norm_bwd_add(buf, buf, buf, NULL, 1, 4, 4, 8, 0, 4, 4, 1, 1, 1, 1, 0, 0, 1, 1, NULL), BCOS_E_INVAL);
    /* ABI v5 entry points */
    EXPECT(bcos_rows_normalize(NULL, buf, NULL, 4, 8, NULL), BCOS_E_INVAL);
    EXPECT(bcos_rows_normalize(buf, NULL, NULL, 4, 8, NULL), BCOS_E_INVAL);                       /* neither output */
    EXPECT(bcos_cosine_grad(buf, buf, NULL, buf, NULL, buf, 4, 8, NULL), BCOS_E_INVAL);
    EXPECT(bcos_weight_row_invnorm(buf, NULL, NULL, 4, 8, NULL), BCOS_E_INVAL);
    EXPECT(bcos_weight_row_invnorm(buf, NULL, buf, 0, 8, NULL), BCOS_E_INVAL);
    EXPECT(bcos_layernorm_fwd(NULL, NULL, NULL, buf, NULL, am, 4, 8, 1e-5f, NULL), BCOS_E_INVAL);
    EXPECT(bcos_layernorm_stats(buf, NULL, NULL, NULL, buf, am, 4, 8, 1e-5f, NULL), BCOS_E_INVAL);               /* rstd_out is required */
    EXPECT(bcos_layernorm_stats(NULL, NULL, NULL, buf, NULL, NULL, 4, 8, 1e-5f, NULL), BCOS_E_INVAL);
    EXPECT(bcos_layernorm_bwd_detached(buf, NULL, buf, NULL, NULL, buf, NULL, am, 4, 8, NULL), BCOS_E_INVAL);   /* maxima of an absent out2 */
    EXPECT(bcos_head_rank1_grad(NULL, buf, buf, NULL, NULL, buf, NULL, am, 2, 4, 10, 8, 1.0f, NULL), BCOS_E_INVAL);    /* no class indices */
    EXPECT(bcos_head_rank1_grad((const int64_t*)buf, buf, buf, NULL, NULL, buf, NULL, am, 2, 4, 10, 6, 1.0f, NULL), BCOS_E_INVAL);   /* D % 4 */
    EXPECT(bcos_head_rank1_grad((const int64_t*)buf, buf, buf + 1, NULL, NULL, buf, NULL, am, 2, 4, 10, 8, 1.0f, NULL), BCOS_E_INVAL);   /* misaligned w */
    EXPECT(bcos_head_rank1_grad_ex((const int64_t*)buf, buf, buf, NULL, buf, buf, NULL, 0, buf, NULL, am, NULL, 2, 4, 10, 8, 1.0f, NULL), BCOS_E_INVAL);   /* mul2 without out2 */
    EXPECT(bcos_head_rank1_grad_ex((const int64_t*)buf, buf, buf, NULL, NULL, NULL, NULL, 1, buf, buf, am, am, 2, 4, 10, 8, 1.0f, NULL), BCOS_E_INVAL);    /* gate from an absent mul */
    /* ABI v9 */
    { int64_t fl = -1;
      EXPECT(bcos_conv2d_wgrad_ws_floats(8, 56, 56, 64, 0, 56, 56, 64, 0, 3, 3, 1, 1, 1, 1, 1, 1, 0, NULL), BCOS_E_INVAL);
      EXPECT(bcos_conv2d_wgrad_ws_floats(8, 56, 56, 64, 0, 56, 56, 64, 0, 0, 3, 1, 1, 1, 1, 1, 1, 0, &fl), BCOS_E_INVAL);      /* kh = 0 */
      EXPECT(bcos_conv2d_wgrad_ws_floats(8, 56, 56, 64, 0, 56, 56, 64, 0, 3, 3, 1, 1, 1, 1, 1, 1, 0, &fl), BCOS_OK);
      if (fl <= 0 || fl % (64 * 9 * 64) != 0) { printf("FAIL wgrad workspace size %lld\n", (long long)fl); ++failures; }
      EXPECT(bcos_conv2d_wgrad_ordered(NULL, buf, buf, buf, 8, 56, 56, 64, 0, 56, 56, 64, 0, 3, 3, 1, 1, 1, 1, 1, 1, 0, NULL), BCOS_E_INVAL);
      EXPECT(bcos_conv2d_wgrad_ordered(buf, buf, buf, NULL, 8, 56, 56, 64, 0, 56, 56, 64, 0, 3, 3, 1, 1, 1, 1, 1, 1, 0, NULL), BCOS_E_INVAL);   /* needs a workspace */
      EXPECT(bcos_conv2d_wgrad_ordered(buf, buf, buf + 1, buf, 8, 56, 56, 64, 0, 56, 56, 64, 0, 3, 3, 1, 1, 1, 1, 1, 1, 0, NULL), BCOS_E_INVAL); /* misaligned gw */
      EXPECT(bcos_conv2d_wgrad_ordered(buf, buf, buf, buf, 8, 56, 56, 64, 0, 56, 56, 64, 0, 3, 3, 1, 1, 1, 1, 1, 1, 68, NULL), BCOS_E_NOSUP);   /* padded gw */ }
    EXPECT(bcos_tapconv_fuses_image_range(NULL, &g, &e), BCOS_E_INVAL);
    EXPECT(bcos_tapconv_fuses_image_range(&o, &g, &e), 0);               /* no out_absmax: nothing to fold */
    { uint32_t* img = am + 64; e.out_imgmax = img; e.out_imgmin_c = img + 8;        /* the pair without out_absmax / outside a fusing launch */
      EXPECT(bcos_tapconv_ops(&o, &g, &e, NULL), BCOS_E_NOSUP); e.out_imgmin_c = NULL;
      EXPECT(bcos_tapconv_ops(&o, &g, &e, NULL), BCOS_E_NOSUP); e.out_imgmax = NULL; }
    { e.rowadd = buf;                                                               /* a row-scaled addend without its scales */
      EXPECT(bcos_tapconv_ops(&o, &g, &e, NULL), BCOS_E_NOSUP); e.rowadd = NULL; }
    EXPECT(bcos_weight_prep_batch(NULL, 1, 64, 64, am, NULL), BCOS_E_INVAL);
    EXPECT(bcos_weight_prep_batch((const bcos_weight_prep_job*)buf, 1, 64, 64, NULL, NULL), BCOS_E_INVAL);     /* no scratch for the rows' maxima */
    EXPECT(bcos_weight_prep_batch((const bcos_weight_prep_job*)buf, 0, 64, 64, am, NULL), BCOS_E_INVAL);
    EXPECT(bcos_weight_prep_batch((const bcos_weight_prep_job*)buf, 1, 0, 64, am, NULL), BCOS_E_INVAL);
    if (sizeof(bcos_weight_prep_job) != 248) { printf("FAIL sizeof(bcos_weight_prep_job) = %zu\n", sizeof(bcos_weight_prep_job)); ++failures; }
    EXPECT(bcos_stream_copy(NULL, buf, 16, NULL), BCOS_E_INVAL);
    EXPECT(bcos_stream_copy(buf, buf + 1, 16, NULL), BCOS_E_INVAL);      /* misaligned */
    EXPECT(bcos_stream_copy(buf, buf, 6, NULL), BCOS_E_INVAL);           /* n % 4 */
    EXPECT(bcos_stream_copy(buf, buf, -4, NULL), BCOS_E_INVAL);
    EXPECT(bcos_stream_copy(buf, buf, 0, NULL), BCOS_OK);                /* nothing to do: no launch */
    EXPECT(bcos_layernorm_bwd_add(buf, buf, NULL, buf, buf, NULL, NULL, 4, 8, NULL), BCOS_E_INVAL);                    /* no output */
    EXPECT(bcos_layernorm_bwd_add(buf, buf, NULL, NULL, buf, buf, NULL, 4, 8, NULL), BCOS_E_INVAL);                    /* no rstd */
    EXPECT(bcos_attention_fwd(buf, NULL, NULL, am, 1, 4, 1, 64, 1.0f, NULL), BCOS_E_INVAL);
    EXPECT(bcos_attention_bwd_v(buf, buf, buf, buf, am, 1, 4, 1, 32, 1.0f, NULL), BCOS_E_NOSUP);               /* head dimension */
    free(buf); free(am);
    if (failures) { printf("%d failure(s)\n", failures); return 1; }
    printf("abi_validation: ok\n");
    return 0;
}
