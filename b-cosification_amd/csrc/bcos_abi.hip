// bcos_abi.hip -- C-ABI entry points that lower the reference-shaped operators
// (include/bcos_hip.h) onto the generic fused implicit GEMM (bcos_tapconv.hip), plus the
// library's version / error reporting.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include "bcos_hip.h"
#include "bcos_internal.h"

namespace {
thread_local char g_err[512] = "";
}

int bcos_set_error(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int bcos_set_hip_error(const char* what, hipError_t err) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(err));
    return BCOS_E_LAUNCH;
}

#ifdef BCOS_DEV_BUILD
extern "C" int bcos_version(void) { return BCOS_ABI_VERSION | BCOS_VERSION_DEV_FLAG; }
#else
extern "C" int bcos_version(void) { return BCOS_ABI_VERSION; }
#endif

// -- option table (include/bcos_hip.h: bcos_option) ------------------------------------------------------------------------
namespace {
struct OptSpec { int64_t def, lo, hi; };
constexpr int64_t TWO31 = (int64_t)1 << 31;
constexpr OptSpec OPT_SPECS[BCOS_OPT_COUNT] = {
    {1, 0, 1},                       // TAIL_SPLIT
    {0, 0, 8},                       // D_ONE_WG
    {0, 0, 1},                       // EPI_GENERIC
    {0, 0, 1},                       // H2_LOOP
    {1, 0, 1},                       // PATCH
    {1, 0, 1},                       // PATCH_WIDE
    {0, 0, 2},                       // H2_TILE
    {1, 0, 1},                       // H2_TALL
    {2 * 256 * 512, 0, TWO31},       // H2_TALL_MIN
    {0, 0, 1},                       // ATTENTION_F32
    {TWO31, 1 << 16, TWO31},         // SPLIT_LIMIT
    {0, 0, 160},                     // LDS_MIN_KB
    {0, 0, 0},                       // reserved
    {1, 0, 1},                       // PATCH_LEVELS
    {7, 4, 8},                       // H2_WIDE_COST
    {768, 64, 4096},                 // WGRAD_WGS
};
std::atomic<int64_t> g_opts[BCOS_OPT_COUNT] = {
    OPT_SPECS[0].def, OPT_SPECS[1].def, OPT_SPECS[2].def, OPT_SPECS[3].def, OPT_SPECS[4].def, OPT_SPECS[5].def, OPT_SPECS[6].def,
    OPT_SPECS[7].def, OPT_SPECS[8].def, OPT_SPECS[9].def, OPT_SPECS[10].def, OPT_SPECS[11].def, OPT_SPECS[12].def, OPT_SPECS[13].def, OPT_SPECS[14].def,
    OPT_SPECS[15].def};
static_assert(BCOS_OPT_COUNT == 16, "one OPT_SPECS row and one initialiser per option");
}  // namespace

int64_t bcos_option(int option) { return g_opts[option].load(std::memory_order_relaxed); }

extern "C" int bcos_set_option(int option, int64_t value) {
    if (option < 0 || option >= BCOS_OPT_COUNT) return bcos_set_error(BCOS_E_INVAL, "bcos_set_option: unknown option");
    if (value < OPT_SPECS[option].lo || value > OPT_SPECS[option].hi)
        return bcos_set_error(BCOS_E_INVAL, "bcos_set_option: value outside the option's range");
    g_opts[option].store(value, std::memory_order_relaxed);
    return BCOS_OK;
}

extern "C" int bcos_get_option(int option, int64_t* value) {
    if (option < 0 || option >= BCOS_OPT_COUNT || !value) return bcos_set_error(BCOS_E_INVAL, "bcos_get_option: bad argument");
    *value = bcos_option(option);
    return BCOS_OK;
}

extern "C" const char* bcos_last_error_string(void) { return g_err; }

static void zero_epilogue(bcos_epilogue* e) { memset(e, 0, sizeof(*e)); e->b = 2.0f; }

extern "C" int bcos_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, float* scale_out,
                               float* norm_out, int N, int Cin, int H, int W, int Cout, int kh, int kw, int sh,
                               int sw, int ph, int pw, int dh, int dw, float b, void* stream) {
    if (!x || !w || !y) return bcos_set_error(BCOS_E_INVAL, "bcos_conv2d_fwd: NULL tensor");
    if (sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || ph < 0 || pw < 0 || kh <= 0 || kw <= 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_conv2d_fwd: bad stride/padding/dilation/kernel");
    const int Ho = (H + 2 * ph - dh * (kh - 1) - 1) / sh + 1;
    const int Wo = (W + 2 * pw - dw * (kw - 1) - 1) / sw + 1;
    if (Ho <= 0 || Wo <= 0) return bcos_set_error(BCOS_E_INVAL, "bcos_conv2d_fwd: empty output");
    bcos_tapconv_geom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.H = H; g.W = W; g.C = Cin;
    g.P = Ho; g.Q = Wo;
    g.in_sh = sh; g.in_sw = sw;
    g.dh0 = -ph; g.dw0 = -pw;
    g.dstep_h = dh; g.dstep_w = dw;
    g.TH = kh; g.TW = kw;
    g.OH = Ho; g.OW = Wo;
    g.out_sh = 1; g.out_sw = 1;
    g.Cout = Cout;
    bcos_epilogue e;
    zero_epilogue(&e);
    e.bias = bias;
    e.out = y;
    e.scale_out = scale_out;
    e.norm_out = norm_out;
    e.bcos_mode = (b == 1.0f) ? BCOS_NONE : BCOS_CONV_EPS;
    e.b = b;
    return bcos_tapconv(x, w, &g, &e, stream);
}

extern "C" int bcos_linear_fwd(const float* x, const float* w, const float* bias, float* y, float* scale_out,
                               float* norm_out, int64_t rows, int Cin, int Cout, float b, void* stream) {
    if (!x || !w || !y) return bcos_set_error(BCOS_E_INVAL, "bcos_linear_fwd: NULL tensor");
    if (rows <= 0 || rows >= ((int64_t)1 << 31)) return bcos_set_error(BCOS_E_INVAL, "bcos_linear_fwd: bad row count");
    bcos_tapconv_geom g;
    memset(&g, 0, sizeof(g));
    g.N = 1; g.H = 1; g.W = (int)rows; g.C = Cin;
    g.P = 1; g.Q = (int)rows;
    g.in_sh = 1; g.in_sw = 1;
    g.dstep_h = 1; g.dstep_w = 1;
    g.TH = 1; g.TW = 1;
    g.OH = 1; g.OW = (int)rows;
    g.out_sh = 1; g.out_sw = 1;
    g.Cout = Cout;
    bcos_epilogue e;
    zero_epilogue(&e);
    e.bias = bias;
    e.out = y;
    e.scale_out = scale_out;
    e.norm_out = norm_out;
    e.bcos_mode = (b == 1.0f) ? BCOS_NONE : BCOS_LINEAR_EPS;
    e.b = b;
    return bcos_tapconv(x, w, &g, &e, stream);
}

extern "C" int bcos_conv2d_dgrad_s1(const float* gylin, const float* wT, float* gx, int N, int Cin, int H, int W,
                                    int Cout, int kh, int kw, int ph, int pw, void* stream) {
    if (!gylin || !wT || !gx) return bcos_set_error(BCOS_E_INVAL, "bcos_conv2d_dgrad_s1: NULL tensor");
    const int Ho = H + 2 * ph - (kh - 1);
    const int Wo = W + 2 * pw - (kw - 1);
    if (Ho <= 0 || Wo <= 0) return bcos_set_error(BCOS_E_INVAL, "bcos_conv2d_dgrad_s1: empty output");
    // gx[n,h,w,ci] = sum_{r,s,co} gylin[n, h + ph - r, w + pw - s, co] * w[co,r,s,ci]
    //             = sum_{r',s'} gylin[n, h - (kh-1-ph) + r', ...] * wT[ci, r', s', co],  r' = kh-1-r
    bcos_tapconv_geom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.H = Ho; g.W = Wo; g.C = Cout;
    g.P = H; g.Q = W;
    g.in_sh = 1; g.in_sw = 1;
    g.dh0 = -(kh - 1 - ph); g.dw0 = -(kw - 1 - pw);
    g.dstep_h = 1; g.dstep_w = 1;
    g.TH = kh; g.TW = kw;
    g.OH = H; g.OW = W;
    g.out_sh = 1; g.out_sw = 1;
    g.Cout = Cin;
    bcos_epilogue e;
    zero_epilogue(&e);
    e.out = gx;
    e.bcos_mode = BCOS_NONE;
    return bcos_tapconv(gylin, wT, &g, &e, stream);
}

extern "C" int bcos_linear_dgrad(const float* gylin, const float* wT, float* gx, int64_t rows, int Cin, int Cout,
                                 void* stream) {
    if (!gylin || !wT || !gx) return bcos_set_error(BCOS_E_INVAL, "bcos_linear_dgrad: NULL tensor");
    if (rows <= 0 || rows >= ((int64_t)1 << 31)) return bcos_set_error(BCOS_E_INVAL, "bcos_linear_dgrad: bad row count");
    bcos_tapconv_geom g;
    memset(&g, 0, sizeof(g));
    g.N = 1; g.H = 1; g.W = (int)rows; g.C = Cout;
    g.P = 1; g.Q = (int)rows;
    g.in_sh = 1; g.in_sw = 1;
    g.dstep_h = 1; g.dstep_w = 1;
    g.TH = 1; g.TW = 1;
    g.OH = 1; g.OW = (int)rows;
    g.out_sh = 1; g.out_sw = 1;
    g.Cout = Cin;
    bcos_epilogue e;
    zero_epilogue(&e);
    e.out = gx;
    e.bcos_mode = BCOS_NONE;
    return bcos_tapconv(gylin, wT, &g, &e, stream);
}
