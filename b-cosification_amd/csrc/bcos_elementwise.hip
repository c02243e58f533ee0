// bcos_elementwise.hip -- the HBM-bound kernels of the B-cos forward / explanation path on
// gfx950: input preparation, pooling, the classification head, the end of the explanation
// pass and the unit-norm weight projection.  All are streaming kernels: 16-byte accesses
// where the layout allows, 256-thread blocks, grid-stride loops capped at 256 CUs x 8 blocks.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bcos_hip.h"
#include "bcos_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TPB = 256;
constexpr int64_t MAX_BLOCKS = 256 * 8;

// one-pixel-per-thread kernels whose passes are chains of dependent loads and stores (prep_input, finalize_explanation): more, shorter
// threads keep more independent memory operations in flight than MAX_BLOCKS workgroups walking 25 passes each
inline unsigned grid_pixels(int64_t pixels) {
    int64_t b = (pixels + TPB - 1) / TPB;
    if (b < 1) b = 1;
    if (b > 256 * 64) b = 256 * 64;
    return (unsigned)b;
}

inline unsigned grid_for(int64_t work_items) {
    int64_t b = (work_items + TPB - 1) / TPB;
    if (b < 1) b = 1;
    if (b > MAX_BLOCKS) b = MAX_BLOCKS;
    return (unsigned)b;
}

inline int check_launch(const char* what) {
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return bcos_set_hip_error(what, err);
    return BCOS_OK;
}

// ---- unit-norm weight projection: one wavefront per row --------------------------------
__global__ __launch_bounds__(TPB) void weight_rownorm_kernel(const float* __restrict__ w,
                                                             const float* __restrict__ gain,
                                                             float* __restrict__ out, int rows, int64_t cols) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * TPB + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * TPB) >> 6;
    for (int r = wave; r < rows; r += nwaves) {
        const float* src = w + (int64_t)r * cols;
        float ss = 0.f;
        for (int64_t c = lane; c < cols; c += 64) ss = fmaf(src[c], src[c], ss);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        const float nrm = sqrtf(ss);
        const float gn = gain ? gain[r] : 1.f;
        float* dst = out + (int64_t)r * cols;
        for (int64_t c = lane; c < cols; c += 64) {
            float v = src[c] / nrm;
            dst[c] = gain ? gn * v : v;
        }
    }
}

__global__ __launch_bounds__(TPB) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, int64_t n4, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * TPB;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += stride) {
        f32x4 x = reinterpret_cast<const f32x4*>(a)[i];
        f32x4 y = reinterpret_cast<const f32x4*>(b)[i];
        reinterpret_cast<f32x4*>(out)[i] = x * y;
    }
    // tail
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) out[i] = a[i] * b[i];
}

__device__ inline float bcos_scale_of(float lin, float nrm, float b) {
    if (b == 2.0f) return fabsf(lin) / nrm;
    return powf(fabsf(lin / nrm) + 1e-6f, b - 1.0f);
}

// ---- MaxOut + scaling, general path ----------------------------------------------------------
__global__ __launch_bounds__(TPB) void maxout_scale_kernel(const float* __restrict__ lin,
                                                           const float* __restrict__ norm, float* __restrict__ y,
                                                           float* __restrict__ scale_out,
                                                           int32_t* __restrict__ argmax_out, int64_t rows, int Cout,
                                                           int max_out, int norm_stride, float b) {
    const int64_t total = rows * Cout;
    const int64_t stride = (int64_t)gridDim.x * TPB;
    const int per_group = Cout / norm_stride;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
        const int64_t r = i / Cout;
        const int c = (int)(i - r * Cout);
        const float* src = lin + r * (int64_t)Cout * max_out + (int64_t)c * max_out;
        float best = src[0];
        int arg = 0;
        for (int m = 1; m < max_out; ++m) {
            const float v = src[m];
            if (v > best) { best = v; arg = m; }
        }
        float s = 1.f;
        if (norm) {
            const float nrm = norm[r * norm_stride + c / per_group];
            s = bcos_scale_of(best, nrm, b);
        }
        y[i] = s * best;
        if (scale_out) scale_out[i] = s;
        if (argmax_out) argmax_out[i] = arg;
    }
}

// ---- network input: AddInverse + Normalize + NCHW -> NHWC (padded) -----------------------------
__global__ __launch_bounds__(TPB) void prep_input_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                         const float* __restrict__ mean6,
                                                         const float* __restrict__ std6, unsigned* __restrict__ absmax,
                                                         int N, int Cx, int HW, int Cpad, int add_inverse) {
    const int64_t total = (int64_t)N * HW;
    const int64_t stride = (int64_t)gridDim.x * TPB;
    float mu[6], sd[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) { mu[c] = mean6[c]; sd[c] = std6[c]; }
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
        const int64_t n = i / HW;
        const int64_t hw = i - n * HW;
        const float* src = x + n * (int64_t)Cx * HW + hw;
        float v[6];
        if (add_inverse) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { v[c] = src[(int64_t)c * HW]; v[c + 3] = 1.0f - v[c]; }
        } else {
#pragma unroll
            for (int c = 0; c < 6; ++c) v[c] = src[(int64_t)c * HW];
        }
        float* dst = out + i * Cpad;
        unsigned m = 0u;
        float o[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            o[c] = (v[c] - mu[c]) / sd[c];
            m = max(m, __float_as_uint(o[c]) & 0x7fffffffu);
        }
        if (Cpad == 8 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {        // the usual padded pixel: two 16-byte stores
            reinterpret_cast<f32x4*>(dst)[0] = f32x4{o[0], o[1], o[2], o[3]};
            reinterpret_cast<f32x4*>(dst)[1] = f32x4{o[4], o[5], 0.f, 0.f};
        } else {
#pragma unroll
            for (int c = 0; c < 6; ++c) dst[c] = o[c];
            for (int c = 6; c < Cpad; ++c) dst[c] = 0.f;
        }
        if (absmax) absmax[i] = m;          // per-pixel max |value| bit pattern (operand scale source of the f16x2 contraction)
    }
}

// ---- end of the explanation pass -----------------------------------------------------------------
__global__ __launch_bounds__(TPB) void finalize_expl_kernel(const float* __restrict__ gxn,
                                                            const float* __restrict__ x,
                                                            const float* __restrict__ std6,
                                                            float* __restrict__ wout, float* __restrict__ cout,
                                                            int N, int Cx, int HW, int Cpad, int add_inverse) {
    const int64_t total = (int64_t)N * HW;
    const int64_t stride = (int64_t)gridDim.x * TPB;
    float sd[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) sd[c] = std6[c];
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
        const int64_t n = i / HW;
        const int64_t hw = i - n * HW;
        const float* g = gxn + i * Cpad;
        const float* src = x + n * (int64_t)Cx * HW + hw;
        float gv[6];
        if (Cpad == 8 && (reinterpret_cast<uintptr_t>(gxn) & 15) == 0) {        // the usual padded pixel: two 16-byte loads
            const f32x4 a = reinterpret_cast<const f32x4*>(g)[0], b = reinterpret_cast<const f32x4*>(g)[1];
            gv[0] = a[0]; gv[1] = a[1]; gv[2] = a[2]; gv[3] = a[3]; gv[4] = b[0]; gv[5] = b[1];
        } else {
#pragma unroll
            for (int c = 0; c < 6; ++c) gv[c] = g[c];
        }
        float contrib = 0.f;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const float wv = gv[c] / sd[c];
            float xv;
            if (add_inverse) xv = c < 3 ? src[(int64_t)c * HW] : 1.0f - src[(int64_t)(c - 3) * HW];
            else xv = src[(int64_t)c * HW];
            if (wout) wout[(n * 6 + c) * (int64_t)HW + hw] = wv;
            contrib += xv * wv;   // same order as torch .sum(1) over 6 channels is not guaranteed; 6 terms
        }
        if (cout) cout[i] = contrib;
    }
}

__global__ __launch_bounds__(TPB) void contrib_map_kernel(const float* __restrict__ x, const float* __restrict__ gx,
                                                          float* __restrict__ out, int N, int C, int HW) {
    const int64_t total = (int64_t)N * HW;
    const int64_t stride = (int64_t)gridDim.x * TPB;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
        const int64_t n = i / HW;
        const int64_t hw = i - n * HW;
        const int64_t base = n * (int64_t)C * HW + hw;
        float acc = 0.f;
        for (int c = 0; c < C; ++c) acc += x[base + (int64_t)c * HW] * gx[base + (int64_t)c * HW];
        out[i] = acc;
    }
}

// ---- AvgPool2d (count_include_pad=True), NHWC, 4 channels per thread -------------------------------
__global__ __launch_bounds__(TPB) void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N,
                                                          int H, int W, int C4, int k, int s, int p, int OH,
                                                          int OW) {
    const int64_t total = (int64_t)N * OH * OW * C4;
    const int64_t stride = (int64_t)gridDim.x * TPB;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
        const int c4 = (int)(i % C4);
        int64_t t = i / C4;
        const int ow = (int)(t % OW); t /= OW;
        const int oh = (int)(t % OH);
        const int64_t n = t / OH;
        int hs = oh * s - p, ws = ow * s - p;
        int he = min(hs + k, H + p), we = min(ws + k, W + p);
        const float pool = (float)((he - hs) * (we - ws));
        hs = max(hs, 0); ws = max(ws, 0); he = min(he, H); we = min(we, W);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int h = hs; h < he; ++h)
            for (int w = ws; w < we; ++w)
                acc += reinterpret_cast<const f32x4*>(x)[((n * H + h) * W + w) * C4 + c4];
        reinterpret_cast<f32x4*>(y)[i] = acc / pool;
    }
}

// The same pool with one workgroup per OUTPUT row (n, oh) and the window expanded at compile time (K x K predicated loads, all in flight
// together; out-of-image taps add +0, which leaves every partial sum as it was): 32-bit index arithmetic, optional per-pixel maxima of
// the result for the f16x2 contraction that reads it (C4 a power of two <= 64: the C4 lanes of a pixel are an aligned lane group) --
// the stem pool of the ResNets ran the flat kernel above at 4.7 TB/s and a separate pass for the maxima.  Same sums in the same
// order as avgpool_fwd_kernel: bit-identical.
template <int K>
__global__ __launch_bounds__(TPB) void avgpool_fwd_row_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned* __restrict__ absmax,
                                                              int H, int W, int C4, int s, int p, int OH, int OW) {
    const int n = blockIdx.x / OH, oh = blockIdx.x - n * OH;
    const int hs = oh * s - p;
    const int he = min(hs + K, H + p);
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x) + (int64_t)n * H * W * C4;
    f32x4* y4 = reinterpret_cast<f32x4*>(y) + (int64_t)blockIdx.x * OW * C4;
    const int c4_shift = (C4 & (C4 - 1)) == 0 ? __builtin_ctz(C4) : -1;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (int i0 = 0; i0 < OW * C4; i0 += TPB) {
        const int i = i0 + threadIdx.x;
        const bool live = i < OW * C4;
        const int ow = live ? (c4_shift >= 0 ? i >> c4_shift : i / C4) : 0, c4 = live ? i - ow * C4 : 0;
        const int ws = ow * s - p;
        const int we = min(ws + K, W + p);
        const float pool = (float)((he - hs) * (we - ws));
        f32x4 v[K][K];
#pragma unroll
        for (int dh = 0; dh < K; ++dh)
#pragma unroll
            for (int dw = 0; dw < K; ++dw) {
                const int h = hs + dh, w = ws + dw;
                const bool ok = live && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
                v[dh][dw] = ok ? x4[(h * W + w) * C4 + c4] : zero4;
            }
        f32x4 acc = zero4;
#pragma unroll
        for (int dh = 0; dh < K; ++dh)
#pragma unroll
            for (int dw = 0; dw < K; ++dw) acc += v[dh][dw];
        acc = acc / pool;
        if (live) y4[i] = acc;
        if (absmax) {
            unsigned m = max(max(__float_as_uint(acc[0]) & 0x7fffffffu, __float_as_uint(acc[1]) & 0x7fffffffu),
                             max(__float_as_uint(acc[2]) & 0x7fffffffu, __float_as_uint(acc[3]) & 0x7fffffffu));
            for (int o = C4 >> 1; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
            if (live && c4 == 0) absmax[(int64_t)blockIdx.x * OW + ow] = m;
        }
    }
}

// One workgroup per input row (n, h): 32-bit index arithmetic only (the flat 64-bit div/mod version was ALU-bound at
// 3.2 TB/s on the stem's 822 MB gradient).
__global__ __launch_bounds__(TPB) void avgpool_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ mul,
                                                          float* __restrict__ gx, unsigned* __restrict__ absmax, int N, int H, int W,
                                                          int C4, int k, int s, int p, int OH, int OW) {
    const int n = blockIdx.x / H, h = blockIdx.x - n * H;
    // windows oh with oh*s - p <= h < oh*s - p + k
    int oh_lo = (h + p - k + s) / s;      // ceil((h+p-k+1)/s) for non-negative numerator
    if (h + p - k + 1 <= 0) oh_lo = 0;
    const int oh_hi = min((h + p) / s, OH - 1);
    const int64_t row = ((int64_t)n * H + h) * W * C4;
    const f32x4* gy4 = reinterpret_cast<const f32x4*>(gy) + (int64_t)n * OH * OW * C4;
    // (every lane runs every trip: the per-pixel maxima below are reduced across the C4 lanes of a pixel by shuffles)
    // The (pixel, channel-chunk) decode and the window bounds are integer divisions by run-time values; with the usual powers of
    // two (C4 = 16, stride 2) they become shifts -- the generic form spent more instructions on them than on the 6 memory
    // operations of an element.
    const int c4_shift = (C4 & (C4 - 1)) == 0 ? __builtin_ctz(C4) : -1;
    const int s_shift = (s & (s - 1)) == 0 ? __builtin_ctz(s) : -1;
    for (int i0 = 0; i0 < W * C4; i0 += TPB) {
        const int i = i0 + threadIdx.x;
        const bool live = i < W * C4;
        const int w = live ? (c4_shift >= 0 ? i >> c4_shift : i / C4) : 0, c4 = live ? i - w * C4 : 0;
        const int lo_num = w + p - k + s;                   // ceil((w + p - k + 1) / s) for a non-negative numerator
        int ow_lo = s_shift >= 0 ? lo_num >> s_shift : lo_num / s;
        if (w + p - k + 1 <= 0) ow_lo = 0;
        const int ow_hi = min(s_shift >= 0 ? (w + p) >> s_shift : (w + p) / s, OW - 1);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            for (int oh = oh_lo; oh <= oh_hi; ++oh) {
                const int hs = oh * s - p;
                const int he = min(hs + k, H + p);
                for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                    const int ws = ow * s - p;
                    const int we = min(ws + k, W + p);
                    const float rpool = 1.0f / (float)((he - hs) * (we - ws));      // one division per window, not four per chunk
                    acc += gy4[(oh * OW + ow) * C4 + c4] * rpool;
                }
            }
            if (mul) acc *= __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mul) + row + i);     // read exactly once
            reinterpret_cast<f32x4*>(gx)[row + i] = acc;
        }
        if (absmax) {      // C4 is a power of two <= 64 (host): the C4 lanes of a pixel are an aligned lane group
            unsigned m = max(max(__float_as_uint(acc[0]) & 0x7fffffffu, __float_as_uint(acc[1]) & 0x7fffffffu),
                             max(__float_as_uint(acc[2]) & 0x7fffffffu, __float_as_uint(acc[3]) & 0x7fffffffu));
            for (int o = C4 >> 1; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
            if (live && c4 == 0) absmax[((int64_t)n * H + h) * W + w] = m;
        }
    }
}

// ... with the windows of an input pixel expanded at compile time: a pixel lies in at most WMAX x WMAX = ceil(k / s)^2 windows (2 x 2 for
// the 3 x 3 / 2 stem pool, 1 for the k = s pools of the CLIP ResNets); the loads of the windows it does not lie in are predicated off and
// add 0 * rpool = +0 in the same place of the same sum (bit-identical), every thread handles TWO items per trip and issues their
// multiplier and gradient loads together: 10 loads in flight where the loop above had 1-5.
template <int WMAX>
__global__ __launch_bounds__(TPB) void avgpool_bwd_win_kernel(const float* __restrict__ gy, const float* __restrict__ mul,
                                                              float* __restrict__ gx, unsigned* __restrict__ absmax, int N, int H, int W,
                                                              int C4, int k, int s, int p, int OH, int OW) {
    constexpr int U = 2;
    const int n = blockIdx.x / H, h = blockIdx.x - n * H;
    int oh_lo = (h + p - k + s) / s;
    if (h + p - k + 1 <= 0) oh_lo = 0;
    const int oh_hi = min((h + p) / s, OH - 1);
    const int64_t row = ((int64_t)n * H + h) * W * C4;
    const f32x4* gy4 = reinterpret_cast<const f32x4*>(gy) + (int64_t)n * OH * OW * C4;
    const f32x4* mul4 = reinterpret_cast<const f32x4*>(mul) + row;
    f32x4* gx4 = reinterpret_cast<f32x4*>(gx) + row;
    const int c4_shift = (C4 & (C4 - 1)) == 0 ? __builtin_ctz(C4) : -1;
    const int s_shift = (s & (s - 1)) == 0 ? __builtin_ctz(s) : -1;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    float rh[WMAX];              // (he - hs) of the row windows
#pragma unroll
    for (int a = 0; a < WMAX; ++a) {
        const int hs = (oh_lo + a) * s - p;
        rh[a] = (float)(min(hs + k, H + p) - hs);
    }
    for (int i0 = 0; i0 < W * C4; i0 += U * TPB) {
        f32x4 g[U][WMAX][WMAX], m4[U];
        float rp[U][WMAX][WMAX];
        bool live[U];
        int wq[U], cq[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * TPB + threadIdx.x;
            live[u] = i < W * C4;
            const int w = live[u] ? (c4_shift >= 0 ? i >> c4_shift : i / C4) : 0, c4 = live[u] ? i - w * C4 : 0;
            wq[u] = w; cq[u] = c4;
            const int lo_num = w + p - k + s;
            int ow_lo = s_shift >= 0 ? lo_num >> s_shift : lo_num / s;
            if (w + p - k + 1 <= 0) ow_lo = 0;
            const int ow_hi = min(s_shift >= 0 ? (w + p) >> s_shift : (w + p) / s, OW - 1);
            m4[u] = (mul && live[u]) ? __builtin_nontemporal_load(mul4 + i) : zero4;
#pragma unroll
            for (int a = 0; a < WMAX; ++a)
#pragma unroll
                for (int b = 0; b < WMAX; ++b) {
                    const int oh = oh_lo + a, ow = ow_lo + b;
                    const bool ok = live[u] && oh <= oh_hi && ow <= ow_hi;
                    const int ws = ow * s - p;
                    rp[u][a][b] = ok ? 1.0f / (rh[a] * (float)(min(ws + k, W + p) - ws)) : 0.f;      // (a window the pixel is not in: 0 x 0, never 0 x inf)
                    g[u][a][b] = ok ? gy4[(oh * OW + ow) * C4 + c4] : zero4;
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f32x4 acc = zero4;
#pragma unroll
            for (int a = 0; a < WMAX; ++a)
#pragma unroll
                for (int b = 0; b < WMAX; ++b) acc += g[u][a][b] * rp[u][a][b];
            if (mul) acc *= m4[u];
            const int i = i0 + u * TPB + threadIdx.x;
            if (live[u]) gx4[i] = acc;
            if (absmax) {
                unsigned m = max(max(__float_as_uint(acc[0]) & 0x7fffffffu, __float_as_uint(acc[1]) & 0x7fffffffu),
                                 max(__float_as_uint(acc[2]) & 0x7fffffffu, __float_as_uint(acc[3]) & 0x7fffffffu));
                for (int o = C4 >> 1; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
                if (live[u] && cq[u] == 0) absmax[((int64_t)n * H + h) * W + wq[u]] = m;
            }
        }
    }
}

// ---- head: global average pool + logit layer -----------------------------------------------------
__global__ __launch_bounds__(TPB) void gap_logits_kernel(const float* __restrict__ x, float* __restrict__ y, int N,
                                                         int HW, int C, float inv_t, float bias) {
    const int64_t total = (int64_t)N * C;
    const int64_t stride = (int64_t)gridDim.x * TPB;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
        const int64_t n = i / C;
        const int c = (int)(i - n * C);
        const float* src = x + n * (int64_t)HW * C + c;
        float acc = 0.f;
        for (int hw = 0; hw < HW; ++hw) acc += src[(int64_t)hw * C];
        float v = acc / (float)HW;
        if (inv_t != 1.0f) v *= inv_t;
        y[i] = v + bias;
    }
}

// the same mean for many positions per image (the token path: 197 tokens x 1000 classes per image ran at 1 TB/s with one thread walking
// all positions of a column): a workgroup takes 64 columns of one image, four thread groups walk every fourth position, their partial
// sums meet in LDS in a fixed order
__global__ __launch_bounds__(256) void gap_logits_wide_kernel(const float* __restrict__ x, float* __restrict__ y, int HW, int C, int cblocks,
                                                              float inv_t, float bias) {
    __shared__ float part[4][64];
    const int n = blockIdx.x / cblocks, cb = blockIdx.x - n * cblocks;
    const int c = cb * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    float a0 = 0.f, a1 = 0.f;
    if (c < C) {
        const float* src = x + (int64_t)n * HW * C + c;
        int hw = g;
        for (; hw + 4 < HW; hw += 8) {
            a0 += src[(int64_t)hw * C];
            a1 += src[(int64_t)(hw + 4) * C];
        }
        if (hw < HW) a0 += src[(int64_t)hw * C];
    }
    part[g][threadIdx.x & 63] = a0 + a1;
    __syncthreads();
    if (g == 0 && c < C) {
        float v = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x])) / (float)HW;
        if (inv_t != 1.0f) v *= inv_t;
        y[(int64_t)n * C + c] = v + bias;
    }
}

// four columns per thread (C % 4 == 0, 16-byte aligned tensors): the tensor is zeros but for one column per image -- 16-byte stores
__global__ __launch_bounds__(TPB) void head_onehot4_kernel(const int64_t* __restrict__ cls, const float* __restrict__ scale,
                                                           float* __restrict__ glin, int N, int HW, int C4, float coef) {
    const int64_t total = (int64_t)N * HW * C4;
    const int64_t stride = (int64_t)gridDim.x * TPB;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
        const int c4 = (int)(i % C4);
        const int64_t n = i / ((int64_t)HW * C4);
        const int k = (int)cls[n];
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((k >> 2) == c4) v[k & 3] = scale[i * 4 + (k & 3)] * coef;
        reinterpret_cast<f32x4*>(glin)[i] = v;
    }
}

__global__ __launch_bounds__(TPB) void head_onehot_kernel(const int64_t* __restrict__ cls,
                                                          const float* __restrict__ scale, float* __restrict__ glin,
                                                          int N, int HW, int C, float coef) {
    const int64_t total = (int64_t)N * HW * C;
    const int64_t stride = (int64_t)gridDim.x * TPB;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % C);
        const int64_t n = i / ((int64_t)HW * C);
        glin[i] = (c == (int)cls[n]) ? scale[i] * coef : 0.f;
    }
}

// The gradient of ONE class's mean logit w.r.t. the head layer's input is rank one per image: d logit[cls] / d lin[n, r, k] is zero but for
// k = cls_n, so  v[n, r, :] = coef * scale[n, r, cls_n] * row_scale[n r] * W[cls_n, :]  -- what the one-hot tensor [N, R, K] followed by
// a K-long contraction computes (head_onehot_kernel + the input-gradient launch), without writing that tensor, reading it back for its
// row maxima and contracting over K - 1 zero columns.  out = v * mul (mul NULL: v), out2 = v (optional), per-row max |out|.
// One wavefront per row, 16-byte columns.
__global__ __launch_bounds__(256) void head_rank1_kernel(const int64_t* __restrict__ cls, const float* __restrict__ scale,
                                                         const float* __restrict__ w, const float* __restrict__ row_scale,
                                                         const float* __restrict__ mul, const float* __restrict__ mul2,
                                                         const float* __restrict__ gate2, float* __restrict__ out, float* __restrict__ out2,
                                                         unsigned* __restrict__ out_absmax, unsigned* __restrict__ out2_absmax, int64_t rows,
                                                         int R, int K, int D4, float coef, int gate_from_mul) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row = wave; row < rows; row += nwaves) {
        const int64_t n = row / R;
        // a class index outside [0, K) explains nothing: a zero gradient, as the one-hot tensor this launch replaces gave (the host
        // validates user-given targets -- bcos_hip/ops.py: check_targets --, this keeps a raw C-ABI caller off out-of-range reads)
        const int64_t kc = cls[n];
        const bool k_ok = kc >= 0 && kc < (int64_t)K;
        const int k = k_ok ? (int)kc : 0;
        float a = k_ok ? coef * scale[row * K + k] : 0.f;
        if (row_scale) a *= row_scale[row];
        const f32x4* wr = reinterpret_cast<const f32x4*>(w + (int64_t)k * D4 * 4);
        unsigned mx = 0u, mx2 = 0u;
        for (int d = lane; d < D4; d += 64) {
            const int64_t i = row * D4 + d;
            const f32x4 v = wr[d] * a;
            f32x4 o = v, m = {0.f, 0.f, 0.f, 0.f};
            if (mul) { m = reinterpret_cast<const f32x4*>(mul)[i]; o *= m; }
            reinterpret_cast<f32x4*>(out)[i] = o;
#pragma unroll
            for (int q = 0; q < 4; ++q) mx = max(mx, __float_as_uint(o[q]) & 0x7fffffffu);
            if (out2) {                      // the second output of the gradient epilogue (bcos_epilogue.out2): v [* mul2] [gated]
                f32x4 o2 = v;
                if (mul2) o2 *= reinterpret_cast<const f32x4*>(mul2)[i];
                if (gate_from_mul) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) o2[q] = (__float_as_uint(m[q]) & 1u) ? o2[q] : 0.f;
                } else if (gate2) {
                    const f32x4 gt = reinterpret_cast<const f32x4*>(gate2)[i];
#pragma unroll
                    for (int q = 0; q < 4; ++q) o2[q] = gt[q] > 0.f ? o2[q] : 0.f;
                }
                reinterpret_cast<f32x4*>(out2)[i] = o2;
#pragma unroll
                for (int q = 0; q < 4; ++q) mx2 = max(mx2, __float_as_uint(o2[q]) & 0x7fffffffu);
            }
        }
        if (out_absmax || out2_absmax) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
                mx2 = max(mx2, (unsigned)__shfl_xor((int)mx2, o));
            }
            if (lane == 0 && out_absmax) out_absmax[row] = mx;
            if (lane == 0 && out2_absmax) out2_absmax[row] = mx2;
        }
    }
}

// one wavefront per row; ties -> lowest index
__global__ __launch_bounds__(TPB) void argmax_rows_kernel(const float* __restrict__ x, int64_t* __restrict__ idx,
                                                          float* __restrict__ val, int N, int C) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * TPB + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * TPB) >> 6;
    for (int r = wave; r < N; r += nwaves) {
        const float* src = x + (int64_t)r * C;
        float best = -INFINITY;
        int arg = 0x7fffffff;
        for (int c = lane; c < C; c += 64) {
            const float v = src[c];
            if (v > best || (v == best && c < arg)) { best = v; arg = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o);
            const int oa = __shfl_xor(arg, o);
            if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
        }
        if (lane == 0) {
            if (idx) idx[r] = arg;
            if (val) val[r] = best;
        }
    }
}

__global__ __launch_bounds__(TPB) void channel_affine_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift, float* __restrict__ y,
                                                             int64_t total4, int C4, int relu) {
    const int64_t stride = (int64_t)gridDim.x * TPB;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total4; i += stride) {
        const int c4 = (int)(i % C4);
        f32x4 v = reinterpret_cast<const f32x4*>(x)[i] * reinterpret_cast<const f32x4*>(scale)[c4];
        if (shift) v += reinterpret_cast<const f32x4*>(shift)[c4];
        if (relu) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
        }
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
}

// y = [relu](x * scale[c] + shift[c] + addend): the normalisation, affine map, residual add and ReLU of a training-mode unit in one pass
__global__ __launch_bounds__(TPB) void channel_affine_add_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, const float* __restrict__ addend,
                                                                 float* __restrict__ y, int64_t total4, int C4, int relu) {
    const int64_t stride = (int64_t)gridDim.x * TPB;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total4; i += stride) {
        const int c4 = (int)(i % C4);
        f32x4 v = reinterpret_cast<const f32x4*>(x)[i] * reinterpret_cast<const f32x4*>(scale)[c4];
        if (shift) v += reinterpret_cast<const f32x4*>(shift)[c4];
        v += reinterpret_cast<const f32x4*>(addend)[i];
        if (relu) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
        }
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
}

// out = act > 0 ? g : 0   (the ReLU gate of a gradient, from the kept activation)
__global__ __launch_bounds__(TPB) void relu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ act, float* __restrict__ out,
                                                       int64_t total4) {
    const int64_t stride = (int64_t)gridDim.x * TPB;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total4; i += stride) {
        const f32x4 a = reinterpret_cast<const f32x4*>(act)[i];
        f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = a[q] > 0.f ? v[q] : 0.f;
        reinterpret_cast<f32x4*>(out)[i] = v;
    }
}

}  // namespace

#define STREAM(s) reinterpret_cast<hipStream_t>(s)

extern "C" int bcos_weight_rownorm_scale(const float* w, const float* gain, float* w_out, int rows, int64_t cols,
                                         void* stream) {
    if (!w || !w_out || rows <= 0 || cols <= 0) return bcos_set_error(BCOS_E_INVAL, "bcos_weight_rownorm_scale: bad argument");
    const unsigned grid = grid_for((int64_t)rows * 64);
    hipLaunchKernelGGL(weight_rownorm_kernel, dim3(grid), dim3(TPB), 0, STREAM(stream), w, gain, w_out, rows, cols);
    return check_launch("weight_rownorm_kernel");
}

namespace {
// y[r, :] = x[r, :] / ||x[r, :]||_2, inv[r] = 1 / ||x[r, :]||_2: one wavefront per row
__global__ __launch_bounds__(TPB) void rows_normalize_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ inv,
                                                             int64_t rows, int C) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * TPB) >> 6;
    for (int64_t r = wave; r < rows; r += nwaves) {
        const float* src = x + r * C;
        float ss = 0.f;
        for (int c = lane; c < C; c += 64) ss = fmaf(src[c], src[c], ss);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        const float nrm = sqrtf(ss);
        if (y) for (int c = lane; c < C; c += 64) y[r * C + c] = src[c] / nrm;
        if (inv && lane == 0) inv[r] = 1.0f / nrm;
    }
}

// gradient of the cosine logit l = u . w (u = f / ||f||, given) w.r.t. f, times a per-row coefficient:
// out[r, :] = coef[r] * inv[r] * (w[r, :] - l[r] * u[r, :])
__global__ __launch_bounds__(TPB) void cosine_grad_kernel(const float* __restrict__ u, const float* __restrict__ w, const float* __restrict__ l,
                                                          const float* __restrict__ inv, const float* __restrict__ coef,
                                                          float* __restrict__ out, int64_t rows, int C) {
    const int64_t n = rows * C, stride = (int64_t)gridDim.x * TPB;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / C;
        const float k = (coef ? coef[r] : 1.f) * inv[r];
        out[i] = k * (w[i] - l[r] * u[i]);
    }
}
}  // namespace

namespace {
__global__ __launch_bounds__(TPB) void weight_row_invnorm_kernel(const float* __restrict__ w, const float* __restrict__ gain,
                                                                 float* __restrict__ inv, int rows, int64_t cols) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * TPB + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * TPB) >> 6;
    for (int r = wave; r < rows; r += nwaves) {
        const float* src = w + (int64_t)r * cols;
        float ss = 0.f;
        for (int64_t c = lane; c < cols; c += 64) ss = fmaf(src[c], src[c], ss);       // (the summation order of weight_rownorm_kernel)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        if (lane == 0) inv[r] = (gain ? gain[r] : 1.f) / sqrtf(ss);
    }
}
}  // namespace

extern "C" int bcos_weight_row_invnorm(const float* w, const float* gain, float* inv, int rows, int64_t cols, void* stream) {
    if (!w || !inv || rows <= 0 || cols <= 0) return bcos_set_error(BCOS_E_INVAL, "bcos_weight_row_invnorm: bad argument");
    hipLaunchKernelGGL(weight_row_invnorm_kernel, dim3(grid_for((int64_t)rows * 64)), dim3(TPB), 0, STREAM(stream), w, gain, inv, rows, cols);
    return check_launch("weight_row_invnorm_kernel");
}

extern "C" int bcos_rows_normalize(const float* x, float* y, float* inv_norm, int64_t rows, int C, void* stream) {
    if (!x || (!y && !inv_norm) || rows <= 0 || C <= 0) return bcos_set_error(BCOS_E_INVAL, "bcos_rows_normalize: bad argument");
    hipLaunchKernelGGL(rows_normalize_kernel, dim3(grid_for(rows * 64)), dim3(TPB), 0, STREAM(stream), x, y, inv_norm, rows, C);
    return check_launch("rows_normalize_kernel");
}

extern "C" int bcos_cosine_grad(const float* u, const float* w, const float* l, const float* inv_norm, const float* coef, float* out,
                                int64_t rows, int C, void* stream) {
    if (!u || !w || !l || !inv_norm || !out || rows <= 0 || C <= 0) return bcos_set_error(BCOS_E_INVAL, "bcos_cosine_grad: bad argument");
    hipLaunchKernelGGL(cosine_grad_kernel, dim3(grid_for(rows * C)), dim3(TPB), 0, STREAM(stream), u, w, l, inv_norm, coef, out, rows, C);
    return check_launch("cosine_grad_kernel");
}

namespace {
// the plainest streaming kernel: 8 x 16 bytes in flight per thread, loads first, then the stores; non-temporal both ways
__global__ __launch_bounds__(TPB) void stream_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int64_t n4) {
    constexpr int U = 8;
    const int64_t stride = (int64_t)gridDim.x * TPB * U;
    for (int64_t base = (int64_t)blockIdx.x * TPB * U + threadIdx.x; base < n4; base += stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + (int64_t)u * TPB;
            if (i < n4) v[u] = __builtin_nontemporal_load(src + i);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + (int64_t)u * TPB;
            if (i < n4) __builtin_nontemporal_store(v[u], dst + i);
        }
    }
}
}  // namespace

extern "C" int bcos_stream_copy(const float* src, float* dst, int64_t n, void* stream) {
    if (!src || !dst || n < 0 || (n & 3)) return bcos_set_error(BCOS_E_INVAL, "bcos_stream_copy: NULL buffer, negative n or n % 4 != 0");
    if ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15)
        return bcos_set_error(BCOS_E_INVAL, "bcos_stream_copy: buffers must be 16-byte aligned");
    if (n == 0) return BCOS_OK;
    const int64_t n4 = n / 4;
    const int64_t want = (n4 + (int64_t)TPB * 8 - 1) / ((int64_t)TPB * 8);
    const unsigned grid = (unsigned)(want < 2048 ? want : 2048);
    hipLaunchKernelGGL(stream_copy_kernel, dim3(grid), dim3(TPB), 0, STREAM(stream), reinterpret_cast<const f32x4*>(src),
                       reinterpret_cast<f32x4*>(dst), n4);
    return check_launch("stream_copy_kernel");
}

extern "C" int bcos_mul(const float* a, const float* b, float* out, int64_t n, void* stream) {
    if (!a || !b || !out || n < 0) return bcos_set_error(BCOS_E_INVAL, "bcos_mul: bad argument");
    if (n == 0) return BCOS_OK;
    int64_t n4 = n / 4;
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out)) & 15) n4 = 0;
    hipLaunchKernelGGL(mul_kernel, dim3(grid_for(n4 > 0 ? n4 : n)), dim3(TPB), 0, STREAM(stream), a, b, out, n4, n);
    return check_launch("mul_kernel");
}

namespace {
__global__ __launch_bounds__(TPB) void maxout_expand_kernel(const float* __restrict__ gy, const float* __restrict__ t,
                                                            float* __restrict__ glin, int64_t n4, int c4, int Cn, int M) {
    const int64_t stride = (int64_t)gridDim.x * TPB;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += stride) {
        const int64_t r = i / c4;
        const int c = (int)(i - r * c4) * 4;
        f32x4 tv = reinterpret_cast<const f32x4*>(t)[i];
        const float* g = gy + r * Cn;
#pragma unroll
        for (int q = 0; q < 4; ++q) tv[q] *= g[(c + q) / M];
        reinterpret_cast<f32x4*>(glin)[i] = tv;
    }
}
}  // namespace

extern "C" int bcos_maxout_expand(const float* gy, const float* t, float* glin, int64_t rows, int Cout, int max_out, void* stream) {
    if (!gy || !t || !glin || rows <= 0 || Cout <= 0 || max_out <= 0 || Cout % 4 != 0 || Cout % max_out != 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_maxout_expand: bad argument");
    const int64_t n4 = rows * (Cout / 4);
    hipLaunchKernelGGL(maxout_expand_kernel, dim3(grid_for(n4)), dim3(TPB), 0, STREAM(stream), gy, t, glin, n4, Cout / 4,
                       Cout / max_out, max_out);
    return check_launch("maxout_expand_kernel");
}

extern "C" int bcos_maxout_scale(const float* lin, const float* norm, float* y, float* scale_out,
                                 int32_t* argmax_out, int64_t rows, int Cout, int max_out, int norm_stride, float b,
                                 void* stream) {
    if (!lin || !y || rows <= 0 || Cout <= 0 || max_out <= 0) return bcos_set_error(BCOS_E_INVAL, "bcos_maxout_scale: bad argument");
    if (norm_stride <= 0 || Cout % norm_stride != 0) return bcos_set_error(BCOS_E_INVAL, "bcos_maxout_scale: Cout % groups != 0");
    hipLaunchKernelGGL(maxout_scale_kernel, dim3(grid_for(rows * Cout)), dim3(TPB), 0, STREAM(stream), lin, norm, y,
                       scale_out, argmax_out, rows, Cout, max_out, norm_stride, b);
    return check_launch("maxout_scale_kernel");
}

extern "C" int bcos_prep_input(const float* x, float* out, const float* mean6, const float* std6, uint32_t* absmax_out, int N,
                               int Cx, int H, int W, int Cpad, int add_inverse, void* stream) {
    if (!x || !out || !mean6 || !std6 || N <= 0 || H <= 0 || W <= 0) return bcos_set_error(BCOS_E_INVAL, "bcos_prep_input: bad argument");
    if (Cx != (add_inverse ? 3 : 6)) return bcos_set_error(BCOS_E_INVAL, "bcos_prep_input: Cx must be 3 (add_inverse) or 6");
    if (Cpad < 6) return bcos_set_error(BCOS_E_INVAL, "bcos_prep_input: Cpad < 6");
    hipLaunchKernelGGL(prep_input_kernel, dim3(grid_pixels((int64_t)N * H * W)), dim3(TPB), 0, STREAM(stream), x, out,
                       mean6, std6, absmax_out, N, Cx, H * W, Cpad, add_inverse);
    return check_launch("prep_input_kernel");
}

extern "C" int bcos_finalize_explanation(const float* gxn, const float* x, const float* std6, float* weights_out,
                                         float* contrib_out, int N, int Cx, int H, int W, int Cpad, int add_inverse,
                                         void* stream) {
    if (!gxn || !x || !std6 || (!weights_out && !contrib_out) || N <= 0 || H <= 0 || W <= 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_finalize_explanation: bad argument");
    if (Cx != (add_inverse ? 3 : 6) || Cpad < 6) return bcos_set_error(BCOS_E_INVAL, "bcos_finalize_explanation: bad channels");
    hipLaunchKernelGGL(finalize_expl_kernel, dim3(grid_pixels((int64_t)N * H * W)), dim3(TPB), 0, STREAM(stream), gxn, x,
                       std6, weights_out, contrib_out, N, Cx, H * W, Cpad, add_inverse);
    return check_launch("finalize_expl_kernel");
}

extern "C" int bcos_contrib_map(const float* x, const float* gx, float* out, int N, int C, int H, int W,
                                void* stream) {
    if (!x || !gx || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0) return bcos_set_error(BCOS_E_INVAL, "bcos_contrib_map: bad argument");
    hipLaunchKernelGGL(contrib_map_kernel, dim3(grid_for((int64_t)N * H * W)), dim3(TPB), 0, STREAM(stream), x, gx, out,
                       N, C, H * W);
    return check_launch("contrib_map_kernel");
}

static int pool_args_ok(int N, int H, int W, int C, int k, int s, int p, int OH, int OW) {
    return N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && k > 0 && s > 0 && p >= 0 && 2 * p <= k && OH > 0 && OW > 0;
}

extern "C" int bcos_avgpool2d_fwd(const float* x, float* y, int N, int H, int W, int C, int k, int s, int p, int OH,
                                  int OW, void* stream) {
    if (!x || !y || !pool_args_ok(N, H, W, C, k, s, p, OH, OW)) return bcos_set_error(BCOS_E_INVAL, "bcos_avgpool2d_fwd: bad argument");
    return bcos_avgpool2d_fwd_absmax(x, y, nullptr, N, H, W, C, k, s, p, OH, OW, stream);
}

extern "C" int bcos_avgpool2d_fwd_absmax(const float* x, float* y, uint32_t* absmax_out, int N, int H, int W, int C, int k, int s, int p,
                                         int OH, int OW, void* stream) {
    if (!x || !y || !pool_args_ok(N, H, W, C, k, s, p, OH, OW)) return bcos_set_error(BCOS_E_INVAL, "bcos_avgpool2d_fwd: bad argument");
    if (absmax_out && (C / 4 > 64 || ((C / 4) & (C / 4 - 1)) != 0 || (k != 2 && k != 3)))
        return bcos_set_error(BCOS_E_NOSUP, "bcos_avgpool2d_fwd_absmax: absmax_out needs k in {2, 3} and C / 4 a power of two <= 64");
    // one workgroup per output row, the window expanded at compile time (k = 3: the stem pool of the ResNets; k = 2: the pools of CLIP's
    // ModifiedResNet); the row kernels index an image with 32 bits
    const bool rows_ok = (int64_t)N * OH < ((int64_t)1 << 31) && (int64_t)H * W * (C / 4) < ((int64_t)1 << 31);
    if (rows_ok && (k == 2 || k == 3)) {
        if (k == 3) hipLaunchKernelGGL(avgpool_fwd_row_kernel<3>, dim3((unsigned)(N * OH)), dim3(TPB), 0, STREAM(stream), x, y, absmax_out, H, W, C / 4, s, p, OH, OW);
        else hipLaunchKernelGGL(avgpool_fwd_row_kernel<2>, dim3((unsigned)(N * OH)), dim3(TPB), 0, STREAM(stream), x, y, absmax_out, H, W, C / 4, s, p, OH, OW);
        return check_launch("avgpool_fwd_row_kernel");
    }
    if (absmax_out) return bcos_set_error(BCOS_E_NOSUP, "bcos_avgpool2d_fwd_absmax: tensor too large for the row kernel");
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(grid_for((int64_t)N * OH * OW * (C / 4))), dim3(TPB), 0, STREAM(stream),
                       x, y, N, H, W, C / 4, k, s, p, OH, OW);
    return check_launch("avgpool_fwd_kernel");
}

extern "C" int bcos_avgpool2d_bwd(const float* gy, const float* mul, float* gx, uint32_t* absmax_out, int N, int H, int W, int C,
                                  int k, int s, int p, int OH, int OW, void* stream) {
    if (!gy || !gx || !pool_args_ok(N, H, W, C, k, s, p, OH, OW)) return bcos_set_error(BCOS_E_INVAL, "bcos_avgpool2d_bwd: bad argument");
    if ((int64_t)N * H >= ((int64_t)1 << 31) || (int64_t)OH * OW * (C / 4) >= ((int64_t)1 << 31))
        return bcos_set_error(BCOS_E_NOSUP, "bcos_avgpool2d_bwd: tensor too large");
    if (absmax_out && (C / 4 > 64 || ((C / 4) & (C / 4 - 1)) != 0))
        return bcos_set_error(BCOS_E_NOSUP, "bcos_avgpool2d_bwd: absmax_out needs C / 4 to be a power of two <= 64");
    const int wmax = (k + s - 1) / s;         // windows per input pixel and axis
    if (wmax == 1) hipLaunchKernelGGL(avgpool_bwd_win_kernel<1>, dim3((unsigned)(N * H)), dim3(TPB), 0, STREAM(stream), gy, mul, gx, absmax_out, N, H, W, C / 4, k, s, p, OH, OW);
    else if (wmax == 2) hipLaunchKernelGGL(avgpool_bwd_win_kernel<2>, dim3((unsigned)(N * H)), dim3(TPB), 0, STREAM(stream), gy, mul, gx, absmax_out, N, H, W, C / 4, k, s, p, OH, OW);
    else hipLaunchKernelGGL(avgpool_bwd_kernel, dim3((unsigned)(N * H)), dim3(TPB), 0, STREAM(stream), gy,
                            mul, gx, absmax_out, N, H, W, C / 4, k, s, p, OH, OW);
    return check_launch("avgpool_bwd_kernel");
}

extern "C" int bcos_global_avgpool_logits(const float* x, float* y, int N, int HW, int C, float inv_temperature,
                                          float logit_bias, void* stream) {
    if (!x || !y || N <= 0 || HW <= 0 || C <= 0) return bcos_set_error(BCOS_E_INVAL, "bcos_global_avgpool_logits: bad argument");
    if (HW >= 64) {        // many positions per image (token path): see gap_logits_wide_kernel
        const int cblocks = (C + 63) / 64;
        hipLaunchKernelGGL(gap_logits_wide_kernel, dim3((unsigned)((int64_t)N * cblocks)), dim3(256), 0, STREAM(stream), x, y, HW, C, cblocks,
                           inv_temperature, logit_bias);
        return check_launch("gap_logits_wide_kernel");
    }
    hipLaunchKernelGGL(gap_logits_kernel, dim3(grid_for((int64_t)N * C)), dim3(TPB), 0, STREAM(stream), x, y, N, HW, C,
                       inv_temperature, logit_bias);
    return check_launch("gap_logits_kernel");
}

extern "C" int bcos_head_onehot_grad(const int64_t* cls, const float* scale, float* glin, int N, int HW, int C,
                                     float inv_temperature, void* stream) {
    if (!cls || !scale || !glin || N <= 0 || HW <= 0 || C <= 0) return bcos_set_error(BCOS_E_INVAL, "bcos_head_onehot_grad: bad argument");
    const float coef = inv_temperature / (float)HW;
    if (C % 4 == 0 && !((reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(glin)) & 15)) {
        hipLaunchKernelGGL(head_onehot4_kernel, dim3(grid_for((int64_t)N * HW * (C / 4))), dim3(TPB), 0, STREAM(stream), cls, scale, glin, N, HW,
                           C / 4, coef);
        return check_launch("head_onehot_kernel");
    }
    hipLaunchKernelGGL(head_onehot_kernel, dim3(grid_for((int64_t)N * HW * C)), dim3(TPB), 0, STREAM(stream), cls, scale,
                       glin, N, HW, C, coef);
    return check_launch("head_onehot_kernel");
}

extern "C" int bcos_head_rank1_grad_ex(const int64_t* cls, const float* scale, const float* w, const float* row_scale, const float* mul,
                                       const float* mul2, const float* gate2, int gate2_from_mul, float* out, float* out2,
                                       uint32_t* out_absmax, uint32_t* out2_absmax, int N, int R, int K, int D, float inv_temperature,
                                       void* stream) {
    if (!cls || !scale || !w || !out || N <= 0 || R <= 0 || K <= 0 || D <= 0 || D % 4 != 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_head_rank1_grad: bad argument (D must be a multiple of 4)");
    if ((mul2 || gate2 || gate2_from_mul || out2_absmax) && !out2)
        return bcos_set_error(BCOS_E_INVAL, "bcos_head_rank1_grad: mul2 / gate2 / out2_absmax belong to out2");
    if (gate2_from_mul && (!mul || gate2)) return bcos_set_error(BCOS_E_INVAL, "bcos_head_rank1_grad: gate2_from_mul needs mul and excludes gate2");
    if ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(mul) | reinterpret_cast<uintptr_t>(mul2) | reinterpret_cast<uintptr_t>(gate2) |
         reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(out2)) & 15)
        return bcos_set_error(BCOS_E_INVAL, "bcos_head_rank1_grad: tensors must be 16-byte aligned");
    const int64_t rows = (int64_t)N * R;
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(head_rank1_kernel, dim3((unsigned)blocks), dim3(256), 0, STREAM(stream), cls, scale, w, row_scale, mul, mul2, gate2, out,
                       out2, out_absmax, out2_absmax, rows, R, K, D / 4, inv_temperature / (float)R, gate2_from_mul ? 1 : 0);
    return check_launch("head_rank1_kernel");
}

extern "C" int bcos_head_rank1_grad(const int64_t* cls, const float* scale, const float* w, const float* row_scale, const float* mul,
                                    float* out, float* out2, uint32_t* out_absmax, int N, int R, int K, int D, float inv_temperature,
                                    void* stream) {
    return bcos_head_rank1_grad_ex(cls, scale, w, row_scale, mul, nullptr, nullptr, 0, out, out2, out_absmax, nullptr, N, R, K, D,
                                   inv_temperature, stream);
}

extern "C" int bcos_argmax_rows(const float* x, int64_t* idx, float* val, int N, int C, void* stream) {
    if (!x || (!idx && !val) || N <= 0 || C <= 0) return bcos_set_error(BCOS_E_INVAL, "bcos_argmax_rows: bad argument");
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(grid_for((int64_t)N * 64)), dim3(TPB), 0, STREAM(stream), x, idx, val, N, C);
    return check_launch("argmax_rows_kernel");
}

extern "C" int bcos_channel_affine_add(const float* x, const float* scale, const float* shift, const float* addend, float* y,
                                       int64_t pixels, int C, int relu, void* stream) {
    if (!x || !scale || !addend || !y || pixels <= 0 || C <= 0 || C % 4 != 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_channel_affine_add: bad argument (C % 4)");
    const int64_t total4 = pixels * (C / 4);
    hipLaunchKernelGGL(channel_affine_add_kernel, dim3(grid_for(total4)), dim3(TPB), 0, STREAM(stream), x, scale, shift, addend, y,
                       total4, C / 4, relu);
    return check_launch("channel_affine_add_kernel");
}

extern "C" int bcos_relu_bwd(const float* g, const float* act, float* out, int64_t n, void* stream) {
    if (!g || !act || !out || n <= 0 || n % 4 != 0) return bcos_set_error(BCOS_E_INVAL, "bcos_relu_bwd: bad argument (n % 4)");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n / 4)), dim3(TPB), 0, STREAM(stream), g, act, out, n / 4);
    return check_launch("relu_bwd_kernel");
}

extern "C" int bcos_channel_affine(const float* x, const float* scale, const float* shift, float* y, int64_t pixels,
                                   int C, int relu, void* stream) {
    if (!x || !scale || !y || pixels <= 0 || C <= 0 || C % 4 != 0) return bcos_set_error(BCOS_E_INVAL, "bcos_channel_affine: bad argument (C % 4)");
    const int64_t total4 = pixels * (C / 4);
    hipLaunchKernelGGL(channel_affine_kernel, dim3(grid_for(total4)), dim3(TPB), 0, STREAM(stream), x, scale, shift, y,
                       total4, C / 4, relu);
    return check_launch("channel_affine_kernel");
}
