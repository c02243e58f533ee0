// bcos_render.hip -- batched RGBA rendering of B-cos explanations on gfx950.
//
// Device restatement of gradient_to_image (reference bcos/common.py:387-436; duplicated in
// interpretability/analyses/text_localisation.py:106-119), for a whole batch:
//   contribs = sum_c x_c W_c;  d = clamp(W / (max_c |W_c| + 1e-12), 0);  rgb = d[:3] / (d[:3] + d[3:] + 1e-12)
//   alpha = ||W||_2 over c, 1e-12 where contribs < 0;  alpha = avg_pool2d(alpha, smooth, 1, (smooth-1)/2)  (zero padded,
//   divisor smooth^2);  alpha = clip(alpha / quantile(alpha, q), 0, 1)  with torch.quantile's linear interpolation.
// Three HBM-bound launches: (1) per-pixel colour + raw alpha (reads W and x once), (2) separable box filter through an
// LDS tile, (3) one workgroup per image: exact order statistics by 4-pass radix select on the float bit patterns
// (alpha >= 0, so the unsigned patterns are ordered like the values) followed by the normalisation of that image.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bcos_hip.h"
#include "bcos_internal.h"

namespace {

constexpr int TPB = 256;

__global__ __launch_bounds__(TPB) void render_rgb_alpha_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               float* __restrict__ rgba, float* __restrict__ alpha,
                                                               int N, int Cx, int HW, int add_inverse) {
    const int64_t total = (int64_t)N * HW;
    const int64_t stride = (int64_t)gridDim.x * TPB;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
        const int64_t n = i / HW, hw = i - n * HW;
        const float* wp = w + n * 6 * (int64_t)HW + hw;
        const float* xp = x + n * (int64_t)Cx * HW + hw;
        float wv[6], contrib = 0.f, mx = 0.f, ss = 0.f;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            wv[c] = wp[(int64_t)c * HW];
            float xv;
            if (add_inverse) xv = c < 3 ? xp[(int64_t)c * HW] : 1.0f - xp[(int64_t)(c - 3) * HW];
            else xv = xp[(int64_t)c * HW];
            contrib += xv * wv[c];
            mx = fmaxf(mx, fabsf(wv[c]));
            ss += wv[c] * wv[c];
        }
        const float inv = 1.0f / (mx + 1e-12f);
        float d[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) d[c] = fmaxf(wv[c] * inv, 0.f);
        float* o = rgba + i * 4;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = d[c] / (d[c] + d[c + 3] + 1e-12f);
        alpha[i] = contrib < 0.f ? 1e-12f : sqrtf(ss);
    }
}

// out = avg_pool2d(in, k, stride 1, pad (k-1)/2), count_include_pad: 32x8 output pixels per workgroup
constexpr int BX = 32, BY = 8;
__global__ __launch_bounds__(TPB) void box_filter_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                         int k) {
    extern __shared__ float tile[];          // [(BY + k - 1)][(BX + k - 1)] input, then [(BY + k - 1)][BX] row sums
    const int r = (k - 1) / 2;
    const int TW = BX + k - 1, TH = BY + k - 1;
    float* rows = tile + TH * TW;
    const int n = blockIdx.z;
    const int x0 = blockIdx.x * BX, y0 = blockIdx.y * BY;
    const float* src = in + (int64_t)n * H * W;
    for (int i = threadIdx.x; i < TH * TW; i += TPB) {
        const int ty = i / TW, tx = i - ty * TW;
        const int yy = y0 + ty - r, xx = x0 + tx - r;
        tile[i] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? src[(int64_t)yy * W + xx] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TH * BX; i += TPB) {
        const int ty = i / BX, tx = i - ty * BX;
        float s = 0.f;
        for (int j = 0; j < k; ++j) s += tile[ty * TW + tx + j];
        rows[i] = s;
    }
    __syncthreads();
    const int tx = threadIdx.x % BX, ty = threadIdx.x / BX;
    const int xx = x0 + tx, yy = y0 + ty;
    if (xx < W && yy < H) {
        float s = 0.f;
        for (int j = 0; j < k; ++j) s += rows[(ty + j) * BX + tx];
        out[(int64_t)n * H * W + (int64_t)yy * W + xx] = s / (float)(k * k);
    }
}

// One workgroup per image: v_lo = element of ascending rank `lo`, v_hi = rank lo + 1 (if it exists), quantile =
// v_lo + frac (v_hi - v_lo)  [torch.quantile, interpolation='linear'], then alpha -> clip(alpha / quantile, 0, 1).
__global__ __launch_bounds__(TPB) void quantile_normalise_kernel(const float* __restrict__ alpha, float* __restrict__ rgba,
                                                                 float* __restrict__ qout, int HW, float q) {
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_rank, s_cnt_le, s_min_gt;
    const int n = blockIdx.x;
    const float* a = alpha + (int64_t)n * HW;
    const float pos = q * (float)(HW - 1);            // rank arithmetic in the tensor dtype, like torch
    const int lo = (int)floorf(pos);
    const float frac = pos - (float)lo;
    unsigned prefix = 0, mask = 0;
    unsigned rank = (unsigned)lo;                     // rank among the elements matching the prefix so far
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < 256; i += TPB) hist[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < HW; i += TPB) {
            const unsigned u = __float_as_uint(a[i]);
            if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned acc = 0;
            int b = 0;
            for (; b < 256; ++b) {
                if (acc + hist[b] > rank) break;
                acc += hist[b];
            }
            s_prefix = prefix | ((unsigned)b << shift);
            s_rank = rank - acc;
        }
        __syncthreads();
        prefix = s_prefix;
        rank = s_rank;
        mask |= 255u << shift;
        __syncthreads();
    }
    const unsigned ulo = prefix;
    if (threadIdx.x == 0) { s_cnt_le = 0; s_min_gt = 0xffffffffu; }
    __syncthreads();
    unsigned cnt = 0, mn = 0xffffffffu;
    for (int i = threadIdx.x; i < HW; i += TPB) {
        const unsigned u = __float_as_uint(a[i]);
        if (u <= ulo) ++cnt;
        else mn = min(mn, u);
    }
    atomicAdd(&s_cnt_le, cnt);
    atomicMin(&s_min_gt, mn);
    __syncthreads();
    const float vlo = __uint_as_float(ulo);
    float vhi = vlo;
    if (lo + 1 < HW && s_cnt_le <= (unsigned)(lo + 1)) vhi = __uint_as_float(s_min_gt);
    const float qv = vlo + frac * (vhi - vlo);
    if (threadIdx.x == 0 && qout) qout[n] = qv;
    float* o = rgba + (int64_t)n * HW * 4;
    for (int i = threadIdx.x; i < HW; i += TPB) o[(int64_t)i * 4 + 3] = fminf(fmaxf(a[i] / qv, 0.f), 1.f);
}

// Grid pointing game (interpretability/analyses/localisation.py:319-321,387-401): one workgroup per attribution map.
// a = clamp(+-attr, 0); mean of a over every cell_h x cell_w cell (avg_pool2d, floor mode); frac = mean / sum of means
// where (sum * mean) > 0, else 0; written in the reference's order (permute(0,1,3,2).reshape): index = col * rows + row.
__global__ __launch_bounds__(TPB) void localisation_kernel(const float* __restrict__ attr, float* __restrict__ frac, int H,
                                                           int W, int cell_h, int cell_w, int neg) {
    __shared__ float s_part[TPB / 64];
    __shared__ float s_mean[64];
    const int t = blockIdx.x;
    const int rows = H / cell_h, cols = W / cell_w;
    const float* a = attr + (int64_t)t * H * W;
    const int cell_px = cell_h * cell_w;
    for (int c = 0; c < rows * cols; ++c) {
        const int r = c / cols, q = c - r * cols;
        float acc = 0.f;
        for (int i = threadIdx.x; i < cell_px; i += TPB) {        // fixed order: deterministic sums
            const int y = i / cell_w, x = i - y * cell_w;
            float v = a[(int64_t)(r * cell_h + y) * W + q * cell_w + x];
            v = neg ? -v : v;
            acc += fmaxf(v, 0.f);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            float tot = 0.f;
            for (int wv = 0; wv < TPB / 64; ++wv) tot += s_part[wv];
            s_mean[c] = tot / (float)cell_px;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float total = 0.f;
        for (int c = 0; c < rows * cols; ++c) total += s_mean[c];
        for (int c = 0; c < rows * cols; ++c) {
            const int r = c / cols, q = c - r * cols;
            const float m = s_mean[c];
            frac[(int64_t)t * rows * cols + q * rows + r] = (total * m > 0.f) ? m / total : 0.f;
        }
    }
}

}  // namespace

extern "C" int bcos_box_filter(const float* in, float* out, int N, int H, int W, int k, void* stream) {
    if (!in || !out || in == out || N <= 0 || H <= 0 || W <= 0 || k < 1 || k % 2 == 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_box_filter: bad argument (odd window, distinct buffers)");
    if (N > 65535) return bcos_set_error(BCOS_E_NOSUP, "bcos_box_filter: more than 65535 maps per call");
    const size_t lds = ((size_t)(BY + k - 1) * (BX + k - 1) + (size_t)(BY + k - 1) * BX) * sizeof(float);
    if (lds > 64 * 1024) return bcos_set_error(BCOS_E_NOSUP, "bcos_box_filter: window too large");
    hipLaunchKernelGGL(box_filter_kernel, dim3((W + BX - 1) / BX, (H + BY - 1) / BY, N), dim3(TPB), lds,
                       reinterpret_cast<hipStream_t>(stream), in, out, H, W, k);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return bcos_set_hip_error("box_filter launch", err);
    return BCOS_OK;
}

extern "C" int bcos_localisation_fractions(const float* attr, float* frac, int T, int H, int W, int cell_h, int cell_w,
                                           int neg, void* stream) {
    if (!attr || !frac || T <= 0 || H <= 0 || W <= 0 || cell_h <= 0 || cell_w <= 0 || cell_h > H || cell_w > W)
        return bcos_set_error(BCOS_E_INVAL, "bcos_localisation_fractions: bad argument");
    if ((H / cell_h) * (W / cell_w) > 64) return bcos_set_error(BCOS_E_NOSUP, "bcos_localisation_fractions: more than 64 cells");
    hipLaunchKernelGGL(localisation_kernel, dim3(T), dim3(TPB), 0, reinterpret_cast<hipStream_t>(stream), attr, frac, H, W,
                       cell_h, cell_w, neg);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return bcos_set_hip_error("localisation launch", err);
    return BCOS_OK;
}

extern "C" int bcos_render_explanations(const float* x, const float* weights, float* rgba, float* scratch, float* quantiles,
                                        int N, int Cx, int H, int W, int smooth, float q, int add_inverse, void* stream) {
    if (!x || !weights || !rgba || !scratch || N <= 0 || H <= 0 || W <= 0 || smooth < 0 || !(q >= 0.f && q <= 1.f))
        return bcos_set_error(BCOS_E_INVAL, "bcos_render_explanations: bad argument");
    if (Cx != (add_inverse ? 3 : 6)) return bcos_set_error(BCOS_E_INVAL, "bcos_render_explanations: bad channels");
    if (smooth > 0 && smooth % 2 == 0) return bcos_set_error(BCOS_E_NOSUP, "bcos_render_explanations: even smoothing window");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t HW = (int64_t)H * W;
    if (HW >= ((int64_t)1 << 24)) return bcos_set_error(BCOS_E_NOSUP, "bcos_render_explanations: image too large");
    float* raw = scratch;
    float* smoothed = smooth > 1 ? scratch + (int64_t)N * HW : scratch;
    const int64_t total = (int64_t)N * HW;
    const unsigned blocks = (unsigned)((total + TPB - 1) / TPB < 65535 * 16 ? (total + TPB - 1) / TPB : 65535 * 16);
    hipLaunchKernelGGL(render_rgb_alpha_kernel, dim3(blocks), dim3(TPB), 0, s, x, weights, rgba, raw, N, Cx, (int)HW, add_inverse);
    if (smooth > 1) {
        const size_t lds = ((size_t)(BY + smooth - 1) * (BX + smooth - 1) + (size_t)(BY + smooth - 1) * BX) * sizeof(float);
        if (lds > 64 * 1024) return bcos_set_error(BCOS_E_NOSUP, "bcos_render_explanations: smoothing window too large");
        hipLaunchKernelGGL(box_filter_kernel, dim3((W + BX - 1) / BX, (H + BY - 1) / BY, N), dim3(TPB), lds, s, raw, smoothed, H, W, smooth);
    }
    hipLaunchKernelGGL(quantile_normalise_kernel, dim3(N), dim3(TPB), 0, s, smoothed, rgba, quantiles, (int)HW, q);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return bcos_set_hip_error("render launch", err);
    return BCOS_OK;
}
