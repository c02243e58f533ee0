// bcos_vit.hip -- the non-B-cos pieces of B-cosified transformers on gfx950:
// DetachableLayerNorm (bcos/modules/norms/centered_norms.py:187-245), MyGELU (bcosify_vit.py:27-32), the softmax
// attention of bcos/models/vit.py:143-158 / bcos/modules/bcosattnpool.py:22-59 with q,k detached in explanation
// mode, the token positional-embedding add and the un-patchify end of the ViT explanation pass.
// All of these are small next to the B-cos linears (ViT-Ti: 30 MFLOP of attention vs 1.75 GFLOP of B-cos GEMMs per
// image), so they are written as streaming / VALU kernels: one wavefront per LayerNorm row, one workgroup per
// (image, head) for attention with K, V (forward) or Q, dOut (backward) resident in LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "bcos_hip.h"
#include "bcos_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TPB = 256;

inline int check_launch(const char* what) {
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return bcos_set_hip_error(what, err);
    return BCOS_OK;
}

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- LayerNorm over the last dimension, one wavefront per row -----------------------------------------------
__global__ __launch_bounds__(TPB) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ y,
                                                            float* __restrict__ rstd_out, int64_t rows, int D,
                                                            float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * TPB) >> 6;
    for (int64_t r = wave; r < rows; r += nwaves) {
        const float* src = x + r * D;
        float s = 0.f;
        for (int c = lane; c < D; c += 64) s += src[c];
        const float mean = wave_sum(s) / (float)D;
        float v = 0.f;
        for (int c = lane; c < D; c += 64) { const float d = src[c] - mean; v = fmaf(d, d, v); }
        const float var = wave_sum(v) / (float)D;
        const float sd = sqrtf(var + eps);
        float* dst = y + r * D;
        for (int c = lane; c < D; c += 64) {
            float o = (src[c] - mean) / sd;
            if (w) o *= w[c];
            if (b) o += b[c];
            dst[c] = o;
        }
        if (rstd_out && lane == 0) rstd_out[r] = 1.0f / sd;
    }
}

// explanation mode: the variance is a constant, the mean is not (centered_norms.py:204-215):
//   y = w * (x - mean(x)) / std   =>   gx = h - mean(h),  h = gy * w / std
// out = gx (+ addend); out2 = out * mul2 (the scale of the B-cos layer that produced x), both optional extras.
__global__ __launch_bounds__(TPB) void layernorm_bwd_detached_kernel(const float* __restrict__ gy,
                                                                     const float* __restrict__ w,
                                                                     const float* __restrict__ rstd,
                                                                     const float* __restrict__ addend,
                                                                     const float* __restrict__ mul2,
                                                                     float* __restrict__ out, float* __restrict__ out2,
                                                                     int64_t rows, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * TPB) >> 6;
    for (int64_t r = wave; r < rows; r += nwaves) {
        const float* g = gy + r * D;
        const float rs = rstd[r];
        float s = 0.f;
        for (int c = lane; c < D; c += 64) s += g[c] * (w ? w[c] : 1.f) * rs;
        const float mh = wave_sum(s) / (float)D;
        for (int c = lane; c < D; c += 64) {
            float o = g[c] * (w ? w[c] : 1.f) * rs - mh;
            if (addend) o += addend[r * D + c];
            if (out) out[r * D + c] = o;
            if (out2) out2[r * D + c] = mul2 ? o * mul2[r * D + c] : o;
        }
    }
}

// ---- GELU with detachable gate: y = gate(x) * x, gate = 0.5 (1 + erf(x / sqrt 2)) ------------------------------
__global__ __launch_bounds__(TPB) void gelu_gate_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        float* __restrict__ gate_out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * TPB;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) {
        const float v = x[i];
        const float gate = 0.5f * (1.0f + erff(v / 1.4142135623730951f));
        y[i] = gate * v;
        if (gate_out) gate_out[i] = gate;
    }
}

// ---- x[b, t, :] += pe[t, :] -----------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void add_rows_bcast_kernel(float* __restrict__ x, const float* __restrict__ pe,
                                                             int64_t total4, int64_t period4) {
    const int64_t stride = (int64_t)gridDim.x * TPB;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total4; i += stride) {
        f32x4 v = reinterpret_cast<f32x4*>(x)[i];
        v += reinterpret_cast<const f32x4*>(pe)[i % period4];
        reinterpret_cast<f32x4*>(x)[i] = v;
    }
}

// ---- softmax attention, one workgroup per (batch, head), head dim 64 ----------------------------------------
// qkv: [B, T, 3*H*64] laid out "(three h d)" like vit.py:145-146; out: [B, T, H*64]
// stats: [B, H, T, 2] = (row max, 1 / row sum) kept for the explanation backward instead of the T x T matrix.
constexpr int DH = 64;

__global__ __launch_bounds__(TPB) void attention_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                            float* __restrict__ stats, int B, int T, int H,
                                                            float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sK = smem;                 // [T][64]
    float* sV = smem + (size_t)T * DH;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int inner = H * DH;
    const float* base = qkv + (int64_t)b * T * 3 * inner;
    for (int i = threadIdx.x; i < T * (DH / 4); i += TPB) {
        const int t = i / (DH / 4), c4 = i % (DH / 4);
        const float* row = base + (int64_t)t * 3 * inner + h * DH + c4 * 4;
        *reinterpret_cast<f32x4*>(sK + t * DH + c4 * 4) = *reinterpret_cast<const f32x4*>(row + inner);
        *reinterpret_cast<f32x4*>(sV + t * DH + c4 * 4) = *reinterpret_cast<const f32x4*>(row + 2 * inner);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < T; i += TPB) {      // query row i
        float q[DH];
        const float* qrow = base + (int64_t)i * 3 * inner + h * DH;
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(qrow + d);
            q[d] = v[0]; q[d + 1] = v[1]; q[d + 2] = v[2]; q[d + 3] = v[3];
        }
        float mx = -INFINITY;
        for (int j = 0; j < T; ++j) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < DH; ++d) s = fmaf(q[d], sK[j * DH + d], s);
            mx = fmaxf(mx, s * scale);
        }
        float l = 0.f;
        for (int j = 0; j < T; ++j) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < DH; ++d) s = fmaf(q[d], sK[j * DH + d], s);
            l += expf(s * scale - mx);
        }
        const float rl = 1.0f / l;
        float acc[DH];
#pragma unroll
        for (int d = 0; d < DH; ++d) acc[d] = 0.f;
        for (int j = 0; j < T; ++j) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < DH; ++d) s = fmaf(q[d], sK[j * DH + d], s);
            const float pj = expf(s * scale - mx) * rl;
#pragma unroll
            for (int d = 0; d < DH; ++d) acc[d] = fmaf(pj, sV[j * DH + d], acc[d]);
        }
        float* orow = out + ((int64_t)b * T + i) * inner + h * DH;
#pragma unroll
        for (int d = 0; d < DH; d += 4) *reinterpret_cast<f32x4*>(orow + d) = f32x4{acc[d], acc[d + 1], acc[d + 2], acc[d + 3]};
        if (stats) {
            float* st = stats + (((int64_t)b * H + h) * T + i) * 2;
            st[0] = mx;
            st[1] = rl;
        }
    }
}

// explanation mode: q and k are detached (vit.py:148-151), so attn is a constant and only v receives gradient:
//   gv[j, :] = sum_i attn[i, j] * gout[i, :],  attn[i, j] = exp(q_i . k_j * scale - max_i) / sum_i  (recomputed)
__global__ __launch_bounds__(TPB) void attention_bwd_v_kernel(const float* __restrict__ qkv,
                                                              const float* __restrict__ stats,
                                                              const float* __restrict__ gout, float* __restrict__ gv,
                                                              int B, int T, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sQ = smem;                         // [T][64]
    float* sG = smem + (size_t)T * DH;        // [T][64]
    float* sS = sG + (size_t)T * DH;          // [T][2]
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int inner = H * DH;
    const float* base = qkv + (int64_t)b * T * 3 * inner;
    for (int i = threadIdx.x; i < T * (DH / 4); i += TPB) {
        const int t = i / (DH / 4), c4 = i % (DH / 4);
        *reinterpret_cast<f32x4*>(sQ + t * DH + c4 * 4) =
            *reinterpret_cast<const f32x4*>(base + (int64_t)t * 3 * inner + h * DH + c4 * 4);
        *reinterpret_cast<f32x4*>(sG + t * DH + c4 * 4) =
            *reinterpret_cast<const f32x4*>(gout + ((int64_t)b * T + t) * inner + h * DH + c4 * 4);
    }
    for (int i = threadIdx.x; i < 2 * T; i += TPB) sS[i] = stats[((int64_t)b * H + h) * T * 2 + i];
    __syncthreads();
    for (int j = threadIdx.x; j < T; j += TPB) {      // key / value row j
        float k[DH];
        const float* krow = base + (int64_t)j * 3 * inner + inner + h * DH;
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(krow + d);
            k[d] = v[0]; k[d + 1] = v[1]; k[d + 2] = v[2]; k[d + 3] = v[3];
        }
        float acc[DH];
#pragma unroll
        for (int d = 0; d < DH; ++d) acc[d] = 0.f;
        for (int i = 0; i < T; ++i) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < DH; ++d) s = fmaf(sQ[i * DH + d], k[d], s);
            const float pij = expf(s * scale - sS[2 * i]) * sS[2 * i + 1];
#pragma unroll
            for (int d = 0; d < DH; ++d) acc[d] = fmaf(pij, sG[i * DH + d], acc[d]);
        }
        float* grow = gv + ((int64_t)b * T + j) * inner + h * DH;
#pragma unroll
        for (int d = 0; d < DH; d += 4) *reinterpret_cast<f32x4*>(grow + d) = f32x4{acc[d], acc[d + 1], acc[d + 2], acc[d + 3]};
    }
}

// ---- end of the ViT explanation pass: patch-major gradient -> W(x) NCHW + contribution map -----------------
// gp: [N, gh, gw, ps, ps, Cpad] (the input gradient of the patch embedding, one row per patch, "(p1 p2 c)" order
// of vit.py:291 with c padded); x: network input [N, Cx, H, W]
__global__ __launch_bounds__(TPB) void finalize_patches_kernel(const float* __restrict__ gp, const float* __restrict__ x,
                                                               const float* __restrict__ std6, float* __restrict__ wout,
                                                               float* __restrict__ cout, int N, int Cx, int H, int W,
                                                               int ps, int Cpad, int add_inverse) {
    const int64_t HW = (int64_t)H * W;
    const int64_t total = (int64_t)N * HW;
    const int64_t stride = (int64_t)gridDim.x * TPB;
    const int gw = W / ps;
    float sd[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) sd[c] = std6[c];
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
        const int64_t n = i / HW;
        const int64_t hw = i - n * HW;
        const int hh = (int)(hw / W), ww = (int)(hw - (int64_t)hh * W);
        const int pi = hh / ps, r = hh - pi * ps, pj = ww / ps, s = ww - pj * ps;
        const float* g = gp + ((((n * (H / ps) + pi) * gw + pj) * ps + r) * ps + s) * Cpad;
        const float* src = x + n * (int64_t)Cx * HW + hw;
        float contrib = 0.f;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const float wv = g[c] / sd[c];
            float xv;
            if (add_inverse) xv = c < 3 ? src[(int64_t)c * HW] : 1.0f - src[(int64_t)(c - 3) * HW];
            else xv = src[(int64_t)c * HW];
            if (wout) wout[(n * 6 + c) * HW + hw] = wv;
            contrib += xv * wv;
        }
        if (cout) cout[i] = contrib;
    }
}

inline unsigned grid_rows(int64_t rows) {
    int64_t b = (rows * 64 + TPB - 1) / TPB;
    if (b < 1) b = 1;
    if (b > 256 * 8) b = 256 * 8;
    return (unsigned)b;
}
inline unsigned grid_elems(int64_t n) {
    int64_t b = (n + TPB - 1) / TPB;
    if (b < 1) b = 1;
    if (b > 256 * 8) b = 256 * 8;
    return (unsigned)b;
}

}  // namespace

#define STREAM(s) reinterpret_cast<hipStream_t>(s)

extern "C" int bcos_layernorm_fwd(const float* x, const float* weight, const float* bias, float* y, float* rstd_out,
                                  int64_t rows, int D, float eps, void* stream) {
    if (!x || !y || rows <= 0 || D <= 0) return bcos_set_error(BCOS_E_INVAL, "bcos_layernorm_fwd: bad argument");
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(grid_rows(rows)), dim3(TPB), 0, STREAM(stream), x, weight, bias, y,
                       rstd_out, rows, D, eps);
    return check_launch("layernorm_fwd_kernel");
}

extern "C" int bcos_layernorm_bwd_detached(const float* gy, const float* weight, const float* rstd, const float* addend,
                                           const float* mul2, float* out, float* out2, int64_t rows, int D,
                                           void* stream) {
    if (!gy || !rstd || (!out && !out2) || rows <= 0 || D <= 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_layernorm_bwd_detached: bad argument");
    hipLaunchKernelGGL(layernorm_bwd_detached_kernel, dim3(grid_rows(rows)), dim3(TPB), 0, STREAM(stream), gy, weight,
                       rstd, addend, mul2, out, out2, rows, D);
    return check_launch("layernorm_bwd_detached_kernel");
}

extern "C" int bcos_gelu_gate(const float* x, float* y, float* gate_out, int64_t n, void* stream) {
    if (!x || !y || n <= 0) return bcos_set_error(BCOS_E_INVAL, "bcos_gelu_gate: bad argument");
    hipLaunchKernelGGL(gelu_gate_kernel, dim3(grid_elems(n)), dim3(TPB), 0, STREAM(stream), x, y, gate_out, n);
    return check_launch("gelu_gate_kernel");
}

extern "C" int bcos_add_rows_bcast(float* x, const float* pe, int64_t total, int64_t period, void* stream) {
    if (!x || !pe || total <= 0 || period <= 0 || total % 4 || period % 4 || total % period)
        return bcos_set_error(BCOS_E_INVAL, "bcos_add_rows_bcast: sizes must be multiples of 4 and of the period");
    hipLaunchKernelGGL(add_rows_bcast_kernel, dim3(grid_elems(total / 4)), dim3(TPB), 0, STREAM(stream), x, pe,
                       total / 4, period / 4);
    return check_launch("add_rows_bcast_kernel");
}

static int attn_lds_ok(int T, size_t floats, const void* fn, const char* what) {
    const size_t bytes = floats * sizeof(float);
    (void)T;
    if (bytes > 160 * 1024) return bcos_set_error(BCOS_E_NOSUP, "attention: sequence too long for the LDS-resident kernel");
    hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (err != hipSuccess) return bcos_set_hip_error(what, err);
    return BCOS_OK;
}

extern "C" int bcos_attention_fwd(const float* qkv, float* out, float* stats, int B, int T, int H, int Dh, float scale,
                                  void* stream) {
    if (!qkv || !out || B <= 0 || T <= 0 || H <= 0) return bcos_set_error(BCOS_E_INVAL, "bcos_attention_fwd: bad argument");
    if (Dh != DH) return bcos_set_error(BCOS_E_NOSUP, "bcos_attention_fwd: head dimension must be 64");
    const size_t floats = (size_t)2 * T * DH;
    int rc = attn_lds_ok(T, floats, reinterpret_cast<const void*>(attention_fwd_kernel), "hipFuncSetAttribute(attention_fwd)");
    if (rc) return rc;
    hipLaunchKernelGGL(attention_fwd_kernel, dim3((unsigned)(B * H)), dim3(TPB), floats * sizeof(float), STREAM(stream), qkv,
                       out, stats, B, T, H, scale);
    return check_launch("attention_fwd_kernel");
}

extern "C" int bcos_attention_bwd_v(const float* qkv, const float* stats, const float* gout, float* gv, int B, int T,
                                    int H, int Dh, float scale, void* stream) {
    if (!qkv || !stats || !gout || !gv || B <= 0 || T <= 0 || H <= 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_attention_bwd_v: bad argument");
    if (Dh != DH) return bcos_set_error(BCOS_E_NOSUP, "bcos_attention_bwd_v: head dimension must be 64");
    const size_t floats = (size_t)2 * T * DH + 2 * T;
    int rc = attn_lds_ok(T, floats, reinterpret_cast<const void*>(attention_bwd_v_kernel), "hipFuncSetAttribute(attention_bwd)");
    if (rc) return rc;
    hipLaunchKernelGGL(attention_bwd_v_kernel, dim3((unsigned)(B * H)), dim3(TPB), floats * sizeof(float), STREAM(stream),
                       qkv, stats, gout, gv, B, T, H, scale);
    return check_launch("attention_bwd_v_kernel");
}

extern "C" int bcos_finalize_explanation_patches(const float* gp, const float* x, const float* std6, float* weights_out,
                                                 float* contrib_out, int N, int Cx, int H, int W, int patch, int Cpad,
                                                 int add_inverse, void* stream) {
    if (!gp || !x || !std6 || (!weights_out && !contrib_out) || N <= 0 || H <= 0 || W <= 0 || patch <= 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_finalize_explanation_patches: bad argument");
    if (H % patch || W % patch || Cx != (add_inverse ? 3 : 6) || Cpad < 6)
        return bcos_set_error(BCOS_E_INVAL, "bcos_finalize_explanation_patches: bad geometry");
    hipLaunchKernelGGL(finalize_patches_kernel, dim3(grid_elems((int64_t)N * H * W)), dim3(TPB), 0, STREAM(stream), gp, x,
                       std6, weights_out, contrib_out, N, Cx, H, W, patch, Cpad, add_inverse);
    return check_launch("finalize_patches_kernel");
}
