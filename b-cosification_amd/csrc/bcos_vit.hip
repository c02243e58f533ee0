// bcos_vit.hip -- the non-B-cos pieces of B-cosified transformers on gfx950:
// DetachableLayerNorm (bcos/modules/norms/centered_norms.py:187-245), MyGELU (bcosify_vit.py:27-32), the softmax
// attention of bcos/models/vit.py:143-158 / bcos/modules/bcosattnpool.py:22-59 with q,k detached in explanation
// mode, the token positional-embedding add and the un-patchify end of the ViT explanation pass.
// LayerNorm / GELU / embedding add are streaming kernels (one wavefront per LayerNorm row); the attention runs on fp32
// MFMA, one workgroup per (image, head), with the walked operand and the transposed value operand resident in LDS and
// the probabilities kept in registers between the two products.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include <initializer_list>
#include <atomic>
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
__device__ inline unsigned wave_max_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o));
    return v;
}
__device__ inline unsigned abs_bits4(const f32x4 v) {
    return max(max(__float_as_uint(v[0]) & 0x7fffffffu, __float_as_uint(v[1]) & 0x7fffffffu),
               max(__float_as_uint(v[2]) & 0x7fffffffu, __float_as_uint(v[3]) & 0x7fffffffu));
}

// Rows of up to 256 floats (D % 4 == 0): the row lives in ONE float4 per lane -- read once, two wavefront reductions, written
// once, 16-byte accesses -- and the row's max |y| (what the split-f16 contraction reading y scales its operand by) falls out
// of the registers.  Other widths take the strided loop.
template <bool VEC>
__global__ __launch_bounds__(TPB) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ y,
                                                            float* __restrict__ rstd_out, unsigned* __restrict__ absmax_out,
                                                            int64_t rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * TPB) >> 6;
    if constexpr (VEC) {
        const bool on = lane * 4 < D;
        f32x4 wv = {1.f, 1.f, 1.f, 1.f}, bv = {0.f, 0.f, 0.f, 0.f};
        if (on && w) wv = *reinterpret_cast<const f32x4*>(w + lane * 4);
        if (on && b) bv = *reinterpret_cast<const f32x4*>(b + lane * 4);
        for (int64_t r = wave; r < rows; r += nwaves) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (on) v = *reinterpret_cast<const f32x4*>(x + r * D + lane * 4);
            const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) / (float)D;
            f32x4 d = v - mean;
            if (!on) d = f32x4{0.f, 0.f, 0.f, 0.f};
            const float var = wave_sum(fmaf(d[0], d[0], d[1] * d[1]) + fmaf(d[2], d[2], d[3] * d[3])) / (float)D;
            const float sd = sqrtf(var + eps);
            f32x4 o = {d[0] / sd, d[1] / sd, d[2] / sd, d[3] / sd};
            o = o * wv + bv;
            if (on) *reinterpret_cast<f32x4*>(y + r * D + lane * 4) = o;
            if (rstd_out && lane == 0) rstd_out[r] = 1.0f / sd;
            if (absmax_out) {
                const unsigned m = wave_max_u32(on ? abs_bits4(o) : 0u);
                if (lane == 0) absmax_out[r] = m;
            }
        }
    } else {
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
            unsigned m = 0u;
            for (int c = lane; c < D; c += 64) {
                float o = (src[c] - mean) / sd;
                if (w) o *= w[c];
                if (b) o += b[c];
                dst[c] = o;
                m = max(m, __float_as_uint(o) & 0x7fffffffu);
            }
            if (rstd_out && lane == 0) rstd_out[r] = 1.0f / sd;
            if (absmax_out) {
                m = wave_max_u32(m);
                if (lane == 0) absmax_out[r] = m;
            }
        }
    }
}

// Row statistics of the LayerNorm for a contraction that reads x itself (bcos_epilogue.row_scale / a_sumsq): same reductions as
// layernorm_fwd_kernel (mean, then the centered second moment), plus |gamma xhat + beta|^2 and max |x|; nothing row-sized is written.
__global__ __launch_bounds__(TPB) void layernorm_stats_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ b, float* __restrict__ rstd_out,
                                                              float* __restrict__ zss_out, unsigned* __restrict__ absmax_out,
                                                              int64_t rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * TPB) >> 6;
    for (int64_t r = wave; r < rows; r += nwaves) {
        const float* src = x + r * D;
        float s = 0.f;
        unsigned m = 0u;
        for (int c = lane; c < D; c += 64) { s += src[c]; m = max(m, __float_as_uint(src[c]) & 0x7fffffffu); }
        const float mean = wave_sum(s) / (float)D;
        float v = 0.f;
        for (int c = lane; c < D; c += 64) { const float d = src[c] - mean; v = fmaf(d, d, v); }
        const float var = wave_sum(v) / (float)D;
        const float sd = sqrtf(var + eps);
        if (lane == 0) rstd_out[r] = 1.0f / sd;
        if (zss_out) {
            float zs = 0.f;
            for (int c = lane; c < D; c += 64) {
                float o = (src[c] - mean) / sd;
                if (w) o *= w[c];
                if (b) o += b[c];
                zs = fmaf(o, o, zs);
            }
            zs = wave_sum(zs);
            if (lane == 0) zss_out[r] = zs;
        }
        if (absmax_out) {
            m = wave_max_u32(m);
            if (lane == 0) absmax_out[r] = m;
        }
    }
}

// The same statistics for rows of up to 256 floats (D % 4 == 0), SIXTEEN lanes per row: a wavefront holds four rows at once (NV
// 16-byte pieces per lane, four-step reductions), so a workgroup keeps 16 rows of loads in flight where the one-row-per-wavefront
// form above keeps four -- the kernel is one dependent chain per row (load, mean, centred moment, output norm) and nothing else.
__device__ inline float sum16(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <int NV>
__global__ __launch_bounds__(TPB) void layernorm_stats16_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ b, float* __restrict__ rstd_out,
                                                                float* __restrict__ zss_out, unsigned* __restrict__ absmax_out,
                                                                int64_t rows, int D, float eps) {
    const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
    const int64_t wave = ((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * TPB) >> 6;
    f32x4 wv[NV], bv[NV];
    bool on[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (sub + 16 * i) * 4;
        on[i] = c < D;
        wv[i] = (on[i] && w) ? *reinterpret_cast<const f32x4*>(w + c) : f32x4{1.f, 1.f, 1.f, 1.f};
        bv[i] = (on[i] && b) ? *reinterpret_cast<const f32x4*>(b + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int64_t r0 = wave * 4; r0 < rows; r0 += nwaves * 4) {
        const int64_t r = r0 + grp;
        const bool live = r < rows;
        f32x4 v[NV];
        float s = 0.f;
        unsigned m = 0u;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i] = (live && on[i]) ? *reinterpret_cast<const f32x4*>(x + r * D + (sub + 16 * i) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
            m = max(m, abs_bits4(v[i]));
        }
        const float mean = sum16(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i] = on[i] ? v[i] - mean : f32x4{0.f, 0.f, 0.f, 0.f};
            q += fmaf(v[i][0], v[i][0], v[i][1] * v[i][1]) + fmaf(v[i][2], v[i][2], v[i][3] * v[i][3]);
        }
        const float sd = sqrtf(sum16(q) / (float)D + eps);
        if (sub == 0 && live) rstd_out[r] = 1.0f / sd;
        if (zss_out) {
            float zs = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                f32x4 o = {v[i][0] / sd, v[i][1] / sd, v[i][2] / sd, v[i][3] / sd};       // (the value layernorm_fwd_kernel writes)
                o = o * wv[i] + bv[i];
                if (!on[i]) o = f32x4{0.f, 0.f, 0.f, 0.f};
                zs += fmaf(o[0], o[0], o[1] * o[1]) + fmaf(o[2], o[2], o[3] * o[3]);
            }
            zs = sum16(zs);
            if (sub == 0 && live) zss_out[r] = zs;
        }
        if (absmax_out) {
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
            if (sub == 0 && live) absmax_out[r] = m;
        }
    }
}

// explanation mode: the variance is a constant, the mean is not (centered_norms.py:204-215):
//   y = w * (x - mean(x)) / std   =>   gx = h - mean(h),  h = gy * w / std
// out = gx (+ addend); out2 = out * mul2 (the scale of the B-cos layer that produced x), both optional extras.
template <bool VEC>
__global__ __launch_bounds__(TPB) void layernorm_bwd_detached_kernel(const float* __restrict__ gy,
                                                                     const float* __restrict__ w,
                                                                     const float* __restrict__ rstd,
                                                                     const float* __restrict__ addend,
                                                                     const float* __restrict__ mul2,
                                                                     float* __restrict__ out, float* __restrict__ out2,
                                                                     unsigned* __restrict__ absmax2_out, int64_t rows, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * TPB) >> 6;
    if constexpr (VEC) {
        const bool on = lane * 4 < D;
        f32x4 wv = {1.f, 1.f, 1.f, 1.f};
        if (on && w) wv = *reinterpret_cast<const f32x4*>(w + lane * 4);
        for (int64_t r = wave; r < rows; r += nwaves) {
            const int64_t off = r * D + lane * 4;
            f32x4 g = {0.f, 0.f, 0.f, 0.f}, ad = {0.f, 0.f, 0.f, 0.f}, m2 = {1.f, 1.f, 1.f, 1.f};
            if (on) g = *reinterpret_cast<const f32x4*>(gy + off);
            if (on && addend) ad = *reinterpret_cast<const f32x4*>(addend + off);
            if (on && mul2) m2 = *reinterpret_cast<const f32x4*>(mul2 + off);
            const float rs = rstd[r];
            const f32x4 h = g * wv * rs;
            const float mh = wave_sum((h[0] + h[1]) + (h[2] + h[3])) / (float)D;
            const f32x4 o = h - mh + ad;
            const f32x4 o2 = o * m2;
            if (on && out) *reinterpret_cast<f32x4*>(out + off) = o;
            if (on && out2) *reinterpret_cast<f32x4*>(out2 + off) = o2;
            if (absmax2_out) {
                const unsigned m = wave_max_u32(on ? abs_bits4(o2) : 0u);
                if (lane == 0) absmax2_out[r] = m;
            }
        }
    } else {
        for (int64_t r = wave; r < rows; r += nwaves) {
            const float* g = gy + r * D;
            const float rs = rstd[r];
            float s = 0.f;
            for (int c = lane; c < D; c += 64) s += g[c] * (w ? w[c] : 1.f) * rs;
            const float mh = wave_sum(s) / (float)D;
            unsigned m = 0u;
            for (int c = lane; c < D; c += 64) {
                float o = g[c] * (w ? w[c] : 1.f) * rs - mh;
                if (addend) o += addend[r * D + c];
                if (out) out[r * D + c] = o;
                const float o2 = mul2 ? o * mul2[r * D + c] : o;
                if (out2) out2[r * D + c] = o2;
                m = max(m, __float_as_uint(o2) & 0x7fffffffu);
            }
            if (absmax2_out) {
                m = wave_max_u32(m);
                if (lane == 0) absmax2_out[r] = m;
            }
        }
    }
}

// ---- GELU with detachable gate: y = gate(x) * x, gate = 0.5 (1 + erf(x / sqrt 2)) ------------------------------
__global__ __launch_bounds__(TPB) void gelu_gate_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        float* __restrict__ gate_out, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * TPB;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) {
        const float v = x[i];
        const float gate = bcos_gelu_gate(v);
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

// ---- softmax attention on fp32 MFMA, one workgroup per (batch, head), head dim 64 -----------------------------
// qkv: [B, T, 3*H*64] laid out "(three h d)" like vit.py:145-146; out: [B, T, H*64]
// stats: [B, H, T, 2] = (row max, 1 / row sum) kept for the explanation backward instead of the T x T matrix.
//
// One kernel serves both directions.  Call the operand that is walked in 32-row chunks X, the one that owns the
// output rows Y, and the value operand Z:
//   forward   out[q]  = sum_k softmax_k(scale q.k) v[k]          X = K, Y = Q, Z = V,    softmax statistics online
//   backward  gv[k]   = sum_q P[q][k] gout[q]   (q, k detached: vit.py:148-151, bcosattnpool.py:37-39)
//                                                                X = Q, Y = K, Z = gout, P from the stored stats
// A wavefront owns a 32-row tile of Y (held in registers as MFMA B fragments) and walks X (LDS, [Tpad][68]):
//   S^T tile = X_chunk . Y_tile^T on v_mfma_f32_32x32x2_f32: D layout puts the Y row on the lane (column) and 16 X
//   rows in registers, so per-Y-row softmax reductions are 16 in-register ops + one exchange with lane^32, and the
//   probabilities are ALREADY in the B-operand layout of the second product  O^T[d][y] += Z^T[d][x] . P^T[x][y]
//   (register r of lane half hf is X row 8(r>>2) + 4hf + (r&3); the A operand Z^T is read from LDS, [64][Tpad+4],
//   with the same row permutation) -- the T x T matrix never leaves the register file.
// Both LDS layouts are padded so that every ds_read_b128 of a 16-lane group hits 16 distinct bank quads.
constexpr int DH = 64;
constexpr int AT_XLD = DH + 4;      // floats per X row in LDS

typedef float f32x16 __attribute__((ext_vector_type(16)));

// 8 wavefronts per workgroup: the 7 column tiles of a 196-token sequence run in one round (with 4 they took two, the
// second one quarter full); the workgroup owns the CU either way (119 KB of LDS)
constexpr int ATPB = 512;

template <bool BWD>
__global__ __launch_bounds__(ATPB) void attention_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ zsrc,
                                                             float* __restrict__ out, float* __restrict__ stats_out,
                                                             const float* __restrict__ stats_in, unsigned* __restrict__ absmax_out,
                                                             int B, int T, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Tpad = (T + 31) & ~31;
    const int zld = Tpad + 4;
    float* sX = smem;                                  // [Tpad][68]
    float* sZT = smem + (size_t)Tpad * AT_XLD;         // [64][Tpad + 4]
    float* sS = sZT + (size_t)DH * zld;                // [Tpad][2]   (backward only)
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int inner = H * DH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* base = qkv + (int64_t)b * T * 3 * inner + h * DH;
    const float* xsrc = base + (BWD ? 0 : inner);      // K (forward) / Q (backward), row stride 3*inner
    const float* ysrc = base + (BWD ? inner : 0);      // Q (forward) / K (backward)
    const float* zrow0 = BWD ? zsrc + (int64_t)b * T * inner + h * DH : base + 2 * inner;
    const int zstride = BWD ? inner : 3 * inner;

    for (int i = tid; i < Tpad * (DH / 4); i += ATPB) {
        const int t = i / (DH / 4), c4 = i % (DH / 4);
        f32x4 xv = {0.f, 0.f, 0.f, 0.f}, zv = {0.f, 0.f, 0.f, 0.f};
        if (t < T) {
            xv = *reinterpret_cast<const f32x4*>(xsrc + (int64_t)t * 3 * inner + c4 * 4);
            zv = *reinterpret_cast<const f32x4*>(zrow0 + (int64_t)t * zstride + c4 * 4);
        }
        *reinterpret_cast<f32x4*>(sX + t * AT_XLD + c4 * 4) = xv;
#pragma unroll
        for (int q = 0; q < 4; ++q) sZT[(c4 * 4 + q) * zld + t] = zv[q];
    }
    if (BWD) {
        const float* st = stats_in + ((int64_t)b * H + h) * T * 2;
        for (int i = tid; i < Tpad; i += ATPB) {
            sS[2 * i] = i < T ? st[2 * i] : INFINITY;          // padded query rows: exp(s - inf) * 0 = 0
            sS[2 * i + 1] = i < T ? st[2 * i + 1] : 0.f;
        }
    }
    __syncthreads();

    const int col = lane & 31, hf = lane >> 5;
    const int ntiles = Tpad / 32;
    for (int tile = wave; tile < ntiles; tile += ATPB / 64) {
        const int y = tile * 32 + col;                 // the Y row (query fwd / key bwd) of this lane
        f32x4 yf[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            yf[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (y < T) yf[g] = *reinterpret_cast<const f32x4*>(ysrc + (int64_t)y * 3 * inner + 8 * g + 4 * hf);
        }
        f32x16 o0, o1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
        float m_run = -INFINITY, l_run = 0.f;
        for (int chunk = 0; chunk < ntiles; ++chunk) {
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
            const float* xr = sX + (chunk * 32 + col) * AT_XLD + 4 * hf;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f32x4 xa = *reinterpret_cast<const f32x4*>(xr + 8 * g);
#pragma unroll
                for (int c = 0; c < 4; ++c) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[c], yf[g][c], sacc, 0, 0, 0);
            }
            // register r <-> X row  chunk*32 + 8(r>>2) + 4hf + (r&3)
            float pr[16];
            if (!BWD) {
                float mx = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int x = chunk * 32 + 8 * (r >> 2) + 4 * hf + (r & 3);
                    pr[r] = x < T ? sacc[r] * scale : -INFINITY;
                    mx = fmaxf(mx, pr[r]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float m_new = fmaxf(m_run, mx);          // finite: every chunk holds at least one valid key
                const float alpha = expf(m_run - m_new);
                float sum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { pr[r] = expf(pr[r] - m_new); sum += pr[r]; }
                sum += __shfl_xor(sum, 32);
                l_run = l_run * alpha + sum;
                m_run = m_new;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float* st = sS + 2 * (chunk * 32 + 8 * g + 4 * hf);
                    const f32x4 s01 = *reinterpret_cast<const f32x4*>(st), s23 = *reinterpret_cast<const f32x4*>(st + 4);
                    pr[4 * g + 0] = expf(sacc[4 * g + 0] * scale - s01[0]) * s01[1];
                    pr[4 * g + 1] = expf(sacc[4 * g + 1] * scale - s01[2]) * s01[3];
                    pr[4 * g + 2] = expf(sacc[4 * g + 2] * scale - s23[0]) * s23[1];
                    pr[4 * g + 3] = expf(sacc[4 * g + 3] * scale - s23[2]) * s23[3];
                }
            }
            const float* z0 = sZT + col * zld + chunk * 32 + 4 * hf;
            const float* z1 = z0 + 32 * zld;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 za = *reinterpret_cast<const f32x4*>(z0 + 8 * g);
                const f32x4 zb = *reinterpret_cast<const f32x4*>(z1 + 8 * g);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(za[c], pr[4 * g + c], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(zb[c], pr[4 * g + c], o1, 0, 0, 0);
                }
            }
        }
        if (y < T) {
            const float rl = BWD ? 1.0f : 1.0f / l_run;
            float* orow = out + ((int64_t)b * T + y) * inner + h * DH + 4 * hf;    // d = 32 dt + 8 g + 4 hf + c
            unsigned mx = 0u;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 va = f32x4{o0[4 * g] * rl, o0[4 * g + 1] * rl, o0[4 * g + 2] * rl, o0[4 * g + 3] * rl};
                const f32x4 vb = f32x4{o1[4 * g] * rl, o1[4 * g + 1] * rl, o1[4 * g + 2] * rl, o1[4 * g + 3] * rl};
                *reinterpret_cast<f32x4*>(orow + 8 * g) = va;
                *reinterpret_cast<f32x4*>(orow + 32 + 8 * g) = vb;
                mx = max(mx, max(abs_bits4(va), abs_bits4(vb)));
            }
            if (absmax_out) {         // max |out| of the token row over this head's 64 columns; the heads meet in one atomic max (zeroed buffer)
                mx = max(mx, (unsigned)__shfl_xor((int)mx, 32));
                if (hf == 0) atomicMax(absmax_out + (int64_t)b * T + y, mx);
            }
            if (!BWD && stats_out && hf == 0) {
                float* st = stats_out + (((int64_t)b * H + h) * T + y) * 2;
                st[0] = m_run;
                st[1] = rl;
            }
        }
    }
}

// ---- the same attention on the f16 matrix pipe (round 3) -------------------------------------------------------------------------
// Same structure -- one workgroup per (image, head), the walked operand X and the transposed value operand Z^T resident in LDS,
// a wavefront owns 32 Y rows, the probabilities never leave the register file -- but both products run as THREE
// v_mfma_f32_32x32x16_f16 on exact two-way splits (x 2^e = h + l, h, l fp16: l_a h_b + h_a l_b + h_a h_b, fp32 accumulation;
// the arithmetic of the contraction kernel, csrc/bcos_tapconv.hip) instead of fp32 MFMAs at 1/16 of the rate:
//   * X rows (keys fwd / queries bwd) and Z^T rows (one per feature d) are scaled per row by a power of two, split ONCE by the
//     loading pass and kept in LDS as (h | l) planes -- the same bytes as the fp32 images they replace;
//   * Y rows are split once per tile in registers; the probabilities (<= 1) are split per chunk after a fixed 2^14;
//   * the powers of two are undone exactly: per X row inside the softmax argument, per Y row in the same factor, per d at the end.
// The k index of both contractions is stored permuted (the middle two 4-blocks of every 16 swapped) so that a lane's 8 k values
// of one matrix instruction are one ds_read_b128 for the LDS operand and plain register order for the register operand.
typedef _Float16 af16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 af16x4 __attribute__((ext_vector_type(4)));
constexpr int AH_XROW = 144;        // bytes per X row and plane: 64 f16 + 16 (16 rows at distinct 16-byte bank groups)
constexpr float AH_LOG2E = 1.4426950408889634f, AH_LN2 = 0.6931471805599453f;
#ifndef AH_KO
#define AH_KO 0
#endif
BCOS_DEV_SWITCH(AH_KO, 0);
#ifndef AH_TWO_CHAINS
#define AH_TWO_CHAINS 0
#endif
constexpr int AH_MAXIT = 9;         // 32-row blocks of the walked operand a workgroup holds at most (288 tokens: what the LDS planes allow)

__device__ __forceinline__ void split_scale_exp(unsigned maxbits, float& sc, float& inv) {
    unsigned E = maxbits >> 23;
    E = E < 15u ? 15u : E;
    sc = __uint_as_float((268u - E) << 23);      // max * sc in [2^14, 2^15)
    inv = __uint_as_float((E - 14u) << 23);
}

template <bool BWD>
__global__ __launch_bounds__(ATPB) void attention_h2_kernel(const float* __restrict__ qkv, const float* __restrict__ zsrc,
                                                           float* __restrict__ out, float* __restrict__ stats_out,
                                                           const float* __restrict__ stats_in, unsigned* __restrict__ absmax_out,
                                                           int B, int T, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Tpad = (T + 31) & ~31;
    const int zrow = Tpad * 2 + 16;                     // bytes per Z^T row and plane
    char* sXh = reinterpret_cast<char*>(smem);
    char* sXl = sXh + (size_t)Tpad * AH_XROW;
    char* sZh = sXl + (size_t)Tpad * AH_XROW;
    char* sZl = sZh + (size_t)DH * zrow;
    float* sXinv = reinterpret_cast<float*>(sZl + (size_t)DH * zrow);     // [Tpad]
    float* sZinv = sXinv + Tpad;                                          // [64]
    unsigned* sZmax = reinterpret_cast<unsigned*>(sZinv + DH);            // [64]
    float* sS = reinterpret_cast<float*>(sZmax + DH);                     // [Tpad][2]   (backward only)
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int inner = H * DH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* base = qkv + (int64_t)b * T * 3 * inner + h * DH;
    const float* xsrc = base + (BWD ? 0 : inner);      // K (forward) / Q (backward), row stride 3*inner
    const float* ysrc = base + (BWD ? inner : 0);      // Q (forward) / K (backward)
    const float* zrow0 = BWD ? zsrc + (int64_t)b * T * inner + h * DH : base + 2 * inner;
    const int zstride = BWD ? inner : 3 * inner;

    if (tid < DH) sZmax[tid] = 0u;
    __syncthreads();
    // pass 1: X rows -> scaled (h | l) planes; per-d maxima of Z.  A thread keeps its 16-byte column c4 = tid % 16 throughout and
    // owns the rows (tid >> 4) + 32 it.  ALL of its X and Z pieces are requested up front (AH_MAXIT x 2 loads in flight, one
    // exposed memory latency per workgroup) and the Z pieces stay in registers for pass 2 -- round 3 walked the rows in a loop of
    // dependent (load, reduce, store) iterations and read Z a second time: with one workgroup per CU (the planes fill the LDS)
    // nothing else covered those latencies.
    const int c4 = tid & 15;
    const int xpos = (16 * (c4 >> 2) + 4 * (((c4 & 3) == 1) ? 2 : ((c4 & 3) == 2) ? 1 : (c4 & 3))) * 2;     // byte offset of d block c4
    const int nit = Tpad / 32;                     // <= AH_MAXIT (host: the planes must fit the LDS)
    f32x4 zreg[AH_MAXIT];
    {
        f32x4 xreg[AH_MAXIT];
#pragma unroll
        for (int it = 0; it < AH_MAXIT; ++it) {
            const int t = (tid >> 4) + 32 * it;
            xreg[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            zreg[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (it < nit && t < T) {
                xreg[it] = *reinterpret_cast<const f32x4*>(xsrc + (int64_t)t * 3 * inner + c4 * 4);
                zreg[it] = *reinterpret_cast<const f32x4*>(zrow0 + (int64_t)t * zstride + c4 * 4);
            }
        }
        unsigned zm[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int it = 0; it < AH_MAXIT; ++it) {
            if (it < nit) {
                const int t = (tid >> 4) + 32 * it;
                const f32x4 xv = xreg[it], zv = zreg[it];
                unsigned m = abs_bits4(xv);
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
                float sc, inv;
                split_scale_exp(m, sc, inv);
                af16x4 hh, ll;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v = xv[q] * sc;
                    hh[q] = (_Float16)v;
                    ll[q] = (_Float16)(v - (float)hh[q]);
                    zm[q] = max(zm[q], __float_as_uint(zv[q]) & 0x7fffffffu);
                }
                *reinterpret_cast<af16x4*>(sXh + t * AH_XROW + xpos) = hh;
                *reinterpret_cast<af16x4*>(sXl + t * AH_XROW + xpos) = ll;
                if (c4 == 0) sXinv[t] = inv;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) atomicMax(&sZmax[c4 * 4 + q], zm[q]);
    }
    if (BWD) {
        const float* st = stats_in + ((int64_t)b * H + h) * T * 2;
        for (int i = tid; i < Tpad; i += ATPB) {
            sS[2 * i] = i < T ? st[2 * i] : INFINITY;          // padded query rows: exp(s - inf) * 0 = 0
            sS[2 * i + 1] = i < T ? st[2 * i + 1] : 0.f;
        }
    }
    __syncthreads();
    // pass 2: Z -> scaled, split, transposed planes (token index permuted like d above), from the registers of pass 1
    {
        float zsc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float inv;
            split_scale_exp(sZmax[c4 * 4 + q], zsc[q], inv);
            if (tid < 16) sZinv[c4 * 4 + q] = inv;
        }
#pragma unroll
        for (int it = 0; it < AH_MAXIT; ++it) {
            if (it < nit && !(AH_KO & 2)) {
                const int t = (tid >> 4) + 32 * it;
                const f32x4 zv = zreg[it];
                const int tb = (t >> 2) & 3;
                const int tpos = ((t & ~15) + 4 * (tb == 1 ? 2 : tb == 2 ? 1 : tb) + (t & 3)) * 2;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v = zv[q] * zsc[q];
                    const _Float16 hh = (_Float16)v;
                    *reinterpret_cast<_Float16*>(sZh + (c4 * 4 + q) * zrow + tpos) = hh;
                    *reinterpret_cast<_Float16*>(sZl + (c4 * 4 + q) * zrow + tpos) = (_Float16)(v - (float)hh);
                }
            }
        }
    }
    __syncthreads();

    const int col = lane & 31, hf = lane >> 5;
    const int ntiles = (AH_KO & 1) ? 0 : Tpad / 32;      // (AH_KO: development knock-outs, timing only)
    for (int tile = wave; tile < ntiles; tile += ATPB / 64) {
        const int y = tile * 32 + col;                 // the Y row (query fwd / key bwd) of this lane
        f32x4 yf[8];
        unsigned ym = 0u;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            yf[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (y < T) yf[g] = *reinterpret_cast<const f32x4*>(ysrc + (int64_t)y * 3 * inner + 8 * g + 4 * hf);
            ym = max(ym, abs_bits4(yf[g]));
        }
        ym = max(ym, (unsigned)__shfl_xor((int)ym, 32));
        float ysc, yinv;
        split_scale_exp(ym, ysc, yinv);
        af16x8 yh[4], yl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float v = yf[2 * j + (q >> 2)][q & 3] * ysc;
                yh[j][q] = (_Float16)v;
                yl[j][q] = (_Float16)(v - (float)yh[j][q]);
            }
        // forward: scores in the base-2 domain (the softmax runs on v_exp_f32 directly); backward: natural scores against the recorded
        // natural maxima, the difference converted
        const float sfac = BWD ? scale * yinv : scale * yinv * AH_LOG2E;
        f32x16 o0, o1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
        float m_run = -INFINITY, l_run = 0.f;
        for (int chunk = 0; chunk < ntiles; ++chunk) {
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
            const char* xh = sXh + (chunk * 32 + col) * AH_XROW + 16 * hf;
            const char* xl = sXl + (chunk * 32 + col) * AH_XROW + 16 * hf;
#if AH_TWO_CHAINS
            // two accumulation chains (k blocks 0-1 and 2-3), added at the end: a v_mfma directly behind the one it depends on waits
            // for its result, and with one workgroup per CU there are at most two waves per SIMD to fill that wait
            f32x16 sacc2;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc2[r] = 0.f;
            af16x8 ah4[4], al4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ah4[j] = *reinterpret_cast<const af16x8*>(xh + 32 * j);
                al4[j] = *reinterpret_cast<const af16x8*>(xl + 32 * j);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al4[j], yh[j], sacc, 0, 0, 0);
                sacc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al4[j + 2], yh[j + 2], sacc2, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah4[j], yl[j], sacc, 0, 0, 0);
                sacc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah4[j + 2], yl[j + 2], sacc2, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah4[j], yh[j], sacc, 0, 0, 0);
                sacc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah4[j + 2], yh[j + 2], sacc2, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] += sacc2[r];
#else
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const af16x8 ah = *reinterpret_cast<const af16x8*>(xh + 32 * j);
                const af16x8 al = *reinterpret_cast<const af16x8*>(xl + 32 * j);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, yh[j], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, yl[j], sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, yh[j], sacc, 0, 0, 0);
            }
#endif
            // register r <-> X row  chunk*32 + 8(r>>2) + 4hf + (r&3); undo the X row's power of two
            float pr[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 xi = *reinterpret_cast<const f32x4*>(sXinv + chunk * 32 + 8 * g + 4 * hf);
#pragma unroll
                for (int c = 0; c < 4; ++c) pr[4 * g + c] = sacc[4 * g + c] * sfac * xi[c];
            }
            // (round 4) the step was bound by its ~530 vector instructions per 24 matrix instructions: the exponentials are the
            // hardware's base-2 ones (1 ulp; the scores already carry the log2 e), keys beyond T exist in the last chunk only, and the
            // running output is rescaled only when some row's maximum moved
            if (!BWD) {
                if (chunk + 1 == ntiles) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int x = chunk * 32 + 8 * (r >> 2) + 4 * hf + (r & 3);
                        pr[r] = x < T ? pr[r] : -INFINITY;
                    }
                }
                float mx = fmaxf(fmaxf(fmaxf(pr[0], pr[1]), fmaxf(pr[2], pr[3])), fmaxf(fmaxf(pr[4], pr[5]), fmaxf(pr[6], pr[7])));
                mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(pr[8], pr[9]), fmaxf(pr[10], pr[11])), fmaxf(fmaxf(pr[12], pr[13]), fmaxf(pr[14], pr[15]))));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float m_new = fmaxf(m_run, mx);          // finite: every chunk holds at least one valid key
                float sum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { pr[r] = __builtin_amdgcn_exp2f(pr[r] - m_new); sum += pr[r]; }
                sum += __shfl_xor(sum, 32);
                if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0ull) {
                    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);       // (exactly 1 for the rows whose maximum stayed)
                    l_run *= alpha;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
                    m_run = m_new;
                }
                l_run += sum;
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float* st = sS + 2 * (chunk * 32 + 8 * g + 4 * hf);
                    const f32x4 s01 = *reinterpret_cast<const f32x4*>(st), s23 = *reinterpret_cast<const f32x4*>(st + 4);
                    pr[4 * g + 0] = __builtin_amdgcn_exp2f((pr[4 * g + 0] - s01[0]) * AH_LOG2E) * s01[1];
                    pr[4 * g + 1] = __builtin_amdgcn_exp2f((pr[4 * g + 1] - s01[2]) * AH_LOG2E) * s01[3];
                    pr[4 * g + 2] = __builtin_amdgcn_exp2f((pr[4 * g + 2] - s23[0]) * AH_LOG2E) * s23[1];
                    pr[4 * g + 3] = __builtin_amdgcn_exp2f((pr[4 * g + 3] - s23[2]) * AH_LOG2E) * s23[3];
                }
            }
            const char* zh = sZh + col * zrow + (chunk * 32 + 8 * hf) * 2;
            const char* zl = sZl + col * zrow + (chunk * 32 + 8 * hf) * 2;
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2) {
                af16x8 ph, pl;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float v = pr[8 * m2 + q] * 16384.0f;
                    ph[q] = (_Float16)v;
                    pl[q] = (_Float16)(v - (float)ph[q]);
                }
                const af16x8 zh0 = *reinterpret_cast<const af16x8*>(zh + 32 * m2), zl0 = *reinterpret_cast<const af16x8*>(zl + 32 * m2);
                const af16x8 zh1 = *reinterpret_cast<const af16x8*>(zh + 32 * zrow + 32 * m2);
                const af16x8 zl1 = *reinterpret_cast<const af16x8*>(zl + 32 * zrow + 32 * m2);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(zl0, ph, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(zl1, ph, o1, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh0, pl, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh1, pl, o1, 0, 0, 0);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh0, ph, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(zh1, ph, o1, 0, 0, 0);
            }
        }
        if (y < T) {
            const float rl = (BWD ? 1.0f : 1.0f / l_run) * (1.0f / 16384.0f);
            float* orow = out + ((int64_t)b * T + y) * inner + h * DH + 4 * hf;    // d = 32 dt + 8 g + 4 hf + c
            unsigned mx = 0u;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 za = *reinterpret_cast<const f32x4*>(sZinv + 8 * g + 4 * hf) * rl;
                const f32x4 zb = *reinterpret_cast<const f32x4*>(sZinv + 32 + 8 * g + 4 * hf) * rl;
                const f32x4 va = f32x4{o0[4 * g] * za[0], o0[4 * g + 1] * za[1], o0[4 * g + 2] * za[2], o0[4 * g + 3] * za[3]};
                const f32x4 vb = f32x4{o1[4 * g] * zb[0], o1[4 * g + 1] * zb[1], o1[4 * g + 2] * zb[2], o1[4 * g + 3] * zb[3]};
                *reinterpret_cast<f32x4*>(orow + 8 * g) = va;
                *reinterpret_cast<f32x4*>(orow + 32 + 8 * g) = vb;
                mx = max(mx, max(abs_bits4(va), abs_bits4(vb)));
            }
            if (absmax_out) {
                mx = max(mx, (unsigned)__shfl_xor((int)mx, 32));
                if (hf == 0) atomicMax(absmax_out + (int64_t)b * T + y, mx);
            }
            if (!BWD && stats_out && hf == 0) {
                float* st = stats_out + (((int64_t)b * H + h) * T + y) * 2;
                st[0] = m_run * AH_LN2;          // (the statistics tensor keeps the natural-log maximum of rounds 1-3)
                st[1] = 1.0f / l_run;
            }
        }
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
    const bool vec = Cpad == 8 && (reinterpret_cast<uintptr_t>(gp) & 15) == 0;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += stride) {
        const int64_t n = i / HW;
        const int64_t hw = i - n * HW;
        const int hh = (int)(hw / W), ww = (int)(hw - (int64_t)hh * W);
        const int pi = hh / ps, r = hh - pi * ps, pj = ww / ps, s = ww - pj * ps;
        const float* g = gp + ((((n * (H / ps) + pi) * gw + pj) * ps + r) * ps + s) * Cpad;
        const float* src = x + n * (int64_t)Cx * HW + hw;
        float gv[6];
        if (vec) {                                         // the usual padded pixel record: two 16-byte loads
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

// ---- GroupNorm over NHWC tensors (DetachableGroupNorm2d, centered_norms.py:93-160): one workgroup per (image, group) ------
// Group g owns the channels [g cg, (g + 1) cg) of every pixel.  Three passes over the group's HW x cg values (mean, centred
// sum of squares, normalise): the group of one image is at most a few MB and stays in L2 between the passes.
__device__ inline float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();                                   // red may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int k = 0; k < TPB / 64; ++k) t += red[k];
    return t;
}

__global__ __launch_bounds__(TPB) void groupnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ y,
                                                            float* __restrict__ rstd_out, int HW, int C, int G, float eps) {
    __shared__ float red[TPB / 64];
    const int n = blockIdx.x / G, g = blockIdx.x - n * G;
    const int cg = C / G;
    const int64_t base = (int64_t)n * HW * C + g * cg;
    const int64_t m = (int64_t)HW * cg;
    float s = 0.f;
    for (int64_t e = threadIdx.x; e < m; e += TPB) s += x[base + (e / cg) * C + e % cg];
    const float mean = block_sum(s, red) / (float)m;
    float v = 0.f;
    for (int64_t e = threadIdx.x; e < m; e += TPB) { const float d = x[base + (e / cg) * C + e % cg] - mean; v = fmaf(d, d, v); }
    const float var = block_sum(v, red) / (float)m;
    const float sd = sqrtf(var + eps);
    for (int64_t e = threadIdx.x; e < m; e += TPB) {
        const int c = (int)(e % cg);
        const int64_t i = base + (e / cg) * C + c;
        float o = (x[i] - mean) / sd;
        if (w) o *= w[g * cg + c];
        if (b) o += b[g * cg + c];
        y[i] = o;
    }
    if (rstd_out && threadIdx.x == 0) rstd_out[blockIdx.x] = 1.0f / sd;
}

// explanation mode (variance constant, mean differentiable):  gx = h - mean_group(h),  h = gy * w / std
__global__ __launch_bounds__(TPB) void groupnorm_bwd_detached_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                                     const float* __restrict__ rstd, float* __restrict__ gx,
                                                                     int HW, int C, int G) {
    __shared__ float red[TPB / 64];
    const int n = blockIdx.x / G, g = blockIdx.x - n * G;
    const int cg = C / G;
    const int64_t base = (int64_t)n * HW * C + g * cg;
    const int64_t m = (int64_t)HW * cg;
    const float rs = rstd[blockIdx.x];
    float s = 0.f;
    for (int64_t e = threadIdx.x; e < m; e += TPB) {
        const int c = (int)(e % cg);
        s += gy[base + (e / cg) * C + c] * (w ? w[g * cg + c] : 1.0f) * rs;
    }
    const float mh = block_sum(s, red) / (float)m;
    for (int64_t e = threadIdx.x; e < m; e += TPB) {
        const int c = (int)(e % cg);
        const int64_t i = base + (e / cg) * C + c;
        gx[i] = gy[i] * (w ? w[g * cg + c] : 1.0f) * rs - mh;
    }
}

// GroupNorm, nothing detached (F.group_norm's gradient): per (image, group) with m values,  x_hat = (x - mean) rstd, h = gy w:
//   gx = rstd (h - mean(h) - x_hat mean(h x_hat));  x_hat is written out for the weight gradient (a column reduction)
__global__ __launch_bounds__(TPB) void groupnorm_bwd_full_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                 const float* __restrict__ w, const float* __restrict__ rstd,
                                                                 float* __restrict__ gx, float* __restrict__ xhat, int HW, int C, int G) {
    __shared__ float red[TPB / 64];
    const int n = blockIdx.x / G, g = blockIdx.x - n * G;
    const int cg = C / G;
    const int64_t base = (int64_t)n * HW * C + g * cg;
    const int64_t m = (int64_t)HW * cg;
    const float rs = rstd[blockIdx.x];
    float s = 0.f;
    for (int64_t e = threadIdx.x; e < m; e += TPB) s += x[base + (e / cg) * C + e % cg];
    const float mean = block_sum(s, red) / (float)m;
    float sh = 0.f, shx = 0.f;
    for (int64_t e = threadIdx.x; e < m; e += TPB) {
        const int c = (int)(e % cg);
        const int64_t i = base + (e / cg) * C + c;
        const float xh = (x[i] - mean) * rs;
        const float h = gy[i] * (w ? w[g * cg + c] : 1.0f);
        sh += h;
        shx = fmaf(h, xh, shx);
    }
    const float mh = block_sum(sh, red) / (float)m;
    const float mhx = block_sum(shx, red) / (float)m;
    for (int64_t e = threadIdx.x; e < m; e += TPB) {
        const int c = (int)(e % cg);
        const int64_t i = base + (e / cg) * C + c;
        const float xh = (x[i] - mean) * rs;
        const float h = gy[i] * (w ? w[g * cg + c] : 1.0f);
        gx[i] = rs * (h - mh - xh * mhx);
        if (xhat) xhat[i] = xh;
    }
}

// ---- training-mode backward of the token path (SURVEY.md section 8(f) N4 for the ViT family) ---------------------------------
// LayerNorm, nothing detached: x_hat = (x - mean) rstd, h = gy w:  gx = rstd (h - mean(h) - x_hat mean(h x_hat)); x_hat is also
// written out for the weight gradient (sum_rows gy x_hat, a column reduction).  One wavefront per row.
__global__ __launch_bounds__(TPB) void layernorm_bwd_full_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                 const float* __restrict__ w, const float* __restrict__ rstd,
                                                                 const float* __restrict__ addend, float* __restrict__ gx,
                                                                 float* __restrict__ xhat, int64_t rows, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * TPB + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * TPB) >> 6;
    for (int64_t r = wave; r < rows; r += nwaves) {
        const float* xs = x + r * D;
        const float* gs = gy + r * D;
        float s = 0.f;
        for (int c = lane; c < D; c += 64) s += xs[c];
        const float mean = wave_sum(s) / (float)D;
        const float rs = rstd[r];
        float sh = 0.f, shx = 0.f;
        for (int c = lane; c < D; c += 64) {
            const float xh = (xs[c] - mean) * rs;
            const float h = gs[c] * (w ? w[c] : 1.0f);
            sh += h;
            shx = fmaf(h, xh, shx);
        }
        const float mh = wave_sum(sh) / (float)D, mhx = wave_sum(shx) / (float)D;
        for (int c = lane; c < D; c += 64) {
            const float xh = (xs[c] - mean) * rs;
            const float h = gs[c] * (w ? w[c] : 1.0f);
            const float g = rs * (h - mh - xh * mhx);
            gx[r * D + c] = addend ? g + addend[r * D + c] : g;        // (the residual stream's gradient joins here)
            if (xhat) xhat[r * D + c] = xh;
        }
    }
}

// GELU, gate not detached: d/dx [x Phi(x)] = Phi(x) + x phi(x)
__global__ __launch_bounds__(TPB) void gelu_bwd_full_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                            float* __restrict__ gx, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * TPB;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) {
        const float v = x[i];
        const float Phi = bcos_gelu_gate(v);
        const float phi = 0.3989422804014327f * expf(-0.5f * v * v);
        gx[i] = gy[i] * (Phi + v * phi);
    }
}

// Softmax attention, nothing detached: per (image, head) one workgroup; K and V of the head stay in LDS, the query rows are walked
// in chunks of ARB; thread j owns key j (dK_j, dV_j accumulate in its registers), then threads (r, d) finish dQ of the chunk.
//   P = softmax(scale Q K^T) (from the stored statistics), dP = dO V^T, D_i = dO_i . O_i, dS = scale P (dP - D)
//   dQ = dS K,  dK = dS^T Q,  dV = P^T dO.   Plain FMA loops: training throughput of the ViTs is not a benchmarked quantity.
constexpr int ARB = 16;
constexpr int AKLD = DH + 1;      // padded K / V rows in LDS: lanes = keys read the same d without bank conflicts

__global__ __launch_bounds__(256) void attention_bwd_full_kernel(const float* __restrict__ qkv, const float* __restrict__ stats,
                                                                 const float* __restrict__ out, const float* __restrict__ gout,
                                                                 float* __restrict__ gqkv, int B, int T, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int inner = H * DH;
    const int tid = threadIdx.x;
    // (one key per thread: the host guarantees T <= 256)
    float* sK = sm;                         // [T][AKLD]
    float* sV = sK + T * AKLD;              // [T][AKLD]
    float* sQ = sV + T * AKLD;              // [ARB][DH]
    float* sdO = sQ + ARB * DH;             // [ARB][DH]
    float* sD = sdO + ARB * DH;             // [ARB]
    float* sSt = sD + ARB;                  // [ARB][2]
    float* sdS = sSt + 2 * ARB;             // [ARB][T]
    const float* base = qkv + (int64_t)b * T * 3 * inner + h * DH;
    for (int i = tid; i < T * DH; i += 256) {
        const int t = i / DH, d = i - t * DH;
        sK[t * AKLD + d] = base[(int64_t)t * 3 * inner + inner + d];
        sV[t * AKLD + d] = base[(int64_t)t * 3 * inner + 2 * inner + d];
    }
    float dk[DH], dv[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
    const bool key = tid < T;
    for (int i0 = 0; i0 < T; i0 += ARB) {
        __syncthreads();                                     // previous chunk consumed (first trip: K / V staged)
        for (int i = tid; i < ARB * DH; i += 256) {
            const int r = i / DH, d = i - r * DH;
            const int t = i0 + r;
            sQ[i] = t < T ? base[(int64_t)t * 3 * inner + d] : 0.f;
            sdO[i] = t < T ? gout[((int64_t)b * T + t) * inner + h * DH + d] : 0.f;
        }
        if (tid < ARB) {
            const int t = i0 + tid;
            float dd = 0.f;
            if (t < T) {
                const float* go = gout + ((int64_t)b * T + t) * inner + h * DH;
                const float* oo = out + ((int64_t)b * T + t) * inner + h * DH;
                for (int d = 0; d < DH; ++d) dd = fmaf(go[d], oo[d], dd);
                sSt[2 * tid] = stats[(((int64_t)b * H + h) * T + t) * 2];
                sSt[2 * tid + 1] = stats[(((int64_t)b * H + h) * T + t) * 2 + 1];
            } else {
                sSt[2 * tid] = INFINITY;                     // padded query rows: p = exp(-inf) * 0 = 0
                sSt[2 * tid + 1] = 0.f;
            }
            sD[tid] = dd;
        }
        __syncthreads();
        if (key) {
            const float* kj = sK + tid * AKLD;
            const float* vj = sV + tid * AKLD;
            for (int r = 0; r < ARB; ++r) {
                const float* q = sQ + r * DH;
                const float* go = sdO + r * DH;
                float sc = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < DH; ++d) { sc = fmaf(q[d], kj[d], sc); dp = fmaf(go[d], vj[d], dp); }
                const float pr = expf(sc * scale - sSt[2 * r]) * sSt[2 * r + 1];
                const float ds = pr * (dp - sD[r]) * scale;
                sdS[r * T + tid] = ds;
#pragma unroll
                for (int d = 0; d < DH; ++d) { dk[d] = fmaf(ds, q[d], dk[d]); dv[d] = fmaf(pr, go[d], dv[d]); }
            }
        }
        __syncthreads();
        for (int i = tid; i < ARB * DH; i += 256) {          // dQ of the chunk: thread (r, d)
            const int r = i / DH, d = i - r * DH;
            const int t = i0 + r;
            if (t >= T) continue;
            float acc = 0.f;
            for (int j = 0; j < T; ++j) acc = fmaf(sdS[r * T + j], sK[j * AKLD + d], acc);
            gqkv[((int64_t)b * T + t) * 3 * inner + h * DH + d] = acc;
        }
    }
    if (key) {
        float* gk = gqkv + ((int64_t)b * T + tid) * 3 * inner + inner + h * DH;
        float* gv = gk + inner;
#pragma unroll
        for (int d = 0; d < DH; ++d) { gk[d] = dk[d]; gv[d] = dv[d]; }
    }
}


// ---- round 5: the full attention gradient on the fp32 matrix pipe --------------------------------------------------------------------
// attention_bwd_full_kernel above evaluates the five T x T x 64 products of a head with scalar FMA chains (one key per thread):
// 1.09 ms per call at ViT-Ti batch 64, HALF of a training step (profiles/r05_kernel_stats_train_vit_ti.csv).  Here one workgroup per
// (image, head) keeps K and V in LDS ([T][65]: a lane reads "its" key's or query's element k without bank conflicts) and walks the
// query rows in tiles of 32; wave w owns the key tiles w and w + 4.  Per (row tile i, key tile j), all on v_mfma_f32_32x32x2_f32:
//   S = Q_i K_j^T, dP = dO_i V_j^T                     (A, B from LDS)
//   P = exp(S scale - m) / l,  dS = P (dP - D) scale   in the accumulator layout (lane = key, 16 query rows per lane)
//   dV_j += P^T dO_i,  dK_j += dS^T Q_i                A = the accumulator registers AS THEY ARE (an accumulator register of lane
//                                                      (key, half) is A[m = key][k = row 8 g + 4 half + r]: the contraction index is walked
//                                                      in that order), B from LDS; dK / dV stay in the owning wave's registers
//   dQ_i += dS K_j                                     the contraction runs over the lane index: dS goes through a wave-private LDS tile
// dQ_i of the four waves meets in LDS in wave order (fixed summation order).  T <= ATP - 1 = 207 tokens, head dim 64.
constexpr int AKP = DH + 1;       // LDS row pitch (floats)
constexpr int ATP = 208;          // K / V rows held (rows T .. ATP - 1 are zero: keys beyond T read a zero row)
constexpr int ASP = 33;           // pitch of a wave's dS tile
constexpr size_t attention_bwd_mfma_lds() {
    return ((size_t)2 * ATP * AKP + 2 * 32 * AKP + 4 * 32 * ASP + 32 * DH + 3 * 32) * sizeof(float);
}

__global__ __launch_bounds__(256) void attention_bwd_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ stats,
                                                                 const float* __restrict__ out, const float* __restrict__ gout,
                                                                 float* __restrict__ gqkv, int B, int T, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sK = sm;                          // [ATP][AKP]
    float* sV = sK + ATP * AKP;              // [ATP][AKP]
    float* sQ = sV + ATP * AKP;              // [32][AKP]
    float* sdO = sQ + 32 * AKP;              // [32][AKP]
    float* sdS = sdO + 32 * AKP;             // [4][32][ASP]   per-wave dS tile
    float* sRed = sdS + 4 * 32 * ASP;        // [32][DH]       dQ_i of the waves, summed in wave order
    float* sM = sRed + 32 * DH;              // [32] row maximum (natural-log units of the scaled scores)
    float* sI = sM + 32;                     // [32] 1 / row sum
    float* sD = sI + 32;                     // [32] sum_d dO O
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int inner = H * DH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, half = lane >> 5;
    const float* base = qkv + (int64_t)b * T * 3 * inner + h * DH;
    for (int i = tid; i < ATP * (DH / 4); i += 256) {
        const int t = i / (DH / 4), d4 = (i - t * (DH / 4)) * 4;
        f32x4 k4 = {0.f, 0.f, 0.f, 0.f}, v4 = {0.f, 0.f, 0.f, 0.f};
        if (t < T) {
            k4 = *reinterpret_cast<const f32x4*>(base + (int64_t)t * 3 * inner + inner + d4);
            v4 = *reinterpret_cast<const f32x4*>(base + (int64_t)t * 3 * inner + 2 * inner + d4);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { sK[t * AKP + d4 + q] = k4[q]; sV[t * AKP + d4 + q] = v4[q]; }
    }
    const int ntile = (T + 31) / 32;
    f32x16 dK[2][2], dV[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dK[a][t][r] = 0.f; dV[a][t][r] = 0.f; }
    float* my_dS = sdS + wave * 32 * ASP;
    for (int it = 0; it < ntile; ++it) {
        const int i0 = it * 32;
        __syncthreads();                                     // the previous row tile is consumed (first trip: K / V staged)
        {   // Q_i, dO_i: thread = (row tid / 8, eight consecutive d); D = sum_d dO O over the row's eight threads
            const int r = tid >> 3, d0 = (tid & 7) * 8;
            const int t = i0 + r;
            f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0, g0 = q0, g1 = q0, o0 = q0, o1 = q0;
            if (t < T) {
                const float* qp = base + (int64_t)t * 3 * inner + d0;
                const float* gp = gout + ((int64_t)b * T + t) * inner + h * DH + d0;
                const float* op = out + ((int64_t)b * T + t) * inner + h * DH + d0;
                q0 = *reinterpret_cast<const f32x4*>(qp); q1 = *reinterpret_cast<const f32x4*>(qp + 4);
                g0 = *reinterpret_cast<const f32x4*>(gp); g1 = *reinterpret_cast<const f32x4*>(gp + 4);
                o0 = *reinterpret_cast<const f32x4*>(op); o1 = *reinterpret_cast<const f32x4*>(op + 4);
            }
            float dd = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sQ[r * AKP + d0 + q] = q0[q]; sQ[r * AKP + d0 + 4 + q] = q1[q];
                sdO[r * AKP + d0 + q] = g0[q]; sdO[r * AKP + d0 + 4 + q] = g1[q];
                dd = fmaf(g0[q], o0[q], dd); dd = fmaf(g1[q], o1[q], dd);
            }
            dd += __shfl_xor(dd, 1); dd += __shfl_xor(dd, 2); dd += __shfl_xor(dd, 4);
            if ((tid & 7) == 0) {
                sD[r] = dd;
                if (t < T) {
                    sM[r] = stats[(((int64_t)b * H + h) * T + t) * 2];
                    sI[r] = stats[(((int64_t)b * H + h) * T + t) * 2 + 1];
                } else {
                    sM[r] = INFINITY;                        // padded query rows: p = exp(-inf) * 0 = 0
                    sI[r] = 0.f;
                }
            }
        }
        __syncthreads();
        f32x16 dQp[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) dQp[t][r] = 0.f;
#pragma unroll
        for (int js = 0; js < 2; ++js) {
            const int j = wave + 4 * js;
            if (j < ntile) {                                 // (wave-uniform)
                const int key = 32 * j + l32;
                const int krow = key < ATP - 1 ? key : ATP - 1;                  // keys beyond T: a zero row
                f32x16 S, dP;
#pragma unroll
                for (int r = 0; r < 16; ++r) { S[r] = 0.f; dP[r] = 0.f; }
#pragma unroll 8
                for (int st = 0; st < DH / 2; ++st) {
                    const int d = 2 * st + half;
                    S = __builtin_amdgcn_mfma_f32_32x32x2f32(sQ[l32 * AKP + d], sK[krow * AKP + d], S, 0, 0, 0);
                    dP = __builtin_amdgcn_mfma_f32_32x32x2f32(sdO[l32 * AKP + d], sV[krow * AKP + d], dP, 0, 0, 0);
                }
                const bool kvalid = key < T;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 8 * (r >> 2) + 4 * half + (r & 3);
                    const float pr = kvalid ? expf(S[r] * scale - sM[row]) * sI[row] : 0.f;
                    const float ds = pr * (dP[r] - sD[row]) * scale;
                    S[r] = pr;
                    dP[r] = ds;
                    my_dS[row * ASP + l32] = ds;
                }
                // dV_j += P^T dO_i, dK_j += dS^T Q_i: A = the accumulator registers, contraction index = their row order
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 8 * (r >> 2) + 4 * half + (r & 3);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        dV[js][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(S[r], sdO[row * AKP + l32 + 32 * t], dV[js][t], 0, 0, 0);
                        dK[js][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(dP[r], sQ[row * AKP + l32 + 32 * t], dK[js][t], 0, 0, 0);
                    }
                }
                // dQ_i += dS K_j: A[m = row][k = key] from the wave's dS tile (its own writes: ordered within the wave)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 4
                for (int st = 0; st < 16; ++st) {
                    const int kk = 2 * st + half;
                    const int kr = 32 * j + kk < ATP - 1 ? 32 * j + kk : ATP - 1;
                    const float a = my_dS[l32 * ASP + kk];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        dQp[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, sK[kr * AKP + l32 + 32 * t], dQp[t], 0, 0, 0);
                }
            }
        }
        // dQ_i: the waves' partial tiles meet in LDS in wave order (waves without a key tile contribute zeros)
        for (int w = 0; w < 4; ++w) {
            __syncthreads();
            if (wave == w) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = 8 * (r >> 2) + 4 * half + (r & 3);
                        float* dst = sRed + row * DH + l32 + 32 * t;
                        *dst = w == 0 ? dQp[t][r] : *dst + dQp[t][r];
                    }
            }
        }
        __syncthreads();
        {
            const int r = tid >> 3, d0 = (tid & 7) * 8;
            const int t = i0 + r;
            if (t < T) {
                float* dst = gqkv + ((int64_t)b * T + t) * 3 * inner + h * DH + d0;
                *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(sRed + r * DH + d0);
                *reinterpret_cast<f32x4*>(dst + 4) = *reinterpret_cast<const f32x4*>(sRed + r * DH + d0 + 4);
            }
        }
    }
    // dK_j, dV_j: accumulator tile (row = key, column = d)
#pragma unroll
    for (int js = 0; js < 2; ++js) {
        const int j = wave + 4 * js;
        if (j < ntile) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = 32 * j + 8 * (r >> 2) + 4 * half + (r & 3);
                    if (key < T) {
                        float* gk = gqkv + ((int64_t)b * T + key) * 3 * inner + inner + h * DH + l32 + 32 * t;
                        gk[0] = dK[js][t][r];
                        gk[inner] = dV[js][t][r];
                    }
                }
        }
    }
}

}  // namespace

#define STREAM(s) reinterpret_cast<hipStream_t>(s)

static bool ln_vec_ok(int D, std::initializer_list<const void*> ptrs) {
    uintptr_t bits = 0;
    for (const void* q : ptrs) bits |= reinterpret_cast<uintptr_t>(q);
    return D % 4 == 0 && D <= 256 && (bits & 15) == 0;
}

extern "C" int bcos_layernorm_fwd(const float* x, const float* weight, const float* bias, float* y, float* rstd_out,
                                  uint32_t* y_absmax, int64_t rows, int D, float eps, void* stream) {
    if (!x || !y || rows <= 0 || D <= 0) return bcos_set_error(BCOS_E_INVAL, "bcos_layernorm_fwd: bad argument");
    if (ln_vec_ok(D, {x, weight, bias, y}))
        hipLaunchKernelGGL(layernorm_fwd_kernel<true>, dim3(grid_rows(rows)), dim3(TPB), 0, STREAM(stream), x, weight, bias, y,
                           rstd_out, y_absmax, rows, D, eps);
    else
        hipLaunchKernelGGL(layernorm_fwd_kernel<false>, dim3(grid_rows(rows)), dim3(TPB), 0, STREAM(stream), x, weight, bias, y,
                           rstd_out, y_absmax, rows, D, eps);
    return check_launch("layernorm_fwd_kernel");
}

extern "C" int bcos_layernorm_stats(const float* x, const float* weight, const float* bias, float* rstd_out, float* zsumsq_out,
                                    uint32_t* x_absmax, int64_t rows, int D, float eps, void* stream) {
    if (!x || !rstd_out || rows <= 0 || D <= 0) return bcos_set_error(BCOS_E_INVAL, "bcos_layernorm_stats: bad argument");
    if (ln_vec_ok(D, {x, weight, bias})) {
        int64_t nb = (rows + 15) / 16;           // 16 rows per workgroup and pass
        if (nb > 256 * 32) nb = 256 * 32;
        const dim3 grid((unsigned)nb), block(TPB);
        hipStream_t st = STREAM(stream);
        switch ((D + 63) / 64) {
            case 1: hipLaunchKernelGGL(layernorm_stats16_kernel<1>, grid, block, 0, st, x, weight, bias, rstd_out, zsumsq_out, x_absmax, rows, D, eps); break;
            case 2: hipLaunchKernelGGL(layernorm_stats16_kernel<2>, grid, block, 0, st, x, weight, bias, rstd_out, zsumsq_out, x_absmax, rows, D, eps); break;
            case 3: hipLaunchKernelGGL(layernorm_stats16_kernel<3>, grid, block, 0, st, x, weight, bias, rstd_out, zsumsq_out, x_absmax, rows, D, eps); break;
            default: hipLaunchKernelGGL(layernorm_stats16_kernel<4>, grid, block, 0, st, x, weight, bias, rstd_out, zsumsq_out, x_absmax, rows, D, eps); break;
        }
    } else
        hipLaunchKernelGGL(layernorm_stats_kernel, dim3(grid_rows(rows)), dim3(TPB), 0, STREAM(stream), x, weight, bias, rstd_out,
                           zsumsq_out, x_absmax, rows, D, eps);
    return check_launch("layernorm_stats_kernel");
}

extern "C" int bcos_layernorm_bwd_detached(const float* gy, const float* weight, const float* rstd, const float* addend,
                                           const float* mul2, float* out, float* out2, uint32_t* out2_absmax, int64_t rows, int D,
                                           void* stream) {
    if (!gy || !rstd || (!out && !out2) || rows <= 0 || D <= 0 || (out2_absmax && !out2))
        return bcos_set_error(BCOS_E_INVAL, "bcos_layernorm_bwd_detached: bad argument");
    if (ln_vec_ok(D, {gy, weight, addend, mul2, out, out2}))
        hipLaunchKernelGGL(layernorm_bwd_detached_kernel<true>, dim3(grid_rows(rows)), dim3(TPB), 0, STREAM(stream), gy, weight,
                           rstd, addend, mul2, out, out2, out2_absmax, rows, D);
    else
        hipLaunchKernelGGL(layernorm_bwd_detached_kernel<false>, dim3(grid_rows(rows)), dim3(TPB), 0, STREAM(stream), gy, weight,
                           rstd, addend, mul2, out, out2, out2_absmax, rows, D);
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

static int attn_launch(bool bwd, const float* qkv, const float* z, float* out, float* stats_out, const float* stats_in,
                       unsigned* absmax_out, int B, int T, int H, float scale, void* stream) {
    const int Tpad = (T + 31) & ~31;
    // f16 matrix pipe (default) or the exact-fp32 MFMA form of rounds 1-2 (BCOS_OPT_ATTENTION_F32; also what the f32 contraction mode means)
    if (!bcos_option(BCOS_OPT_ATTENTION_F32)) {
        const size_t hb = 2 * (size_t)Tpad * AH_XROW + 2 * (size_t)DH * (Tpad * 2 + 16) + ((size_t)Tpad + 2 * DH + 2 * (size_t)Tpad) * 4;
        if (hb <= 160 * 1024 && Tpad <= 32 * AH_MAXIT) {
            const void* fn2 = bwd ? reinterpret_cast<const void*>(attention_h2_kernel<true>)
                                  : reinterpret_cast<const void*>(attention_h2_kernel<false>);
            static std::atomic<size_t> lds_hw2[2];
            hipError_t e2 = bcos_ensure_dynamic_lds(fn2, hb, lds_hw2[bwd ? 1 : 0]);
            if (e2 != hipSuccess) return bcos_set_hip_error("hipFuncSetAttribute(attention)", e2);
            if (bwd)
                hipLaunchKernelGGL(attention_h2_kernel<true>, dim3((unsigned)(B * H)), dim3(ATPB), hb, STREAM(stream), qkv, z, out,
                                   stats_out, stats_in, absmax_out, B, T, H, scale);
            else
                hipLaunchKernelGGL(attention_h2_kernel<false>, dim3((unsigned)(B * H)), dim3(ATPB), hb, STREAM(stream), qkv, z, out,
                                   stats_out, stats_in, absmax_out, B, T, H, scale);
            return check_launch(bwd ? "attention_h2_kernel<bwd>" : "attention_h2_kernel<fwd>");
        }
    }
    const size_t bytes = ((size_t)Tpad * AT_XLD + (size_t)DH * (Tpad + 4) + 2 * (size_t)Tpad) * sizeof(float);
    if (bytes > 160 * 1024) return bcos_set_error(BCOS_E_NOSUP, "attention: sequence too long for the LDS-resident kernel");
    const void* fn = bwd ? reinterpret_cast<const void*>(attention_mfma_kernel<true>)
                         : reinterpret_cast<const void*>(attention_mfma_kernel<false>);
    static std::atomic<size_t> lds_hw[2];
    hipError_t err = bcos_ensure_dynamic_lds(fn, bytes, lds_hw[bwd ? 1 : 0]);
    if (err != hipSuccess) return bcos_set_hip_error("hipFuncSetAttribute(attention)", err);
    if (bwd)
        hipLaunchKernelGGL(attention_mfma_kernel<true>, dim3((unsigned)(B * H)), dim3(ATPB), bytes, STREAM(stream), qkv, z, out,
                           stats_out, stats_in, absmax_out, B, T, H, scale);
    else
        hipLaunchKernelGGL(attention_mfma_kernel<false>, dim3((unsigned)(B * H)), dim3(ATPB), bytes, STREAM(stream), qkv, z, out,
                           stats_out, stats_in, absmax_out, B, T, H, scale);
    return check_launch(bwd ? "attention_mfma_kernel<bwd>" : "attention_mfma_kernel<fwd>");
}

extern "C" int bcos_attention_fwd(const float* qkv, float* out, float* stats, uint32_t* out_absmax, int B, int T, int H, int Dh,
                                  float scale, void* stream) {
    if (!qkv || !out || B <= 0 || T <= 0 || H <= 0) return bcos_set_error(BCOS_E_INVAL, "bcos_attention_fwd: bad argument");
    if (Dh != DH) return bcos_set_error(BCOS_E_NOSUP, "bcos_attention_fwd: head dimension must be 64");
    return attn_launch(false, qkv, nullptr, out, stats, nullptr, out_absmax, B, T, H, scale, stream);
}

extern "C" int bcos_attention_bwd_v(const float* qkv, const float* stats, const float* gout, float* gv, uint32_t* gv_absmax, int B,
                                    int T, int H, int Dh, float scale, void* stream) {
    if (!qkv || !stats || !gout || !gv || B <= 0 || T <= 0 || H <= 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_attention_bwd_v: bad argument");
    if (Dh != DH) return bcos_set_error(BCOS_E_NOSUP, "bcos_attention_bwd_v: head dimension must be 64");
    return attn_launch(true, qkv, gout, gv, nullptr, stats, gv_absmax, B, T, H, scale, stream);
}

extern "C" int bcos_finalize_explanation_patches(const float* gp, const float* x, const float* std6, float* weights_out,
                                                 float* contrib_out, int N, int Cx, int H, int W, int patch, int Cpad,
                                                 int add_inverse, void* stream) {
    if (!gp || !x || !std6 || (!weights_out && !contrib_out) || N <= 0 || H <= 0 || W <= 0 || patch <= 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_finalize_explanation_patches: bad argument");
    if (H % patch || W % patch || Cx != (add_inverse ? 3 : 6) || Cpad < 6)
        return bcos_set_error(BCOS_E_INVAL, "bcos_finalize_explanation_patches: bad geometry");
    // (one pixel per thread and pass: a grid of 16 workgroups per CU keeps enough independent loads in flight -- the 2 048 of
    //  grid_elems ran 25 dependent passes per thread at ViT-Ti batch 256)
    int64_t fb = ((int64_t)N * H * W + TPB - 1) / TPB;
    if (fb > 256 * 64) fb = 256 * 64;
    hipLaunchKernelGGL(finalize_patches_kernel, dim3((unsigned)fb), dim3(TPB), 0, STREAM(stream), gp, x,
                       std6, weights_out, contrib_out, N, Cx, H, W, patch, Cpad, add_inverse);
    return check_launch("finalize_patches_kernel");
}

extern "C" int bcos_groupnorm_fwd(const float* x, const float* weight, const float* bias, float* y, float* rstd_out, int N, int HW,
                                  int C, int G, float eps, void* stream) {
    if (!x || !y || N <= 0 || HW <= 0 || C <= 0 || G <= 0 || C % G != 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_groupnorm_fwd: bad argument (C must be divisible by the group count)");
    if ((int64_t)N * G >= ((int64_t)1 << 31)) return bcos_set_error(BCOS_E_NOSUP, "bcos_groupnorm_fwd: too many groups");
    hipLaunchKernelGGL(groupnorm_fwd_kernel, dim3((unsigned)(N * G)), dim3(TPB), 0, STREAM(stream), x, weight, bias, y, rstd_out, HW,
                       C, G, eps);
    return check_launch("groupnorm_fwd_kernel");
}

extern "C" int bcos_groupnorm_bwd_detached(const float* gy, const float* weight, const float* rstd, float* gx, int N, int HW, int C,
                                           int G, void* stream) {
    if (!gy || !rstd || !gx || N <= 0 || HW <= 0 || C <= 0 || G <= 0 || C % G != 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_groupnorm_bwd_detached: bad argument");
    if ((int64_t)N * G >= ((int64_t)1 << 31)) return bcos_set_error(BCOS_E_NOSUP, "bcos_groupnorm_bwd_detached: too many groups");
    hipLaunchKernelGGL(groupnorm_bwd_detached_kernel, dim3((unsigned)(N * G)), dim3(TPB), 0, STREAM(stream), gy, weight, rstd, gx, HW,
                       C, G);
    return check_launch("groupnorm_bwd_detached_kernel");
}

extern "C" int bcos_layernorm_bwd_add(const float* gy, const float* x, const float* weight, const float* rstd, const float* addend,
                                      float* gx, float* xhat_out, int64_t rows, int D, void* stream) {
    if (!gy || !x || !rstd || !gx || rows <= 0 || D <= 0) return bcos_set_error(BCOS_E_INVAL, "bcos_layernorm_bwd: bad argument");
    hipLaunchKernelGGL(layernorm_bwd_full_kernel, dim3(grid_rows(rows)), dim3(TPB), 0, STREAM(stream), gy, x, weight, rstd, addend, gx,
                       xhat_out, rows, D);
    return check_launch("layernorm_bwd_full_kernel");
}

extern "C" int bcos_layernorm_bwd(const float* gy, const float* x, const float* weight, const float* rstd, float* gx, float* xhat_out,
                                  int64_t rows, int D, void* stream) {
    return bcos_layernorm_bwd_add(gy, x, weight, rstd, nullptr, gx, xhat_out, rows, D, stream);
}

extern "C" int bcos_gelu_bwd(const float* gy, const float* x, float* gx, int64_t n, void* stream) {
    if (!gy || !x || !gx || n <= 0) return bcos_set_error(BCOS_E_INVAL, "bcos_gelu_bwd: bad argument");
    int64_t blocks = (n + TPB - 1) / TPB;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gelu_bwd_full_kernel, dim3((unsigned)blocks), dim3(TPB), 0, STREAM(stream), gy, x, gx, n);
    return check_launch("gelu_bwd_full_kernel");
}

extern "C" int bcos_attention_bwd(const float* qkv, const float* stats, const float* out, const float* gout, float* gqkv, int B, int T,
                                  int H, int Dh, float scale, void* stream) {
    if (!qkv || !stats || !out || !gout || !gqkv || B <= 0 || T <= 0 || H <= 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_attention_bwd: bad argument");
    if (Dh != DH) return bcos_set_error(BCOS_E_NOSUP, "bcos_attention_bwd: head dim must be 64");
    if (T > 256) return bcos_set_error(BCOS_E_NOSUP, "bcos_attention_bwd: at most 256 tokens");
    if (T <= ATP - 1 && !((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(gout) |
                           reinterpret_cast<uintptr_t>(gqkv)) & 15)) {
        // the five products of a head on the fp32 matrix pipe (attention_bwd_mfma_kernel); longer sequences keep the scalar kernel
        const size_t lds = attention_bwd_mfma_lds();
        static std::atomic<size_t> lds_hw2;
        hipError_t e2 = bcos_ensure_dynamic_lds(reinterpret_cast<const void*>(attention_bwd_mfma_kernel), lds, lds_hw2);
        if (e2 != hipSuccess) return bcos_set_hip_error("hipFuncSetAttribute", e2);
        hipLaunchKernelGGL(attention_bwd_mfma_kernel, dim3((unsigned)(B * H)), dim3(256), lds, STREAM(stream), qkv, stats, out, gout, gqkv, B, T,
                           H, scale);
        return check_launch("attention_bwd_mfma_kernel");
    }
    const size_t bytes = ((size_t)2 * T * AKLD + 2 * ARB * DH + 3 * ARB + (size_t)ARB * T) * sizeof(float);
    static std::atomic<size_t> lds_hw;
    hipError_t err = bcos_ensure_dynamic_lds(reinterpret_cast<const void*>(attention_bwd_full_kernel), bytes, lds_hw);
    if (err != hipSuccess) return bcos_set_hip_error("hipFuncSetAttribute", err);
    hipLaunchKernelGGL(attention_bwd_full_kernel, dim3((unsigned)(B * H)), dim3(256), bytes, STREAM(stream), qkv, stats, out, gout, gqkv,
                       B, T, H, scale);
    return check_launch("attention_bwd_full_kernel");
}

extern "C" int bcos_groupnorm_bwd(const float* gy, const float* x, const float* weight, const float* rstd, float* gx, float* xhat_out,
                                  int N, int HW, int C, int G, void* stream) {
    if (!gy || !x || !rstd || !gx || N <= 0 || HW <= 0 || C <= 0 || G <= 0 || C % G != 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_groupnorm_bwd: bad argument");
    if ((int64_t)N * G >= ((int64_t)1 << 31)) return bcos_set_error(BCOS_E_NOSUP, "bcos_groupnorm_bwd: too many groups");
    hipLaunchKernelGGL(groupnorm_bwd_full_kernel, dim3((unsigned)(N * G)), dim3(TPB), 0, STREAM(stream), gy, x, weight, rstd, gx, xhat_out,
                       HW, C, G);
    return check_launch("groupnorm_bwd_full_kernel");
}
