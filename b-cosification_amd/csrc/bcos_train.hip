// bcos_train.hip -- training-mode backward of the B-cos layers on gfx950.
//
// Outside explanation mode the dynamic scale s = |lin| / norm is NOT detached (bcos/modules/bcosconv2d.py:176-194,
// bcosifyconv2d.py:85-101), so for y = s(lin, norm) * lin with lin = conv(x, W) (+ bias), norm = ||patch(x)||:
//     dL/dlin  = gy * dy/dlin                      B == 2:  dy/dlin  = 2 s
//     dL/dnorm = sum_c gy_c * dy_c/dnorm           B == 2:  dy/dnorm = -y / norm
//     gx = dgrad(dL/dlin, W)  +  x (.) PatchSum^T(dL/dnorm / norm)        (d norm / d x_j = x_j / norm)
//     gW = wgrad(dL/dlin, x),  gbias = sum_pixels dL/dlin
// The input gradient reuses the tapconv kernel (bcos_tapconv.hip) with the norm term as its epilogue addend; this
// file holds what is new: the per-pixel scale derivative, the transposed patch sum, the weight-gradient contraction
// on v_mfma_f32_32x32x2_f32 (its operand layout -- 32 consecutive channels of one pixel per half-wavefront -- is exactly
// how NHWC tensors lie in memory, so the pixel-contraction needs no transposes), and per-channel reductions for bias
// gradients and the batch statistics of BatchNormUncentered2d (batchnorm_uncentered.py:36-44).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "bcos_hip.h"
#include "bcos_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

inline int check_launch(const char* what) {
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return bcos_set_hip_error(what, err);
    return BCOS_OK;
}

// ---- dy/dlin and dL/dnorm of one output pixel: one wavefront per row (grid-stride) ---------------------------------------
// General form: s = c^(B-1), c = q + 1e-6, q = |lin| / norm.  q is rebuilt from lin = y / s (s > 0 in this form) rather than
// from s^(1/(B-1)): a learnable B starts at 1 + 1e-6 (bcos/training/trainer.py:463), where s is 1 to fp32 precision and
// carries no information about q.  `bgrad` (optional): dL/dB_eff = sum gy * y * ln c, one atomic per workgroup.
// BN (round 5): `gy` is the gradient w.r.t. the OUTPUT of the BatchNormUncentered2d behind the layer and the kernel forms the gradient
// w.r.t. y itself: gy_y = gy * bn_g[c] + (y - bn_mean[c]) * bn_coef[c] (bn_coef NULL: variance a constant) -- bcos_channel_axpby's
// pass over the tensor (one write and one read of a layer output per layer and training step) folded into this one.
template <bool BN>
__global__ __launch_bounds__(256) void scale_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                        const float* __restrict__ s, const float* __restrict__ norm,
                                                        const float* __restrict__ bn_g, const float* __restrict__ bn_mean,
                                                        const float* __restrict__ bn_coef,
                                                        float* __restrict__ glin, float* __restrict__ rnorm, float* __restrict__ bgrad,
                                                        unsigned* __restrict__ glin_absmax,
                                                        int64_t rows, int C, int linear_eps, float b, int pow_form) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const float bm1 = b - 1.0f;
    float bacc = 0.f;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const float nrm = norm[row];
        float acc = 0.f;
        unsigned mx = 0u;
        for (int c = lane * 4; c < C; c += 256) {
            const int64_t i = row * C + c;
            f32x4 g4 = *reinterpret_cast<const f32x4*>(gy + i);
            const f32x4 y4 = *reinterpret_cast<const f32x4*>(y + i);
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(s + i);
            if constexpr (BN) {
                g4 *= *reinterpret_cast<const f32x4*>(bn_g + c);
                if (bn_coef) {
                    f32x4 w = y4;
                    if (bn_mean) w -= *reinterpret_cast<const f32x4*>(bn_mean + c);
                    g4 += w * *reinterpret_cast<const f32x4*>(bn_coef + c);
                }
            }
            f32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (!pow_form) {                       // s = |lin| / norm:  dy/dlin = 2 s,  dy/dnorm = -y / norm
                    o[q] = g4[q] * 2.0f * s4[q];
                    acc = fmaf(g4[q], -y4[q] / nrm, acc);
                } else {
                    const float qq = fabsf(y4[q] / s4[q]) / nrm;       // |lin| / norm
                    const float cc = qq + 1e-6f;
                    const float ratio = qq / cc;
                    o[q] = g4[q] * s4[q] * (1.0f + bm1 * ratio);
                    acc = fmaf(g4[q], -bm1 * y4[q] * ratio / nrm, acc);
                    if (bgrad) bacc = fmaf(g4[q] * y4[q], logf(cc), bacc);
                }
            }
            *reinterpret_cast<f32x4*>(glin + i) = o;
#pragma unroll
            for (int q = 0; q < 4; ++q) mx = max(mx, __float_as_uint(o[q]) & 0x7fffffffu);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (glin_absmax) {      // per-row max |glin| (fp32 bit pattern): the operand scale of the input-gradient launch that reads glin
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
            if (lane == 0) glin_absmax[row] = mx;
        }
        if (lane == 0) {
            // d norm / d x_j = x_j / ||.||: sqrt(S + 1e-6) differentiates to x / norm, ||x|| + 1e-12 to x / (norm - 1e-12)
            const float div = linear_eps ? fmaxf(nrm - 1e-12f, 1e-30f) : nrm;
            rnorm[row] = acc / div;
        }
    }
    if (bgrad) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bacc += __shfl_xor(bacc, o);
        if (lane == 0) red[wave] = bacc;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(bgrad, (red[0] + red[1]) + (red[2] + red[3]));
    }
}

// ---- backward of the unit-norm projection w_eff[r,:] = gain[r] * w[r,:] / ||w[r,:]||: one wavefront per row ------------------
//      gw[r,:] = gain[r] / ||w[r]|| * (g[r,:] - w_hat[r,:] <w_hat[r], g[r]>),   ggain[r] = <w_hat[r], g[r]>
__global__ __launch_bounds__(256) void rownorm_bwd_kernel(const float* __restrict__ w, const float* __restrict__ g,
                                                          const float* __restrict__ gain, float* __restrict__ gw,
                                                          float* __restrict__ ggain, int rows, int64_t cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* wr = w + (int64_t)row * cols;
    const float* gr = g + (int64_t)row * cols;
    float ss = 0.f, dot = 0.f;
    for (int64_t c = lane; c < cols; c += 64) {
        ss = fmaf(wr[c], wr[c], ss);
        dot = fmaf(wr[c], gr[c], dot);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ss += __shfl_xor(ss, o);
        dot += __shfl_xor(dot, o);
    }
    const float nrm = sqrtf(ss);
    const float inv = 1.0f / nrm;
    const float gn = gain ? gain[row] : 1.0f;
    const float dh = dot * inv;                     // <w_hat, g>
    if (gw) {
        float* o = gw + (int64_t)row * cols;
        for (int64_t c = lane; c < cols; c += 64) o[c] = gn * inv * (gr[c] - wr[c] * inv * dh);
    }
    if (ggain && lane == 0) ggain[row] = dh;
}

// ---- MaxOut routing by index: full[r, c * M + argmax[r, c]] = g[r, c], zero elsewhere ---------------------------------------
__global__ __launch_bounds__(256) void maxout_scatter_kernel(const float* __restrict__ g, const int* __restrict__ argmax,
                                                             float* __restrict__ full, int64_t n, int M) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float v = g[i];
        const int a = argmax[i];
        for (int m = 0; m < M; ++m) full[i * M + m] = m == a ? v : 0.f;
    }
}

// ---- transposed patch sum times x: out[n,h,w,:] = x[n,h,w,:] * sum_{(i,j): patch(i,j) contains (h,w)} r[n,i,j] -----------
__global__ __launch_bounds__(256) void patch_norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ add,
                                                             float* __restrict__ out, int N, int H, int W, int C, int x_pitch,
                                                             int P, int Q, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
                                                             int dw, int vec) {
    const int lane = threadIdx.x & 63;
    const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t total = (int64_t)N * H * W;
    if (pix >= total) return;
    const int w = (int)(pix % W);
    const int h = (int)((pix / W) % H);
    const int n = (int)(pix / ((int64_t)W * H));
    float t = 0.f;
    for (int tap = lane; tap < kh * kw; tap += 64) {
        const int th = tap / kw, tw = tap - th * kw;
        const int hn = h + ph - th * dh, wn = w + pw - tw * dw;
        if (hn >= 0 && wn >= 0 && hn % sh == 0 && wn % sw == 0) {
            const int i = hn / sh, j = wn / sw;
            if (i < P && j < Q) t += r[((int64_t)n * P + i) * Q + j];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    const float* src = x + pix * x_pitch;
    float* dst = out + pix * C;
    const float* ad = add ? add + pix * C : nullptr;          // (bcos_patch_norm_bwd_add: a gradient that reaches the same tensor by another path)
    if (vec) {
        for (int c = lane * 4; c < C; c += 256) {
            f32x4 v = *reinterpret_cast<const f32x4*>(src + c) * t;
            if (ad) v += *reinterpret_cast<const f32x4*>(ad + c);
            *reinterpret_cast<f32x4*>(dst + c) = v;
        }
    } else {                                       // C % 4 != 0 (e.g. the 6-channel network input): element by element
        for (int c = lane; c < C; c += 64) dst[c] = src[c] * t + (ad ? ad[c] : 0.f);
    }
}

// The same for C = 64 / 128 channels (the 3 x 3 layers at 56^2 / 28^2, where one wave per pixel left 48 / 32 of its lanes without a
// float4): LPP = C / 4 lanes per pixel, 64 / LPP pixels per wave; the taps go to the first lanes of the pixel's group and meet by a
// butterfly over the group -- for kh kw <= LPP the very additions of the kernel above (its upper butterfly steps add zeros), so the bits
// do not change.
template <int LPP>
__global__ __launch_bounds__(256) void patch_norm_bwd_group_kernel(const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ add,
                                                                   float* __restrict__ out, int N, int H, int W, int C, int x_pitch,
                                                                   int P, int Q, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
                                                                   int dw) {
    constexpr int PPW = 64 / LPP;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPP;
    const int64_t total = (int64_t)N * H * W;
    const int64_t pix = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * PPW + lane / LPP;
    const bool live = pix < total;                 // (whole groups: the shuffles below stay inside a group)
    float t = 0.f;
    if (live) {
        const int w = (int)(pix % W);
        const int h = (int)((pix / W) % H);
        const int n = (int)(pix / ((int64_t)W * H));
        if (sub < kh * kw) {
            const int th = sub / kw, tw = sub - th * kw;
            const int hn = h + ph - th * dh, wn = w + pw - tw * dw;
            if (hn >= 0 && wn >= 0 && hn % sh == 0 && wn % sw == 0) {
                const int i = hn / sh, j = wn / sw;
                if (i < P && j < Q) t = r[((int64_t)n * P + i) * Q + j];
            }
        }
    }
#pragma unroll
    for (int o = LPP / 2; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if (!live) return;
    f32x4 v = *reinterpret_cast<const f32x4*>(x + pix * x_pitch + sub * 4) * t;
    if (add) v += *reinterpret_cast<const f32x4*>(add + pix * C + sub * 4);
    *reinterpret_cast<f32x4*>(out + pix * C + sub * 4) = v;
}

// ---- weight gradient -------------------------------------------------------------------------------------------------------
// gw[co][th][tw][ci] += sum_m glin[m, co] * x[pix(m, th, tw), ci]: per (128 co x 128 ci tile, tap, pixel chunk) one workgroup;
// operands staged pixel-major through LDS ([32 pixels][128 channels], the memory order), fragments of v_mfma_f32_32x32x2_f32
// are 32 consecutive channels of two pixels = two conflict-free ds_read_b32 rows; partial tiles of the pixel chunks are
// combined with fp32 atomics (the caller zeroes gw).
struct WgradArgs {
    const float* glin;
    const float* x;
    float* gw;
    int N, H, W, C, x_pitch;       // x: [N,H,W,x_pitch], C channels used
    int P, Q, Cout, g_pitch;       // glin: [N,P,Q,g_pitch]
    int kh, kw, sh, sw, ph, pw, dh, dw;
    int gw_cin;                    // channels per tap in gw (its innermost pitch)
    int tiles_ci;
    int64_t M;                     // N*P*Q
    int64_t chunk;                 // pixels per workgroup (multiple of 32)
    int fuse;                      // 1: narrow inputs (C < 128, several taps): the tile's "ci" columns run over (tap, channel) pairs --
                                   // column c = tap c / C, channel c % C -- instead of one tap per workgroup with most of the tile empty
                                   // (the 7 x 7 stem over 8 channels: 4 column tiles instead of 49 nearly empty ones)
    int cf;                        // ... channel stride of a tap among the fused columns: C rounded up to 4 (<= x_pitch; columns >= C are skipped)
    int ctot;                      // kh * kw * cf
};

constexpr int WG_T = 128;          // tile edge (channels)
constexpr int WG_K = 32;           // pixels per stage
constexpr int WG_LD = WG_T + 4;    // LDS row pitch (floats): rows of different pixels start in different banks

typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned wg_u32x4 __attribute__((ext_vector_type(4)));

// X3 = true (round 5): the products run on the bf16 matrix pipe over EXACT 3-way splits of both operands, x = h + m + l (8 + 8 + 8
// significand bits, fp32's exponent range: no scaling), a b = a_h b_h + (a_h b_m + a_m b_h) + (a_m b_m + a_h b_l + a_l b_h): six
// v_mfma_f32_32x32x16_bf16 per 16 pixels (192 matrix-pipe cycles) where eight v_mfma_f32_32x32x2_f32 took 512; the dropped terms are
// <= 2^-21 |a b| -- the arithmetic of the forward / input-gradient contractions in mode bf16x3 (csrc/bcos_tapconv.hip: tile_body_x3).
// The fp32 operands stay in LDS as they are (pixel-major); a lane gathers its 8 pixels of a channel and splits them in registers.
// X3 = false: exact fp32 MFMA (contraction mode f32).
template <bool X3>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs p) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float* sA = wsm;                               // [2][WG_K][WG_LD] glin
    float* sB = wsm + 2 * WG_K * WG_LD;            // [2][WG_K][WG_LD] x
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int tile = blockIdx.x;
    const int tile_co = tile / p.tiles_ci, tile_ci = tile - tile_co * p.tiles_ci;
    const int co0 = tile_co * WG_T, ci0 = tile_ci * WG_T;
    // this thread's 16-byte column chunk of the x operand: (tap, first channel); fused launches derive the tap from the column
    const int cq_ = threadIdx.x & 31;
    const int bcol = ci0 + cq_ * 4;
    const int tap = p.fuse ? (bcol < p.ctot ? bcol / p.cf : 0) : (int)blockIdx.y;
    const int bci = p.fuse ? bcol - tap * p.cf : bcol;
    const bool bcol_ok = p.fuse ? bcol < p.ctot : true;
    const int th = tap / p.kw, tw = tap - th * p.kw;
    const int64_t m_lo = (int64_t)blockIdx.z * p.chunk;
    const int64_t m_hi = m_lo + p.chunk < p.M ? m_lo + p.chunk : p.M;
    const int PQ = p.P * p.Q;

    // staging: thread -> (pixel row lr + 8 j, 16-byte channel chunk cq) of both operands
    const int cq = tid & 31, lr = tid >> 5;
    f32x4 ra[4], rb[4];
    auto load = [&](int64_t m0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t m = m0 + lr + 8 * j;
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
            if (m < m_hi) {
                const int n = (int)(m / PQ);
                const int rem = (int)(m - (int64_t)n * PQ);
                const int i = rem / p.Q, jj = rem - i * p.Q;
                const int co = co0 + cq * 4;
                if (co + 3 < p.Cout) a = *reinterpret_cast<const f32x4*>(p.glin + m * p.g_pitch + co);
                else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (co + q < p.Cout) a[q] = p.glin[m * p.g_pitch + co + q];
                }
                const int ih = i * p.sh - p.ph + th * p.dh, iw = jj * p.sw - p.pw + tw * p.dw;
                if (bcol_ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) {
                    const float* src = p.x + (((int64_t)n * p.H + ih) * p.W + iw) * p.x_pitch;
                    const int ci = bci;
                    if (p.fuse || ci + 3 < p.C) b = *reinterpret_cast<const f32x4*>(src + ci);      // (fused: the chunk lies inside [0, cf) <= x_pitch)
                    else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (ci + q < p.C) b[q] = src[ci + q];
                    }
                }
            }
            ra[j] = a;
            rb[j] = b;
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<f32x4*>(sA + (buf * WG_K + lr + 8 * j) * WG_LD + cq * 4) = ra[j];
            *reinterpret_cast<f32x4*>(sB + (buf * WG_K + lr + 8 * j) * WG_LD + cq * 4) = rb[j];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fr = lane & 31, fk = lane >> 5;
    const int nst = (int)((m_hi - m_lo + WG_K - 1) / WG_K);
    if (nst > 0) {
        load(m_lo);
        store(0);
        __syncthreads();
        for (int st = 0; st < nst; ++st) {
            const int cur = st & 1;
            if (st + 1 < nst) load(m_lo + (int64_t)(st + 1) * WG_K);
            const float* a = sA + cur * WG_K * WG_LD + wave_m * 64 + fr;
            const float* b = sB + cur * WG_K * WG_LD + wave_n * 64 + fr;
            if constexpr (X3) {
                // exact split of 8 fp32 values (pixels kk + 8 fk .. + 7 of one channel) into three bf16x8 fragments: truncation keeps
                // every remainder exact (x - h has <= 16 significant bits, x - h - m <= 8: a bf16 holds it as is)
                auto frag3 = [&](const float* src, wg_bf16x8 (&out)[3]) {
                    unsigned hh[8], mm[8], ll[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float x = src[q * WG_LD];
                        const unsigned hu = __float_as_uint(x) & 0xffff0000u;
                        const float r1 = x - __uint_as_float(hu);
                        const unsigned mu = __float_as_uint(r1) & 0xffff0000u;
                        const float r2 = r1 - __uint_as_float(mu);
                        hh[q] = hu; mm[q] = mu; ll[q] = __float_as_uint(r2);
                    }
                    wg_u32x4 ph, pm, pl;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        ph[q] = hh[2 * q + 1] | (hh[2 * q] >> 16);
                        pm[q] = mm[2 * q + 1] | (mm[2 * q] >> 16);
                        pl[q] = (ll[2 * q + 1] & 0xffff0000u) | (ll[2 * q] >> 16);
                    }
                    out[0] = __builtin_bit_cast(wg_bf16x8, ph);
                    out[1] = __builtin_bit_cast(wg_bf16x8, pm);
                    out[2] = __builtin_bit_cast(wg_bf16x8, pl);
                };
#pragma unroll
                for (int kk = 0; kk < WG_K; kk += 16) {
                    wg_bf16x8 af[2][3], bf[2][3];
#pragma unroll
                    for (int i = 0; i < 2; ++i) frag3(a + (kk + 8 * fk) * WG_LD + i * 32, af[i]);
#pragma unroll
                    for (int j = 0; j < 2; ++j) frag3(b + (kk + 8 * fk) * WG_LD + j * 32, bf[j]);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {       // smallest terms first
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
                        }
                }
            } else {
#pragma unroll
            for (int kk = 0; kk < WG_K; kk += 2) {
                float af[2], bf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i] = a[(kk + fk) * WG_LD + i * 32];
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[j] = b[(kk + fk) * WG_LD + j * 32];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
            }
            if (st + 1 < nst) store(cur ^ 1);
            __syncthreads();
        }
    }
    // accumulator (row = co, column = ci): lane = column, 16 rows per lane
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = ci0 + wave_n * 64 + j * 32 + (lane & 31);
            const int otap = p.fuse ? (col < p.ctot ? col / p.cf : 0) : tap;      // (fused: this lane's column names its own tap)
            const int ci = p.fuse ? col - otap * p.cf : col;
            const bool col_ok = p.fuse ? (col < p.ctot && ci < p.C) : ci < p.C;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wave_m * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co < p.Cout && col_ok)
                    atomicAdd(p.gw + ((int64_t)co * p.kh * p.kw + otap) * p.gw_cin + ci, acc[i][j][r]);
            }
        }
}

// ---- weight gradient, round 6: split once at staging, fragments by ds_read_b128, fixed-order combine --------------------------------
// The kernel above hands every wave 8 scalar LDS reads and ~60 vector instructions per fragment (each value of a stage is gathered and split
// by the two waves that share its half tile): 17.8 M vector for 1.2 M matrix instructions, 16 % of the bf16 pipe (profiles/r05_small_probes.txt
// (9)).  Here a 16-pixel stage of both operands is split into its three bf16 planes ONCE, by the thread that loaded it -- a thread holds a
// 4 pixel x 4 channel block, i.e. for each of its channels four consecutive k of the contraction -- and written to LDS pixel-contiguous:
//   plane image [k-half 0 / 1][128 channel rows][8 pixels] bf16 = 4 KB, rows permuted (see `store` below) so that the 8-byte stores of a
//   wave spread over the banks; a fragment of v_mfma_f32_32x32x16_bf16 (row = channel, 8 consecutive pixels) is then ONE conflict-free
//   ds_read_b128 per plane.  Waves 4 x 1: a wave owns 32 output rows and all 128 columns (a row tile beyond Cout is skipped whole), and
//   its lanes hold four consecutive input channels per output row -- the partial tile leaves as 16-byte stores.
// Per wave and stage: 15 fragment reads, 24 matrix instructions, ~90 vector instructions of split + 12 ds_write_b64 for its share of the
// next stage; two LDS buffers of 24 KB, one barrier per stage; <= 168 registers: three workgroups per CU.  Same arithmetic as above (exact
// 3-way bf16 splits by truncation, the six leading products, smallest terms first).
// Combine: every (tile, tap, pixel chunk) workgroup STORES its partial tile into slab `chunk` of a workspace laid out like gw; a second
// launch adds the slabs of every element in chunk order -- one fixed order whatever the dispatch: the weight gradient is reproducible bit
// for bit from run to run (the atomics of the kernel above are not).  ws == NULL: atomics into a zeroed gw as before.
constexpr int W3_K = 16;           // pixels per stage
constexpr int W3_PLANE = 2 * WG_T * 16;      // bytes of one plane image of one operand: [2][128][16 B]
constexpr int W3_BUF = 6 * W3_PLANE;         // one stage: 2 operands x 3 planes

__global__ __launch_bounds__(256, 3) void wgrad3_kernel(const WgradArgs p, float* __restrict__ ws, const int64_t slab) {
    extern __shared__ __attribute__((aligned(16))) char w3sm[];       // [2 buffers][operand g | x][plane h | m | l][k-half][128 rows][16 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x;
    const int tile_co = tile / p.tiles_ci, tile_ci = tile - tile_co * p.tiles_ci;
    const int co0 = tile_co * WG_T, ci0 = tile_ci * WG_T;
    // staging: waves 0, 1 load the gradient rows, waves 2, 3 the input rows; a thread = (16-byte channel chunk cq, pixel group pg of 4)
    const bool is_x = wave >= 2;
    const int cq = tid & 31, pg = (tid >> 5) & 3;
    const int bcol = ci0 + cq * 4;
    const int tap = p.fuse ? (bcol < p.ctot ? bcol / p.cf : 0) : (int)blockIdx.y;
    const int bci = p.fuse ? bcol - tap * p.cf : bcol;
    const bool bcol_ok = p.fuse ? bcol < p.ctot : true;
    const int th = tap / p.kw, tw = tap - th * p.kw;
    const int64_t m_lo = (int64_t)blockIdx.z * p.chunk;
    const int64_t m_hi = m_lo + p.chunk < p.M ? m_lo + p.chunk : p.M;
    const int PQ = p.P * p.Q;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    f32x4 rv[4];                    // the thread's 4 pixels x 4 channels of its operand, one stage ahead
    // (n, i, j) of the thread's four output pixels: divided out once, then ADVANCED by the 16 pixels of a stage (the two waves that stage
    // the input rows would otherwise spend eight integer divisions per thread and stage); a 1 x 1 / stride-1 / unpadded layer over the
    // whole image (P Q = H W) needs none of it: its input pixel IS the output pixel
    const bool direct = p.kh * p.kw == 1 && p.sh == 1 && p.sw == 1 && p.ph == 0 && p.pw == 0 && p.P == p.H && p.Q == p.W;
    int pn[4], pi[4], pj[4];
    if (is_x && !direct) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t m = m_lo + pg * 4 + j;
            pn[j] = (int)(m / PQ);
            const int rem = (int)(m - (int64_t)pn[j] * PQ);
            pi[j] = rem / p.Q;
            pj[j] = rem - pi[j] * p.Q;
        }
    }
    auto load = [&](int64_t m0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t m = m0 + pg * 4 + j;
            f32x4 v = zero4;
            if (!is_x) {
                if (m < m_hi) {
                    const int co = co0 + cq * 4;
                    if (co + 3 < p.Cout) v = *reinterpret_cast<const f32x4*>(p.glin + m * p.g_pitch + co);
                    else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (co + q < p.Cout) v[q] = p.glin[m * p.g_pitch + co + q];
                    }
                }
            } else {
                const float* src = nullptr;
                if (direct) {
                    if (m < m_hi && bcol_ok) src = p.x + m * p.x_pitch;
                } else {
                    const int ih = pi[j] * p.sh - p.ph + th * p.dh, iw = pj[j] * p.sw - p.pw + tw * p.dw;
                    if (m < m_hi && bcol_ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
                        src = p.x + (((int64_t)pn[j] * p.H + ih) * p.W + iw) * p.x_pitch;
                    pj[j] += W3_K;                         // the same thread's pixel of the next stage
                    while (pj[j] >= p.Q) { pj[j] -= p.Q; if (++pi[j] == p.P) { pi[j] = 0; ++pn[j]; } }
                }
                if (src) {
                    if (p.fuse || bci + 3 < p.C) v = *reinterpret_cast<const f32x4*>(src + bci);
                    else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (bci + q < p.C) v[q] = src[bci + q];
                    }
                }
            }
            rv[j] = v;
        }
    };
    // split + store: 8 bytes (4 pixels of one channel) into k-half pg >> 1 of the channel's plane row.  Row of tile channel 4 cq + c:
    //   gradient rows (output rows of the tile):  32 (cq / 8) + 8 c + cq % 8  -- 32-channel blocks stay together (a row tile beyond Cout is
    //                                             skipped whole), the 8-byte stores of 16 lanes fall on 8 distinct 16-byte slots;
    //   input rows (columns of the tile):         32 c + cq                   -- the four channels of a chunk sit at the SAME column of the
    //                                             four column tiles: a lane of the epilogue holds 4 consecutive ci (one 16-byte store).
    const int row0 = is_x ? cq : 32 * (cq >> 3) + (cq & 7);
    const int rstep = is_x ? 32 : 8;
    const int st_off = (is_x ? 3 * W3_PLANE : 0) + (pg >> 1) * (WG_T * 16) + row0 * 16 + (pg & 1) * 8;
    auto store = [&](int buf) {
        char* base = w3sm + buf * W3_BUF + st_off;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            unsigned hh[4], mm[4], ll[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = rv[j][c];
                const unsigned hu = __float_as_uint(x) & 0xffff0000u;
                const float r1 = x - __uint_as_float(hu);
                const unsigned mu = __float_as_uint(r1) & 0xffff0000u;
                const float r2 = r1 - __uint_as_float(mu);
                hh[j] = hu; mm[j] = mu; ll[j] = __float_as_uint(r2);
            }
            // two bf16 per word: the high halves of (odd pixel, even pixel)
            uint2 ph, pm, pl;
            ph.x = __builtin_amdgcn_perm(hh[1], hh[0], 0x07060302u); ph.y = __builtin_amdgcn_perm(hh[3], hh[2], 0x07060302u);
            pm.x = __builtin_amdgcn_perm(mm[1], mm[0], 0x07060302u); pm.y = __builtin_amdgcn_perm(mm[3], mm[2], 0x07060302u);
            pl.x = __builtin_amdgcn_perm(ll[1], ll[0], 0x07060302u); pl.y = __builtin_amdgcn_perm(ll[3], ll[2], 0x07060302u);
            char* row = base + c * (rstep * 16);
            *reinterpret_cast<uint2*>(row) = ph;
            *reinterpret_cast<uint2*>(row + W3_PLANE) = pm;
            *reinterpret_cast<uint2*>(row + 2 * W3_PLANE) = pl;
        }
    };

    // wave w owns output rows co0 + 32 w .. + 31 (plane rows 32 w + rho <-> channel 32 w + 4 (rho % 8) + rho / 8) and all four column tiles
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const bool rows_live = co0 + 32 * wave < p.Cout;           // (wave-uniform: a row tile beyond Cout multiplies nothing)

    // fragment (row tile t of the 128 plane rows, plane): lane (row rho = lane & 31, k-half lane >> 5) reads 16 bytes
    const int fr_off = (lane >> 5) * (WG_T * 16) + (lane & 31) * 16;
    const int nst = (int)((m_hi - m_lo + W3_K - 1) / W3_K);
    if (nst > 0) {
        load(m_lo);
        store(0);
        __syncthreads();
        for (int st = 0; st < nst; ++st) {
            const int cur = st & 1;
            if (st + 1 < nst) load(m_lo + (int64_t)(st + 1) * W3_K);
            if (rows_live) {
                const char* a = w3sm + cur * W3_BUF + fr_off + wave * (32 * 16);
                const char* b = w3sm + cur * W3_BUF + fr_off + 3 * W3_PLANE;
                wg_bf16x8 af[3];
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) af[sp] = *reinterpret_cast<const wg_bf16x8*>(a + sp * W3_PLANE);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    wg_bf16x8 bf[3];
#pragma unroll
                    for (int sp = 0; sp < 3; ++sp) bf[sp] = *reinterpret_cast<const wg_bf16x8*>(b + sp * W3_PLANE + j * (32 * 16));
                    // smallest terms first
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[0], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[2], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], acc[j], 0, 0, 0);
                }
            }
            if (st + 1 < nst) store(cur ^ 1);
            __syncthreads();
        }
    }
    // accumulator j, register r of lane (kappa = lane & 31, half = lane >> 5): output row rho = (r & 3) + 8 (r >> 2) + 4 half of the wave's
    // row tile, column kappa of column tile j = tile column 4 kappa + j: the lane's four accumulators are four consecutive ci
    if (!rows_live) return;
    float* dst = ws ? ws + (int64_t)blockIdx.z * slab : p.gw;
    const int col = ci0 + 4 * (lane & 31);
    const int otap = p.fuse ? (col < p.ctot ? col / p.cf : 0) : tap;
    const int ci = p.fuse ? col - otap * p.cf : col;
    const bool col_any = p.fuse ? (col < p.ctot && ci < p.C) : ci < p.C;
    const bool vec = col_any && ci + 3 < p.C && (p.gw_cin & 3) == 0 && ws != nullptr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rho = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int co = co0 + 32 * wave + 4 * (rho & 7) + (rho >> 3);
        if (co < p.Cout && col_any) {
            float* q = dst + ((int64_t)co * p.kh * p.kw + otap) * p.gw_cin + ci;
            if (vec) {
                const f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
                *reinterpret_cast<f32x4*>(q) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (ci + e < p.C) { if (ws) q[e] = acc[e][r]; else atomicAdd(q + e, acc[e][r]); }
            }
        }
    }
}

// ... for MANY slabs over a small gw (the 56^2 layers: 64 x 64 weights, several hundred pixel chunks -- one thread walking them all is a
// chain of several hundred dependent-latency loads): 16 threads per 16-byte column, thread s adds slabs s, s + 16, ... in order, the 16
// partial sums meet in LDS and are added in the order s = 0 .. 15.  One fixed association of the sum, whatever the dispatch.
__global__ __launch_bounds__(256) void wgrad_combine_wide_kernel(const f32x4* __restrict__ ws, f32x4* __restrict__ gw, int64_t n4, int64_t slab4, int split) {
    __shared__ f32x4 part[16][16];
    const int el = threadIdx.x & 15, sg = threadIdx.x >> 4;
    const int64_t e = (int64_t)blockIdx.x * 16 + el;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (e < n4) {
        int z = sg;
        for (; z + 48 < split; z += 64) {
            const f32x4 v0 = ws[e + (int64_t)z * slab4], v1 = ws[e + (int64_t)(z + 16) * slab4];
            const f32x4 v2 = ws[e + (int64_t)(z + 32) * slab4], v3 = ws[e + (int64_t)(z + 48) * slab4];
            acc += v0; acc += v1; acc += v2; acc += v3;
        }
        for (; z < split; z += 16) acc += ws[e + (int64_t)z * slab4];
    }
    part[sg][el] = acc;
    __syncthreads();
    if (sg == 0 && e < n4) {
        f32x4 t = part[0][el];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += part[k][el];
        gw[e] = t;
    }
}

// gw[e] = sum over the pixel chunks' slabs, in chunk order (16-byte columns, four slabs in flight)
__global__ __launch_bounds__(256) void wgrad_combine_kernel(const f32x4* __restrict__ ws, f32x4* __restrict__ gw, int64_t n4, int64_t slab4, int split) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += stride) {
        f32x4 acc = ws[e];
        int z = 1;
        for (; z + 3 < split; z += 4) {
            const f32x4 v0 = ws[e + (int64_t)z * slab4], v1 = ws[e + (int64_t)(z + 1) * slab4];
            const f32x4 v2 = ws[e + (int64_t)(z + 2) * slab4], v3 = ws[e + (int64_t)(z + 3) * slab4];
            acc += v0; acc += v1; acc += v2; acc += v3;
        }
        for (; z < split; ++z) acc += ws[e + (int64_t)z * slab4];
        gw[e] = acc;
    }
}

// ---- per-channel sums over rows: out[c] += sum_r (a[r,c] - sa[c]) * (b ? b[r,c] - sb[c] : 1) -----------------------------------
// `partial` != NULL: the workgroup's sums go to partial[blockIdx.x][C] instead (no atomics; colsum_finish_kernel adds the rows of
// `partial` in order: every sum of the launch then has ONE fixed order -- bcos_colsum_ws)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     const float* __restrict__ sa, const float* __restrict__ sb,
                                                     float* __restrict__ out, int64_t rows, int C, int64_t rows_per_block,
                                                     float* __restrict__ partial) {
    __shared__ float red[256 * 4];
    const int c4 = C / 4;                                  // float4 columns
    const int tpc = c4 < 256 ? c4 : 256;                   // threads along the channel dimension
    const int rstride = 256 / tpc;                         // rows handled concurrently
    const int tc = threadIdx.x % tpc, tr = threadIdx.x / tpc;
    const int64_t r_lo = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r_hi = r_lo + rows_per_block < rows ? r_lo + rows_per_block : rows;
    for (int base = 0; base < c4; base += tpc) {            // uniform trip count: the loop body holds barriers
        const int cg = base + tc;
        const bool live = cg < c4 && tr < rstride;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        f32x4 sa4 = {0.f, 0.f, 0.f, 0.f}, sb4 = {0.f, 0.f, 0.f, 0.f};
        if (live && sa) sa4 = *reinterpret_cast<const f32x4*>(sa + cg * 4);
        if (live && sb) sb4 = *reinterpret_cast<const f32x4*>(sb + cg * 4);
        if (live)
            for (int64_t r = r_lo + tr; r < r_hi; r += rstride) {
                f32x4 va = *reinterpret_cast<const f32x4*>(a + r * C + cg * 4) - sa4;
                if (b) va *= *reinterpret_cast<const f32x4*>(b + r * C + cg * 4) - sb4;
                acc += va;
            }
        // combine the rstride partial sums of this channel group through LDS
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) red[threadIdx.x * 4 + q] = acc[q];
        __syncthreads();
        if (tr == 0 && cg < c4) {
            for (int k = 1; k < rstride; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] += red[(k * tpc + tc) * 4 + q];
            if (partial) {
                *reinterpret_cast<f32x4*>(partial + (int64_t)blockIdx.x * C + cg * 4) = acc;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) atomicAdd(out + cg * 4 + q, acc[q]);
            }
        }
    }
}

// out[c] = sum_b partial[b][c] in ONE fixed order: workgroup = 16 channels x 16 lanes; lane t adds rows t, t + 16, ... in four
// interleaved chains (independent loads in flight), the 64 partial sums of a channel meet in a fixed tree
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ partial, float* __restrict__ out, int nblk, int C) {
    __shared__ float red[256];
    const int tc = threadIdx.x & 15, tr = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + tc;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < C) {
        int b = tr;
        for (; b + 48 < nblk; b += 64) {
            a0 += partial[(int64_t)b * C + c];
            a1 += partial[(int64_t)(b + 16) * C + c];
            a2 += partial[(int64_t)(b + 32) * C + c];
            a3 += partial[(int64_t)(b + 48) * C + c];
        }
        for (; b < nblk; b += 16) a0 += partial[(int64_t)b * C + c];
    }
    red[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (tr == 0 && c < C) {
        float acc = red[tc];
        for (int k = 1; k < 16; ++k) acc += red[k * 16 + tc];
        out[c] = acc;
    }
}

// the same sums in a fixed order: workgroup = 16 float4 column groups (64 channels) x 16 row lanes over ALL rows; no atomics
__global__ __launch_bounds__(256) void colsum_ordered_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             const float* __restrict__ sa, const float* __restrict__ sb,
                                                             float* __restrict__ out, int64_t rows, int C) {
    __shared__ float red[256 * 4];
    const int c4 = C / 4;
    const int tc = threadIdx.x & 15, tr = threadIdx.x >> 4;
    const int cg = blockIdx.x * 16 + tc;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (cg < c4) {
        f32x4 sa4 = {0.f, 0.f, 0.f, 0.f}, sb4 = {0.f, 0.f, 0.f, 0.f};
        if (sa) sa4 = *reinterpret_cast<const f32x4*>(sa + cg * 4);
        if (sb) sb4 = *reinterpret_cast<const f32x4*>(sb + cg * 4);
        for (int64_t r = tr; r < rows; r += 16) {
            f32x4 va = *reinterpret_cast<const f32x4*>(a + r * C + cg * 4) - sa4;
            if (b) va *= *reinterpret_cast<const f32x4*>(b + r * C + cg * 4) - sb4;
            acc += va;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) red[threadIdx.x * 4 + q] = acc[q];
    __syncthreads();
    if (tr == 0 && cg < c4) {
        for (int k = 1; k < 16; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += red[(k * 16 + tc) * 4 + q];
        *reinterpret_cast<f32x4*>(out + cg * 4) = acc;
    }
}

// out[r,c] = a[r,c] * sa[c] + (b[r,c] - mb[c]) * sb[c]        (b, mb, sb optional as a group)
__global__ __launch_bounds__(256) void channel_axpby_kernel(const float* __restrict__ a, const float* __restrict__ sa,
                                                            const float* __restrict__ b, const float* __restrict__ mb,
                                                            const float* __restrict__ sb, float* __restrict__ out,
                                                            int64_t n4, int c4) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const int cg = (int)(i % c4);
        f32x4 v = reinterpret_cast<const f32x4*>(a)[i] * reinterpret_cast<const f32x4*>(sa)[cg];
        if (b) {
            f32x4 w = reinterpret_cast<const f32x4*>(b)[i];
            if (mb) w -= reinterpret_cast<const f32x4*>(mb)[cg];
            v += w * reinterpret_cast<const f32x4*>(sb)[cg];
        }
        reinterpret_cast<f32x4*>(out)[i] = v;
    }
}


// ---- round 5: the batch statistics of a BatchNormUncentered2d in ONE pass over y, and the sums of its backward fused with the ReLU gate
// (profiles/r04_kernel_stats_train_resnet50.csv: colsum_kernel was the largest entry of a training step -- four streaming passes per
// BatchNorm, each behind ~10 tiny torch launches) ----------------------------------------------------------------------------------------
// Workgroup b owns rows [b rpb, (b + 1) rpb).  It sums (y - c) and (y - c)^2 per channel with the shift c = its FIRST row (the data
// itself: |mean_b - c| is of the order of the spread, so the one-pass second moment loses no more than a bit or two to cancellation)
// and leaves (c, S1, S2) in the workspace; bn_stats_finish_kernel combines the workgroups' (n_b, mean_b, M2_b) in one fixed order
// (Chan et al.): mean = sum n_b mean_b / m, M2 = sum [M2_b + n_b (mean_b - mean)^2] -- the centred variance x.var(unbiased=False)
// of batchnorm_uncentered.py:36-44 without a second pass over the tensor.
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ y, float* __restrict__ partial, int64_t rows, int C,
                                                               int64_t rows_per_block) {
    __shared__ float red[2][256 * 4];
    const int c4 = C / 4;
    const int tpc = c4 < 256 ? c4 : 256;
    const int rstride = 256 / tpc;
    const int tc = threadIdx.x % tpc, tr = threadIdx.x / tpc;
    const int64_t r_lo = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r_hi = r_lo + rows_per_block < rows ? r_lo + rows_per_block : rows;
    for (int base = 0; base < c4; base += tpc) {
        const int cg = base + tc;
        const bool live = cg < c4 && tr < rstride;
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            c = *reinterpret_cast<const f32x4*>(y + r_lo * C + cg * 4);
            int64_t r = r_lo + tr;
            for (; r + 3 * rstride < r_hi; r += 4 * rstride) {      // four rows in flight per thread (one fixed order: the chain below)
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(y + (r + u * rstride) * C + cg * 4);
#pragma unroll
                for (int u = 0; u < 4; ++u) { const f32x4 d = v[u] - c; s1 += d; s2 += d * d; }
            }
            for (; r < r_hi; r += rstride) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(y + r * C + cg * 4) - c;
                s1 += v;
                s2 += v * v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) { red[0][threadIdx.x * 4 + q] = s1[q]; red[1][threadIdx.x * 4 + q] = s2[q]; }
        __syncthreads();
        if (tr == 0 && cg < c4) {
            for (int k = 1; k < rstride; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) { s1[q] += red[0][(k * tpc + tc) * 4 + q]; s2[q] += red[1][(k * tpc + tc) * 4 + q]; }
            float* dst = partial + (int64_t)blockIdx.x * 3 * C + cg * 4;
            *reinterpret_cast<f32x4*>(dst) = c;
            *reinterpret_cast<f32x4*>(dst + C) = s1;
            *reinterpret_cast<f32x4*>(dst + 2 * C) = s2;
        }
    }
}

// fixed-order sum over the workgroups b of f(b) for one channel: 16 lanes, lane t takes b = t, t + 16, ... in order, the 16 partial
// sums meet in order (the shape of colsum_finish_kernel)
template <typename F>
__device__ __forceinline__ float ordered_block_sum(float* red, int tc, int tr, int nblk, bool live, F f) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;       // four interleaved chains per lane: independent loads in flight
    if (live) {
        int b = tr;
        for (; b + 48 < nblk; b += 64) { a0 += f(b); a1 += f(b + 16); a2 += f(b + 32); a3 += f(b + 48); }
        for (; b < nblk; b += 16) a0 += f(b);
    }
    __syncthreads();
    red[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    float acc = red[tc];
    for (int k = 1; k < 16; ++k) acc += red[k * 16 + tc];
    return acc;
}

__global__ __launch_bounds__(256) void bn_stats_finish_kernel(const float* __restrict__ partial, const float* __restrict__ weight,
                                                              float* __restrict__ running_var, float* __restrict__ mean_out,
                                                              float* __restrict__ var_out, float* __restrict__ rstd_out, float* __restrict__ g_out,
                                                              int nblk, int C, int64_t rows, int64_t rpb, float eps, float momentum) {
    __shared__ float red[256];
    const int tc = threadIdx.x & 15, tr = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + tc;
    const bool live = c < C;
    auto nb = [&](int b) { const int64_t lo = (int64_t)b * rpb; return (float)((lo + rpb < rows ? lo + rpb : rows) - lo); };
    const float m = (float)rows;
    const float total = ordered_block_sum(red, tc, tr, nblk, live, [&](int b) {
        const float* q = partial + (int64_t)b * 3 * C + c;
        return nb(b) * q[0] + q[C];
    });
    const float mean = total / m;
    const float m2 = ordered_block_sum(red, tc, tr, nblk, live, [&](int b) {
        const float* q = partial + (int64_t)b * 3 * C + c;
        const float n = nb(b), s1 = q[C];
        const float d = q[0] + s1 / n - mean;
        return q[2 * C] - s1 * s1 / n + n * d * d;
    });
    if (tr == 0 && live) {
        const float var = fmaxf(m2 / m, 0.f);
        const float rstd = rsqrtf(var + eps);
        mean_out[c] = mean;
        var_out[c] = var;
        rstd_out[c] = rstd;
        g_out[c] = weight ? weight[c] * rstd : rstd;
        if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * var;
    }
}

// ga = act > 0 ? g : 0 (act NULL: ga = g, not written) with the two column sums of the BatchNorm backward from the same pass:
// partial[b] = (sum ga y, sum ga) per channel
__global__ __launch_bounds__(256) void relu_bwd_colsums_kernel(const float* __restrict__ g, const float* __restrict__ act, const float* __restrict__ y,
                                                               float* __restrict__ ga, float* __restrict__ partial, int64_t rows, int C,
                                                               int64_t rows_per_block) {
    __shared__ float red[2][256 * 4];
    const int c4 = C / 4;
    const int tpc = c4 < 256 ? c4 : 256;
    const int rstride = 256 / tpc;
    const int tc = threadIdx.x % tpc, tr = threadIdx.x / tpc;
    const int64_t r_lo = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r_hi = r_lo + rows_per_block < rows ? r_lo + rows_per_block : rows;
    for (int base = 0; base < c4; base += tpc) {
        const int cg = base + tc;
        const bool live = cg < c4 && tr < rstride;
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            auto rows_u = [&](auto u_c, int64_t r) {                 // U rows in flight per thread: all loads first, then the chain in row order
                constexpr int U = decltype(u_c)::value;
                f32x4 v[U], a[U], yy[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t o = (r + u * rstride) * C + cg * 4;
                    v[u] = *reinterpret_cast<const f32x4*>(g + o);
                    yy[u] = *reinterpret_cast<const f32x4*>(y + o);
                    if (act) a[u] = *reinterpret_cast<const f32x4*>(act + o);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (act) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[u][q] = a[u][q] > 0.f ? v[u][q] : 0.f;
                        *reinterpret_cast<f32x4*>(ga + (r + u * rstride) * C + cg * 4) = v[u];
                    }
                    s1 += v[u] * yy[u];
                    s2 += v[u];
                }
            };
            int64_t r = r_lo + tr;
            for (; r + 3 * rstride < r_hi; r += 4 * rstride) rows_u(std::integral_constant<int, 4>{}, r);
            for (; r < r_hi; r += rstride) rows_u(std::integral_constant<int, 1>{}, r);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) { red[0][threadIdx.x * 4 + q] = s1[q]; red[1][threadIdx.x * 4 + q] = s2[q]; }
        __syncthreads();
        if (tr == 0 && cg < c4) {
            for (int k = 1; k < rstride; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) { s1[q] += red[0][(k * tpc + tc) * 4 + q]; s2[q] += red[1][(k * tpc + tc) * 4 + q]; }
            float* dst = partial + (int64_t)blockIdx.x * 2 * C + cg * 4;
            *reinterpret_cast<f32x4*>(dst) = s1;
            *reinterpret_cast<f32x4*>(dst + C) = s2;
        }
    }
}

// sgx = sum_b partial[b][0], sg = sum_b partial[b][1] in one fixed order; with the forward's rstd and g = weight rstd also the weight
// gradient sgx rstd and the coefficient of the variance term of the input gradient, -(g sgx) rstd^2 / m (batchnorm_uncentered.py:36-44)
__global__ __launch_bounds__(256) void bn_bwd_finish_kernel(const float* __restrict__ partial, const float* __restrict__ rstd, const float* __restrict__ gvec,
                                                            float* __restrict__ sgx_out, float* __restrict__ sg_out, float* __restrict__ gw_out,
                                                            float* __restrict__ coef_out, int nblk, int C, int64_t rows) {
    __shared__ float red[256];
    const int tc = threadIdx.x & 15, tr = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + tc;
    const bool live = c < C;
    const float sgx = ordered_block_sum(red, tc, tr, nblk, live, [&](int b) { return partial[(int64_t)b * 2 * C + c]; });
    const float sg = ordered_block_sum(red, tc, tr, nblk, live, [&](int b) { return partial[(int64_t)b * 2 * C + C + c]; });
    if (tr == 0 && live) {
        sgx_out[c] = sgx;
        if (sg_out) sg_out[c] = sg;
        if (gw_out) gw_out[c] = sgx * rstd[c];
        if (coef_out) coef_out[c] = -(gvec[c] * sgx) * rstd[c] * rstd[c] / (float)rows;
    }
}


// y = [relu](x * scale[c] + shift[c] (+ addend)) row by row, with the per-row max |y| (fp32 bit pattern) the next contraction takes as
// its operand scale (round 5: forward and input-gradient contractions of a training step on the 3-product split-f16 loop instead of the
// 6-product bf16 one).  LPR lanes per row: C / 4 when that divides 64 (8 / 4 / 2 rows per wavefront), else 64 lanes looping over the row.
template <int LPR>
__global__ __launch_bounds__(256) void channel_affine_rows_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, const float* __restrict__ addend,
                                                                  float* __restrict__ y, unsigned* __restrict__ absmax, int64_t rows, int C,
                                                                  int relu) {
    constexpr int RPW = 64 / LPR;                        // rows per wavefront
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR, l = lane % LPR;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r0 = wave * RPW; r0 < rows; r0 += nwaves * RPW) {        // (wave-uniform trip count: the shuffles below see every lane)
        const int64_t row = r0 + sub;
        const bool live = row < rows;
        unsigned mx = 0u;
        if (live)
            for (int c = l * 4; c < C; c += LPR * 4) {
                const int64_t i = row * C + c;
                f32x4 v = *reinterpret_cast<const f32x4*>(x + i) * *reinterpret_cast<const f32x4*>(scale + c);
                if (shift) v += *reinterpret_cast<const f32x4*>(shift + c);
                if (addend) v += *reinterpret_cast<const f32x4*>(addend + i);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (relu) v[q] = fmaxf(v[q], 0.f);
                    mx = max(mx, __float_as_uint(v[q]) & 0x7fffffffu);
                }
                *reinterpret_cast<f32x4*>(y + i) = v;
            }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
        if (live && l == 0) absmax[row] = mx;
    }
}

}  // namespace

extern "C" int bcos_train_scale_bwd_absmax(const float* gy, const float* y, const float* s, const float* norm, float* glin,
                                           float* rnorm, float* bgrad, uint32_t* glin_absmax, int64_t rows, int C, int bcos_mode, float b,
                                           int force_pow, void* stream) {
    if (!gy || !y || !s || !norm || !glin || !rnorm || rows <= 0 || C <= 0 || C % 4 != 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_train_scale_bwd: bad argument (C must be a multiple of 4)");
    if (bcos_mode != BCOS_CONV_EPS && bcos_mode != BCOS_LINEAR_EPS)
        return bcos_set_error(BCOS_E_INVAL, "bcos_train_scale_bwd: bcos_mode must be BCOS_CONV_EPS or BCOS_LINEAR_EPS");
    if (b == 1.0f) return bcos_set_error(BCOS_E_INVAL, "bcos_train_scale_bwd: B == 1 has no dynamic scale");
    const int pow_form = (b != 2.0f || force_pow) ? 1 : 0;
    if (bgrad && !pow_form)
        return bcos_set_error(BCOS_E_INVAL, "bcos_train_scale_bwd: the |lin| / norm form (B == 2 without force_pow) does not depend on B");
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(scale_bwd_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       gy, y, s, norm, nullptr, nullptr, nullptr, glin, rnorm, bgrad, glin_absmax, rows, C, bcos_mode == BCOS_LINEAR_EPS ? 1 : 0, b,
                       pow_form);
    return check_launch("train_scale_bwd launch");
}

extern "C" int bcos_train_scale_bwd_bn(const float* g_out, const float* y, const float* s, const float* norm, const float* bn_g,
                                       const float* bn_mean, const float* bn_coef, float* glin, float* rnorm, uint32_t* glin_absmax,
                                       int64_t rows, int C, int bcos_mode, float b, int force_pow, void* stream) {
    if (!g_out || !y || !s || !norm || !bn_g || !glin || !rnorm || rows <= 0 || C <= 0 || C % 4 != 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_train_scale_bwd_bn: bad argument (C must be a multiple of 4)");
    if (bn_mean && !bn_coef) return bcos_set_error(BCOS_E_INVAL, "bcos_train_scale_bwd_bn: bn_mean without bn_coef");
    if (bcos_mode != BCOS_CONV_EPS && bcos_mode != BCOS_LINEAR_EPS)
        return bcos_set_error(BCOS_E_INVAL, "bcos_train_scale_bwd_bn: bcos_mode must be BCOS_CONV_EPS or BCOS_LINEAR_EPS");
    if (b == 1.0f) return bcos_set_error(BCOS_E_INVAL, "bcos_train_scale_bwd_bn: B == 1 has no dynamic scale");
    if ((reinterpret_cast<uintptr_t>(bn_g) | reinterpret_cast<uintptr_t>(bn_mean) | reinterpret_cast<uintptr_t>(bn_coef)) & 15)
        return bcos_set_error(BCOS_E_INVAL, "bcos_train_scale_bwd_bn: the channel vectors must be 16-byte aligned");
    const int pow_form = (b != 2.0f || force_pow) ? 1 : 0;
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(scale_bwd_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       g_out, y, s, norm, bn_g, bn_mean, bn_coef, glin, rnorm, nullptr, glin_absmax, rows, C,
                       bcos_mode == BCOS_LINEAR_EPS ? 1 : 0, b, pow_form);
    return check_launch("train_scale_bwd_bn launch");
}

extern "C" int bcos_train_scale_bwd(const float* gy, const float* y, const float* s, const float* norm, float* glin,
                                    float* rnorm, float* bgrad, int64_t rows, int C, int bcos_mode, float b, int force_pow,
                                    void* stream) {
    return bcos_train_scale_bwd_absmax(gy, y, s, norm, glin, rnorm, bgrad, nullptr, rows, C, bcos_mode, b, force_pow, stream);
}

extern "C" int bcos_channel_affine_rows(const float* x, const float* scale, const float* shift, const float* addend, float* y,
                                        uint32_t* y_absmax, int64_t rows, int C, int relu, void* stream) {
    if (!x || !scale || !y || !y_absmax || rows <= 0 || C <= 0 || C % 4 != 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_channel_affine_rows: bad argument (C % 4)");
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(addend) | reinterpret_cast<uintptr_t>(scale) |
         reinterpret_cast<uintptr_t>(shift)) & 15)
        return bcos_set_error(BCOS_E_INVAL, "bcos_channel_affine_rows: tensors must be 16-byte aligned");
    const int C4 = C / 4;
    const int lpr = (C4 < 64 && 64 % C4 == 0) ? C4 : 64;
    const int64_t waves = (rows + (64 / lpr) - 1) / (64 / lpr);
    int64_t blocks = (waves + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define BCOS_CAR(L) hipLaunchKernelGGL(channel_affine_rows_kernel<L>, dim3((unsigned)blocks), dim3(256), 0, st, x, scale, shift, addend, y, y_absmax, rows, C, relu)
    switch (lpr) {
        case 1: BCOS_CAR(1); break;
        case 2: BCOS_CAR(2); break;
        case 4: BCOS_CAR(4); break;
        case 8: BCOS_CAR(8); break;
        case 16: BCOS_CAR(16); break;
        case 32: BCOS_CAR(32); break;
        default: BCOS_CAR(64); break;
    }
#undef BCOS_CAR
    return check_launch("channel_affine_rows launch");
}

extern "C" int bcos_weight_rownorm_bwd(const float* w, const float* g_eff, const float* gain, float* gw, float* ggain, int rows,
                                       int64_t cols, void* stream) {
    if (!w || !g_eff || (!gw && !ggain) || rows <= 0 || cols <= 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_weight_rownorm_bwd: bad argument");
    hipLaunchKernelGGL(rownorm_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       w, g_eff, gain, gw, ggain, rows, cols);
    return check_launch("weight_rownorm_bwd launch");
}

extern "C" int bcos_maxout_scatter(const float* g, const int32_t* argmax, float* full, int64_t rows, int Cout, int max_out,
                                   void* stream) {
    if (!g || !argmax || !full || rows <= 0 || Cout <= 0 || max_out <= 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_maxout_scatter: bad argument");
    const int64_t n = rows * Cout;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(maxout_scatter_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, argmax,
                       full, n, max_out);
    return check_launch("maxout_scatter launch");
}

extern "C" int bcos_patch_norm_bwd_add(const float* x, const float* rnorm, const float* addend, float* out, int N, int H, int W, int C,
                                       int x_pitch, int P, int Q, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, void* stream) {
    if (!x || !rnorm || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0 || P <= 0 || Q <= 0 || kh <= 0 || kw <= 0 ||
        sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_patch_norm_bwd: bad argument");
    if (x_pitch == 0) x_pitch = C;
    if (x_pitch < C) return bcos_set_error(BCOS_E_INVAL, "bcos_patch_norm_bwd: bad x_pitch");
    const int vec = (C % 4 == 0 && x_pitch % 4 == 0 &&
                     !((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(addend)) & 15)) ? 1 : 0;
    const int64_t total = (int64_t)N * H * W;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (vec && (C == 64 || C == 128) && kh * kw <= C / 4) {
        const int ppb = 4 * (256 / C);             // pixels per workgroup
        const dim3 grid((unsigned)((total + ppb - 1) / ppb));
        if (C == 64)
            hipLaunchKernelGGL(patch_norm_bwd_group_kernel<16>, grid, dim3(256), 0, s, x, rnorm, addend, out, N, H, W, C, x_pitch, P, Q, kh, kw,
                               sh, sw, ph, pw, dh, dw);
        else
            hipLaunchKernelGGL(patch_norm_bwd_group_kernel<32>, grid, dim3(256), 0, s, x, rnorm, addend, out, N, H, W, C, x_pitch, P, Q, kh, kw,
                               sh, sw, ph, pw, dh, dw);
        return check_launch("patch_norm_bwd launch");
    }
    hipLaunchKernelGGL(patch_norm_bwd_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s,
                       x, rnorm, addend, out, N, H, W, C, x_pitch, P, Q, kh, kw, sh, sw, ph, pw, dh, dw, vec);
    return check_launch("patch_norm_bwd launch");
}

extern "C" int bcos_patch_norm_bwd(const float* x, const float* rnorm, float* out, int N, int H, int W, int C, int x_pitch,
                                   int P, int Q, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, void* stream) {
    return bcos_patch_norm_bwd_add(x, rnorm, nullptr, out, N, H, W, C, x_pitch, P, Q, kh, kw, sh, sw, ph, pw, dh, dw, stream);
}

extern "C" int bcos_conv2d_wgrad(const float* glin, const float* x, float* gw, int N, int H, int W, int C, int x_pitch, int P,
                                 int Q, int Cout, int g_pitch, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                                 int gw_cin, void* stream) {
    if (!glin || !x || !gw || N <= 0 || H <= 0 || W <= 0 || C <= 0 || P <= 0 || Q <= 0 || Cout <= 0 || kh <= 0 || kw <= 0 ||
        sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_conv2d_wgrad: bad argument");
    if (x_pitch == 0) x_pitch = C;
    if (g_pitch == 0) g_pitch = Cout;
    if (gw_cin == 0) gw_cin = C;
    if (x_pitch % 4 != 0 || g_pitch % 4 != 0 || x_pitch < C || g_pitch < Cout || gw_cin < C ||
        ((reinterpret_cast<uintptr_t>(glin) | reinterpret_cast<uintptr_t>(x)) & 15))
        return bcos_set_error(BCOS_E_INVAL, "bcos_conv2d_wgrad: operands must be 16-byte addressable per pixel");
    WgradArgs p;
    p.glin = glin; p.x = x; p.gw = gw;
    p.N = N; p.H = H; p.W = W; p.C = C; p.x_pitch = x_pitch;
    p.P = P; p.Q = Q; p.Cout = Cout; p.g_pitch = g_pitch;
    p.kh = kh; p.kw = kw; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.dh = dh; p.dw = dw;
    p.gw_cin = gw_cin;
    p.M = (int64_t)N * P * Q;
    const int tiles_co = (Cout + WG_T - 1) / WG_T;
    p.cf = (C + 3) & ~3;
    p.ctot = kh * kw * p.cf;
    p.fuse = (p.cf < WG_T && kh * kw > 1 && x_pitch >= p.cf) ? 1 : 0;
    p.tiles_ci = ((p.fuse ? p.ctot : C) + WG_T - 1) / WG_T;
    const int grid_y = p.fuse ? 1 : kh * kw;
    const int64_t tiles = (int64_t)tiles_co * p.tiles_ci * grid_y;
    // split the pixels so that ~4 workgroups per CU are in flight, chunks of at least 256 pixels
    int64_t split = (1024 + tiles - 1) / tiles;
    const int64_t max_split = (p.M + 255) / 256;
    if (split > max_split) split = max_split;
    if (split < 1) split = 1;
    if (split > 65535) split = 65535;
    p.chunk = (((p.M + split - 1) / split) + WG_K - 1) / WG_K * WG_K;
    split = (p.M + p.chunk - 1) / p.chunk;
    const size_t lds = (size_t)4 * WG_K * WG_LD * sizeof(float);
    static std::atomic<size_t> lds_hw[2];
    // exact fp32 MFMA in contraction mode f32, the 6-product bf16 split otherwise (the split modes of the other contractions)
    const bool x3 = bcos_get_contraction_mode() != 0;
    const dim3 grid((unsigned)(tiles_co * p.tiles_ci), (unsigned)grid_y, (unsigned)split);
    hipError_t e = x3 ? bcos_ensure_dynamic_lds(reinterpret_cast<const void*>(wgrad_kernel<true>), lds, lds_hw[1])
                      : bcos_ensure_dynamic_lds(reinterpret_cast<const void*>(wgrad_kernel<false>), lds, lds_hw[0]);
    if (e != hipSuccess) return bcos_set_hip_error("hipFuncSetAttribute", e);
    if (x3) hipLaunchKernelGGL(wgrad_kernel<true>, grid, dim3(256), lds, reinterpret_cast<hipStream_t>(stream), p);
    else hipLaunchKernelGGL(wgrad_kernel<false>, grid, dim3(256), lds, reinterpret_cast<hipStream_t>(stream), p);
    return check_launch("wgrad launch");
}

// ---- round 6: the same weight gradient with the split at staging and a fixed-order combine (wgrad3_kernel) ------------------------------
namespace {
struct Wgrad3Plan { WgradArgs p; int tiles_co, grid_y; int64_t split, slab; };

int wgrad3_plan(Wgrad3Plan& pl, const float* glin, const float* x, float* gw, int N, int H, int W, int C, int x_pitch, int P, int Q, int Cout,
                int g_pitch, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int gw_cin) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || P <= 0 || Q <= 0 || Cout <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_conv2d_wgrad: bad argument");
    if (x_pitch == 0) x_pitch = C;
    if (g_pitch == 0) g_pitch = Cout;
    if (gw_cin == 0) gw_cin = C;
    if (x_pitch % 4 != 0 || g_pitch % 4 != 0 || x_pitch < C || g_pitch < Cout || gw_cin < C)
        return bcos_set_error(BCOS_E_INVAL, "bcos_conv2d_wgrad: operands must be 16-byte addressable per pixel");
    WgradArgs& p = pl.p;
    p.glin = glin; p.x = x; p.gw = gw;
    p.N = N; p.H = H; p.W = W; p.C = C; p.x_pitch = x_pitch;
    p.P = P; p.Q = Q; p.Cout = Cout; p.g_pitch = g_pitch;
    p.kh = kh; p.kw = kw; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.dh = dh; p.dw = dw;
    p.gw_cin = gw_cin;
    p.M = (int64_t)N * P * Q;
    pl.tiles_co = (Cout + WG_T - 1) / WG_T;
    p.cf = (C + 3) & ~3;
    p.ctot = kh * kw * p.cf;
    p.fuse = (p.cf < WG_T && kh * kw > 1 && x_pitch >= p.cf) ? 1 : 0;
    p.tiles_ci = ((p.fuse ? p.ctot : C) + WG_T - 1) / WG_T;
    pl.grid_y = p.fuse ? 1 : kh * kw;
    const int64_t tiles = (int64_t)pl.tiles_co * p.tiles_ci * pl.grid_y;
    // pixel chunks: ~3 workgroups per CU in flight (what the kernel is compiled for), chunks of at least 256 pixels; every chunk costs a
    // slab of the workspace and a term of the combine
    const int64_t want = bcos_option(BCOS_OPT_WGRAD_WGS);
    int64_t split = (want + tiles - 1) / tiles;
    const int64_t max_split = (p.M + 255) / 256;
    if (split > max_split) split = max_split;
    if (split < 1) split = 1;
    if (split > 65535) split = 65535;
    p.chunk = (((p.M + split - 1) / split) + W3_K - 1) / W3_K * W3_K;
    pl.split = (p.M + p.chunk - 1) / p.chunk;
    pl.slab = (((int64_t)Cout * kh * kw * gw_cin) + 3) & ~(int64_t)3;
    return BCOS_OK;
}
}  // namespace

extern "C" int bcos_conv2d_wgrad_ws_floats(int N, int H, int W, int C, int x_pitch, int P, int Q, int Cout, int g_pitch, int kh, int kw,
                                           int sh, int sw, int ph, int pw, int dh, int dw, int gw_cin, int64_t* floats) {
    if (!floats) return bcos_set_error(BCOS_E_INVAL, "bcos_conv2d_wgrad_ws_floats: NULL result");
    Wgrad3Plan pl;
    const int rc = wgrad3_plan(pl, nullptr, nullptr, nullptr, N, H, W, C, x_pitch, P, Q, Cout, g_pitch, kh, kw, sh, sw, ph, pw, dh, dw, gw_cin);
    if (rc != BCOS_OK) return rc;
    *floats = pl.split > 1 ? pl.split * pl.slab : 0;        // (one chunk: the kernel stores straight into gw)
    return BCOS_OK;
}

extern "C" int bcos_conv2d_wgrad_ordered(const float* glin, const float* x, float* gw, float* ws, int N, int H, int W, int C, int x_pitch,
                                         int P, int Q, int Cout, int g_pitch, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                                         int gw_cin, void* stream) {
    if (!glin || !x || !gw) return bcos_set_error(BCOS_E_INVAL, "bcos_conv2d_wgrad_ordered: NULL tensor");
    if ((reinterpret_cast<uintptr_t>(glin) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gw) | reinterpret_cast<uintptr_t>(ws)) & 15)
        return bcos_set_error(BCOS_E_INVAL, "bcos_conv2d_wgrad_ordered: tensors must be 16-byte aligned");
    if (bcos_get_contraction_mode() == 0)
        return bcos_set_error(BCOS_E_NOSUP, "bcos_conv2d_wgrad_ordered: the bf16x3 weight gradient (contraction modes bf16x3 / f16x2); mode f32 uses bcos_conv2d_wgrad");
    Wgrad3Plan pl;
    const int rc = wgrad3_plan(pl, glin, x, gw, N, H, W, C, x_pitch, P, Q, Cout, g_pitch, kh, kw, sh, sw, ph, pw, dh, dw, gw_cin);
    if (rc != BCOS_OK) return rc;
    if (pl.split > 1 && !ws) return bcos_set_error(BCOS_E_INVAL, "bcos_conv2d_wgrad_ordered: this geometry needs a workspace (bcos_conv2d_wgrad_ws_floats)");
    if ((int64_t)Cout * kh * kw * pl.p.gw_cin % 4 != 0 || pl.p.gw_cin != C)
        return bcos_set_error(BCOS_E_NOSUP, "bcos_conv2d_wgrad_ordered: gw must be dense (gw_cin == C) and hold a multiple of 4 floats");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t lds = (size_t)2 * W3_BUF;
    static std::atomic<size_t> lds_hw;
    hipError_t e = bcos_ensure_dynamic_lds(reinterpret_cast<const void*>(wgrad3_kernel), lds, lds_hw);
    if (e != hipSuccess) return bcos_set_hip_error("hipFuncSetAttribute", e);
    const dim3 grid((unsigned)(pl.tiles_co * pl.p.tiles_ci), (unsigned)pl.grid_y, (unsigned)pl.split);
    // one chunk: its slab IS gw (plain stores, every element written once: no zero fill either)
    hipLaunchKernelGGL(wgrad3_kernel, grid, dim3(256), lds, s, pl.p, pl.split > 1 ? ws : gw, pl.slab);
    int rc2 = check_launch("wgrad3 launch");
    if (rc2 != BCOS_OK || pl.split == 1) return rc2;
    const int64_t n4 = (int64_t)Cout * kh * kw * pl.p.gw_cin / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (pl.split >= 32 && n4 <= ((int64_t)1 << 20))          // many slabs, small gw: 16 threads per column (wgrad_combine_wide_kernel)
        hipLaunchKernelGGL(wgrad_combine_wide_kernel, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, s, reinterpret_cast<const f32x4*>(ws),
                           reinterpret_cast<f32x4*>(gw), n4, pl.slab / 4, (int)pl.split);
    else
        hipLaunchKernelGGL(wgrad_combine_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const f32x4*>(ws), reinterpret_cast<f32x4*>(gw),
                           n4, pl.slab / 4, (int)pl.split);
    return check_launch("wgrad combine launch");
}

extern "C" int bcos_colsum(const float* a, const float* b, const float* shift_a, const float* shift_b, float* out,
                           int64_t rows, int C, void* stream) {
    if (!a || !out || rows <= 0 || C <= 0 || C % 4 != 0) return bcos_set_error(BCOS_E_INVAL, "bcos_colsum: bad argument");
    int64_t blocks = (rows + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    const int64_t rpb = (rows + blocks - 1) / blocks;
    blocks = (rows + rpb - 1) / rpb;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, b, shift_a,
                       shift_b, out, rows, C, rpb, static_cast<float*>(nullptr));
    return check_launch("colsum launch");
}

static void colsum_ws_plan(int64_t rows, int64_t* blocks, int64_t* rpb) {
    int64_t nb = (rows + 255) / 256;        // >= 256 rows per workgroup, <= 1024 workgroups (the second launch reads them all)
    if (nb > 1024) nb = 1024;
    *rpb = (rows + nb - 1) / nb;
    *blocks = (rows + *rpb - 1) / *rpb;
}

extern "C" int bcos_colsum_ws_floats(int64_t rows, int C, int64_t* floats) {
    if (rows <= 0 || C <= 0 || C % 4 != 0 || !floats) return bcos_set_error(BCOS_E_INVAL, "bcos_colsum_ws_floats: bad argument");
    int64_t blocks, rpb;
    colsum_ws_plan(rows, &blocks, &rpb);
    *floats = blocks * C;
    return BCOS_OK;
}

extern "C" int bcos_colsum_ws(const float* a, const float* b, const float* shift_a, const float* shift_b, float* out, float* workspace,
                              int64_t workspace_floats, int64_t rows, int C, void* stream) {
    if (!a || !out || !workspace || rows <= 0 || C <= 0 || C % 4 != 0) return bcos_set_error(BCOS_E_INVAL, "bcos_colsum_ws: bad argument");
    if ((reinterpret_cast<uintptr_t>(workspace) & 15)) return bcos_set_error(BCOS_E_INVAL, "bcos_colsum_ws: workspace must be 16-byte aligned");
    int64_t blocks, rpb;
    colsum_ws_plan(rows, &blocks, &rpb);
    if (workspace_floats < blocks * C) return bcos_set_error(BCOS_E_INVAL, "bcos_colsum_ws: workspace smaller than bcos_colsum_ws_floats");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, b, shift_a, shift_b, out, rows, C, rpb, workspace);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)((C + 15) / 16)), dim3(256), 0, s, workspace, out, (int)blocks, C);
    return check_launch("colsum_ws launch");
}

static void bn_ws_plan(int64_t rows, int64_t* blocks, int64_t* rpb) {
    int64_t nb = (rows + 127) / 128;        // >= 128 rows per workgroup, <= 2048 workgroups (eight per CU keep the loads of a pass in flight)
    if (nb > 2048) nb = 2048;
    *rpb = (rows + nb - 1) / nb;
    *blocks = (rows + *rpb - 1) / *rpb;
}

extern "C" int bcos_bn_train_ws_floats(int64_t rows, int C, int64_t* floats) {
    if (rows <= 0 || C <= 0 || C % 4 != 0 || !floats) return bcos_set_error(BCOS_E_INVAL, "bcos_bn_train_ws_floats: bad argument");
    int64_t blocks, rpb;
    bn_ws_plan(rows, &blocks, &rpb);
    *floats = blocks * 3 * C;
    return BCOS_OK;
}

extern "C" int bcos_bn_batch_stats(const float* y, const float* weight, float* running_var, float* mean, float* var, float* rstd, float* g,
                                   float* workspace, int64_t workspace_floats, int64_t rows, int C, float eps, float momentum, void* stream) {
    if (!y || !mean || !var || !rstd || !g || !workspace || rows <= 0 || C <= 0 || C % 4 != 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_bn_batch_stats: bad argument (C must be a multiple of 4)");
    if ((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(y)) & 15)
        return bcos_set_error(BCOS_E_INVAL, "bcos_bn_batch_stats: y and workspace must be 16-byte aligned");
    int64_t blocks, rpb;
    bn_ws_plan(rows, &blocks, &rpb);
    if (workspace_floats < blocks * 3 * C) return bcos_set_error(BCOS_E_INVAL, "bcos_bn_batch_stats: workspace smaller than bcos_bn_train_ws_floats");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, s, y, workspace, rows, C, rpb);
    hipLaunchKernelGGL(bn_stats_finish_kernel, dim3((unsigned)((C + 15) / 16)), dim3(256), 0, s, workspace, weight, running_var, mean, var, rstd, g,
                       (int)blocks, C, rows, rpb, eps, momentum);
    return check_launch("bn_batch_stats launch");
}

extern "C" int bcos_relu_bwd_colsums(const float* g, const float* act, const float* y, float* ga, const float* rstd, const float* gvec,
                                     float* sgx, float* sg, float* gw, float* coef, float* workspace, int64_t workspace_floats, int64_t rows,
                                     int C, void* stream) {
    if (!g || !y || !sgx || !workspace || rows <= 0 || C <= 0 || C % 4 != 0 || (act && !ga) || ((gw || coef) && !rstd) || (coef && !gvec))
        return bcos_set_error(BCOS_E_INVAL, "bcos_relu_bwd_colsums: bad argument (C % 4 == 0; act needs ga; gw / coef need rstd, coef needs gvec)");
    if ((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(act) |
         reinterpret_cast<uintptr_t>(ga)) & 15)
        return bcos_set_error(BCOS_E_INVAL, "bcos_relu_bwd_colsums: tensors must be 16-byte aligned");
    int64_t blocks, rpb;
    bn_ws_plan(rows, &blocks, &rpb);
    if (workspace_floats < blocks * 2 * C) return bcos_set_error(BCOS_E_INVAL, "bcos_relu_bwd_colsums: workspace smaller than bcos_bn_train_ws_floats");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(relu_bwd_colsums_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g, act, y, ga, workspace, rows, C, rpb);
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((unsigned)((C + 15) / 16)), dim3(256), 0, s, workspace, rstd, gvec, sgx, sg, gw, coef, (int)blocks, C,
                       rows);
    return check_launch("relu_bwd_colsums launch");
}

extern "C" int bcos_colsum_ordered(const float* a, const float* b, const float* shift_a, const float* shift_b, float* out,
                                   int64_t rows, int C, void* stream) {
    if (!a || !out || rows <= 0 || C <= 0 || C % 4 != 0) return bcos_set_error(BCOS_E_INVAL, "bcos_colsum_ordered: bad argument");
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out) |
         reinterpret_cast<uintptr_t>(shift_a) | reinterpret_cast<uintptr_t>(shift_b)) & 15)
        return bcos_set_error(BCOS_E_INVAL, "bcos_colsum_ordered: tensors must be 16-byte aligned");
    hipLaunchKernelGGL(colsum_ordered_kernel, dim3((unsigned)((C / 4 + 15) / 16)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, b,
                       shift_a, shift_b, out, rows, C);
    return check_launch("colsum_ordered launch");
}

extern "C" int bcos_channel_axpby(const float* a, const float* sa, const float* b, const float* mb, const float* sb, float* out,
                                  int64_t rows, int C, void* stream) {
    if (!a || !sa || !out || rows <= 0 || C <= 0 || C % 4 != 0 || (b && !sb))
        return bcos_set_error(BCOS_E_INVAL, "bcos_channel_axpby: bad argument");
    const int64_t n4 = rows * (C / 4);
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(channel_axpby_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, sa, b,
                       mb, sb, out, n4, C / 4);
    return check_launch("channel_axpby launch");
}
