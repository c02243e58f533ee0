// bcos_skinny.hip -- direct convolution for very narrow outputs (Cout <= 8) on gfx950.
//
// The last input-gradient of a B-cosified CNN (stem dgrad: 7x7/2 conv 6->64, explanation pass of
// bcos/common.py:177) produces only 6 channels per pixel.  On the 32x32 MFMA tile of bcos_tapconv.hip
// 26 of 32 output columns are padding, and -- worse -- with so few columns per row the im2col expansion of the
// A operand (every input pixel re-read once per tap, 13 GB per launch from L2) is what bounds the kernel, not
// the matrix pipe (measured: 2.3 ms per parity class either way).  This kernel removes both:
//   * a workgroup owns a 16x16 patch of output pixels; the (16+TH-1)x(16+TW-1) input pixels it touches are
//     loaded ONCE (coalesced 16-byte loads, zero-filled outside the image) into LDS as [pixel][C+4] and every
//     tap then reads its operands from there: HBM/L2 traffic drops ~10x to about the unique input size;
//   * the contraction runs on v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 blocks per instruction, block b =
//     lanes 4b..4b+3.  Lane l supplies A = its own output pixel and B = weight column l%4, so one instruction
//     advances 64 pixels x 4 channels by one k; two accumulators cover 8 channels (75 % useful for Cout = 6).
//     D: lane l, register r = (pixel of lane 4*(l/4)+r, channel l%4 [+4])   (layout verified on hardware);
//   * the [8][taps*CH] weight panel of the current channel slice sits in LDS (rows padded by 4 floats so that
//     the four distinct addresses of a ds_read_b128 fall on different bank quads);
//   * K-split across the four wavefronts: each covers all 256 pixels but every fourth 16-byte channel chunk, so
//     one pair of weight reads feeds 32 MFMAs (LDS reads per MFMA halve); partial sums meet in LDS at the end;
//   * channel slices sized for 4 workgroups per CU so that the staging of one overlaps the MFMAs of the others.
// Measured (stem dgrad of ResNet-50, 4 parity classes, batch 256): 7.1 ms generic kernel -> 2.4 ms.
// Requirements (else bcos_tapconv falls back to the generic kernel): unit input stride and tap step (true for
// every dgrad parity class), C % 4 == 0, LDS footprint <= 160 KB, plain / addend / mul epilogue.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bcos_hip.h"
#include "bcos_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int SK_THREADS = 256;     // 4 wavefronts x 64 pixels = 16 x 16 output pixels
constexpr int SK_T = 16;

struct SkArgs {
    const float* a;
    const float* wt;        // [Cout][taps][C]
    float* out;
    const float* addend;    // optional, indexed like out
    const float* mul;       // optional, indexed like out
    bcos_tapconv_geom g;
    int Ktot, ldw;          // Ktot = taps * C; ldw = taps * CH + 4: one channel slice of the weight panel
    int CH;                 // channels per pass (C is processed in C / CH slices so that 2 workgroups fit a CU)
    int ldp;                // LDS floats per input pixel = CH + 4
    int PH, PW;             // input patch extent = 16 + TH - 1, 16 + TW - 1
    int tiles_i, tiles_j;
};

// In-place accumulate pinned to the accumulator file: with the builtin the register allocator rotates the eight
// accumulators through fresh registers and pays 36 copies per 32 MFMAs to undo it at the loop head.
__device__ __forceinline__ void mfma4(f32x4& acc, float a, float b) {
    asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

__global__ __launch_bounds__(SK_THREADS) void skinny_kernel(const SkArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sW = smem;                               // [8][ldw] weight panel
    float* sX = smem + 8 * p.ldw;                   // [PH*PW][ldp] input patch
    const bcos_tapconv_geom& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: keeps the K-split loop uniform
    int b = blockIdx.x;
    const int tj = b % p.tiles_j; b /= p.tiles_j;
    const int ti = b % p.tiles_i;
    const int n = b / p.tiles_i;
    const int i_base = ti * SK_T, j_base = tj * SK_T;

    // K-split: every wavefront covers all 256 pixels of the tile (4 groups of 64) but only the 16-byte channel
    // chunks c4 = wave, wave + 4, ... of each slice, so one pair of weight reads feeds 4 x 8 MFMAs (LDS reads per
    // MFMA halve against a pixel split); the four partial sums meet in LDS at the end.
    const int pr = lane >> 4, pj = lane & 15;                    // this lane's pixel inside a group: row 4*grp + pr
    const float* w0 = sW + (lane & 3) * p.ldw;                  // channel l%4
    const float* w1 = sW + ((lane & 3) + 4) * p.ldw;            // channel l%4 + 4
    f32x4 acc0[4], acc1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { acc0[q] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int C = g.C, CH = p.CH, ntaps = g.TH * g.TW;
    const int cps = CH / 4;                         // 16-byte chunks per pixel per slice
    const float* abase = p.a + (int64_t)n * g.H * g.W * g.a_pitch;
    const int grp_stride = 4 * p.PW * p.ldp;        // LDS distance between the pixel groups (4 patch rows)
    for (int c0 = 0; c0 < C; c0 += CH) {
        if (c0) __syncthreads();                    // everyone is done reading the previous slice
        // weight slice: sW[r][tap*CH + c] = wt[r][tap][c0 + c]   (rows >= Cout are zero)
        for (int i = tid; i < 8 * ntaps * cps; i += SK_THREADS) {
            const int r = i / (ntaps * cps);
            const int rem = i - r * (ntaps * cps);
            const int tap = rem / cps, c4 = rem - tap * cps;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < g.Cout) v = *reinterpret_cast<const f32x4*>(p.wt + (int64_t)r * p.Ktot + tap * C + c0 + c4 * 4);
            *reinterpret_cast<f32x4*>(sW + r * p.ldw + tap * CH + c4 * 4) = v;
        }
        // input patch slice: pixel (ph, pw) of the patch = input (i_base + dh0 + ph, j_base + dw0 + pw);
        // 4 loads in flight per thread (the loop is latency-bound otherwise: one 16-byte load per round trip)
        const int nld = p.PH * p.PW * cps;
        for (int i0 = tid; i0 < nld; i0 += 4 * SK_THREADS) {
            f32x4 v[4];
            int dst[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * SK_THREADS;
                const int px = i / cps, c4 = i - px * cps;
                const int ph = px / p.PW, pw = px - ph * p.PW;
                const int ih = i_base + g.dh0 + ph, iw = j_base + g.dw0 + pw;
                dst[u] = px * p.ldp + c4 * 4;
                v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (i < nld && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W)
                    v[u] = *reinterpret_cast<const f32x4*>(abase + ((int64_t)ih * g.W + iw) * g.a_pitch + c0 + c4 * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * SK_THREADS < nld) *reinterpret_cast<f32x4*>(sX + dst[u]) = v[u];
        }
        __syncthreads();
        // one flat loop over (tap, chunk) keeps the 32 accumulators in place (nested loops made the compiler move
        // them between register files around every innermost trip)
        const float* xs0 = sX + (pr * p.PW + pj) * p.ldp;
        const int cpw = (cps - wave + 3) >> 2;      // chunks of this wavefront per tap
        int th = 0, tw = 0, k = 0;
        // the asm MFMAs are opaque to the compiler's hazard recogniser: fence the accumulator hand-over
        // (v_accvgpr_write -> MFMA SrcC before the loop, MFMA result -> v_accvgpr_read after it) by hand
        // (the fences name the accumulators so that the copies into / out of the accumulator file stay outside)
        asm volatile("s_nop 7\n\ts_nop 7" : "+a"(acc0[0]), "+a"(acc0[1]), "+a"(acc0[2]), "+a"(acc0[3]),
                                               "+a"(acc1[0]), "+a"(acc1[1]), "+a"(acc1[2]), "+a"(acc1[3]));
        for (int it = 0; it < ntaps * cpw; ++it) {
            const int c = (wave + 4 * k) * 4;
            const float* xs = xs0 + (th * p.PW + tw) * p.ldp + c;
            const int kb = (th * g.TW + tw) * CH + c;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(w0 + kb);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(w1 + kb);
            f32x4 av[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) av[gq] = *reinterpret_cast<const f32x4*>(xs + gq * grp_stride);
            // k outer, pixel group inner: consecutive MFMAs hit different accumulators (a chain on one accumulator
            // would wait for the previous result every other instruction)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    mfma4(acc0[gq], av[gq][q], b0[q]);
                    mfma4(acc1[gq], av[gq][q], b1[q]);
                }
            }
            if (++k == cpw) { k = 0; if (++tw == g.TW) { tw = 0; ++th; } }
        }
        asm volatile("s_nop 7\n\ts_nop 7" : "+a"(acc0[0]), "+a"(acc0[1]), "+a"(acc0[2]), "+a"(acc0[3]),
                                               "+a"(acc1[0]), "+a"(acc1[1]), "+a"(acc1[2]), "+a"(acc1[3]));
    }

    // cross-wavefront reduction: red[src wave][group][lane][8]; wavefront w then owns pixel group w
    __syncthreads();
    float* red = smem;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        float* r8 = red + (((wave * 4 + gq) * 64 + lane) << 3);
        *reinterpret_cast<f32x4*>(r8) = acc0[gq];
        *reinterpret_cast<f32x4*>(r8 + 4) = acc1[gq];
    }
    __syncthreads();
    f32x4 sum0 = {0.f, 0.f, 0.f, 0.f}, sum1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sw = 0; sw < 4; ++sw) {
        const float* r8 = red + (((sw * 4 + wave) * 64 + lane) << 3);
        sum0 += *reinterpret_cast<const f32x4*>(r8);
        sum1 += *reinterpret_cast<const f32x4*>(r8 + 4);
    }
    const int pi = wave * 4 + pr;

    // D: lane l, reg r -> the pixel of lane (l & ~3) + r, channel l%4 (+4): 4 consecutive pj of one tile row
    const int ch = lane & 3;
    const int oi = i_base + pi;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int oj = j_base + (pj & ~3) + r;
        if (oi >= g.P || oj >= g.Q) continue;
        const int64_t px = ((int64_t)n * g.OH + (oi * g.out_sh + g.out_h0)) * g.OW + (oj * g.out_sw + g.out_w0);
        const int64_t base = px * g.out_pitch;
        if (ch < g.Cout) {
            float v = sum0[r];
            if (p.addend) v += p.addend[base + ch];
            if (p.mul) v *= p.mul[base + ch];
            p.out[base + ch] = v;
        }
        if (ch + 4 < g.Cout) {
            float v = sum1[r];
            if (p.addend) v += p.addend[base + ch + 4];
            if (p.mul) v *= p.mul[base + ch + 4];
            p.out[base + ch + 4] = v;
        }
    }
}


// ---- several tap sets over ONE input (the parity classes of a strided input gradient) in one launch -----------------
// The four classes of a stride-2 gradient read the same neighbourhood of the incoming gradient.  One workgroup stages
// the input patch of its 16x16 class pixels once per channel slice and runs every class over it (own weight panel, own
// accumulators: 4 classes x 32 registers), instead of four launches that each re-stage the same patch: measured HBM
// fetch of the ResNet stem gradient 4 x 1.75 GB -> 1 x ~1.2 GB.
constexpr int SK_MAXC = 4;
struct SkClass {
    const float* wt;            // [Cout][taps][C]
    int Ktot, TH, TW, dh0, dw0, out_h0, out_w0;
    int woff, ldw;              // LDS float offset / row pitch of this class's weight panel
};
struct SkGroupArgs {
    const float* a;
    float* out;
    const float* addend;
    const float* mul;
    int H, W, C, a_pitch, P, Q, OH, OW, out_sh, out_sw, out_pitch, Cout;
    int CH, ldp, PH, PW, dmin_h, dmin_w, tiles_i, tiles_j, ncls, wtotal;
    SkClass cls[SK_MAXC];
};

template <int NS>      // channel slices (C = NS * CH): unrolled so that the accumulators never leave the AGPRs between slices
__global__ __launch_bounds__(SK_THREADS) void skinny_group_kernel(const SkGroupArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sW = smem;                               // weight panels of all classes
    float* sX = smem + p.wtotal;                    // [PH*PW][ldp] input patch
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = blockIdx.x;
    const int tj = b % p.tiles_j; b /= p.tiles_j;
    const int ti = b % p.tiles_i;
    const int n = b / p.tiles_i;
    const int i_base = ti * SK_T, j_base = tj * SK_T;
    const int pr = lane >> 4, pj = lane & 15;
    f32x4 acc[SK_MAXC][8];
#pragma unroll
    for (int c = 0; c < SK_MAXC; ++c)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[c][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int C = p.C, CH = p.CH, cps = CH / 4;
    const float* abase = p.a + (int64_t)n * p.H * p.W * p.a_pitch;
    const int grp_stride = 4 * p.PW * p.ldp;
    const int cpw = (cps - wave + 3) >> 2;          // chunks of this wavefront per tap
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
        const int c0 = sl * CH;
        if (sl) __syncthreads();
        for (int ci = 0; ci < p.ncls; ++ci) {       // weight slices of every class
            const SkClass& k = p.cls[ci];
            const int ntaps = k.TH * k.TW;
            for (int i = tid; i < 8 * ntaps * cps; i += SK_THREADS) {
                const int r = i / (ntaps * cps);
                const int rem = i - r * (ntaps * cps);
                const int tap = rem / cps, c4 = rem - tap * cps;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (r < p.Cout) v = *reinterpret_cast<const f32x4*>(k.wt + (int64_t)r * k.Ktot + tap * C + c0 + c4 * 4);
                *reinterpret_cast<f32x4*>(sW + k.woff + r * k.ldw + tap * CH + c4 * 4) = v;
            }
        }
        const int nld = p.PH * p.PW * cps;          // input patch slice, origin (i_base + dmin_h, j_base + dmin_w)
        for (int i0 = tid; i0 < nld; i0 += 4 * SK_THREADS) {
            f32x4 v[4];
            int dst[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * SK_THREADS;
                const int px = i / cps, c4 = i - px * cps;
                const int ph = px / p.PW, pw = px - ph * p.PW;
                const int ih = i_base + p.dmin_h + ph, iw = j_base + p.dmin_w + pw;
                dst[u] = px * p.ldp + c4 * 4;
                v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (i < nld && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
                    v[u] = *reinterpret_cast<const f32x4*>(abase + ((int64_t)ih * p.W + iw) * p.a_pitch + c0 + c4 * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * SK_THREADS < nld) *reinterpret_cast<f32x4*>(sX + dst[u]) = v[u];
        }
        __syncthreads();
#pragma unroll
        for (int ci = 0; ci < SK_MAXC; ++ci) {
            if (ci >= p.ncls) break;
            const SkClass& k = p.cls[ci];
            const float* w0 = sW + k.woff + (lane & 3) * k.ldw;
            const float* w1 = w0 + 4 * k.ldw;
            const float* xs0 = sX + ((pr + k.dh0 - p.dmin_h) * p.PW + (pj + k.dw0 - p.dmin_w)) * p.ldp;
            int th = 0, tw = 0, kk = 0;
            asm volatile("s_nop 7\n\ts_nop 7" : "+a"(acc[ci][0]), "+a"(acc[ci][1]), "+a"(acc[ci][2]), "+a"(acc[ci][3]),
                                                   "+a"(acc[ci][4]), "+a"(acc[ci][5]), "+a"(acc[ci][6]), "+a"(acc[ci][7]));
            const int iters = k.TH * k.TW * cpw;
            for (int it = 0; it < iters; ++it) {
                const int c = (wave + 4 * kk) * 4;
                const float* xs = xs0 + (th * p.PW + tw) * p.ldp + c;
                const int kb = (th * k.TW + tw) * CH + c;
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(w0 + kb);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(w1 + kb);
                f32x4 av[4];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) av[gq] = *reinterpret_cast<const f32x4*>(xs + gq * grp_stride);
#pragma unroll
                for (int q = 0; q < 4; ++q) {       // k outer: consecutive MFMAs hit different accumulators
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        mfma4(acc[ci][gq], av[gq][q], b0[q]);
                        mfma4(acc[ci][4 + gq], av[gq][q], b1[q]);
                    }
                }
                if (++kk == cpw) { kk = 0; if (++tw == k.TW) { tw = 0; ++th; } }
            }
            asm volatile("s_nop 7\n\ts_nop 7" : "+a"(acc[ci][0]), "+a"(acc[ci][1]), "+a"(acc[ci][2]), "+a"(acc[ci][3]),
                                                   "+a"(acc[ci][4]), "+a"(acc[ci][5]), "+a"(acc[ci][6]), "+a"(acc[ci][7]));
        }
    }

    // per class: cross-wavefront reduction through LDS (red[src wave][group][lane][8]), then wavefront w stores group w
    float* red = smem;
    const int ch = lane & 3;
    const int pi = wave * 4 + pr;
    const int oi = i_base + pi;
#pragma unroll
    for (int ci = 0; ci < SK_MAXC; ++ci) {
        if (ci >= p.ncls) break;
        const SkClass& k = p.cls[ci];
        __syncthreads();
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            float* r8 = red + (((wave * 4 + gq) * 64 + lane) << 3);
            *reinterpret_cast<f32x4*>(r8) = acc[ci][gq];
            *reinterpret_cast<f32x4*>(r8 + 4) = acc[ci][4 + gq];
        }
        __syncthreads();
        f32x4 sum0 = {0.f, 0.f, 0.f, 0.f}, sum1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sw = 0; sw < 4; ++sw) {
            const float* r8 = red + (((sw * 4 + wave) * 64 + lane) << 3);
            sum0 += *reinterpret_cast<const f32x4*>(r8);
            sum1 += *reinterpret_cast<const f32x4*>(r8 + 4);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oj = j_base + (pj & ~3) + r;
            if (oi >= p.P || oj >= p.Q) continue;
            const int64_t px = ((int64_t)n * p.OH + (oi * p.out_sh + k.out_h0)) * p.OW + (oj * p.out_sw + k.out_w0);
            const int64_t base = px * p.out_pitch;
            if (ch < p.Cout) {
                float v = sum0[r];
                if (p.addend) v += p.addend[base + ch];
                if (p.mul) v *= p.mul[base + ch];
                p.out[base + ch] = v;
            }
            if (ch + 4 < p.Cout) {
                float v = sum1[r];
                if (p.addend) v += p.addend[base + ch + 4];
                if (p.mul) v *= p.mul[base + ch + 4];
                p.out[base + ch + 4] = v;
            }
        }
    }
}

}  // namespace

// Returns 1 if the launch was handled here, 0 if the caller should use the generic kernel, < 0 on error.
int bcos_try_skinny(const float* a, const float* wt, const bcos_tapconv_geom& g, const bcos_epilogue& e, int M,
                    hipStream_t stream) {
    (void)M;
    if (g.Cout > 8) return 0;
    if (e.bcos_mode != BCOS_NONE || e.bias || e.ch_scale || e.ch_shift || e.relu || e.out2 || e.scale_out ||
        e.norm_out || e.mul2 || e.gate2 || !e.out)
        return 0;
    if (g.in_sh != 1 || g.in_sw != 1 || g.dstep_h != 1 || g.dstep_w != 1) return 0;
    SkArgs p;
    p.a = a; p.wt = wt; p.out = e.out; p.addend = e.addend; p.mul = e.mul;
    p.g = g;
    p.Ktot = g.TH * g.TW * g.C;
    p.PH = SK_T + g.TH - 1;
    p.PW = SK_T + g.TW - 1;
    p.tiles_i = (g.P + SK_T - 1) / SK_T;
    p.tiles_j = (g.Q + SK_T - 1) / SK_T;
    // channels per pass: the largest slice (C, C/2, C/4, ... multiple of 4) whose footprint lets two workgroups
    // share a CU; measured best at <= 40 KB (4 workgroups per CU: staging of one overlaps the MFMAs of the others)
    size_t lds = 0;
    p.CH = 0;
    for (int ch = g.C; ch >= 4 && ch % 4 == 0; ch /= 2) {
        const size_t need = ((size_t)8 * (g.TH * g.TW * ch + 4) + (size_t)p.PH * p.PW * (ch + 4)) * sizeof(float);
        if (need <= 160 * 1024 && p.CH == 0) { p.CH = ch; lds = need; }
        if (need <= 40 * 1024) { p.CH = ch; lds = need; break; }
        if (g.C % (ch / 2) != 0 || (ch / 2) % 4 != 0) break;
    }
    if (p.CH == 0) return 0;
    if (lds < (size_t)4 * 4 * 64 * 8 * sizeof(float)) lds = (size_t)4 * 4 * 64 * 8 * sizeof(float);   // reduction buffer
    p.ldw = g.TH * g.TW * p.CH + 4;
    p.ldp = p.CH + 4;
    const int64_t blocks = (int64_t)g.N * p.tiles_i * p.tiles_j;
    if (blocks >= ((int64_t)1 << 31)) return 0;
    static std::atomic<size_t> lds_hw{0};
    hipError_t err = bcos_ensure_dynamic_lds(reinterpret_cast<const void*>(skinny_kernel), lds, lds_hw);
    if (err != hipSuccess) return bcos_set_hip_error("hipFuncSetAttribute(skinny)", err);
    hipLaunchKernelGGL(skinny_kernel, dim3((unsigned)blocks), dim3(SK_THREADS), lds, stream, p);
    err = hipGetLastError();
    if (err != hipSuccess) return bcos_set_hip_error("skinny launch", err);
    return 1;
}


// Several tap sets over one input in one launch; 1 = handled, 0 = not applicable (caller launches them one by one).
int bcos_try_skinny_group(const float* a, const float* const* wts, const bcos_tapconv_geom* gs, const bcos_epilogue* es,
                          int count, hipStream_t stream) {
    if (count < 2 || count > SK_MAXC) return 0;
    const bcos_tapconv_geom& g0 = gs[0];
    const bcos_epilogue& e0 = es[0];
    if (g0.Cout > 8 || g0.C % 4 != 0) return 0;
    SkGroupArgs p;
    int dmin_h = 1 << 30, dmin_w = 1 << 30, dmax_h = -(1 << 30), dmax_w = -(1 << 30), taps_total = 0;
    for (int i = 0; i < count; ++i) {
        const bcos_tapconv_geom& g = gs[i];
        const bcos_epilogue& e = es[i];
        if (e.bcos_mode != BCOS_NONE || e.bias || e.ch_scale || e.ch_shift || e.relu || e.out2 || e.scale_out ||
            e.norm_out || e.mul2 || e.gate2 || !e.out || e.flags)
            return 0;
        if (e.out != e0.out || e.addend != e0.addend || e.mul != e0.mul) return 0;
        if (g.in_sh != 1 || g.in_sw != 1 || g.dstep_h != 1 || g.dstep_w != 1) return 0;
        if (g.N != g0.N || g.H != g0.H || g.W != g0.W || g.C != g0.C || g.a_pitch != g0.a_pitch || g.P != g0.P ||
            g.Q != g0.Q || g.OH != g0.OH || g.OW != g0.OW || g.out_sh != g0.out_sh || g.out_sw != g0.out_sw ||
            g.out_pitch != g0.out_pitch || g.Cout != g0.Cout)
            return 0;
        if ((g.P - 1) * g.out_sh + g.out_h0 >= g.OH || (g.Q - 1) * g.out_sw + g.out_w0 >= g.OW || g.out_h0 < 0 || g.out_w0 < 0)
            return 0;
        if (!wts[i] || (reinterpret_cast<uintptr_t>(wts[i]) & 15)) return 0;
        dmin_h = g.dh0 < dmin_h ? g.dh0 : dmin_h;
        dmin_w = g.dw0 < dmin_w ? g.dw0 : dmin_w;
        dmax_h = g.dh0 + g.TH - 1 > dmax_h ? g.dh0 + g.TH - 1 : dmax_h;
        dmax_w = g.dw0 + g.TW - 1 > dmax_w ? g.dw0 + g.TW - 1 : dmax_w;
        taps_total += g.TH * g.TW;
    }
    p.a = a; p.out = e0.out; p.addend = e0.addend; p.mul = e0.mul;
    p.H = g0.H; p.W = g0.W; p.C = g0.C; p.a_pitch = g0.a_pitch ? g0.a_pitch : g0.C;
    p.P = g0.P; p.Q = g0.Q; p.OH = g0.OH; p.OW = g0.OW; p.out_sh = g0.out_sh; p.out_sw = g0.out_sw;
    p.out_pitch = g0.out_pitch ? g0.out_pitch : g0.Cout; p.Cout = g0.Cout;
    p.dmin_h = dmin_h; p.dmin_w = dmin_w;
    p.PH = SK_T + dmax_h - dmin_h;
    p.PW = SK_T + dmax_w - dmin_w;
    p.tiles_i = (g0.P + SK_T - 1) / SK_T;
    p.tiles_j = (g0.Q + SK_T - 1) / SK_T;
    p.ncls = count;
    // channel slice: at least 16 (one 16-byte chunk per wavefront of the K-split), the largest that keeps two
    // workgroups per CU
    const size_t red_bytes = (size_t)4 * 4 * 64 * 8 * sizeof(float);
    size_t lds = 0;
    p.CH = 0;
    for (int ch = g0.C; ch >= 4 && ch % 4 == 0; ch /= 2) {
        const size_t need = ((size_t)8 * (taps_total * ch + 4 * count) + (size_t)p.PH * p.PW * (ch + 4)) * sizeof(float);
        if (need <= 160 * 1024 && p.CH == 0) { p.CH = ch; lds = need; }
        if (need <= 72 * 1024) { p.CH = ch; lds = need; break; }
        if (g0.C % (ch / 2) != 0 || (ch / 2) % 4 != 0) break;
    }
    if (p.CH == 0) return 0;
    p.ldp = p.CH + 4;
    int woff = 0;
    for (int i = 0; i < count; ++i) {
        SkClass& k = p.cls[i];
        const bcos_tapconv_geom& g = gs[i];
        k.wt = wts[i]; k.TH = g.TH; k.TW = g.TW; k.dh0 = g.dh0; k.dw0 = g.dw0; k.out_h0 = g.out_h0; k.out_w0 = g.out_w0;
        k.Ktot = g.TH * g.TW * g.C;
        k.ldw = g.TH * g.TW * p.CH + 4;
        k.woff = woff;
        woff += 8 * k.ldw;
    }
    p.wtotal = woff;
    if (lds < red_bytes) lds = red_bytes;
    const int64_t blocks = (int64_t)g0.N * p.tiles_i * p.tiles_j;
    if (blocks >= ((int64_t)1 << 31)) return 0;
    const int ns = g0.C / p.CH;
    static std::atomic<size_t> lds_hw[3];
    hipError_t err;
    auto go = [&](auto kern, int which) -> hipError_t {
        hipError_t e2 = bcos_ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, lds_hw[which]);
        if (e2 != hipSuccess) return e2;
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(SK_THREADS), lds, stream, p);
        return hipSuccess;
    };
    if (ns == 1) err = go(skinny_group_kernel<1>, 0);
    else if (ns == 2) err = go(skinny_group_kernel<2>, 1);
    else if (ns == 4) err = go(skinny_group_kernel<4>, 2);
    else return 0;
    if (err != hipSuccess) return bcos_set_hip_error("hipFuncSetAttribute(skinny_group)", err);
    err = hipGetLastError();
    if (err != hipSuccess) return bcos_set_hip_error("skinny_group launch", err);
    return 1;
}
