// bcos_tapconv.hip -- fused implicit-GEMM "tap convolution" for gfx950 (MI355X, CDNA4).
//
// One kernel family serves BcosConv2d / BcosifyConv2d forward (reference
// bcos/modules/bcosconv2d.py:153-231, bcosifyconv2d.py:50-102), BcosLinear / BcosifyLinear
// forward (bcoslinear.py:88-130, bcosifylinear.py:42-95) and the explanation-mode input
// gradients (the autograd convolution_backward behind bcos/common.py:177).
//
// Structure (written for CDNA4, not translated from anything):
//   * GEMM view: rows m = output pixels (n,i,j), cols = Cout, K = taps x Cin, NHWC activations
//     and [Cout][taps][Cin] weights so that both operands are K-contiguous.
//   * 256 threads = 4 wavefronts of 64; block tile BM x BN x 32, each wave owns TM x TN
//     tiles of 32x32 computed with v_mfma_f32_32x32x2_f32 (exact fp32, 16 acc VGPRs per tile).
//   * operands are staged global -> VGPR -> LDS (padded rows, 36 floats, conflict-free
//     ds_read_b128) so that out-of-image taps are zero-filled in registers; the loads for
//     K-step k+1 are issued before the MFMAs of step k and written to the other LDS
//     buffer after them (one barrier per K-step).
//   * each lane feeds 4 MFMAs from one ds_read_b128 per operand: lanes 0-31 hold
//     k = 8q..8q+3 and lanes 32-63 k = 8q+4..8q+7 of their row, so MFMA c consumes the
//     k-pair (8q+c, 8q+4+c) -- a permutation of K applied identically to A and B.
//   * the per-row patch norm sum_k A[m,k]^2 is accumulated from the A fragments already in
//     registers (8 FMAs per 16 MFMAs) and finished with one cross-half shuffle.
//   * the fused epilogue (include/bcos_hip.h: bcos_epilogue) runs on the accumulators.
//   * block -> tile mapping is XCD-aware: the 8 XCDs (private L2s) each get a contiguous
//     range of tiles, n-tile fastest, so blocks sharing an A row-panel share an L2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include <type_traits>
#include "bcos_hip.h"
#include "bcos_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));


namespace {

// process-wide DEFAULT for calls that do not choose (bcos_operands.contraction == 0): 0 = v_mfma_f32_32x32x2_f32,
// 1 = 6 x v_mfma_f32_32x32x16_bf16 on exact 3-way splits, 2 (default) = 3 x v_mfma_f32_32x32x16_f16 on scaled 2-way splits
// where the call provides what that needs (a_absmax + wt_f16x2), else 1
std::atomic<int> g_contraction_mode{2};

#ifndef BCOS_PHASE_TIMING
#define BCOS_PHASE_TIMING 0       // development builds: per-workgroup clock of prologue / K loop / epilogue of tile_body_d, summed in g_phase
#endif
#if BCOS_PHASE_TIMING
__device__ unsigned long long g_phase[8];          // [0..2] clock sums of prologue, loop, epilogue; [3] workgroups; [4] launch-wide first / [5] last clock;
                                                   // [6] / [7] inside the epilogue: row records + accumulators -> LDS of part 0, the stores of part 0
#define BCOS_PHASE_MARK(var) const unsigned long long var = wall_clock64()
#else
#define BCOS_PHASE_MARK(var)
#endif
constexpr int BK = 32;            // K floats per step
constexpr int LDS_LD = BK + 4;    // padded LDS row (floats): 144 B, 16 rows hit 16 distinct 16-B slots
constexpr int NTHREADS = 256;
constexpr int NXCD = 8;
constexpr int EPI_ROWS = 128;     // rows of a tile drained per epilogue part (LDS transpose buffer = EPI_ROWS x 132 floats)

struct KArgs {
    const float* a;
    const float* wt;
    bcos_tapconv_geom g;
    bcos_epilogue e;
    int M;          // N*P*Q
    int PQ;
    int Ktot;       // taps*C
    int cpt;        // 16-byte chunks per tap = C/4
    int nchunks;    // Ktot/4
    int nk;         // K steps
    int tiles_n;            // column tiles
    int n_big, n_small;     // tiles of BM rows, then tiles of BM/2 rows (tail of the launch)
    int rows_big;           // rows covered by the BM-row tiles
    const void* wt3;              // optional pre-split weights in MFMA fragment order (bcos_split_weights), else NULL
    unsigned wt3_bytes;
    unsigned a_bytes, wt_bytes;   // operand sizes for the buffer descriptors of the split-bf16 path (< 2 GiB there)
    const unsigned* a_absmax;     // split-f16 path: per-pixel max |A| bit patterns (the operand scale source)
    unsigned absmax_bytes;
    const unsigned* a_imgmax;     // ... per-image maxima of those (input-patch loop, tile_body_p), else NULL
    const unsigned* a_imgmin;     // ... per-image minima over the NONZERO pixels of those (dynamic range inside an image), else NULL
    const unsigned* a_imgmin_c;   // ... or a LOWER BOUND of those minima, stored complemented (~v; 0 = no nonzero pixel), as the epilogue of the launch that
                                  //     produced A leaves it (bcos_epilogue.out_imgmin_c): decides only whether a tile runs the level bookkeeping, never a bit
    int lvl_on;                   // input-patch loop: 1 = one pass per operand-scale level present in a tile (BCOS_OPT_PATCH_LEVELS)
    int lvl_off;                  // input-patch loop: byte offset, inside the dynamic LDS, of the row-level table (beyond everything the epilogue uses)
    const void* wt2;              // split-f16 path: pre-split, pre-scaled weights in MFMA fragment order (bcos_split_weights_f16x2)
    unsigned wt2_bytes;
    const float* wt2_cinv;        // ... their inverse column scales [padded Cout]
    int h2;          // contraction on split-f16 MFMA (see tile_body_h2)
    int x3;          // contraction on split-bf16 MFMA (see tile_body_x3) instead of fp32 MFMA
    int uniform_tap; // C % 32 == 0: every K-step lies inside one tap
    int vec_ok;     // every per-element epilogue tensor is 16-byte addressable (pitch % 4 == 0, aligned bases)
    int epi_kind;   // 0 = general epilogue, k + 1 = specialisation k of EPI_KINDS_FWD (norm launches) / EPI_KINDS_BWD (tile_epilogue_fast)
    unsigned out_bytes;   // size of the [N*OH*OW][out_pitch] epilogue tensors (epi_kind > 0: < 2 GiB)
    // 2-D row tiles (input-patch loop on wide images, tile_body_p<..., T2D>): tile t = (image t / t2_nb, block t % t2_nb) and row r of a
    // tile is position (r / t2_bw, r % t2_bw) of that t2_bh x t2_bw block of the row grid; 0 = rows are m = n P Q + i Q + j
    int t2_bw, t2_nbx, t2_nb, t2_rows;
};

// row r of the tile at m0 -> (image, grid position), false: the tile has no such row (shared by the epilogues)
template <typename P>
__device__ __forceinline__ bool tile_row_nij(const P& p, int m0, int r, int& n, int& i, int& jj) {
    if (p.t2_bw > 0) {
        const int t = m0 / p.t2_rows;
        n = t / p.t2_nb;
        const int b = t - n * p.t2_nb;
        const int by = b / p.t2_nbx;
        const int ri = r / p.t2_bw;
        i = by * (p.t2_rows / p.t2_bw) + ri;
        jj = (b - by * p.t2_nbx) * p.t2_bw + (r - ri * p.t2_bw);
        return i < p.g.P && jj < p.g.Q;
    }
    const int m = m0 + r;
    if (m >= p.M) return false;
    n = m / p.PQ;
    const int rem = m - n * p.PQ;
    i = rem / p.g.Q;
    jj = rem - i * p.g.Q;
    return true;
}

// max over groups of G consecutive lanes (G = 8, 16, 32; groups aligned to G) on the vector ALU (DPP), no LDS traffic.
// After the call the lanes with (lane % G) == group_max_lane<G>() hold the group's max.
template <int G>
__device__ __forceinline__ unsigned group_max_u32(unsigned v) {
    static_assert(G == 8 || G == 16 || G == 32, "group size");
    auto dpp = [](unsigned x, auto ctrl) { return (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, decltype(ctrl)::value, 0xF, 0xF, false); };
    v = max(v, dpp(v, std::integral_constant<int, 0xB1>{}));       // quad_perm [1,0,3,2]
    v = max(v, dpp(v, std::integral_constant<int, 0x4E>{}));       // quad_perm [2,3,0,1]
    v = max(v, dpp(v, std::integral_constant<int, 0x141>{}));      // row_half_mirror: 8 lanes agree
    if (G >= 16) v = max(v, dpp(v, std::integral_constant<int, 0x140>{}));   // row_mirror: 16 lanes agree
    if (G == 32)                                                    // row_bcast15 into rows 1 and 3: lanes 16..31 of each half
        v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xA, 0xF, false));
    return v;
}
template <int G> constexpr int group_max_lane() { return G == 32 ? 16 : 0; }

// column parts of an epilogue: 192 -> 3 x 64, 256 -> 2 x 128; EPI_SPLIT_128 (development switch): 128 -> 2 x 64, which halves the
// transpose buffer (35 instead of 68 KB) so that three 128 x 128 workgroups fit a CU's LDS
#ifndef EPI_SPLIT_128
#define EPI_SPLIT_128 0
#endif
#ifndef D_WGS3
#define D_WGS3 0                  // 1 = the 128 x 128 LDS-DMA kernels are compiled for three resident workgroups per CU (needs EPI_SPLIT_128)
#endif
template <int BN> constexpr int epi_pn() { return BN == 192 ? 3 : (BN > 128 ? BN / 128 : ((EPI_SPLIT_128 && BN == 128) ? 2 : 1)); }
// tile / part geometry of an epilogue (shared by the functions below)
// part (pm, pn) = accumulator tiles i in [pm*TM/PM, ...), j in [pn*TN/PN, ...) of EVERY wave, so each wave retires half
// of its accumulator registers per part; local row l of a part is tile row (l / HM) * WM + pm * HM + l % HM
#define BCOS_EPI_SHAPE                                                                                                     \
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;                                                                    \
    constexpr int TM = WM / 32, TN = WN / 32;                                                                              \
    constexpr int PM = (BM > EPI_ROWS && (BM / WAVES_M) % (32 * (BM / EPI_ROWS)) == 0) ? BM / EPI_ROWS : 1;                \
    constexpr int PN = epi_pn<BN>();                                                                                       \
    constexpr int SBM = BM / PM, SBN = BN / PN;                                                                            \
    constexpr int HM = WM / PM, HN = WN / PN;                                                                              \
    constexpr int TMP = TM / PM, TNP = TN / PN;                                                                            \
    static_assert(TM % PM == 0 && TN % PN == 0 && HN % 4 == 0, "parts split the wave tile evenly");                        \
    constexpr int LDC = SBN + 4;                                                                                           \
    const int tid = threadIdx.x;                                                                                           \
    const int lane = tid & 63;                                                                                             \
    const int wave = tid >> 6;                                                                                             \
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;                                                            \
    (void)TM; (void)TN; (void)SBM; (void)HM; (void)HN; (void)TMP; (void)TNP; (void)LDC; (void)lane; (void)wave_m; (void)wave_n;
// The epilogue reads the launch descriptor through a laundered pointer to the kernel-argument segment: its ~60 scalars
// (20 tensor pointers, pitches, flags) are then loaded HERE, by scalar loads, instead of at kernel entry.  Loaded at
// entry they are live across the whole main loop, the kernel exceeds the 102 scalar registers by > 100, and every
// use of a spilled pointer inside the per-element loop becomes a v_readlane on the vector ALU.
// (KArgs is the kernel's only explicit argument, so it sits at offset 0 of the segment.)
// `kpin`: NULL inside a kernel; the segment pointer handed down where the epilogue runs in an out-of-line function
// (tile_body_p_more) -- llvm.amdgcn.kernarg.segment.ptr is NULL outside kernels.
typedef const __attribute__((address_space(4))) struct KArgs* KArgsSegPtr;
#define BCOS_EPI_KARGS                                                                                                     \
    const __attribute__((address_space(4))) KArgs* kp =                                                                    \
        kpin ? kpin : (const __attribute__((address_space(4))) KArgs*)__builtin_amdgcn_kernarg_segment_ptr();              \
    asm volatile("" : "+s"(kp));                                                                                           \
    const __attribute__((address_space(4))) KArgs& p = *kp;                                                                \
    const auto& g = p.g;                                                                                                   \
    const auto& e = p.e;                                                                                                   \
    (void)g; (void)e;

// [BN] floats behind the epilogue's transpose buffer and row records: 1 / ||w_c|| of the tile's columns (BCOS_EPI_UNIT_NORM_W)
template <int BM, int BN, int WAVES_M>
__device__ __forceinline__ float* epi_col_table(float* smem) {
    constexpr int PM = (BM > EPI_ROWS && (BM / WAVES_M) % (32 * (BM / EPI_ROWS)) == 0) ? BM / EPI_ROWS : 1;
    constexpr int SBM = BM / PM, SBN = BN / epi_pn<BN>();
    return smem + SBM * (SBN + 4) + BM * 5;
}

// General epilogue (include/bcos_hip.h: bcos_epilogue), every feature decided at run time; shared by the fp32, split-bf16 and
// split-f16 main loops.  `ss` = per-lane partial row sums in MFMA fragment layout, or `ROWSS` = partial row sums in staging
// layout.  SCALED (split-f16 loop): accumulators carry the power-of-two operand scales; `AINV` (staging layout) holds the
// inverse row scales, p.wt2_cinv the inverse column scales.
//   epi_rows_generic: per-row metadata (output pixel, patch norm, inverse scales) into LDS behind the transpose buffer;
//   epi_part_generic: one part of the tile, LDS transpose buffer -> epilogue math -> tensors.
// `lvl` (input-patch loop on images of wide dynamic range, tile_body_p): NULL, or one byte per tile row -- only the rows whose byte
// equals `pass` belong to this pass of the tile; the others are dropped like rows beyond M
template <int BM, int BN, int WAVES_M, int WAVES_N, bool NORM, bool SCALED, int NT, bool ROWX = true>
__device__ __forceinline__ void epi_rows_generic(float* smem, const float* ss, const float* ROWSS, const float* AINV, const int m0,
                                                 const unsigned char* lvl, const int pass, KArgsSegPtr kpin) {
    BCOS_EPI_SHAPE
    BCOS_EPI_KARGS
    float* sC = smem;
    int64_t* sPix = reinterpret_cast<int64_t*>(smem + SBM * LDC);  // [BM] output pixel index or -1
    float* sNorm = reinterpret_cast<float*>(sPix + BM);            // [BM] patch norm
    float* sRinv = sNorm + BM;                                     // [BM] 1 / norm
    float* sAinv = sRinv + BM;                                     // [BM] inverse operand scale of the row (SCALED)
    for (int r = tid; r < BM; r += NT) {
        int64_t pix = -1, apix = -1;
        int n, i, jj;
        if (tile_row_nij(p, m0, r, n, i, jj) && (lvl == nullptr || lvl[r] == pass)) {
            pix = ((int64_t)n * g.OH + (i * g.out_sh + g.out_h0)) * g.OW + (jj * g.out_sw + g.out_w0);
            if (!NORM && e.addend_sub > 1) {     // subsampled addend: this row's pixel in [N, ceil(OH / s), ceil(OW / s)] or -1
                const int s = e.addend_sub;
                const int h = i * g.out_sh + g.out_h0, w = jj * g.out_sw + g.out_w0;
                apix = (h % s == 0 && w % s == 0) ? ((int64_t)n * ((g.OH + s - 1) / s) + h / s) * ((g.OW + s - 1) / s) + w / s : -1;
            }
        }
        sPix[r] = pix;
        if (!NORM && e.addend_sub > 1) reinterpret_cast<int64_t*>(sNorm)[r] = apix;
        if (ROWX && !SCALED && e.row_scale) sAinv[r] = pix >= 0 ? e.row_scale[pix] : 0.f;      // (bcos_epilogue.row_scale: rides in the row factor)
    }
    // output pixel of tile row r (bcos_epilogue.row_scale / a_sumsq are indexed by it), or -1
    auto row_pix = [&](int r) -> int64_t {
        int n, i, jj;
        if (!tile_row_nij(p, m0, r, n, i, jj)) return -1;
        return ((int64_t)n * g.OH + (i * g.out_sh + g.out_h0)) * g.OW + (jj * g.out_sw + g.out_w0);
    };
    if (NORM && ROWSS == nullptr) {
        // row sums of squares in MFMA fragment layout (fp32 kernel): lane (row, k-half), two halves per row
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float t = ss[i] + __shfl_xor(ss[i], 32);
            if (wave_n == 0 && lane < 32) {
                const int row = wave_m * WM + i * 32 + lane;
                if (ROWX && e.a_sumsq) { const int64_t px = row_pix(row); t = px >= 0 ? e.a_sumsq[px] : 0.f; }
                float nrm = e.bcos_mode == BCOS_LINEAR_EPS ? sqrtf(t) + 1e-12f : sqrtf(t + 1e-6f);
                sNorm[row] = nrm;
                sRinv[row] = 1.0f / nrm;
            }
        }
    } else if (NORM) {
        // row sums of squares in staging layout (split kernels): thread (r0 + RP j, chunk), 4 chunk-lanes per row
        constexpr int RP = NT / 4;
#pragma unroll
        for (int j = 0; j < BM / RP; ++j) {
            float t = ROWSS[j];
            t += __shfl_xor(t, 1);
            t += __shfl_xor(t, 2);
            if ((tid & 3) == 0) {
                const int row = (tid >> 2) + RP * j;
                if (ROWX && e.a_sumsq) { const int64_t px = row_pix(row); t = px >= 0 ? e.a_sumsq[px] : 0.f; }
                float nrm = e.bcos_mode == BCOS_LINEAR_EPS ? sqrtf(t) + 1e-12f : sqrtf(t + 1e-6f);
                sNorm[row] = nrm;
                sRinv[row] = 1.0f / nrm;
            }
        }
    }
    if (SCALED) {
#pragma unroll
        for (int j = 0; j < BM / (NT / 4); ++j)
            if ((tid & 3) == 0) {
                const int row = (tid >> 2) + (NT / 4) * j;
                float ai = AINV[j];
                if (ROWX && e.row_scale) { const int64_t px = row_pix(row); ai *= px >= 0 ? e.row_scale[px] : 0.f; }
                sAinv[row] = ai;
            }
    }

}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool NORM, bool SCALED, int NT>
__device__ __forceinline__ void epi_part_generic(float* smem, const int pm, const int pn, const int part, const int n0, const int tile_n,
                                                 KArgsSegPtr kpin) {
    BCOS_EPI_SHAPE
    BCOS_EPI_KARGS
    float* sC = smem;
    int64_t* sPix = reinterpret_cast<int64_t*>(smem + SBM * LDC);  // [BM] output pixel index or -1
    float* sNorm = reinterpret_cast<float*>(sPix + BM);            // [BM] patch norm
    float* sRinv = sNorm + BM;                                     // [BM] 1 / norm
    float* sAinv = sRinv + BM;                                     // [BM] inverse operand scale of the row (SCALED)
    const bool b_is_2 = e.b == 2.0f && !(e.flags & BCOS_EPI_FORCE_POW);
    const bool norm_only = (e.flags & BCOS_EPI_NORM_ONLY) != 0;
    const bool gate_lsb = (e.flags & BCOS_EPI_SCALE_GATE_LSB) != 0;
    const bool gate_mul = (e.flags & BCOS_EPI_GATE2_FROM_MUL) != 0 && e.mul != nullptr;
    const bool mul_from_act = (e.flags & BCOS_EPI_MUL_FROM_ACT) != 0 && e.mul != nullptr && e.mul_norm != nullptr;
    const bool want_max = e.out_absmax != nullptr || e.out2_absmax != nullptr;
    const float bm1 = e.b - 1.0f;
    const int Cout = ((int)blockIdx.y + 1) * g.Cout;      // end of this launch's (group's) column range; blockIdx.y = group
    constexpr int CPR = SBN / 4;             // 16-byte chunks per part row
    constexpr int RPP = NT / CPR;            // rows per pass
    constexpr int PASSES = SBM / RPP;
    constexpr int EPI_G = NT > 256 ? 2 : 4;  // chunks whose loads are in flight together (half the registers per wave at 8 waves)
    static_assert(PASSES % EPI_G == 0, "epilogue grouping");
    const int cq = tid % CPR;
    const int rbase = tid / CPR;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    if (NORM && e.norm_out != nullptr && tile_n == 0 && part == 0) {
        for (int r = tid; r < BM; r += NT) {
            const int64_t pix = sPix[r];
            if (pix >= 0) e.norm_out[pix * g.norm_pitch + blockIdx.y] = sNorm[r];
        }
    }
    const int col = n0 + ((cq * 4) / HN) * WN + pn * HN + (cq * 4) % HN;
    const bool vec = p.vec_ok && (col + 3 < Cout);     // whole chunk inside the tensor and 16-byte addressable
    // element offset of the column chunk inside its output row: the column itself, or -- depth-to-space launches
    // (bcos_tapconv_geom.out_cgroup) -- channel col % G of the pixel (dh * OW + dw) further on
    int coladd = col;
    if (g.out_cgroup > 0) {
        const int cls = col / g.out_cgroup;
        coladd = ((cls / g.out_sw) * g.OW + cls % g.out_sw) * g.out_pitch + (col - cls * g.out_cgroup);
    }
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, csc4 = {1.f, 1.f, 1.f, 1.f}, csh4 = {0.f, 0.f, 0.f, 0.f}, cinv4 = {1.f, 1.f, 1.f, 1.f};
    // col_scale (the unit-norm projection 1 / ||w_c|| of NormedConv2d / NormedLinear, folded into the contraction) multiplies
    // the accumulator ahead of everything else: it rides in the column factor the scaled loop applies anyway
    const bool col_pre = SCALED || e.col_scale != nullptr || (e.flags & BCOS_EPI_UNIT_NORM_W);
    const bool row_pre = !SCALED && e.row_scale != nullptr;      // (SCALED: the row factor is folded into sAinv)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = col + q < Cout ? col + q : 0;
        if (e.bias) bias4[q] = e.bias[c];
        if (e.ch_scale) csc4[q] = e.ch_scale[c];
        if (e.ch_shift) csh4[q] = e.ch_shift[c];
        if (SCALED) cinv4[q] = p.wt2_cinv[c];
        if (e.col_scale) cinv4[q] *= e.col_scale[c];
    }
    if (!SCALED && (e.flags & BCOS_EPI_UNIT_NORM_W)) {       // 1 / ||w_c|| gathered by the main loop (tile columns, local index)
        const float* sCol = epi_col_table<BM, BN, WAVES_M>(smem);
        const int lc = ((cq * 4) / HN) * WN + pn * HN + (cq * 4) % HN;
#pragma unroll
        for (int q = 0; q < 4; ++q) cinv4[q] *= sCol[lc + q < BN ? lc + q : 0];
    }
    f32x4 mcsc4 = {1.f, 1.f, 1.f, 1.f}, mcsh4 = {0.f, 0.f, 0.f, 0.f};
    if (mul_from_act) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = col + q < Cout ? col + q : 0;
            if (e.mul_csc) mcsc4[q] = e.mul_csc[c];
            if (e.mul_csh) mcsh4[q] = e.mul_csh[c];
        }
    }
    // `mul` given as the kept ACTIVATION a = relu(lin s csc + csh) of the layer below (B = 2, s = |lin| / norm) instead of its
    // stored multiplier t = s csc gate: |lin|^2 = |a - csh| norm / |csc|, so t = csc sqrt(|a - csh| / (|csc| norm)) where a > 0
    auto rebuild_t = [&](float a, float nrm, float csc, float csh) {
        const float z = fabsf(a - csh);
        const float den = fabsf(csc) * nrm;
        // hardware reciprocal / square root (1 ulp each): the IEEE division + sqrt sequences are ~20 instructions per
        // element in an epilogue that is issue-bound, and t enters a product whose other factor carries fp32 rounding anyway
        return (a > 0.f && den > 0.f) ? csc * __builtin_amdgcn_sqrtf(z * __builtin_amdgcn_rcpf(den)) : 0.f;
    };
    // subsampled addend (bcos_epilogue.addend_sub = s > 1, gradient launches only): the addend tensor holds the pixels
    // (h % s == 0, w % s == 0) alone; epi_rows_generic left each row's pixel index in it (or -1) where forward launches keep
    // their patch norms
    const int64_t* sApix = reinterpret_cast<const int64_t*>(sNorm);
    const bool asub = !NORM && e.addend != nullptr && e.addend_sub > 1;
    struct EpiIn { f32x4 ad[EPI_G], m1[EPI_G]; };
    auto issue = [&](int p0, EpiIn& in) {
#pragma unroll
        for (int u = 0; u < EPI_G; ++u) {
            const int lrow = rbase + (p0 + u) * RPP;
            const int64_t pix = sPix[(lrow / HM) * WM + pm * HM + lrow % HM];
            const int64_t idx = (pix >= 0 ? pix : 0) * g.out_pitch + coladd;
            int64_t aidx = idx;
            bool a_ok = true;
            if constexpr (!NORM) {
                if (asub) {
                    const int64_t ap = sApix[(lrow / HM) * WM + pm * HM + lrow % HM];
                    a_ok = ap >= 0;
                    aidx = (a_ok ? ap : 0) * g.out_pitch + coladd;
                }
            }
            in.ad[u] = (vec && e.addend && a_ok) ? *reinterpret_cast<const f32x4*>(e.addend + aidx) : zero4;
            in.m1[u] = (vec && e.mul) ? *reinterpret_cast<const f32x4*>(e.mul + idx) : zero4;
        }
    };
    auto process = [&](int p0, const EpiIn& in) {
        unsigned mx1[EPI_G], mx2[EPI_G];
        int64_t pixs[EPI_G];
#pragma unroll
        for (int u = 0; u < EPI_G; ++u) {
            const int lrow = rbase + (p0 + u) * RPP;
            mx1[u] = 0u; mx2[u] = 0u;
            pixs[u] = sPix[(lrow / HM) * WM + pm * HM + lrow % HM];
        }
        if (e.max_out > 1) {
            // MaxOut (bcosconv2d.py:166-170): the accumulator columns are the M filters of each output unit, adjacent;
            // a thread's 4 columns hold 4 / M whole units.  out is [pixels, Cout / M] (pitch out_pitch), scale_out keeps
            // the contraction's width (pitch Cout): the scale at the winning filter, 0 at the others -- d out / d lin.
            // (the host only takes this path for M in {2, 4}, Cout % 4 == 0 and plain forward epilogues)
            const int M_ = e.max_out;
#pragma unroll
            for (int u = 0; u < EPI_G; ++u) {
                const int lrow = rbase + (p0 + u) * RPP;
                const int row = (lrow / HM) * WM + pm * HM + lrow % HM;
                const int64_t pix = pixs[u];
                if (pix < 0 || col >= Cout) continue;
                f32x4 val = *reinterpret_cast<const f32x4*>(sC + lrow * LDC + cq * 4);
                if (SCALED) val = val * sAinv[row] * cinv4;
                else {
                    if (row_pre) val = val * sAinv[row];
                    if (col_pre) val = val * cinv4;
                }
                val += bias4;
                f32x4 tf = {0.f, 0.f, 0.f, 0.f};
                for (int u0 = 0; u0 < 4; u0 += M_) {
                    int arg = u0;
                    for (int q = u0 + 1; q < u0 + M_; ++q) arg = val[q] > val[arg] ? q : arg;      // first maximum wins
                    const float m = val[arg];
                    float sc = 1.f;
                    if (NORM && !norm_only)
                        sc = b_is_2 ? fabsf(m) * sRinv[row] : powf(fabsf(m / sNorm[row]) + 1e-6f, bm1);
                    if (e.out) e.out[pix * g.out_pitch + (col + u0) / M_] = m * sc;
                    tf[arg] = sc;
                }
                if (e.scale_out) *reinterpret_cast<f32x4*>(e.scale_out + pix * (int64_t)Cout + col) = tf;
            }
        } else if (vec) {
            f32x4 v[EPI_G], ad[EPI_G], m1[EPI_G], m2[EPI_G], g2[EPI_G], rg[EPI_G];
            int64_t idx[EPI_G];
            bool ok[EPI_G];
            float rinv[EPI_G], nrm[EPI_G];
#pragma unroll
            for (int u = 0; u < EPI_G; ++u) {
                const int lrow = rbase + (p0 + u) * RPP;
                const int row = (lrow / HM) * WM + pm * HM + lrow % HM;
                const int64_t pix = pixs[u];
                ok[u] = pix >= 0;
                idx[u] = (ok[u] ? pix : 0) * g.out_pitch + coladd;
                v[u] = *reinterpret_cast<const f32x4*>(sC + lrow * LDC + cq * 4);
                if (SCALED) v[u] = v[u] * sAinv[row] * cinv4;
                else {
                    if (row_pre) v[u] = v[u] * sAinv[row];
                    if (col_pre) v[u] = v[u] * cinv4;
                }
                rinv[u] = NORM ? sRinv[row] : 1.f;
                nrm[u] = NORM ? sNorm[row] : 1.f;
                ad[u] = in.ad[u];
                rg[u] = e.relu_gate ? *reinterpret_cast<const f32x4*>(e.relu_gate + idx[u]) : zero4;
                m1[u] = in.m1[u];
                if (mul_from_act) {
                    const float mn = e.mul_norm[ok[u] ? pix : 0];
#pragma unroll
                    for (int q = 0; q < 4; ++q) m1[u][q] = rebuild_t(m1[u][q], mn, mcsc4[q], mcsh4[q]);
                }
                m2[u] = e.mul2 ? *reinterpret_cast<const f32x4*>(e.mul2 + idx[u]) : zero4;
                g2[u] = e.gate2 ? *reinterpret_cast<const f32x4*>(e.gate2 + idx[u]) : zero4;
            }
#pragma unroll
            for (int u = 0; u < EPI_G; ++u) {
                f32x4 val = v[u] + bias4;
                f32x4 s = {1.f, 1.f, 1.f, 1.f};
                if (NORM && !norm_only) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        s[q] = b_is_2 ? fabsf(val[q]) * rinv[u] : powf(fabsf(val[q] / nrm[u]) + 1e-6f, bm1);
                    val *= s;
                }
                val = val * csc4 + csh4;
                s *= csc4;
                if (e.addend) val += ad[u];
                if (e.relu == 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool open_gate = e.relu_gate ? rg[u][q] > 0.f : val[q] > 0.f;
                        s[q] = open_gate ? (gate_lsb ? __uint_as_float(__float_as_uint(s[q]) | 1u) : s[q]) : 0.f;
                        val[q] = open_gate ? val[q] : 0.f;
                    }
                } else if (e.relu == 2) {     // GELU with the gate treated as a constant (MyGELU, bcosify_vit.py:27-32)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float gate = bcos_gelu_gate(val[q]);
                        s[q] *= gate;
                        val[q] *= gate;
                    }
                }
                if (ok[u]) {
                    const f32x4 o1 = e.mul ? val * m1[u] : val;
                    if (e.out) *reinterpret_cast<f32x4*>(e.out + idx[u]) = o1;
                    if (want_max)
                        mx1[u] = max(max(__float_as_uint(o1[0]) & 0x7fffffffu, __float_as_uint(o1[1]) & 0x7fffffffu),
                                     max(__float_as_uint(o1[2]) & 0x7fffffffu, __float_as_uint(o1[3]) & 0x7fffffffu));
                    if (e.out2) {
                        f32x4 o2 = val;
                        if (e.mul2) o2 *= m2[u];
                        if (gate_mul) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) o2[q] = (__float_as_uint(m1[u][q]) & 1u) ? o2[q] : 0.f;
                        } else if (e.gate2) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) o2[q] = g2[u][q] > 0.f ? o2[q] : 0.f;
                        }
                        *reinterpret_cast<f32x4*>(e.out2 + idx[u]) = o2;
                        if (want_max)
                            mx2[u] = max(max(__float_as_uint(o2[0]) & 0x7fffffffu, __float_as_uint(o2[1]) & 0x7fffffffu),
                                         max(__float_as_uint(o2[2]) & 0x7fffffffu, __float_as_uint(o2[3]) & 0x7fffffffu));
                    }
                    if (e.scale_out) *reinterpret_cast<f32x4*>(e.scale_out + idx[u]) = s;
                }
            }
        } else if (col < Cout) {
            // ragged right edge (Cout % 4 != 0) or unaligned tensors: same math, element by element
            for (int u = 0; u < EPI_G; ++u) {
                const int lrow = rbase + (p0 + u) * RPP;
                const int row = (lrow / HM) * WM + pm * HM + lrow % HM;
                const int64_t pix = pixs[u];
                if (pix < 0) continue;
                for (int q = 0; q < 4 && col + q < Cout; ++q) {
                    const int64_t idx = pix * g.out_pitch + col + q;
                    float v = sC[lrow * LDC + cq * 4 + q];
                    if (SCALED) v = v * sAinv[row] * cinv4[q];
                    else {
                        if (row_pre) v = v * sAinv[row];
                        if (col_pre) v = v * cinv4[q];
                    }
                    v += bias4[q];
                    float s = 1.f;
                    if (NORM && !norm_only) {
                        s = b_is_2 ? fabsf(v) * sRinv[row] : powf(fabsf(v / sNorm[row]) + 1e-6f, bm1);
                        v *= s;
                    }
                    v = v * csc4[q] + csh4[q];
                    s *= csc4[q];
                    if (!NORM && asub) v += sApix[row] >= 0 ? e.addend[sApix[row] * g.out_pitch + col + q] : 0.f;
                    else if (e.addend) v += e.addend[idx];
                    if (e.relu == 1) {
                        const bool open_gate = e.relu_gate ? e.relu_gate[idx] > 0.f : v > 0.f;
                        s = open_gate ? (gate_lsb ? __uint_as_float(__float_as_uint(s) | 1u) : s) : 0.f;
                        v = open_gate ? v : 0.f;
                    } else if (e.relu == 2) {
                        const float gate = bcos_gelu_gate(v);
                        s *= gate;
                        v *= gate;
                    }
                    const float o1 = e.mul ? v * (mul_from_act ? rebuild_t(e.mul[idx], e.mul_norm[pix], mcsc4[q], mcsh4[q]) : e.mul[idx]) : v;
                    if (e.out) e.out[idx] = o1;
                    mx1[u] = max(mx1[u], __float_as_uint(o1) & 0x7fffffffu);
                    if (e.out2) {
                        float o2 = v;
                        if (e.mul2) o2 *= e.mul2[idx];
                        if (gate_mul) o2 = (__float_as_uint(e.mul[idx]) & 1u) ? o2 : 0.f;
                        else if (e.gate2) o2 = e.gate2[idx] > 0.f ? o2 : 0.f;
                        e.out2[idx] = o2;
                        mx2[u] = max(mx2[u], __float_as_uint(o2) & 0x7fffffffu);
                    }
                    if (e.scale_out) e.scale_out[idx] = s;
                }
            }
        }
        if (want_max) {
            // per-pixel max |value| of what this tile wrote (bit pattern: monotonic for non-negative floats), for the
            // operand scaling of the split-f16 contraction in the layer that reads the tensor; all lanes are active here
#pragma unroll
            for (int u = 0; u < EPI_G; ++u) {
                unsigned a1 = e.out_absmax ? group_max_u32<CPR>(mx1[u]) : 0u;
                unsigned a2 = e.out2_absmax ? group_max_u32<CPR>(mx2[u]) : 0u;
                if (cq == group_max_lane<CPR>() && pixs[u] >= 0) {
                    if (e.out_absmax && a1) atomicMax(e.out_absmax + pixs[u], a1);
                    if (e.out2_absmax && a2) atomicMax(e.out2_absmax + pixs[u], a2);
                }
            }
        }
    };
    // (requesting group g+1's inputs before group g is computed -- two register sets -- was measured: 30 more VGPRs and
    //  the forward HBM-bound launches got 10 % SLOWER, 2.47 -> 2.77 ms for the four 64 -> 256 @ 56^2 layers; not kept)
    EpiIn in;
#pragma unroll 1
    for (int p0 = 0; p0 < PASSES; p0 += EPI_G) {
        issue(p0, in);
        process(p0, in);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Specialised epilogues.  The layers around the 4x-wide block tensors (K <= 256) move ~2.7 GB per launch through the
// epilogue and are bound by instruction issue there, not by the contraction: the general
// epilogue above decides every feature at run time and spends ~210 vector instructions per 4 output elements.  The
// launches of a ResNet-style forward + explanation pass use a handful of feature sets; for those the host selects a
// compile-time specialisation (KArgs.epi_kind) that
//   * knows its tensors at compile time (EF_* bits), computes the same expressions in the same order as the general
//     epilogue (bit-identical results; tests/test_gpu_parity.py::test_fast_epilogue_bit_identical),
//   * addresses every tensor through raw buffer descriptors with 32-bit byte offsets (tensors < 2 GiB): rows beyond M and
//     columns beyond Cout carry an out-of-range offset, the hardware bounds check drops their stores and zero-fills their
//     loads, so the loop body has no predicates,
//   * keeps the per-row values (output offset, 1 / norm, inverse operand scale, pixel index) as one 16-byte LDS record.
// Anything else (MaxOut, B != 2, GELU, replayed gates, ragged Cout, tensors >= 2 GiB, ...) takes the general epilogue.
enum : int { EF_ADDEND = 1, EF_RELU = 2, EF_SCALE_OUT = 4, EF_MUL = 8, EF_OUT2 = 16, EF_MUL2 = 32, EF_GELU = 64, EF_MULACT = 128, EF_ROWADD = 256, EF_NONE = -1 };
// forward kinds (NORM kernels): B = 2 scale, optional bias / channel affine;  backward kinds (no norm): gradient multipliers
// (the GELU kinds are the linear1 layers of the B-cosified ViTs: MyGELU with its gate folded into the stored multiplier)
constexpr int EPI_KINDS_FWD[] = {EF_RELU | EF_SCALE_OUT, EF_RELU | EF_SCALE_OUT | EF_ADDEND, EF_SCALE_OUT, EF_RELU, EF_RELU | EF_ADDEND, 0,
                                 EF_GELU | EF_SCALE_OUT, EF_GELU};
constexpr int EPI_KINDS_BWD[] = {EF_MUL, EF_MUL | EF_ADDEND | EF_OUT2, EF_MUL | EF_ADDEND | EF_OUT2 | EF_MUL2, 0, EF_ADDEND, EF_MUL | EF_OUT2,
                                 EF_MUL | EF_MULACT, EF_ADDEND | EF_ROWADD};       // MULACT: the multiplier is rebuilt from the kept activation (BCOS_EPI_MUL_FROM_ACT);
                                                                                   // ROWADD: out = acc + (rowadd_scale[pixel] rowadd + addend), the addend optional (bcos_epilogue.rowadd)
constexpr int N_EPI_KINDS = 8;

struct __attribute__((aligned(16))) EpiRow { unsigned off; float rinv; float ainv; int pix; };

// a * b rounded, THEN + c rounded (no fused multiply-add: the two passes this stands in for round twice)
__device__ __forceinline__ float mul_then_add(float a, float b, float c) {
#pragma clang fp contract(off)
    const float prod = a * b;
    return prod + c;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int BM, int BN, int WAVES_M, int WAVES_N, bool NORM, bool SCALED, int NT, bool ROWX = true>
__device__ __forceinline__ void epi_rows_fast(float* smem, const float* ss, const float* ROWSS, const float* AINV, const int m0,
                                              const unsigned char* lvl, const int pass, KArgsSegPtr kpin) {
    BCOS_EPI_SHAPE
    BCOS_EPI_KARGS
    constexpr unsigned OOB = 0x80000000u;
    float* sC = smem;
    EpiRow* sRow = reinterpret_cast<EpiRow*>(smem + SBM * LDC);    // [BM]
    float* sNorm = reinterpret_cast<float*>(sRow + BM);            // [BM] patch norm (for norm_out)
    const int out_pitch = g.out_pitch;
    if (e.out_imgmax != nullptr && tid < 33) {
        // per-image range of the per-pixel maxima this tile emits (bcos_epilogue.out_imgmax / out_imgmin_c): [16] maxima, [16] complemented
        // minima over the nonzero pixels, [1] the image of the tile's first row -- a tile spans at most 16 images (host)
        unsigned* sImg = reinterpret_cast<unsigned*>(epi_col_table<BM, BN, WAVES_M>(smem) + BN) + 2 * BM;
        int n0i = 0, i0i = 0, j0i = 0;
        tile_row_nij(p, m0, 0, n0i, i0i, j0i);
        sImg[tid] = tid < 32 ? 0u : (unsigned)n0i;
    }
    for (int r = tid; r < BM; r += NT) {
        int pix = -1;
        int n = 0, i = 0, jj = 0;
        if (tile_row_nij(p, m0, r, n, i, jj) && (lvl == nullptr || lvl[r] == pass))
            pix = (n * g.OH + (i * g.out_sh + g.out_h0)) * g.OW + (jj * g.out_sw + g.out_w0);
        sRow[r].pix = pix;
        sRow[r].off = pix >= 0 ? (unsigned)pix * (unsigned)out_pitch * 4u : OOB;
        if (!NORM && e.addend_sub > 1) {
            // subsampled addend (gradient kinds, whose rows carry no 1 / norm): byte offset of this row in the
            // [N, ceil(OH / s), ceil(OW / s), out_pitch] addend, out of range where the pixel is off the s-grid
            const int s = e.addend_sub;
            unsigned aoff = OOB;
            if (pix >= 0) {
                const int h = i * g.out_sh + g.out_h0, w = jj * g.out_sw + g.out_w0;
                if (h % s == 0 && w % s == 0)
                    aoff = (unsigned)((n * ((g.OH + s - 1) / s) + h / s) * ((g.OW + s - 1) / s) + w / s) * (unsigned)out_pitch * 4u;
            }
            sRow[r].rinv = __uint_as_float(aoff);
        }
        if (!NORM && e.rowadd != nullptr) sRow[r].rinv = pix >= 0 ? e.rowadd_scale[pix] : 0.f;      // (host: rowadd excludes addend_sub > 1)
    }
    // output pixel of tile row r (bcos_epilogue.row_scale / a_sumsq are indexed by it), or -1
    auto row_pix = [&](int r) -> int {
        int n, i, jj;
        if (!tile_row_nij(p, m0, r, n, i, jj)) return -1;
        return (n * g.OH + (i * g.out_sh + g.out_h0)) * g.OW + (jj * g.out_sw + g.out_w0);
    };
    if (NORM && ROWSS == nullptr) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float t = ss[i] + __shfl_xor(ss[i], 32);
            if (wave_n == 0 && lane < 32) {
                const int row = wave_m * WM + i * 32 + lane;
                if (ROWX && e.a_sumsq) { const int px = row_pix(row); t = px >= 0 ? e.a_sumsq[px] : 0.f; }
                float nrm = e.bcos_mode == BCOS_LINEAR_EPS ? sqrtf(t) + 1e-12f : sqrtf(t + 1e-6f);
                sNorm[row] = nrm;
                sRow[row].rinv = 1.0f / nrm;
            }
        }
    } else if (NORM) {
        constexpr int RP = NT / 4;
#pragma unroll
        for (int j = 0; j < BM / RP; ++j) {
            float t = ROWSS[j];
            t += __shfl_xor(t, 1);
            t += __shfl_xor(t, 2);
            if ((tid & 3) == 0) {
                const int row = (tid >> 2) + RP * j;
                if (ROWX && e.a_sumsq) { const int px = row_pix(row); t = px >= 0 ? e.a_sumsq[px] : 0.f; }
                float nrm = e.bcos_mode == BCOS_LINEAR_EPS ? sqrtf(t) + 1e-12f : sqrtf(t + 1e-6f);
                sNorm[row] = nrm;
                sRow[row].rinv = 1.0f / nrm;
            }
        }
    }
    if (SCALED) {
        // (bcos_epilogue.row_scale rides in the inverse operand scale of the row: launches that carry one and are not SCALED take the
        //  general epilogue)
#pragma unroll
        for (int j = 0; j < BM / (NT / 4); ++j)
            if ((tid & 3) == 0) {
                const int row = (tid >> 2) + (NT / 4) * j;
                float ai = AINV[j];
                if (ROWX && e.row_scale) { const int px = row_pix(row); ai *= px >= 0 ? e.row_scale[px] : 0.f; }
                sRow[row].ainv = ai;
            }
    }

}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool NORM, bool SCALED, int NT, int EF>
__device__ __forceinline__ void epi_part_fast(float* smem, const int pm, const int pn, const int part, const int n0, const int tile_n,
                                              KArgsSegPtr kpin) {
    BCOS_EPI_SHAPE
    BCOS_EPI_KARGS
    constexpr unsigned OOB = 0x80000000u;
    constexpr bool ADDEND = (EF & EF_ADDEND) != 0, RELU = (EF & EF_RELU) != 0, SCALE_OUT = (EF & EF_SCALE_OUT) != 0;
    constexpr bool MUL = (EF & EF_MUL) != 0, OUT2 = (EF & EF_OUT2) != 0, MUL2 = (EF & EF_MUL2) != 0, GELU = (EF & EF_GELU) != 0;
    static_assert(NORM ? !(MUL || OUT2 || MUL2) : !(RELU || SCALE_OUT || GELU), "forward kinds scale, backward kinds multiply");
    static_assert(!(RELU && GELU), "one activation");
    constexpr bool MULACT = (EF & EF_MULACT) != 0;
    static_assert(!MULACT || (MUL && !OUT2), "rebuilt multipliers: plain gradient launches");
    constexpr bool ROWADD = (EF & EF_ROWADD) != 0;
    static_assert(!ROWADD || (!NORM && ADDEND && !MUL && !OUT2), "row-scaled addend: plain gradient launches");
    float* sC = smem;
    EpiRow* sRow = reinterpret_cast<EpiRow*>(smem + SBM * LDC);    // [BM]
    float* sNorm = reinterpret_cast<float*>(sRow + BM);            // [BM] patch norm (for norm_out)
    const unsigned lsb = (RELU && SCALE_OUT && (e.flags & BCOS_EPI_SCALE_GATE_LSB)) ? 1u : 0u;
    const bool want_max1 = e.out_absmax != nullptr, want_max2 = OUT2 && e.out2_absmax != nullptr;
    const int Cout = ((int)blockIdx.y + 1) * g.Cout;      // end of this launch's (group's) column range; blockIdx.y = group
    const unsigned tbytes = p.out_bytes;
    auto rsrc = [&](const void* q) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q), 0, tbytes, 0x00020000); };
    const __amdgpu_buffer_rsrc_t r_out = rsrc(e.out);
    // (ROWADD: the addend is optional -- a descriptor of zero records answers every load with zeros)
    const __amdgpu_buffer_rsrc_t r_ad = (ROWADD && e.addend == nullptr)
        ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e.out), 0, 0, 0x00020000) : rsrc(ADDEND ? e.addend : e.out);
    const __amdgpu_buffer_rsrc_t r_radd = rsrc(ROWADD ? e.rowadd : e.out);
    const __amdgpu_buffer_rsrc_t r_mul = rsrc(MUL ? e.mul : e.out);
    const __amdgpu_buffer_rsrc_t r_mul2 = rsrc(MUL2 ? e.mul2 : e.out);
    const __amdgpu_buffer_rsrc_t r_out2 = rsrc(OUT2 ? e.out2 : e.out);
    const __amdgpu_buffer_rsrc_t r_sc = rsrc(SCALE_OUT ? e.scale_out : e.out);
    // Cache policy of the streamed tensors (aux bits of the buffer instructions: 2 = nt, non-temporal).  The stored multipliers are
    // written in the forward and read once, much later, by the explanation pass; addends / multipliers are read exactly once.
    // Same-node A/B on ResNet-50: nt on the multiplier stores and on the input loads of the forward kinds and of the >= 128-column
    // gradient tiles is -0.25 ms per step (wide launches -3 % each); nt loads in the 64-column gradient tiles cost +5 % there and
    // nt on the activation stores (re-read by the next launch) is neutral to negative, so those keep the default policy.
    constexpr int AUX_IN = (NORM || BN >= 128) ? 2 : 0;
    constexpr int AUX_SC = 2;
    auto ldq = [](const __amdgpu_buffer_rsrc_t& r, unsigned voff) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, AUX_IN));
    };
    auto stq = [](const __amdgpu_buffer_rsrc_t& r, unsigned voff, const f32x4& v) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)voff, 0, 0);
    };
    auto stq_sc = [](const __amdgpu_buffer_rsrc_t& r, unsigned voff, const f32x4& v) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)voff, 0, AUX_SC);
    };
    auto absmax4 = [](const f32x4& o) {
        return max(max(__float_as_uint(o[0]) & 0x7fffffffu, __float_as_uint(o[1]) & 0x7fffffffu),
                   max(__float_as_uint(o[2]) & 0x7fffffffu, __float_as_uint(o[3]) & 0x7fffffffu));
    };
    constexpr int CPR = SBN / 4;
    constexpr int RPP = NT / CPR;
    constexpr int PASSES = SBM / RPP;
    constexpr int NIN = (ADDEND ? 1 : 0) + (MUL ? 1 : 0) + (MUL2 ? 1 : 0) + (ROWADD ? 1 : 0);
    // (the narrow split-f16 forward kernels are compiled for three workgroups per CU -- 168 registers: 4 chunks there)
    constexpr int G = (NIN <= 1 && PASSES % 8 == 0 && NT <= 256 && PM * PN == 1 && !(SCALED && NORM && BN <= 64)) ? 8 : (PASSES % 4 == 0 ? 4 : 2);
    static_assert(PASSES % G == 0, "epilogue grouping");
    const int cq = tid % CPR;
    const int rbase = tid / CPR;
    const bool asub = ADDEND && !NORM && e.addend_sub > 1;      // subsampled addend: its row offsets are in EpiRow.rinv

    if (NORM && e.norm_out != nullptr && tile_n == 0 && part == 0) {
        for (int r = tid; r < BM; r += NT) {
            const int pix = sRow[r].pix;
            if (pix >= 0) e.norm_out[(int64_t)pix * g.norm_pitch + blockIdx.y] = sNorm[r];
        }
    }
    const int col = n0 + ((cq * 4) / HN) * WN + pn * HN + (cq * 4) % HN;
    const bool col_ok = col < Cout;                       // Cout % 4 == 0 (host): the whole chunk is inside or outside
    unsigned coloff = col_ok ? (unsigned)col * 4u : OOB;
    if (g.out_cgroup > 0 && col_ok) {      // depth to space: column block (dh, dw) is the pixel dh * OW + dw further on
        const int cls = col / g.out_cgroup;
        coloff = (unsigned)(((cls / g.out_sw) * g.OW + cls % g.out_sw) * g.out_pitch + (col - cls * g.out_cgroup)) * 4u;
    }
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, csc4 = {1.f, 1.f, 1.f, 1.f}, csh4 = {0.f, 0.f, 0.f, 0.f}, cinv4 = {1.f, 1.f, 1.f, 1.f};
    if (col_ok) {
        if (e.bias) bias4 = *reinterpret_cast<const f32x4*>(e.bias + col);
        if (NORM && e.ch_scale) csc4 = *reinterpret_cast<const f32x4*>(e.ch_scale + col);
        if (NORM && e.ch_shift) csh4 = *reinterpret_cast<const f32x4*>(e.ch_shift + col);
    }
    if (SCALED) {
#pragma unroll
        for (int q = 0; q < 4; ++q) cinv4[q] = p.wt2_cinv[col_ok ? col + q : 0];
    }
    f32x4 mcsc4 = {1.f, 1.f, 1.f, 1.f}, mcsh4 = {0.f, 0.f, 0.f, 0.f};
    if (MULACT && col_ok) {
        if (e.mul_csc) mcsc4 = *reinterpret_cast<const f32x4*>(e.mul_csc + col);
        if (e.mul_csh) mcsh4 = *reinterpret_cast<const f32x4*>(e.mul_csh + col);
    }
#pragma unroll 1
    for (int p0 = 0; p0 < PASSES; p0 += G) {
        EpiRow rw[G];
        unsigned voff[G];
        f32x4 ad[G], m1[G], m2[G], xr[G];
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int lrow = rbase + (p0 + u) * RPP;
            rw[u] = sRow[(lrow / HM) * WM + pm * HM + lrow % HM];
            voff[u] = (rw[u].off + coloff) | ((rw[u].off | coloff) & OOB);
            if (ROWADD) xr[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_radd, (int)voff[u], 0, 0));      // (x is read again by the weight gradient: default cache policy)
            if (ADDEND && !NORM && asub) {
                const unsigned aoff = __float_as_uint(rw[u].rinv);
                ad[u] = ldq(r_ad, (aoff + coloff) | ((aoff | coloff) & OOB));
            } else if (ADDEND) ad[u] = ldq(r_ad, voff[u]);
            if (MUL) m1[u] = ldq(r_mul, voff[u]);
            if (MUL2) m2[u] = ldq(r_mul2, voff[u]);
        }
        unsigned mx1[G], mx2[G];
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int lrow = rbase + (p0 + u) * RPP;
            f32x4 v = *reinterpret_cast<const f32x4*>(sC + lrow * LDC + cq * 4);
            if (SCALED) v = v * rw[u].ainv * cinv4;
            f32x4 val = v + bias4;
            f32x4 s = {1.f, 1.f, 1.f, 1.f};
            if (NORM) {
#pragma unroll
                for (int q = 0; q < 4; ++q) s[q] = fabsf(val[q]) * rw[u].rinv;
                val *= s;
            }
            val = val * csc4 + csh4;
            s *= csc4;
            if (ROWADD) {
                // (the roundings of bcos_patch_norm_bwd_add followed by a plain addend -- acc + ((x r) + addend), product and sum rounded
                //  separately as that kernel does: the fused launch gives the bits of the two passes it replaces)
#pragma unroll
                for (int q = 0; q < 4; ++q) val[q] += mul_then_add(xr[u][q], rw[u].rinv, ad[u][q]);
            } else if (ADDEND) val += ad[u];
            if (RELU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool open_gate = val[q] > 0.f;
                    s[q] = open_gate ? __uint_as_float(__float_as_uint(s[q]) | lsb) : 0.f;
                    val[q] = open_gate ? val[q] : 0.f;
                }
            }
            if (GELU) {           // GELU with the gate treated as a constant (MyGELU, bcosify_vit.py:27-32)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float gate = bcos_gelu_gate(val[q]);
                    s[q] *= gate;
                    val[q] *= gate;
                }
            }
            if (MULACT) {          // t = csc sqrt(|a - csh| / (|csc| norm)) where the kept activation a is positive (B = 2, no residual)
                const float mn = e.mul_norm[rw[u].pix >= 0 ? rw[u].pix : 0];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float a = m1[u][q];
                    const float den = fabsf(mcsc4[q]) * mn;
                    m1[u][q] = (a > 0.f && den > 0.f) ? mcsc4[q] * __builtin_amdgcn_sqrtf(fabsf(a - mcsh4[q]) * __builtin_amdgcn_rcpf(den)) : 0.f;
                }
            }
            const f32x4 o1 = MUL ? val * m1[u] : val;
            stq(r_out, voff[u], o1);
            mx1[u] = want_max1 ? absmax4(o1) : 0u;
            if (OUT2) {
                f32x4 o2 = val;
                if (MUL2) o2 *= m2[u];
                if (MUL && (e.flags & BCOS_EPI_GATE2_FROM_MUL)) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) o2[q] = (__float_as_uint(m1[u][q]) & 1u) ? o2[q] : 0.f;
                }
                stq(r_out2, voff[u], o2);
                mx2[u] = want_max2 ? absmax4(o2) : 0u;
            }
            if (SCALE_OUT) stq_sc(r_sc, voff[u], s);
        }
        if (want_max1 || want_max2) {
            // Per-pixel maxima of what the tile wrote.  A row's maximum over the tile's column PARTS is carried in LDS (the same lane
            // leads the same row in every part), so a tile contributes ONE update per row and tensor; a launch with a single column
            // tile (Cout <= BN) owns its pixels and writes them with a plain store -- round 3 issued one fp-order-free but L2-serialised
            // atomicMax per row, part and tensor: 72 us of the 444 us of the 64 -> 256 @56^2 forward launch (scripts/probe/epi_knock.py).
            unsigned* sMax = reinterpret_cast<unsigned*>(epi_col_table<BM, BN, WAVES_M>(smem) + BN);      // [2][BM]
            const bool own = p.tiles_n == 1 && g.out_pitch == g.Cout;      // (a launch that writes a column slice of wider pixels shares them)
            const bool img_on = e.out_imgmax != nullptr;
#pragma unroll
            for (int u = 0; u < G; ++u) {
                // lanes outside the tensor hold zeros or values whose stores were dropped: keep them out of the maxima
                const bool live = (int)voff[u] >= 0;
                unsigned a1 = want_max1 ? group_max_u32<CPR>(live ? mx1[u] : 0u) : 0u;
                unsigned a2 = want_max2 ? group_max_u32<CPR>(live ? mx2[u] : 0u) : 0u;
                if (cq == group_max_lane<CPR>() && rw[u].pix >= 0) {
                    const int lrow = rbase + (p0 + u) * RPP;
                    const int trow = (lrow / HM) * WM + pm * HM + lrow % HM;
                    if constexpr (PN > 1) {
                        if (pn > 0) { a1 = max(a1, sMax[trow]); a2 = max(a2, sMax[BM + trow]); }
                        if (pn + 1 < PN) { sMax[trow] = a1; sMax[BM + trow] = a2; }
                    }
                    if (pn + 1 == PN) {
                        if (own) {
                            if (want_max1) e.out_absmax[rw[u].pix] = a1;
                            if (want_max2) e.out2_absmax[rw[u].pix] = a2;
                        } else {
                            if (want_max1 && a1) atomicMax(e.out_absmax + rw[u].pix, a1);
                            if (want_max2 && a2) atomicMax(e.out2_absmax + rw[u].pix, a2);
                        }
                        if (want_max1 && img_on) sMax[trow] = a1;       // (kept for the per-image range: tile_epilogue folds the rows' maxima behind the last part)
                    }
                }
            }
        }
    }
}

// Epilogue of one tile: row metadata, then per part: accumulators -> LDS transpose buffer sC[SBM][SBN+4] (MFMA layout:
// lane = column, 16 rows per lane) and the specialisation the host selected for this launch, or the general code.  The
// accumulators are only touched HERE: the (large) part functions take no reference to them, so the registers of the
// parts still waiting stay put while one part is drained.
// Tiles larger than 128 x 128 are drained in 128 x 128 parts (one part = half the accumulators of every wave) so that the
// LDS transpose buffer stays at 66 KB and two workgroups fit a CU.
// ROWX = false (the input-patch loop): bcos_epilogue.row_scale / a_sumsq are compiled out -- they belong to launches that read a LayerNorm's
// input (1 x 1 geometries), and their row lookups cost the patch kernels registers they do not have (248 -> 256 VGPRs, +0.16 ms per step)
template <int BM, int BN, int WAVES_M, int WAVES_N, bool NORM, bool SCALED, int NT = NTHREADS, bool ROWX = true>
__device__ __forceinline__ void tile_epilogue(const auto& p, float* smem, f32x16 (&acc)[(BM / WAVES_M) / 32][(BN / WAVES_N) / 32],
                                              const float* ss, const float* ROWSS, const float* AINV, const int m0, const int n0,
                                              const int tile_n, const unsigned char* lvl = nullptr, const int pass = 0) {
    BCOS_EPI_SHAPE
    const int kind = p.epi_kind;
    BCOS_PHASE_MARK(pe_t0);
    KArgsSegPtr kpin = nullptr;          // (see BCOS_EPI_KARGS)
    if constexpr (!std::is_same_v<std::remove_cvref_t<decltype(p)>, KArgs>) kpin = &p;
    // (all waves are past the last barrier of the main loop: the staging buffers are free)
    if (kind) epi_rows_fast<BM, BN, WAVES_M, WAVES_N, NORM, SCALED, NT, ROWX>(smem, ss, ROWSS, AINV, m0, lvl, pass, kpin);
    else epi_rows_generic<BM, BN, WAVES_M, WAVES_N, NORM, SCALED, NT, ROWX>(smem, ss, ROWSS, AINV, m0, lvl, pass, kpin);
    float* sC = smem;
    // (the parts are expanded at compile time: a loop the optimiser declines to unroll would index the accumulators at run time)
    auto drain = [&](auto part_c) {
        constexpr int part = decltype(part_c)::value;
        constexpr int pm = part / PN, pn = part - pm * PN;
        if (part > 0) __syncthreads();           // the previous part's sC has been consumed
#pragma unroll
        for (int ii = 0; ii < TMP; ++ii)
#pragma unroll
            for (int jj = 0; jj < TNP; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wave_m * HM + ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int colt = wave_n * HN + jj * 32 + (lane & 31);
                    sC[row * LDC + colt] = acc[pm * TMP + ii][pn * TNP + jj][r];
                }
        __syncthreads();                         // (part 0: also publishes the row metadata written above)
#if BCOS_PHASE_TIMING
        unsigned long long pe_t1 = 0;
        if constexpr (part == 0) pe_t1 = wall_clock64();
#endif
#define BCOS_EPI_CASE(I)                                                                                                     \
    case I + 1:                                                                                                              \
        epi_part_fast<BM, BN, WAVES_M, WAVES_N, NORM, SCALED, NT, (NORM ? EPI_KINDS_FWD[I] : EPI_KINDS_BWD[I])>(smem, pm, pn, part, \
                                                                                                                n0, tile_n, kpin); \
        break;
        switch (kind) {
            BCOS_EPI_CASE(0) BCOS_EPI_CASE(1) BCOS_EPI_CASE(2) BCOS_EPI_CASE(3) BCOS_EPI_CASE(4) BCOS_EPI_CASE(5)
            BCOS_EPI_CASE(6) BCOS_EPI_CASE(7)
            default: epi_part_generic<BM, BN, WAVES_M, WAVES_N, NORM, SCALED, NT>(smem, pm, pn, part, n0, tile_n, kpin);
        }
#undef BCOS_EPI_CASE
#if BCOS_PHASE_TIMING
        if constexpr (part == 0) {
            if (threadIdx.x == 0) {
                atomicAdd(&g_phase[6], pe_t1 - pe_t0);
                atomicAdd(&g_phase[7], wall_clock64() - pe_t1);
            }
        }
#endif
    };
    static_assert(PM * PN <= 4, "parts");
    drain(std::integral_constant<int, 0>{});
    if constexpr (PM * PN > 1) drain(std::integral_constant<int, 1>{});
    if constexpr (PM * PN > 2) drain(std::integral_constant<int, 2>{});
    if constexpr (PM * PN > 3) drain(std::integral_constant<int, 3>{});
    // Per-image range of the per-pixel maxima (bcos_epilogue.out_imgmax / out_imgmin_c, specialised epilogues): the rows of the tile have
    // folded their maxima into at most 16 LDS cells per array; one atomic per touched image and array hands them to the launch-wide
    // arrays, which the reader of the tensor (a 3 x 3 launch over an LDS-resident patch) takes as its per-image operand scale -- the
    // separate bcos_image_absrange pass over the maxima (28 launches of a ResNet-50 step, ~6 us each on the critical path) is gone.
    // The maximum is exact (max is associative); the minimum over the NONZERO pixels is exact when the tile owns its pixels (one column
    // tile) and a lower bound otherwise (a column tile sees its own columns' maxima) -- it only switches the level bookkeeping on.
    if (kind && p.e.out_imgmax != nullptr) {
        __syncthreads();
        unsigned* sMaxR = reinterpret_cast<unsigned*>(epi_col_table<BM, BN, WAVES_M>(smem) + BN);      // [BM]: every row's maximum over the tile's columns
        unsigned* sImg = sMaxR + 2 * BM;
        const EpiRow* sRowR = reinterpret_cast<const EpiRow*>(smem + SBM * LDC);
        const int img_px = p.g.OH * p.g.OW;
        for (int r = tid; r < BM; r += NT) {
            const int pix = sRowR[r].pix;
            const unsigned a1 = pix >= 0 ? sMaxR[r] : 0u;
            if (a1) {
                const int slot = pix / img_px - (int)sImg[32];
                atomicMax(sImg + slot, a1);
                atomicMax(sImg + 16 + slot, ~a1);
            }
        }
        __syncthreads();
        if (tid < 32) {
            const unsigned v = sImg[tid];
            if (v) atomicMax((tid < 16 ? p.e.out_imgmax : p.e.out_imgmin_c) + sImg[32] + (tid & 15), v);
        }
    }
}

// One output tile [m0, m0+BM) x [n0, n0+BN): main loop + epilogue.  `tile_n` only tells whether this block is the one
// that writes the per-row norms.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool NORM>
__device__ __forceinline__ void tile_body(const KArgs& p, float* smem, const int m0, const int n0, const int tile_n) {
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_LD = BM / 32, B_LD = BN / 32;   // float4 loads per thread per K-step
    constexpr int BUF = (BM + BN) * LDS_LD;          // floats per LDS buffer
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");
    static_assert(TM >= 1 && TN >= 1, "wave tile");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    const int chunk = tid & 7;      // 16-B chunk within the 32-float K-step
    const int r0 = tid >> 3;        // staging row 0..31 (+32 per pass)

    const bcos_tapconv_geom& g = p.g;
    const int H = g.H, W = g.W;
    const int a_pitch = g.a_pitch;

    // ---- per-thread staging rows (fixed over the K loop) -------------------------------
    int64_t a_nbase[A_LD];
    int a_ih0[A_LD], a_iw0[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        const int m = m0 + r0 + 32 * j;
        if (m < p.M) {
            const int n = m / p.PQ;
            const int rem = m - n * p.PQ;
            const int i = rem / g.Q;
            const int jj = rem - i * g.Q;
            a_nbase[j] = (int64_t)n * H * W * a_pitch;
            a_ih0[j] = i * g.in_sh + g.dh0;
            a_iw0[j] = jj * g.in_sw + g.dw0;
        } else {
            a_nbase[j] = 0;
            a_ih0[j] = -(1 << 28);   // fails every bounds check -> zero rows
            a_iw0[j] = -(1 << 28);
        }
    }
    int64_t b_off[B_LD];
    bool b_ok[B_LD];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
        const int co = n0 + r0 + 32 * j;
        b_ok[j] = co < ((int)blockIdx.y + 1) * g.Cout;          // (grouped launches: blockIdx.y = group, n0 counts global columns)
        b_off[j] = (int64_t)(b_ok[j] ? co : 0) * p.Ktot;
    }

    f32x4 ra[A_LD], rb[B_LD];
    // Position of the NEXT K-step to load inside the (tap, channel) space.  When C % 32 == 0 a K-step lies inside
    // one tap, so the position is wave-uniform and advanced incrementally (scalar registers, no divisions);
    // otherwise (the 6->8 channel stem) every lane derives its own tap from its chunk index.
    int s_cc = 0, s_th = 0, s_tw = 0;      // uniform mode: chunk offset inside the tap, tap coordinates
    auto load_step = [&](int ks) {
        int cc, dh, dw;
        bool kvalid;
        const int q = ks * 8 + chunk;
        if (p.uniform_tap) {
            cc = s_cc + chunk;
            dh = s_th * g.dstep_h;
            dw = s_tw * g.dstep_w;
            kvalid = true;
            s_cc += 8;
            if (s_cc == p.cpt) {
                s_cc = 0;
                if (++s_tw == g.TW) { s_tw = 0; ++s_th; }
            }
        } else {
            kvalid = q < p.nchunks;
            const int tap = q / p.cpt;
            cc = q - tap * p.cpt;
            const int th = tap / g.TW;
            const int tw = tap - th * g.TW;
            dh = th * g.dstep_h;
            dw = tw * g.dstep_w;
        }
        // branch-free: out-of-image taps / rows beyond M load a valid dummy address and are zeroed afterwards
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            const int ih = a_ih0[j] + dh, iw = a_iw0[j] + dw;
            const bool ok = kvalid && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(p.a + (int)blockIdx.y * g.C + a_nbase[j] + ((int64_t)ih * W + iw) * a_pitch + cc * 4);
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            const bool ok = kvalid && b_ok[j];
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(p.wt + b_off[j] + (int64_t)q * 4);
            rb[j] = v;
        }
    };
    // BCOS_EPI_UNIT_NORM_W: sum of squares of every weight row of the tile, gathered from the staging registers the rows pass
    // through anyway (the unit-norm projection of NormedConv2d / NormedLinear fused into the contraction, bcosconv2d.py:26-35)
    const bool unit_w = (p.e.flags & BCOS_EPI_UNIT_NORM_W) != 0;
    float colss[B_LD];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) colss[j] = 0.f;
    auto store_step = [&](int buf) {
        float* sA = smem + buf * BUF;
        float* sB = sA + BM * LDS_LD;
#pragma unroll
        for (int j = 0; j < A_LD; ++j)
            *reinterpret_cast<f32x4*>(sA + (r0 + 32 * j) * LDS_LD + chunk * 4) = ra[j];
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            *reinterpret_cast<f32x4*>(sB + (r0 + 32 * j) * LDS_LD + chunk * 4) = rb[j];
            if (unit_w) colss[j] = fmaf(rb[j][0], rb[j][0], fmaf(rb[j][1], rb[j][1], fmaf(rb[j][2], rb[j][2], fmaf(rb[j][3], rb[j][3], colss[j]))));
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float ss[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) ss[i] = 0.f;

    const int frag_row = lane & 31, frag_half = lane >> 5;
    const int a_frag = (wave_m * WM + frag_row) * LDS_LD + frag_half * 4;
    const int b_frag = BM * LDS_LD + (wave_n * WN + frag_row) * LDS_LD + frag_half * 4;

    // ---- main loop ------------------------------------------------------------------------
    // per K-step: fragments of sub-step kk+1 are read from LDS while the 4*TM*TN MFMAs of kk execute (two
    // register sets), the global loads of K-step ks+1 are issued behind the first fragment reads and land in
    // the other LDS buffer after the MFMAs; one barrier per K-step.
    load_step(0);
    store_step(0);
    __syncthreads();
    for (int ks = 0; ks < p.nk; ++ks) {
        const int cur = ks & 1;
        const bool more = ks + 1 < p.nk;
        const float* sbuf = smem + cur * BUF;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const f32x4*>(sbuf + a_frag + i * 32 * LDS_LD + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[j] = *reinterpret_cast<const f32x4*>(sbuf + b_frag + j * 32 * LDS_LD + kk * 8);
            if (kk == 0 && more) load_step(ks + 1);
            if (NORM) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ss[i] = fmaf(af[i][0], af[i][0], ss[i]);
                    ss[i] = fmaf(af[i][1], af[i][1], ss[i]);
                    ss[i] = fmaf(af[i][2], af[i][2], ss[i]);
                    ss[i] = fmaf(af[i][3], af[i][3], ss[i]);
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][c], bf[j][c], acc[i][j], 0, 0, 0);
        }
        if (more) store_step(cur ^ 1);
        __syncthreads();
    }

    if (unit_w) {        // 1 / ||w_c|| of the tile's columns: the 8 chunk lanes of a staging row meet, one value per column
        float* sCol = epi_col_table<BM, BN, WAVES_M>(smem);
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            float t = colss[j];
            t += __shfl_xor(t, 1); t += __shfl_xor(t, 2); t += __shfl_xor(t, 4);
            if (chunk == 0) sCol[r0 + 32 * j] = 1.0f / sqrtf(t);
        }
    }
    tile_epilogue<BM, BN, WAVES_M, WAVES_N, NORM, false>(p, smem, acc, ss, nullptr, nullptr, m0, n0, tile_n);
}

// ---------------------------------------------------------------------------------------------------------------
// Split-bf16 main loop ("bf16x3"): every fp32 operand x is split EXACTLY into three bf16 numbers x = h + m + l (8 + 8 + 8
// significand bits, bf16 has fp32's exponent range so no scaling is needed), and a*b is evaluated as
//     a_h b_h + (a_h b_m + a_m b_h) + (a_m b_m + a_h b_l + a_l b_h)
// i.e. 6 products on v_mfma_f32_32x32x16_bf16 (products of bf16 numbers are exact in fp32, accumulation is fp32).
// The dropped terms (a_m b_l, a_l b_m, a_l b_l) are <= 2^-21 |a b|: fp32-rounding class, but 6 bf16 MFMAs cost 192
// matrix-pipe cycles per 16 k against 512 for 8 fp32 MFMAs.  Same tiling / staging / epilogue as tile_body; the
// splits are produced while the tile is written to LDS (one 16-byte global load -> three 8-byte LDS stores), LDS rows
// are 16 bf16 = 32 B with an XOR swizzle of the two halves (conflict-free stores and ds_read_b128); K advances 16 per
// step through two LDS buffers: while the 6*TM*TN MFMAs of step k run, step k+1 is converted and written to the
// other buffer and the global loads of step k+2 are in flight (one barrier per step).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int X3_BK = 16;      // k per step = one MFMA k-block; two LDS buffers
constexpr int X3_ROW = 32;     // bytes per LDS row of one split: 16 bf16, unpadded; the two 16-B halves of a row are swapped
                               // when bit 3 of the row index is set, which makes both the 8-byte staging stores (4 rows x 32 B
                               // per 16-lane group) and the 16-byte fragment reads (16 rows per group) bank-conflict free

template <int BM, int BN, int WAVES_M, int WAVES_N, bool NORM, bool PRE>
__device__ __forceinline__ void tile_body_x3(const KArgs& p, float* smem, const int m0, const int n0, const int tile_n) {
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_LD = BM / 64, B_LD = BN / 64 > 0 ? BN / 64 : 1;     // float4 loads per thread per 16-k step
    constexpr bool B_HALF = BN < 64;                                     // BN = 32: only half the threads stage B
    char* lds = reinterpret_cast<char*>(smem);
    constexpr int A_SPLIT = BM * X3_ROW, B_SPLIT = BN * X3_ROW, B_BASE = 3 * A_SPLIT;
    constexpr int BUF = 3 * (A_SPLIT + B_SPLIT);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    const int chunk = tid & 3;       // 16-byte chunk (4 k) within the 16-k step
    const int r0 = tid >> 2;         // staging row 0..63 (+64 per pass)
    const bcos_tapconv_geom& g = p.g;
    const int H = g.H, W = g.W;
    const int a_pitch = g.a_pitch;

    // Operands are fetched with raw buffer loads: a lane whose source is outside the image / past the last weight row
    // carries the offset OOB (>= num_records, the host guarantees both operands are < 2 GiB) and the hardware returns
    // zeros -- no exec-mask branches, no zero-filling moves, 32-bit offsets.  Inside a tap the K walk advances through
    // the scalar offset operand, so a steady-state step issues its loads with no vector ALU work at all.
    constexpr unsigned OOB = 0x80000000u;
    // (grouped launches: blockIdx.y = group; it reads its channel slice of A, n0 counts global output columns)
    const unsigned a_goff = (unsigned)blockIdx.y * (unsigned)g.C * 4u;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.a) + a_goff), 0, p.a_bytes - a_goff, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wt), 0, p.wt_bytes, 0x00020000);
    unsigned a_nbase[A_LD];          // byte offset of the row's image (+ this lane's 16-byte chunk)
    int a_ih0[A_LD], a_iw0[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        const int m = m0 + r0 + 64 * j;
        if (m < p.M) {
            const int n = m / p.PQ;
            const int rem = m - n * p.PQ;
            const int i = rem / g.Q;
            const int jj = rem - i * g.Q;
            a_nbase[j] = ((unsigned)n * H * W * a_pitch + chunk * 4) * 4u;
            a_ih0[j] = i * g.in_sh + g.dh0;
            a_iw0[j] = jj * g.in_sw + g.dw0;
        } else {
            a_nbase[j] = 0;
            a_ih0[j] = -(1 << 28);
            a_iw0[j] = -(1 << 28);
        }
    }
    unsigned b_off[B_LD];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
        const int co = n0 + r0 + 64 * j;
        const bool ok = co < ((int)blockIdx.y + 1) * g.Cout && (!B_HALF || r0 < BN);
        b_off[j] = ok ? ((unsigned)co * p.Ktot + chunk * 4) * 4u : OOB;
    }

    const int nk = (p.nchunks + 3) / 4;
    const bool uniform = (g.C % X3_BK) == 0;
    int s_cc = 0, s_th = 0, s_tw = 0;
    // Per-row source offset of the CURRENT tap is cached and only recomputed when the K walk enters a new tap (uniform
    // mode: C % 16 == 0); inside a tap a step just advances the scalar offset by 64 bytes.
    unsigned a_cur[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) a_cur[j] = OOB;
    auto ldq = [](const __amdgpu_buffer_rsrc_t& r, unsigned voff, int soff) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
    };
    auto load_step = [&](int ks, f32x4 (&ra)[A_LD], f32x4 (&rb)[B_LD]) {
        if (uniform) {
            if (s_cc == 0) {
                const int dh = s_th * g.dstep_h, dw = s_tw * g.dstep_w;
#pragma unroll
                for (int j = 0; j < A_LD; ++j) {
                    const int ih = a_ih0[j] + dh, iw = a_iw0[j] + dw;
                    const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                    a_cur[j] = ok ? a_nbase[j] + (unsigned)((ih * W + iw) * a_pitch) * 4u : OOB;
                }
            }
#pragma unroll
            for (int j = 0; j < A_LD; ++j) ra[j] = ldq(a_rsrc, a_cur[j], s_cc * 16);
            s_cc += 4;
            if (s_cc == p.cpt) {
                s_cc = 0;
                if (++s_tw == g.TW) { s_tw = 0; ++s_th; }
            }
            if (!PRE) {
#pragma unroll
                for (int j = 0; j < B_LD; ++j) rb[j] = ldq(b_rsrc, b_off[j], ks * 64);
            }
        } else {
            const int q = ks * 4 + chunk;
            const bool kvalid = q < p.nchunks;
            const int tap = q / p.cpt;
            const int cc = q - tap * p.cpt;
            const int th = tap / g.TW;
            const int tw = tap - th * g.TW;
            const int dh = th * g.dstep_h, dw = tw * g.dstep_w;
#pragma unroll
            for (int j = 0; j < A_LD; ++j) {
                const int ih = a_ih0[j] + dh, iw = a_iw0[j] + dw;
                const bool ok = kvalid && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                // a_nbase carries this lane's chunk within a 16-k step; here the chunk within the tap is cc instead
                ra[j] = ldq(a_rsrc, ok ? a_nbase[j] + (unsigned)((ih * W + iw) * a_pitch + (cc - chunk) * 4) * 4u : OOB, 0);
            }
            if (!PRE) {
#pragma unroll
                for (int j = 0; j < B_LD; ++j) rb[j] = ldq(b_rsrc, kvalid ? b_off[j] : OOB, ks * 64);
            }
        }
    };
    // x = h + m + l with h, m, l the three successive 8-bit significand slices (truncation; every step exact)
    auto split_store = [&](const f32x4& v, char* dst, const int split_stride) {
        unsigned h[4], m[4], l[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned u = __float_as_uint(v[q]);
            h[q] = u & 0xffff0000u;
            const float r1 = v[q] - __uint_as_float(h[q]);
            m[q] = __float_as_uint(r1) & 0xffff0000u;
            const float r2 = r1 - __uint_as_float(m[q]);
            l[q] = __float_as_uint(r2);
        }
        uint2 ph, pm, pl;
        ph.x = (h[0] >> 16) | h[1]; ph.y = (h[2] >> 16) | h[3];
        pm.x = (m[0] >> 16) | m[1]; pm.y = (m[2] >> 16) | m[3];
        pl.x = (l[0] >> 16) | (l[1] & 0xffff0000u); pl.y = (l[2] >> 16) | (l[3] & 0xffff0000u);
        *reinterpret_cast<uint2*>(dst) = ph;
        *reinterpret_cast<uint2*>(dst + split_stride) = pm;
        *reinterpret_cast<uint2*>(dst + 2 * split_stride) = pl;
    };
    float rowss[BM / 32];            // only the first A_LD entries are used (tile_epilogue reads them in staging layout)
#pragma unroll
    for (int j = 0; j < BM / 32; ++j) rowss[j] = 0.f;
    const bool unit_w = !PRE && (p.e.flags & BCOS_EPI_UNIT_NORM_W) != 0;      // see tile_body
    float colss[B_LD];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) colss[j] = 0.f;
    auto store_step = [&](const f32x4 (&ra)[A_LD], const f32x4 (&rb)[B_LD], int buf) {
        char* base = lds + buf * BUF;
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            if (NORM) {
                rowss[j] = fmaf(ra[j][0], ra[j][0], rowss[j]);
                rowss[j] = fmaf(ra[j][1], ra[j][1], rowss[j]);
                rowss[j] = fmaf(ra[j][2], ra[j][2], rowss[j]);
                rowss[j] = fmaf(ra[j][3], ra[j][3], rowss[j]);
            }
            split_store(ra[j], base + (r0 + 64 * j) * X3_ROW + ((chunk * 8) ^ (((r0 >> 3) & 1) << 4)), A_SPLIT);
        }
        if (!PRE)
#pragma unroll
        for (int j = 0; j < B_LD; ++j)
            if (!B_HALF || r0 < BN) {
                split_store(rb[j], base + B_BASE + (r0 + 64 * j) * X3_ROW + ((chunk * 8) ^ (((r0 >> 3) & 1) << 4)), B_SPLIT);
                if (unit_w) colss[j] = fmaf(rb[j][0], rb[j][0], fmaf(rb[j][1], rb[j][1], fmaf(rb[j][2], rb[j][2], fmaf(rb[j][3], rb[j][3], colss[j]))));
            }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag_row = lane & 31, frag_half = lane >> 5;
    const int frag_off = (frag_half * 16) ^ (((frag_row >> 3) & 1) << 4);
    const int a_frag = (wave_m * WM + frag_row) * X3_ROW + frag_off;
    const int b_frag = B_BASE + (wave_n * WN + frag_row) * X3_ROW + frag_off;

    // Pre-split weights (bcos_split_weights): [32-column tile][16-k step][plane][lane][8 bf16] = exactly the B fragment a
    // wavefront feeds to v_mfma_f32_32x32x16_bf16, 1 KB per (tile, step, plane).  A step's B operand is then six
    // perfectly coalesced 16-byte loads per lane straight into registers: no conversion, no LDS store, no LDS read,
    // and the weights never take part in the workgroup barrier.  Offsets are scalar (voffset = lane * 16).
    struct BFrag { bf16x8 b[3][TN]; };
    const __amdgpu_buffer_rsrc_t b3_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wt3), 0, p.wt3_bytes, 0x00020000);
    const int b3_tile0 = __builtin_amdgcn_readfirstlane((n0 + wave_n * WN) >> 5);
    auto load_bfrag = [&](int ks, BFrag& f) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) {
                const int soff = (((b3_tile0 + j) * nk + ks) * 3 + sp) * 1024;
                f.b[sp][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(b3_rsrc, lane * 16, soff, 0));
            }
    };
    auto mma_step = [&](int buf, const BFrag& pf) {
        const char* base = lds + buf * BUF;
        bf16x8 af[3][TM], bf[3][TN];
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[sp][i] = *reinterpret_cast<const bf16x8*>(base + a_frag + sp * A_SPLIT + i * 32 * X3_ROW);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (PRE) bf[sp][j] = pf.b[sp][j];
                else bf[sp][j] = *reinterpret_cast<const bf16x8*>(base + b_frag + sp * B_SPLIT + j * 32 * X3_ROW);
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                // smallest terms first
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bf[0][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[2][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[1][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
            }
    };

    // buf[ks & 1] holds step ks.  Three-stage software pipeline: global loads of step ks+2 are issued into the free
    // register set; step ks+1 -- loaded one step ago -- is split and written to the other LDS buffer next to the MFMAs
    // of step ks.  Two steps per loop trip so that the register sets (and, with pre-split weights, the two B fragment
    // sets: the fragments of step ks+1 are loaded during step ks) swap roles without copies; the accumulators stay in
    // AGPRs; no branches between the MFMAs and the split/stores.
    f32x4 ra0[A_LD], rb0[B_LD], ra1[A_LD], rb1[B_LD];
    BFrag bf0, bf1;
    load_step(0, ra0, rb0);
    if (PRE) load_bfrag(0, bf0);
    store_step(ra0, rb0, 0);
    if (nk > 1) load_step(1, ra0, rb0);
    __syncthreads();
    int ks = 0;
    for (; ks + 3 < nk; ks += 2) {
        load_step(ks + 2, ra1, rb1);
        if (PRE) load_bfrag(ks + 1, bf1);
        mma_step(0, bf0);
        store_step(ra0, rb0, 1);
        __syncthreads();
        load_step(ks + 3, ra0, rb0);
        if (PRE) load_bfrag(ks + 2, bf0);
        mma_step(1, bf1);
        store_step(ra1, rb1, 0);
        __syncthreads();
    }
    // the last 1-3 steps (ks is even here)
    auto tail_step = [&](int k, f32x4 (&la)[A_LD], f32x4 (&lb)[B_LD], const f32x4 (&sa)[A_LD], const f32x4 (&sb)[B_LD],
                         const BFrag& fc, BFrag& fn) {
        if (k + 2 < nk) load_step(k + 2, la, lb);
        if (PRE && k + 1 < nk) load_bfrag(k + 1, fn);
        mma_step(k & 1, fc);
        if (k + 1 < nk) store_step(sa, sb, (k + 1) & 1);
        __syncthreads();
    };
    if (ks < nk) tail_step(ks, ra1, rb1, ra0, rb0, bf0, bf1);
    if (ks + 1 < nk) tail_step(ks + 1, ra0, rb0, ra1, rb1, bf1, bf0);
    if (ks + 2 < nk) tail_step(ks + 2, ra1, rb1, ra0, rb0, bf0, bf1);
    if (unit_w) {
        float* sCol = epi_col_table<BM, BN, WAVES_M>(smem);
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            float t = colss[j];
            t += __shfl_xor(t, 1); t += __shfl_xor(t, 2);
            if (chunk == 0 && (!B_HALF || r0 < BN)) sCol[r0 + 64 * j] = 1.0f / sqrtf(t);
        }
    }
    tile_epilogue<BM, BN, WAVES_M, WAVES_N, NORM, false>(p, smem, acc, nullptr, NORM ? rowss : nullptr, nullptr, m0, n0, tile_n);
}

// ---------------------------------------------------------------------------------------------------------------
// Split-f16 main loop ("f16x2"): every fp32 operand x is written as x * 2^e = h + l with h, l fp16 (round to nearest:
// |x 2^e - h - l| <= 2^-22 |x|, fp32-rounding class) and a*b is evaluated as  l_a h_b + h_a l_b + h_a h_b  on
// v_mfma_f32_32x32x16_f16: 3 matrix instructions per 16 k instead of 6 (bf16x3) or 8 x 64 cycles (fp32); products of fp16
// numbers are exact in fp32, accumulation is fp32, the dropped l_a l_b term is <= 2^-22 |a b|.
// fp16 has a 5-bit exponent, so the operands are brought into range by POWER-OF-TWO scales that are exact to apply and
// to undo: one per GEMM row (from the per-pixel max |A| side tensor `a_absmax` that the producer of A emitted: the row's
// max over its taps has its leading bit moved to 2^14) and one per weight row (static, stored with the pre-split image).
// Elements within 2^-17 of their row's max keep full precision; below that the absolute error stays <= 2^-40 of the max.
// Structure: BM x BN x 16 steps through two LDS buffers as in tile_body_x3, but
//   * the weights arrive pre-split in MFMA fragment order ([32-col tile][16-k step][plane][lane][8 f16], 1 KB blocks) and
//     are copied verbatim into LDS by whole wavefronts (coalesced 16-byte loads, conflict-free 16-byte stores and fragment
//     reads): the four waves share one copy instead of each fetching its own fragments through the texture path;
//   * the activation split costs ~4 VALU per element (scale, 2 converts, 1 mixed FMA) instead of ~9;
//   * tiles of 256 x 128 / 128 x 256 (8 accumulator tiles per wave) halve the bytes staged per MFMA.
constexpr int LVL_STEP = 16;       // operand-scale ladder of the per-image scales (tile_body_p): a row is computed with a scale within 2^LVL_STEP of
                                   // its own maximum.  The split's absolute error is 2^-25 in scaled units and a row's maximum sits at >= 2^(14 - LVL_STEP)
                                   // there: every element of the row is off by <= 2^-23 of the row's maximum, ~1e-7 of ||patch|| ||w|| in the contraction
                                   // (measured: tests/test_gpu_parity.py::test_patch_loop_dynamic_range_inside_an_image); 12 costs 1.1 % of the ResNet-50
                                   // step on smooth synthetic images (level contours cross many tiles of the 56^2 / 112^2 gradient launches), 16 nothing
constexpr int H2_MAX_TAPS = 16;    // channel-chunk-major K walk for up to this many taps (offset table: taps x BM x 4 bytes of LDS)
#ifndef H2_KO
#define H2_KO 0                   // development knock-outs (timing only, wrong results): 1 no MFMA, 2 no split VALU, 4 no global loads in the loop, 8 no fragment reads
#endif
BCOS_DEV_SWITCH(H2_KO, 0);
#ifndef H2_MFMA_ORDER
#define H2_MFMA_ORDER 0           // 0 = chosen by tile shape (see mma_step), 1 = accumulator-major, 2 = product-major
#endif
#ifndef H2_PIPE_SMALL
#define H2_PIPE_SMALL 2           // pipeline of the <= 4-accumulator tiles: 2 = one 16-k step per barrier, 3 = two
#endif
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
#ifndef H2_PRIO
#define H2_PRIO 0                 // wave priority of the split-f16 main loops (s_setprio): 0 = never raised, 1 = raised around the matrix instructions of a step,
                                  // 2 = raised for the whole K loop (prologue and epilogue at priority 0: the other workgroup of the CU is usually in one of them)
#endif
#ifndef H2_MIX_SPLIT
#define H2_MIX_SPLIT 1            // 1 = the (h, l) split as two v_fma_mix{lo,hi}_f16 per element (split4_f16), 0 = the scalar expressions left to the compiler (rounds 2-5); same bits
#endif
// (h, l) f16 split of fp32 values times a power of two `s`, two values into the two halves of one register each:
//   h = f16(x s),  l = f16(x s - h).
// x s and x s - h are exact in fp32 (s is a power of two, h holds the leading 11 bits of x s), so each result is rounded ONCE, to f16
// -- bit for bit what (_Float16)(x * s) and (_Float16)(x * s - (float)h) give.  Written as the instructions themselves: v_fma_mixlo_f16 /
// v_fma_mixhi_f16 take fp32 (or, per source, f16-half) operands, compute the fma in fp32 and write the f16 result into the low / high
// half of the destination -- two vector instructions per element and no packing.  Left to the compiler the same expressions become
// multiply / convert / convert back / subtract / convert / pack over register PAIRS (v_pk_mul_f32, v_pk_fma_f32 ...: ~6 instructions
// per element plus the moves that build the pairs, and packed fp32 instructions beside matrix instructions cost more than their issue slot).
// One 16-byte piece (four fp32 values) -> two registers of h halves and two of l halves; SUMSQ: acc += x^2 of the four values (one
// v_fmac_f32 each: the compiler pairs neighbouring chains into v_pk_fma_f32 behind two moves).  ONE asm statement, so that nothing is
// padded between its instructions; inside it every instruction that reads a register half written by a v_fma_mix{lo,hi}_f16 is at
// least two instructions behind that write (gfx940+: a result written with a destination half-select needs one wait state before a
// vector instruction reads it).
template <bool SUMSQ>
__device__ __forceinline__ void split4_f16(const f32x4& x, const float s, unsigned& h01, unsigned& h23, unsigned& l01, unsigned& l23, float& acc) {
    unsigned a, b, c, d;
    if constexpr (SUMSQ) {
        asm("v_fma_mixlo_f16 %0, %5, %9, 0\n\t"
            "v_fma_mixlo_f16 %1, %7, %9, 0\n\t"
            "v_fma_mixhi_f16 %0, %6, %9, 0\n\t"
            "v_fma_mixhi_f16 %1, %8, %9, 0\n\t"
            "v_fmac_f32 %4, %5, %5\n\t"
            "v_fma_mixlo_f16 %2, %5, %9, -%0 op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixlo_f16 %3, %7, %9, -%1 op_sel_hi:[0,0,1]\n\t"
            "v_fmac_f32 %4, %6, %6\n\t"
            "v_fma_mixhi_f16 %2, %6, %9, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
            "v_fmac_f32 %4, %7, %7\n\t"
            "v_fma_mixhi_f16 %3, %8, %9, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
            "v_fmac_f32 %4, %8, %8"
            : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "+v"(acc)
            : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(s));
    } else {
        asm("v_fma_mixlo_f16 %0, %4, %8, 0\n\t"
            "v_fma_mixlo_f16 %1, %6, %8, 0\n\t"
            "v_fma_mixhi_f16 %0, %5, %8, 0\n\t"
            "v_fma_mixhi_f16 %1, %7, %8, 0\n\t"
            "v_fma_mixlo_f16 %2, %4, %8, -%0 op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixlo_f16 %3, %6, %8, -%1 op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %2, %5, %8, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %3, %7, %8, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
            : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
            : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(s));
    }
    h01 = a; h23 = b; l01 = c; l23 = d;
}
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// pipeline of a split-f16 tile: 1 for the 8-accumulator tiles, H2_PIPE_SMALL otherwise -- except the 32-column gradient tiles
// (the depth-to-space stem gradient: 3 matrix instructions per 16-k step and wave), which take two sub-steps per barrier
// (same-node A/B of that launch: 1.38 -> 1.26 ms; the 64- and 128-column tiles lose with it)
template <int BM, int BN, int WAVES_M, int WAVES_N, bool NORM>
constexpr int h2_pipe() {
    return (BM / WAVES_M) * (BN / WAVES_N) > 64 * 64 ? 1 : ((BN == 32 && !NORM && BM <= 128) ? 3 : H2_PIPE_SMALL);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool NORM, int NT = NTHREADS,
          int PIPE = h2_pipe<BM, BN, WAVES_M, WAVES_N, NORM>()>
__device__ __forceinline__ void tile_body_h2(const KArgs& p, float* smem, const int m0, const int n0, const int tile_n) {
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NW = NT / 64;                     // wavefronts of the workgroup (4, or 8 for the 512-thread variant)
    constexpr int RP = NT / 4;                      // staging rows per pass
    static_assert(WAVES_M * WAVES_N == NW && BM % RP == 0, "wave layout");
    constexpr int A_LD = BM / RP;                   // float4 loads per thread per 16-k step
    constexpr int NBLK = (BN / 32) * 2;             // 1-KB fragment blocks of B per step: (32-column tile, plane)
    constexpr int B_LD = (NBLK + NW - 1) / NW;      // blocks per wave
    char* lds = reinterpret_cast<char*>(smem);
    constexpr int A_SPLIT = BM * X3_ROW, B_BASE = 2 * A_SPLIT;
    constexpr int BUF = B_BASE + NBLK * 1024;
    constexpr int H2_NBUF = PIPE == 3 ? 4 : 2;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    const int chunk = tid & 3;       // 16-byte chunk (4 k) within the 16-k step
    const int r0 = tid >> 2;         // staging row 0..RP-1 (+RP per pass)
    const bcos_tapconv_geom& g = p.g;
    const int H = g.H, W = g.W;
    const int a_pitch = g.a_pitch;

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wt2), 0, p.wt2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t m_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(p.a_absmax), 0, p.absmax_bytes, 0x00020000);
    unsigned a_nbase[A_LD];          // byte offset of the row's image (+ this lane's 16-byte chunk)
    int a_ih0[A_LD], a_iw0[A_LD];
    float a_scale[A_LD], a_inv[A_LD];
    {
        // per-row operand scale: max over the row's taps of the per-pixel max |A|; the 4 chunk-lanes of a row share the taps
        unsigned rmax[A_LD];
        int pix0[A_LD];
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            const int m = m0 + r0 + RP * j;
            rmax[j] = 0u;
            if (m < p.M) {
                const int n = m / p.PQ;
                const int rem = m - n * p.PQ;
                const int i = rem / g.Q;
                const int jj = rem - i * g.Q;
                a_nbase[j] = ((unsigned)n * H * W * a_pitch + chunk * 4) * 4u;
                pix0[j] = n * H * W;
                a_ih0[j] = i * g.in_sh + g.dh0;
                a_iw0[j] = jj * g.in_sw + g.dw0;
            } else {
                a_nbase[j] = 0;
                pix0[j] = 0;
                a_ih0[j] = -(1 << 28);
                a_iw0[j] = -(1 << 28);
            }
        }
        const int ntaps = g.TH * g.TW;
        for (int t = chunk; t < ntaps; t += 4) {
            const int th = t / g.TW, tw = t - th * g.TW;
            const int dh = th * g.dstep_h, dw = tw * g.dstep_w;
#pragma unroll
            for (int j = 0; j < A_LD; ++j) {
                const int ih = a_ih0[j] + dh, iw = a_iw0[j] + dw;
                const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                const unsigned v = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(m_rsrc, ok ? (unsigned)(pix0[j] + ih * W + iw) * 4u : OOB, 0, 0);
                rmax[j] = max(rmax[j], v);
            }
        }
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            unsigned v = rmax[j];
            v = max(v, (unsigned)__shfl_xor((int)v, 1));
            v = max(v, (unsigned)__shfl_xor((int)v, 2));
            unsigned E = v >> 23;                  // biased exponent of the row max (the bit patterns carry no sign)
            E = E < 15u ? 15u : E;
            a_scale[j] = __uint_as_float((268u - E) << 23);     // max * scale in [2^14, 2^15)
            a_inv[j] = __uint_as_float((E - 14u) << 23);
        }
    }

    const int nk = (p.nchunks + 3) / 4;
    const bool uniform = (g.C % X3_BK) == 0;
    // C = 4 or 8 (the 6 -> 8 channel network input): a 16-k step spans 16 / C taps and every lane walks its own tap
    // coordinates incrementally instead of dividing its chunk index by runtime values each step
    const bool small_c = g.C == 4 || g.C == 8;
    const int tps = small_c ? 16 / g.C : 1;                                // taps per step
    const int l_cc = small_c ? (g.C == 8 ? (chunk & 1) : 0) : 0;            // this lane's chunk inside its tap
    int l_th = 0, l_tw = 0;                                                 // this lane's tap (small_c)
    if (small_c) {
        const int sub = g.C == 8 ? (chunk >> 1) : chunk;
        l_th = sub / g.TW;
        l_tw = sub - l_th * g.TW;
    }
    // Multi-tap launches (3x3 convolutions and their input gradients) walk K channel-chunk-major: step ks covers tap
    // ks % taps of the 16-channel chunk ks / taps (the pre-split weight image is stored in the same order,
    // bcos_split_weights_f16x2_conv).  A workgroup then re-reads the same ~160 pixels x 64 B for nine consecutive steps -- they
    // stay in L1 / L2 -- instead of streaming its whole row panel (164 KB at 256 channels) once per tap through a 4 MB L2
    // that 64 co-resident workgroups share.  Per-(tap, row) source offsets live in LDS behind the staging buffers.
    const int ntaps = g.TH * g.TW;
    const bool kmajor = ntaps > 1 && ntaps <= H2_MAX_TAPS && uniform;
    unsigned* s_tapoff = reinterpret_cast<unsigned*>(lds + H2_NBUF * BUF);         // [tap][BM] byte offsets (without the lane's chunk)
    if (kmajor) {
        for (int t = chunk; t < ntaps; t += 4) {
            const int th = t / g.TW, tw = t - th * g.TW;
            const int dh = th * g.dstep_h, dw = tw * g.dstep_w;
#pragma unroll
            for (int j = 0; j < A_LD; ++j) {
                const int ih = a_ih0[j] + dh, iw = a_iw0[j] + dw;
                const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                s_tapoff[t * BM + r0 + RP * j] = ok ? a_nbase[j] - chunk * 16u + (unsigned)((ih * W + iw) * a_pitch) * 4u : OOB;
            }
        }
    }
    int s_tap = 0;
    int s_cc = 0, s_th = 0, s_tw = 0;
    unsigned a_cur[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) a_cur[j] = OOB;
    auto ldq = [](const __amdgpu_buffer_rsrc_t& r, unsigned voff, int soff) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
    };
    // B: wave w copies blocks w, w + 4, ... of the step (block = (32-column tile c, plane sp) = 1 KB, lane-linear)
    const int b_tile0 = n0 >> 5;
    auto load_step = [&](int ks, f32x4 (&ra)[A_LD], f32x4 (&rb)[B_LD]) {
        if (kmajor) {
#pragma unroll
            for (int j = 0; j < A_LD; ++j) ra[j] = ldq(a_rsrc, s_tapoff[s_tap * BM + r0 + RP * j] + chunk * 16u, s_cc * 16);
            if (++s_tap == ntaps) { s_tap = 0; s_cc += 4; }
        } else if (uniform) {
            if (s_cc == 0) {
                const int dh = s_th * g.dstep_h, dw = s_tw * g.dstep_w;
#pragma unroll
                for (int j = 0; j < A_LD; ++j) {
                    const int ih = a_ih0[j] + dh, iw = a_iw0[j] + dw;
                    const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                    a_cur[j] = ok ? a_nbase[j] + (unsigned)((ih * W + iw) * a_pitch) * 4u : OOB;
                }
            }
#pragma unroll
            for (int j = 0; j < A_LD; ++j) ra[j] = ldq(a_rsrc, a_cur[j], s_cc * 16);
            s_cc += 4;
            if (s_cc == p.cpt) {
                s_cc = 0;
                if (++s_tw == g.TW) { s_tw = 0; ++s_th; }
            }
        } else if (small_c) {
            const bool kvalid = l_th < g.TH;
            const int dh = l_th * g.dstep_h, dw = l_tw * g.dstep_w;
#pragma unroll
            for (int j = 0; j < A_LD; ++j) {
                const int ih = a_ih0[j] + dh, iw = a_iw0[j] + dw;
                const bool ok = kvalid && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                ra[j] = ldq(a_rsrc, ok ? a_nbase[j] + (unsigned)((ih * W + iw) * a_pitch + (l_cc - chunk) * 4) * 4u : OOB, 0);
            }
            l_tw += tps;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (l_tw >= g.TW) { l_tw -= g.TW; ++l_th; }
        } else {
            const int q = ks * 4 + chunk;
            const bool kvalid = q < p.nchunks;
            const int tap = q / p.cpt;
            const int cc = q - tap * p.cpt;
            const int th = tap / g.TW;
            const int tw = tap - th * g.TW;
            const int dh = th * g.dstep_h, dw = tw * g.dstep_w;
#pragma unroll
            for (int j = 0; j < A_LD; ++j) {
                const int ih = a_ih0[j] + dh, iw = a_iw0[j] + dw;
                const bool ok = kvalid && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                ra[j] = ldq(a_rsrc, ok ? a_nbase[j] + (unsigned)((ih * W + iw) * a_pitch + (cc - chunk) * 4) * 4u : OOB, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            const int blk = wave + NW * j;
            if (NBLK % NW == 0 || blk < NBLK) {
                const int soff = ((b_tile0 + (blk >> 1)) * nk + ks) * 2048 + (blk & 1) * 1024;
                rb[j] = ldq(b_rsrc, lane * 16, soff);
            }
        }
    };
    float rowss[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) rowss[j] = 0.f;
    auto store_step = [&](const f32x4 (&ra)[A_LD], const f32x4 (&rb)[B_LD], int buf) {
        char* base = lds + buf * BUF;
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            if (NORM) {
                rowss[j] = fmaf(ra[j][0], ra[j][0], rowss[j]);
                rowss[j] = fmaf(ra[j][1], ra[j][1], rowss[j]);
                rowss[j] = fmaf(ra[j][2], ra[j][2], rowss[j]);
                rowss[j] = fmaf(ra[j][3], ra[j][3], rowss[j]);
            }
            f16x4 h, l;
#if H2_KO & 2
            h = __builtin_bit_cast(f16x4, __builtin_shufflevector(ra[j], ra[j], 0, 1));
            l = __builtin_bit_cast(f16x4, __builtin_shufflevector(ra[j], ra[j], 2, 3));
#else
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float xs = ra[j][q] * a_scale[j];
                const _Float16 hh = (_Float16)xs;
                h[q] = hh;
                l[q] = (_Float16)(xs - (float)hh);
            }
#endif
            char* dst = base + (r0 + RP * j) * X3_ROW + ((chunk * 8) ^ (((r0 >> 3) & 1) << 4));
            *reinterpret_cast<f16x4*>(dst) = h;
            *reinterpret_cast<f16x4*>(dst + A_SPLIT) = l;
        }
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            const int blk = wave + NW * j;
            if (NBLK % NW == 0 || blk < NBLK) *reinterpret_cast<f32x4*>(base + B_BASE + blk * 1024 + lane * 16) = rb[j];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag_row = lane & 31, frag_half = lane >> 5;
    const int frag_off = (frag_half * 16) ^ (((frag_row >> 3) & 1) << 4);
    const int a_frag = (wave_m * WM + frag_row) * X3_ROW + frag_off;
    const int b_frag = B_BASE + (wave_n * TN) * 2048 + lane * 16;

    auto mma_step = [&](int buf) {
        const char* base = lds + buf * BUF;
        f16x8 af[2][TM], bf[2][TN];
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
#if H2_KO & 8
#pragma unroll
            for (int i = 0; i < TM; ++i) { af[sp][i] = f16x8{}; asm volatile("" : "+v"(af[sp][i])); }
#pragma unroll
            for (int j = 0; j < TN; ++j) { bf[sp][j] = f16x8{}; asm volatile("" : "+v"(bf[sp][j])); }
#else
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[sp][i] = *reinterpret_cast<const f16x8*>(base + a_frag + sp * A_SPLIT + i * 32 * X3_ROW);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[sp][j] = *reinterpret_cast<const f16x8*>(base + b_frag + j * 2048 + sp * 1024);
#endif
        }
        // smallest terms first.  Product-major order (consecutive matrix instructions write DIFFERENT accumulators: a dependent
        // v_mfma on the same accumulator waits for the previous one to retire) for the 2- and 4-accumulator wave tiles; the
        // 8-accumulator tiles (128 x 256) measured 3-5 % faster accumulator-major (same-node A/B; the 4-accumulator tiles 3-6 %
        // slower).  Per accumulator the three products arrive in the same order either way: identical bits.
        // H2_MFMA_ORDER 0 = by tile shape, 1 = accumulator-major everywhere, 2 = product-major everywhere
        constexpr bool ACC_MAJOR = H2_MFMA_ORDER == 1 || (H2_MFMA_ORDER == 0 && TM * TN >= 8);
        if constexpr (!ACC_MAJOR) {
#pragma unroll
            for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                    {
#if H2_KO & 1
                        asm volatile("" :: "v"(af[pr == 0 ? 1 : 0][i]), "v"(bf[pr == 1 ? 1 : 0][j]));
#else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[pr == 0 ? 1 : 0][i], bf[pr == 1 ? 1 : 0][j], acc[i][j], 0, 0, 0);
#endif
                    }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][i], bf[0][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][i], bf[1][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
                }
        }
    };

    if (kmajor) __syncthreads();        // the (tap, row) offset table is complete
    if constexpr (PIPE == 2) {
        // same three-stage pipeline as tile_body_x3: loads of step ks+2 in flight, step ks+1 converted and written to the
        // other LDS buffer next to the MFMAs of step ks; two steps per trip so the register sets swap roles without copies
        f32x4 ra0[A_LD], rb0[B_LD], ra1[A_LD], rb1[B_LD];
        load_step(0, ra0, rb0);
        store_step(ra0, rb0, 0);
        if (nk > 1) load_step(1, ra0, rb0);
        __syncthreads();
        int ks = 0;
        for (; ks + 3 < nk; ks += 2) {
#if !(H2_KO & 4)
            load_step(ks + 2, ra1, rb1);
#endif
            mma_step(0);
            store_step(ra0, rb0, 1);
            __syncthreads();
#if !(H2_KO & 4)
            load_step(ks + 3, ra0, rb0);
#endif
            mma_step(1);
            store_step(ra1, rb1, 0);
            __syncthreads();
        }
        auto tail_step = [&](int k, f32x4 (&la)[A_LD], f32x4 (&lb)[B_LD], const f32x4 (&sa)[A_LD], const f32x4 (&sb)[B_LD]) {
            if (k + 2 < nk) load_step(k + 2, la, lb);
            mma_step(k & 1);
            if (k + 1 < nk) store_step(sa, sb, (k + 1) & 1);
            __syncthreads();
        };
        if (ks < nk) tail_step(ks, ra1, rb1, ra0, rb0);
        if (ks + 1 < nk) tail_step(ks + 1, ra0, rb0, ra1, rb1);
        if (ks + 2 < nk) tail_step(ks + 2, ra1, rb1, ra0, rb0);
    } else if constexpr (PIPE == 4) {
        // three staging register sets: the loads of step ks+3 are issued at step ks and have three steps to arrive
        f32x4 ra0[A_LD], rb0[B_LD], ra1[A_LD], rb1[B_LD], ra2[A_LD], rb2[B_LD];
        load_step(0, ra0, rb0);
        store_step(ra0, rb0, 0);
        if (nk > 1) load_step(1, ra1, rb1);
        if (nk > 2) load_step(2, ra2, rb2);
        __syncthreads();
        auto body = [&](int k, f32x4 (&la)[A_LD], f32x4 (&lb)[B_LD], const f32x4 (&sa)[A_LD], const f32x4 (&sb)[B_LD]) {
            if (k + 3 < nk) load_step(k + 3, la, lb);
            mma_step(k & 1);
            if (k + 1 < nk) store_step(sa, sb, (k + 1) & 1);
            __syncthreads();
        };
        for (int ks = 0; ks < nk; ks += 3) {
            body(ks, ra0, rb0, ra1, rb1);
            if (ks + 1 < nk) body(ks + 1, ra1, rb1, ra2, rb2);
            if (ks + 2 < nk) body(ks + 2, ra2, rb2, ra0, rb0);
        }
    } else if constexpr (PIPE == 3) {
        // two 16-k sub-steps per barrier (four LDS sub-buffers): the loads of macro-step m+1 are issued before the 24 MFMAs
        // of macro-step m and written to the other buffer pair after them -- the same latency budget as the pipeline above
        // with half the barriers
        f32x4 ra0[A_LD], rb0[B_LD], ra1[A_LD], rb1[B_LD];
        const int nm = (nk + 1) / 2;
        load_step(0, ra0, rb0);
        if (nk > 1) load_step(1, ra1, rb1);
        store_step(ra0, rb0, 0);
        if (nk > 1) store_step(ra1, rb1, 1);
        __syncthreads();
        for (int m = 0; m < nm; ++m) {
            const int cur = (m & 1) * 2, nxt = cur ^ 2;
            const bool n0_ = 2 * m + 2 < nk, n1_ = 2 * m + 3 < nk;
            if (n0_) load_step(2 * m + 2, ra0, rb0);
            if (n1_) load_step(2 * m + 3, ra1, rb1);
            mma_step(cur);
            if (2 * m + 1 < nk) mma_step(cur + 1);
            if (n0_) store_step(ra0, rb0, nxt);
            if (n1_) store_step(ra1, rb1, nxt + 1);
            __syncthreads();
        }
    } else {
        // 8-accumulator tiles leave room for ONE staging register set: the loads of step ks+2 are issued right after the
        // set has been written to LDS (end of step ks) and have one whole step -- the co-resident workgroup's MFMAs
        // included -- to arrive
        f32x4 ra[A_LD], rb[B_LD];
        load_step(0, ra, rb);
        store_step(ra, rb, 0);
        if (nk > 1) load_step(1, ra, rb);
        __syncthreads();
        for (int ks = 0; ks < nk; ++ks) {
            mma_step(ks & 1);
            if (ks + 1 < nk) store_step(ra, rb, (ks + 1) & 1);
            if (ks + 2 < nk) load_step(ks + 2, ra, rb);
            __syncthreads();
        }
    }
    tile_epilogue<BM, BN, WAVES_M, WAVES_N, NORM, true, NT>(p, smem, acc, nullptr, NORM ? rowss : nullptr, a_inv, m0, n0, tile_n);
}

// ---- split-f16 contraction, ASYNCHRONOUS staging (round 3) ---------------------------------------------------------------------
// Same arithmetic as tile_body_h2 -- the same split of every operand element, the same three products per accumulator in the
// same order over the same K walk, the same partial sums of the patch norm: results are bit-identical -- with the operands moved
// by the LDS-DMA path instead of through registers:
//   * A stays fp32 on its way into LDS: `buffer_load_dwordx4 ... lds` writes 64 lanes x 16 bytes straight into the slot (lanes
//     whose tap is out of the image fail the buffer bounds check and land as zeros: scripts/probe/lds_dma_probe.hip); the slot
//     is [BM rows][16 k] fp32 = 64 B rows whose four 16-byte pieces are permuted by (row >> 2) & 3 ON THE SOURCE SIDE (the DMA
//     writes lane-linear), which makes the fragment reads (two ds_read_b128 per 32 x 16 fragment) bank-conflict free;
//   * the fp32 -> (h, l) f16 split happens on the fragment, in the wave that feeds it to the matrix pipe.  Waves are laid out
//     WAVES_M x 1 wherever the tile allows, so every A element is still converted exactly once per tile and the conversion
//     (~3 VALU per element with the packed f32 / cvt_pk forms) sits in the shadow of that wave's own matrix instructions;
//   * B (pre-split weights in fragment order) is DMA'd verbatim, one 1-KB block per wave instruction;
//   * no staging registers, no ds_write, no VALU on the load path: a ring of three or four slots (d_nslot), the loads of step ks + 2 / ks + 3
//     are issued at the top of step ks and have two / three steps of matrix work to land; ONE raw s_barrier per step, counted s_waitcnt vmcnt
//     (never 0 inside the loop).  The compiler keeps ds_reads clear of the DMA queue as long as all LDS is one array.
#ifndef D_NSLOT
#define D_NSLOT 0                 // 0 = per tile configuration (d_nslot), 3 / 4 = that many ring slots everywhere
#endif
#ifndef D_KO
#define D_KO 0                    // development knock-outs (timing only, wrong results): 1 no DMA issue in the steady loop, 2 no split of the next A rows
#endif
BCOS_DEV_SWITCH(D_KO, 0);
#ifndef D_DEAL_DMA
#define D_DEAL_DMA 0              // 1 = issue the DMA pieces one by one behind the step's matrix instructions instead of together behind the barrier
#endif
#ifndef D_EARLY
#define D_EARLY 1                 // 1 = the prologue's DMA pieces go out between the loads of the operand maxima and their use, no workgroup barrier ahead of the loop (every walk but the channel-chunk-major one); 0 = behind the scales and a barrier (round 3)
#endif
BCOS_DEV_SWITCH(D_EARLY, 1);
#ifndef D_A_AUX
#define D_A_AUX 0                 // cache policy bits of the A operand's LDS-DMA loads (2 = non-temporal: measured in round 4)
#endif
BCOS_DEV_SWITCH(D_A_AUX, 0);
#ifndef D_SCHED
#define D_SCHED 1                 // 1 = pin the issue order of a step's fragment reads / matrix / vector instructions (sched_group_barrier)
#endif
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#define BCOS_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int BM, int BN>
constexpr int d_slot_bytes() { return BM * 64 + (BN / 32) * 2048; }
// Ring depth of a tile configuration: FOUR slots (three steps of DMA in flight behind the step being multiplied) where four slots, the row
// scales and the largest tap table still leave room for two workgroups per CU (80 KB each) -- the <= 128-column tiles of 128 rows and the
// 256 x 32 tile; three slots otherwise (128 x 192 / 128 x 256 / 256 x 64).  The K loops of these launches wait on DMA latency, not on a pipe
// (a lone 128 x 128 workgroup runs a 16-k step of 12 matrix instructions in ~1 500 cycles, profiles/r04_phase_times.txt): a deeper ring is
// more bytes in flight per workgroup.  D_NSLOT = 3 / 4 forces one depth for every configuration (development A/B).
template <int BM, int BN>
constexpr int d_nslot() {
    if (D_NSLOT != 0) return D_NSLOT;
    return 4 * d_slot_bytes<BM, BN>() + BM * 4 + 1024 + H2_MAX_TAPS * BM * 4 <= 80 * 1024 ? 4 : 3;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool NORM, int NT = NTHREADS>
__device__ __forceinline__ void tile_body_d(const KArgs& p, float* smem, const int m0, const int n0, const int tile_n) {
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NW = NT / 64;
    constexpr int RP = NT / 4;                      // rows covered by one DMA pass of the workgroup (16 per wave)
    static_assert(WAVES_M * WAVES_N == NW && BM % RP == 0, "wave layout");
    constexpr int A_LD = BM / RP;                   // A DMA instructions per wave per 16-k step (1 KB = 16 rows each)
    static_assert(WM == 16 * WAVES_N * A_LD, "a wave row-group's rows are loaded by its own waves");
    constexpr int NBLK = (BN / 32) * 2;             // 1-KB fragment blocks of B per step: (32-column tile, plane)
    constexpr int B_LD = (NBLK + NW - 1) / NW;      // B DMA instructions per wave per step (waves without a block issue a dummy: equal counts)
    constexpr int A_BYTES = BM * 64;
    constexpr int SLOT = d_slot_bytes<BM, BN>();
    constexpr bool PRIV = WAVES_N == 1;             // a wave's fragment rows are loaded by that wave alone: it converts them a step ahead
    char* lds = reinterpret_cast<char*>(smem);

    BCOS_PHASE_MARK(ph_t0);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    // DMA instruction j of wave (wave_m, wave_n) fills rows wave_m WM + 16 (wave_n A_LD + j) + [0, 16) of the slot: lane l
    // writes byte 16 l of that 1-KB piece = physical position l & 3 of row l >> 2, which holds LOGICAL chunk
    // (l & 3) ^ ((row >> 2) & 3) (4 k of the 16-k step; the permutation makes the fragment reads bank-conflict free)
    const int r0 = wave_m * WM + 16 * wave_n * A_LD + (lane >> 2);     // row of pass 0 (+16 per pass)
    const int chunk = (lane & 3) ^ ((lane >> 4) & 3);                  // (row >> 2) & 3 == (lane >> 4) & 3: the row bases are multiples of 16
    const bcos_tapconv_geom& g = p.g;
    const int H = g.H, W = g.W;
    const int a_pitch = g.a_pitch;

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wt2), 0, p.wt2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t m_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(p.a_absmax), 0, p.absmax_bytes, 0x00020000);
    unsigned a_nbase[A_LD];          // byte offset of the row's image (+ this lane's 16-byte chunk)
    int a_ih0[A_LD], a_iw0[A_LD];
    constexpr int NS = d_nslot<BM, BN>();           // ring slots
    static_assert(NS == 3 || NS == 4, "ring depth");
    float* s_scale = reinterpret_cast<float*>(lds + NS * SLOT);                          // [BM] row scales
    char* s_dummy = lds + NS * SLOT + BM * 4;                                           // 1 KB: target of the dummy B DMAs
    unsigned* s_tapoff = reinterpret_cast<unsigned*>(lds + NS * SLOT + BM * 4 + 1024);    // [tap][BM] byte offsets (without the lane's chunk)
    // rows of the tile this lane's DMA pieces cover: image base, first tap position (rows beyond M fail every bounds check)
    int pix0[A_LD], row_img[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        const int m = m0 + r0 + 16 * j;
        if (m < p.M) {
            const int n = m / p.PQ;
            const int rem = m - n * p.PQ;
            const int i = rem / g.Q;
            const int jj = rem - i * g.Q;
            a_nbase[j] = ((unsigned)n * H * W * a_pitch + chunk * 4) * 4u;
            pix0[j] = n * H * W;
            row_img[j] = n;
            a_ih0[j] = i * g.in_sh + g.dh0;
            a_iw0[j] = jj * g.in_sw + g.dw0;
        } else {
            a_nbase[j] = 0;
            pix0[j] = 0;
            row_img[j] = -1;
            a_ih0[j] = -(1 << 28);
            a_iw0[j] = -(1 << 28);
        }
    }
    const int nk = (p.nchunks + 3) / 4;
    const bool uniform = (g.C % X3_BK) == 0;
    const bool small_c = g.C == 4 || g.C == 8;      // see tile_body_h2
    const int tps = small_c ? 16 / g.C : 1;
    const int l_cc = small_c ? (g.C == 8 ? (chunk & 1) : 0) : 0;
    int l_th = 0, l_tw = 0;
    if (small_c) {
        const int sub = g.C == 8 ? (chunk >> 1) : chunk;
        l_th = sub / g.TW;
        l_tw = sub - l_th * g.TW;
    }
    const int ntaps = g.TH * g.TW;
    const bool kmajor = ntaps > 1 && ntaps <= H2_MAX_TAPS && uniform;      // channel-chunk-major K walk (see tile_body_h2)
    if (kmajor) {
        for (int t = chunk; t < ntaps; t += 4) {
            const int th = t / g.TW, tw = t - th * g.TW;
            const int dh = th * g.dstep_h, dw = tw * g.dstep_w;
#pragma unroll
            for (int j = 0; j < A_LD; ++j) {
                const int ih = a_ih0[j] + dh, iw = a_iw0[j] + dw;
                const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                s_tapoff[t * BM + r0 + 16 * j] = ok ? a_nbase[j] - chunk * 16u + (unsigned)((ih * W + iw) * a_pitch) * 4u : OOB;
            }
        }
    }
    int s_tap = 0;
    int s_cc = 0, s_th = 0, s_tw = 0;
    unsigned a_cur[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) a_cur[j] = OOB;
    const int b_tile0 = n0 >> 5;
    const int a_dst0 = (wave_m * WM + 16 * wave_n * A_LD) * 64;           // this wave's first 1-KB piece inside the A slot
    // issue the DMA of 16-k step ks into ring slot `slot`: the wave's A pieces first, then its B blocks (calls come in
    // increasing ks: the walk state advances with them)
    // K walk of the launch, fixed outside the loop: 0 channel-chunk-major taps (3x3 ...), 1 one tap per step (C % 16 == 0), 2 C = 4 / 8
    // (16 / C taps per step, every lane walks its own tap), 3 general
    const int walk = kmajor ? 0 : uniform ? 1 : small_c ? 2 : 3;
    // (calls come in increasing ks: the walk state advances with them)
    // walk_a: the source offsets of the wave's A pieces of the next step of the walk, handed to dma_a(j, voffset, soffset)
    auto walk_a = [&](auto walk_c, int ks, auto dma_a) {
        constexpr int WALK = decltype(walk_c)::value;
        if constexpr (WALK == 0) {
#pragma unroll
            for (int j = 0; j < A_LD; ++j) dma_a(j, s_tapoff[s_tap * BM + r0 + 16 * j] + chunk * 16u, s_cc * 16);
            if (++s_tap == ntaps) { s_tap = 0; s_cc += 4; }
        } else if constexpr (WALK == 1) {
            if (s_cc == 0) {
                const int dh = s_th * g.dstep_h, dw = s_tw * g.dstep_w;
#pragma unroll
                for (int j = 0; j < A_LD; ++j) {
                    const int ih = a_ih0[j] + dh, iw = a_iw0[j] + dw;
                    const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                    a_cur[j] = ok ? a_nbase[j] + (unsigned)((ih * W + iw) * a_pitch) * 4u : OOB;
                }
            }
#pragma unroll
            for (int j = 0; j < A_LD; ++j) dma_a(j, a_cur[j], s_cc * 16);
            s_cc += 4;
            if (s_cc == p.cpt) {
                s_cc = 0;
                if (++s_tw == g.TW) { s_tw = 0; ++s_th; }
            }
        } else if constexpr (WALK == 2) {
            const bool kvalid = l_th < g.TH;
            const int dh = l_th * g.dstep_h, dw = l_tw * g.dstep_w;
#pragma unroll
            for (int j = 0; j < A_LD; ++j) {
                const int ih = a_ih0[j] + dh, iw = a_iw0[j] + dw;
                const bool ok = kvalid && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                dma_a(j, ok ? a_nbase[j] + (unsigned)((ih * W + iw) * a_pitch + (l_cc - chunk) * 4) * 4u : OOB, 0);
            }
            l_tw += tps;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (l_tw >= g.TW) { l_tw -= g.TW; ++l_th; }
        } else {
            const int q = ks * 4 + chunk;
            const bool kvalid = q < p.nchunks;
            const int tap = q / p.cpt;
            const int cc = q - tap * p.cpt;
            const int th = tap / g.TW;
            const int tw = tap - th * g.TW;
            const int dh = th * g.dstep_h, dw = tw * g.dstep_w;
#pragma unroll
            for (int j = 0; j < A_LD; ++j) {
                const int ih = a_ih0[j] + dh, iw = a_iw0[j] + dw;
                const bool ok = kvalid && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                dma_a(j, ok ? a_nbase[j] + (unsigned)((ih * W + iw) * a_pitch + (cc - chunk) * 4) * 4u : OOB, 0);
            }
        }
    };
    auto piece_a = [&](int j, int slot_off, unsigned voff, int soff) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, BCOS_LDS_PTR(lds + slot_off + a_dst0 + j * 1024), 16, (int)voff, soff, 0, D_A_AUX);
    };
    auto issue_a = [&](auto walk_c, int ks, int slot_off) {
        walk_a(walk_c, ks, [&](int j, unsigned voff, int soff) { piece_a(j, slot_off, voff, soff); });
    };
    // B: wave w copies blocks w, w + NW, ... of the step (block = (32-column tile c, plane sp) = 1 KB, lane-linear)
    // (the source offset of block j at step 0 is fixed per wave: kept in scalar registers, a step adds ks * 2048 -- recomputed from
    //  the tile index it cost a scalar multiply and half a dozen scalar instructions per piece and step)
    int b_soff0[B_LD];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
        const int blk = wave + NW * j;
        b_soff0[j] = __builtin_amdgcn_readfirstlane((b_tile0 + (blk >> 1)) * nk * 2048 + (blk & 1) * 1024);
    }
    auto piece_b = [&](int j, int ks, int slot_off) {
        const int blk = wave + NW * j;
        if (NBLK % NW == 0 || blk < NBLK) {
            const int soff = b_soff0[j] + ks * 2048;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, BCOS_LDS_PTR(lds + slot_off + A_BYTES + blk * 1024), 16, lane * 16, soff, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, BCOS_LDS_PTR(s_dummy), 16, (int)OOB, 0, 0, 0);   // keeps every wave's DMA count equal
        }
    };
    auto issue_b = [&](int ks, int slot_off) {
#pragma unroll
        for (int j = 0; j < B_LD; ++j) piece_b(j, ks, slot_off);
    };
    constexpr int LA = A_LD, LB = B_LD;      // DMA instructions per wave and step: A pieces, then B blocks
    // The DMA of the first steps needs addresses only -- the row scales are applied when a fragment is converted -- so for every K walk
    // that derives its addresses from registers (1 x 1 layers, the small-channel stem, the general walk: everything but the
    // channel-chunk-major multi-tap walk, whose (tap, row) table lives in LDS) the prologue's pieces go out right behind the loads of
    // the operand maxima and ahead of their use (D_EARLY), and the tile enters its loop without a workgroup barrier (see no_barrier
    // below).  Same pieces, same order, same arithmetic: bit-identical results.
    auto prologue_issue = [&](auto walk_c) {
        if constexpr (PRIV) {
            // A(0), then the pairs [A(j + 1), B(j)] for j = 0 .. NS - 2: A runs one step ahead of B (see `run`)
            issue_a(walk_c, 0, 0);
            if (nk > 1) issue_a(walk_c, 1, SLOT);
            issue_b(0, 0);
            if (nk > 2) issue_a(walk_c, 2, 2 * SLOT);
            if (nk > 1) issue_b(1, SLOT);
            if constexpr (NS == 4) {
                if (nk > 3) issue_a(walk_c, 3, 3 * SLOT);
                if (nk > 2) issue_b(2, 2 * SLOT);
            }
        } else {
            issue_a(walk_c, 0, 0); issue_b(0, 0);
            if (nk > 1) { issue_a(walk_c, 1, SLOT); issue_b(1, SLOT); }
        }
    };
    float rsc[A_LD];                 // this lane's rows' operand scales (the four lanes of a row agree)
    {
        // per-row operand scale: max over the row's taps of the per-pixel max |A|; the 4 lanes of a row share the taps.
        // Launches with 25 or more taps that were given the per-image range of those maxima (the 7 x 7 stem: 49 taps = 13
        // dependent-latency loads per lane and tile before the first DMA) take the row's IMAGE maximum instead -- for the rows of
        // images whose nonzero pixels all lie within 2^LVL_STEP of that maximum (then it is within 2^LVL_STEP of every row's own
        // maximum too, and every row keeps the full 22 bits: the B-cos network input [x, 1 - x] always qualifies); the rows of
        // other images scan their taps.  Either way the scale of a row is a function of its image alone.
        const bool img_scale = p.a_imgmax != nullptr && p.a_imgmin != nullptr && g.TH * g.TW >= 25;
        bool scan = !img_scale;
        unsigned rmax[A_LD];
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            rmax[j] = 0u;
            if (img_scale && row_img[j] >= 0) {
                const unsigned mx = p.a_imgmax[row_img[j]], mn = p.a_imgmin[row_img[j]];
                const unsigned Ei = max(mx >> 23, 15u);
                if (Ei - min(mn >> 23, Ei) <= (unsigned)LVL_STEP) rmax[j] = mx; else scan = true;
            }
        }
        const int ntaps = g.TH * g.TW;
        // (the four lanes of a row take the same decision; a pixel maximum never exceeds its image's, so a row that already holds
        //  its image maximum is not changed by lanes of the same wave that scan)
        if (scan) {
            for (int t = chunk; t < ntaps; t += 4) {
                const int th = t / g.TW, tw = t - th * g.TW;
                const int dh = th * g.dstep_h, dw = tw * g.dstep_w;
#pragma unroll
                for (int j = 0; j < A_LD; ++j) {
                    const int ih = a_ih0[j] + dh, iw = a_iw0[j] + dw;
                    const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                    const unsigned v = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(m_rsrc, ok ? (unsigned)(pix0[j] + ih * W + iw) * 4u : OOB, 0, 0);
                    rmax[j] = max(rmax[j], v);
                }
            }
        }
        // (round 4) the prologue's DMA pieces go out behind the loads of the maxima and ahead of their use: both latencies overlap
#if D_EARLY
        if (walk == 1) prologue_issue(std::integral_constant<int, 1>{});
        else if (walk == 2) prologue_issue(std::integral_constant<int, 2>{});
        else if (walk == 3) prologue_issue(std::integral_constant<int, 3>{});
#endif
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            unsigned v = rmax[j];
            v = max(v, (unsigned)__shfl_xor((int)v, 1));
            v = max(v, (unsigned)__shfl_xor((int)v, 2));
            unsigned E = v >> 23;                  // biased exponent of the row max (the bit patterns carry no sign)
            E = E < 15u ? 15u : E;
            rsc[j] = __uint_as_float((268u - E) << 23);                                          // max * scale in [2^14, 2^15)
            if ((lane & 3) == 0) s_scale[r0 + 16 * j] = rsc[j];
        }
    }


    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float pa[TM], pb[TM];            // patch-norm partial sums of the lane's two chunks: the chains of tile_body_h2's chunk lanes
#pragma unroll
    for (int i = 0; i < TM; ++i) { pa[i] = 0.f; pb[i] = 0.f; }

    const int frag_row = lane & 31, frag_half = lane >> 5;
    int a_frag[TM];                  // byte offset of the fragment row's first piece (chunk 2 * half) inside the A slot
    float f_scale[TM];
    const int b_frag = A_BYTES + (wave_n * TN) * 2048 + lane * 16;

    // A wave whose fragment rows are the rows of its own DMA pieces (PRIV) already HOLDS their scales -- lane l has the rows
    // (l >> 2) + 16 j -- so for every walk without the LDS tap table the scales travel by lane permutation and the tile starts its
    // loop without a workgroup barrier (and without the drain of the prologue's DMA that barrier implies): the K <= 256 layers run a
    // dozen short tiles per slot and paid that barrier in each.  The LDS copy of the scales is still written: the epilogue's staging
    // layout reads other waves' rows, behind the loop's barriers.
    const bool no_barrier = D_EARLY && PRIV && walk != 0;
    if (!no_barrier) __syncthreads();                 // scales and the (tap, row) offset table are complete
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int R = wave_m * WM + i * 32 + frag_row;
        a_frag[i] = R * 64 + (((2 * frag_half) ^ ((R >> 2) & 3)) << 4);
        if constexpr (PRIV) {
            // local row i * 32 + frag_row = 16 j + (lane >> 2) of lane 4 (frag_row & 15), j = 2 i + (frag_row >> 4)
            const float v0 = __shfl(rsc[(2 * i) % A_LD], 4 * (frag_row & 15));
            const float v1 = __shfl(rsc[(2 * i + 1) % A_LD], 4 * (frag_row & 15));
            f_scale[i] = no_barrier ? ((frag_row & 16) ? v1 : v0) : s_scale[R];
        } else {
            f_scale[i] = s_scale[R];
        }
    }

    f16x8 af[2][TM];                 // the A fragments of the step about to be multiplied: [h | l][row tile]
    // fp32 fragment rows of the slot at byte offset `off` -> registers (two 16-byte pieces per 32 x 16 fragment) ...
    auto read_a = [&](int off, f32x4 (&x0)[TM], f32x4 (&x1)[TM]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            x0[i] = *reinterpret_cast<const f32x4*>(lds + off + a_frag[i]);            // k = 8 half .. + 3
            x1[i] = *reinterpret_cast<const f32x4*>(lds + off + (a_frag[i] ^ 16));     // k = 8 half + 4 .. + 7
        }
    };
#if H2_MIX_SPLIT
    // ... -> patch-norm partial sums and the (h, l) f16 fragments, in 2 TM slices of one 16-byte piece each (slice = (row tile i, piece
    // x0 | x1)) so that the vector instructions can be dealt out between matrix instructions
    constexpr int NSLICE = 2 * TM;
    auto split_slice = [&](auto slice_c, const f32x4 (&x0)[TM], const f32x4 (&x1)[TM]) {
        constexpr int SL = decltype(slice_c)::value;
        constexpr int i = SL / 2, piece = SL & 1;
        u32x4v h4 = __builtin_bit_cast(u32x4v, af[0][i]), l4 = __builtin_bit_cast(u32x4v, af[1][i]);
        unsigned h01, h23, l01, l23;
        split4_f16<NORM>(piece ? x1[i] : x0[i], f_scale[i], h01, h23, l01, l23, piece ? pb[i] : pa[i]);
        h4[piece * 2] = h01; h4[piece * 2 + 1] = h23;
        l4[piece * 2] = l01; l4[piece * 2 + 1] = l23;
        af[0][i] = __builtin_bit_cast(f16x8, h4);
        af[1][i] = __builtin_bit_cast(f16x8, l4);
    };
    auto split_a = [&](const f32x4 (&x0)[TM], const f32x4 (&x1)[TM]) {
        split_slice(std::integral_constant<int, 0>{}, x0, x1); split_slice(std::integral_constant<int, 1>{}, x0, x1);
        if constexpr (TM > 1) { split_slice(std::integral_constant<int, 2>{}, x0, x1); split_slice(std::integral_constant<int, 3>{}, x0, x1); }
        static_assert(TM <= 2, "slices");
    };
#else
    // ... -> patch-norm partial sums and the (h, l) f16 fragments, in 4 TM slices of two elements each (slice = (row tile i,
    // piece x0 | x1, element pair)) so that the vector instructions can be dealt out between matrix instructions
    constexpr int NSLICE = 4 * TM;
    auto split_slice = [&](auto slice_c, const f32x4 (&x0)[TM], const f32x4 (&x1)[TM]) {
        constexpr int SL = decltype(slice_c)::value;
        constexpr int i = SL / 4, piece = (SL / 2) & 1, q0 = (SL & 1) * 2;
        const f32x4 x = piece ? x1[i] : x0[i];
#pragma unroll
        for (int q = q0; q < q0 + 2; ++q) {
            if (NORM) {
                if (piece) pb[i] = fmaf(x[q], x[q], pb[i]);
                else pa[i] = fmaf(x[q], x[q], pa[i]);
            }
            const float xs = x[q] * f_scale[i];
            const _Float16 h = (_Float16)xs;
            af[0][i][piece * 4 + q] = h;
            af[1][i][piece * 4 + q] = (_Float16)(xs - (float)h);
        }
    };
    auto split_a = [&](const f32x4 (&x0)[TM], const f32x4 (&x1)[TM]) {
        split_slice(std::integral_constant<int, 0>{}, x0, x1); split_slice(std::integral_constant<int, 1>{}, x0, x1);
        split_slice(std::integral_constant<int, 2>{}, x0, x1); split_slice(std::integral_constant<int, 3>{}, x0, x1);
        if constexpr (TM > 1) {
            split_slice(std::integral_constant<int, 4>{}, x0, x1); split_slice(std::integral_constant<int, 5>{}, x0, x1);
            split_slice(std::integral_constant<int, 6>{}, x0, x1); split_slice(std::integral_constant<int, 7>{}, x0, x1);
        }
        static_assert(TM <= 2, "slices");
    };
#endif
    // The matrix instructions of one step on slot `off`, af = this step's A fragments.  `NEXT` (wave-private A rows only): the
    // next step's A rows (slot `off_nx`, already landed) are read up front and split slice by slice BETWEEN this step's matrix
    // instructions, whose shadow hides the vector work.  The issue order is pinned (sched_barrier between the chunks): left
    // to itself the scheduler sinks every fragment read to just before its first use and the whole split behind the last
    // matrix instruction.  Product order per accumulator as in tile_body_h2 (smallest terms first): l_a h_b, h_a l_b, h_a h_b.
    auto mma_step = [&](int off, auto next_c, int off_nx, auto npiece_c, auto piece) {
        constexpr bool NEXT = decltype(next_c)::value;
        constexpr int NPIECE = decltype(npiece_c)::value;           // DMA pieces of later steps to issue behind this step's matrix instructions
        const char* bb = lds + off + b_frag;
        f32x4 x0[TM], x1[TM];
        f16x8 a1[TM], a0[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) { a0[i] = af[0][i]; a1[i] = af[1][i]; }
        constexpr int G = 3 * TM * TN;                              // matrix instructions of the step
        // "actions" dealt out behind the matrix instructions, one group per instruction from the second one on: first the DMA
        // pieces (a piece issued right behind a matrix instruction hides ~half of its issue time in that instruction's shadow;
        // issued at the top of the step, behind the barrier, all of it is exposed), then the slices of the split (the A
        // reads have returned by then)
        constexpr int NACT = NPIECE + ((NEXT && !(D_KO & 2)) ? NSLICE : 0);
        constexpr int D0 = (NPIECE == 0 && G >= 12) ? G / 4 : (G > 1 ? 1 : 0);     // matrix instructions ahead of the first action (slices: the A reads need a few of them to return)
        constexpr int PER = (NACT + (G - D0) - 1) / (G - D0);
        auto act = [&](auto a_c) {
            constexpr int a = decltype(a_c)::value;
            if constexpr (a < NPIECE) piece(a_c);
            else if constexpr (a < NACT) split_slice(std::integral_constant<int, (a < NACT ? a - NPIECE : 0)>{}, x0, x1);
        };
        auto deal = [&](auto m_c) {
            constexpr int m = decltype(m_c)::value;
            if constexpr (m >= D0) {
                constexpr int a0 = (m - D0) * PER;
                [&]<int... As>(std::integer_sequence<int, As...>) { (act(std::integral_constant<int, a0 + As>{}), ...); }(std::make_integer_sequence<int, PER>{});
            }
        };
        constexpr bool ACC_MAJOR = TM == 1 && (H2_MFMA_ORDER == 1 || (H2_MFMA_ORDER == 0 && TM * TN >= 8));
        if constexpr (!ACC_MAJOR) {
            f16x8 bf[2][TN];
#pragma unroll
            for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[sp][j] = *reinterpret_cast<const f16x8*>(bb + j * 2048 + sp * 1024);
            if constexpr (NEXT) read_a(off_nx, x0, x1);
            if constexpr (H2_PRIO == 1) __builtin_amdgcn_s_setprio(1);
            __builtin_amdgcn_sched_barrier(0);
            auto mm = [&](auto m_c) {
                constexpr int m = decltype(m_c)::value;
                constexpr int pr = m / (TM * TN), i = (m / TN) % TM, j = m % TN;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 0 ? a1[i] : a0[i], bf[pr == 1 ? 1 : 0][j], acc[i][j], 0, 0, 0);
                deal(m_c);
                __builtin_amdgcn_sched_barrier(0);
            };
            [&]<int... Ms>(std::integer_sequence<int, Ms...>) { (mm(std::integral_constant<int, Ms>{}), ...); }(std::make_integer_sequence<int, G>{});
        } else {
            // (TM = 1) column tiles in PAIRS: the six matrix instructions of tiles (2p, 2p + 1) alternate between the two accumulators
            // (a dependent v_mfma directly behind its producer waits for it, and any instruction placed between the two costs ~40
            // cycles more: the pair's chains hide each other, and the reads / split slices go between independent instructions);
            // the B fragments of pair p + 1 are read behind the first two instructions of pair p
            static_assert(TM == 1 && TN % 2 == 0, "pair order");
            f16x8 bq[TN][2];
            auto read_b = [&](auto j_c) {
                constexpr int j = decltype(j_c)::value;
                bq[j][0] = *reinterpret_cast<const f16x8*>(bb + j * 2048);
                bq[j][1] = *reinterpret_cast<const f16x8*>(bb + j * 2048 + 1024);
            };
            read_b(std::integral_constant<int, 0>{});
            read_b(std::integral_constant<int, 1>{});
            if constexpr (NEXT) read_a(off_nx, x0, x1);
            if constexpr (H2_PRIO == 1) __builtin_amdgcn_s_setprio(1);
            __builtin_amdgcn_sched_barrier(0);
            auto mm = [&](auto m_c) {
                constexpr int m = decltype(m_c)::value;
                constexpr int pp = m / 6, pr = (m % 6) / 2, j = 2 * pp + (m & 1);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 0 ? a1[0] : a0[0], bq[j][pr == 1 ? 1 : 0], acc[0][j], 0, 0, 0);
                if constexpr (m % 6 < 2 && 2 * pp + 2 + (m & 1) < TN) read_b(std::integral_constant<int, (2 * pp + 2 + (m & 1) < TN ? 2 * pp + 2 + (m & 1) : 0)>{});
                deal(m_c);
                __builtin_amdgcn_sched_barrier(0);
            };
            [&]<int... Ms>(std::integer_sequence<int, Ms...>) { (mm(std::integral_constant<int, Ms>{}), ...); }(std::make_integer_sequence<int, G>{});
        }
        static_assert((G - D0) * PER >= NACT, "every action is dealt out");
        if constexpr (H2_PRIO == 1) { __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); }
    };

    // Ring: step s lives in slot s % NS (A rows and B blocks).  The loops are peeled so that the steady state has no branch.
    auto rot = [](int o) { return o + SLOT == NS * SLOT ? 0 : o + SLOT; };
    auto barrier = []() {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");         // (the raw barrier does not order the compiler's memory operations)
    };
    auto run = [&](auto walk_c) {
        if constexpr (PRIV) {
            // A wave converts its OWN A rows one step ahead of the matrix instructions that use them, so A runs one step ahead
            // of B in the DMA queue: at the top of step ks the queue of a wave holds, oldest first,
            //   A(ks+1) B(ks) | A(ks+2) B(ks+1)            (A(ks+3) B(ks+2) are issued behind the barrier)
            // and ONE counted wait (everything but the last pair) covers what step ks needs: B(ks) for its matrix instructions,
            // A(ks+1) for the split that runs in their shadow.  Every piece has two steps to land.
            if constexpr (decltype(walk_c)::value == 0 || !D_EARLY) prologue_issue(walk_c);      // (the other walks: issued ahead of the scales, above)
            // The queue of a wave, oldest first, is A(0), then the PAIRS j = [A(j + 1), B(j)]; the prologue issued the pairs 0 .. NS - 2, step
            // ks issues pair ks + NS - 1 (A(ks + NS) into the slot of step ks, whose rows were split during step ks - 1; B(ks + NS - 1) into the
            // slot of step ks - 1, whose blocks were read during step ks - 1) and needs pair ks landed: B(ks) for its matrix instructions,
            // A(ks + 1) for the split that runs in their shadow -- ONE counted wait that leaves the NS - 2 younger pairs in flight (those
            // that exist near the end of the walk).  Every piece has NS - 1 steps to land.
            // before the first split: everything behind A(0) may still be in flight
            if constexpr (NS == 3) { if (nk > 2) wait_vmcnt<2 * LA + 2 * LB>(); else if (nk > 1) wait_vmcnt<LA + 2 * LB>(); else wait_vmcnt<LB>(); }
            else { if (nk > 3) wait_vmcnt<3 * LA + 3 * LB>(); else if (nk > 2) wait_vmcnt<2 * LA + 3 * LB>(); else if (nk > 1) wait_vmcnt<LA + 2 * LB>(); else wait_vmcnt<LB>(); }
            {
                f32x4 x0[TM], x1[TM];
                read_a(0, x0, x1);
                split_a(x0, x1);
            }
            int off = 0, off_nx = SLOT, off_pv = (NS - 1) * SLOT;      // slots of step ks, ks + 1, ks - 1
            auto step = [&](auto ia_c, auto ib_c, auto wait_c, auto next_c, int ks) {
                wait_vmcnt<decltype(wait_c)::value>();
                barrier();                         // every wave's B blocks of step ks have landed; the slot of step ks - 1 is free
                constexpr bool IA = decltype(ia_c)::value && !(D_KO & 1), IB = decltype(ib_c)::value && !(D_KO & 1);
#if D_DEAL_DMA
                // the DMA pieces dealt out one by one behind the step's matrix instructions.  Measured AGAINST issuing them
                // together behind the barrier (same node, batch 256): 3x3 @14^2 190 vs 181 us, 3x3 @28^2 226 vs 211 us,
                // 1024 -> 256 @14^2 93.5 vs 87.5 us -- a piece between two matrix instructions delays the second one by more than
                // the piece's issue time saved at the top; not the default
                unsigned va[A_LD];
                int sa = 0;
                if constexpr (IA) walk_a(walk_c, ks + NS, [&](int j, unsigned voff, int soff) { va[j] = voff; sa = soff; });
                const int o_a = off, o_b = off_pv;
                mma_step(off, next_c, off_nx, std::integral_constant<int, (IA ? LA : 0) + (IB ? LB : 0)>{}, [&](auto q_c) {
                    constexpr int q = decltype(q_c)::value;
                    if constexpr (IA && q < LA) piece_a(q, o_a, va[q < LA ? q : 0], sa);
                    else piece_b(q - (IA ? LA : 0), ks + NS - 1, o_b);
                });
#else
                if constexpr (IA) issue_a(walk_c, ks + NS, off);
                if constexpr (IB) issue_b(ks + NS - 1, off_pv);
                mma_step(off, next_c, off_nx, std::integral_constant<int, 0>{}, [](auto) {});
#endif
                off_pv = off; off = off_nx; off_nx = rot(off_nx);
            };
            using T = std::true_type; using F = std::false_type;
            int ks = 0;
            if constexpr (NS == 3) {
                for (; ks + 3 < nk; ++ks) step(T{}, T{}, std::integral_constant<int, LA + LB>{}, T{}, ks);
                if (ks + 2 < nk) { step(F{}, T{}, std::integral_constant<int, LA + LB>{}, T{}, ks); ++ks; }
                if (ks + 1 < nk) { step(F{}, F{}, std::integral_constant<int, LB>{}, T{}, ks); ++ks; }
            } else {
                // pairs in flight behind pair ks: ks + 1 and ks + 2, as far as they exist (r = nk - ks steps left)
                for (; ks + 4 < nk; ++ks) step(T{}, T{}, std::integral_constant<int, 2 * LA + 2 * LB>{}, T{}, ks);        // r >= 5
                if (ks + 3 < nk) { step(F{}, T{}, std::integral_constant<int, 2 * LA + 2 * LB>{}, T{}, ks); ++ks; }       // r == 4: B(ks + 3) is the last piece
                if (ks + 2 < nk) { step(F{}, F{}, std::integral_constant<int, LA + 2 * LB>{}, T{}, ks); ++ks; }           // r == 3: pairs [A(ks+2) B(ks+1)] [B(ks+2)]
                if (ks + 1 < nk) { step(F{}, F{}, std::integral_constant<int, LB>{}, T{}, ks); ++ks; }                    // r == 2: [B(ks+1)]
            }
            step(F{}, F{}, std::integral_constant<int, 0>{}, F{}, ks);
        } else {
            // A rows shared by the waves of a row group (half-height tiles): A and B of step ks + 2 are issued together at the top of
            // step ks, the split happens at the top of the step behind the barrier
            if constexpr (decltype(walk_c)::value == 0 || !D_EARLY) prologue_issue(walk_c);
            int off = 0, off_in = (2 * SLOT) % (NS * SLOT);
            for (int ks = 0; ks < nk; ++ks) {
                if (ks + 1 < nk) wait_vmcnt<LA + LB>(); else wait_vmcnt<0>();
                barrier();
                if (ks + 2 < nk) { issue_a(walk_c, ks + 2, off_in); issue_b(ks + 2, off_in); }
                f32x4 x0[TM], x1[TM];
                read_a(off, x0, x1);
                split_a(x0, x1);
                mma_step(off, std::false_type{}, 0, std::integral_constant<int, 0>{}, [](auto) {});
                off = rot(off); off_in = rot(off_in);
            }
        }
    };
    BCOS_PHASE_MARK(ph_t1);
    if constexpr (H2_PRIO == 2) __builtin_amdgcn_s_setprio(1);
    if (walk == 0) run(std::integral_constant<int, 0>{});
    else if (walk == 1) run(std::integral_constant<int, 1>{});
    else if (walk == 2) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 3>{});
    if constexpr (H2_PRIO == 2) __builtin_amdgcn_s_setprio(0);
    __syncthreads();                           // the ring is free: the epilogue reuses it
    float a_inv[BM / (NT / 4)];      // inverse row scales in staging layout (what tile_epilogue takes): 2^-e from the row's 2^e
#pragma unroll
    for (int j = 0; j < BM / (NT / 4); ++j)
        a_inv[j] = __uint_as_float((254u - (__float_as_uint(s_scale[(tid >> 2) + (NT / 4) * j]) >> 23)) << 23);
    __syncthreads();                           // (the epilogue's row records may overlap the scale table)
    float ss[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) ss[i] = pa[i] + pb[i];
    BCOS_PHASE_MARK(ph_t2);
    tile_epilogue<BM, BN, WAVES_M, WAVES_N, NORM, true, NT>(p, smem, acc, NORM ? ss : nullptr, nullptr, a_inv, m0, n0, tile_n);
#if BCOS_PHASE_TIMING
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long ph_t3 = wall_clock64();
        atomicAdd(&g_phase[0], ph_t1 - ph_t0);
        atomicAdd(&g_phase[1], ph_t2 - ph_t1);
        atomicAdd(&g_phase[2], ph_t3 - ph_t2);
        atomicAdd(&g_phase[3], 1ull);
        atomicMin(&g_phase[4], ph_t0);
        atomicMax(&g_phase[5], ph_t3);
    }
#endif
}

// ---- split-f16 contraction over an LDS-RESIDENT INPUT PATCH (round 3, multi-tap launches) --------------------------------------
// tile_body_d moves and converts every input element once per TAP (9 x for a 3 x 3 layer, 16 x for the depth-to-space stem
// gradient): the fp32 -> (h, l) split and the A traffic into LDS are what its narrow multi-tap launches wait on.  Here the
// K walk is the same (16-channel chunk major, tap inner) but the A operand of a chunk is the tile's input PATCH: the union of the
// pixels its BM rows touch over all taps, laid out as rows of a virtually zero-padded image (PW = (Q - 1) s + TW pixels per
// row, HP = (P - 1) s + TH rows per image), loaded and split ONCE per chunk -- global -> registers -> (h, l) f16 -> LDS, NI items
// of (pixel, 8 channels) per thread, issued a chunk ahead -- and then read as MFMA fragments by every tap: row r, tap (th, tw)
// is patch pixel base[r] + th PW + tw.  Halo pixels are zeros in the patch, so no tap needs a bounds check.
//   * LDS: four planes [h | l] x [k-half] of 16 bytes per pixel (a fragment read is one conflict-free ds_read_b128 per plane:
//     consecutive rows are consecutive pixels), single-buffered (one extra barrier per chunk), + a ring of three B slots fed by
//     LDS-DMA exactly as in tile_body_d (one raw barrier + one counted vmcnt per step).
//   * no per-step A traffic, no per-step vector work: the waves are free to be laid out 2 x 2 (a 64 x 128 wave tile reads
//     12 fragments for 24 matrix instructions; tile_body_d's 32 x 256 wave tiles read 18).
//   * operand scale: ONE power of two per IMAGE (max over the image's per-pixel maxima), not per row -- a pixel serves rows
//     with different tap sets, and a per-tile scale would make an image's bits depend on its batch neighbours.  Elements
//     within 2^-17 of their image's max keep the full 22 bits, smaller ones an absolute error <= 2^-40 of that max.  Same
//     product order per accumulator (l_a h_b, h_a l_b, h_a h_b) and the same K walk as the other split-f16 loops; results agree
//     with them to fp32 rounding, not bit for bit (different scale, patch norm summed per pixel then per tap).
#ifndef P_KO
#define P_KO 0                    // development knock-outs (timing only, wrong results): 1 no patch refill, 2 no B DMA in the steady loop, 4 one barrier per chunk only, 8 no fragment reads after the first step, 16 first product only, 32 no vmcnt waits, 64 no image maxima
#endif
BCOS_DEV_SWITCH(P_KO, 0);
#ifndef P_NSLOT
#define P_NSLOT 3                 // B ring slots: 3 = the DMA of step ks + 2 is issued in step ks (two steps to land), 2 = of step ks + 1 (one step)
#endif
#ifndef P_WGS3
#define P_WGS3 0                  // 1 = the 32-column configuration (depth-to-space stem gradient) is compiled for three resident workgroups per CU
#endif
#ifndef P_DBUF
#define P_DBUF 0                  // 1 = two patch buffers: the refill of chunk c + 1 is written during chunk c, no barrier at the end of a chunk
#endif
#ifndef P_OPT
#define P_OPT 7                   // development switches: 1 = DMA issue behind the fragment reads (else ahead of them), 2 = refill converted from tap 3 on (else at the end of the chunk), 4 = ... one item per tap (else all at tap 3)
#endif
template <int BM, int BN, int PXL>
constexpr size_t p_lds_bytes() { return (size_t)PXL * 64 * (1 + P_DBUF) + P_NSLOT * (size_t)(BN / 32) * 2048 + 1024 + (size_t)BM * 8 + 192; }

template <int BM, int BN, int WAVES_M, int WAVES_N, bool NORM, int NI, int PXL, int NTAPS, int T2BW, bool MORE>
__device__ __noinline__ void tile_body_p_more(const __attribute__((address_space(4))) KArgs* kp, unsigned lds_base, int m0, int n0, int tile_n, int pass, unsigned lvl_mask);

// MORE = false: the tile's first (normally only) pass, inlined into the kernel.  MORE = true: a further pass of a tile whose rows span
// several operand-scale levels (see "Dynamic range" below), run from tile_body_p_more -- an out-of-line function, so that the loop
// over the levels and what it keeps live cost the common path nothing.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool NORM, int NI, int PXL, int NTAPS, int T2BW = 0, bool MORE = false, int NT = NTHREADS, typename PT>
__device__ __forceinline__ void tile_body_p(const PT& p, float* smem, const int m0, const int n0, const int tile_n, int pass = 0, unsigned lvl_mask = 1u) {
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NW = NT / 64;
    static_assert(WAVES_M * WAVES_N == NW && BM % (NT / 4) == 0, "wave layout");
    constexpr int PX = PXL;                         // physical pixels the patch buffer holds (PR rows of a power-of-two pitch); the NI items
                                                    // per thread cover the PR x PW pixels that are used
    constexpr int PLANE = PX * 16;                  // bytes of one (split half, k-half) plane
    constexpr int NBLK = (BN / 32) * 2;             // 1-KB fragment blocks of B per step: (32-column tile, plane)
    constexpr int LB = (NBLK + NW - 1) / NW;        // B DMA instructions per wave per step
    constexpr int BSLOT = (BN / 32) * 2048;
    constexpr int NP = 2 * NI;                      // global loads per thread per patch refill
    constexpr unsigned OOB = 0x80000000u;
    char* lds = reinterpret_cast<char*>(smem);
    char* ring = lds + 4 * PLANE * (1 + P_DBUF);
    char* s_dummy = ring + P_NSLOT * BSLOT;
    int* s_base = reinterpret_cast<int*>(s_dummy + 1024);                 // [BM] (patch row << 16) | rotated column of the row's tap (0, 0)
    float* s_rowinv = reinterpret_cast<float*>(s_base + BM);             // [BM] inverse operand scale of the row (its image's)
    unsigned* s_imgmax = reinterpret_cast<unsigned*>(s_rowinv + BM);     // [16]
    float* s_imgscale = reinterpret_cast<float*>(s_imgmax + 16);         // [16]
    unsigned* s_imgmin = reinterpret_cast<unsigned*>(s_imgscale + 16);   // [16]
    // beyond everything the epilogue touches (KArgs.lvl_off): the level of every tile row and the set of levels present (wide images)
    unsigned char* s_lvl = reinterpret_cast<unsigned char*>(lds + p.lvl_off);     // [BM]
    unsigned* s_lvlmask = reinterpret_cast<unsigned*>(lds + p.lvl_off + BM);
    unsigned* s_pixmax = reinterpret_cast<unsigned*>(lds);               // [PX] per-pixel maxima (wide images; before the first patch is written)
    float* s_pixss = reinterpret_cast<float*>(ring);                     // [PX] per-pixel sums of squares (after the loop: the ring is free)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    const auto& g = p.g;
    const int H = g.H, W = g.W, st = g.in_sh;
    // T2BW > 0: the tile is a (BM / T2BW) x T2BW block of ONE image's row grid (wide images: the rows of a linear tile would drag whole
    // image rows into the patch); the patch is then the block's own halo'd window and `rotq` (the rotation per patch row) its width
    constexpr bool T2D = T2BW > 0;
    constexpr int T2BH = T2D ? BM / (T2D ? T2BW : 1) : 0;
    const int HP = (g.P - 1) * st + g.TH, PW = T2D ? (T2BW - 1) * st + g.TW : (g.Q - 1) * st + g.TW;
    const int rotq = T2D ? T2BW : g.Q;
    // Patch rows have a power-of-two pitch and are ROTATED: logical column cc of patch row jr sits at physical column
    // (cc + jr Q) & (pitch - 1).  Consecutive GEMM rows (j -> j + 1, and (i, Q - 1) -> (i + 1, 0)) are then consecutive 16-byte
    // slots modulo the pitch for every tap, so a fragment read (16 rows per LDS pass) is bank-conflict free; with the rows
    // stored back to back the two halo columns between (i, Q - 1) and (i + 1, 0) cost a 2-way conflict on every pass.
    const int LP = 32 - __builtin_clz((unsigned)(PW - 1) | 1u);
    const int cmask = (1 << LP) - 1;
    const int ntaps = g.TH * g.TW;
    const int nk = (p.nchunks + 3) / 4;
    const int nch = g.C / 16;

    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wt2), 0, p.wt2_bytes, 0x00020000);

    // linear tiles: the rows [m0, m0 + BM) span images n_first .. n_last and the virtual input rows G0 .. G0 + PR - 1;
    // 2-D tiles: one image, the block's first input row / column ih0 / iw0 (before the tap offset)
    int n_first, n_last, G0, PR, ih0 = 0, iw0 = 0;
    if constexpr (T2D) {
        const int t = m0 / BM;
        n_first = n_last = t / p.t2_nb;
        const int b = t - n_first * p.t2_nb;
        const int by = b / p.t2_nbx;
        ih0 = by * T2BH * st;
        iw0 = (b - by * p.t2_nbx) * T2BW * st;
        G0 = 0;
        PR = (T2BH - 1) * st + g.TH;
    } else {
        n_first = m0 / p.PQ;
        const int i_first = (m0 - n_first * p.PQ) / g.Q;
        G0 = n_first * HP + i_first * st;
        const int m_last = (m0 + BM < p.M ? m0 + BM : p.M) - 1;
        n_last = m_last / p.PQ;
        const int i_last = (m_last - n_last * p.PQ) / g.Q;
        PR = n_last * HP + i_last * st + g.TH - G0;
    }
    // (the tile's image maxima: the OLDEST memory operation of the prologue, so that waiting for it leaves the loads below in flight)
    const unsigned my_imgmax = (tid < 16 && n_first + tid <= n_last) ? p.a_imgmax[n_first + tid] : 0u;      // (at most 16 images per tile: patch_fits)
    const unsigned my_imgmin = (tid < 16 && n_first + tid <= n_last && (p.a_imgmin || p.a_imgmin_c))      // (not given: range unknown = wide)
                                   ? (p.a_imgmin ? p.a_imgmin[n_first + tid] : ~p.a_imgmin_c[n_first + tid]) : 0u;
    // this thread's NI items of the patch: item q = tid + NT it is (used pixel q >> 1 in row-major order of the PR x PW patch,
    // k-half q & 1): 8 channels = two 16-byte loads, stored at the pixel's rotated physical position
    unsigned voff[NI];
    float isc[NI];
    int iimg[NI];                    // image of the item (index into the tile's image scales)
    int idst[NI];                    // byte offset of the item in the h half (the l half at + 2 PLANE), or -1: no such pixel
    // item `it` -> its pixel: patch row / column, image and input coordinates; false: no such pixel in the input (halo, padding)
    auto item_pixel = [&](int it, int& jr, int& cc, int& n, int& ih, int& iw) {
        const int px = (tid >> 1) + (NT / 2) * it;
        jr = px / PW;
        cc = px - jr * PW;
        if constexpr (T2D) {
            n = n_first;
            ih = ih0 + jr + g.dh0;
            iw = iw0 + cc + g.dw0;
        } else {
            const int G = G0 + jr;
            n = G / HP;
            ih = G - n * HP + g.dh0;
            iw = cc + g.dw0;
        }
        return jr < PR && n < g.N && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    };
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        int jr, cc, n, ih, iw;
        const bool ok = item_pixel(it, jr, cc, n, ih, iw);
        voff[it] = ok ? ((unsigned)((n * H + ih) * W + iw) * (unsigned)g.a_pitch + (tid & 1) * 8u) * 4u : OOB;
        int k = n - n_first;
        iimg[it] = k < 0 ? 0 : (k > 15 ? 15 : k);
        idst[it] = jr < PR ? (tid & 1) * PLANE + (((jr << LP) + ((cc + jr * rotq) & cmask)) << 4) : -1;
    }
    f32x4 xr[NI][2];
    float pss[NI];
#pragma unroll
    for (int it = 0; it < NI; ++it) pss[it] = 0.f;
    auto load_items = [&](int soff) {
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            xr[it][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, (int)voff[it], soff, 0));
            xr[it][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, (int)(voff[it] + 16u), soff, 0));
        }
    };
    const int b_tile0 = n0 >> 5;
    auto issue_b = [&](int ks, int slot_off) {
#pragma unroll
        for (int j = 0; j < LB; ++j) {
            const int blk = wave + NW * j;
            if (NBLK % NW == 0 || blk < NBLK) {
                const int soff = ((b_tile0 + (blk >> 1)) * nk + ks) * 2048 + (blk & 1) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, BCOS_LDS_PTR(ring + slot_off + blk * 1024), 16, lane * 16, soff, 0, 0);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, BCOS_LDS_PTR(s_dummy), 16, (int)OOB, 0, 0, 0);   // keeps every wave's DMA count equal
            }
        }
    };

    // The first chunk's loads and the first B blocks go out BEFORE the rest of the prologue (image scales, row table: two barriers
    // and a few hundred integer instructions), which then runs under their latency instead of ahead of it.
    constexpr int AHEAD = P_NSLOT - 1;               // the DMA of step ks + AHEAD is issued in step ks
    // (first pass, ladder enabled: the per-pixel maxima of the patch go out FIRST -- NI 4-byte loads per even thread, L2 hits -- so
    //  that a tile of a wide-range image can derive its rows' levels behind a counted wait instead of draining the chunk loads and
    //  DMAs issued right behind them; tiles of narrow images never look at them)
    unsigned pmx[NI];
#pragma unroll
    for (int it = 0; it < NI; ++it) pmx[it] = 0u;
    if constexpr (!MORE) {
        if (p.lvl_on && !(tid & 1)) {
            const __amdgpu_buffer_rsrc_t m_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(p.a_absmax), 0, p.absmax_bytes, 0x00020000);
#pragma unroll
            for (int it = 0; it < NI; ++it) {
                int jr, cc, n, ih, iw;
                const bool ok = item_pixel(it, jr, cc, n, ih, iw);
                pmx[it] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(m_rsrc, ok ? (unsigned)((n * H + ih) * W + iw) * 4u : OOB, 0, 0);
            }
        }
    }
    load_items(0);
    issue_b(0, 0);
    if constexpr (AHEAD == 2) issue_b(1, BSLOT);
    // (barriers of the prologue: LDS traffic only -- __syncthreads() would also wait for the loads and DMAs issued above)
    auto lds_barrier = []() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    if (tid < 16) { s_imgmax[tid] = my_imgmax; s_imgmin[tid] = my_imgmin; }
    lds_barrier();
    // Dynamic range INSIDE the tile's images.  One operand scale per image keeps the full 22 bits of the elements within 2^-17 of the
    // image maximum only; a row whose whole tap window is far darker than its image's brightest pixel (a blob on a dark background,
    // the sparse gradients of the explanation pass) would be computed with an absolute error of 2^-40 of that maximum -- large next
    // to its own patch.  So the scales form a LADDER: row r of image n belongs to level l(r) = floor((E_n - E_r) / LVL_STEP), E the
    // exponents of the image maximum and of the row's maximum over its taps (from the per-pixel maxima `a_absmax`), and the tile is
    // contracted once per level present, level l with the scale 2^(l LVL_STEP) above the image's, each pass writing its own rows
    // (a row of level l reads no pixel brighter than 2^-(l LVL_STEP) of the image maximum, so nothing it reads overflows; brighter
    // pixels are clamped).  Level, scale and result of a row are functions of its image alone -- not of the tile or the batch.
    // Images whose nonzero pixels all lie within 2^LVL_STEP of the maximum (`a_imgmin`: every synthetic benchmark input, most
    // activations) have level 0 everywhere and take none of this.
    bool wide = MORE;
    if constexpr (!MORE) {
        for (int k = 0; k <= n_last - n_first && k < 16; ++k) {
            const unsigned Ei = max(s_imgmax[k] >> 23, 15u);
            wide = wide || Ei - min(s_imgmin[k] >> 23, Ei) > (unsigned)LVL_STEP;
        }
        wide = __builtin_amdgcn_readfirstlane((int)(wide && p.lvl_on)) != 0;      // (every thread read the same table)
    }
    const unsigned char* lvl_rows = wide ? s_lvl : nullptr;
    if (!MORE && wide) {
        if (tid == 0) *s_lvlmask = 0u;
        if (!(tid & 1)) {
#pragma unroll
            for (int it = 0; it < NI; ++it)
                if (idst[it] >= 0) s_pixmax[idst[it] >> 4] = pmx[it];
        }
        lds_barrier();
        for (int r = tid; r < BM; r += NT) {
            int n = n_first, jr = -1, jc = 0;
            if constexpr (T2D) {
                jr = (r / T2BW) * st;
                jc = (r % T2BW) * st;
            } else {
                const int m = m0 + r;
                if (m < p.M) {
                    n = m / p.PQ;
                    const int rem = m - n * p.PQ;
                    const int i = rem / g.Q;
                    jr = n * HP + i * st - G0;
                    jc = (rem - i * g.Q) * st;
                }
            }
            int level = 0;
            if (jr >= 0) {
                const int c0 = (jc + jr * rotq) & cmask;
                unsigned rmax = 0u;
                for (int t = 0; t < ntaps; ++t) {
                    const int th = t / g.TW, tw = t - th * g.TW;
                    rmax = max(rmax, s_pixmax[((jr + th) << LP) + ((c0 + tw + th * rotq) & cmask)]);
                }
                const unsigned Ei = max(s_imgmax[(n - n_first) & 15] >> 23, 15u);
                const unsigned Er = min(max(rmax >> 23, 1u), Ei);
                const unsigned lmax = (Ei - 15u) / LVL_STEP;                       // the scale's exponent stays representable
                level = rmax ? (int)min(min((Ei - Er) / LVL_STEP, lmax), 31u) : 0;   // (an all-zero window is exact at any level)
                atomicOr(s_lvlmask, 1u << level);
            }
            s_lvl[r] = (unsigned char)level;
        }
        lds_barrier();
        lvl_mask = (unsigned)__builtin_amdgcn_readfirstlane((int)*s_lvlmask);
        if (lvl_mask == 0u) lvl_mask = 1u;
        pass = __builtin_ctz(lvl_mask);
        lvl_mask >>= pass;
    }

    for (int r = tid; r < BM; r += NT) {
        int base = 0;
        float inv = 1.0f;
        int n = n_first, jr = -1, jc = 0;
        if constexpr (T2D) {
            jr = (r / T2BW) * st;             // (rows outside the image contract zeros and are dropped by the epilogue)
            jc = (r % T2BW) * st;
        } else {
            const int m = m0 + r;
            if (m < p.M) {
                n = m / p.PQ;
                const int rem = m - n * p.PQ;
                const int i = rem / g.Q;
                jr = n * HP + i * st - G0;
                jc = (rem - i * g.Q) * st;
            }
        }
        if (jr >= 0) {
            base = (jr << 16) | ((jc + jr * rotq) & cmask);
            unsigned E = s_imgmax[(n - n_first) & 15] >> 23;
            E = E < 15u ? 15u : E;
            E -= (unsigned)(pass * LVL_STEP);      // (>= 15: a level never exceeds (E - 15) / LVL_STEP)
            inv = __uint_as_float((E - 14u) << 23);
        }
        s_base[r] = base;
        s_rowinv[r] = inv;
    }
    if (tid < 16) {
        unsigned E = s_imgmax[tid] >> 23;          // biased exponent of the image max (the bit patterns carry no sign)
        E = E < 15u ? 15u : E;
        const unsigned Es = 268u - E + (unsigned)(pass * LVL_STEP);
        s_imgscale[tid] = __uint_as_float((Es > 254u ? 254u : Es) << 23);       // max * scale in [2^14, 2^15) at level 0
    }
    lds_barrier();
#pragma unroll
    for (int it = 0; it < NI; ++it) isc[it] = s_imgscale[iimg[it]];
    const bool clamp_h = pass > 0;         // brighter pixels than this level's rows read: keep them finite

    // the refill in two halves: registers -> (h, l) f16 registers as soon as the loads have landed (vector work in the shadow of a
    // step's matrix instructions), registers -> LDS at the end of the chunk, behind the barrier that retires the old patch
    f16x8 ph[NI], pl[NI];
    auto convert_item = [&](auto it_c) {
        constexpr int it = decltype(it_c)::value;
#if H2_MIX_SPLIT
        if (!clamp_h) {            // (uniform; level 0, i.e. every pass of an image without the ladder: two instructions per element, see split4_f16)
            u32x4v h4, l4;
            unsigned h01, h23, l01, l23;
            split4_f16<NORM>(xr[it][0], isc[it], h01, h23, l01, l23, pss[it]);
            h4[0] = h01; h4[1] = h23; l4[0] = l01; l4[1] = l23;
            split4_f16<NORM>(xr[it][1], isc[it], h01, h23, l01, l23, pss[it]);
            h4[2] = h01; h4[3] = h23; l4[2] = l01; l4[3] = l23;
            ph[it] = __builtin_bit_cast(f16x8, h4);
            pl[it] = __builtin_bit_cast(f16x8, l4);
            return;
        }
#endif
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float x = xr[it][q >> 2][q & 3];
            if (NORM) pss[it] = fmaf(x, x, pss[it]);
            float xs = x * isc[it];
            if (clamp_h) xs = __builtin_amdgcn_fmed3f(xs, -32768.f, 32768.f);      // (uniform branch; only passes above level 0)
            const _Float16 hh = (_Float16)xs;
            ph[it][q] = hh;
            pl[it][q] = (_Float16)(xs - (float)hh);
        }
    };
    auto convert_items = [&]() {
        [&]<int... Is>(std::integer_sequence<int, Is...>) { (convert_item(std::integral_constant<int, Is>{}), ...); }(std::make_integer_sequence<int, NI>{});
    };
    auto write_items = [&](int buf_off, bool publish) {
#pragma unroll
        for (int it = 0; it < NI; ++it)
            if (idst[it] >= 0) {
                *reinterpret_cast<f16x8*>(lds + buf_off + idst[it]) = ph[it];
                *reinterpret_cast<f16x8*>(lds + buf_off + idst[it] + 2 * PLANE) = pl[it];
            }
        if (publish) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the raw s_barrier that publishes the patch does not wait for LDS writes;
                                                                             //  with two buffers the fragment reads of the following taps do)
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int a_row[TM], a_c0[TM];         // fragment row: byte offset of its patch row in its k-half plane of the h half; rotated column of tap (0, 0)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int b = s_base[wave_m * WM + i * 32 + (lane & 31)];
        a_row[i] = (lane >> 5) * PLANE + (((b >> 16) << LP) << 4);
        a_c0[i] = b & 0xffff;
    }
    const int b_frag = (wave_n * TN) * 2048 + lane * 16;
    // A fragments of tap (th, tw): patch row + th, rotated column + tw + th Q; the l half (first product) and the h half are read
    // at different points of a step
    f16x8 a_h[TM], a_l[TM];
    int pbuf = 0;                    // byte offset of the patch buffer of the current chunk (two buffers: P_DBUF)
    auto a_addr = [&](int i, int th, int tw) -> const char* {
        return lds + pbuf + a_row[i] + ((th << LP) << 4) + (((a_c0[i] + tw + th * rotq) & cmask) << 4);
    };
    auto read_al = [&](int th, int tw) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a_l[i] = *reinterpret_cast<const f16x8*>(a_addr(i, th, tw) + 2 * PLANE);
    };
    auto read_ah = [&](int th, int tw) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a_h[i] = *reinterpret_cast<const f16x8*>(a_addr(i, th, tw));
    };
    auto barrier = []() {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // prologue: chunk 0 of the patch (its loads and the B blocks of steps 0 and 1 were issued at the top; nk >= 4)
    wait_vmcnt<AHEAD * LB>();
    convert_items();
    write_items(0, true);
    int ks = 0;
    int boff = 0, boff_nx = BSLOT, boff_in = AHEAD == 2 ? 2 * BSLOT : BSLOT;
    // One step = one tap of the chunk.  The taps of a chunk are expanded at compile time (NTAPS, square tap grids): every wait
    // count is then a constant, and the compiler's own s_waitcnt for the refill registers sees how many DMA instructions were
    // issued behind their loads -- inside a run-time tap loop it assumes none and waits for the DMAs issued a moment ago.
    // LAST: the last chunk (no refill; its last two taps issue no B).
    constexpr int TWC = NTAPS == 4 ? 2 : NTAPS == 9 ? 3 : 4;
    constexpr int TCONV = 3;           // the refill loads are forced complete by the wait at the top of tap 3
    static_assert(NTAPS == TWC * TWC && NTAPS > TCONV && (!(P_OPT & 4) || TCONV + NI <= NTAPS), "square tap grids; one converted item per tap");
    f16x8 bf[2][TN];
    auto step = [&]<int T, bool LAST>(int c) {
        constexpr int th = T / TWC, tw = T % TWC;
        // B(ks) has landed: the queue behind it holds B(ks + 1) and, for two steps after a refill was issued, its NP loads
        if constexpr (!(P_KO & 32)) {
            constexpr int BEHIND = (AHEAD - 1) * LB;          // B(ks + 1) where the ring is three slots deep
            if constexpr (LAST) {
                if constexpr (T + 1 < NTAPS) wait_vmcnt<BEHIND>(); else wait_vmcnt<0>();
            } else {
                if constexpr ((T == 1 || (T == 2 && AHEAD == 2)) && !(P_KO & 1)) wait_vmcnt<BEHIND + NP>(); else wait_vmcnt<BEHIND>();
            }
        }
        if constexpr (!(P_KO & 4) || T == 0) barrier();      // every wave's B blocks of step ks (and the patch writes of this chunk) are visible; the slot of step ks + 2 is free
        constexpr bool ISSUE = !LAST || T + AHEAD < NTAPS;
        constexpr bool REFILL = T == 0 && !LAST && !(P_KO & 1);
        if constexpr (!(P_OPT & 1)) {
            if constexpr (ISSUE) { if (!(P_KO & 2) || ks == 0) issue_b(ks + AHEAD, boff_in); }
            if constexpr (REFILL) load_items((c + 1) * 64);
        }
        if constexpr (T == 0) read_al(0, 0);      // (later taps: read behind the previous step's first product; the patch does not change inside a chunk)
        // Product-major over the wave's accumulators: a dependent v_mfma never sits directly behind its producer.  Fragment reads
        // in the order of first use: l_a and h_b for the first product, then h_a, l_b; the NEXT tap's l_a goes into the registers
        // the first product has just released and lands under the other two products.
        const char* bb = ring + boff + b_frag;
        if (!(P_KO & 8) || ks == 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[0][j] = *reinterpret_cast<const f16x8*>(bb + j * 2048);
            read_ah(th, tw);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[1][j] = *reinterpret_cast<const f16x8*>(bb + j * 2048 + 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (P_OPT & 1) {     // the DMA of step ks + 2 (and, first tap, the refill loads) behind the fragment reads: their issue time hides the reads' latency
            if constexpr (ISSUE) { if (!(P_KO & 2) || ks == 0) issue_b(ks + AHEAD, boff_in); }
            if constexpr (REFILL) load_items((c + 1) * 64);
        }
        if constexpr (H2_PRIO == 1) __builtin_amdgcn_s_setprio(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l[i], bf[0][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (T + 1 < NTAPS && !(P_KO & 8)) read_al((T + 1) / TWC, (T + 1) % TWC);
        // the refill's conversion, one item per step from tap TCONV on: ~30 vector instructions in the shadow of the step's second and third product
        if constexpr ((P_OPT & 2) && !LAST && !(P_KO & 1)) {
            if constexpr (P_OPT & 4) {
                if constexpr (T >= TCONV && T - TCONV < NI) convert_item(std::integral_constant<int, (T >= TCONV && T - TCONV < NI) ? T - TCONV : 0>{});
            } else if constexpr (T == TCONV) {
                convert_items();
            }
        }
        if constexpr (P_DBUF && (P_OPT & 2) && T == TCONV + 1 && !LAST && !(P_KO & 1)) write_items(4 * PLANE - pbuf, false);   // the other buffer: last read in chunk c - 1
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pr = 1; pr < ((P_KO & 16) ? 1 : 3); ++pr)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[i], bf[pr == 1 ? 1 : 0][j], acc[i][j], 0, 0, 0);
        if constexpr (H2_PRIO == 1) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        ++ks;
        if constexpr (AHEAD == 2) { const int o = boff; boff = boff_nx; boff_nx = boff_in; boff_in = o; }
        else { const int o = boff; boff = boff_in; boff_in = o; }
    };
    auto chunk = [&]<bool LAST, int... Ts>(int c, std::integer_sequence<int, Ts...>) {
        (step.template operator()<Ts, LAST>(c), ...);
        if constexpr (!LAST && !(P_KO & 1) && !P_DBUF) {
            barrier();                 // every wave is done reading this chunk's patch
            if constexpr (!(P_OPT & 2)) convert_items();
            write_items(0, true);
        }
        if constexpr (P_DBUF) pbuf = 4 * PLANE - pbuf;
    };
    if constexpr (H2_PRIO == 2) __builtin_amdgcn_s_setprio(1);
    for (int c = 0; c + 1 < nch; ++c) chunk.template operator()<false>(c, std::make_integer_sequence<int, NTAPS>{});
    chunk.template operator()<true>(nch - 1, std::make_integer_sequence<int, NTAPS>{});
    if constexpr (H2_PRIO == 2) __builtin_amdgcn_s_setprio(0);
    __syncthreads();                   // patch and ring are free
    float rowss[BM / (NT / 4)], a_inv[BM / (NT / 4)];
    if (NORM) {
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const float v = pss[it] + __shfl_xor(pss[it], 1);
            if (!(tid & 1) && idst[it] >= 0) s_pixss[idst[it] >> 4] = v;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < BM / (NT / 4); ++j) {
            const int b = s_base[(tid >> 2) + (NT / 4) * j];
            const int jr = b >> 16, c0 = b & 0xffff;
            float v = 0.f;
            for (int t = tid & 3; t < ntaps; t += 4) {
                const int th = t / g.TW;
                const int tw = t - th * g.TW;
                v += s_pixss[((jr + th) << LP) + ((c0 + tw + th * rotq) & cmask)];
            }
            rowss[j] = v;              // staging layout: the 4 lanes of a row hold partial sums
        }
    }
#pragma unroll
    for (int j = 0; j < BM / (NT / 4); ++j) a_inv[j] = s_rowinv[(tid >> 2) + (NT / 4) * j];
    __syncthreads();                   // the epilogue reuses all of it
    tile_epilogue<BM, BN, WAVES_M, WAVES_N, NORM, true, NT, false>(p, smem, acc, nullptr, NORM ? rowss : nullptr, a_inv, m0, n0, tile_n, lvl_rows, pass);
    if constexpr (!MORE) {      // the other levels present, if any
        lvl_mask >>= 1;
        if (lvl_mask)
            tile_body_p_more<BM, BN, WAVES_M, WAVES_N, NORM, NI, PXL, NTAPS, T2BW, true>(
                (const __attribute__((address_space(4))) KArgs*)__builtin_amdgcn_kernarg_segment_ptr(),      // (KArgs is the kernel's only argument)
                (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)smem, m0, n0, tile_n, pass + 1, lvl_mask);
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool NORM, int NI, int PXL, int NTAPS, int T2BW, bool MORE>
__device__ __noinline__ void tile_body_p_more(const __attribute__((address_space(4))) KArgs* kp, unsigned lds_base, int m0, int n0, int tile_n, int pass, unsigned lvl_mask) {
    // (arguments of a non-kernel function arrive in vector registers: every one of them is wave-uniform)
    {
        const uint64_t v = (uint64_t)(uintptr_t)kp;
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
        kp = (const __attribute__((address_space(4))) KArgs*)(uintptr_t)(((uint64_t)hi << 32) | lo);
    }
    lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_base);
    m0 = __builtin_amdgcn_readfirstlane(m0);
    n0 = __builtin_amdgcn_readfirstlane(n0);
    tile_n = __builtin_amdgcn_readfirstlane(tile_n);
    pass = __builtin_amdgcn_readfirstlane(pass);
    lvl_mask = (unsigned)__builtin_amdgcn_readfirstlane((int)lvl_mask);
    // (the kernel's dynamic LDS, handed over as its LDS offset so that the address space stays known on this side of the call)
    float* smem = (float*)(__attribute__((address_space(3))) float*)(uintptr_t)lds_base;
    for (; lvl_mask; ++pass, lvl_mask >>= 1) {
        if (!(lvl_mask & 1u)) continue;
        __syncthreads();                   // the previous pass's epilogue is done with the LDS
        tile_body_p<BM, BN, WAVES_M, WAVES_N, NORM, NI, PXL, NTAPS, T2BW, MORE>(*kp, smem, m0, n0, tile_n, pass, 0u);
    }
}

// Narrow split-f16 tiles (128 x 64, 128 x 32: the stem, the 56^2 3x3 layers) hold 16-32 accumulator registers per wave and run
// 6-12 matrix instructions per 16-k step: their steps wait on load latency, not on a pipe.  The FORWARD kernels of these
// tiles are compiled for three resident workgroups per CU (168 registers): same-node A/B on ResNet-50, stem forward
// 1.36 -> 1.07 ms, 3x3 @56^2 forward 1.10 -> 0.92 ms; the gradient kernels (no norm) measured no gain (128 x 64) or a small
// loss (128 x 32, the depth-to-space stem gradient) and stay at two; four workgroups (128 registers) spill.
#ifndef H2_NARROW_WGS
#define H2_NARROW_WGS 3           // resident workgroups per CU the narrow forward kernels are compiled for
#endif
#ifndef H2_NARROW_BN
#define H2_NARROW_BN 64
#endif
// XCD-aware id remap: the 8 XCDs (private L2s) each get a contiguous range of `nt` work items (bijective for any nt)
__device__ __forceinline__ int xcd_remap(int bid, int nt) {
    const int xcd = bid % NXCD, idx = bid / NXCD;
    const int q = nt / NXCD, r = nt % NXCD;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Grid = n_big tiles of BM rows followed by n_small tiles of BM/2 rows covering the remaining rows.  Workgroups are
// dispatched in blockIdx order, so the half-height tiles fill the tail of the launch: with a few hundred equal
// tiles on 256 CUs x 2 resident workgroups the last "round" otherwise runs at ~50 % occupancy (e.g. 784 tiles
// = 1.53 rounds cost 2 rounds).  Results are bit-identical for any split: an output element's k-order is fixed.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool NORM, int X3>     // X3: 0 fp32 MFMA, 1 split-bf16, 2 split-bf16 with pre-split weights, 3 split-f16 (register staging), 4 split-f16 (LDS-DMA staging)
__global__ __launch_bounds__(NTHREADS, ((X3 >= 3 && NORM && BN <= H2_NARROW_BN && BM <= 128) ? H2_NARROW_WGS : (D_WGS3 && X3 == 4 && BM == 128 && BN == 128) ? 3 : 2)) void tapconv_kernel(const KArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bid = blockIdx.x;
    if (bid < p.n_big) {
        const int tile = xcd_remap(bid, p.n_big);
        const int tile_m = tile / p.tiles_n;
        const int tile_n = tile - tile_m * p.tiles_n;
        const int gcol = (int)blockIdx.y * p.g.Cout;        // first global column of this group (0 unless a grouped launch)
        if constexpr (X3 == 4) tile_body_d<BM, BN, WAVES_M, WAVES_N, NORM>(p, smem, tile_m * BM, tile_n * BN, tile_n);
        else if constexpr (X3 == 3) tile_body_h2<BM, BN, WAVES_M, WAVES_N, NORM>(p, smem, tile_m * BM, tile_n * BN, tile_n);
        else if constexpr (X3 == 2) tile_body_x3<BM, BN, WAVES_M, WAVES_N, NORM, true>(p, smem, tile_m * BM, gcol + tile_n * BN, tile_n);
        else if constexpr (X3 == 1) tile_body_x3<BM, BN, WAVES_M, WAVES_N, NORM, false>(p, smem, tile_m * BM, gcol + tile_n * BN, tile_n);
        else tile_body<BM, BN, WAVES_M, WAVES_N, NORM>(p, smem, tile_m * BM, gcol + tile_n * BN, tile_n);
    } else {
        constexpr int BMS = BM / 2;
        constexpr int WMS = (BMS / 32 >= WAVES_M) ? WAVES_M : BMS / 32;    // waves along M of the small tile
        constexpr int WNS = 4 / WMS;
        if constexpr (BN / WNS >= 32) {
            const int tile = xcd_remap(bid - p.n_big, p.n_small);
            const int tile_m = tile / p.tiles_n;
            const int tile_n = tile - tile_m * p.tiles_n;
            const int gcol = (int)blockIdx.y * p.g.Cout;
            if constexpr (X3 == 4) tile_body_d<BMS, BN, WMS, WNS, NORM>(p, smem, p.rows_big + tile_m * BMS, tile_n * BN, tile_n);
            else if constexpr (X3 == 3) tile_body_h2<BMS, BN, WMS, WNS, NORM>(p, smem, p.rows_big + tile_m * BMS, tile_n * BN, tile_n);
            else if constexpr (X3 == 2) tile_body_x3<BMS, BN, WMS, WNS, NORM, true>(p, smem, p.rows_big + tile_m * BMS, gcol + tile_n * BN, tile_n);
            else if constexpr (X3 == 1) tile_body_x3<BMS, BN, WMS, WNS, NORM, false>(p, smem, p.rows_big + tile_m * BMS, gcol + tile_n * BN, tile_n);
            else tile_body<BMS, BN, WMS, WNS, NORM>(p, smem, p.rows_big + tile_m * BMS, gcol + tile_n * BN, tile_n);
        }
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool NORM, int NI, int PXL, int NTAPS, int T2BW>
__global__ __launch_bounds__(NTHREADS, (BN <= 32 && P_WGS3) ? 3 : 2) void tappatch_kernel(const KArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tile = xcd_remap(blockIdx.x, p.n_big);
    const int tile_m = tile / p.tiles_n;
    const int tile_n = tile - tile_m * p.tiles_n;
    tile_body_p<BM, BN, WAVES_M, WAVES_N, NORM, NI, PXL, NTAPS, T2BW>(p, smem, tile_m * BM, tile_n * BN, tile_n);
}

// (An 8-wavefront / 512-thread form of the split-f16 loop -- same tile and LDS images, eight waves of 64 x 32 with 32
//  accumulator registers, four waves per SIMD -- was measured on the ResNet-50 shapes: 186 vs 211 TFLOP/s at M = 50 176,
//  K = 2304 and within +-3 % elsewhere.  More waves per SIMD do not lift the loop; tile_body_h2 keeps its thread-count
//  parameter, the kernel variant was dropped.)
constexpr int SLOTS = 512;   // 256 CUs x 2 resident workgroups (LDS- and VGPR-limited)

// tile counts of a launch: n_big tiles of BM rows, then n_small half-height tiles (see tapconv_kernel)
template <int BM, int BN, int WAVES_M>
void plan_tiles(KArgs& p) {
    p.tiles_n = (p.g.Cout + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int64_t total = (int64_t)tiles_m * p.tiles_n;
    // how many BM-row tiles to keep at full height: whole "rounds" of SLOTS
    constexpr bool can_split = (BM / 2 >= 32) && (BN / (4 / ((BM / 64 >= WAVES_M) ? WAVES_M : BM / 64)) >= 32);
    int m_big = tiles_m;
    // Measured policy (ResNet-50 layer shapes, split-bf16 kernel): the split pays from two full rounds on (e.g. 1568
    // tiles: -5 %); below that a CU left with a single resident workgroup runs it ~1.5x faster, which already hides
    // most of the tail, and the half-height tiles' lower efficiency dominates (392 tiles: +25 % when split).
    if (can_split && total >= 2 * SLOTS && total < 8 * SLOTS && bcos_option(BCOS_OPT_TAIL_SPLIT)) {
        const int64_t full = (total / SLOTS) * SLOTS;
        const int64_t rem = total - full;
        if (rem > 0 && rem < (SLOTS * 9) / 10) m_big = (int)(full / p.tiles_n);
    }
    // (Between half a round and one round -- 392 tiles on 512 slots: 136 CUs carry two tiles, 120 one -- keeping one full-height
    // tile per CU and cutting the rest into half-height tiles was measured with the LDS-DMA loop too: 243 against 181 us at
    // M = 50 176, N = 256, K = 2304; the 64-row bodies convert every A row twice and cannot run the split a step ahead.)
    p.rows_big = m_big * BM;
    if (p.rows_big > p.M) p.rows_big = p.M;
    p.n_big = m_big * p.tiles_n;
    const int rows_small = p.M - p.rows_big;
    p.n_small = rows_small > 0 ? ((rows_small + BM / 2 - 1) / (BM / 2)) * p.tiles_n : 0;
}

template <int BM, int BN, int WAVES_M>
constexpr size_t epilogue_lds() {
    constexpr int PM = (BM > EPI_ROWS && (BM / WAVES_M) % (32 * (BM / EPI_ROWS)) == 0) ? BM / EPI_ROWS : 1;
    constexpr int SBM = BM / PM, SBN = BN / epi_pn<BN>();
    return (size_t)SBM * (SBN + 4) * sizeof(float) + (size_t)BM * 20 + (size_t)BN * 4 + (size_t)BM * 8 + 144;      // (+ the row maxima of the column parts + the per-image range cells)
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch_cfg(const KArgs& base, bool norm, hipStream_t stream) {
    KArgs p = base;
    plan_tiles<BM, BN, WAVES_M>(p);
    size_t lds = 2 * (size_t)(BM + BN) * LDS_LD * sizeof(float);
    const size_t lds_epi = epilogue_lds<BM, BN, WAVES_M>();
    if (lds_epi > lds) lds = lds_epi;
    const dim3 grid((unsigned)(p.n_big + p.n_small), (unsigned)(p.g.groups > 1 ? p.g.groups : 1)), block(NTHREADS);     // y = group
    hipError_t err;
    const size_t lds_x3 = 2 * 3 * (size_t)(BM + BN) * X3_ROW;
    if (p.x3 && lds_x3 > lds_epi) lds = lds_x3;
    else if (p.x3) lds = lds_epi;
    static std::atomic<size_t> lds_hw[6];           // per kernel instantiation of this tile configuration
    auto launch = [&](auto k, int which) -> hipError_t {
        hipError_t e2 = bcos_ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds, lds_hw[which]);
        if (e2 != hipSuccess) return e2;
        hipLaunchKernelGGL(k, grid, block, lds, stream, p);
        return hipSuccess;
    };
    if (p.x3 && p.wt3) err = norm ? launch(tapconv_kernel<BM, BN, WAVES_M, WAVES_N, true, 2>, 0)
                                  : launch(tapconv_kernel<BM, BN, WAVES_M, WAVES_N, false, 2>, 1);
    else if (p.x3) err = norm ? launch(tapconv_kernel<BM, BN, WAVES_M, WAVES_N, true, 1>, 2)
                              : launch(tapconv_kernel<BM, BN, WAVES_M, WAVES_N, false, 1>, 3);
    else err = norm ? launch(tapconv_kernel<BM, BN, WAVES_M, WAVES_N, true, 0>, 4)
                    : launch(tapconv_kernel<BM, BN, WAVES_M, WAVES_N, false, 0>, 5);
    if (err != hipSuccess) return bcos_set_hip_error("hipFuncSetAttribute", err);
    err = hipGetLastError();
    if (err != hipSuccess) return bcos_set_hip_error("tapconv launch", err);
    return BCOS_OK;
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
constexpr size_t h2_staging_lds() {      // (sized for the deeper of the norm / no-norm pipelines of the tile)
    constexpr int nbuf = (h2_pipe<BM, BN, WAVES_M, WAVES_N, false>() == 3 || h2_pipe<BM, BN, WAVES_M, WAVES_N, true>() == 3) ? 4 : 2;
    return nbuf * ((size_t)2 * BM * X3_ROW + (size_t)(BN / 32) * 2048) + (size_t)H2_MAX_TAPS * BM * 4;
}

// split-f16 launches (their own tile configurations: only this loop is instantiated for them)
template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch_h2(const KArgs& base, bool norm, hipStream_t stream) {
    KArgs p = base;
    plan_tiles<BM, BN, WAVES_M>(p);
    // staging buffers of the full-height body and of the half-height tail body (its own wave layout and pipeline)
    constexpr int BMS = BM / 2, WMS = (BMS / 32 >= WAVES_M) ? WAVES_M : BMS / 32, WNS = 4 / WMS;
    size_t lds = h2_staging_lds<BM, BN, WAVES_M, WAVES_N>();
    if (BMS >= 32 && BN / WNS >= 32 && h2_staging_lds<BMS, BN, WMS, WNS>() > lds) lds = h2_staging_lds<BMS, BN, WMS, WNS>();
    const size_t lds_epi = epilogue_lds<BM, BN, WAVES_M>();
    if (lds_epi > lds) lds = lds_epi;
    const dim3 grid((unsigned)(p.n_big + p.n_small)), block(NTHREADS);
    static std::atomic<size_t> lds_hw[2];
    auto launch = [&](auto k, int which) -> hipError_t {
        hipError_t e2 = bcos_ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds, lds_hw[which]);
        if (e2 != hipSuccess) return e2;
        hipLaunchKernelGGL(k, grid, block, lds, stream, p);
        return hipSuccess;
    };
    hipError_t err = norm ? launch(tapconv_kernel<BM, BN, WAVES_M, WAVES_N, true, 3>, 0)
                          : launch(tapconv_kernel<BM, BN, WAVES_M, WAVES_N, false, 3>, 1);
    if (err != hipSuccess) return bcos_set_hip_error("hipFuncSetAttribute", err);
    err = hipGetLastError();
    if (err != hipSuccess) return bcos_set_hip_error("tapconv launch", err);
    return BCOS_OK;
}

// split-f16 launches with LDS-DMA staging (tile_body_d): ring + row scales + (channel-chunk-major launches) the tap table
template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch_d(const KArgs& base, bool norm, hipStream_t stream) {
    KArgs p = base;
    plan_tiles<BM, BN, WAVES_M>(p);
    const int ntaps = p.g.TH * p.g.TW;
    const bool kmajor = ntaps > 1 && ntaps <= H2_MAX_TAPS && p.g.C % X3_BK == 0;
    size_t lds = (size_t)d_nslot<BM, BN>() * d_slot_bytes<BM, BN>() + (size_t)BM * 4 + 1024 + (kmajor ? (size_t)ntaps * BM * 4 : 0);    // (the half-height body needs less)
    const size_t lds_epi = epilogue_lds<BM, BN, WAVES_M>();
    if (lds_epi > lds) lds = lds_epi;
    if (const int64_t one = bcos_option(BCOS_OPT_D_ONE_WG)) {      // development switch: LDS request that leaves room for `one` workgroups per CU only
        const size_t want = (size_t)160 * 1024 / (size_t)one - 512;
        if (want > lds) lds = want;
    }
    if (const int64_t kb = bcos_option(BCOS_OPT_LDS_MIN_KB)) { if ((size_t)kb * 1024 > lds) lds = (size_t)kb * 1024; }
    const dim3 grid((unsigned)(p.n_big + p.n_small)), block(NTHREADS);
    static std::atomic<size_t> lds_hw[2];
    auto launch = [&](auto k, int which) -> hipError_t {
        hipError_t e2 = bcos_ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds, lds_hw[which]);
        if (e2 != hipSuccess) return e2;
        hipLaunchKernelGGL(k, grid, block, lds, stream, p);
        return hipSuccess;
    };
    hipError_t err = norm ? launch(tapconv_kernel<BM, BN, WAVES_M, WAVES_N, true, 4>, 0)
                          : launch(tapconv_kernel<BM, BN, WAVES_M, WAVES_N, false, 4>, 1);
    if (err != hipSuccess) return bcos_set_hip_error("hipFuncSetAttribute", err);
    err = hipGetLastError();
    if (err != hipSuccess) return bcos_set_hip_error("tapconv launch", err);
    return BCOS_OK;
}

// split-f16 launches over an LDS-resident input patch (tile_body_p); every tile has BM rows
template <int BM, int BN, int WAVES_M, int WAVES_N, int NI, int PXL, int NTAPS, int T2BW = 0>
int launch_p(const KArgs& base, bool norm, hipStream_t stream) {
    KArgs p = base;
    p.tiles_n = (p.g.Cout + BN - 1) / BN;
    p.n_big = ((p.M + BM - 1) / BM) * p.tiles_n;
    if constexpr (T2BW > 0) {          // 2-D row tiles: (BM / T2BW) x T2BW blocks of every image's row grid
        constexpr int BH = BM / T2BW;
        p.t2_bw = T2BW;
        p.t2_rows = BM;
        p.t2_nbx = (p.g.Q + T2BW - 1) / T2BW;
        p.t2_nb = p.t2_nbx * ((p.g.P + BH - 1) / BH);
        p.n_big = p.g.N * p.t2_nb * p.tiles_n;
    }
    p.n_small = 0;
    p.rows_big = p.M;
    size_t lds = p_lds_bytes<BM, BN, PXL>();
    const size_t lds_epi = epilogue_lds<BM, BN, WAVES_M>();
    if (lds_epi > lds) lds = lds_epi;
    lds = (lds + 15) & ~(size_t)15;
    p.lvl_off = (int)lds;              // row levels + level mask of a tile (tile_body_p), kept across the tile's epilogues
    lds += BM + 16;
    if (const int64_t kb = bcos_option(BCOS_OPT_LDS_MIN_KB)) { if ((size_t)kb * 1024 > lds) lds = (size_t)kb * 1024; }
    const dim3 grid((unsigned)p.n_big), block(NTHREADS);
    static std::atomic<size_t> lds_hw[2];
    auto launch = [&](auto k, int which) -> hipError_t {
        hipError_t e2 = bcos_ensure_dynamic_lds(reinterpret_cast<const void*>(k), lds, lds_hw[which]);
        if (e2 != hipSuccess) return e2;
        hipLaunchKernelGGL(k, grid, block, lds, stream, p);
        return hipSuccess;
    };
    hipError_t err = norm ? launch(tappatch_kernel<BM, BN, WAVES_M, WAVES_N, true, NI, PXL, NTAPS, T2BW>, 0)
                          : launch(tappatch_kernel<BM, BN, WAVES_M, WAVES_N, false, NI, PXL, NTAPS, T2BW>, 1);
    if (err != hipSuccess) return bcos_set_hip_error("hipFuncSetAttribute", err);
    err = hipGetLastError();
    if (err != hipSuccess) return bcos_set_hip_error("tapconv launch", err);
    return BCOS_OK;
}

// Does the input patch of every BM-row tile of the launch fit `px` pixels?  Upper bound over the tile positions: a tile touches
// at most R = (BM + Q - 2) / Q + 1 output rows, which cross at most ceil((R - 1) / P) image boundaries (TH - s extra virtual rows each)
inline bool patch_fits(const bcos_tapconv_geom& g, int BM, int items_px, int lds_px) {
    const int s = g.in_sh;
    const int R = (BM + g.Q - 2) / g.Q + 1;
    const int nb = (R - 1 + g.P - 1) / g.P;
    const int extra = g.TH > s ? g.TH - s : 0;
    const int64_t rows = (int64_t)(R - 1) * s + g.TH + (int64_t)nb * extra;
    const int pw = (g.Q - 1) * s + g.TW;
    int pitch = 1;
    while (pitch < pw) pitch <<= 1;                                  // (rows are stored with a power-of-two pitch, rotated: tile_body_p)
    return rows * pw <= items_px && rows * pitch <= lds_px && (int64_t)g.P * g.Q * 14 >= BM;      // (at most 16 images per tile)
}

}  // namespace

#if BCOS_PHASE_TIMING
// development builds: fetch and reset the phase clocks of tile_body_d (100 MHz wall-clock ticks); every slice of the file owns a copy
// of g_phase, so every slice exports its own bcos_debug_phase_p<slice>
#ifndef BCOS_TAPCONV_PART
#define BCOS_TAPCONV_PART -1
#endif
#define BCOS_PH_CAT(a, b) a##b
#define BCOS_PH_NAME(n) BCOS_PH_CAT(bcos_debug_phase_p, n)
#if BCOS_TAPCONV_PART >= 0
extern "C" int BCOS_PH_NAME(BCOS_TAPCONV_PART)(unsigned long long* out) {
    unsigned long long z[8] = {0, 0, 0, 0, ~0ull, 0, 0, 0};
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(z));
    hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z));
    return 0;
}
#endif
#endif

// The file is compiled either whole or in slices (-DBCOS_TAPCONV_PART=k, bcos_hip/lib.py: build): slice 0 carries the C ABI
// and the small tile configurations, slices 1.. the kernel instantiations of one or two tile configurations each.  The
// launchers cross slices as plain functions taking the launch descriptor by address (every slice compiles the same KArgs).
#ifndef BCOS_TAPCONV_PART
#define BCOS_TAPCONV_PART -1
#endif
#define BCOS_TC_IN(k) (BCOS_TAPCONV_PART == -1 || BCOS_TAPCONV_PART == (k))
#define BCOS_TC_LAUNCHER(name) __attribute__((visibility("hidden"))) int name(const void* kargs, int norm, hipStream_t s)
BCOS_TC_LAUNCHER(bcos_tc_cfg_128x128);
BCOS_TC_LAUNCHER(bcos_tc_cfg_128x64);
BCOS_TC_LAUNCHER(bcos_tc_cfg_128x32);
BCOS_TC_LAUNCHER(bcos_tc_h2_128x256);
BCOS_TC_LAUNCHER(bcos_tc_h2_128x128);
BCOS_TC_LAUNCHER(bcos_tc_h2_128x64);
BCOS_TC_LAUNCHER(bcos_tc_h2_128x32);
BCOS_TC_LAUNCHER(bcos_tc_h2_256x64);
BCOS_TC_LAUNCHER(bcos_tc_h2_256x32);
BCOS_TC_LAUNCHER(bcos_tc_d_128x256);
BCOS_TC_LAUNCHER(bcos_tc_d_128x128);
BCOS_TC_LAUNCHER(bcos_tc_d_128x192);
BCOS_TC_LAUNCHER(bcos_tc_d_128x64);
BCOS_TC_LAUNCHER(bcos_tc_d_128x32);
BCOS_TC_LAUNCHER(bcos_tc_d_256x64);
BCOS_TC_LAUNCHER(bcos_tc_d_256x32);
BCOS_TC_LAUNCHER(bcos_tc_p_128x256_a);
BCOS_TC_LAUNCHER(bcos_tc_p_128x128_a);
BCOS_TC_LAUNCHER(bcos_tc_p_128x128_b);
BCOS_TC_LAUNCHER(bcos_tc_p_128x128_c);
BCOS_TC_LAUNCHER(bcos_tc_p_256x64_a);
BCOS_TC_LAUNCHER(bcos_tc_p2_256x32_t16);
BCOS_TC_LAUNCHER(bcos_tc_p2_256x32_t9);
BCOS_TC_LAUNCHER(bcos_tc_p2_256x64_b32);
#define BCOS_TC_DEFINE(name, call) BCOS_TC_LAUNCHER(name) { return call(*static_cast<const KArgs*>(kargs), norm != 0, s); }
#if BCOS_TC_IN(1)
BCOS_TC_DEFINE(bcos_tc_cfg_128x128, (launch_cfg<128, 128, 2, 2>))
#endif
#if BCOS_TC_IN(2)
BCOS_TC_DEFINE(bcos_tc_cfg_128x64, (launch_cfg<128, 64, 2, 2>))
BCOS_TC_DEFINE(bcos_tc_cfg_128x32, (launch_cfg<128, 32, 4, 1>))
#endif
#if BCOS_TC_IN(3)
BCOS_TC_DEFINE(bcos_tc_h2_128x256, (launch_h2<128, 256, 2, 2>))
BCOS_TC_DEFINE(bcos_tc_h2_128x128, (launch_h2<128, 128, 2, 2>))
#endif
#if BCOS_TC_IN(4)
BCOS_TC_DEFINE(bcos_tc_h2_256x64, (launch_h2<256, 64, 4, 1>))
BCOS_TC_DEFINE(bcos_tc_h2_256x32, (launch_h2<256, 32, 4, 1>))
#endif
#if BCOS_TC_IN(5)
#ifndef D_WIDE_WM
#define D_WIDE_WM 4                // development switch: wave layout of the 128 x 256 LDS-DMA configuration (4 x 1; 2 x 2 measured in round 4)
#define D_WIDE_WN 1
#endif
BCOS_TC_DEFINE(bcos_tc_d_128x256, (launch_d<128, 256, D_WIDE_WM, D_WIDE_WN>))
#endif
#if BCOS_TC_IN(6)
BCOS_TC_DEFINE(bcos_tc_d_128x128, (launch_d<128, 128, 4, 1>))
BCOS_TC_DEFINE(bcos_tc_d_128x64, (launch_d<128, 64, 4, 1>))
#endif
#if BCOS_TC_IN(8)
BCOS_TC_DEFINE(bcos_tc_d_128x192, (launch_d<128, 192, 4, 1>))
#endif
#if BCOS_TC_IN(7)
BCOS_TC_DEFINE(bcos_tc_d_256x64, (launch_d<256, 64, 4, 1>))
BCOS_TC_DEFINE(bcos_tc_d_256x32, (launch_d<256, 32, 4, 1>))
BCOS_TC_DEFINE(bcos_tc_d_128x32, (launch_d<128, 32, 4, 1>))
#endif
// input-patch configurations: <BM, BN, waves M x N, items per thread (128 pixels each), LDS pixels, taps>
#if BCOS_TC_IN(9)
BCOS_TC_DEFINE(bcos_tc_p_128x256_a, (launch_p<128, 256, 2, 2, 2, 256, 9>))      // 14^2: 15 rows of 16
BCOS_TC_DEFINE(bcos_tc_p_256x64_a, (launch_p<256, 64, 4, 1, 5, 640, 9>))        // 56^2: 10 rows of 58 (pitch 64)
BCOS_TC_DEFINE(bcos_tc_p2_256x64_b32, (launch_p<256, 64, 4, 1, 3, 640, 9, 32>))      // 3 x 3 on wide images: 8 x 32 blocks, 10 rows of 34 (pitch 64)
BCOS_TC_DEFINE(bcos_tc_p2_256x32_t9, (launch_p<256, 32, 4, 1, 3, 576, 9, 16>))     // 3 x 3 with <= 32 output channels on wide images (CLIP stem, 32 -> 32 @112^2): 16 x 16 blocks, 18 rows of 18
BCOS_TC_DEFINE(bcos_tc_p2_256x32_t16, (launch_p<256, 32, 4, 1, 3, 608, 16, 16>))   // 4 x 4 taps (the depth-to-space stem gradient): 16 x 16 blocks, 19 rows of 19 (pitch 32)
#endif
#if BCOS_TC_IN(10)
BCOS_TC_DEFINE(bcos_tc_p_128x128_a, (launch_p<128, 128, 2, 2, 2, 256, 9>))      // 14^2
BCOS_TC_DEFINE(bcos_tc_p_128x128_b, (launch_p<128, 128, 2, 2, 2, 448, 9>))      // 7^2: 28 rows of 9 (pitch 16)
BCOS_TC_DEFINE(bcos_tc_p_128x128_c, (launch_p<128, 128, 2, 2, 3, 320, 9>))      // 28^2: 10 rows of 30 (pitch 32)
#endif
#if BCOS_TC_IN(0)
BCOS_TC_DEFINE(bcos_tc_h2_128x64, (launch_h2<128, 64, 2, 2>))
BCOS_TC_DEFINE(bcos_tc_h2_128x32, (launch_h2<128, 32, 4, 1>))

extern "C" int bcos_set_contraction_mode(int mode) {
    if (mode < 0 || mode > 2)
        return bcos_set_error(BCOS_E_INVAL, "bcos_set_contraction_mode: 0 = fp32 MFMA, 1 = split-bf16 MFMA, 2 = split-f16 MFMA");
    g_contraction_mode = mode;
    return BCOS_OK;
}

extern "C" int bcos_get_contraction_mode(void) { return g_contraction_mode; }


namespace {

// wt [rows][Ktot] fp32 -> wt3 [32-row tile][16-k step][plane h|m|l][lane][8 bf16]: lane = (row % 32) + 32 * ((k % 16) / 8),
// i.e. the B operand of v_mfma_f32_32x32x16_bf16 for that (tile, step); rows are padded to a multiple of 128 and k to a
// multiple of 16 with zeros.  Same truncating split as tile_body_x3::split_store (bit-identical products).
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ wt, uint4* __restrict__ wt3, int rows,
                                                            int Ktot, int nk, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;        // (tile, ks, lane)
    if (i >= total) return;
    const int lane = (int)(i & 63);
    const int64_t ts = i >> 6;
    const int ks = (int)(ts % nk);
    const int tile = (int)(ts / nk);
    const int row = tile * 32 + (lane & 31);
    const int k0 = ks * 16 + 8 * (lane >> 5);
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = (row < rows && k0 + e < Ktot) ? wt[(int64_t)row * Ktot + k0 + e] : 0.f;
        h[e] = __float_as_uint(x) & 0xffff0000u;
        const float r1 = x - __uint_as_float(h[e]);
        m[e] = __float_as_uint(r1) & 0xffff0000u;
        l[e] = __float_as_uint(r1 - __uint_as_float(m[e])) & 0xffff0000u;
    }
    auto pack = [](const unsigned (&v)[8]) {
        return uint4{(v[0] >> 16) | v[1], (v[2] >> 16) | v[3], (v[4] >> 16) | v[5], (v[6] >> 16) | v[7]};
    };
    uint4* dst = wt3 + (ts * 3) * 64 + lane;
    dst[0] = pack(h);
    dst[64] = pack(m);
    dst[128] = pack(l);
}

inline int64_t split_bytes(int rows, int Ktot) {
    const int64_t tiles = (((int64_t)rows + 127) / 128) * 4, nk = ((int64_t)Ktot + 15) / 16;
    return tiles * nk * 3 * 1024;
}

}  // namespace

extern "C" int bcos_split_weights_bytes(int rows, int Ktot, int64_t* bytes) {
    if (rows <= 0 || Ktot <= 0 || !bytes) return bcos_set_error(BCOS_E_INVAL, "bcos_split_weights_bytes: bad argument");
    *bytes = split_bytes(rows, Ktot);
    return BCOS_OK;
}

extern "C" int bcos_split_weights(const float* wt, void* wt3, int rows, int Ktot, void* stream) {
    if (!wt || !wt3 || rows <= 0 || Ktot <= 0) return bcos_set_error(BCOS_E_INVAL, "bcos_split_weights: bad argument");
    if (reinterpret_cast<uintptr_t>(wt3) & 15) return bcos_set_error(BCOS_E_INVAL, "bcos_split_weights: wt3 must be 16-byte aligned");
    const int nk = (Ktot + 15) / 16;
    const int64_t total = split_bytes(rows, Ktot) / (3 * 16);
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), wt, reinterpret_cast<uint4*>(wt3), rows, Ktot, nk, total);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return bcos_set_hip_error("split_weights launch", err);
    return BCOS_OK;
}

namespace {

// f16x2 image of wt [rows][Ktot]: [32-row tile][16-k step][plane h|l][lane][8 f16] (B fragments of v_mfma_f32_32x32x16_f16,
// rows padded to a multiple of 128, k to a multiple of 16) followed by the inverse row scales float[padded rows].
// Row r is scaled by 2^e_r so that its max |w| lands in [2^14, 2^15) before the split (exact), cinv[r] = 2^-e_r.
__global__ __launch_bounds__(256) void weight_rowscale_kernel(const float* __restrict__ wt, float* __restrict__ cinv, int rows,
                                                              int rows_pad, int Ktot) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows_pad) return;
    unsigned m = 0u;
    if (row < rows)
        for (int k = lane; k < Ktot; k += 64) m = max(m, __float_as_uint(wt[(int64_t)row * Ktot + k]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    unsigned E = m >> 23;
    E = E < 15u ? 15u : E;
    if (lane == 0) cinv[row] = __uint_as_float((E - 14u) << 23);
}

__global__ __launch_bounds__(256) void split_weights_h2_kernel(const float* __restrict__ wt, uint4* __restrict__ wt2,
                                                               const float* __restrict__ cinv, int rows, int Ktot, int nk,
                                                               int64_t total, int taps, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;        // (tile, ks, lane)
    if (i >= total) return;
    const int lane = (int)(i & 63);
    const int64_t ts = i >> 6;
    const int ks = (int)(ts % nk);
    const int tile = (int)(ts / nk);
    const int row = tile * 32 + (lane & 31);
    // k of the 8 elements: plain order, or (taps > 1) channel-chunk-major: step ks = tap ks % taps of 16-channel chunk ks / taps
    const int k0 = taps > 1 ? (ks % taps) * C + (ks / taps) * 16 + 8 * (lane >> 5) : ks * 16 + 8 * (lane >> 5);
    const float scale = 1.0f / cinv[row];                             // exact: a power of two
    f16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = (row < rows && k0 + e < Ktot) ? wt[(int64_t)row * Ktot + k0 + e] * scale : 0.f;
        const _Float16 hh = (_Float16)x;
        h[e] = hh;
        l[e] = (_Float16)(x - (float)hh);
    }
    uint4* dst = wt2 + (ts * 2) * 64 + lane;
    dst[0] = __builtin_bit_cast(uint4, h);
    dst[64] = __builtin_bit_cast(uint4, l);
}

// both steps in one launch for small weight tensors (one workgroup per 32-row tile: row maxima, then its fragments) -- a training step
// re-splits every weight twice (the forward and the transposed input-gradient image), ~100 images of 37-400 K elements for ViT-Ti:
// half the launches of that for the block weights.  Same arithmetic as the two kernels above.
__global__ __launch_bounds__(256) void split_weights_h2_fused_kernel(const float* __restrict__ wt, uint4* __restrict__ wt2,
                                                                     float* __restrict__ cinv, int rows, int Ktot, int nk, int taps, int C) {
    // grid (32-row tile, K part): every workgroup takes the maxima of its tile's rows (coalesced 16-byte loads, the rows meet in LDS
    // atomics; redundant across the K parts: the tensors are L2-sized) and writes the fragments of its share of the 16-k steps
    __shared__ unsigned s_max[32];
    __shared__ float s_scale[32];
    const int tile = blockIdx.x, tid = threadIdx.x;
    if (tid < 32) s_max[tid] = 0u;
    __syncthreads();
    const int row0 = tile * 32;
    const int live = rows - row0 < 32 ? rows - row0 : 32;
    const float* base = wt + (int64_t)row0 * Ktot;
    if ((Ktot & 3) == 0 && !(reinterpret_cast<uintptr_t>(wt) & 15)) {
        const int per_row = Ktot >> 2, n4 = live * per_row;
        for (int i = tid; i < n4; i += 256) {
            const f32x4 v = reinterpret_cast<const f32x4*>(base)[i];
            unsigned m = 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) m = max(m, __float_as_uint(v[q]) & 0x7fffffffu);
            atomicMax(&s_max[i / per_row], m);
        }
    } else {
        const int n = live * Ktot;
        for (int i = tid; i < n; i += 256) atomicMax(&s_max[i / Ktot], __float_as_uint(base[i]) & 0x7fffffffu);
    }
    __syncthreads();
    if (tid < 32) {
        unsigned E = s_max[tid] >> 23;
        E = E < 15u ? 15u : E;
        const float ci = __uint_as_float((E - 14u) << 23);
        if (blockIdx.y == 0) cinv[row0 + tid] = ci;
        s_scale[tid] = 1.0f / ci;                                     // exact: a power of two
    }
    __syncthreads();
    const int lane = tid & 63;
    const int row = row0 + (lane & 31);
    const float scale = s_scale[lane & 31];
    for (int ks = (int)blockIdx.y * 4 + (tid >> 6); ks < nk; ks += 4 * (int)gridDim.y) {
        const int k0 = taps > 1 ? (ks % taps) * C + (ks / taps) * 16 + 8 * (lane >> 5) : ks * 16 + 8 * (lane >> 5);
        f16x8 h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = (row < rows && k0 + e < Ktot) ? wt[(int64_t)row * Ktot + k0 + e] * scale : 0.f;
            const _Float16 hh = (_Float16)x;
            h[e] = hh;
            l[e] = (_Float16)(x - (float)hh);
        }
        uint4* dst = wt2 + ((int64_t)tile * nk + ks) * 2 * 64 + lane;
        dst[0] = __builtin_bit_cast(uint4, h);
        dst[64] = __builtin_bit_cast(uint4, l);
    }
}

__host__ __device__ inline int64_t h2_tiles(int rows) { return (((int64_t)rows + 127) / 128) * 4; }
__host__ __device__ inline int64_t h2_image_bytes(int rows, int Ktot) { return h2_tiles(rows) * (((int64_t)Ktot + 15) / 16) * 2 * 1024; }

}  // namespace

extern "C" int bcos_split_weights_f16x2_bytes(int rows, int Ktot, int64_t* bytes) {
    if (rows <= 0 || Ktot <= 0 || !bytes) return bcos_set_error(BCOS_E_INVAL, "bcos_split_weights_f16x2_bytes: bad argument");
    *bytes = h2_image_bytes(rows, Ktot) + h2_tiles(rows) * 32 * 4;
    return BCOS_OK;
}

extern "C" int bcos_split_weights_f16x2(const float* wt, void* wt2, int rows, int Ktot, void* stream) {
    return bcos_split_weights_f16x2_conv(wt, wt2, rows, 1, Ktot, stream);
}

extern "C" int bcos_split_weights_f16x2_conv(const float* wt, void* wt2, int rows, int taps, int C, void* stream) {
    if (taps <= 0 || C <= 0) return bcos_set_error(BCOS_E_INVAL, "bcos_split_weights_f16x2: bad argument");
    const int Ktot = taps * C;
    // the kernel walks K channel-chunk-major exactly when this holds (tile_body_h2: kmajor)
    const int ktaps = (taps > 1 && taps <= H2_MAX_TAPS && C % 16 == 0) ? taps : 1;
    if (!wt || !wt2 || rows <= 0 || Ktot <= 0) return bcos_set_error(BCOS_E_INVAL, "bcos_split_weights_f16x2: bad argument");
    if (reinterpret_cast<uintptr_t>(wt2) & 15) return bcos_set_error(BCOS_E_INVAL, "bcos_split_weights_f16x2: image must be 16-byte aligned");
    const int nk = (Ktot + 15) / 16;
    const int rows_pad = (int)h2_tiles(rows) * 32;
    float* cinv = reinterpret_cast<float*>(static_cast<char*>(wt2) + h2_image_bytes(rows, Ktot));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // (one workgroup per 32 rows walks its whole K: beyond ~150 K elements or K > 1024 the two wide launches are faster -- at 2^20
    //  elements the one-launch form cost a ResNet-50 training step 43 us per image, 4.6 ms per step)
    if ((int64_t)rows_pad * Ktot <= 160 * 1024 && Ktot <= 1024) {
        int parts = nk / 8;                                   // two 16-k steps per wave and workgroup
        parts = parts < 1 ? 1 : (parts > 8 ? 8 : parts);
        hipLaunchKernelGGL(split_weights_h2_fused_kernel, dim3((unsigned)h2_tiles(rows), (unsigned)parts), dim3(256), 0, s, wt,
                           reinterpret_cast<uint4*>(wt2), cinv, rows, Ktot, nk, ktaps, C);
    } else {
        hipLaunchKernelGGL(weight_rowscale_kernel, dim3((unsigned)((rows_pad + 3) / 4)), dim3(256), 0, s, wt, cinv, rows, rows_pad, Ktot);
        const int64_t total = h2_tiles(rows) * nk * 64;
        hipLaunchKernelGGL(split_weights_h2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, wt,
                           reinterpret_cast<uint4*>(wt2), cinv, rows, Ktot, nk, total, ktaps, C);
    }
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return bcos_set_hip_error("split_weights_f16x2 launch", err);
    return BCOS_OK;
}

// ---- banks + images of many layers from one call (bcos_weight_prep_batch) ---------------------------------------------------------------
namespace {
constexpr int PREP_KCH = 1024;          // bank elements of a row per workgroup of the gather
// (1) grid (32-row tile x K part, jobs): gather the part's elements of the tile's rows from the strided source into the bank (coalesced
//     along K) and fold their maxima into row_max[job.row_offset + row] (zero-filled by the caller): per thread, per wave, then ONE atomic per
//     row and workgroup.
__global__ __launch_bounds__(256) void weight_prep_gather_kernel(const bcos_weight_prep_job* __restrict__ jobs, unsigned* __restrict__ row_max) {
    const bcos_weight_prep_job& jb = jobs[blockIdx.y];
    const int rows = jb.rows, Cp = jb.Cp, channels = jb.channels;
    const int Ktot = jb.taps * Cp;
    const int kparts = (Ktot + PREP_KCH - 1) / PREP_KCH;
    const int tile = (int)blockIdx.x / kparts, kpart = (int)blockIdx.x - tile * kparts;
    const int row0 = tile * 32;
    if (row0 >= rows) return;
    const int live = rows - row0 < 32 ? rows - row0 : 32;
    const int k_lo = kpart * PREP_KCH, k_hi = k_lo + PREP_KCH < Ktot ? k_lo + PREP_KCH : Ktot;
    const float* __restrict__ src = jb.src;
    float* __restrict__ bank = jb.bank;
    const int tid = threadIdx.x;
    // (t, c) of this thread's elements do not depend on the row: decoded once
    int off[PREP_KCH / 256];
    bool in[PREP_KCH / 256];
#pragma unroll
    for (int q = 0; q < PREP_KCH / 256; ++q) {
        const int k = k_lo + q * 256 + tid;
        const int t = k / Cp, c = k - t * Cp;
        in[q] = k < k_hi && c < channels && t < jb.taps;
        off[q] = in[q] ? c * jb.ch_stride + jb.tap_offset[t] : 0;
    }
    // rows in groups of eight: eight independent loads in flight per thread and K part, then the stores and the rows' maxima
#pragma unroll
    for (int q = 0; q < PREP_KCH / 256; ++q) {
        const int k = k_lo + q * 256 + tid;
        if (k_lo + q * 256 >= k_hi) break;
        for (int r0 = 0; r0 < live; r0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = (in[q] && r0 + u < live) ? src[(int64_t)(row0 + r0 + u) * jb.row_stride + off[q]] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (r0 + u < live) {            // (uniform over the workgroup)
                    if (k < k_hi) bank[(int64_t)(row0 + r0 + u) * Ktot + k] = v[u];
                    unsigned m = k < k_hi ? __float_as_uint(v[u]) & 0x7fffffffu : 0u;
#pragma unroll
                    for (int o = 32; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
                    if ((tid & 63) == 0 && m) atomicMax(row_max + jb.row_offset + row0 + r0 + u, m);
                }
            }
        }
    }
}

// (2) grid (32-row tile x group of 16-k steps, jobs): the image of the bank -- the arithmetic of weight_rowscale_kernel + split_weights_h2_kernel
//     on the bank's values and the rows' maxima: that pair's image bit for bit.
__global__ __launch_bounds__(256) void weight_prep_split_kernel(const bcos_weight_prep_job* __restrict__ jobs, const unsigned* __restrict__ row_max) {
    const bcos_weight_prep_job& jb = jobs[blockIdx.y];
    if (jb.image == nullptr) return;
    const int rows = jb.rows, taps = jb.taps, Cp = jb.Cp;
    const int Ktot = taps * Cp;
    const int nk = (Ktot + 15) / 16;
    constexpr int KS_PER_WG = PREP_KCH / 16;                    // 64 steps (16 per wave)
    const int kgroups = (nk + KS_PER_WG - 1) / KS_PER_WG;
    const int tiles = (int)h2_tiles(rows);                       // (the image is padded to a multiple of 128 rows: empty tiles hold zeros)
    const int tile = (int)blockIdx.x / kgroups, kgroup = (int)blockIdx.x - tile * kgroups;
    if (tile >= tiles) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int row = tile * 32 + (lane & 31);
    unsigned E = (row < rows ? row_max[jb.row_offset + row] : 0u) >> 23;
    E = E < 15u ? 15u : E;
    const float ci = __uint_as_float((E - 14u) << 23);
    const float scale = 1.0f / ci;                                // exact: a power of two
    uint4* wt2 = reinterpret_cast<uint4*>(jb.image);
    if (kgroup == 0 && tid < 32) reinterpret_cast<float*>(static_cast<char*>(jb.image) + h2_image_bytes(rows, Ktot))[tile * 32 + tid] = ci;
    const float* __restrict__ bank = jb.bank;
    const int ktaps = (taps > 1 && taps <= H2_MAX_TAPS && Cp % 16 == 0) ? taps : 1;      // (bcos_split_weights_f16x2_conv: channel-chunk-major K)
    const int ks_hi = (kgroup + 1) * KS_PER_WG < nk ? (kgroup + 1) * KS_PER_WG : nk;
    for (int ks = kgroup * KS_PER_WG + (tid >> 6); ks < ks_hi; ks += 4) {
        const int k0 = ktaps > 1 ? (ks % ktaps) * Cp + (ks / ktaps) * 16 + 8 * (lane >> 5) : ks * 16 + 8 * (lane >> 5);
        f16x8 h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = (row < rows && k0 + e < Ktot) ? bank[(int64_t)row * Ktot + k0 + e] * scale : 0.f;
            const _Float16 hh = (_Float16)x;
            h[e] = hh;
            l[e] = (_Float16)(x - (float)hh);
        }
        uint4* dst = wt2 + ((int64_t)tile * nk + ks) * 2 * 64 + lane;
        dst[0] = __builtin_bit_cast(uint4, h);
        dst[64] = __builtin_bit_cast(uint4, l);
    }
}
}  // namespace

extern "C" int bcos_weight_prep_batch(const bcos_weight_prep_job* jobs, int njobs, int max_rows, int max_ktot, uint32_t* row_max, void* stream) {
    if (!jobs || !row_max || njobs <= 0 || njobs > 65535 || max_rows <= 0 || max_ktot <= 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_weight_prep_batch: bad argument");
    if (reinterpret_cast<uintptr_t>(jobs) & 7) return bcos_set_error(BCOS_E_INVAL, "bcos_weight_prep_batch: job table must be 8-byte aligned");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t tiles = h2_tiles(max_rows);
    const int64_t kparts = ((int64_t)max_ktot + PREP_KCH - 1) / PREP_KCH;
    if (tiles * kparts > 0x7fffffff) return bcos_set_error(BCOS_E_NOSUP, "bcos_weight_prep_batch: too large");
    hipLaunchKernelGGL(weight_prep_gather_kernel, dim3((unsigned)(tiles * kparts), (unsigned)njobs), dim3(256), 0, s, jobs, row_max);
    hipLaunchKernelGGL(weight_prep_split_kernel, dim3((unsigned)(tiles * kparts), (unsigned)njobs), dim3(256), 0, s, jobs, row_max);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return bcos_set_hip_error("weight_prep_batch launch", err);
    return BCOS_OK;
}

// per-pixel max |x| bit patterns of an NHWC tensor (for tensors whose producer is not a tapconv epilogue)
namespace {
__global__ __launch_bounds__(256) void rows_absmax_kernel(const float* __restrict__ x, unsigned* __restrict__ out, int64_t rows,
                                                          int C, int pitch) {
    // LPR lanes per row (power of two <= 64), float4 loads
    int lpr = 1;
    while (lpr < 64 && lpr * 4 < C) lpr <<= 1;
    const int rpw = 64 / lpr;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t row = wave * rpw + lane / lpr;
    const int sub = lane % lpr;
    unsigned m = 0u;
    if (row < rows) {
        const float* src = x + row * pitch;
        for (int c = sub * 4; c < C; c += lpr * 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
#pragma unroll
            for (int q = 0; q < 4; ++q) m = max(m, __float_as_uint(v[q]) & 0x7fffffffu);
        }
    }
    for (int o = lpr >> 1; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if (row < rows && sub == 0) out[row] = m;
}
}  // namespace

namespace {
__global__ __launch_bounds__(1024) void image_absmax_kernel(const unsigned* __restrict__ absmax, unsigned* __restrict__ out,
                                                            unsigned* __restrict__ out_min, int hw) {
    // one workgroup of 16 waves per image, four independent loads in flight per thread: the kernel is one load latency long.
    // out_min (optional): the smallest NONZERO per-pixel maximum of the image (0xffffffff: every pixel is zero) -- with the
    // image maximum, the dynamic range of the pixels inside the image (bcos_operands.a_imgmin)
    __shared__ unsigned s_max[16], s_min[16];
    const unsigned* src = absmax + (size_t)blockIdx.x * hw;
    unsigned v = 0u, w = 0xffffffffu;
    for (int i0 = 0; i0 < hw; i0 += 4096) {
        unsigned u[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + q * 1024 + (int)threadIdx.x;
            u[q] = i < hw ? src[i] : 0u;
        }
        v = max(max(v, u[0]), max(max(u[1], u[2]), u[3]));
#pragma unroll
        for (int q = 0; q < 4; ++q) w = min(w, u[q] - 1u);      // (0 - 1 wraps to the top: zero pixels never win)
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        v = max(v, (unsigned)__shfl_xor((int)v, o));
        w = min(w, (unsigned)__shfl_xor((int)w, o));
    }
    if ((threadIdx.x & 63) == 0) { s_max[threadIdx.x >> 6] = v; s_min[threadIdx.x >> 6] = w; }
    __syncthreads();
    if (threadIdx.x < 16) {
        unsigned m = s_max[threadIdx.x], mn = s_min[threadIdx.x];
#pragma unroll
        for (int o = 8; o; o >>= 1) {
            m = max(m, (unsigned)__shfl_xor((int)m, o));
            mn = min(mn, (unsigned)__shfl_xor((int)mn, o));
        }
        if (threadIdx.x == 0) {
            out[blockIdx.x] = m;
            if (out_min) out_min[blockIdx.x] = mn == 0xffffffffu ? mn : mn + 1u;
        }
    }
}
}  // namespace

namespace {
// the same range from SEVERAL workgroups per image (grid: image x part), for images of many pixels or a device that is busy with
// another stream's launch: the parts meet by atomicMax in ZERO-FILLED words -- the minimum over the nonzero pixels as its complement
// (~v; 0 = no nonzero pixel), the form the producing epilogues leave (bcos_epilogue.out_imgmin_c)
__global__ __launch_bounds__(256) void image_absrange_c_kernel(const unsigned* __restrict__ absmax, unsigned* __restrict__ out_max,
                                                               unsigned* __restrict__ out_min_c, int hw, int per_part) {
    __shared__ unsigned s_max[4], s_min[4];
    const unsigned* src = absmax + (size_t)blockIdx.x * hw;
    const int lo = (int)blockIdx.y * per_part, hi = lo + per_part < hw ? lo + per_part : hw;
    unsigned v = 0u, w = 0xffffffffu;
    for (int i0 = lo; i0 < hi; i0 += 1024) {
        unsigned u[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + q * 256 + (int)threadIdx.x;
            u[q] = i < hi ? src[i] : 0u;
        }
        v = max(max(v, u[0]), max(max(u[1], u[2]), u[3]));
#pragma unroll
        for (int q = 0; q < 4; ++q) w = min(w, u[q] - 1u);      // (0 - 1 wraps to the top: zero pixels never win)
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        v = max(v, (unsigned)__shfl_xor((int)v, o));
        w = min(w, (unsigned)__shfl_xor((int)w, o));
    }
    if ((threadIdx.x & 63) == 0) { s_max[threadIdx.x >> 6] = v; s_min[threadIdx.x >> 6] = w; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned m = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
        const unsigned mn = min(min(s_min[0], s_min[1]), min(s_min[2], s_min[3]));
        if (m) atomicMax(out_max + blockIdx.x, m);
        if (mn != 0xffffffffu) atomicMax(out_min_c + blockIdx.x, ~(mn + 1u));
    }
}
}  // namespace

extern "C" int bcos_image_absrange_c(const uint32_t* absmax, uint32_t* out_max, uint32_t* out_min_c, int n_images, int pixels_per_image,
                                     void* stream) {
    if (!absmax || !out_max || !out_min_c || n_images <= 0 || pixels_per_image <= 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_image_absrange_c: bad argument");
    // parts of >= 2 048 pixels, as many as put ~1 024 workgroups on the device
    int parts = (pixels_per_image + 2047) / 2048;
    const int want = (1024 + n_images - 1) / n_images;
    parts = parts < 1 ? 1 : (parts > want ? want : parts);
    const int per_part = (((pixels_per_image + parts - 1) / parts) + 255) & ~255;
    parts = (pixels_per_image + per_part - 1) / per_part;
    hipLaunchKernelGGL(image_absrange_c_kernel, dim3((unsigned)n_images, (unsigned)parts), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       absmax, out_max, out_min_c, pixels_per_image, per_part);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return bcos_set_hip_error("image_absrange_c launch", err);
    return BCOS_OK;
}

extern "C" int bcos_image_absmax(const uint32_t* absmax, uint32_t* out, int n_images, int pixels_per_image, void* stream) {
    return bcos_image_absrange(absmax, out, nullptr, n_images, pixels_per_image, stream);
}

extern "C" int bcos_image_absrange(const uint32_t* absmax, uint32_t* out_max, uint32_t* out_min, int n_images, int pixels_per_image,
                                   void* stream) {
    uint32_t* out = out_max;
    if (!absmax || !out || n_images <= 0 || pixels_per_image <= 0) return bcos_set_error(BCOS_E_INVAL, "bcos_image_absmax: bad argument");
    hipLaunchKernelGGL(image_absmax_kernel, dim3((unsigned)n_images), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), absmax, out,
                       out_min, pixels_per_image);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return bcos_set_hip_error("image_absmax launch", err);
    return BCOS_OK;
}

extern "C" int bcos_rows_absmax(const float* x, uint32_t* out, int64_t rows, int C, int pitch, void* stream) {
    if (!x || !out || rows <= 0 || C <= 0 || C % 4 != 0) return bcos_set_error(BCOS_E_INVAL, "bcos_rows_absmax: bad argument");
    if (pitch == 0) pitch = C;
    if (pitch % 4 != 0 || pitch < C || (reinterpret_cast<uintptr_t>(x) & 15))
        return bcos_set_error(BCOS_E_INVAL, "bcos_rows_absmax: rows must be 16-byte addressable");
    int lpr = 1;
    while (lpr < 64 && lpr * 4 < C) lpr <<= 1;
    const int64_t rows_per_block = 4 * (64 / lpr);
    hipLaunchKernelGGL(rows_absmax_kernel, dim3((unsigned)((rows + rows_per_block - 1) / rows_per_block)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x, out, rows, C, pitch);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return bcos_set_hip_error("rows_absmax launch", err);
    return BCOS_OK;
}

extern "C" int bcos_tapconv(const float* a, const float* wt, const bcos_tapconv_geom* geom,
                            const bcos_epilogue* epi, void* stream) {
    bcos_operands o = {a, nullptr, wt, nullptr, nullptr, BCOS_CONTRACT_DEFAULT, nullptr, nullptr};
    return bcos_tapconv_ops(&o, geom, epi, stream);
}

extern "C" int bcos_tapconv_presplit(const float* a, const float* wt, const void* wt3, const bcos_tapconv_geom* geom,
                                     const bcos_epilogue* epi, void* stream) {
    bcos_operands o = {a, nullptr, wt, wt3, nullptr, BCOS_CONTRACT_DEFAULT, nullptr, nullptr};
    return bcos_tapconv_ops(&o, geom, epi, stream);
}

namespace { thread_local bool t_query_image_range = false; }

extern "C" int bcos_tapconv_ops(const bcos_operands* ops, const bcos_tapconv_geom* geom, const bcos_epilogue* epi, void* stream);

// 1: a bcos_tapconv_ops call with these arguments folds the per-image range of its out_absmax into bcos_epilogue.out_imgmax / out_imgmin_c
// itself; 0: it does not (general epilogue, depth-to-space or grouped launch, fewer than 19 rows per image, no out_absmax) and rejects
// the two fields; < 0: the arguments are invalid anyway.  Nothing is launched: the call walks bcos_tapconv_ops' own validation and
// epilogue selection and stops ahead of the dispatch.
extern "C" int bcos_tapconv_fuses_image_range(const bcos_operands* ops, const bcos_tapconv_geom* geom, const bcos_epilogue* epi) {
    t_query_image_range = true;
    const int rc = bcos_tapconv_ops(ops, geom, epi, nullptr);
    t_query_image_range = false;
    return rc;
}

extern "C" int bcos_tapconv_ops(const bcos_operands* ops, const bcos_tapconv_geom* geom, const bcos_epilogue* epi, void* stream) {
    if (!ops || !geom || !epi) return bcos_set_error(BCOS_E_INVAL, "bcos_tapconv: NULL argument");
    const float* a = ops->a;
    const float* wt = ops->wt;
    const void* wt3 = ops->wt_bf16x3;
    if (!a || !wt) return bcos_set_error(BCOS_E_INVAL, "bcos_tapconv: NULL operand");
    if (ops->contraction < 0 || ops->contraction > BCOS_CONTRACT_F16X2)
        return bcos_set_error(BCOS_E_INVAL, "bcos_tapconv: bad contraction selector");
    const int mode = ops->contraction == BCOS_CONTRACT_DEFAULT ? g_contraction_mode.load(std::memory_order_relaxed)
                                                               : ops->contraction - 1;
    const bcos_tapconv_geom& g = *geom;
    if (g.N <= 0 || g.H <= 0 || g.W <= 0 || g.C <= 0 || g.P <= 0 || g.Q <= 0 || g.TH <= 0 || g.TW <= 0 ||
        g.Cout <= 0 || g.OH <= 0 || g.OW <= 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_tapconv: non-positive dimension");
    if (g.C % 4 != 0) return bcos_set_error(BCOS_E_INVAL, "bcos_tapconv: C must be a multiple of 4");
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(wt)) & 15)
        return bcos_set_error(BCOS_E_INVAL, "bcos_tapconv: operands must be 16-byte aligned");
    const int64_t M64 = (int64_t)g.N * g.P * g.Q;
    if (M64 >= (int64_t)1 << 31) return bcos_set_error(BCOS_E_NOSUP, "bcos_tapconv: more than 2^31 rows");
    if (!epi->out && !epi->out2 && !epi->scale_out)
        return bcos_set_error(BCOS_E_INVAL, "bcos_tapconv: no output buffer");
    if (epi->max_out < 0) return bcos_set_error(BCOS_E_INVAL, "bcos_tapconv: negative max_out");
    if ((epi->flags & BCOS_EPI_MUL_FROM_ACT) && (!epi->mul || !epi->mul_norm || (epi->flags & BCOS_EPI_GATE2_FROM_MUL)))
        return bcos_set_error(BCOS_E_INVAL, "bcos_tapconv: BCOS_EPI_MUL_FROM_ACT needs mul and mul_norm and excludes GATE2_FROM_MUL");
    if (epi->max_out > 1) {
        if ((epi->max_out != 2 && epi->max_out != 4) || g.Cout % 4 != 0)
            return bcos_set_error(BCOS_E_NOSUP, "bcos_tapconv: fused MaxOut needs max_out in {2, 4} and Cout % 4 == 0");
        if (epi->addend || epi->mul || epi->mul2 || epi->gate2 || epi->relu_gate || epi->out2 || epi->ch_scale || epi->ch_shift ||
            epi->relu || epi->out_absmax || epi->out2_absmax || (epi->flags & ~(BCOS_EPI_FORCE_POW | BCOS_EPI_UNIT_NORM_W)))
            return bcos_set_error(BCOS_E_NOSUP, "bcos_tapconv: fused MaxOut supports bias, the B-cos scale, out, scale_out and norm_out only");
        if ((reinterpret_cast<uintptr_t>(epi->scale_out) & 15))
            return bcos_set_error(BCOS_E_INVAL, "bcos_tapconv: scale_out must be 16-byte aligned");
    }
    if (epi->a_sumsq && epi->bcos_mode == BCOS_NONE)
        return bcos_set_error(BCOS_E_INVAL, "bcos_tapconv: a_sumsq belongs to a B-cos launch (bcos_mode != BCOS_NONE)");
    if ((epi->row_scale || epi->a_sumsq) && (g.groups > 1 || g.out_cgroup != 0))
        return bcos_set_error(BCOS_E_NOSUP, "bcos_tapconv: row_scale / a_sumsq exclude grouped and depth-to-space launches");
    if (epi->addend_sub < 0 || (epi->addend_sub > 1 && (!epi->addend || g.out_cgroup != 0 || g.groups > 1 || epi->bcos_mode != BCOS_NONE)))
        return bcos_set_error(BCOS_E_INVAL, "bcos_tapconv: addend_sub > 1 needs an addend and a plain gradient launch (bcos_mode BCOS_NONE, "
                                            "no out_cgroup, no groups)");
    // the last row/col written must be inside the output tensor
    if ((g.P - 1) * g.out_sh + g.out_h0 >= g.OH || (g.Q - 1) * g.out_sw + g.out_w0 >= g.OW || g.out_h0 < 0 ||
        g.out_w0 < 0)
        return bcos_set_error(BCOS_E_INVAL, "bcos_tapconv: output mapping outside [OH,OW]");
    const int G = g.groups > 1 ? g.groups : 1;
    if (g.groups < 0 || g.groups > 65535) return bcos_set_error(BCOS_E_INVAL, "bcos_tapconv: bad group count");
    if (G > 1 && (g.out_cgroup != 0 || epi->max_out > 1 || epi->out_absmax || epi->out2_absmax))
        return bcos_set_error(BCOS_E_NOSUP, "bcos_tapconv: grouped launches exclude out_cgroup, fused MaxOut and *_absmax");
    if (g.out_cgroup != 0) {     // depth to space: columns = (output parity class, channel)
        if (g.out_cgroup < 0 || g.out_cgroup % 4 != 0 || g.out_sh <= 0 || g.out_sw <= 0 || g.out_h0 != 0 || g.out_w0 != 0 ||
            g.Cout != g.out_sh * g.out_sw * g.out_cgroup || g.P * g.out_sh > g.OH || g.Q * g.out_sw > g.OW ||
            (g.out_pitch != 0 && g.out_pitch < g.out_cgroup))
            return bcos_set_error(BCOS_E_INVAL, "bcos_tapconv: out_cgroup needs Cout == out_sh * out_sw * out_cgroup, out_cgroup % 4 == 0, "
                                                "zero output offsets, P * out_sh <= OH, Q * out_sw <= OW and out_pitch >= out_cgroup");
        if (epi->max_out > 1 || epi->out_absmax || epi->out2_absmax || epi->norm_out || epi->bias || epi->ch_scale || epi->ch_shift ||
            (epi->flags & BCOS_EPI_MUL_FROM_ACT) || epi->bcos_mode != BCOS_NONE)
            return bcos_set_error(BCOS_E_NOSUP, "bcos_tapconv: out_cgroup launches are plain gradient launches (addend / mul / mul2 / "
                                                "gate2 / out / out2 only)");
    }

    KArgs p;
    p.a = a;
    p.wt = wt;
    p.g = g;
    if (p.g.a_pitch == 0) p.g.a_pitch = G * g.C;
    if (p.g.out_pitch == 0) p.g.out_pitch = g.out_cgroup > 0 ? g.out_cgroup : G * g.Cout;
    if (p.g.norm_pitch == 0) p.g.norm_pitch = G;
    if (p.g.a_pitch % 4 != 0 || p.g.a_pitch < G * g.C) return bcos_set_error(BCOS_E_INVAL, "bcos_tapconv: bad a_pitch");
    if (epi->max_out > 1 && g.out_pitch == 0) p.g.out_pitch = g.Cout / epi->max_out;
    if (g.out_cgroup == 0 && p.g.out_pitch < G * g.Cout / (epi->max_out > 1 ? epi->max_out : 1))
        return bcos_set_error(BCOS_E_INVAL, "bcos_tapconv: bad out_pitch");
    if (p.g.norm_pitch < G) return bcos_set_error(BCOS_E_INVAL, "bcos_tapconv: norm_pitch smaller than the group count");
    p.e = *epi;
    p.M = (int)M64;
    p.PQ = g.P * g.Q;
    p.Ktot = g.TH * g.TW * g.C;
    p.cpt = g.C / 4;
    p.nchunks = p.Ktot / 4;
    p.nk = (p.nchunks + 7) / 8;
    p.tiles_n = p.n_big = p.n_small = p.rows_big = 0;
    p.uniform_tap = (g.C % BK == 0) ? 1 : 0;
    p.x3 = mode >= 1 ? 1 : 0;
    p.h2 = 0;
    p.a_absmax = nullptr;
    p.a_imgmax = nullptr;
    p.a_imgmin = nullptr;
    p.a_imgmin_c = nullptr;
    p.lvl_off = 0;
    p.lvl_on = bcos_option(BCOS_OPT_PATCH_LEVELS) != 0;
    p.t2_bw = p.t2_nbx = p.t2_nb = p.t2_rows = 0;
    p.absmax_bytes = 0;
    p.wt2 = nullptr;
    p.wt2_bytes = 0;
    p.wt2_cinv = nullptr;
    {   // the split-bf16 path addresses its operands through 32-bit buffer offsets: keep each launch below 2 GiB of A
        // by splitting the batch (every tensor of the call is per-image separable); fall back to fp32 MFMA otherwise
        const int64_t img_bytes = (int64_t)g.H * g.W * p.g.a_pitch * 4;
        const int64_t a_bytes = img_bytes * g.N, wt_bytes = (int64_t)G * g.Cout * p.Ktot * 4;
        const int64_t lim = (int64_t)1 << 31;
        // (BCOS_OPT_SPLIT_LIMIT: tests lower the chunking threshold to drive this path with small tensors)
        const int64_t chunk_lim = bcos_option(BCOS_OPT_SPLIT_LIMIT);
        if (p.x3 && a_bytes >= chunk_lim && g.N > 1 && img_bytes < chunk_lim && wt_bytes < lim) {
            const int per = (int)((chunk_lim - 1) / img_bytes);
            for (int n0 = 0; n0 < g.N; n0 += per) {
                bcos_tapconv_geom g2 = p.g;
                bcos_epilogue e2 = *epi;
                g2.N = g.N - n0 < per ? g.N - n0 : per;
                const int64_t opix = (int64_t)n0 * g.OH * g.OW;
                const float** cin[] = {&e2.addend, &e2.mul, &e2.mul2, &e2.gate2, &e2.relu_gate};
                for (const float** q : cin) if (*q) *q += opix * p.g.out_pitch;
                if (e2.addend && e2.addend_sub > 1) {      // subsampled addend: its own image size
                    const int sb = e2.addend_sub;
                    e2.addend = epi->addend + (int64_t)n0 * ((g.OH + sb - 1) / sb) * ((g.OW + sb - 1) / sb) * p.g.out_pitch;
                }
                float** cout[] = {&e2.out, &e2.out2, &e2.scale_out};
                for (float** q : cout) if (*q) *q += opix * p.g.out_pitch;
                if (e2.norm_out) e2.norm_out += opix * p.g.norm_pitch;
                if (e2.mul_norm) e2.mul_norm += opix;
                if (e2.row_scale) e2.row_scale += opix;
                if (e2.a_sumsq) e2.a_sumsq += opix;
                if (e2.out_absmax) e2.out_absmax += opix;
                if (e2.out2_absmax) e2.out2_absmax += opix;
                if (e2.out_imgmax) e2.out_imgmax += n0;
                if (e2.out_imgmin_c) e2.out_imgmin_c += n0;
                if (e2.rowadd) e2.rowadd += opix * p.g.out_pitch;
                if (e2.rowadd_scale) e2.rowadd_scale += opix;
                bcos_operands o2 = *ops;
                o2.a = a + (int64_t)n0 * g.H * g.W * p.g.a_pitch;
                if (o2.a_absmax) o2.a_absmax += (int64_t)n0 * g.H * g.W;
                if (o2.a_imgmax) o2.a_imgmax += n0;          // (per-image maxima are indexed by the chunk's local image index)
                if (o2.a_imgmin) o2.a_imgmin += n0;
                if (o2.a_imgmin_c) o2.a_imgmin_c += n0;
                const int rc = bcos_tapconv_ops(&o2, &g2, &e2, stream);
                if (rc != BCOS_OK) return rc;
            }
            return BCOS_OK;
        }
        if (a_bytes >= lim || wt_bytes >= lim) p.x3 = 0;
        p.a_bytes = (unsigned)(a_bytes < lim ? a_bytes : 0);
        p.wt_bytes = (unsigned)(wt_bytes < lim ? wt_bytes : 0);
        const int64_t w3b = split_bytes(G * g.Cout, p.Ktot);
        // (grouped launches index the image by global weight row: a group's rows must start on a 32-row fragment tile)
        const bool unit_w = (epi->flags & BCOS_EPI_UNIT_NORM_W) != 0;     // norms come from the fp32 weight rows in the staging registers
        p.wt3 = (p.x3 && !unit_w && wt3 && w3b < lim && !(reinterpret_cast<uintptr_t>(wt3) & 15) && (G == 1 || g.Cout % 32 == 0)) ? wt3 : nullptr;
        p.wt3_bytes = (unsigned)(w3b < lim ? w3b : 0);
        const int64_t w2b = h2_image_bytes(g.Cout, p.Ktot), pixb = (int64_t)g.N * g.H * g.W * 4;
        // below K = 256 a launch is HBM-bound and the bf16x3 loop (no operand maxima to produce) is as fast, unless the caller
        // insists on f16x2; at K = 256 the six bf16 products still occupy a third of the SIMD cycles (14^2 layers of ResNet-50:
        // -12 % per launch with three f16 products).  The stem's K is all taps over 8 channels: compute-bound at any K.
        const bool h2_pays = p.Ktot >= 256 || (g.C <= 16 && p.Ktot >= 128) || ops->contraction == BCOS_CONTRACT_F16X2;
        if (mode == 2 && G == 1 && h2_pays && !unit_w && p.x3 && ops->a_absmax && ops->wt_f16x2 && w2b < lim && pixb < lim &&
            !(reinterpret_cast<uintptr_t>(ops->wt_f16x2) & 15)) {
            p.h2 = 1;
            p.a_absmax = ops->a_absmax;
            p.a_imgmax = ops->a_imgmax;
            p.a_imgmin = ops->a_imgmax ? ops->a_imgmin : nullptr;
            p.a_imgmin_c = (ops->a_imgmax && !ops->a_imgmin) ? ops->a_imgmin_c : nullptr;
            p.absmax_bytes = (unsigned)pixb;
            p.wt2 = ops->wt_f16x2;
            p.wt2_bytes = (unsigned)w2b;
            p.wt2_cinv = reinterpret_cast<const float*>(static_cast<const char*>(ops->wt_f16x2) + w2b);
        }
    }
    {
        uintptr_t bits = 0;
        const void* ptrs[] = {epi->addend, epi->mul, epi->mul2, epi->gate2, epi->relu_gate, epi->out, epi->out2, epi->scale_out};
        for (const void* q : ptrs) bits |= reinterpret_cast<uintptr_t>(q);
        p.vec_ok = ((bits & 15) == 0 && p.g.out_pitch % 4 == 0) ? 1 : 0;
        if (g.out_cgroup > 0 && !p.vec_ok)
            return bcos_set_error(BCOS_E_INVAL, "bcos_tapconv: out_cgroup needs 16-byte addressable epilogue tensors");
    }
    {   // specialised epilogue (tile_epilogue_fast) when the launch's feature set is one of the compiled kinds
        p.epi_kind = 0;
        p.out_bytes = 0;
        const bcos_epilogue& e = *epi;
        const int64_t obytes = (int64_t)g.N * g.OH * g.OW * p.g.out_pitch * 4;
        const bool norm_l = e.bcos_mode != BCOS_NONE;
        const bool off = bcos_option(BCOS_OPT_EPI_GENERIC) != 0;                // development / test switch
        // (a row_scale rides in the inverse operand scale of the split-f16 loops; the other loops carry it through the general epilogue)
        bool ok = !off && p.vec_ok && g.Cout % 4 == 0 && obytes < ((int64_t)1 << 31) && e.max_out <= 1 && e.out != nullptr && !e.col_scale && !(e.flags & BCOS_EPI_UNIT_NORM_W) &&
                  (!e.row_scale || p.h2) &&
                  !e.gate2 && !e.relu_gate && !(e.flags & (BCOS_EPI_NORM_ONLY | BCOS_EPI_FORCE_POW)) &&
                  ((reinterpret_cast<uintptr_t>(e.bias) | reinterpret_cast<uintptr_t>(e.ch_scale) | reinterpret_cast<uintptr_t>(e.ch_shift)) & 15) == 0;
        int ef = 0;
        if (norm_l) {
            ok = ok && e.b == 2.0f && e.relu >= 0 && e.relu <= 2 && !e.mul && !e.mul2 && !e.out2 && !e.out2_absmax &&
                 !(e.flags & BCOS_EPI_MUL_FROM_ACT);
            ef = (e.addend ? EF_ADDEND : 0) | (e.relu == 1 ? EF_RELU : 0) | (e.relu == 2 ? EF_GELU : 0) | (e.scale_out ? EF_SCALE_OUT : 0);
        } else {
            ok = ok && e.relu == 0 && !e.ch_scale && !e.ch_shift && !e.scale_out && (!e.mul2 || e.out2) &&
                 (!e.out2 || e.mul) && (!e.out2_absmax || e.out2);
            // out2 is either ungated or gated by the low bit of mul (BCOS_EPI_GATE2_FROM_MUL): both are what the kinds compute
            ef = (e.addend ? EF_ADDEND : 0) | (e.mul ? EF_MUL : 0) | (e.out2 ? EF_OUT2 : 0) | (e.mul2 ? EF_MUL2 : 0) |
                 ((e.flags & BCOS_EPI_MUL_FROM_ACT) ? EF_MULACT : 0) | (e.rowadd ? (EF_ROWADD | EF_ADDEND) : 0);
            if (e.rowadd) ok = ok && e.rowadd_scale && e.addend_sub <= 1 && !(reinterpret_cast<uintptr_t>(e.rowadd) & 15) && g.out_cgroup == 0;
            if (e.flags & BCOS_EPI_MUL_FROM_ACT)
                ok = ok && ((reinterpret_cast<uintptr_t>(e.mul_csc) | reinterpret_cast<uintptr_t>(e.mul_csh)) & 15) == 0;
        }
        if (ok) {
            const int* kinds = norm_l ? EPI_KINDS_FWD : EPI_KINDS_BWD;
            for (int k = 0; k < N_EPI_KINDS; ++k)
                if (kinds[k] == ef) { p.epi_kind = k + 1; p.out_bytes = (unsigned)obytes; break; }
        }
    }
    if (epi->rowadd && (p.epi_kind == 0 || (epi->bcos_mode != BCOS_NONE)))
        return bcos_set_error(BCOS_E_NOSUP, "bcos_tapconv: rowadd / rowadd_scale belong to a plain gradient launch that takes a specialised epilogue "
                                            "(16-byte addressable tensors < 2 GiB, Cout % 4 == 0, no mul / out2 / addend_sub): use bcos_patch_norm_bwd_add");
    {   // per-image range of the emitted maxima (bcos_epilogue.out_imgmax / out_imgmin_c): folded into the specialised epilogues only, for
        // launches whose tiles span at most 16 images (256 rows at >= 19 rows per image), plain output mapping, one group
        const bool fuses = p.epi_kind > 0 && epi->out_absmax && g.out_cgroup == 0 && G == 1 && (int64_t)g.P * g.Q >= 19;
        if (t_query_image_range) return fuses ? 1 : 0;
        if (epi->out_imgmax || epi->out_imgmin_c) {
            if (!epi->out_imgmax || !epi->out_imgmin_c || !fuses)
                return bcos_set_error(BCOS_E_NOSUP, "bcos_tapconv: out_imgmax / out_imgmin_c come as a pair, with out_absmax, and only for launches "
                                                    "bcos_tapconv_fuses_image_range() answers 1 for");
        }
    }
    const bool norm = epi->bcos_mode != BCOS_NONE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (g.Cout <= 8 && G == 1 && !epi->out_absmax && !epi->out2_absmax && epi->max_out <= 1 && epi->addend_sub <= 1 && !epi->row_scale && !epi->a_sumsq && !epi->rowadd) {
        const int handled = bcos_try_skinny(a, wt, p.g, p.e, p.M, s);
        if (handled != 0) return handled < 0 ? handled : BCOS_OK;
    }
    if (p.h2) {
        // staging of the split-f16 loop: LDS-DMA (tile_body_d, default) or registers (tile_body_h2: BCOS_H2_LOOP=regs); same bits
        const bool dma = bcos_option(BCOS_OPT_H2_LOOP) == 0;
        {
            // multi-tap launches over an LDS-resident input patch (tile_body_p).  The choice depends on the layer's geometry alone,
            // never on the batch: the patch loop rounds differently from the per-tap loops (one operand scale per image), and an
            // image's bits must not depend on how many images share its launch.  BCOS_OPT_PATCH = 0: development / test switch.
            const bool patch_on = bcos_option(BCOS_OPT_PATCH) != 0;
            if (!patch_on) { p.a_imgmax = p.a_imgmin = nullptr; p.a_imgmin_c = nullptr; }      // (per-row scales everywhere: the per-tap loops as they were)
            const int ntaps = g.TH * g.TW;
            const bool geom_ok = ((ntaps == 9 && g.TH == 3) || (ntaps == 16 && g.TH == 4)) && g.C % X3_BK == 0 && g.in_sh == 1 && g.in_sw == 1 &&
                                 g.dstep_h == 1 && g.dstep_w == 1 && p.g.a_pitch >= g.C;
            if (dma && geom_ok && p.a_imgmax && patch_on && !epi->row_scale && !epi->a_sumsq) {      // (the patch kernels compile those two out)
                if (ntaps == 16) {
                    if (g.Cout <= 32 && g.in_sh == 1) return bcos_tc_p2_256x32_t16(&p, norm, s);
                } else {
                if (g.Cout > 128 && g.Cout <= 256 && patch_fits(g, 128, 256, 256) && bcos_option(BCOS_OPT_PATCH_WIDE)) return bcos_tc_p_128x256_a(&p, norm, s);
                if (g.Cout > 64 && patch_fits(g, 128, 256, 256)) return bcos_tc_p_128x128_a(&p, norm, s);
                if (g.Cout > 64 && patch_fits(g, 128, 256, 448)) return bcos_tc_p_128x128_b(&p, norm, s);
                if (g.Cout > 64 && patch_fits(g, 128, 384, 320)) return bcos_tc_p_128x128_c(&p, norm, s);
                if (g.Cout > 32 && g.Cout <= 64 && patch_fits(g, 256, 640, 640)) return bcos_tc_p_256x64_a(&p, norm, s);
                // wider images (112^2: the 3 x 3 stem convolutions of the CLIP ResNets): 8 x 32 blocks.  (At 56^2 the linear tiles win,
                // 224 against 253 us: the blocks leave an eighth of the rows empty there.)
                if (g.Cout > 32 && g.Cout <= 64 && g.in_sh == 1 && g.Q > 64) return bcos_tc_p2_256x64_b32(&p, norm, s);
                if (g.Cout > 8 && g.Cout <= 32 && g.in_sh == 1 && g.Q > 64) return bcos_tc_p2_256x32_t9(&p, norm, s);
                }
            }
        }
        if (g.Cout > 64 && dma && bcos_option(BCOS_OPT_H2_TILE) == 0) {
            // FEW-ROW launches (M <= a few hundred rows: the pooled token of CLIP's attention pool at batch 256 -- 2 row tiles --, small
            // batches through every head): with 128- or 256-column tiles they occupy a handful of CUs and run at the latency of ONE
            // workgroup's K loop (256 x 2048 -> 1024: 8 workgroups, 111 us against 28 for the vendor's GEMM).  Narrower
            // column tiles multiply the workgroups -- 64 or 32 columns each, as many as it takes to put a workgroup on half the CUs -- at
            // no cost in bits (an output element's K order does not depend on its tile).
            const int64_t tm = (M64 + 127) / 128;
            const int64_t t128 = tm * ((g.Cout + 127) / 128);
            if (t128 < 64) {
                if (tm * ((g.Cout + 63) / 64) >= 96 || g.Cout <= 128) return bcos_tc_d_128x64(&p, norm, s);
                return bcos_tc_d_128x32(&p, norm, s);
            }
        }
        if (g.Cout > 64) {
            // 128 x 256 tiles stage half the A bytes per MFMA; they pay when they do not cost an extra round of tiles
            // (measured on the ResNet-50 shapes: M = 50176, N = 256: -3 %; M = 12544, N = 512: +17 % -> stays 128 x 128)
            const int64_t force = bcos_option(BCOS_OPT_H2_TILE);      // development switch: 1 = 128 x 128, 2 = 128 x 256
            const int64_t tm = (M64 + 127) / 128;
            const int64_t t1 = tm * ((g.Cout + 127) / 128), t2 = tm * ((g.Cout + 255) / 256);
            // (a finer cost model -- a wide tile = 1.75 narrow ones, which moves M = 200 704 with N = 256 / 512 and M = 50 176 with
            // N = 1024, K = 512 to wide tiles -- wins 11-18 % on those launches timed one by one and loses 0.1 ms on the step, where
            // consecutive launches overlap their tails: not adopted)
            // rounds of tiles, a wide tile priced at BCOS_OPT_H2_WIDE_COST / 4 narrow ones (8 = two, the model of rounds 2-3; the wide
            // tile shares the A operand's DMA and split between its halves, so 7 is nearer what the layers measure)
            const int64_t wc = bcos_option(BCOS_OPT_H2_WIDE_COST);
            const int64_t c1 = 4 * ((t1 + SLOTS - 1) / SLOTS), c2 = wc * ((t2 + SLOTS - 1) / SLOTS);
            bool wide = g.Cout > 128 && (c2 < c1 || (c2 == c1 && (wc < 8 || p.Ktot >= 1024)));
            if (force) wide = g.Cout > 128 && force == 2;
            // 129 ... 192 columns (the 192-wide linears of the SimpleViTs: to_out, linear2, their gradients): ONE tile of 192 columns
            // (six accumulator tiles per wave) instead of 128 + a half-empty second 128
            if (dma && g.Cout > 128 && g.Cout <= 192 && !force) return bcos_tc_d_128x192(&p, norm, s);
            // multiples of 192 that are not multiples of 256 (the 576-wide to_qkv of the SimpleViTs: 3 x 192 exactly, against 5 x 128 with
            // 64 empty columns or 3 x 256 with 192): whole 192-column tiles -- ViT-Ti batch 512 forward+explanation 18.26 -> 17.97 ms per step
            // in three same-node pairs (round 5); 768 = 4 x 192 against 3 x 256: no gain, stays on the 256-column tiles.  Bits unchanged.
            if (dma && g.Cout % 192 == 0 && g.Cout % 256 != 0 && g.Cout <= 1152 && !force) return bcos_tc_d_128x192(&p, norm, s);
            if (wide) return dma ? bcos_tc_d_128x256(&p, norm, s) : bcos_tc_h2_128x256(&p, norm, s);
            return dma ? bcos_tc_d_128x128(&p, norm, s) : bcos_tc_h2_128x128(&p, norm, s);
        }
        if (g.Cout > 32) {
            // 256 x 64 tiles (four waves of 64 x 64: the A operand's split, its LDS image and the per-tile prologue / epilogue
            // latencies are shared by twice the matrix instructions of a 128 x 64 tile) from two rounds of tiles on.  Same-node
            // A/B on the ResNet-50 step at batch 256 (M = 802 816): 3x3 64 -> 64 gradient 1.03 -> 0.90 ms, 256 -> 64 gradient
            // 1.21 -> 1.13 ms, stem forward 1.05 -> 1.01 ms, the forward launches -1 %.  Results are identical bit for bit (same
            // K walk, same product order per accumulator).
            const bool tall = bcos_option(BCOS_OPT_H2_TALL) != 0;       // development switch: 0 keeps the 128-row tiles
            const int64_t tall_min = bcos_option(BCOS_OPT_H2_TALL_MIN);     // default 2 * 256 * SLOTS (batch 128, M = 401 408: +0.4 % per step; M = 200 704: neutral)
            if (M64 >= tall_min && tall) return dma ? bcos_tc_d_256x64(&p, norm, s) : bcos_tc_h2_256x64(&p, norm, s);
            return dma ? bcos_tc_d_128x64(&p, norm, s) : bcos_tc_h2_128x64(&p, norm, s);
        }
        {   // 256 x 32 tiles likewise (the depth-to-space stem gradient, M = 3.2 M: 1.29-1.38 -> 1.17-1.20 ms in a same-node A/B)
            if (M64 >= 2 * 256 * SLOTS && bcos_option(BCOS_OPT_H2_TALL)) return dma ? bcos_tc_d_256x32(&p, norm, s) : bcos_tc_h2_256x32(&p, norm, s);
        }
        return dma ? bcos_tc_d_128x32(&p, norm, s) : bcos_tc_h2_128x32(&p, norm, s);
    }
    if (g.Cout > 64) {
        // few-row launches (see the split-f16 branch above): narrower column tiles put a workgroup on more CUs; same bits
        const int64_t tm = (M64 + 127) / 128;
        if (G == 1 && tm * ((g.Cout + 127) / 128) < 64 && bcos_option(BCOS_OPT_H2_TILE) == 0)
            return (tm * ((g.Cout + 63) / 64) >= 96 || g.Cout <= 128) ? bcos_tc_cfg_128x64(&p, norm, s) : bcos_tc_cfg_128x32(&p, norm, s);
        return bcos_tc_cfg_128x128(&p, norm, s);
    }
    if (g.Cout > 32) return bcos_tc_cfg_128x64(&p, norm, s);
    return bcos_tc_cfg_128x32(&p, norm, s);
}


extern "C" int bcos_tapconv_group(const float* a, const float* const* wts, const bcos_tapconv_geom* geoms,
                                  const bcos_epilogue* epis, int count, void* stream) {
    if (!a || !wts || !geoms || !epis || count <= 0) return bcos_set_error(BCOS_E_INVAL, "bcos_tapconv_group: bad argument");
    const int fused = bcos_try_skinny_group(a, wts, geoms, epis, count, reinterpret_cast<hipStream_t>(stream));
    if (fused < 0) return fused;
    if (fused == 1) return BCOS_OK;
    for (int i = 0; i < count; ++i) {
        const int rc = bcos_tapconv(a, wts[i], &geoms[i], &epis[i], stream);
        if (rc != BCOS_OK) return rc;
    }
    return BCOS_OK;
}
#endif  // BCOS_TC_IN(0)
