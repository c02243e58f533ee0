// What does the ACCESS PATTERN of the A operand cost on its own?  The LDS-DMA loop of the 1 x 1 launches (csrc/bcos_tapconv.hip: tile_body_d)
// reads a [M][K] fp32 matrix tile by tile (256 or 128 consecutive rows = one contiguous block of memory), 64 bytes of every row per 16-k
// step, two steps in flight.  This probe reads the same matrix with nothing else going on -- plain 16-byte loads, no LDS, no arithmetic
// beyond one add per value -- in that order (piece = 64 B of a row per step), with 128 / 256 / 1024-byte pieces per row and step, and
// fully contiguously, for K = 64 ... 512 at constant size (822 MB).   hipcc --offload-arch=gfx950 -O3 a_pattern_probe.hip -o a_pattern_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// one workgroup = ROWS consecutive rows; a "step" covers PIECE bytes of every row; INFL steps in flight
template <int ROWS, int PIECE, int INFL>
__global__ __launch_bounds__(256) void tile_pattern(const f32x4* __restrict__ a, float* __restrict__ sink, int K4, int ntiles) {
    constexpr int LPR = PIECE / 16;            // lanes per row piece
    constexpr int RPP = 256 / LPR;             // rows per pass of the workgroup
    constexpr int PASSES = ROWS / RPP;
    static_assert(PASSES >= 1, "rows");
    const int lane_c = threadIdx.x % LPR, lane_r = threadIdx.x / LPR;
    const int steps = (K4 * 16) / PIECE;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const f32x4* base = a + (size_t)tile * ROWS * K4;
        for (int s0 = 0; s0 < steps; s0 += INFL) {
            f32x4 v[INFL][PASSES];
#pragma unroll
            for (int u = 0; u < INFL; ++u)
#pragma unroll
                for (int p = 0; p < PASSES; ++p)
                    if (s0 + u < steps) v[u][p] = base[(size_t)(lane_r + p * RPP) * K4 + (s0 + u) * LPR + lane_c];
#pragma unroll
            for (int u = 0; u < INFL; ++u)
#pragma unroll
                for (int p = 0; p < PASSES; ++p)
                    if (s0 + u < steps) acc += v[u][p];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

__global__ __launch_bounds__(256) void contiguous(const f32x4* __restrict__ a, float* __restrict__ sink, size_t n4) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * 256 * 8;
    for (size_t i = (size_t)blockIdx.x * 256 * 8 + threadIdx.x; i < n4; i += stride) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (i + u * 256 < n4) v[u] = a[i + u * 256];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (i + u * 256 < n4) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

template <typename F> static float time_ms(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

int main() {
    const size_t n = (size_t)802816 * 256;            // floats
    float *a, *sink;
    hipMalloc(&a, n * 4); hipMalloc(&sink, 64);
    hipMemset(a, 0x3c, n * 4);
    const f32x4* a4 = (const f32x4*)a;
    printf("# read-only streaming of %.0f MB, GB/s\n", n * 4 / 1e6);
    printf("contiguous (8 x 16 B in flight per thread, 2048 workgroups)   %7.0f\n", n * 4 / time_ms([&] { hipLaunchKernelGGL(contiguous, dim3(2048), dim3(256), 0, 0, a4, sink, n / 4); }) / 1e6);
    for (int K : {64, 128, 256, 512}) {
        const int K4 = K / 4;
        const int nt256 = (int)(n / K / 256), nt128 = (int)(n / K / 128);
#define RUN(ROWS, PIECE, INFL, WGS) \
        if (PIECE <= K * 4) printf("K = %3d  rows/tile %3d  piece %4d B  %d steps in flight  %4d workgroups   %7.0f\n", K, ROWS, PIECE, INFL, WGS, \
               n * 4 / time_ms([&] { hipLaunchKernelGGL((tile_pattern<ROWS, PIECE, INFL>), dim3(WGS), dim3(256), 0, 0, a4, sink, K4, (ROWS == 256 ? nt256 : nt128)); }) / 1e6);
        RUN(256, 64, 2, 768) RUN(256, 64, 2, 2048) RUN(256, 64, 4, 768) RUN(256, 128, 2, 768) RUN(256, 128, 1, 768) RUN(256, 256, 1, 768) RUN(256, 256, 2, 768)
        RUN(256, 1024, 1, 768) RUN(128, 64, 2, 512) RUN(128, 64, 2, 1024) RUN(128, 128, 2, 512) RUN(128, 256, 1, 512)
    }
    return 0;
}
