// Probe: how many independent VALU instructions can run in the shadow of one v_mfma_f32_32x32x16_bf16?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            if (u == 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            if (u == 2) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
            if (u == 3) acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc3, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV; ++q) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[q & 7]) : "v"(1.0f));
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV>
void run(int blocks_per_cu, float* d) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NV><<<256 * blocks_per_cu, 256>>>(d, 100);
    hipEventRecord(e0);
    k<NV><<<256 * blocks_per_cu, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: blocks_per_cu waves, each iters*4 MFMAs
    double mfma_per_simd = (double)blocks_per_cu * iters * 4;
    printf("NV=%2d waves/SIMD=%d: %.3f ms, %.1f ns per MFMA-slot (per SIMD), = %.1f cycles @2.4GHz\n", NV, blocks_per_cu, ms,
           ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4);
}

int main() {
    float* d; hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
    for (int w = 1; w <= 2; ++w) {
        run<0>(w, d); run<2>(w, d); run<4>(w, d); run<6>(w, d); run<8>(w, d); run<12>(w, d); run<16>(w, d);
    }
    return 0;
}
