// Does handing a CU 6 instead of 8 wave blocks shorten a matrix-bound launch?  (VERDICT r03 "Next round" 1(a): a balanced
// schedule of the 1568 32-row wave blocks of M = 50 176 -- every CU 6 or 7 blocks instead of 4 or 8.)
// A workgroup is 4 waves = one wave per SIMD; a CU holds two such workgroups (LDS- and register-limited, as the contraction
// kernels are).  A wave block = one wave's chain of v_mfma_f32_32x32x16_f16 (eight independent accumulators, `iters` steps);
// a wave with no block exits at once.  Variants of ONE launch of 512 workgroups (two per CU):
//   full     : 4 + 4 blocks per CU                                   (8 blocks: what 136 of the 256 CUs carry today)
//   idle3    : waves 0-2 active in every workgroup (3 + 3)            (6 blocks per CU)
//   idle3/0  : wave 3 idle in even workgroups, wave 0 in odd ones     (6 blocks per CU, spread over the SIMDs)
//   half     : waves 0-1 active (2 + 2)                               (4 blocks: what a CU with ONE workgroup carries)
//   lone     : 256 workgroups of 4 active waves (one per CU)
// hipcc --offload-arch=gfx950 -O2 simd_balance_probe.hip -o simd_balance_probe.bin ; ./simd_balance_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256, 2) void probe(float* out, int iters, int variant) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6;
    bool active = true;
    if (variant == 1) active = wave < 3;
    if (variant == 2) active = (blockIdx.x & 1) ? wave > 0 : wave < 3;
    if (variant == 3) active = wave < 2;
    if (!active) return;
    f16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(threadIdx.x * 0.001f + q); b[q] = (_Float16)(0.5f + q * 0.01f); }
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.f) out[threadIdx.x] = s + lds[0];
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    const int lds = 70000, iters = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const char* names[] = {"full (4+4 blocks per CU)", "idle3 (3+3, same SIMD idle)", "idle3/0 (3+3, idle wave alternates)", "half (2+2)", "lone (one workgroup per CU)"};
    for (int v = 0; v < 5; ++v) {
        const int grid = v == 4 ? 256 : 512, variant = v == 4 ? 0 : v;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(probe, dim3(grid), dim3(256), lds, 0, out, iters, variant);
        hipDeviceSynchronize();
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(grid), dim3(256), lds, 0, out, iters, variant);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-40s %8.1f us\n", names[v], best * 1e3f);
    }
    return 0;
}
