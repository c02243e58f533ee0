// Does `buffer_load_dwordx4 ... lds` (LDS-DMA) write ZEROS to LDS for lanes whose offset fails the buffer bounds check?
// (the tap convolution relies on the bounds check to zero-fill out-of-image taps).  hipcc --offload-arch=gfx950 -O3 -o lds_dma_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* a, float* out, unsigned bytes) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a), 0, bytes, 0x00020000);
    unsigned voff = threadIdx.x * 16;
    if (threadIdx.x & 1) voff = 0x80000000u;          // out of range
    if ((threadIdx.x & 7) == 2) voff = bytes - 8;     // straddles the end: partial
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) ((float*)lds)[i] = -7.f;
    __syncthreads();
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + (threadIdx.x >> 6) * 1024), 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = ((float*)lds)[i];
}
int main() {
    const int n = 4096;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = 1.f + i;
    float *a, *o;
    hipMalloc(&a, n * 4); hipMalloc(&o, 1024 * 4);
    hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice);
    k<<<1, 256, 16384>>>(a, o, n * 4);
    std::vector<float> r(1024);
    hipMemcpy(r.data(), o, 1024 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) {
        for (int q = 0; q < 4; ++q) {
            float got = r[t * 4 + q], want;
            if ((t & 7) == 2) want = q < 2 ? 1.f + (n - 2 + q) : 0.f;
            else if (t & 1) want = 0.f;
            else want = 1.f + t * 4 + q;
            if (got != want) { if (bad < 12) printf("lane %d q %d got %g want %g\n", t, q, got, want); ++bad; }
        }
    }
    printf("lds_dma_probe: %d mismatches (0 = out-of-range lanes write zeros)\n", bad);
    return bad != 0;
}
