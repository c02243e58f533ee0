// Where does the hardware place the workgroups of a launch that is smaller than one round of resident slots?
// (development aid)  Each workgroup records (XCC, SE, CU) and its start / end time; the host prints how many CUs received
// 0 / 1 / 2 workgroups.   hipcc --offload-arch=gfx950 -O2 dispatch_probe.hip -o dispatch_probe.bin ; ./dispatch_probe.bin 392 70000
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <tuple>
#include <vector>

__global__ __launch_bounds__(256, 2) void probe(unsigned* ids, unsigned long long* t0, unsigned long long* t1, int spin) {
    extern __shared__ float lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long a = wall_clock64();
    float v = threadIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    lds[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        ids[blockIdx.x * 2] = hw;
        ids[blockIdx.x * 2 + 1] = xcc;
        t0[blockIdx.x] = a;
        t1[blockIdx.x] = wall_clock64() + (lds[1] == 12345.f);
    }
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 392;
    const int lds = argc > 2 ? atoi(argv[2]) : 70000;
    const int spin = argc > 3 ? atoi(argv[3]) : 200000;
    unsigned* ids; unsigned long long *t0, *t1;
    hipMalloc(&ids, grid * 8); hipMalloc(&t0, grid * 8); hipMalloc(&t1, grid * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(grid), dim3(256), lds, 0, ids, t0, t1, spin);
    hipDeviceSynchronize();
    std::vector<unsigned> h(grid * 2); std::vector<unsigned long long> a(grid), b(grid);
    hipMemcpy(h.data(), ids, grid * 8, hipMemcpyDeviceToHost);
    hipMemcpy(a.data(), t0, grid * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), t1, grid * 8, hipMemcpyDeviceToHost);
    std::map<std::tuple<unsigned, unsigned, unsigned, unsigned>, int> per_cu;
    unsigned long long first = ~0ull, last = 0, late = 0;
    for (int i = 0; i < grid; ++i) if (a[i] < first) first = a[i];
    for (int i = 0; i < grid; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_cu[{xcc, se, sh, cu}]++;
        if (b[i] > last) last = b[i];
        if (a[i] - first > (b[i] - a[i]) / 4) ++late;       // started well after the first: a second round
    }
    std::map<int, int> hist;
    for (auto& kv : per_cu) hist[kv.second]++;
    printf("grid %d, lds %d: %zu distinct CUs used;", grid, lds, per_cu.size());
    for (auto& kv : hist) printf("  %d CUs x %d WGs", kv.second, kv.first);
    printf(";  %llu WGs started late;  span %.1f us (100 MHz clock)\n", late, (last - first) / 100.0);
    for (int i = 0; i < 20 && i < grid; ++i)
        printf("%s wg %d: xcc %u se %u cu %u", i ? "," : "first:", i, h[2 * i + 1] & 0xf, (h[2 * i] >> 13) & 7, (h[2 * i] >> 8) & 0xf);
    printf("\n");
    return 0;
}
