// tools/lds_granule.hip -- how many one-wavefront workgroups with N bytes of dynamic LDS does a CU of this GPU hold?
// build: hipcc --offload-arch=gfx950 -O2 tools/lds_granule.hip -o build/exp/lds_granule
// (a) what the runtime's occupancy query says, (b) measured: workgroups that spin a fixed time, grid = CUs x k, time vs k.
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ char smem[];
__global__ __launch_bounds__(64) void spin(long long ticks, int *sink)
{
    const long long t0 = __builtin_amdgcn_s_memtime();
    smem[threadIdx.x] = 1;
    while (__builtin_amdgcn_s_memtime() - t0 < ticks) {}
    if (smem[threadIdx.x] == 7) *sink = 1;
}
int main()
{
    hipDeviceProp_t pr;
    (void)hipGetDeviceProperties(&pr, 0);
    const int n_cu = pr.multiProcessorCount;
    int *sink;
    (void)hipMalloc(&sink, 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&spin), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const size_t sizes[] = {12032, 12800, 13056, 13312, 14080, 14336, 16384, 18000, 19712, 20480, 20736, 22528, 27136, 32768, 40960};
    for (size_t lds : sizes) {
        int occ = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin, 64, lds);
        printf("lds %6zu: query %2d  naive %2zu  granule1280 %2zu | measured ms at k per CU:", lds, occ, (size_t)163840 / lds,
               (size_t)163840 / ((lds + 1279) / 1280 * 1280));
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        const int k0 = (int)(163840 / lds);
        for (int k = k0 - 2; k <= k0 + 1; k++) {
            if (k < 1) continue;
            hipLaunchKernelGGL(spin, dim3(n_cu * k), dim3(64), lds, 0, 100000LL, sink);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(spin, dim3(n_cu * k), dim3(64), lds, 0, 100000LL, sink);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            float ms;
            (void)hipEventElapsedTime(&ms, a, b);
            printf("  k=%d %.3f", k, ms);
        }
        printf("\n");
    }
    return 0;
}
