// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on byte counts that are known, in the access patterns of the grbda
// kernels (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ...
// other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
// Each kernel moves N bytes once (N far beyond the 256 MiB Infinity Cache):
//   copy_row4      4 B per lane, one 256-byte row per wave instruction, read + write   (slab rows of the chain kernels)
//   copy_row16     16 B per lane, read + write                                          (the guide's reference pattern)
//   read_ldsdma4   global -> LDS copies of 4 B per lane, no write                       (tile prologue, stage_issue)
// build: hipcc --offload-arch=gfx950 -O3 tools/traffic_calib.hip -o build/tools/traffic_calib
// run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- build/tools/traffic_calib   (and once more with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

__global__ void copy_row4(const float *__restrict__ in, float *__restrict__ out, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void copy_row16(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void read_ldsdma4(const unsigned *__restrict__ in, unsigned *__restrict__ sink, size_t n)
{
    // 64 threads per block: a wave copies 256-byte rows straight into LDS; one value per wave leaves so that nothing is elided
    for (size_t base = blockIdx.x * (size_t)64; base < n; base += (size_t)gridDim.x * 64)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(in + base + threadIdx.x),
                                         (__attribute__((address_space(3))) void *)smem, 4, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = reinterpret_cast<unsigned *>(smem)[0];
}

int main()
{
    const size_t bytes = 1ull << 30;  // 1 GiB per array
    float *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
    hipMemset(a, 1, bytes);
    hipMemset(b, 0, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(copy_row4, dim3(256 * 8), dim3(64), 0, 0, a, b, bytes / 4);
        hipLaunchKernelGGL(copy_row16, dim3(256 * 8), dim3(64), 0, 0, (const float4 *)a, (float4 *)b, bytes / 16);
        hipLaunchKernelGGL(read_ldsdma4, dim3(256 * 8), dim3(64), 1024, 0, (const unsigned *)a, (unsigned *)b, bytes / 4);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    std::printf("bytes per kernel: read %zu, written %zu (read_ldsdma4 writes ~8 KiB)\n", bytes, bytes);
    return 0;
}
