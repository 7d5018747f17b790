// Per-launch cost of back-to-back launches on one stream (hipEvents around N launches), for kernels that do nothing: the floor under every
// small-batch time of tools/lm_waves_ab.py.   build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/launch_gap.hip -o /tmp/launch_gap && /tmp/launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void nothing(int *p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void touch(const float *in, float *out, int n)
{   // one 256-byte row in, one out per wavefront: a load-to-store round trip through HBM / L2
    extern __shared__ float lds[];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + 1.0f;
}
int main()
{
    hipStream_t s;
    (void)hipStreamCreate(&s);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    float *in, *out;
    (void)hipMalloc(&in, 1 << 24);
    (void)hipMalloc(&out, 1 << 24);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&touch), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int N = 200;
    for (int grid : {1, 256, 512, 2048})
        for (int block : {64, 256})
            for (int lds : {0, 81920}) {
                float ms_nothing = 0, ms_touch = 0;
                for (int rep = 0; rep < 3; rep++) {
                    (void)hipEventRecord(a, s);
                    for (int i = 0; i < N; i++) hipLaunchKernelGGL(nothing, dim3(grid), dim3(block), lds, s, nullptr);
                    (void)hipEventRecord(b, s);
                    (void)hipEventSynchronize(b);
                    (void)hipEventElapsedTime(&ms_nothing, a, b);
                    (void)hipEventRecord(a, s);
                    for (int i = 0; i < N; i++) hipLaunchKernelGGL(touch, dim3(grid), dim3(block), lds, s, in, out, grid * block);
                    (void)hipEventRecord(b, s);
                    (void)hipEventSynchronize(b);
                    (void)hipEventElapsedTime(&ms_touch, a, b);
                }
                std::printf("grid %5d block %3d lds %6d: empty %.2f us per launch, load+store %.2f us\n", grid, block, lds, 1e3 * ms_nothing / N, 1e3 * ms_touch / N);
            }
    return 0;
}
