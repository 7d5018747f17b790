// Issue rate of v_fma_f32, v_pk_fma_f32 and v_fma_f64 on one SIMD at one, two and four wavefronts, with 1-8 independent accumulators
// per wavefront (dependent-issue latency against throughput; is two states per lane in packed registers worth it?).
// hipcc --offload-arch=gfx950 -O3 tools/pk_rate.hip -o build/tools/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <class T, int NACC>
__global__ void __launch_bounds__(64) k(T *out, float a, float b, int iters)
{
    T acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = T(threadIdx.x + i);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = acc[i] * a + b;
    }
    T s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; i++) s += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <class T, int NACC>
void run(const char *name, int waves_per_simd)
{
    T *out;
    const int n_cu = 256;
    const int blocks = n_cu * 4 * waves_per_simd;
    (void)hipMalloc(&out, blocks * 64 * sizeof(T));
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<T, NACC><<<blocks, 64>>>(out, 1.0001f, 0.5f, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<T, NACC><<<blocks, 64>>>(out, 1.0001f, 0.5f, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr = double(iters) * 8 * NACC * waves_per_simd;  // per SIMD
    std::printf("%-10s acc %2d waves/SIMD %d: %.3f ms, %.2f cycles per instruction and SIMD at 2.4 GHz\n", name, NACC, waves_per_simd, ms,
                ms * 1e-3 * 2.4e9 / instr);
    (void)hipFree(out);
}
int main()
{
    for (int w = 1; w <= 4; w *= 2) {
        run<float, 1>("fma", w); run<float, 2>("fma", w); run<float, 4>("fma", w); run<float, 8>("fma", w);
        run<f2, 1>("pk_fma", w); run<f2, 2>("pk_fma", w); run<f2, 4>("pk_fma", w); run<f2, 8>("pk_fma", w);
        run<double, 1>("fma_f64", w); run<double, 2>("fma_f64", w); run<double, 4>("fma_f64", w); run<double, 8>("fma_f64", w);
    }
    return 0;
}
