// Issue rate of v_fma_f32, v_pk_fma_f32 and v_fma_f64 on one SIMD at one, two and four wavefronts, with 1-8 independent accumulators
// per wavefront (dependent-issue latency against throughput; is two states per lane in packed registers worth it?).
// hipcc --offload-arch=gfx950 -O3 tools/pk_rate.hip -o build/tools/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
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
// matrix cores: v_mfma_f32_16x16x4_f32 and v_mfma_f64_16x16x4_f64, NACC independent accumulator tiles (1 024 FMAs per instruction = 16 wavefront FMAs)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d4 __attribute__((ext_vector_type(4)));
template <class T, int NACC>
__global__ void __launch_bounds__(64) km(T *out, float a, float b, int iters)
{
    using V = typename std::conditional<sizeof(T) == 4, f4, d4>::type;
    V acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = V{T(threadIdx.x + i), 0, 0, 0};
    const T av = T(a) * T(1e-3) * T(threadIdx.x & 3), bv = T(b);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) {
                if constexpr (sizeof(T) == 4) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
            }
    }
    T s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <class T, int NACC>
void run_m(const char *name, int waves_per_simd)
{
    T *out;
    const int blocks = 256 * 4 * waves_per_simd;
    (void)hipMalloc(&out, blocks * 64 * sizeof(T));
    const int iters = 5000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    km<T, NACC><<<blocks, 64>>>(out, 1.0001f, 0.5f, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    km<T, NACC><<<blocks, 64>>>(out, 1.0001f, 0.5f, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr = double(iters) * 8 * NACC * waves_per_simd;  // per SIMD
    const double cyc = ms * 1e-3 * 2.4e9 / instr;
    std::printf("%-10s acc %2d waves/SIMD %d: %.3f ms, %.2f cycles per instruction and SIMD at 2.4 GHz = %.1f FMA lanes per cycle, %.1f TFLOP/s chip\n", name,
                NACC, waves_per_simd, ms, cyc, 1024.0 / cyc, 2.0 * 1024.0 * instr * 1024 / (ms * 1e-3) * 1e-12);
    (void)hipFree(out);
}
int main()
{
    for (int w = 1; w <= 2; w *= 2) {
        run_m<float, 1>("mfma_f32", w); run_m<float, 2>("mfma_f32", w); run_m<float, 4>("mfma_f32", w);
        run_m<double, 1>("mfma_f64", w); run_m<double, 2>("mfma_f64", w); run_m<double, 4>("mfma_f64", w);
    }
    for (int w = 1; w <= 4; w *= 2) {
        run<float, 1>("fma", w); run<float, 2>("fma", w); run<float, 4>("fma", w); run<float, 8>("fma", w);
        run<f2, 1>("pk_fma", w); run<f2, 2>("pk_fma", w); run<f2, 4>("pk_fma", w); run<f2, 8>("pk_fma", w);
        run<double, 1>("fma_f64", w); run<double, 2>("fma_f64", w); run<double, 4>("fma_f64", w); run<double, 8>("fma_f64", w);
    }
    return 0;
}
