// valu_ubench.hip -- issue-rate / latency micro-benchmarks that the kernel design rests on (MI355X, gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o build/valu_ubench tools/valu_ubench.hip ; run on the GPU box.
// Every number is cycles (s_memtime ticks) per wave-instruction as seen by ONE wave, for w waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

enum { T_FMA_IND = 0, T_FMA_DEP, T_PKFMA_IND, T_FMA_SGPR, T_MUL_ADD, T_MOV, T_LDS_DEP, T_LDS_IND, T_SLOAD_DEP, T_GLOAD_DEP, T_RCP, T_SIN, N_TESTS };
static const char *kNames[N_TESTS] = {"v_fma_f32 x8 independent", "v_fma_f32 dependent chain", "v_pk_fma_f32 x8 independent",
                                      "v_fma_f32 sgpr operand x8", "v_mul+v_add x8 independent", "v_mov_b32 x8",
                                      "ds_read_b32 dependent (latency)", "ds_read_b32 x8 independent", "s_load_dword dependent (latency)",
                                      "global_load_dword dependent L2-hit (latency)", "v_rcp_f32 x8 independent", "v_sin_f32 x8 independent"};

__global__ __launch_bounds__(64) void bench(int test, int iters, float *out, unsigned long long *cyc, const int *chase, float s0)
{
    __shared__ int lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (i * 4 + 64 * 4) & 4095;  // byte address of the next element
    __syncthreads();
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
    f2 pb = {b, b}, pc = {c, c};
    int idx = threadIdx.x * 4;
    const int *gp = chase + threadIdx.x;
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int n_inst = 64;
    for (int it = 0; it < iters; it++) {
        switch (test) {
        case T_FMA_IND:
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
            break;
        case T_FMA_DEP:
            REP64(asm volatile("v_fma_f32 %0, %0, %1, %2\n" : "+v"(a0) : "v"(b), "v"(c));)
            break;
        case T_PKFMA_IND:
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                              "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));)
            break;
        case T_FMA_SGPR:
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s0), "v"(c));)
            break;
        case T_MUL_ADD:
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
                              "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
            break;
        case T_MOV:
            REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                              "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
            break;
        case T_LDS_DEP:
            REP64(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n" : "+v"(idx) :: "memory");)
            break;
        case T_LDS_IND: {
            int r0, r1, r2, r3, r4, r5, r6, r7;
            REP8(asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                              "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n"
                              "s_waitcnt lgkmcnt(0)\n"
                              : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(idx) : "memory");
                 a0 += r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;)
            break;
        }
        case T_SLOAD_DEP: {
            int sidx = 0;
            const int *base = chase;
            REP64(asm volatile("s_load_dword %0, %1, %0\n s_waitcnt lgkmcnt(0)\n" : "+s"(sidx) : "s"(base) : "memory");)
            a0 += sidx;
            break;
        }
        case T_GLOAD_DEP: {
            int off = threadIdx.x * 4;
            REP64(asm volatile("global_load_dword %0, %0, %1\n s_waitcnt vmcnt(0)\n" : "+v"(off) : "s"(chase) : "memory");)
            a0 += off;
            break;
        }
        case T_RCP:
            REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                              "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
            break;
        case T_SIN:
            REP8(asm volatile("v_sin_f32 %0, %0\n v_sin_f32 %1, %1\n v_sin_f32 %2, %2\n v_sin_f32 %3, %3\n"
                              "v_sin_f32 %4, %4\n v_sin_f32 %5, %5\n v_sin_f32 %6, %6\n v_sin_f32 %7, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
            break;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    (void)n_inst;
    (void)gp;
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.x + p6.x + p7.x + idx;
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; }
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, n_cu, prop.clockRate);
    const int max_blocks = n_cu * 32;
    float *out; unsigned long long *cyc; int *chase;
    hipMalloc(&out, max_blocks * 64 * sizeof(float));
    hipMalloc(&cyc, 2 * max_blocks * sizeof(unsigned long long));
    std::vector<int> h(4096);
    for (int i = 0; i < 4096; i++) h[i] = ((i + 64) % 1024) * 4;  // byte offset of the next element (same lane)
    hipMalloc(&chase, 4096 * sizeof(int));
    hipMemcpy(chase, h.data(), 4096 * sizeof(int), hipMemcpyHostToDevice);
    const int iters = 20000;
    std::vector<unsigned long long> hc(2 * max_blocks);
    for (int test = 0; test < N_TESTS; test++) {
        printf("%-48s", kNames[test]);
        for (int w : {1, 2, 4, 8}) {
            const int blocks = n_cu * 4 * w;
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            bench<<<blocks, 64>>>(test, iters / 4, out, cyc, chase, 1.0001f);
            hipEventRecord(e0);
            bench<<<blocks, 64>>>(test, iters, out, cyc, chase, 1.0001f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(hc.data(), cyc, 2 * blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            double sum = 0, rsum = 0; for (int i = 0; i < blocks; i++) { sum += hc[2 * i]; rsum += hc[2 * i + 1]; }
            const double per = sum / blocks / (iters * 64.0);
            // SIMD throughput view: wall cycles at the nominal clock per instruction per SIMD
            const double wall_cyc = ms * 1e-3 * prop.clockRate * 1e3 / (iters * 64.0 * w);
            printf("  w=%d: %6.2f tick/inst/wave, tick %4.0f MHz, %5.2f ns/inst/SIMD |", w, per, sum / rsum * 100.0, ms * 1e6 / (iters * 64.0 * w));
            (void)wall_cyc;
        }
        printf("\n");
    }
    return 0;
}
