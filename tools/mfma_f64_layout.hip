// Which element of D = A B does lane l hold in accumulator register r of v_mfma_f64_16x16x4_f64?  (gfx950; minv_kernels.hip, Mfma<double>::row)
// build: hipcc --offload-arch=gfx950 -O2 tools/mfma_f64_layout.hip -o build/tools/mfma_f64_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(double *out)
{
    const int l = threadIdx.x;
    // A[i][k] = i + 1 when k == 0 else 0 ; B[k][j] = 100 (j + 1) when k == 0: D[i][j] = (i + 1) * 100 (j + 1)
    const int i = l & 15, k = l >> 4;
    const double a = k == 0 ? double(i + 1) : 0.0, b = k == 0 ? 100.0 * double((l & 15) + 1) : 0.0;
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[l * 4 + r] = c[r];
}
int main()
{
    double *d, h[256];
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 1;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    int ok_a = 1, ok_b = 1;
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < 4; r++) {
            const int v = (int)(h[l * 4 + r] + 0.5), row = v % 100 - 1 + (v % 100 == 0 ? 100 : 0), col = v / 100 - 1;
            const int i = (int)(h[l * 4 + r] / 100.0 + 0.5);
            (void)i;
            const int rr = ((int)(h[l * 4 + r] + 0.5) / 100);   // (i + 1) * (j + 1): decode below instead
            (void)rr; (void)row; (void)col;
        }
    // decode: value = (i + 1) * 100 * (j + 1); j = l & 15 is the natural guess for the column, so i + 1 = value / (100 (j + 1))
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < 4; r++) {
            const int j = l & 15;
            const double irow = h[l * 4 + r] / (100.0 * (j + 1)) - 1.0;
            const int ir = (int)(irow + 0.5);
            if (ir != 4 * r + (l >> 4)) ok_a = 0;
            if (ir != 4 * (l >> 4) + r) ok_b = 0;
            if (l % 16 == 0) printf("lane %2d reg %d: row %d (column %d assumed)\n", l, r, ir, j);
        }
    printf("rows 4 r + (l >> 4): %s;  rows 4 (l >> 4) + r: %s\n", ok_a ? "YES" : "no", ok_b ? "YES" : "no");
    return 0;
}
