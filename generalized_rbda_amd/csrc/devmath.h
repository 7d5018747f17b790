// devmath.h -- device-side building blocks shared by the kernel translation units (kernels.hip: the general
// interpreter; chain_kernels.hip: the chain-structured fast path): constant-address-space plan access, the LDS
// symbol, spatial algebra on (E, r) transforms, the in-register Cholesky, and the tile prologue / epilogue that
// moves a tile between the row-major batch arrays and coordinate-major rows.
// Included inside namespace grbda_hip, after plan.h.
#pragma once

// ---------------------------------------------------------------------------------------------
// plan tables live in the constant address space: uniform loads from it are scalar (s_load),
// which also makes every branch on a plan field a scalar branch
// ---------------------------------------------------------------------------------------------
template <class U>
using cptr = const U __attribute__((address_space(4))) *;

// copy a plan record (all-int32 POD) out of the constant address space; unused fields fold away
template <class U>
__device__ __forceinline__ U load_rec(cptr<U> p)
{
    static_assert(sizeof(U) % 4 == 0, "plan records are arrays of int32");
    U out;
    cptr<int32_t> src = (cptr<int32_t>)p;
    int32_t *dst = reinterpret_cast<int32_t *>(&out);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(U) / 4); i++) dst[i] = src[i];
    return out;
}

// The LDS array is always addressed through this symbol (never through a generic pointer), so
// every access compiles to ds_read / ds_write and never to a flat instruction.
extern __shared__ __attribute__((aligned(16))) unsigned char grbda_smem[];

// ---------------------------------------------------------------------------------------------
// spatial algebra on (E, r) transforms -- src/Utils/SpatialTransforms.cpp:32-157
// ---------------------------------------------------------------------------------------------
__host__ __device__ constexpr int sidx(int i, int j)
{  // packed upper-triangular index of a symmetric 6x6
    return i <= j ? (i * 6 - i * (i - 1) / 2 + (j - i)) : (j * 6 - j * (j - 1) / 2 + (i - j));
}

// E = R_axis(theta) * Et  (ori::coordinateRotation, OrientationTools.h:46-68; XJ * Xtree)
template <class T>
__device__ __forceinline__ void build_E(int axis, T s, T c, cptr<T> Et, T (&E)[9])
{
    if (axis == 0) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            E[j] = Et[j];
            E[3 + j] = c * Et[3 + j] + s * Et[6 + j];
            E[6 + j] = c * Et[6 + j] - s * Et[3 + j];
        }
    } else if (axis == 1) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            E[j] = c * Et[j] - s * Et[6 + j];
            E[3 + j] = Et[3 + j];
            E[6 + j] = s * Et[j] + c * Et[6 + j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            E[j] = c * Et[j] + s * Et[3 + j];
            E[3 + j] = c * Et[3 + j] - s * Et[j];
            E[6 + j] = Et[6 + j];
        }
    }
}

// transformMotionVector: [E w ; E (v - r x w)]
template <class T, class R3>
__device__ __forceinline__ void xmotion(const T (&E)[9], R3 r, const T (&m)[6], T (&o)[6])
{
    const T t0 = m[3] - (r[1] * m[2] - r[2] * m[1]);
    const T t1 = m[4] - (r[2] * m[0] - r[0] * m[2]);
    const T t2 = m[5] - (r[0] * m[1] - r[1] * m[0]);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        o[i] = E[3 * i] * m[0] + E[3 * i + 1] * m[1] + E[3 * i + 2] * m[2];
        o[3 + i] = E[3 * i] * t0 + E[3 * i + 1] * t1 + E[3 * i + 2] * t2;
    }
}

// inverseTransformForceVector: [E^T n + r x (E^T f) ; E^T f]
template <class T, class R3>
__device__ __forceinline__ void xforce_inv(const T (&E)[9], R3 r, const T (&f)[6], T (&o)[6])
{
    T n[3], l[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        n[i] = E[i] * f[0] + E[3 + i] * f[1] + E[6 + i] * f[2];
        l[i] = E[i] * f[3] + E[3 + i] * f[4] + E[6 + i] * f[5];
    }
    o[0] = n[0] + (r[1] * l[2] - r[2] * l[1]);
    o[1] = n[1] + (r[2] * l[0] - r[0] * l[2]);
    o[2] = n[2] + (r[0] * l[1] - r[1] * l[0]);
    o[3] = l[0];
    o[4] = l[1];
    o[5] = l[2];
}

// R = E^T M E for a general 3x3 M (row-major)
template <class T>
__device__ __forceinline__ void rot3(const T (&E)[9], const T (&M)[9], T (&R)[9])
{
    T t[9];  // t = M E
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) t[3 * i + j] = M[3 * i] * E[j] + M[3 * i + 1] * E[3 + j] + M[3 * i + 2] * E[6 + j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) R[3 * i + j] = E[i] * t[j] + E[3 + i] * t[3 + j] + E[6 + i] * t[6 + j];
}

// R = E^T M E for a SYMMETRIC 3x3 M: only the upper triangle of the second product is computed
template <class T>
__device__ __forceinline__ void rot3_sym(const T (&E)[9], const T (&M)[9], T (&R)[9])
{
    T t[9];  // t = M E
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) t[3 * i + j] = M[3 * i] * E[j] + M[3 * i + 1] * E[3 + j] + M[3 * i + 2] * E[6 + j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = i; j < 3; j++) {
            const T v = E[i] * t[j] + E[3 + i] * t[3 + j] + E[6 + i] * t[6 + j];
            R[3 * i + j] = v;
            R[3 * j + i] = v;
        }
}

// B = X^T A X for symmetric 6x6 A (packed), X = (E, r):
// Transform::inverseTransformSpatialInertia (SpatialTransforms.cpp:111-135)
template <class T, class R3, class A21>
__device__ __forceinline__ void congruence(const T (&E)[9], R3 r, const A21 &A, T (&B)[21])
{
    T A11[9], A12[9], A22[9], R11[9], R12[9], R22[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            A11[3 * i + j] = A[sidx(i, j)];
            A12[3 * i + j] = A[sidx(i, 3 + j)];
            A22[3 * i + j] = A[sidx(3 + i, 3 + j)];
        }
    rot3_sym(E, A11, R11);
    rot3(E, A12, R12);
    rot3_sym(E, A22, R22);
    // TR = R12 + r^ R22 ; column j of r^ R22 is r x R22[:, j]
    T TR[9];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        TR[j] = R12[j] + (r[1] * R22[6 + j] - r[2] * R22[3 + j]);
        TR[3 + j] = R12[3 + j] + (r[2] * R22[j] - r[0] * R22[6 + j]);
        TR[6 + j] = R12[6 + j] + (r[0] * R22[3 + j] - r[1] * R22[j]);
    }
    // N = R12 r^ (row i = R12[i,:] x r),  P = TR r^ ;  TL = R11 - N^T - P
    T N[9], Pm[9];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        N[3 * i + 0] = R12[3 * i + 1] * r[2] - R12[3 * i + 2] * r[1];
        N[3 * i + 1] = R12[3 * i + 2] * r[0] - R12[3 * i + 0] * r[2];
        N[3 * i + 2] = R12[3 * i + 0] * r[1] - R12[3 * i + 1] * r[0];
        Pm[3 * i + 0] = TR[3 * i + 1] * r[2] - TR[3 * i + 2] * r[1];
        Pm[3 * i + 1] = TR[3 * i + 2] * r[0] - TR[3 * i + 0] * r[2];
        Pm[3 * i + 2] = TR[3 * i + 0] * r[1] - TR[3 * i + 1] * r[0];
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            if (j >= i) {
                B[sidx(i, j)] = R11[3 * i + j] - N[3 * j + i] - Pm[3 * i + j];
                B[sidx(3 + i, 3 + j)] = R22[3 * i + j];
            }
            B[sidx(i, 3 + j)] = TR[3 * i + j];
        }
}

// The same congruence for a RIGID-BODY inertia A = [[Ibar, h^], [h^T, m 1]] (what SpatialInertia holds: mass, first moment h = m c, rotational
// inertia about the origin) under a proper rotation E:  E^T h^ E = (E^T h)^,  a^ b^ = b a^T - (a . b) 1,  so with h' = E^T h and g = h' + m r
//   B22 = m 1,   B12 = g^,   B11 = E^T Ibar E - (r h'^T + h' r^T + m r r^T) + (2 h' . r + m r . r) 1
// -- ~100 multiply-adds instead of ~260: one symmetric 3 x 3 rotation instead of two and a general one.  Used by the derivative
// recursion and the articulated-inertia factor kernel (every E there is a product of joint rotations and axis permutations; the plan compiler
// refuses them for a model whose body inertias do not have this structure: plan.cpp, DerivProgram::ok).
template <class T, class R3, class A21>
__device__ __forceinline__ void congruence_rigid(const T (&E)[9], R3 r, const A21 &A, T (&B)[21])
{
    T A11[9], R11[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) A11[3 * i + j] = A[sidx(i, j)];
    rot3_sym(E, A11, R11);
    // h^ = [[0, -h2, h1], [h2, 0, -h0], [-h1, h0, 0]] is the upper right block
    const T h0 = A[sidx(2, 4)], h1 = A[sidx(0, 5)], h2 = A[sidx(1, 3)], m = A[sidx(3, 3)];
    T hp[3], g[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        hp[j] = E[j] * h0 + E[3 + j] * h1 + E[6 + j] * h2;  // E^T h
        g[j] = hp[j] + m * r[j];
    }
    const T s = T(2) * (hp[0] * r[0] + hp[1] * r[1] + hp[2] * r[2]) + m * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = i; j < 3; j++) {
            B[sidx(i, j)] = R11[3 * i + j] - (r[i] * hp[j] + hp[i] * r[j] + m * r[i] * r[j]) + (i == j ? s : T(0));
            B[sidx(3 + i, 3 + j)] = i == j ? m : T(0);
        }
    // B12 = g^
    B[sidx(0, 3)] = 0;      B[sidx(0, 4)] = -g[2];  B[sidx(0, 5)] = g[1];
    B[sidx(1, 3)] = g[2];   B[sidx(1, 4)] = 0;      B[sidx(1, 5)] = -g[0];
    B[sidx(2, 3)] = -g[1];  B[sidx(2, 4)] = g[0];   B[sidx(2, 5)] = 0;
}

// y = A x for packed symmetric A
template <class T>
__device__ __forceinline__ void symv(const T (&A)[21], const T (&x)[6], T (&y)[6])
{
#pragma unroll
    for (int i = 0; i < 6; i++) {
        T s = 0;
#pragma unroll
        for (int j = 0; j < 6; j++) s += A[sidx(i, j)] * x[j];
        y[i] = s;
    }
}
template <class T, class A21>
__device__ __forceinline__ void symv_c(const A21 &A /* wave-uniform constants */, const T (&x)[6], T (&y)[6])
{
#pragma unroll
    for (int i = 0; i < 6; i++) {
        T s = 0;
#pragma unroll
        for (int j = 0; j < 6; j++) s += A[sidx(i, j)] * x[j];
        y[i] = s;
    }
}

// forceCrossProduct(a, b) (Spatial.h:177-188)
template <class T>
__device__ __forceinline__ void crf(const T (&a)[6], const T (&b)[6], T (&o)[6])
{
    o[0] = b[2] * a[1] - b[1] * a[2] - b[4] * a[5] + b[5] * a[4];
    o[1] = b[0] * a[2] - b[2] * a[0] + b[3] * a[5] - b[5] * a[3];
    o[2] = b[1] * a[0] - b[0] * a[1] - b[3] * a[4] + b[4] * a[3];
    o[3] = b[5] * a[1] - b[4] * a[2];
    o[4] = b[3] * a[2] - b[5] * a[0];
    o[5] = b[4] * a[0] - b[3] * a[1];
}

// c = motionCrossProduct(v, e_axis * qd) (Spatial.h:131-143): the velocity-product
// acceleration of a revolute joint about a coordinate axis
template <class T>
__device__ __forceinline__ void vxaxis(int axis, const T (&v)[6], T qd, T (&c)[6])
{
    if (axis == 0) {
        c[0] = 0; c[1] = v[2] * qd; c[2] = -v[1] * qd;
        c[3] = 0; c[4] = v[5] * qd; c[5] = -v[4] * qd;
    } else if (axis == 1) {
        c[0] = -v[2] * qd; c[1] = 0; c[2] = v[0] * qd;
        c[3] = -v[5] * qd; c[4] = 0; c[5] = v[3] * qd;
    } else {
        c[0] = v[1] * qd; c[1] = -v[0] * qd; c[2] = 0;
        c[3] = v[4] * qd; c[4] = -v[3] * qd; c[5] = 0;
    }
}

template <class T>
__device__ __forceinline__ T pick(const T (&x)[6], int axis)
{
    return axis == 0 ? x[0] : (axis == 1 ? x[1] : x[2]);
}
template <class T>
__device__ __forceinline__ void column(const T (&A)[21], int axis, T (&h)[6])
{
    if (axis == 0) {
#pragma unroll
        for (int i = 0; i < 6; i++) h[i] = A[sidx(i, 0)];
    } else if (axis == 1) {
#pragma unroll
        for (int i = 0; i < 6; i++) h[i] = A[sidx(i, 1)];
    } else {
#pragma unroll
        for (int i = 0; i < 6; i++) h[i] = A[sidx(i, 2)];
    }
}
template <class T>
__device__ __forceinline__ void add_axis(T (&x)[6], int axis, T val)
{
    if (axis == 0) x[0] += val;
    else if (axis == 1) x[1] += val;
    else x[2] += val;
}

// f32: the hardware sine / cosine (v_sin_f32 / v_cos_f32 on x / 2 pi, 11 instructions) instead of the
// library's sincosf (~155, with a Payne-Hanek path for huge arguments).  Measured on the MIT humanoid and
// JVRC-1 against the fp64 oracle, joint angles up to +-20 rad: max relative error of ydd 2.5e-6 against 0.8e-6
// (tolerance of the path: 1e-3); 3-4 % of the ABA kernel time.  The absolute error grows like |x| * 6e-8 for
// very large angles -- callers who wind joints past ~1e4 rad use the f64 entry points
// (-DGRBDA_PRECISE_SINCOS restores sincosf).
#ifdef GRBDA_PRECISE_SINCOS
__device__ __forceinline__ void sincos_t(float x, float *s, float *c) { sincosf(x, s, c); }
#else
__device__ __forceinline__ void sincos_t(float x, float *s, float *c) { __sincosf(x, s, c); }
#endif
__device__ __forceinline__ void sincos_t(double x, double *s, double *c) { sincos(x, s, c); }
// implicit constraints always use the library functions: K_d^-1 amplifies their error near singular poses
__device__ __forceinline__ void sincos_precise(float x, float *s, float *c) { sincosf(x, s, c); }
__device__ __forceinline__ void sincos_precise(double x, double *s, double *c) { sincos(x, s, c); }

// Implicit constraints in fp32: Cody-Waite reduction to [-pi/4, pi/4] (two fused steps) and the cephes minimax polynomials, ~25
// instructions and ~1 ulp for |x| < 1e4.  The hardware v_sin_f32 / v_cos_f32 behind sincos_t are about ten times less accurate
// (|x| 6e-8 from the 1 / 2 pi scaling alone) and K_d^-1 amplifies that by the constraint's condition number -- measured against the
// oracle compiled in `float` (profiles/r4_gate_f32_oracle.txt): with sincos_t the differential segments were 10x less accurate than
// the dense float restatement, with these they match it; the library's sincosf costs ~155 instructions.
__device__ __forceinline__ void sincos_cw(float x, float *s, float *c)
{
    const float k = __builtin_rintf(x * 0.636619772f);
    float r = __builtin_fmaf(-k, 1.57079637f, x);
    r = __builtin_fmaf(-k, -4.37113883e-8f, r);
    const float z = r * r;
    float sp = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = __builtin_fmaf(z, sp, -1.6666654611e-1f);
    const float sn = __builtin_fmaf(r * z, sp, r);
    float cp = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = __builtin_fmaf(z, cp, 4.166664568298827e-2f);
    const float cs = __builtin_fmaf(z * z, cp, __builtin_fmaf(z, -0.5f, 1.0f));
    const int n = (int)k;
    const float a = (n & 1) ? cs : sn, b = (n & 1) ? sn : cs;
    *s = (n & 2) ? -a : a;
    *c = ((n + 1) & 2) ? -b : b;
}
__device__ __forceinline__ void sincos_cw(double x, double *s, double *c) { sincos(x, s, c); }

// reciprocal and reciprocal square root: f32 takes the hardware approximations (1 ulp; the IEEE division
// expands to ~10 instructions), f64 the exact operations
__device__ __forceinline__ float rcp_t(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double rcp_t(double x) { return 1.0 / x; }
__device__ __forceinline__ float rsqrt_t(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ double rsqrt_t(double x) { return 1.0 / sqrt(x); }

// in-place Cholesky factor + solves for an N x N SPD matrix held in registers.
// The reference inverts D = S^T IA S with ColPivHouseholderQR (ClusterTreeNode.cpp:33-37,
// Utilities.h:325-329); D is SPD so LL^T agrees to rounding.
template <class T, int N>
struct Chol {
    T L[N][N];
    T inv[N];
    __device__ __forceinline__ void factor(const T (&A)[N][N])
    {
#pragma unroll
        for (int j = 0; j < N; j++) {
            T d = A[j][j];
#pragma unroll
            for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k];
            const T rs = rsqrt_t(d);
            inv[j] = rs;
            L[j][j] = d * rs;
#pragma unroll
            for (int i = j + 1; i < N; i++) {
                T s = A[i][j];
#pragma unroll
                for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
                L[i][j] = s * rs;
            }
        }
    }
    __device__ __forceinline__ void solve(T (&b)[N]) const
    {
#pragma unroll
        for (int i = 0; i < N; i++) {
            T s = b[i];
#pragma unroll
            for (int k = 0; k < i; k++) s -= L[i][k] * b[k];
            b[i] = s * inv[i];
        }
#pragma unroll
        for (int i = N - 1; i >= 0; i--) {
            T s = b[i];
#pragma unroll
            for (int k = i + 1; k < N; k++) s -= L[k][i] * b[k];
            b[i] = s * inv[i];
        }
    }
};

// Tile prologue: the batch is row-major ([state][coordinate], the reference's natural vector
// layout), so one wave's 64 states are ONE contiguous block of 64 * ncols scalars.  The wave copies
// that block straight into LDS with asynchronous global->LDS loads (no VGPR round trip, all copies
// of a tile in flight together; LDS holds no live slot at a tile boundary), then reads it back
// transposed and writes coordinate-major rows into its global slab.  Every later access to an
// input is a coalesced 64-element row instead of a 64-line strided gather.
template <class T>
__device__ __forceinline__ void stage_issue(const T *__restrict__ src, size_t tile, int rows_valid, int ncols,
                                            unsigned lds_byte_off, int lane)
{
    // copy as dwords: element type does not matter for a linear copy
    const unsigned *blk = reinterpret_cast<const unsigned *>(src + tile * (size_t)kWave * (size_t)ncols);
    const int n_dw = kWave * ncols * (int)(sizeof(T) / 4);
    const int n_valid = rows_valid * ncols * (int)(sizeof(T) / 4);
    for (int base = 0; base < n_dw; base += kWave) {
        if (base + lane < n_valid)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(blk + base + lane),
                                             (__attribute__((address_space(3))) void *)(grbda_smem + lds_byte_off + (unsigned)base * 4u),
                                             4, 0, 0);
    }
}
// A workgroup is one wavefront, so exchanging data between lanes through LDS needs no s_barrier (and none of
// the memory-wide waits __syncthreads implies): LDS operations of a wave execute in order, the LDS counter
// only has to reach zero and the compiler must not move accesses across the point.
__device__ __forceinline__ void wave_lds_fence()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
// The staged block is the row-major [state][column] copy of the inputs; lane l reads its own row.  With an even number of
// columns a column-at-a-time read hits the same LDS banks from many lanes (24 columns: 8-way conflicts, measured as 39 %
// of the LDS-active cycles of the forward dynamics); 16-byte reads of the lane's own row (rows are 16-byte aligned when
// the row size is a multiple of 16 bytes) conflict at most 2-way.
template <class T>
__device__ __forceinline__ void stage_transpose(int ncols, unsigned lds_byte_off, T *slab_rows, int lane)
{
    const T *stage = reinterpret_cast<const T *>(grbda_smem + lds_byte_off);
    constexpr int V = 16 / (int)sizeof(T);  // elements per 16-byte read
    if ((ncols % V) == 0 && (lds_byte_off % 16u) == 0) {
        struct alignas(16) Vec { T v[V]; };
        const Vec *rows = reinterpret_cast<const Vec *>(stage + lane * ncols);
        for (int c = 0; c < ncols; c += V) {
            const Vec x = rows[c / V];
#pragma unroll
            for (int k = 0; k < V; k++) slab_rows[(size_t)(c + k) * kWave + lane] = x.v[k];
        }
        return;
    }
    for (int c = 0; c < ncols; c++) slab_rows[(size_t)c * kWave + lane] = stage[lane * ncols + c];
}
template <class T>
__device__ __forceinline__ void stage_inputs(const T *__restrict__ q, const T *__restrict__ qd, const T *__restrict__ x,
                                             size_t tile, int rows_valid, int nq, int nv, T *slab, int lane,
                                             int lds_bytes)
{
    const unsigned bq = (unsigned)(kWave * nq) * (unsigned)sizeof(T), bv = (unsigned)(kWave * nv) * (unsigned)sizeof(T);
    if ((int)(bq + 2 * bv) <= lds_bytes) {
        stage_issue(q, tile, rows_valid, nq, 0u, lane);
        stage_issue(qd, tile, rows_valid, nv, bq, lane);
        stage_issue(x, tile, rows_valid, nv, bq + bv, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_lds_fence();
        stage_transpose(nq, 0u, slab, lane);
        stage_transpose(nv, bq, slab + (size_t)nq * kWave, lane);
        stage_transpose(nv, bq + bv, slab + (size_t)(nq + nv) * kWave, lane);
        wave_lds_fence();
    } else {
        // LDS too small for the whole tile: one array at a time (capi.cpp guarantees each one fits)
        stage_issue(q, tile, rows_valid, nq, 0u, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_lds_fence();
        stage_transpose(nq, 0u, slab, lane);
        wave_lds_fence();
        stage_issue(qd, tile, rows_valid, nv, 0u, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_lds_fence();
        stage_transpose(nv, 0u, slab + (size_t)nq * kWave, lane);
        wave_lds_fence();
        stage_issue(x, tile, rows_valid, nv, 0u, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_lds_fence();
        stage_transpose(nv, 0u, slab + (size_t)(nq + nv) * kWave, lane);
        wave_lds_fence();
    }
}

// Tile epilogue: the nv result rows of the slab ([coordinate][state]) become the tile's [state][coordinate]
// block of the output array, transposed through LDS (free again: the tile's state is dead) so that the
// global stores are contiguous.
template <class T>
__device__ __forceinline__ void write_outputs(const T *rows, T *__restrict__ out, size_t tile, int rows_valid, int nv,
                                              int lane)
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the row stores have landed
    T *stage = reinterpret_cast<T *>(grbda_smem);
    constexpr int V = 16 / (int)sizeof(T);
    if ((nv % V) == 0) {  // 16-byte writes of the lane's own row (see stage_transpose)
        struct alignas(16) Vec { T v[V]; };
        Vec *mine = reinterpret_cast<Vec *>(stage + lane * nv);
        for (int c = 0; c < nv; c += V) {
            Vec x;
#pragma unroll
            for (int k = 0; k < V; k++) x.v[k] = rows[(size_t)(c + k) * kWave + lane];
            mine[c / V] = x;
        }
    } else {
        for (int c = 0; c < nv; c++) stage[lane * nv + c] = rows[(size_t)c * kWave + lane];
    }
    wave_lds_fence();
    T *dst = out + tile * (size_t)kWave * (size_t)nv;
    const int total = rows_valid * nv;
    for (int i = 0; i < nv; i++) {
        const int j = i * kWave + lane;
        if (j < total) dst[j] = stage[j];
    }
    wave_lds_fence();
}

// The same with the result rows in LDS ([coordinate][lane] from row `out_row` on, written by the acceleration run): no slab
// round trip and no wait for global stores.  The state-major staging block is the first nv rows of LDS when the result rows sit
// above them, else the nv rows behind the result rows (the plan compiler reserved them).
template <class T>
__device__ __forceinline__ void write_outputs_lds(int out_row, T *__restrict__ out, size_t tile, int rows_valid, int nv, int lane)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const T *rows = reinterpret_cast<const T *>(grbda_smem) + out_row * kWave;
    T *stage = reinterpret_cast<T *>(grbda_smem) + (out_row >= nv ? 0 : (out_row + nv) * kWave);
    constexpr int V = 16 / (int)sizeof(T);
    if ((nv % V) == 0) {
        struct alignas(16) Vec { T v[V]; };
        Vec *mine = reinterpret_cast<Vec *>(stage + lane * nv);
        for (int c = 0; c < nv; c += V) {
            Vec x;
#pragma unroll
            for (int k = 0; k < V; k++) x.v[k] = rows[(c + k) * kWave + lane];
            mine[c / V] = x;
        }
    } else {
        for (int c = 0; c < nv; c++) stage[lane * nv + c] = rows[c * kWave + lane];
    }
    wave_lds_fence();
    T *dst = out + tile * (size_t)kWave * (size_t)nv;
    const int total = rows_valid * nv;
    for (int i = 0; i < nv; i++) {
        const int j = i * kWave + lane;
        if (j < total) dst[j] = stage[j];
    }
    wave_lds_fence();
}

// quaternionToRotationMatrix (OrientationTools.h:251-269) / rpyToRotMat (:121-130)
template <class T>
__device__ __forceinline__ void free_rotation(int ori_repr, const T *o, T (&E)[9])
{
    if (ori_repr == 0) {
        const T e0 = o[0], e1 = o[1], e2 = o[2], e3 = o[3];
        E[0] = 1 - 2 * (e2 * e2 + e3 * e3); E[3] = 2 * (e1 * e2 - e0 * e3);     E[6] = 2 * (e1 * e3 + e0 * e2);
        E[1] = 2 * (e1 * e2 + e0 * e3);     E[4] = 1 - 2 * (e1 * e1 + e3 * e3); E[7] = 2 * (e2 * e3 - e0 * e1);
        E[2] = 2 * (e1 * e3 - e0 * e2);     E[5] = 2 * (e2 * e3 + e0 * e1);     E[8] = 1 - 2 * (e1 * e1 + e2 * e2);
    } else {
        T sx, cx, sy, cy, sz, cz;
        sincos_t(o[0], &sx, &cx);
        sincos_t(o[1], &sy, &cy);
        sincos_t(o[2], &sz, &cz);
        // Rx * Ry * Rz with coordinate rotations
        const T Rxy[9] = {cy, 0, -sy, sx * sy, cx, sx * cy, cx * sy, -sx, cx * cy};
#pragma unroll
        for (int i = 0; i < 3; i++) {
            E[3 * i + 0] = Rxy[3 * i] * cz - Rxy[3 * i + 1] * sz;
            E[3 * i + 1] = Rxy[3 * i] * sz + Rxy[3 * i + 1] * cz;
            E[3 * i + 2] = Rxy[3 * i + 2];
        }
    }
}

template <class T, class E9>
__device__ __forceinline__ void rotate_z(T s, T c, const E9 &Et, T (&E)[9])
{
#pragma unroll
    for (int j = 0; j < 3; j++) {
        E[j] = c * Et[j] + s * Et[3 + j];
        E[3 + j] = c * Et[3 + j] - s * Et[j];
        E[6 + j] = Et[6 + j];
    }
}

// ---------------------------------------------------------------------------------------------
// Links whose tree rotation is a cyclic permutation of the axes (plan.cpp: Et in {1, C, C^2} after the canonical joint axes -- every
// joint of JVRC-1, every joint of the other robots whose URDF <origin> carries no rotation): E = Rz(q) P_K with
// (P_K x)_i = x_((i + K) % 3).  The products with E then cost 4 multiply-adds per 3-vector instead of 9 and a 3 x 3 block congruence
// 22-24 instead of 45-54; the permutation is compile-time re-indexing.  K is wave-uniform (ChainLink::perm): the callers switch.
// ---------------------------------------------------------------------------------------------
template <class T, int K>
__device__ __forceinline__ void rzp_apply(T s, T c, const T (&x)[3], T (&o)[3])  // o = Rz P_K x
{
    const T w0 = x[K % 3], w1 = x[(1 + K) % 3], w2 = x[(2 + K) % 3];
    o[0] = c * w0 + s * w1;
    o[1] = c * w1 - s * w0;
    o[2] = w2;
}
template <class T, int K>
__device__ __forceinline__ void rzp_apply_t(T s, T c, const T (&y)[3], T (&o)[3])  // o = P_K^T Rz^T y
{
    o[K % 3] = c * y[0] - s * y[1];
    o[(1 + K) % 3] = s * y[0] + c * y[1];
    o[(2 + K) % 3] = y[2];
}
template <class T, int K, class R3>
__device__ __forceinline__ void xmotion_p(T s, T c, R3 r, const T (&m)[6], T (&o)[6])
{
    const T w[3] = {m[0], m[1], m[2]};
    const T t[3] = {m[3] - (r[1] * m[2] - r[2] * m[1]), m[4] - (r[2] * m[0] - r[0] * m[2]), m[5] - (r[0] * m[1] - r[1] * m[0])};
    T a[3], b[3];
    rzp_apply<T, K>(s, c, w, a);
    rzp_apply<T, K>(s, c, t, b);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        o[i] = a[i];
        o[3 + i] = b[i];
    }
}
template <class T, int K, class R3>
__device__ __forceinline__ void xforce_inv_p(T s, T c, R3 r, const T (&f)[6], T (&o)[6])
{
    const T fa[3] = {f[0], f[1], f[2]}, fl[3] = {f[3], f[4], f[5]};
    T n[3], l[3];
    rzp_apply_t<T, K>(s, c, fa, n);
    rzp_apply_t<T, K>(s, c, fl, l);
    o[0] = n[0] + (r[1] * l[2] - r[2] * l[1]);
    o[1] = n[1] + (r[2] * l[0] - r[0] * l[2]);
    o[2] = n[2] + (r[0] * l[1] - r[1] * l[0]);
    o[3] = l[0];
    o[4] = l[1];
    o[5] = l[2];
}
// R = E^T M E for E = Rz P_K: M' = Rz^T M Rz (rows and columns 0, 1 mix), then R[a][b] = M'[sigma^-1(a)][sigma^-1(b)]
template <class T, int K, bool SYM>
__device__ __forceinline__ void rot3_p(T s, T c, const T (&M)[9], T (&R)[9])
{
    T t[9], m[9];  // t = M Rz ; m = Rz^T t
#pragma unroll
    for (int i = 0; i < 3; i++) {
        t[3 * i] = c * M[3 * i] - s * M[3 * i + 1];
        t[3 * i + 1] = s * M[3 * i] + c * M[3 * i + 1];
        t[3 * i + 2] = M[3 * i + 2];
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
        if (!SYM || j >= 0) m[j] = c * t[j] - s * t[3 + j];
        if (!SYM || j >= 1) m[3 + j] = s * t[j] + c * t[3 + j];
        m[6 + j] = t[6 + j];
    }
    if (SYM) {
        m[3] = m[1];
        m[6] = m[2];
        m[7] = m[5];
    }
    // sigma(i) = (i + K) % 3 maps an index of the permuted vector to the original one: (P x)_i = x_sigma(i); E^T M E = P^T M' P,
    // (P^T M' P)[sigma(i)][sigma(j)] = M'[i][j]
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) R[3 * ((i + K) % 3) + (j + K) % 3] = m[3 * i + j];
}
template <class T, int K, class R3, class A21>
__device__ __forceinline__ void congruence_p(T s, T c, R3 r, const A21 &A, T (&B)[21])
{
    T A11[9], A12[9], A22[9], R11[9], R12[9], R22[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            A11[3 * i + j] = A[sidx(i, j)];
            A12[3 * i + j] = A[sidx(i, 3 + j)];
            A22[3 * i + j] = A[sidx(3 + i, 3 + j)];
        }
    rot3_p<T, K, true>(s, c, A11, R11);
    rot3_p<T, K, false>(s, c, A12, R12);
    rot3_p<T, K, true>(s, c, A22, R22);
    T TR[9];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        TR[j] = R12[j] + (r[1] * R22[6 + j] - r[2] * R22[3 + j]);
        TR[3 + j] = R12[3 + j] + (r[2] * R22[j] - r[0] * R22[6 + j]);
        TR[6 + j] = R12[6 + j] + (r[0] * R22[3 + j] - r[1] * R22[j]);
    }
    T N[9], Pm[9];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        N[3 * i + 0] = R12[3 * i + 1] * r[2] - R12[3 * i + 2] * r[1];
        N[3 * i + 1] = R12[3 * i + 2] * r[0] - R12[3 * i + 0] * r[2];
        N[3 * i + 2] = R12[3 * i + 0] * r[1] - R12[3 * i + 1] * r[0];
        Pm[3 * i + 0] = TR[3 * i + 1] * r[2] - TR[3 * i + 2] * r[1];
        Pm[3 * i + 1] = TR[3 * i + 2] * r[0] - TR[3 * i + 0] * r[2];
        Pm[3 * i + 2] = TR[3 * i + 0] * r[1] - TR[3 * i + 1] * r[0];
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            if (j >= i) {
                B[sidx(i, j)] = R11[3 * i + j] - N[3 * j + i] - Pm[3 * i + j];
                B[sidx(3 + i, 3 + j)] = R22[3 * i + j];
            }
            B[sidx(i, 3 + j)] = TR[3 * i + j];
        }
}

// y = A x for packed symmetric A and a revolute-about-z velocity product x = (x0, x1, 0, x3, x4, 0)
template <class T, class A21>
__device__ __forceinline__ void symv_z(const A21 &A, const T (&x)[6], T (&y)[6])
{
#pragma unroll
    for (int i = 0; i < 6; i++)
        y[i] = A[sidx(i, 0)] * x[0] + A[sidx(i, 1)] * x[1] + A[sidx(i, 3)] * x[3] + A[sidx(i, 4)] * x[4];
}

