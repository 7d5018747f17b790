// deriv_kernels.hip -- first-order derivatives of the cluster dynamics (BASELINE config 5: d ydd / d q, d qd, d tau).
//
// The reference has no derivative ALGORITHM: it differentiates CasADi graphs of its forward dynamics and validates them
// against central differences (UnitTests/testForwardDynamicsDerivatives.cpp, testHelpers.hpp:50-112).  Here:
//   ydd = FD(q, qd, tau)                                              (chain_kernels.hip / kernels.hip)
//   Dq = d ID / d q, Dqd = d ID / d qd at (q, qd, ydd)                 rnea_deriv_kernel, analytic, one state per lane
//   H                                                                 crba_kernels.hip
//   d ydd / d tau = H^-1,  d ydd / d q = -H^-1 Dq,  d ydd / d qd = -H^-1 Dqd      spd_solve_kernel, one state per wavefront
// (ID(q, qd, FD(q, qd, tau)) = tau differentiated).  Explicit (constant G) clusters only; models with implicit loops keep
// the central differences of capi.cpp.
//
// Inverse-dynamics derivatives.  The recursion is the spatial-vector form of the RNEA derivatives (Carpentier & Mansard,
// RSS 2018; Singh, Russell & Wensing, RA-L 2022) applied to the SPANNING tree, every body a 1-DoF revolute joint (plus
// the free base), and then projected with the clusters' constant G:  d tau_y / d y = G^T (d tau_span / d q_span) G.
// All spatial quantities are expressed in ONE inertial frame F that coincides with the floating base at this instant
// (the world for fixed-base models), so composite quantities add without transforms and an entry of the result is a dot
// product of a descendant-side and an ancestor-side 6-vector.  With S_j the joint axis, Sd_j = v_j x S_j,
// Pd_j = v_parent x S_j (= Sd_j for a revolute joint), Pdd_j = a_parent x S_j + v_parent x Pd_j and, per body,
// B_i = (v x*) I - I (v x) + (I v) xbar*,  Ic / Bc / Fc the sums of I, B, f = I a + v x* I v over the subtree:
//    j ancestor of or equal to k:  d tau_k / d q_j  = Pd_j . (Bc_k^T S_k) + Pdd_j . (Ic_k S_k)
//                                  d tau_k / d qd_j = S_j . (Bc_k^T S_k) + (Sd_j + Pd_j) . (Ic_k S_k)
//    k strict ancestor of j:       d tau_k / d q_j  = S_k . (S_j x* Fc_j + Bc_j Pd_j + Ic_j Pdd_j)
//                                  d tau_k / d qd_j = S_k . (Bc_j S_j + Ic_j (Sd_j + Pd_j))
// The columns of the free base are the body-frame twists of the reference's tangent step (pos += R^T d,
// quat += quat (x) (0, d) / 2): S = 1, Sd = v x 1, Pd = 0, Pdd = a_0 x 1 with a_0 = -gravity seen from the base.
// tests/deriv_recursion_numpy.py is the numpy statement of the same recursion, checked against differences of the oracle.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "devplan.h"

namespace grbda_hip {

#include "devmath.h"

// optional in-kernel cycle accounting of the derivative recursion (make expd NAME=dprof DEFS=-DGRBDA_DERIV_PROFILE, tools/deriv_prof.py;
// never in the shipped library): s_memtime deltas per phase, lane i keeps bucket i, summed over wavefronts at the end
#ifdef GRBDA_DERIV_PROFILE
__device__ unsigned long long grbda_deriv_prof[64];
#define DPROF_T0() unsigned long long dprof_t = __builtin_amdgcn_s_memtime(), dprof_acc = 0, dprof_cnt = 0
#define DPROF_ADD(i)                                                                      \
    do {                                                                                  \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                     \
        if (lane == (i)) {                                                                \
            dprof_acc += now_ - dprof_t;                                                  \
            dprof_cnt += 1;                                                               \
        }                                                                                 \
        dprof_t = now_;                                                                   \
    } while (0)
#define DPROF_END()                                                                       \
    do {                                                                                  \
        if (lane < 32) {                                                                  \
            atomicAdd(&grbda_deriv_prof[lane], dprof_acc);                                \
            atomicAdd(&grbda_deriv_prof[32 + lane], dprof_cnt);                           \
        }                                                                                 \
    } while (0)
extern "C" int grbda_debug_deriv_profile(unsigned long long *out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(grbda_deriv_prof), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[64] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(grbda_deriv_prof), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#else
#define DPROF_T0()
#define DPROF_ADD(i)
#define DPROF_END()
#endif

// ---------------------------------------------------------------------------------------------------------------
// spatial helpers in the common frame
// ---------------------------------------------------------------------------------------------------------------
// motion cross product a x b
template <class T>
__device__ __forceinline__ void crm(const T (&a)[6], const T (&b)[6], T (&o)[6])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
    o[3] = a[1] * b[5] - a[2] * b[4] + a[4] * b[2] - a[5] * b[1];
    o[4] = a[2] * b[3] - a[0] * b[5] + a[5] * b[0] - a[3] * b[2];
    o[5] = a[0] * b[4] - a[1] * b[3] + a[3] * b[1] - a[4] * b[0];
}
template <class T>
__device__ __forceinline__ T dot6(const T (&a)[6], const T (&b)[6])
{
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}
// y = M x, y = M^T x for a row-major 6 x 6
template <class T>
__device__ __forceinline__ void mv6(const T (&M)[36], const T (&x)[6], T (&y)[6])
{
#pragma unroll
    for (int i = 0; i < 6; i++) {
        T s = 0;
#pragma unroll
        for (int j = 0; j < 6; j++) s += M[6 * i + j] * x[j];
        y[i] = s;
    }
}
template <class T>
__device__ __forceinline__ void mtv6(const T (&M)[36], const T (&x)[6], T (&y)[6])
{
#pragma unroll
    for (int j = 0; j < 6; j++) {
        T s = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) s += M[6 * i + j] * x[i];
        y[j] = s;
    }
}
// B = (v x*) I - I (v x) + (I v) xbar*, column by column: B e_j = v x* (I e_j) - I (v x e_j) + e_j x* (I v)
template <class T>
__device__ __forceinline__ void body_B(const T (&I)[21], const T (&v)[6], const T (&h)[6], T (&Bm)[36])
{
#pragma unroll
    for (int j = 0; j < 6; j++) {
        T e[6], col[6], c1[6], ve[6], c2[6], c3[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            e[i] = i == j ? T(1) : T(0);
            col[i] = I[sidx(i, j)];
        }
        crf(v, col, c1);
        crm(v, e, ve);
        symv(I, ve, c2);
        crf(e, h, c3);
#pragma unroll
        for (int i = 0; i < 6; i++) Bm[6 * i + j] = c1[i] - c2[i] + c3[i];
    }
}

// slab rows of one lane
template <class T>
struct Rows {
    T *p;  // slab + lane
    template <int N>
    __device__ __forceinline__ void ld(int row, T (&x)[N]) const
    {
#pragma unroll
        for (int i = 0; i < N; i++) x[i] = p[(size_t)(row + i) * kWave];
    }
    template <int N>
    __device__ __forceinline__ void st(int row, const T (&x)[N]) const
    {
#pragma unroll
        for (int i = 0; i < N; i++) p[(size_t)(row + i) * kWave] = x[i];
    }
};

// kinematics of a revolute body in F from its parent's: E, p (F -> body), v, a, S, Sd = v x S, Pdd = a_p x S + v_p x Sd
template <class T>
__device__ __forceinline__ void deriv_kin(cptr<T> C, bool axisym, T qi, T qdi, T qddi, const T (&Ep)[9], const T (&pp)[3],
                                          const T (&vp)[6], const T (&ap)[6], T (&E)[9], T (&p)[3], T (&v)[6], T (&a)[6], T (&S)[6],
                                          T (&Sd)[6], T (&Pdd)[6])
{
    T sn = 0, cs = 1, El[9];
    if (!axisym) sincos_t(qi, &sn, &cs);   // a rotor's inertia, axis and velocity in F do not depend on its own angle
    rotate_z(sn, cs, C, El);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) E[3 * i + j] = El[3 * i] * Ep[j] + El[3 * i + 1] * Ep[3 + j] + El[3 * i + 2] * Ep[6 + j];
#pragma unroll
    for (int i = 0; i < 3; i++) p[i] = pp[i] + Ep[i] * C[9] + Ep[3 + i] * C[10] + Ep[6 + i] * C[11];
    // joint axis: the body's z axis seen from F, through the body origin
    S[0] = E[6]; S[1] = E[7]; S[2] = E[8];
    S[3] = p[1] * S[2] - p[2] * S[1];
    S[4] = p[2] * S[0] - p[0] * S[2];
    S[5] = p[0] * S[1] - p[1] * S[0];
    crm(vp, S, Sd);
    T t[6];
    crm(ap, S, Pdd);
    crm(vp, Sd, t);
#pragma unroll
    for (int j = 0; j < 6; j++) {
        Pdd[j] += t[j];
        v[j] = vp[j] + S[j] * qdi;
        a[j] = ap[j] + S[j] * qddi + Sd[j] * qdi;
    }
}

// NMAX: the largest number of coordinates of a cluster of the model (1, 2 or NMAX).  The cluster-level
// vectors are sized by it, and with them the register budget (the kernel runs one wavefront per SIMD: it is bound by
// the traffic of its slab rows, two wavefronts per SIMD measured no faster).
// a row of the six base columns of d tau / d q: from body-twist columns [rotation; translation] to the columns of the
// roll-pitch-yaw coordinates [position; angles] (see the kernel)
template <class T>
__device__ __forceinline__ void rpy_columns(const T (&Rb)[9], const T (&Tb)[9], T (&row)[6])
{
    T out[6];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        out[a] = row[3] * Rb[a] + row[4] * Rb[3 + a] + row[5] * Rb[6 + a];
        out[3 + a] = row[0] * Tb[a] + row[1] * Tb[3 + a] + row[2] * Tb[6 + a];
    }
#pragma unroll
    for (int a = 0; a < 6; a++) row[a] = out[a];
}

extern __shared__ __attribute__((aligned(16))) unsigned char deriv_smem[];

// The 63 composites [I | B | f] a cluster hands to its parent body.  fp32: registers.  fp64: LDS, entry j of lane l at
// [j][l] -- the fp64 kernel is far beyond its 512 registers, and what does not fit goes to scratch memory.
template <class T, bool IN_LDS>
struct PartStore {
    T r[IN_LDS ? 1 : 63];
    T *lds;  // (IN_LDS) the wave's block, already offset by the lane
    __device__ __forceinline__ T get(int j) const
    {
        if constexpr (IN_LDS) return lds[j * kWave];
        else return r[j];
    }
    __device__ __forceinline__ void set(int j, T v)
    {
        if constexpr (IN_LDS) lds[j * kWave] = v;
        else r[j] = v;
    }
};

// IL: interleave factor of the result workspace.  1: state-major, [state][entry].  kDerivGroup (= 4): [group of 4 states][entry][4]
// -- what the matrix-core solve reads.  With one state per lane a store instruction of the state-major layout opens 64 cache
// lines per result stream, 192 per wavefront; they do not survive in L2 until their other 31 entries arrive, are written back
// partially and fetched again: 35 KB written and 30 KB fetched per JVRC-1 state for 14.4 KB of results, 3.9 TB/s -- the kernel
// was bound by that (round 3, calibrated FETCH_SIZE / WRITE_SIZE).  Interleaving four states makes every store a run of
// 16-byte pieces, a quarter of the open lines (measured 2.2 -> 1.45 ms per 131 072 states at four wavefronts per CU;
// factors 8, 16 and 64 add 2-4 % more and cost the solve its cooperative 4-wavefront copy).
// (experiment builds: -DGRBDA_EXP_DERIV_WPS=2 asks for two wavefronts per SIMD, -DGRBDA_EXP_PART_LDS keeps the carried composites in LDS
// for fp32 too)
#ifndef GRBDA_EXP_DERIV_WPS
#define GRBDA_EXP_DERIV_WPS 1
#endif
// (experiment builds: -DGRBDA_EXP_ANC_LEVELS=n keeps n levels of the ancestor-row cache in LDS instead of kDerivAncLevels -- the cache is 4.6 KB per
// level and wavefront, 36.9 KB at eight levels: FOUR wavefronts per CU whatever the register count)
// Round 5: the cache is OFF by default (0 levels) -- with it or without it the kernel takes the same time (28.0 against 28.7 ms per million
// JVRC-1 states, profiles/r5_deriv_recursion_experiments.txt) -- and its LDS holds the tile's staged inputs instead (kStageInputsF32).
#ifndef GRBDA_EXP_ANC_LEVELS
#define GRBDA_EXP_ANC_LEVELS 0
#endif
// fp32: the tile's q / qd / ydd blocks are copied to LDS once (coalesced LDS-DMA, as the chain kernels stage theirs) and every body reads its
// coordinates from the lane's own row there.  Read straight from the caller's arrays they are 4-byte accesses 150 bytes apart: 64 cache
// lines per load instruction, three loads per body in pass 1 and again per leaf body in pass 2 -- the counters showed 27 KB fetched per state
// for a kernel whose slab rows account for 12 (profiles/r5_rocprofv3_pmc_derivatives.txt; the kernel moves 4.5 TB/s at the fabric).
#ifndef GRBDA_EXP_NO_STAGE
constexpr bool kStageInputsF32 = true;
#else
constexpr bool kStageInputsF32 = false;
#endif
constexpr int kAncLevelsLds = GRBDA_EXP_ANC_LEVELS;
#ifdef GRBDA_EXP_PART_LDS
constexpr bool kPartLdsF32 = true;
#else
constexpr bool kPartLdsF32 = false;
#endif
template <class T, int NMAX, int IL>
__global__ __launch_bounds__(kWave, GRBDA_EXP_DERIV_WPS) void rnea_deriv_kernel(DevPlan<T> DP, const DerivBody *__restrict__ db_, int n_clusters, int n_rows,
                                                              const T *__restrict__ q, const T *__restrict__ qd,
                                                              const T *__restrict__ ydd, T *__restrict__ Dq, T *__restrict__ Dqd,
                                                              T *__restrict__ H, size_t B, T *__restrict__ scratch)
{
    cptr<ClusterRec> clusters = (cptr<ClusterRec>)DP.clusters;
    cptr<BodyRec> bodies = (cptr<BodyRec>)DP.bodies;
    cptr<T> consts = (cptr<T>)DP.consts;
    cptr<DerivBody> db = (cptr<DerivBody>)db_;
    const int lane = threadIdx.x, nq = DP.nq, nv = DP.nv;
    Rows<T> R;
    R.p = scratch + (size_t)blockIdx.x * (size_t)n_rows * kWave + lane;
    const size_t n_tiles = (B + kWave - 1) / kWave;
    DPROF_T0();
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t r = tile * kWave + lane;
        const size_t st = r < B ? r : B - 1;  // lanes past the end redo the last state and do not store
#ifdef GRBDA_EXP_NO_STORE
        const bool live = r < B && DP.nq < 0;
#else
        const bool live = r < B;
#endif
        const T *qs = q + st * (size_t)nq, *qds = qd + st * (size_t)nv, *ydds = ydd + st * (size_t)nv;
        if constexpr (sizeof(T) == 4 && kStageInputsF32) {
            // (lanes past the end of the batch read the last valid row, as they redo the last state)
            const size_t left = B - tile * kWave;
            const int rows_valid = left < (size_t)kWave ? (int)left : kWave;
            const unsigned in0 = (unsigned)((kPartLdsF32 ? 63 * kWave : 0) + kAncLevelsLds * 18 * kWave) * (unsigned)sizeof(T);
            const unsigned bq = (unsigned)(kWave * nq) * (unsigned)sizeof(T), bv = (unsigned)(kWave * nv) * (unsigned)sizeof(T);
            wave_lds_fence();  // the previous tile's reads of the block are done
            stage_issue(q, tile, rows_valid, nq, in0, lane);
            stage_issue(qd, tile, rows_valid, nv, in0 + bq, lane);
            stage_issue(ydd, tile, rows_valid, nv, in0 + bq + bv, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wave_lds_fence();
            const int my = lane < rows_valid ? lane : rows_valid - 1;
            const T *blk = reinterpret_cast<const T *>(deriv_smem + in0);
            qs = blk + my * nq;
            qds = blk + kWave * nq + my * nv;
            ydds = blk + kWave * (nq + nv) + my * nv;
        }
        // output layout, per state and matrix nv^2 entries: coordinate r owns the run [r^2, (r + 1)^2): first d tau_r / d x_c
        // for c = 0 .. r, then d tau_c / d x_r for c = 0 .. r-1.  Every store of a coordinate's pass (its own cluster, then
        // ancestor by ancestor) lands in that coordinate's run, so a cache line is completed by consecutive instructions
        // instead of being revisited from every descendant (the scattered transposed stores of the plain layout cost 2 of
        // 6.6 ms on JVRC-1); spd_solve_kernel reads column c as entry c of the runs r >= c, and the second half of run c.
        // (interleaved: state r sits at sub-position r % IL of group r / IL; groups are IL * nv^2 entries apart)
        const size_t grp_of = st / IL, sub_of = st % IL;
        T *Dqs = Dq ? Dq + grp_of * (size_t)nv * nv * IL + sub_of : nullptr, *Dqds = Dqd ? Dqd + grp_of * (size_t)nv * nv * IL + sub_of : nullptr;
        // the joint-space inertia matrix falls out of the same composites: H[k][j] = S_j . (Ic_k S_k) for j ancestor of or
        // equal to k (the CRBA in the common frame); the rows of its lower triangle, back to back, when the caller wants it
        T *Hs = H ? H + grp_of * (size_t)nv * nv * IL + sub_of : nullptr;
        // (Dq == Dqd == nullptr: the caller wants H alone -- the spanning-tree route of capi.cpp's projection_run)
        auto put = [&](T *P, int r, int c, T v) {
            if (!P) return;
            if (c <= r) P[(size_t)(r * r + c) * IL] = v;
            else P[(size_t)(c * c + c + 1 + r) * IL] = v;
        };
        auto put_h = [&](int r, int c, T v) {
            if (Hs && c <= r) Hs[(size_t)(r * (r + 1) / 2 + c) * IL] = v;
        };
        // gravity as the acceleration of the frame F (TreeModel.cpp:40-43: a_root = -gravity), base velocity
        T a0[6], vb[6];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            a0[j] = DP.a_root[j];
            vb[j] = 0;
        }
        // roll-pitch-yaw base: the reference's tangent step is plain q + dq there (position in world coordinates, then the
        // angles), so the base's body-twist columns [rotation; translation] of d tau / d q go through
        // d(twist) / d(pos, rpy) = [[0, Tb], [Rb, 0]]: Rb the base rotation, Tb the body angular velocity per unit angle rate
        T Rb[9], Tb[9];
#pragma unroll
        for (int j = 0; j < 9; j++) Rb[j] = Tb[j] = (j % 4 == 0) ? T(1) : T(0);
        const bool rpy = DP.ori_repr == 1;
        // ---- pass 1, root side first: kinematics in F of every body that has children ----
        int last_gb = -1;
        T last_kin[24];
#pragma unroll
        for (int j = 0; j < 24; j++) last_kin[j] = 0;
        for (int c = 0; c < n_clusters; c++) {
            const ClusterRec cr = load_rec(clusters + c);
            if (cr.kind == CK_FREE) {
                T o[4], Eb[9], rb[3], g[6], kin[24];
#pragma unroll
                for (int j = 0; j < 4; j++) o[j] = (rpy && j == 3) ? T(0) : qs[cr.q_index + 3 + j];
                free_rotation(DP.ori_repr, o, Eb);
                if (rpy) {
                    T sx, cx, sy, cy;
                    sincos_t(o[0], &sx, &cx);
                    sincos_t(o[1], &sy, &cy);
                    Tb[0] = 1; Tb[1] = 0; Tb[2] = -sy;
                    Tb[3] = 0; Tb[4] = cx; Tb[5] = sx * cy;
                    Tb[6] = 0; Tb[7] = -sx; Tb[8] = cx * cy;
#pragma unroll
                    for (int j = 0; j < 9; j++) Rb[j] = Eb[j];
                }
#pragma unroll
                for (int j = 0; j < 3; j++) rb[j] = qs[cr.q_index + j];
#pragma unroll
                for (int j = 0; j < 6; j++) g[j] = DP.a_root[j];
                xmotion(Eb, rb, g, a0);
#pragma unroll
                for (int j = 0; j < 9; j++) kin[j] = (j % 4 == 0) ? T(1) : T(0);
#pragma unroll
                for (int j = 0; j < 3; j++) kin[9 + j] = 0;
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    vb[j] = qds[cr.v_index + j];
                    kin[12 + j] = vb[j];
                    kin[18 + j] = a0[j] + ydds[cr.v_index + j];
                }
                const DerivBody x = load_rec(db + cr.first_body);
                if (x.kin_row >= 0) R.st(x.kin_row, kin);
                last_gb = cr.first_body;
#pragma unroll
                for (int j = 0; j < 24; j++) last_kin[j] = kin[j];
                continue;
            }
            for (int i = 0; i < cr.k; i++) {
                if (!((cr.child_mask >> i) & 1)) continue;
                const int gb = cr.first_body + i;
                const BodyRec b = load_rec(bodies + gb);
                const DerivBody x = load_rec(db + gb);
                cptr<T> C = consts + b.cofs;
                T qi = 0, qdi = 0, qddi = 0;
                for (int a2 = 0; a2 < cr.n; a2++) {
                    const T g = C[kBodyConstFixed + a2];
                    qi += g * qs[cr.q_index + a2];
                    qdi += g * qds[cr.v_index + a2];
                    qddi += g * ydds[cr.v_index + a2];
                }
                T kp[24];
                if (b.parent >= 0 && b.parent == last_gb) {
                    // (along a chain the parent is the body before: its kinematics are still in registers -- without this every body
                    // waits for the store -> load round trip of its parent's slab row: ~10 k cycles per body in the phase profile)
#pragma unroll
                    for (int j = 0; j < 24; j++) kp[j] = last_kin[j];
                } else if (b.parent >= 0) {
                    const DerivBody xp = load_rec(db + b.parent);
                    R.ld(xp.kin_row, kp);
                } else {
#pragma unroll
                    for (int j = 0; j < 9; j++) kp[j] = (j % 4 == 0) ? T(1) : T(0);
#pragma unroll
                    for (int j = 9; j < 18; j++) kp[j] = 0;
#pragma unroll
                    for (int j = 0; j < 6; j++) kp[18 + j] = a0[j];
                }
                T Ep[9], pp[3], vp[6], ap[6], kin[24], anc[18];
#pragma unroll
                for (int j = 0; j < 9; j++) Ep[j] = kp[j];
#pragma unroll
                for (int j = 0; j < 3; j++) pp[j] = kp[9 + j];
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    vp[j] = kp[12 + j];
                    ap[j] = kp[18 + j];
                }
                T E[9], p3[3], v[6], a[6], S[6], Sd[6], Pdd[6];
                deriv_kin(C, false, qi, qdi, qddi, Ep, pp, vp, ap, E, p3, v, a, S, Sd, Pdd);
#pragma unroll
                for (int j = 0; j < 9; j++) kin[j] = E[j];
#pragma unroll
                for (int j = 0; j < 3; j++) kin[9 + j] = p3[j];
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    kin[12 + j] = v[j];
                    kin[18 + j] = a[j];
                    anc[j] = S[j];
                    anc[6 + j] = Sd[j];
                    anc[12 + j] = Pdd[j];
                }
                R.st(x.kin_row, kin);
                R.st(x.anc_row, anc);
                last_gb = gb;
#pragma unroll
                for (int j = 0; j < 24; j++) last_kin[j] = kin[j];
            }
        }
        DPROF_ADD(0);   // pass 1
        // ---- pass 2, leaf side first ----
        PartStore<T, sizeof(T) == 8 || kPartLdsF32> part;  // what the in-cluster roots of a cluster hand to the parent body: one read-modify-write per cluster, or
                     // no memory traffic at all along chains (DerivBody::carry_out: it stays here for the next cluster)
        part.lds = reinterpret_cast<T *>(deriv_smem) + lane;
        // fp32: LDS cache of the [S | Sd | Pdd] rows of the current root path, block l = the ancestor l bodies below the base
        // (plan.h, kDerivAncLevels; behind the composites when an experiment build keeps those in LDS too)
        T *anc_cache = reinterpret_cast<T *>(deriv_smem) + (kPartLdsF32 ? 63 * kWave : 0) + lane;
#pragma unroll
        for (int j = 0; j < 63; j++) part.set(j, T(0));
        for (int c = n_clusters - 1; c >= 0; c--) {
            const ClusterRec cr = load_rec(clusters + c);
            if (cr.kind == CK_FREE) {
                // base: E = 1, p = 0; S = 1, Sd = v x 1, Pd = 0, Pdd = a0 x 1; every pair of its columns counts as "ancestor or equal"
                const BodyRec b = load_rec(bodies + cr.first_body);
                const DerivBody x = load_rec(db + cr.first_body);
                cptr<T> Ib = consts + b.cofs + 12;
                T Ic[21], Bc[36], h[6];
#pragma unroll
                for (int j = 0; j < 21; j++) Ic[j] = Ib[j];
                symv(Ic, vb, h);
                body_B(Ic, vb, h, Bc);
                if (x.acc_row >= 0) {
                    T acc[57];
                    R.ld(x.acc_row, acc);
#pragma unroll
                    for (int j = 0; j < 21; j++) Ic[j] += acc[j];
#pragma unroll
                    for (int j = 0; j < 36; j++) Bc[j] += acc[21 + j];
                }
                // d tau_b / d q_b = Ic crm(a0), d tau_b / d qd_b = Bc + Ic crm(vb)
                T Mq[36];
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    T e[6], c1[6], c2[6], y1[6], y2[6];
#pragma unroll
                    for (int i = 0; i < 6; i++) e[i] = i == j ? T(1) : T(0);
                    crm(a0, e, c1);
                    crm(vb, e, c2);
                    symv(Ic, c1, y1);
                    symv(Ic, c2, y2);
#pragma unroll
                    for (int i = 0; i < 6; i++) {
                        Mq[6 * i + j] = y1[i];
                        Bc[6 * i + j] += y2[i];
                    }
                }
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    T row[6];
#pragma unroll
                    for (int j = 0; j < 6; j++) row[j] = Mq[6 * i + j];
                    if (rpy) rpy_columns(Rb, Tb, row);
                    if (live) {
#pragma unroll
                        for (int j = 0; j < 6; j++) {
                            put(Dqs, cr.v_index + i, cr.v_index + j, row[j]);
                            put(Dqds, cr.v_index + i, cr.v_index + j, Bc[6 * i + j]);
                            put_h(cr.v_index + i, cr.v_index + j, Ic[sidx(i, j)]);
                        }
                    }
                }
                DPROF_ADD(4);   // base cluster
                continue;
            }
            const int n = cr.n;
            // cluster-level descendant-side vectors and the contribution to the parent body's accumulator
            T T1[NMAX][6], T2[NMAX][6], T3[NMAX][6], T4[NMAX][6];
            T Cq[NMAX][NMAX], Cqd[NMAX][NMAX], Ch[NMAX][NMAX];
            const DerivBody xf = load_rec(db + cr.first_body);
            const int first_i = xf.carry_body >= 0 ? xf.carry_body - cr.first_body : -1;
            if (first_i < 0) {
#pragma unroll
                for (int j = 0; j < 63; j++) part.set(j, T(0));
            }
#pragma unroll
            for (int a2 = 0; a2 < NMAX; a2++) {
#pragma unroll
                for (int j = 0; j < 6; j++) T1[a2][j] = T2[a2][j] = T3[a2][j] = T4[a2][j] = 0;
#pragma unroll
                for (int b2 = 0; b2 < NMAX; b2++) Cq[a2][b2] = Cqd[a2][b2] = Ch[a2][b2] = 0;
            }
            // (the body that receives carried composites comes first, the others last body first)
            for (int step = first_i >= 0 ? -1 : 0; step < cr.k; step++) {
                int i = first_i;
                if (step >= 0) {
                    i = cr.k - 1 - step;
                    if (i == first_i) continue;
                }
                const int gb = cr.first_body + i;
                const BodyRec b = load_rec(bodies + gb);
                const DerivBody x = load_rec(db + gb);
                cptr<T> C = consts + b.cofs;
                T Gi[NMAX];
#pragma unroll
                for (int a2 = 0; a2 < NMAX; a2++) Gi[a2] = a2 < n ? C[kBodyConstFixed + a2] : T(0);
                T E[9], p3[3], v[6], a[6], S[6], Sd[6], Pdd[6];
                if (x.kin_row >= 0) {
                    T kin[24], anc[18];
                    R.ld(x.kin_row, kin);
                    R.ld(x.anc_row, anc);
#pragma unroll
                    for (int j = 0; j < 9; j++) E[j] = kin[j];
#pragma unroll
                    for (int j = 0; j < 3; j++) p3[j] = kin[9 + j];
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        v[j] = kin[12 + j];
                        a[j] = kin[18 + j];
                        S[j] = anc[j];
                        Sd[j] = anc[6 + j];
                        Pdd[j] = anc[12 + j];
                    }
                } else {
                    T qi = 0, qdi = 0, qddi = 0;
#pragma unroll
                    for (int a2 = 0; a2 < NMAX; a2++)
                        if (a2 < n) {
                            qi += Gi[a2] * qs[cr.q_index + a2];
                            qdi += Gi[a2] * qds[cr.v_index + a2];
                            qddi += Gi[a2] * ydds[cr.v_index + a2];
                        }
                    T kp[24];
                    if (b.parent >= 0) {
                        const DerivBody xp = load_rec(db + b.parent);
                        R.ld(xp.kin_row, kp);
                    } else {
#pragma unroll
                        for (int j = 0; j < 9; j++) kp[j] = (j % 4 == 0) ? T(1) : T(0);
#pragma unroll
                        for (int j = 9; j < 18; j++) kp[j] = 0;
#pragma unroll
                        for (int j = 0; j < 6; j++) kp[18 + j] = a0[j];
                    }
                    T Ep[9], pp[3], vp[6], ap[6];
#pragma unroll
                    for (int j = 0; j < 9; j++) Ep[j] = kp[j];
#pragma unroll
                    for (int j = 0; j < 3; j++) pp[j] = kp[9 + j];
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        vp[j] = kp[12 + j];
                        ap[j] = kp[18 + j];
                    }
                    deriv_kin(C, b.axisym != 0, qi, qdi, qddi, Ep, pp, vp, ap, E, p3, v, a, S, Sd, Pdd);
                }
                // body inertia, force and B in F; composites
                T Ic[21], Bc[36], Fc[6], h[6];
                congruence_rigid(E, p3, C + 12, Ic);
                symv(Ic, v, h);
                {
                    T Ia[6], vh[6];
                    symv(Ic, a, Ia);
                    crf(v, h, vh);
#pragma unroll
                    for (int j = 0; j < 6; j++) Fc[j] = Ia[j] + vh[j];
                }
                body_B(Ic, v, h, Bc);
                if (step < 0) {  // the composites of the subtree arrived in registers
#pragma unroll
                    for (int j = 0; j < 21; j++) Ic[j] += part.get(j);
#pragma unroll
                    for (int j = 0; j < 36; j++) Bc[j] += part.get(21 + j);
#pragma unroll
                    for (int j = 0; j < 6; j++) Fc[j] += part.get(57 + j);
#pragma unroll
                    for (int j = 0; j < 63; j++) part.set(j, T(0));
                } else if (x.acc_row >= 0) {
                    T acc[63];
                    R.ld(x.acc_row, acc);
#pragma unroll
                    for (int j = 0; j < 21; j++) Ic[j] += acc[j];
#pragma unroll
                    for (int j = 0; j < 36; j++) Bc[j] += acc[21 + j];
#pragma unroll
                    for (int j = 0; j < 6; j++) Fc[j] += acc[57 + j];
                }
                DPROF_ADD(1);   // body: records, kinematics (loads or recomputed), composites
                // descendant-side vectors of this joint (Pd = Sd for a revolute joint)
                T t1[6], t2[6], t3[6], t4[6];
                mtv6(Bc, S, t1);
                symv(Ic, S, t2);
                {
                    T u1[6], u2[6], u3[6], u4[6];
                    mv6(Bc, S, u1);
                    symv(Ic, Sd, u2);
                    crf(S, Fc, u3);
                    mv6(Bc, Sd, u4);
                    T u5[6];
                    symv(Ic, Pdd, u5);
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        t3[j] = u1[j] + 2 * u2[j];
                        t4[j] = u3[j] + u4[j] + u5[j];
                    }
                }
                // the joint with itself
                {
                    const T sq = dot6(Sd, t1) + dot6(Pdd, t2), sqd = dot6(S, t1) + 2 * dot6(Sd, t2), sh = dot6(S, t2);
#pragma unroll
                    for (int a2 = 0; a2 < NMAX; a2++)
#pragma unroll
                        for (int b2 = 0; b2 < NMAX; b2++) {
                            Cq[a2][b2] += Gi[a2] * Gi[b2] * sq;
                            Cqd[a2][b2] += Gi[a2] * Gi[b2] * sqd;
                            Ch[a2][b2] += Gi[a2] * Gi[b2] * sh;
                        }
                }
                // in-cluster ancestors
                int l = b.lam;
                while (l >= 0) {
                    const BodyRec bl = load_rec(bodies + l);
                    const DerivBody xl = load_rec(db + l);
                    cptr<T> Cl = consts + bl.cofs;
                    T anc[18], Sl[6], Sdl[6], Pddl[6], Gl[NMAX];
                    R.ld(xl.anc_row, anc);
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        Sl[j] = anc[j];
                        Sdl[j] = anc[6 + j];
                        Pddl[j] = anc[12 + j];
                    }
#pragma unroll
                    for (int a2 = 0; a2 < NMAX; a2++) Gl[a2] = a2 < n ? Cl[kBodyConstFixed + a2] : T(0);
                    const T kq = dot6(Sdl, t1) + dot6(Pddl, t2), kqd = dot6(Sl, t1) + 2 * dot6(Sdl, t2);  // (k = this body, j = l)
                    const T lq = dot6(Sl, t4), lqd = dot6(Sl, t3);                                          // (k = l, j = this body)
                    const T lh = dot6(Sl, t2);
#pragma unroll
                    for (int a2 = 0; a2 < NMAX; a2++)
#pragma unroll
                        for (int b2 = 0; b2 < NMAX; b2++) {
                            Cq[a2][b2] += Gi[a2] * Gl[b2] * kq + Gl[a2] * Gi[b2] * lq;
                            Cqd[a2][b2] += Gi[a2] * Gl[b2] * kqd + Gl[a2] * Gi[b2] * lqd;
                            Ch[a2][b2] += (Gi[a2] * Gl[b2] + Gl[a2] * Gi[b2]) * lh;
                        }
                    l = bl.lam;
                }
#pragma unroll
                for (int a2 = 0; a2 < NMAX; a2++)
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        T1[a2][j] += Gi[a2] * t1[j];
                        T2[a2][j] += Gi[a2] * t2[j];
                        T3[a2][j] += Gi[a2] * t3[j];
                        T4[a2][j] += Gi[a2] * t4[j];
                    }
                // composites to the tree parent: in-cluster parents through their accumulator rows (the first writer stores,
                // the others add), the parent body of the cluster through `part`
                if (b.lam >= 0) {
                    const DerivBody xl = load_rec(db + b.lam);
                    if (!x.acc_first) {
                        T acc[63];
                        R.ld(xl.acc_row, acc);
#pragma unroll
                        for (int j = 0; j < 21; j++) Ic[j] += acc[j];
#pragma unroll
                        for (int j = 0; j < 36; j++) Bc[j] += acc[21 + j];
#pragma unroll
                        for (int j = 0; j < 6; j++) Fc[j] += acc[57 + j];
                    }
                    R.st(xl.acc_row, Ic);
                    R.st(xl.acc_row + 21, Bc);
                    R.st(xl.acc_row + 57, Fc);
                } else {
#pragma unroll
                    for (int j = 0; j < 21; j++) part.set(j, part.get(j) + Ic[j]);
#pragma unroll
                    for (int j = 0; j < 36; j++) part.set(21 + j, part.get(21 + j) + Bc[j]);
#pragma unroll
                    for (int j = 0; j < 6; j++) part.set(57 + j, part.get(57 + j) + Fc[j]);
                }
            }
            DPROF_ADD(2);   // body: joint terms, in-cluster ancestors, hand-over
            if (cr.parent_body >= 0 && !xf.carry_out) {
                const DerivBody xp = load_rec(db + cr.parent_body);
                T out[63];
#pragma unroll
                for (int j = 0; j < 63; j++) out[j] = part.get(j);
                if (!xf.cluster_acc_first) {
                    T acc[63];
                    R.ld(xp.acc_row, acc);
#pragma unroll
                    for (int j = 0; j < 63; j++) out[j] += acc[j];
                }
                R.st(xp.acc_row, out);
            }
            if (live) {
#pragma unroll
                for (int a2 = 0; a2 < NMAX; a2++)
#pragma unroll
                    for (int b2 = 0; b2 < NMAX; b2++)
                        if (a2 < n && b2 < n) {
                            put(Dqs, cr.v_index + a2, cr.v_index + b2, Cq[a2][b2]);
                            put(Dqds, cr.v_index + a2, cr.v_index + b2, Cqd[a2][b2]);
                            put_h(cr.v_index + a2, cr.v_index + b2, Ch[a2][b2]);
                        }
            }
            DPROF_ADD(3);   // accumulator row of the parent body, the cluster's own entries stored
            // ---- up the ancestors outside the cluster, block by block ----
#ifdef GRBDA_EXP_NO_WALK
            int j = -1;
#else
            int j = cr.parent_body;
#endif
            while (j >= 0) {
                const DerivBody xj = load_rec(db + j);
                const ClusterRec cd = load_rec(clusters + xj.cluster);
                if (cd.kind == CK_FREE) {
#pragma unroll
                    for (int a2 = 0; a2 < NMAX; a2++)
                        if (a2 < n) {
                            T w1[6], w2[6];
                            crf(a0, T2[a2], w1);   // (a0 x e_k) . t = -(a0 x* t)_k
                            crf(vb, T2[a2], w2);
#pragma unroll
                            for (int k6 = 0; k6 < 6; k6++) w1[k6] = -w1[k6];
                            if (rpy) rpy_columns(Rb, Tb, w1);
                            if (live) {
#pragma unroll
                                for (int k6 = 0; k6 < 6; k6++) {
                                    put(Dqs, cr.v_index + a2, cd.v_index + k6, w1[k6]);
                                    put(Dqds, cr.v_index + a2, cd.v_index + k6, T1[a2][k6] - w2[k6]);
                                    put(Dqs, cd.v_index + k6, cr.v_index + a2, T4[a2][k6]);
                                    put(Dqds, cd.v_index + k6, cr.v_index + a2, T3[a2][k6]);
                                    put_h(cr.v_index + a2, cd.v_index + k6, T2[a2][k6]);
                                }
                            }
                        }
                    break;
                }
                T Bq[NMAX][NMAX], Bqd[NMAX][NMAX];   // [c][d]
                T Uq[NMAX][NMAX], Uqd[NMAX][NMAX];   // [d][c]
                T Bh[NMAX][NMAX];
#pragma unroll
                for (int a2 = 0; a2 < NMAX; a2++)
#pragma unroll
                    for (int b2 = 0; b2 < NMAX; b2++) Bq[a2][b2] = Bqd[a2][b2] = Uq[a2][b2] = Uqd[a2][b2] = Bh[a2][b2] = 0;
                int jj = j, next = -1;
                for (;;) {
                    const BodyRec bb = load_rec(bodies + jj);
                    const DerivBody xb = load_rec(db + jj);
                    cptr<T> Cj = consts + bb.cofs;
                    T anc[18], Sj[6], Sdj[6], Pddj[6];
                    if constexpr (sizeof(T) == 4) {
                        // (wave-uniform: the block's validity was worked out by the plan compiler, DerivBody::walk_resident)
                        if (xb.anc_lds >= 0 && xb.anc_lds < kAncLevelsLds) {
                            T *blk = anc_cache + xb.anc_lds * 18 * kWave;
                            if ((xf.walk_resident >> xb.anc_lds) & 1) {
#pragma unroll
                                for (int i2 = 0; i2 < 18; i2++) anc[i2] = blk[i2 * kWave];
                            } else {
                                R.ld(xb.anc_row, anc);
#pragma unroll
                                for (int i2 = 0; i2 < 18; i2++) blk[i2 * kWave] = anc[i2];
                            }
                        } else {
                            R.ld(xb.anc_row, anc);
                        }
                    } else {
                        R.ld(xb.anc_row, anc);
                    }
#pragma unroll
                    for (int i2 = 0; i2 < 6; i2++) {
                        Sj[i2] = anc[i2];
                        Sdj[i2] = anc[6 + i2];
                        Pddj[i2] = anc[12 + i2];
                    }
#pragma unroll
                    for (int a2 = 0; a2 < NMAX; a2++) {
                        const T eq = dot6(Sdj, T1[a2]) + dot6(Pddj, T2[a2]), eqd = dot6(Sj, T1[a2]) + 2 * dot6(Sdj, T2[a2]);
                        const T uq = dot6(Sj, T4[a2]), uqd = dot6(Sj, T3[a2]), hh = dot6(Sj, T2[a2]);
#pragma unroll
                        for (int b2 = 0; b2 < NMAX; b2++)
                            if (b2 < cd.n) {
                                const T g = Cj[kBodyConstFixed + b2];
                                Bq[a2][b2] += eq * g;
                                Bqd[a2][b2] += eqd * g;
                                Uq[b2][a2] += uq * g;
                                Uqd[b2][a2] += uqd * g;
                                Bh[a2][b2] += hh * g;
                            }
                    }
                    next = bb.parent;
                    if (bb.lam < 0) break;  // left the cluster
                    jj = bb.lam;
                }
                if (live) {
#pragma unroll
                    for (int a2 = 0; a2 < NMAX; a2++)
#pragma unroll
                        for (int b2 = 0; b2 < NMAX; b2++)
                            if (a2 < n && b2 < cd.n) {
                                put(Dqs, cr.v_index + a2, cd.v_index + b2, Bq[a2][b2]);
                                put(Dqds, cr.v_index + a2, cd.v_index + b2, Bqd[a2][b2]);
                                put(Dqs, cd.v_index + b2, cr.v_index + a2, Uq[b2][a2]);
                                put(Dqds, cd.v_index + b2, cr.v_index + a2, Uqd[b2][a2]);
                                put_h(cr.v_index + a2, cd.v_index + b2, Bh[a2][b2]);
                            }
                }
                j = next;
                DPROF_ADD(5);   // one ancestor cluster of the walk (rows, dot products, stores)
            }
            DPROF_ADD(6);   // walk: the base's columns, loop exit
        }
        DPROF_ADD(7);
        // entries between clusters on different branches are structural zeros: never written, and never read by the solve
        // (DerivProgram::related)
    }
    DPROF_END();
}

template <class T, int IL>
static hipError_t launch_rnea_deriv_il(const DevPlan<T> &P, const DerivBody *db, int n_clusters, int n_rows, int n_max, const T *q,
                                       const T *qd, const T *ydd, T *Dq, T *Dqd, T *H, size_t B, T *scratch, int grid, hipStream_t stream)
{
    const size_t part_lds = ((sizeof(T) == 8 || kPartLdsF32) ? 63 * kWave * sizeof(T) : 0)   // PartStore
                            + (sizeof(T) == 4 ? kAncLevelsLds * 18 * kWave * sizeof(T) : 0)  // fp32: ancestor-row cache
                            + ((sizeof(T) == 4 && kStageInputsF32) ? static_cast<size_t>(kWave) * (P.nq + 2 * P.nv) * sizeof(T) : 0);  // staged inputs
    if (n_max <= 1)
        hipLaunchKernelGGL((rnea_deriv_kernel<T, 1, IL>), dim3(grid), dim3(kWave), part_lds, stream, P, db, n_clusters, n_rows, q, qd, ydd, Dq,
                           Dqd, H, B, scratch);
    else if (n_max <= 2)
        hipLaunchKernelGGL((rnea_deriv_kernel<T, 2, IL>), dim3(grid), dim3(kWave), part_lds, stream, P, db, n_clusters, n_rows, q, qd, ydd, Dq,
                           Dqd, H, B, scratch);
    else
        hipLaunchKernelGGL((rnea_deriv_kernel<T, kMaxClusterDof, IL>), dim3(grid), dim3(kWave), part_lds, stream, P, db, n_clusters, n_rows, q,
                           qd, ydd, Dq, Dqd, H, B, scratch);
    return hipGetLastError();
}
template <class T>
hipError_t launch_rnea_deriv(const DevPlan<T> &P, const DerivBody *db, int n_clusters, int n_rows, int n_max, const T *q, const T *qd,
                             const T *ydd, T *Dq, T *Dqd, T *H, size_t B, T *scratch, int grid, hipStream_t stream, int interleave)
{
    // (f64: the minv route of capi.cpp -- minv_kernels.hip reads the interleaved blocks through LDS)
    if (interleave == kDerivGroup)
        return launch_rnea_deriv_il<T, kDerivGroup>(P, db, n_clusters, n_rows, n_max, q, qd, ydd, Dq, Dqd, H, B, scratch, grid, stream);
    if (interleave == kWave && n_max <= 1) {
        // tile-interleaved results [tile][entry][lane]: what a one-state-per-lane consumer reads as coalesced rows (the
        // spanning-tree pass of manifold_kernels.hip; single-body clusters only)
        const size_t part_lds = ((sizeof(T) == 8 || kPartLdsF32) ? 63 * kWave * sizeof(T) : 0) + (sizeof(T) == 4 ? kAncLevelsLds * 18 * kWave * sizeof(T) : 0) +
                                ((sizeof(T) == 4 && kStageInputsF32) ? static_cast<size_t>(kWave) * (P.nq + 2 * P.nv) * sizeof(T) : 0);
        hipLaunchKernelGGL((rnea_deriv_kernel<T, 1, kWave>), dim3(grid), dim3(kWave), part_lds, stream, P, db, n_clusters, n_rows, q, qd, ydd, Dq,
                           Dqd, H, B, scratch);
        return hipGetLastError();
    }
    if (interleave != 1) return hipErrorInvalidValue;
    return launch_rnea_deriv_il<T, 1>(P, db, n_clusters, n_rows, n_max, q, qd, ydd, Dq, Dqd, H, B, scratch, grid, stream);
}
template hipError_t launch_rnea_deriv<float>(const DevPlan<float> &, const DerivBody *, int, int, int, const float *, const float *,
                                             const float *, float *, float *, float *, size_t, float *, int, hipStream_t, int);
template hipError_t launch_rnea_deriv<double>(const DevPlan<double> &, const DerivBody *, int, int, int, const double *, const double *,
                                              const double *, double *, double *, double *, size_t, double *, int, hipStream_t, int);

// ---------------------------------------------------------------------------------------------------------------
// Batched SPD solve, one state per wavefront.  Lane i holds row i of H, then of its Cholesky factor L (left-looking by
// columns: the entry L[k][m] another row needs comes from lane k by v_readlane, a scalar operand); each finished column
// of L is also written to LDS as a row of L^T.  For the triangular solves every lane carries one or two right-hand-side
// columns in registers and reads the factor row by row from LDS, every lane the same address (a 16-byte broadcast read,
// on the LDS port rather than the VALU's): forward substitution by columns of L, backward by rows of L^T -- both are
// rows of the stored L^T.  With two columns per lane the f32 updates are packed (v_pk_fma_f32).  NV is the compile-time
// size the loops are unrolled for (nv <= NV <= 64; H is padded with the identity).  Right-hand sides: the nv columns of
// P1, of P2 (rnea_deriv_kernel's packed layout) and of the identity (H^-1), each optional; results, in plain row-major
// layout, are scaled by -1 for P1 / P2 (-H^-1 D).  Hinv may be H: the factor is in registers / LDS before H^-1 is stored.
// `related` (DerivProgram::related, may be null): the entries that are not structural zeros; the others are not used,
// whatever the arrays hold.  Global traffic: every matrix read / written once.  TIO: array element type; TC: arithmetic.
// ---------------------------------------------------------------------------------------------------------------
// f32 solves run on the matrix cores (spd_mfma_kernel) when a workgroup's tiles and right-hand sides fit the LDS of a CU (nv <= 48
// with two right-hand sides) and GRBDA_SOLVE_VALU=1 does not keep the triangular solves (A/B runs)
static size_t spd_mfma_lds_bytes(int nv, int n_rhs)
{
    const int nvb = nv <= 16 ? 16 : (nv <= 24 ? 24 : (nv <= 32 ? 32 : (nv <= 40 ? 40 : (nv <= 48 ? 48 : 64))));
    const int nt = (nvb + 15) / 16, ws = nt == 1 ? 16 : (nt <= 3 ? 48 : 80);
    // per WORKGROUP of kDerivGroup wavefronts; the H block is staged where the right-hand sides go
    return static_cast<size_t>(kDerivGroup) * (static_cast<size_t>(nvb) * ws + static_cast<size_t>(n_rhs > 1 ? n_rhs : 1) * nv * nv) * 4;
}
// wavefronts per SIMD (= workgroups per CU) of the matrix-core solve for nv <= 24: the kernel is bound by the issue of mostly dependent
// instruction chains, which a third wavefront fills in -- MIT Humanoid all three matrices 3.03 -> 2.75 ms per 262 144 states, Mini Cheetah
// 0.675 -> 0.614 per 65 536; a fourth costs spills (3.03); nv > 24 needs the registers (246 .. 256) and the LDS of two
// (experiment builds: -DGRBDA_EXP_MF_WAVES24=2|4)
#ifndef GRBDA_EXP_MF_WAVES24
#define GRBDA_EXP_MF_WAVES24 3
#endif
int spd_mfma_workgroups_per_cu(int nv) { return nv <= 24 ? GRBDA_EXP_MF_WAVES24 : 2; }  // (= wavefronts per SIMD: workgroups of four)
bool spd_solve_on_mfma(size_t elem, int nv, int n_rhs)
{
    static const bool valu = [] { const char *e = std::getenv("GRBDA_SOLVE_VALU"); return e && std::atoi(e) != 0; }();
    return elem == 4 && !valu && nv <= kWave && spd_mfma_lds_bytes(nv, n_rhs) <= 160u * 1024u;
}
size_t spd_solve_lds_bytes(int nv, size_t elem, int n_rhs);

__device__ __forceinline__ float lane_value(float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); }
__device__ __forceinline__ double lane_value(double x, int l)
{
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// States whose joint-space inertia is not positive definite to working precision (a pivot of the factorisation <= 0 or not finite:
// massless chains, a singular pose of an implicit cluster): their results are NaN / Inf, and the solve kernels count them here so
// that a caller can ASK (grbda_spd_bad_pivots, include/grbda_hip.h) instead of scanning nv^2 numbers per state.
__device__ unsigned long long grbda_spd_bad_count = 0;
unsigned long long *spd_bad_count_address()
{
    void *p = nullptr;
    return hipGetSymbolAddress(&p, HIP_SYMBOL(grbda_spd_bad_count)) == hipSuccess ? static_cast<unsigned long long *>(p) : nullptr;
}
hipError_t spd_bad_pivots(unsigned long long *count, int reset)
{
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return e;
    e = hipMemcpyFromSymbol(count, HIP_SYMBOL(grbda_spd_bad_count), sizeof *count);
    if (e != hipSuccess || !reset) return e;
    const unsigned long long z = 0;
    return hipMemcpyToSymbol(HIP_SYMBOL(grbda_spd_bad_count), &z, sizeof z);
}

// 1 / sqrt(d): the hardware estimate refined by Newton steps (y <- y (1.5 - 0.5 d y^2)) to working precision, a fraction
// of the instructions of the IEEE division and square root
__device__ __forceinline__ float inv_sqrt(float d)
{
    const float y = __builtin_amdgcn_rsqf(d);
    return y * (1.5f - 0.5f * d * y * y);
}
__device__ __forceinline__ double inv_sqrt(double d)
{
    double y = __builtin_amdgcn_rsq(d);
    y = y * (1.5 - 0.5 * d * y * y);
    return y * (1.5 - 0.5 * d * y * y);
}

template <class TIO, class TC, int NV, int KC>
__global__ __launch_bounds__(kWave)
__attribute__((amdgpu_waves_per_eu((KC == 2 || (sizeof(TC) == 8 && NV <= 48)) ? 2 : 1, KC == 2 ? 2 : 8)))
void spd_solve_kernel(const TIO *H, int h_packed, const TIO *P1, const TIO *P2, TIO *Hinv, TIO *X1, TIO *X2,
                      const uint64_t *__restrict__ related, int nv, size_t B)
{
    constexpr int V = 16 / (int)sizeof(TC);
    typedef TC Vec __attribute__((ext_vector_type(V)));
    typedef TC XV __attribute__((ext_vector_type(KC)));
    static_assert(NV % V == 0, "rows of the factor are read as 16-byte vectors");
    __shared__ __attribute__((aligned(16))) TC Lt[NV * NV];  // Lt[k][i] = L[i][k], i >= k; Lt[k][k] = 1 / L[k][k]
    const int lane = threadIdx.x;
    const size_t nn = (size_t)nv * nv;
    // row k of Lt from entry `from` on, every lane the same address (an LDS broadcast, no bank conflicts)
    auto factor_row = [&](int k, int from, TC(&l)[NV]) {
#pragma unroll
        for (int i = (from / V) * V; i < NV; i += V) {
            const Vec v = *reinterpret_cast<const Vec *>(&Lt[k * NV + i]);
#pragma unroll
            for (int e = 0; e < V; e++) l[i + e] = v[e];
        }
    };
    // a lane's own run of n consecutive entries at `row` (n wave-uniform): 16-byte loads where they stay inside the run
    auto read_row = [&](const TIO *row, int n, TIO(&out)[NV]) {
        constexpr int W = 16 / (int)sizeof(TIO);
        typedef TIO VIO __attribute__((ext_vector_type(W)));
#pragma unroll
        for (int i = 0; i < NV; i += W) {
            if (i + W <= n) {
                VIO v;
                __builtin_memcpy(&v, row + i, sizeof v);
#pragma unroll
                for (int e = 0; e < W; e++) out[i + e] = v[e];
            } else {
#pragma unroll
                for (int e = 0; e < W; e++) out[i + e] = (i + e < n) ? row[i + e] : TIO(0);
            }
        }
    };
    const uint64_t rel_mine = related ? related[lane < nv ? lane : 0] : ~uint64_t(0);
    for (size_t s = blockIdx.x; s < B; s += gridDim.x) {
        {
            TC Lr[NV];  // row `lane` of H, then of L; the diagonal entry holds 1 / L[k][k]
            // (only the lower triangle of H is read, and of that only the entries between coordinates on one root path;
            // h_packed: rows of the lower triangle back to back, as rnea_deriv_kernel writes them)
            const size_t lrow = lane < nv ? lane : 0;
            TIO hrow[NV];
            read_row(H + s * nn + (h_packed ? lrow * (lrow + 1) / 2 : lrow * nv), nv, hrow);
#pragma unroll
            for (int j = 0; j < NV; j++)
                Lr[j] = (lane < nv && j < nv) ? ((j <= lane && ((rel_mine >> j) & 1)) ? (TC)hrow[j] : TC(0)) : (lane == j ? TC(1) : TC(0));
            __syncthreads();  // the previous state's solves are done with Lt
            bool bad = false;
#pragma unroll
            for (int k = 0; k < NV; k++) {
                TC sum = Lr[k];
#pragma unroll
                for (int m2 = 0; m2 < k; m2++) sum -= Lr[m2] * lane_value(Lr[m2], k);
                const TC d = lane_value(sum, k);
                bad = bad || !(d > TC(0)) || !(d < TC(3e38));
                const TC r = inv_sqrt(d);
                Lr[k] = lane == k ? r : sum * r;
                if (lane >= k && lane < NV) Lt[k * NV + lane] = Lr[k];
            }
            if (bad && lane == 0) atomicAdd(&grbda_spd_bad_count, 1ull);
            __syncthreads();
        }
        // right-hand sides: KC columns per lane and pass; forward substitution by columns of L, backward by rows of L^T,
        // both of which are rows of Lt
        const int n1 = P1 ? nv : 0, n2 = P2 ? nv : 0, n3 = Hinv ? nv : 0;
        for (int c0 = 0; c0 < n1 + n2 + n3; c0 += KC * kWave) {
            asm volatile("" ::: "memory");  // the factor is re-read from LDS every pass, not kept in NV^2/2 registers
            const TIO *src[KC];
            TIO *dst[KC];
            int jc[KC];
            TC scale[KC];
            XV x[NV];  // x[i][u]: entry i of this lane's u-th column (a register pair when KC == 2: packed f32 arithmetic)
#pragma unroll
            for (int u = 0; u < KC; u++) {
                const int col = c0 + u * kWave + lane;
                src[u] = nullptr;
                dst[u] = nullptr;
                jc[u] = 0;
                scale[u] = 1;
                if (col < n1) { src[u] = P1 + s * nn; dst[u] = X1 + s * nn; jc[u] = col; scale[u] = -1; }
                else if (col < n1 + n2) { src[u] = P2 + s * nn; dst[u] = X2 + s * nn; jc[u] = col - n1; scale[u] = -1; }
                else if (col < n1 + n2 + n3) { dst[u] = Hinv + s * nn; jc[u] = col - n1 - n2; }
                // column c of a packed right-hand side (rnea_deriv_kernel): the entries on and below the diagonal sit at
                // r^2 + c of the runs r >= c (across lanes: consecutive addresses), the ones above it are the second half of
                // run c, c^2 + c + 1 + r.  Structural zeros are never written, and not used here.
                const uint64_t rel = related ? related[jc[u]] : ~uint64_t(0);
                if (src[u]) {
                    TIO urow[NV];
                    read_row(src[u] + (size_t)jc[u] * jc[u] + jc[u] + 1, nv - 1, urow);
#pragma unroll
                    for (int i = 0; i < NV; i++) {
                        const size_t r = i < nv ? i : 0;
                        const TIO v = src[u][r * r + (jc[u] <= (int)r ? jc[u] : 0)];
                        x[i][u] = (i < nv && ((rel >> i) & 1)) ? (TC)(i < jc[u] ? urow[i] : v) : TC(0);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < NV; i++) x[i][u] = (dst[u] && i == jc[u]) ? TC(1) : TC(0);
                }
            }
            // (the row after the current one is in flight while this one is used; the clobbers and scheduling barriers keep
            // the compiler from hoisting every LDS read of the unrolled loop to the top, which spills)
            // (fp64: no row in flight -- the registers it would take are what keeps a second wavefront off the SIMD, and that
            // wavefront hides the LDS latency better)
            constexpr bool AHEAD = sizeof(TC) == 4;
            TC l[NV], ln[AHEAD ? NV : 1];
            if constexpr (AHEAD) factor_row(0, 0, l);
#pragma unroll
            for (int m2 = 0; m2 < NV; m2++) {
                if constexpr (AHEAD) factor_row(m2 + 1 < NV ? m2 + 1 : NV - 1, m2 + 1 < NV ? m2 + 1 : NV - 1, ln);
                else factor_row(m2, m2, l);
                x[m2] *= l[m2];
#pragma unroll
                for (int i = m2 + 1; i < NV; i++) x[i] -= l[i] * x[m2];
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (AHEAD) {
#pragma unroll
                    for (int i = 0; i < NV; i++) l[i] = ln[i];
                }
            }
#pragma unroll
            for (int i = NV - 1; i >= 0; i--) {
                if constexpr (AHEAD) {
                    if (i > 0) factor_row(i - 1, i - 1, ln);
                } else {
                    factor_row(i, i, l);
                }
                XV acc = x[i];
#pragma unroll
                for (int m2 = i + 1; m2 < NV; m2++) acc -= l[m2] * x[m2];
                x[i] = acc * l[i];
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (AHEAD) {
#pragma unroll
                    for (int j = 0; j < NV; j++) l[j] = ln[j];
                }
            }
#pragma unroll
            for (int u = 0; u < KC; u++)
                if (dst[u]) {
#pragma unroll
                    for (int i = 0; i < NV; i++)
                        if (i < nv) dst[u][(size_t)i * nv + jc[u]] = (TIO)(scale[u] * x[i][u]);
                }
            // a use of the solution outside the conditional stores: without it the compiler sinks the whole substitution
            // into the store block, below the LDS reads, and every factor row stays live (spills)
#pragma unroll
            for (int i = 0; i < NV; i++)
#pragma unroll
                for (int u = 0; u < KC; u++) asm volatile("" ::"v"(x[i][u]));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same solve on the matrix cores (f32).  A workgroup is kDerivGroup = 4 wavefronts and takes one GROUP of four states at
// a time, one state per wavefront; the group's inputs are one contiguous block of memory -- [entry][4] when the derivative
// recursion wrote them (interleaved workspace, see rnea_deriv_kernel), [state][entry] when H comes from the CRBA kernel or
// the caller -- which the four wavefronts copy into LDS together, 16 bytes per lane and instruction, fully coalesced.
// Per wavefront:
//   1. Cholesky factor of H as above (row i in lane i, v_readlane), rows of L^T to LDS;
//   2. W = L^-1 by forward substitution on the identity (column j in lane j), written to LDS;
//   3. H^-1 = W^T W              v_mfma_f32_16x16x4_f32, NT x NT tiles of 16 x 16, W lower triangular: tile (mt, nt) only
//                                sums over rows >= 16 max(mt, nt);
//   4. [X1 | X2] = -H^-1 [P1 | P2]   the same instruction; A fragments from the H^-1 tile in LDS (symmetric: read by rows),
//                                B fragments gathered from the group's PACKED right-hand sides (rnea_deriv_kernel's runs) in
//                                LDS; structural zeros (DerivProgram::related) and the padding are masked to 0 at the gather.
// The group's H block is staged in the LDS region of the right-hand sides first (every lane takes its row from there), then
// the right-hand sides are copied over it while steps 1 - 3 run.  2 nv^2 (nv + n_rhs) flops per state go through 16 x 16 x 4
// tiles (JVRC-1: 38 + 150 MFMAs) instead of ~5 000 VALU instructions of triangular solves; what stays on the VALU is the
// factorisation and the inversion of the factor (~nv^2 FMAs + nv^2 / 2 v_readlane).  Fragment maps
// (cdna_hip_programming.md 3): A[l & 15][l >> 4], B[l >> 4][l & 15], D: column l & 15, rows 4 (l >> 4) + 0..3.  LDS rows of
// the tile have a stride of WS floats with WS = 16 (mod 32): the four 16-lane groups of a fragment read (rows 4k .. 4k + 3)
// fall into different banks.  LDS per workgroup: 4 NVV WS floats of tiles + 4 n_rhs nv^2 of right-hand sides (JVRC-1:
// 30.7 + 46.2 KB, two workgroups per CU).
// ---------------------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifdef GRBDA_EXP_MF_PROF
__device__ unsigned long long mf_prof[8];
#define MF_STAMP(i) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); prof_acc[i] += now_ - prof_t; prof_t = now_; }
#else
#define MF_STAMP(i)
#endif

template <int NVV>
__global__ __launch_bounds__(kWave * kDerivGroup)
    __attribute__((amdgpu_waves_per_eu(NVV > 48 ? 1 : (NVV <= 24 ? GRBDA_EXP_MF_WAVES24 : 2), NVV <= 24 ? GRBDA_EXP_MF_WAVES24 : 2)))
void spd_mfma_kernel(const float *H, int h_packed, int h_il, const float *P1, const float *P2, int p_il, float *Hinv, float *X1, float *X2,
                     const uint64_t *__restrict__ related, int nv, size_t B)
{
    constexpr int NT = (NVV + 15) / 16;               // row / column tiles of H^-1
    constexpr int WS = NT == 1 ? 16 : (NT <= 3 ? 48 : 80);
    constexpr int NCT = (2 * NVV + 15) / 16;          // column tiles of [P1 | P2]
    constexpr int KS = NVV / 4;                       // k steps
    constexpr int G = kDerivGroup;
    static_assert(NVV % 8 == 0 && NT * 16 <= WS, "sizes");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
    const int nn = nv * nv;
    float *A = reinterpret_cast<float *>(grbda_smem) + wave * (NVV * WS);  // this wavefront's tile: rows of L^T, then W, then H^-1
    float *Pg = reinterpret_cast<float *>(grbda_smem) + G * (NVV * WS);   // the group's H block, then its right-hand sides
    const float *src[2] = {P1 ? P1 : P2, P1 ? P2 : nullptr};
    float *const dst0 = P1 ? X1 : X2, *const dst1 = P1 ? X2 : nullptr;
    const int n_mat = (P1 ? 1 : 0) + (P2 ? 1 : 0);
    const int n_cols = n_mat * nv, nct = (n_cols + 15) / 16;
    // element e of this wavefront's state inside a group block: interleaved [entry][G] or state-major [state][entry]
    const int h_es = h_il == G ? G : 1, h_ss = h_il == G ? 1 : nn;
    const int p_es = p_il == G ? G : 1, p_ss = p_il == G ? 1 : nn;
    auto factor_row = [&](int k, int from, float(&l)[NVV]) {
#pragma unroll
        for (int i = (from / 4) * 4; i < NVV; i += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(&A[k * WS + i]);
#pragma unroll
            for (int e = 0; e < 4; e++) l[i + e] = v[e];
        }
    };
    // the four wavefronts copy `count` floats from `from` (16-byte aligned when count % 4 == 0 and the block is) to LDS at `to`
    auto group_copy = [&](const float *from, float *to, int count) {
        const unsigned *blk = reinterpret_cast<const unsigned *>(from);
        if ((count & 3) == 0 && (reinterpret_cast<uintptr_t>(from) & 15) == 0) {
            for (int base = wave * 4 * kWave; base < count; base += G * 4 * kWave)
                if (base + 4 * lane < count)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(blk + base + 4 * lane),
                                                     (__attribute__((address_space(3))) void *)(to + base), 16, 0, 0);
        } else {
            for (int base = wave * kWave; base < count; base += G * kWave)
                if (base + lane < count)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(blk + base + lane),
                                                     (__attribute__((address_space(3))) void *)(to + base), 4, 0, 0);
        }
    };
    // per column tile of the right-hand sides: where this lane's column lives in the packed copy (entry index, before the
    // layout's element stride)
    int lo_base[NCT], hi_base[NCT], ccol[NCT], mat[NCT];
    uint64_t rel[NCT];
#pragma unroll
    for (int t = 0; t < NCT; t++) {
        const int c = 16 * t + c16;
        const bool valid = c < n_cols;
        const int m = (valid && c >= nv) ? 1 : 0, cc = valid ? c - m * nv : 0;
        mat[t] = valid ? m : -1;
        ccol[t] = cc;
        lo_base[t] = cc;
        hi_base[t] = cc * cc + cc + 1;
        rel[t] = valid ? (related ? related[cc] : ~uint64_t(0)) : uint64_t(0);
    }
    const uint64_t rel_mine = related ? related[lane < nv ? lane : 0] : ~uint64_t(0);
    const size_t n_groups = (B + G - 1) / G;
#ifdef GRBDA_EXP_MF_PROF
    unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_t = __builtin_amdgcn_s_memtime();
#endif
    for (size_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const size_t s = grp * G + wave;
        const bool live = s < B;  // (a wavefront past the end of the batch works on whatever LDS holds and stores nothing)
        const int n_valid = (int)(B - grp * G < (size_t)G ? B - grp * G : (size_t)G);
        // ---- 0. the group's H block into LDS; every lane takes its row of H ----
        __syncthreads();  // the previous group's fragment reads are done
        // (interleaved blocks are whole by construction of the workspace; a state-major block ends with the batch)
        group_copy(H + grp * (size_t)G * nn, Pg, (h_il == G ? G : n_valid) * nn);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        MF_STAMP(0)
        float Lr[NVV];  // row `lane` of H, then of L; the diagonal entry holds 1 / L[k][k]
        {
            const int lrow = lane < nv ? lane : 0;
            const float *hrow = Pg + (size_t)(h_packed ? lrow * (lrow + 1) / 2 : lrow * nv) * h_es + wave * h_ss;
#pragma unroll
            for (int j = 0; j < NVV; j++) {
                const float h = (j < nv && j <= lrow) ? hrow[j * h_es] : 0.0f;
                Lr[j] = (lane < nv && j < nv) ? ((j <= lane && ((rel_mine >> j) & 1)) ? h : 0.0f) : (lane == j ? 1.0f : 0.0f);
            }
        }
        __syncthreads();  // every wavefront has its H: the right-hand sides may land over the block
        for (int m = 0; m < n_mat; m++) group_copy(src[m] + grp * (size_t)G * nn, Pg + (size_t)m * G * nn, (p_il == G ? G : n_valid) * nn);
        MF_STAMP(1)
        // ---- 1. Cholesky ----
        bool bad = false;
#pragma unroll
        for (int k = 0; k < NVV; k++) {
            float sum = Lr[k];
#pragma unroll
            for (int m2 = 0; m2 < k; m2++) sum -= Lr[m2] * lane_value(Lr[m2], k);
            const float d = lane_value(sum, k);
            bad = bad || !(d > 0.0f) || !(d < 3e38f);
            const float r = inv_sqrt(d);
            Lr[k] = lane == k ? r : sum * r;
            if (lane >= k && lane < NVV) A[k * WS + lane] = Lr[k];
        }
        if (bad && lane == 0 && live) atomicAdd(&grbda_spd_bad_count, 1ull);
        wave_lds_fence();
        MF_STAMP(2)
        // ---- 2. W = L^-1: column `lane`, forward substitution by columns of L (rows of the stored L^T) ----
        {
            float x[NVV], l[NVV];
#pragma unroll
            for (int i = 0; i < NVV; i++) x[i] = lane == i ? 1.0f : 0.0f;
#pragma unroll
            for (int m2 = 0; m2 < NVV; m2++) {
                factor_row(m2, m2, l);
                x[m2] *= l[m2];
#pragma unroll
                for (int i = m2 + 1; i < NVV; i++) x[i] -= l[i] * x[m2];
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            // (a use of the solution outside the conditional stores: without it the compiler sinks the whole substitution
            // into the store block, below the LDS reads, and every factor row stays live)
#pragma unroll
            for (int i = 0; i < NVV; i++) asm volatile("" ::"v"(x[i]));
            wave_lds_fence();  // every row of L^T has been read: W goes over it
            if (lane < NT * 16) {
#pragma unroll
                for (int i = 0; i < NVV; i++) A[i * WS + lane] = lane < NVV ? x[i] : 0.0f;
            }
        }
        wave_lds_fence();
        MF_STAMP(3)
        // ---- 3. H^-1 = W^T W (rows and columns NVV .. 16 NT - 1 of the tile are padding: never stored, read as zero) ----
        f32x4 hi[NT][NT];
#pragma unroll
        for (int a = 0; a < NT; a++)
#pragma unroll
            for (int b = 0; b < NT; b++) hi[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KS; k++) {
            float w[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) w[t] = (16 * t <= 4 * k + 3) ? A[(4 * k + g) * WS + 16 * t + c16] : 0.0f;
#pragma unroll
            for (int a = 0; a < NT; a++)
#pragma unroll
                for (int b = 0; b < NT; b++)
                    if (16 * (a > b ? a : b) <= 4 * k + 3) hi[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[a], w[b], hi[a][b], 0, 0, 0);
        }
        // the right-hand sides have long arrived; waiting for them HERE, before the H^-1 stores are issued, keeps the wait from
        // also draining those stores (loads and stores share the counter)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_lds_fence();  // W has been read: H^-1 goes over it
#pragma unroll
        for (int a = 0; a < NT; a++)
#pragma unroll
            for (int b = 0; b < NT; b++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int row = 16 * a + 4 * g + j, col = 16 * b + c16;
                    if (row < NVV) A[row * WS + col] = hi[a][b][j];
                }
        if (Hinv && live) {
            // H^-1 is symmetric: the four values a lane holds of tile (a, b) -- rows 16 a + 4 g + 0..3 of column 16 b + c16 -- are
            // also columns 16 a + 4 g + 0..3 of ROW 16 b + c16: one 16-byte store instead of four 4-byte ones
            float *hout = Hinv + s * (size_t)nn;
#pragma unroll
            for (int a = 0; a < NT; a++)
#pragma unroll
                for (int b = 0; b < NT; b++) {
                    const int row = 16 * b + c16, col0 = 16 * a + 4 * g;
                    if (row < nv) {
                        if (col0 + 3 < nv) {
                            __builtin_memcpy(hout + (size_t)row * nv + col0, &hi[a][b], 16);
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; j++)
                                if (col0 + j < nv) hout[(size_t)row * nv + col0 + j] = hi[a][b][j];
                        }
                    }
                }
        }
        MF_STAMP(4)
        if (n_mat == 0) continue;
        __syncthreads();  // every wavefront's part of the copy has landed (and this wavefront's H^-1 is in LDS)
        // ---- 4. X = -H^-1 [P1 | P2] ----
        f32x4 acc[NT][NCT];
#pragma unroll
        for (int a = 0; a < NT; a++)
#pragma unroll
            for (int t = 0; t < NCT; t++) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *Pw = Pg + wave * p_ss;
#pragma unroll
        for (int k = 0; k < KS; k++) {
            int r = 4 * k + g;
            asm volatile("" : "+v"(r));  // (the gather addresses and masks do not depend on the state: left to itself the
                                         // compiler computes all NCT * KS of them before the loop and spills them)
            const int r2 = r * r;
            float av[NT], bv[NCT];
#pragma unroll
            for (int a = 0; a < NT; a++) av[a] = A[r * WS + 16 * a + c16];  // H^-1[a][r] = H^-1[r][a]
#pragma unroll
            for (int t = 0; t < NCT; t++) {
                if (t < nct) {
                    const bool ok = (rel[t] >> r) & 1;  // bit r of related[column]: r < nv, the column exists, not a structural zero
                    const int idx = ccol[t] <= r ? r2 + lo_base[t] : r + hi_base[t];
                    const float v = Pw[ok ? (mat[t] * G * nn + idx * p_es) : 0];
                    bv[t] = ok ? v : 0.0f;
                }
            }
#pragma unroll
            for (int t = 0; t < NCT; t++)
                if (t < nct) {
#pragma unroll
                    for (int a = 0; a < NT; a++) acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[t], av[a], acc[a][t], 0, 0, 0);
                }
        }
        MF_STAMP(5)
        // The product was taken transposed, (P^T H^-1)^T: with the right-hand-side fragment as the A operand a lane's four
        // accumulator values of tile (a, t) are X[16 a + c16][16 t + 4 g + 0..3] -- four consecutive entries of a row of the
        // result: one 16-byte store (two 8-byte ones at the odd alignment of an odd row when nv = 2 mod 4) instead of four.
        if (live) {
#pragma unroll
            for (int t = 0; t < NCT; t++)
                if (t < nct) {
                    const int c0 = 16 * t + 4 * g;  // first of this lane's four columns of [X1 | X2]
#pragma unroll
                    for (int a = 0; a < NT; a++) {
                        const int row = 16 * a + c16;
                        if (row >= nv) continue;
                        const f32x4 v = -acc[a][t];
                        const bool second = c0 >= nv;
                        const int cc0 = second ? c0 - nv : c0;
                        float *o0 = (second ? dst1 : dst0) + s * (size_t)nn + (size_t)row * nv + cc0;
                        if (c0 + 3 < n_cols && cc0 + 3 < nv) {  // all four in one matrix
                            __builtin_memcpy(o0, &v, 16);
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const int c = c0 + j;
                                if (c < n_cols) {
                                    if (c >= nv && !second) dst1[s * (size_t)nn + (size_t)row * nv + (c - nv)] = v[j];
                                    else o0[j] = v[j];
                                }
                            }
                        }
                    }
                }
        }
        MF_STAMP(6)
    }
#ifdef GRBDA_EXP_MF_PROF
    if (lane == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&mf_prof[i], prof_acc[i]);
#endif
}
#ifdef GRBDA_EXP_MF_PROF
extern "C" int grbda_debug_mf_prof(unsigned long long *out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(mf_prof), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(mf_prof), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif

template <int NVV>
static hipError_t launch_spd_mfma_n(const float *H, int h_packed, int h_il, const float *P1, const float *P2, int p_il, float *Hinv, float *X1,
                                    float *X2, const uint64_t *related, int nv, size_t B, int grid, hipStream_t stream)
{
    const size_t lds = spd_mfma_lds_bytes(nv, (P1 ? 1 : 0) + (P2 ? 1 : 0));  // (above the 64 KiB default: set_max_dynamic_lds_deriv, per device)
    hipLaunchKernelGGL((spd_mfma_kernel<NVV>), dim3(grid), dim3(kWave * kDerivGroup), lds, stream, H, h_packed, h_il, P1, P2, p_il, Hinv, X1,
                       X2, related, nv, B);
    return hipGetLastError();
}
// the matrix-core solve needs up to 160 KiB of dynamic LDS: raised per device by capi.cpp's ensure_device, like every other kernel
hipError_t set_max_dynamic_lds_deriv()
{
    const void *fns[] = {reinterpret_cast<const void *>(&spd_mfma_kernel<16>), reinterpret_cast<const void *>(&spd_mfma_kernel<24>),
                         reinterpret_cast<const void *>(&spd_mfma_kernel<32>), reinterpret_cast<const void *>(&spd_mfma_kernel<40>),
                         reinterpret_cast<const void *>(&spd_mfma_kernel<48>), reinterpret_cast<const void *>(&spd_mfma_kernel<64>)};
    for (const void *f : fns) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
static hipError_t launch_spd_mfma(const float *H, int h_packed, int h_il, const float *P1, const float *P2, int p_il, float *Hinv, float *X1,
                                  float *X2, const uint64_t *related, int nv, size_t B, int grid, hipStream_t stream)
{
    if (nv <= 16) return launch_spd_mfma_n<16>(H, h_packed, h_il, P1, P2, p_il, Hinv, X1, X2, related, nv, B, grid, stream);
    if (nv <= 24) return launch_spd_mfma_n<24>(H, h_packed, h_il, P1, P2, p_il, Hinv, X1, X2, related, nv, B, grid, stream);
    if (nv <= 32) return launch_spd_mfma_n<32>(H, h_packed, h_il, P1, P2, p_il, Hinv, X1, X2, related, nv, B, grid, stream);
    if (nv <= 40) return launch_spd_mfma_n<40>(H, h_packed, h_il, P1, P2, p_il, Hinv, X1, X2, related, nv, B, grid, stream);
    if (nv <= 48) return launch_spd_mfma_n<48>(H, h_packed, h_il, P1, P2, p_il, Hinv, X1, X2, related, nv, B, grid, stream);
    if (nv <= 64) return launch_spd_mfma_n<64>(H, h_packed, h_il, P1, P2, p_il, Hinv, X1, X2, related, nv, B, grid, stream);
    return hipErrorInvalidValue;
}

template <class TIO, class TC, int NV>
static hipError_t launch_spd_solve_n(const TIO *H, int h_packed, const TIO *P1, const TIO *P2, TIO *Hinv, TIO *X1, TIO *X2,
                                     const uint64_t *related, int nv, size_t B, int grid, hipStream_t stream)
{
    // two columns per lane: f32 arithmetic and more than one pass of 64 columns (register budget: 4 NV values per lane)
    const int ncols = (P1 ? nv : 0) + (P2 ? nv : 0) + (Hinv ? nv : 0);
    static const int kc_env = [] { const char *e = std::getenv("GRBDA_SOLVE_KC"); return e ? std::atoi(e) : 0; }();
    if constexpr (sizeof(TC) == 4 && NV == 40) {  // (measured: it pays for JVRC-1's 114 columns only)
        if (kc_env != 1 && (ncols > kWave || kc_env == 2)) {
            hipLaunchKernelGGL((spd_solve_kernel<TIO, TC, NV, 2>), dim3(grid), dim3(kWave), 0, stream, H, h_packed, P1, P2, Hinv, X1, X2,
                               related, nv, B);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((spd_solve_kernel<TIO, TC, NV, 1>), dim3(grid), dim3(kWave), 0, stream, H, h_packed, P1, P2, Hinv, X1, X2, related, nv,
                       B);
    return hipGetLastError();
}
// interleave: layout of H, P1, P2 -- 1 state-major, kDerivGroup the interleaved workspace of rnea_deriv_kernel (matrix-core kernel only)
// ---------------------------------------------------------------------------------------------------------------
// Tiny systems (nv <= 6: four_bar.urdf, six_bar.urdf, a leg on a bench): one state per LANE, everything in registers -- the
// one-state-per-wavefront kernels above pad such a system to 16 x 16 and spend their time on barriers and identity rows (six_bar,
// nv = 3: 0.70 ms per 262 144 states on the matrix-core kernel against 0.06 ms for the whole forward dynamics).  Same inputs
// (packed lower rows of H, packed runs of the right-hand sides, any interleave factor IL), same outputs.
// ---------------------------------------------------------------------------------------------------------------
template <class TIO, class TC, int NV>
__global__ __launch_bounds__(kWave) void spd_small_kernel(const TIO *__restrict__ H, int il, const TIO *__restrict__ P1, const TIO *__restrict__ P2,
                                                        TIO *__restrict__ Hinv, TIO *__restrict__ X1, TIO *__restrict__ X2,
                                                        const uint64_t *__restrict__ related, int nv, size_t B)
{
    const size_t nn = (size_t)nv * nv;
    for (size_t st = (size_t)blockIdx.x * kWave + threadIdx.x; st < B; st += (size_t)gridDim.x * kWave) {
        const size_t base = (st / il) * nn * il + st % il;
        TC D[NV][NV];
#pragma unroll
        for (int r = 0; r < NV; r++)
#pragma unroll
            for (int c = 0; c < NV; c++) {
                D[r][c] = r == c ? TC(1) : TC(0);
                if (r < nv && c <= r) {
                    const bool rel = !related || ((related[r] >> c) & 1);
                    const TC v = rel ? (TC)H[base + (size_t)(r * (r + 1) / 2 + c) * il] : TC(0);
                    D[r][c] = v;
                }
            }
#pragma unroll
        for (int r = 0; r < NV; r++)
#pragma unroll
            for (int c = r + 1; c < NV; c++) D[r][c] = D[c][r];
        Chol<TC, NV> ch;
        ch.factor(D);
        bool bad = false;
#pragma unroll
        for (int j = 0; j < NV; j++) bad = bad || !(ch.inv[j] > TC(0)) || !(ch.inv[j] < TC(1e30));
        if (bad) atomicAdd(&grbda_spd_bad_count, 1ull);
        for (int m = 0; m < 3; m++) {
            const TIO *P = m == 0 ? P1 : (m == 1 ? P2 : nullptr);
            TIO *X = m == 0 ? X1 : (m == 1 ? X2 : Hinv);
            if (!X || (m < 2 && !P)) continue;
            for (int c = 0; c < nv; c++) {
                TC b[NV];
#pragma unroll
                for (int r = 0; r < NV; r++) {
                    b[r] = 0;
                    if (r < nv) {
                        if (m == 2) {
                            b[r] = r == c ? TC(1) : TC(0);
                        } else if (!related || ((related[r] >> c) & 1)) {
                            const size_t e = c <= r ? (size_t)(r * r + c) : (size_t)(c * c + c + 1 + r);
                            b[r] = -(TC)P[base + e * il];
                        }
                    }
                }
                ch.solve(b);
#pragma unroll
                for (int r = 0; r < NV; r++)
                    if (r < nv) X[st * nn + (size_t)r * nv + c] = (TIO)b[r];
            }
        }
    }
}
template <class TIO, class TC>
static hipError_t launch_spd_small(const TIO *H, const TIO *P1, const TIO *P2, TIO *Hinv, TIO *X1, TIO *X2, const uint64_t *related, int nv, size_t B,
                                   hipStream_t stream, int il)
{
    size_t grid = (B + kWave - 1) / kWave;
    if (grid > 65535u * 4u) grid = 65535u * 4u;
    if (nv <= 2) hipLaunchKernelGGL((spd_small_kernel<TIO, TC, 2>), dim3(grid), dim3(kWave), 0, stream, H, il, P1, P2, Hinv, X1, X2, related, nv, B);
    else if (nv <= 4) hipLaunchKernelGGL((spd_small_kernel<TIO, TC, 4>), dim3(grid), dim3(kWave), 0, stream, H, il, P1, P2, Hinv, X1, X2, related, nv, B);
    else hipLaunchKernelGGL((spd_small_kernel<TIO, TC, 6>), dim3(grid), dim3(kWave), 0, stream, H, il, P1, P2, Hinv, X1, X2, related, nv, B);
    return hipGetLastError();
}

template <class TIO, class TC>
hipError_t launch_spd_solve(const TIO *H, int h_packed, const TIO *P1, const TIO *P2, TIO *Hinv, TIO *X1, TIO *X2, const uint64_t *related,
                            int nv, size_t B, int grid, hipStream_t stream, int interleave)
{
    if (nv <= 6 && h_packed) return launch_spd_small<TIO, TC>(H, P1, P2, Hinv, X1, X2, related, nv, B, stream, interleave);
    if constexpr (sizeof(TIO) == 4 && sizeof(TC) == 4) {
        if (spd_solve_on_mfma(4, nv, (P1 ? 1 : 0) + (P2 ? 1 : 0)))
            return launch_spd_mfma(H, h_packed, interleave, P1, P2, interleave, Hinv, X1, X2, related, nv, B, grid, stream);
    }
    if (interleave != 1) return hipErrorInvalidValue;
    if (nv <= 16) return launch_spd_solve_n<TIO, TC, 16>(H, h_packed, P1, P2, Hinv, X1, X2, related, nv, B, grid, stream);
    if (nv <= 24) return launch_spd_solve_n<TIO, TC, 24>(H, h_packed, P1, P2, Hinv, X1, X2, related, nv, B, grid, stream);
    if (nv <= 32) return launch_spd_solve_n<TIO, TC, 32>(H, h_packed, P1, P2, Hinv, X1, X2, related, nv, B, grid, stream);
    if (nv <= 40) return launch_spd_solve_n<TIO, TC, 40>(H, h_packed, P1, P2, Hinv, X1, X2, related, nv, B, grid, stream);
    if (nv <= 48) return launch_spd_solve_n<TIO, TC, 48>(H, h_packed, P1, P2, Hinv, X1, X2, related, nv, B, grid, stream);
    if (nv <= 64) return launch_spd_solve_n<TIO, TC, 64>(H, h_packed, P1, P2, Hinv, X1, X2, related, nv, B, grid, stream);
    return hipErrorInvalidValue;
}
template hipError_t launch_spd_solve<float, float>(const float *, int, const float *, const float *, float *, float *, float *, const uint64_t *,
                                                   int, size_t, int, hipStream_t, int);
template hipError_t launch_spd_solve<float, double>(const float *, int, const float *, const float *, float *, float *, float *,
                                                    const uint64_t *, int, size_t, int, hipStream_t, int);
template hipError_t launch_spd_solve<double, double>(const double *, int, const double *, const double *, double *, double *, double *,
                                                     const uint64_t *, int, size_t, int, hipStream_t, int);

// LDS of one workgroup of the solve: the [NVV][WS] tiles of its four wavefronts plus the group's right-hand sides (matrix-core
// kernel), or the factor at the compile-time size the launch picks for nv (VALU kernel, one wavefront)
size_t spd_solve_lds_bytes(int nv, size_t elem, int n_rhs)
{
    if (spd_solve_on_mfma(elem, nv, n_rhs)) return spd_mfma_lds_bytes(nv, n_rhs);
    const int nvb = nv <= 16 ? 16 : (nv <= 24 ? 24 : (nv <= 32 ? 32 : (nv <= 40 ? 40 : (nv <= 48 ? 48 : 64))));
    return static_cast<size_t>(nvb) * nvb * elem;
}

}  // namespace grbda_hip
