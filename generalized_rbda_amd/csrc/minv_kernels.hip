// minv_kernels.hip -- d ydd / d tau = H^-1 and the products -H^-1 [dID/dq | dID/dqd] of the derivative pipeline (BASELINE config 5)
// WITHOUT a dense factorisation of H: the cluster ABA already factorises the joint-space inertia,
//     H^-1 = W^T W,      W = D^-1/2 (1 - psi)          (block row of cluster a, column j in a or below a)
// (the innovations factorisation of Rodriguez / Jain; the backward sweep of Carpentier's computeMinverse), with D_a = S_a^T IA_a S_a of
// ClusterTreeModel::updateArticulatedBodies (ClusterTreeDynamics.cpp:157-191) and psi the force propagators U D^-1 S^T along the tree.
//   abi_factor_kernel   one state per lane: articulated-inertia recursion in the common frame F of deriv_kernels.hip (F coincides with the
//                       floating base at this instant: composites add without transforms), per state a record block
//                       [K = F D^-1 | L^-1 (D = L L^T) | S_ab] per cluster (plan.h, MinvProgram);
//   minv_mfma_kernel    one state per wavefront, lane = column j: walks j up its root path through the record block in LDS
//                       (f = K_c e_j;  per ancestor a:  sigma = S_ab^T f,  W[a][j] = -L_a^-1 sigma,  f -= K_a sigma), then
//                       H^-1 = W^T W and [X1 | X2] = -H^-1 [P1 | P2] as 16 x 16 x 4 matrix-core tiles (f32 and f64).
// What this replaces (deriv_kernels.hip, spd_mfma_kernel / spd_solve_kernel): the Cholesky factorisation of H by v_readlane (22 % of that
// kernel's time) and the inversion of its factor (16 %), ~2.8 k of its ~5.7 k instructions per state, and H itself (741 scalars per JVRC-1
// state written by the recursion and read back; the record block is 437).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "devplan.h"

namespace grbda_hip {

#include "devmath.h"

template <class T>
__device__ __forceinline__ T dot6m(const T (&a)[6], const T (&b)[6])
{
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}

// kinematics in F of a revolute body from its parent's [E 9 | p 3]: E (F -> body), p (body origin in F), S (joint axis in F)
template <class T>
__device__ __forceinline__ void minv_kin(cptr<T> C, bool axisym, T qi, const T (&kp)[12], T (&kin)[12], T (&S)[6])
{
    T sn = 0, cs = 1, El[9];
    if (!axisym) sincos_t(qi, &sn, &cs);  // (a rotor's inertia and axis in F do not depend on its own angle)
    rotate_z(sn, cs, C, El);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) kin[3 * i + j] = El[3 * i] * kp[j] + El[3 * i + 1] * kp[3 + j] + El[3 * i + 2] * kp[6 + j];
#pragma unroll
    for (int i = 0; i < 3; i++) kin[9 + i] = kp[9 + i] + kp[i] * C[9] + kp[3 + i] * C[10] + kp[6 + i] * C[11];
    S[0] = kin[6]; S[1] = kin[7]; S[2] = kin[8];
    S[3] = kin[10] * S[2] - kin[11] * S[1];
    S[4] = kin[11] * S[0] - kin[9] * S[2];
    S[5] = kin[9] * S[1] - kin[10] * S[0];
}
// the PARENT's [E | p] from a revolute body's own: the step above undone (E_p = E_l^T E, p_p = p - E_p^T r)
template <class T>
__device__ __forceinline__ void minv_kin_up(cptr<T> C, bool axisym, T qi, const T (&kin)[12], T (&kp)[12])
{
    T sn = 0, cs = 1, El[9];
    if (!axisym) sincos_t(qi, &sn, &cs);
    rotate_z(sn, cs, C, El);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) kp[3 * r + c] = El[r] * kin[c] + El[3 + r] * kin[3 + c] + El[6 + r] * kin[6 + c];
#pragma unroll
    for (int i = 0; i < 3; i++) kp[9 + i] = kin[9 + i] - (kp[i] * C[9] + kp[3 + i] * C[10] + kp[6 + i] * C[11]);
}
template <class T>
__device__ __forceinline__ void axis_of(const T (&kin)[12], T (&S)[6])
{
    S[0] = kin[6]; S[1] = kin[7]; S[2] = kin[8];
    S[3] = kin[10] * S[2] - kin[11] * S[1];
    S[4] = kin[11] * S[0] - kin[9] * S[2];
    S[5] = kin[9] * S[1] - kin[10] * S[0];
}

// ---------------------------------------------------------------------------------------------------------------
// abi_factor_kernel.  Same processing order and the same slab rows as rnea_deriv_kernel (plan.h, DerivBody: kin_row -- of which the
// first 12 rows [E | p] are used here --, acc_row -- the first 21 --, the first-writer flags and the register hand-over along chains),
// so no plan of its own: the two kernels run one after the other on one stream and one slab.
// ---------------------------------------------------------------------------------------------------------------
template <class T, int NMAX, int IL>
__global__ __launch_bounds__(kWave, 2) void abi_factor_kernel(DevPlan<T> DP, const DerivBody *__restrict__ db_, const MinvBody *__restrict__ mb_,
                                                              int n_clusters, int n_rows, int n_entries, const T *__restrict__ q,
                                                              T *__restrict__ rec, size_t B, T *__restrict__ scratch,
                                                              unsigned long long *__restrict__ bad_count)
{
    cptr<ClusterRec> clusters = (cptr<ClusterRec>)DP.clusters;
    cptr<BodyRec> bodies = (cptr<BodyRec>)DP.bodies;
    cptr<T> consts = (cptr<T>)DP.consts;
    cptr<DerivBody> db = (cptr<DerivBody>)db_;
    cptr<MinvBody> mb = (cptr<MinvBody>)mb_;
    const int lane = threadIdx.x, nq = DP.nq;
    T *slab = scratch + (size_t)blockIdx.x * (size_t)n_rows * kWave + lane;
    auto ld12 = [&](int row, T(&x)[12]) {
#pragma unroll
        for (int i = 0; i < 12; i++) x[i] = slab[(size_t)(row + i) * kWave];
    };
    auto ld21 = [&](int row, T(&x)[21]) {
#pragma unroll
        for (int i = 0; i < 21; i++) x[i] = slab[(size_t)(row + i) * kWave];
    };
    const size_t n_tiles = (B + kWave - 1) / kWave;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t r = tile * kWave + lane;
        const size_t st = r < B ? r : B - 1;  // lanes past the end redo the last state and do not store
        const bool live = r < B;
        bool bad = false;  // a pivot of some D that is not positive and finite: the state's results are NaN / Inf, and it is counted
        // the tile's q block through LDS (coalesced LDS-DMA; every lane then reads its own row): read from the caller's array the
        // coordinates are 4-byte accesses nq scalars apart, 64 cache lines per load instruction
        const T *qs;
        {
            const size_t left = B - tile * kWave;
            const int rows_valid = left < (size_t)kWave ? (int)left : kWave;
            wave_lds_fence();  // the previous tile's reads of the block are done
            stage_issue(q, tile, rows_valid, nq, 0u, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wave_lds_fence();
            const int my = lane < rows_valid ? lane : rows_valid - 1;
            qs = reinterpret_cast<const T *>(grbda_smem) + my * nq;
        }
        T *recs = rec + (st / IL) * (size_t)n_entries * IL + st % IL;
        auto put = [&](int e, T v) {
            if (live) recs[(size_t)e * IL] = v;
        };
        // ---- pass 1, root side first: [E | p] in F of every body that has children ----
        int last_gb = -1;
        T last_kin[12];
#pragma unroll
        for (int j = 0; j < 12; j++) last_kin[j] = 0;
        for (int c = 0; c < n_clusters; c++) {
            const ClusterRec cr = load_rec(clusters + c);
            if (cr.kind == CK_FREE) {
                T kin[12];
#pragma unroll
                for (int j = 0; j < 12; j++) kin[j] = (j < 9 && j % 4 == 0) ? T(1) : T(0);
                const DerivBody x = load_rec(db + cr.first_body);
                if (x.kin_row >= 0 && load_rec(mb + cr.first_body).keep) {
#pragma unroll
                    for (int j = 0; j < 12; j++) slab[(size_t)(x.kin_row + j) * kWave] = kin[j];
                }
                last_gb = cr.first_body;
#pragma unroll
                for (int j = 0; j < 12; j++) last_kin[j] = kin[j];
                continue;
            }
            for (int i = 0; i < cr.k; i++) {
                if (!((cr.child_mask >> i) & 1)) continue;
                const int gb = cr.first_body + i;
                const BodyRec b = load_rec(bodies + gb);
                const DerivBody x = load_rec(db + gb);
                cptr<T> C = consts + b.cofs;
                T qi = 0;
                for (int a2 = 0; a2 < cr.n; a2++) qi += C[kBodyConstFixed + a2] * qs[cr.q_index + a2];
                T kp[12];
                if (b.parent >= 0 && b.parent == last_gb) {
#pragma unroll
                    for (int j = 0; j < 12; j++) kp[j] = last_kin[j];
                } else if (b.parent >= 0) {
                    const DerivBody xp = load_rec(db + b.parent);
                    ld12(xp.kin_row, kp);
                } else {
#pragma unroll
                    for (int j = 0; j < 12; j++) kp[j] = (j < 9 && j % 4 == 0) ? T(1) : T(0);
                }
                T kin[12], S[6];
                minv_kin(C, false, qi, kp, kin, S);
                if (load_rec(mb + gb).keep) {  // (only the rows somebody loads: plan.h, MinvBody::keep)
#pragma unroll
                    for (int j = 0; j < 12; j++) slab[(size_t)(x.kin_row + j) * kWave] = kin[j];
                }
                last_gb = gb;
#pragma unroll
                for (int j = 0; j < 12; j++) last_kin[j] = kin[j];
            }
        }
        // ---- pass 2, leaf side first: articulated inertias, D, K, L^-1 ----
        T part[21];  // articulated inertia the in-cluster roots of a cluster hand to the parent body (registers along chains)
        T kin_carry[12];  // ... and, beside it, the kinematics of that parent body: the next cluster derives ITS parent's from them
#pragma unroll
        for (int j = 0; j < 21; j++) part[j] = 0;
#pragma unroll
        for (int j = 0; j < 12; j++) kin_carry[j] = 0;
        for (int c = n_clusters - 1; c >= 0; c--) {
            const ClusterRec cr = load_rec(clusters + c);
            if (cr.kind == CK_FREE) {
                // base: S = 1 in F, D = its articulated inertia (6 x 6)
                const BodyRec b = load_rec(bodies + cr.first_body);
                const DerivBody x = load_rec(db + cr.first_body);
                const MinvBody xm = load_rec(mb + cr.first_body);
                cptr<T> Ib = consts + b.cofs + 12;
                T IA[21];
#pragma unroll
                for (int j = 0; j < 21; j++) IA[j] = Ib[j];
                if (x.acc_row >= 0) {
                    T acc[21];
                    ld21(x.acc_row, acc);
#pragma unroll
                    for (int j = 0; j < 21; j++) IA[j] += acc[j];
                }
                T Dm[6][6];
#pragma unroll
                for (int i = 0; i < 6; i++)
#pragma unroll
                    for (int j = 0; j < 6; j++) Dm[i][j] = IA[sidx(i, j)];
                Chol<T, 6> ch;
                ch.factor(Dm);
#pragma unroll
                for (int j = 0; j < 6; j++) bad = bad || !(ch.inv[j] > T(0)) || !(ch.inv[j] < T(1e30));
                // L^-1 column by column (forward substitution on the identity), stored as the packed lower triangle, row-major
#pragma unroll
                for (int e = 0; e < 6; e++) {
                    T xcol[6];
#pragma unroll
                    for (int i = 0; i < 6; i++) {
                        if (i < e) { xcol[i] = 0; continue; }
                        T s = i == e ? T(1) : T(0);
#pragma unroll
                        for (int m2 = 0; m2 < i; m2++)
                            if (m2 >= e) s -= ch.L[i][m2] * xcol[m2];
                        xcol[i] = s * ch.inv[i];
                        put(xm.clus_off + i * (i + 1) / 2 + e, xcol[i]);
                    }
                }
                continue;
            }
            const int n = cr.n;
            T Dm[NMAX][NMAX], F[NMAX][6];
            const DerivBody xf = load_rec(db + cr.first_body);
            const MinvBody xmf = load_rec(mb + cr.first_body);
            const int first_i = xf.carry_body >= 0 ? xf.carry_body - cr.first_body : -1;
            if (first_i < 0) {
#pragma unroll
                for (int j = 0; j < 21; j++) part[j] = 0;
            }
            // kinematics of the cluster's parent body: undone from the carried body's along a chain, loaded otherwise
            T kin_pb[12];
            if (first_i >= 0) {
                const BodyRec bc = load_rec(bodies + (cr.first_body + first_i));
                cptr<T> Cc = consts + bc.cofs;
                T qc = 0;
                for (int a2 = 0; a2 < n; a2++) qc += Cc[kBodyConstFixed + a2] * qs[cr.q_index + a2];
                minv_kin_up(Cc, bc.axisym != 0, qc, kin_carry, kin_pb);
            } else if (cr.parent_body >= 0) {
                ld12(load_rec(db + cr.parent_body).kin_row, kin_pb);
            } else {
#pragma unroll
                for (int j = 0; j < 12; j++) kin_pb[j] = (j < 9 && j % 4 == 0) ? T(1) : T(0);
            }
#pragma unroll
            for (int a2 = 0; a2 < NMAX; a2++) {
#pragma unroll
                for (int j = 0; j < 6; j++) F[a2][j] = 0;
#pragma unroll
                for (int b2 = 0; b2 < NMAX; b2++) Dm[a2][b2] = 0;
            }
            // (the body that receives a carried inertia comes first, the others last body first: rnea_deriv_kernel's order)
            for (int step = first_i >= 0 ? -1 : 0; step < cr.k; step++) {
                int i = first_i;
                if (step >= 0) {
                    i = cr.k - 1 - step;
                    if (i == first_i) continue;
                }
                const int gb = cr.first_body + i;
                const BodyRec b = load_rec(bodies + gb);
                const DerivBody x = load_rec(db + gb);
                const MinvBody xm = load_rec(mb + gb);
                cptr<T> C = consts + b.cofs;
                T Gi[NMAX];
#pragma unroll
                for (int a2 = 0; a2 < NMAX; a2++) Gi[a2] = a2 < n ? C[kBodyConstFixed + a2] : T(0);
                T kin[12], S[6];
                if (step < 0) {  // the body a child cluster handed its inertia to: its kinematics came along
#pragma unroll
                    for (int j = 0; j < 12; j++) kin[j] = kin_carry[j];
                    axis_of(kin, S);
                } else {
                    T qi = 0;
#pragma unroll
                    for (int a2 = 0; a2 < NMAX; a2++)
                        if (a2 < n) qi += Gi[a2] * qs[cr.q_index + a2];
                    T kp[12];
                    if (b.lam >= 0) {  // in-cluster parent: its row is kept
                        ld12(load_rec(db + b.lam).kin_row, kp);
                    } else {
#pragma unroll
                        for (int j = 0; j < 12; j++) kp[j] = kin_pb[j];
                    }
                    minv_kin(C, b.axisym != 0, qi, kp, kin, S);
                }
                T IA[21];
                {
                    T E[9], p3[3];
#pragma unroll
                    for (int j = 0; j < 9; j++) E[j] = kin[j];
#pragma unroll
                    for (int j = 0; j < 3; j++) p3[j] = kin[9 + j];
                    congruence_rigid(E, p3, C + 12, IA);
                }
                if (step < 0) {
#pragma unroll
                    for (int j = 0; j < 21; j++) {
                        IA[j] += part[j];
                        part[j] = 0;
                    }
                } else if (x.acc_row >= 0) {
                    T acc[21];
                    ld21(x.acc_row, acc);
#pragma unroll
                    for (int j = 0; j < 21; j++) IA[j] += acc[j];
                }
                T tA[6];
                symv(IA, S, tA);
                const T sh = dot6m(S, tA);
#pragma unroll
                for (int a2 = 0; a2 < NMAX; a2++)
#pragma unroll
                    for (int b2 = 0; b2 < NMAX; b2++) Dm[a2][b2] += Gi[a2] * Gi[b2] * sh;
                // motion of this body per unit cluster coordinate (bodies that carry child clusters): its own axis and its in-cluster
                // ancestors'
                T Sab[NMAX][6];
#pragma unroll
                for (int a2 = 0; a2 < NMAX; a2++)
#pragma unroll
                    for (int j = 0; j < 6; j++) Sab[a2][j] = Gi[a2] * S[j];
                int l = b.lam;
                while (l >= 0) {
                    const BodyRec bl = load_rec(bodies + l);
                    const DerivBody xl = load_rec(db + l);
                    cptr<T> Cl = consts + bl.cofs;
                    T kl[12], Sl[6], Gl[NMAX];
                    ld12(xl.kin_row, kl);
                    axis_of(kl, Sl);
#pragma unroll
                    for (int a2 = 0; a2 < NMAX; a2++) Gl[a2] = a2 < n ? Cl[kBodyConstFixed + a2] : T(0);
                    const T lh = dot6m(Sl, tA);
#pragma unroll
                    for (int a2 = 0; a2 < NMAX; a2++) {
#pragma unroll
                        for (int b2 = 0; b2 < NMAX; b2++) Dm[a2][b2] += (Gi[a2] * Gl[b2] + Gl[a2] * Gi[b2]) * lh;
#pragma unroll
                        for (int j = 0; j < 6; j++) Sab[a2][j] += Gl[a2] * Sl[j];
                    }
                    l = bl.lam;
                }
                if (xm.blk_off >= 0) {
                    const int so = xm.blk_off + 6 * n + n * (n + 1) / 2;
#pragma unroll
                    for (int a2 = 0; a2 < NMAX; a2++)
                        if (a2 < n) {
#pragma unroll
                            for (int j = 0; j < 6; j++) put(so + 6 * a2 + j, Sab[a2][j]);
                        }
                }
#pragma unroll
                for (int a2 = 0; a2 < NMAX; a2++)
#pragma unroll
                    for (int j = 0; j < 6; j++) F[a2][j] += Gi[a2] * tA[j];
                // inertia to the tree parent: in-cluster parents through their accumulator rows (the first writer stores, the
                // others add), the parent body of the cluster through `part`
                if (b.lam >= 0) {
                    const DerivBody xl = load_rec(db + b.lam);
                    if (!x.acc_first) {
                        T acc[21];
                        ld21(xl.acc_row, acc);
#pragma unroll
                        for (int j = 0; j < 21; j++) IA[j] += acc[j];
                    }
#pragma unroll
                    for (int j = 0; j < 21; j++) slab[(size_t)(xl.acc_row + j) * kWave] = IA[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 21; j++) part[j] += IA[j];
                }
            }
            // D = L L^T;  L^-1;  K = F D^-1;  part -= K F^T
            T Li[NMAX][NMAX];  // L^-1, lower triangular
            if constexpr (NMAX == 1) {
                const T d = Dm[0][0];
                bad = bad || !(d > T(0)) || !(d < T(1e30));
                Li[0][0] = rsqrt_t(d);
                if constexpr (sizeof(T) == 4) Li[0][0] = Li[0][0] * (T(1.5) - T(0.5) * d * Li[0][0] * Li[0][0]);
            } else {
#pragma unroll
                for (int a2 = 0; a2 < NMAX; a2++)
                    if (a2 >= n) Dm[a2][a2] = T(1);  // (padding: identity)
                Chol<T, NMAX> ch;
                ch.factor(Dm);
#pragma unroll
                for (int j = 0; j < NMAX; j++) bad = bad || !(ch.inv[j] > T(0)) || !(ch.inv[j] < T(1e30));
#pragma unroll
                for (int e = 0; e < NMAX; e++)
#pragma unroll
                    for (int i = 0; i < NMAX; i++) {
                        if (i < e) { Li[i][e] = 0; continue; }
                        T s = i == e ? T(1) : T(0);
#pragma unroll
                        for (int m2 = 0; m2 < i; m2++)
                            if (m2 >= e) s -= ch.L[i][m2] * Li[m2][e];
                        Li[i][e] = s * ch.inv[i];
                    }
            }
            // D^-1 = L^-T L^-1
            T K[NMAX][6];
#pragma unroll
            for (int a2 = 0; a2 < NMAX; a2++) {
#pragma unroll
                for (int j = 0; j < 6; j++) K[a2][j] = 0;
#pragma unroll
                for (int b2 = 0; b2 < NMAX; b2++) {
                    T di = 0;
#pragma unroll
                    for (int m2 = 0; m2 < NMAX; m2++)
                        if (m2 >= a2 && m2 >= b2) di += Li[m2][a2] * Li[m2][b2];
#pragma unroll
                    for (int j = 0; j < 6; j++) K[a2][j] += di * F[b2][j];
                }
            }
            // [K | L^-1] into every block of the cluster (one per body that carries child clusters; the first one always exists)
            for (int i = -1; i < cr.k; i++) {
                int o = xmf.clus_off;
                if (i >= 0) {
                    if (!((cr.child_mask >> i) & 1)) continue;
                    o = load_rec(mb + (cr.first_body + i)).blk_off;
                    if (o < 0 || o == xmf.clus_off) continue;
                }
#pragma unroll
                for (int a2 = 0; a2 < NMAX; a2++)
                    if (a2 < n) {
#pragma unroll
                        for (int j = 0; j < 6; j++) put(o + 6 * a2 + j, K[a2][j]);
#pragma unroll
                        for (int m2 = 0; m2 < NMAX; m2++)
                            if (m2 <= a2) put(o + 6 * n + a2 * (a2 + 1) / 2 + m2, Li[a2][m2]);
                    }
            }
            if (cr.parent_body >= 0) {
#pragma unroll
                for (int i = 0; i < 6; i++)
#pragma unroll
                    for (int j = i; j < 6; j++) {
                        T s = 0;
#pragma unroll
                        for (int a2 = 0; a2 < NMAX; a2++) s += K[a2][i] * F[a2][j];
                        part[sidx(i, j)] -= s;
                    }
                if (!xf.carry_out) {
                    const DerivBody xp = load_rec(db + cr.parent_body);
                    if (!xf.cluster_acc_first) {
                        T acc[21];
                        ld21(xp.acc_row, acc);
#pragma unroll
                        for (int j = 0; j < 21; j++) part[j] += acc[j];
                    }
#pragma unroll
                    for (int j = 0; j < 21; j++) slab[(size_t)(xp.acc_row + j) * kWave] = part[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 12; j++) kin_carry[j] = kin_pb[j];
                }
            }
        }
        if (bad && live && bad_count) atomicAdd(bad_count, 1ull);
    }
}

template <class T, int IL>
static hipError_t launch_abi_factor_il(const DevPlan<T> &P, const DerivBody *db, const MinvBody *mb, int n_clusters, int n_rows, int n_max,
                                       int n_entries, const T *q, T *rec, size_t B, T *scratch, int grid, hipStream_t stream,
                                       unsigned long long *bad_count)
{
    const size_t lds = static_cast<size_t>(kWave) * P.nq * sizeof(T);
    if (n_max <= 1)
        hipLaunchKernelGGL((abi_factor_kernel<T, 1, IL>), dim3(grid), dim3(kWave), lds, stream, P, db, mb, n_clusters, n_rows, n_entries, q, rec, B,
                           scratch, bad_count);
    else if (n_max <= 2)
        hipLaunchKernelGGL((abi_factor_kernel<T, 2, IL>), dim3(grid), dim3(kWave), lds, stream, P, db, mb, n_clusters, n_rows, n_entries, q, rec, B,
                           scratch, bad_count);
    else
        hipLaunchKernelGGL((abi_factor_kernel<T, kMaxClusterDof, IL>), dim3(grid), dim3(kWave), lds, stream, P, db, mb, n_clusters, n_rows,
                           n_entries, q, rec, B, scratch, bad_count);
    return hipGetLastError();
}
template <class T>
hipError_t launch_abi_factor(const DevPlan<T> &P, const DerivBody *db, const MinvBody *mb, int n_clusters, int n_rows, int n_max, int n_entries,
                             const T *q, T *rec, size_t B, T *scratch, int grid, hipStream_t stream, int interleave,
                             unsigned long long *bad_count)
{
    if (interleave == kDerivGroup)
        return launch_abi_factor_il<T, kDerivGroup>(P, db, mb, n_clusters, n_rows, n_max, n_entries, q, rec, B, scratch, grid, stream, bad_count);
    if (interleave != 1) return hipErrorInvalidValue;
    return launch_abi_factor_il<T, 1>(P, db, mb, n_clusters, n_rows, n_max, n_entries, q, rec, B, scratch, grid, stream, bad_count);
}
template hipError_t launch_abi_factor<float>(const DevPlan<float> &, const DerivBody *, const MinvBody *, int, int, int, int, const float *, float *,
                                             size_t, float *, int, hipStream_t, int, unsigned long long *);
template hipError_t launch_abi_factor<double>(const DevPlan<double> &, const DerivBody *, const MinvBody *, int, int, int, int, const double *,
                                              double *, size_t, double *, int, hipStream_t, int, unsigned long long *);

// ---------------------------------------------------------------------------------------------------------------
// minv_mfma_kernel<T, NVV, NMAX>.  A workgroup is G = kDerivGroup wavefronts and takes one GROUP of G states at a time, one state per
// wavefront; the group's record blocks and packed right-hand sides are contiguous ([entry][G], the interleaved workspace of the two
// one-state-per-lane kernels) and the G wavefronts copy them into LDS together (global_load_lds, 16 bytes per lane and instruction).
// Per wavefront:
//   1. the tile [NVV][WS] is cleared and lane j walks column j up its root path (coltab in registers, loaded once per kernel): W;
//   2. H^-1 = W^T W     NT x NT tiles of 16 x 16 (W is upper block-triangular: row k only reaches columns >= k's cluster);
//   3. [X1 | X2] = -H^-1 [P1 | P2]   A fragments from the H^-1 tile (symmetric: read by rows), B fragments gathered from the packed
//      right-hand sides at LDS addresses that do not depend on the state: every lane keeps the KS x NCT byte addresses of its
//      fragment elements in registers (worked out once per kernel; structural zeros and padding point at a zero word), so the
//      gather is one ds_read per element and no address arithmetic.  (spd_mfma_kernel recomputed index and mask per element, ~8
//      VALU instructions each: it had no registers to keep them in beside the rows of the factorisation.)
// The record block is staged in the LDS region of the right-hand sides first; they are copied over it while steps 1 - 2 run.
// Fragment maps (cdna_hip_programming.md 3; the f64 instruction has the same A / B maps): A[l & 15][l >> 4], B[l >> 4][l & 15];
// D f32: column l & 15, rows 4 (l >> 4) + 0..3;  D f64 (v_mfma_f64_16x16x4_f64): column l & 15, rows 4 r + (l >> 4), r = 0..3.
// ---------------------------------------------------------------------------------------------------------------
// optional in-kernel phase profile (make expv NAME=mvprof DEFS=-DGRBDA_EXP_MV_PROF, tools/time_solve.py; never in the shipped library)
#ifdef GRBDA_EXP_MV_PROF
__device__ unsigned long long mv_prof[8];
#define MV_STAMP(i) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); prof_acc[i] += now_ - prof_t; prof_t = now_; }
extern "C" int grbda_debug_mv_prof(unsigned long long *out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(mv_prof), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(mv_prof), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#else
#define MV_STAMP(i)
#endif
// phase ablation (experiment builds only: -DGRBDA_EXP_MV_ABL, GRBDA_MV_ABL=bits; wrong results): 1 no result stores, 2 no H^-1 store,
// 4 no walk, 8 no second product, 16 no right-hand-side copy, 32 no first product, 64 no record copy
#ifdef GRBDA_EXP_MV_ABL
__device__ int mv_abl = 0;
extern "C" int grbda_debug_mv_abl(int bits) { return hipMemcpyToSymbol(HIP_SYMBOL(mv_abl), &bits, sizeof bits) == hipSuccess ? 0 : -1; }
#define MV_ABL(b) (mv_abl & (b))
#else
#define MV_ABL(b) 0
#endif
typedef float f32x4m __attribute__((ext_vector_type(4)));
typedef double f64x4m __attribute__((ext_vector_type(4)));
template <class T>
struct Mfma;
template <>
struct Mfma<float> {
    typedef f32x4m Acc;
    static __device__ __forceinline__ Acc run(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    // row of accumulator value r of lane group g inside the 16 x 16 tile
    static __device__ __forceinline__ int row(int g, int r) { return 4 * g + r; }
};
template <>
struct Mfma<double> {
    typedef f64x4m Acc;
    static __device__ __forceinline__ Acc run(double a, double b, Acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int g, int r) { return 4 * r + g; }
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a release / acquire fence at workgroup scope and compiles to
// s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier: it drains every global STORE the wavefront has in flight, which is exactly what the pipeline of
// minv_mfma_kernel must not do (measured: with __syncthreads() the result stores of a group never overlapped the next group's arithmetic).
// Data that arrives by global_load_lds is waited for explicitly (s_waitcnt vmcnt) by the wavefront that issued the copy, before the barrier.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int NVV>
struct MinvShape {
    static constexpr int NT = (NVV + 15) / 16;        // row / column tiles of H^-1
    // row stride of a wavefront's tile: NVV, not 16 NT -- fragment reads of the columns NVV .. 16 NT - 1 then run into the next row (or the
    // next tile: always inside the workgroup's LDS) and fetch values that only ever reach padding rows / columns of the products, which are
    // never stored; WRITES to the tile are guarded.  JVRC-1 (NVV = 40): 25.6 instead of 30.7 KB of tiles per workgroup.
    static constexpr int WS = NVV;
    static constexpr int NCT = (2 * NVV + 15) / 16;   // column tiles of [P1 | P2]
    static constexpr int KS = NVV / 4;
};
static int minv_nvb(int nv) { return nv <= 16 ? 16 : (nv <= 24 ? 24 : (nv <= 32 ? 32 : (nv <= 40 ? 40 : (nv <= 48 ? 48 : 64)))); }
// LDS per workgroup: G tiles, the group's packed right-hand sides and G zero words, the group's record blocks
static size_t minv_lds_bytes(int nv, int n_rhs, int n_entries, size_t elem)
{
    const int nvb = minv_nvb(nv);
    const size_t region = static_cast<size_t>(n_rhs) * nv * nv + 1;  // (+ the zero word of each state)
    const size_t rec = (static_cast<size_t>(n_entries) + 3) / 4 * 4;
    return (static_cast<size_t>(kDerivGroup) * (static_cast<size_t>(nvb) * nvb + region + rec) * elem + 15) / 16 * 16;
}
size_t minv_solve_lds_bytes(int nv, int n_rhs, int n_entries, size_t elem)
{
    return minv_lds_bytes(nv, n_rhs, n_entries, elem);
}

template <class T, int NVV, int NMAX>
__global__ __launch_bounds__(kWave *kDerivGroup)
    __attribute__((amdgpu_waves_per_eu((sizeof(T) == 8 || NVV > 48) ? 1 : 2, (sizeof(T) == 8 || NVV > 32) ? 2 : 3)))
void minv_mfma_kernel(const T *__restrict__ rec, int n_entries, int r_il, const int32_t *__restrict__ coltab, int max_depth, int base_off,
                      const T *P1, const T *P2, int p_il, T *Hinv, T *X1, T *X2, const uint64_t *__restrict__ related, int nv, size_t B)
{
    // Schedule of one group (G states, one per wavefront): every global access is issued a compute phase before it is needed.
    //   wait vmcnt(0), barrier A         the record blocks of this group (copied during the previous group's second product) have landed;
    //                                    every wavefront has finished the previous group's second product (its reads of the right-hand sides)
    //   issue: right-hand sides of THIS group -> LDS (global_load_lds)
    //   clear the tile, walk the columns through the record blocks (LDS), W^T W
    //   wait vmcnt(0), barrier B         everyone's part of the right-hand sides has landed; everyone has finished with the record blocks
    //   issue: record blocks of the NEXT group -> LDS;  H^-1 -> tile (and out);  second product;  result stores
    // Barriers order LDS traffic only (lds_barrier): __syncthreads() would drain the stores in flight.
    // What was measured on the way here (JVRC-1, 1 048 576 states, profiles/r6_minv_experiments.txt): the phases of a group do not overlap
    // chip-wide whatever the schedule -- copies + stores alone take 6.8 ms (3.9 TB/s), the arithmetic alone 6.1 ms, everything 9.5 ms; three
    // workgroups per CU (one right-hand side at a time through half the LDS), records prefetched through registers, workgroups started
    // out of step: all within 3 % of each other.
    using Sh = MinvShape<NVV>;
    constexpr int NT = Sh::NT, WS = Sh::WS, NCT = Sh::NCT, KS = Sh::KS, G = kDerivGroup;
    constexpr int kReach = 5;
    typedef typename Mfma<T>::Acc Acc;
    static_assert(NVV % 8 == 0, "sizes");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
    const int nn = nv * nv;
    const int n_mat = (P1 ? 1 : 0) + (P2 ? 1 : 0);
    const int n_cols = n_mat * nv, nct = (n_cols + 15) / 16;
    const size_t region = static_cast<size_t>(n_mat) * nn + 1;
    T *A = reinterpret_cast<T *>(grbda_smem) + wave * (NVV * WS);  // this wavefront's tile: W, then H^-1
    T *Pg = reinterpret_cast<T *>(grbda_smem) + G * (NVV * WS);    // the group's right-hand sides, [matrix][entry][G], then G zero words
    T *Rg = Pg + G * region;                                       // the group's record blocks, [entry][G]
    const int zero_at = static_cast<int>(G * (region - 1));        // (element index in Pg of the first of the G zero words)
    const T *src[2] = {P1 ? P1 : P2, P1 ? P2 : nullptr};
    T *const dst0 = P1 ? X1 : X2, *const dst1 = P1 ? X2 : nullptr;
    // (both workspaces are interleaved by groups of G states: element e of the wavefront's state sits at e * G + wave)
    constexpr int r_es = G, p_es = G, p_ss = 1;
    (void)r_il;
    (void)p_il;
    // the four wavefronts copy `count` scalars from `from` to LDS at `to`
    auto group_copy = [&](const T *from, T *to, size_t count) {
        const unsigned *blk = reinterpret_cast<const unsigned *>(from);
        const int dw = static_cast<int>(count * (sizeof(T) / 4));
        unsigned *to_dw = reinterpret_cast<unsigned *>(to);
        if ((dw & 3) == 0 && (reinterpret_cast<uintptr_t>(from) & 15) == 0) {
            for (int base = wave * 4 * kWave; base < dw; base += G * 4 * kWave)
                if (base + 4 * lane < dw)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(blk + base + 4 * lane),
                                                     (__attribute__((address_space(3))) void *)(to_dw + base), 16, 0, 0);
        } else {
            for (int base = wave * kWave; base < dw; base += G * kWave)
                if (base + lane < dw)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(blk + base + lane),
                                                     (__attribute__((address_space(3))) void *)(to_dw + base), 4, 0, 0);
        }
    };
    // ---- once per kernel: where this lane's right-hand-side fragment elements sit in LDS (state-independent), two 16-bit element
    // indices per register ----
    constexpr int NG = (KS * NCT + 1) / 2;
    unsigned gpk[NG];
    {
#pragma unroll
        for (int i = 0; i < NG; i++) gpk[i] = 0;
        uint64_t relc[NCT];   // related[] of this lane's column of every column tile
        int mcol[NCT], ccol[NCT];
#pragma unroll
        for (int t = 0; t < NCT; t++) {
            const int c = 16 * t + c16;
            const int m = c >= nv ? 1 : 0;
            mcol[t] = m;
            ccol[t] = c - m * nv;
            relc[t] = (n_mat && c < n_cols) ? (related ? related[ccol[t]] : ~uint64_t(0)) : uint64_t(0);
        }
#pragma unroll
        for (int k = 0; k < KS; k++)
#pragma unroll
            for (int t = 0; t < NCT; t++) {
                const int r = 4 * k + g, cc = ccol[t];
                int e = zero_at;
                if (r < nv && ((relc[t] >> r) & 1)) e = mcol[t] * G * nn + (cc <= r ? r * r + cc : cc * cc + cc + 1 + r) * p_es;
                const int idx = k * NCT + t;
                gpk[idx / 2] |= static_cast<unsigned>(e + wave * p_ss) << (16 * (idx & 1));
            }
#pragma unroll
        for (int i = 0; i < NG; i++) asm volatile("" : "+v"(gpk[i]));  // (kept, not recomputed inside the loop over the groups)
        if (threadIdx.x < G) Pg[zero_at + threadIdx.x] = T(0);
    }
    // ---- once per kernel: this lane's column program ----
    int own_k, own_l, own_m, st[kMinvMaxDepth];
    {
        const int32_t *col = coltab + lane * kMinvColInts;
        own_k = col[0];
        own_l = col[1];
        own_m = col[2];
#pragma unroll
        for (int t = 0; t < kMinvMaxDepth; t++) st[t] = col[3 + t];
    }
    const size_t n_groups = (B + G - 1) / G;
#ifdef GRBDA_EXP_MV_PROF
    unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_t = __builtin_amdgcn_s_memtime();
#endif
    // prologue: the first group's record blocks
    if (blockIdx.x < n_groups && !MV_ABL(64)) group_copy(rec + blockIdx.x * (size_t)G * n_entries, Rg, (size_t)G * n_entries);
    for (size_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const size_t s = grp * G + wave;
        const bool live = s < B;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();  // barrier A
        if (!MV_ABL(16))
            for (int m = 0; m < n_mat; m++) group_copy(src[m] + grp * (size_t)G * nn, Pg + (size_t)m * G * nn, (size_t)G * nn);
        const size_t next = grp + gridDim.x;
        // clear the tile (16-byte stores, the tile is NVV * WS scalars)
        {
            constexpr int V = 16 / (int)sizeof(T);
            typedef T TV __attribute__((ext_vector_type(V)));
            TV z;
#pragma unroll
            for (int e = 0; e < V; e++) z[e] = T(0);
            for (int i = lane * V; i < NVV * WS; i += kWave * V) *reinterpret_cast<TV *>(&A[i]) = z;
        }
        wave_lds_fence();
        MV_STAMP(0)
        // ---- 1. W: lane j walks column j ----
        if (!MV_ABL(4)) {
            const T *R = Rg + wave;
            const bool valid = (own_m >> 21) & 1;
            const int n_own = own_m & 15, e_own = (own_m >> 4) & 15, v_own = (own_m >> 8) & 63;
            const bool is_base_col = valid && base_off >= 0 && own_l == base_off;
            T f[6];
#pragma unroll
            for (int i = 0; i < 6; i++) f[i] = (valid && !is_base_col) ? R[(own_k + i) * r_es] : T(0);
            // own cluster: W[v + i][j] = L^-1[i][e], i >= e
#pragma unroll
            for (int i = 0; i < 6; i++)
                if (valid && i >= e_own && i < n_own) A[(v_own + i) * WS + lane] = R[(own_l + i * (i + 1) / 2 + e_own) * r_es];
#pragma unroll
            for (int t = 0; t < kMinvMaxDepth; t++) {
                const unsigned d1 = (unsigned)st[t];
                if (t < max_depth && (d1 >> 31)) {
                    const int na = NMAX == 1 ? 1 : (int)((d1 >> 16) & 15), va = (d1 >> 20) & 63;
                    const int ko = d1 & 0xffff, lo = ko + 6 * na, sab = lo + na * (na + 1) / 2;
                    T sig[NMAX];
#pragma unroll
                    for (int a2 = 0; a2 < NMAX; a2++) {
                        sig[a2] = 0;
                        if (a2 < na) {
#pragma unroll
                            for (int i = 0; i < 6; i++) sig[a2] += R[(sab + 6 * a2 + i) * r_es] * f[i];
                        }
                    }
#pragma unroll
                    for (int a2 = 0; a2 < NMAX; a2++)
                        if (a2 < na) {
                            T w = 0;
#pragma unroll
                            for (int m2 = 0; m2 < NMAX; m2++)
                                if (m2 <= a2) w -= R[(lo + a2 * (a2 + 1) / 2 + m2) * r_es] * sig[m2];
                            A[(va + a2) * WS + lane] = w;
#pragma unroll
                            for (int i = 0; i < 6; i++) f[i] -= R[(ko + 6 * a2 + i) * r_es] * sig[a2];
                        }
                }
            }
            if (valid && !is_base_col && ((own_m >> 20) & 1)) {
                // the floating base: sigma = f, W[i][j] = -(L_b^-1 f)_i
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    T w = 0;
#pragma unroll
                    for (int m2 = 0; m2 <= i; m2++) w -= R[(base_off + i * (i + 1) / 2 + m2) * r_es] * f[m2];
                    A[i * WS + lane] = w;
                }
            }
        }
        wave_lds_fence();
        MV_STAMP(1)
        // ---- 2. H^-1 = W^T W (rows and columns nv .. NVV - 1 of the tile are zero) ----
        Acc hi[NT][NT];
#pragma unroll
        for (int a = 0; a < NT; a++)
#pragma unroll
            for (int b = 0; b < NT; b++)
#pragma unroll
                for (int j = 0; j < 4; j++) hi[a][b][j] = T(0);
#pragma unroll
        for (int k = 0; k < KS; k++) {
            if (MV_ABL(32)) break;
            T w[NT];
#pragma unroll
            // (row r of W reaches the columns from the first coordinate of r's own cluster on -- L^-1 is LOWER triangular inside the
            // cluster's block, at most kReach = 5 columns to the left of r (the base's six) -- and nothing before them)
            for (int t = 0; t < NT; t++) w[t] = (16 * t + 15 + kReach >= 4 * k) ? A[(4 * k + g) * WS + 16 * t + c16] : T(0);
#pragma unroll
            for (int a = 0; a < NT; a++)
#pragma unroll
                for (int b = 0; b < NT; b++)
                    if (16 * (a < b ? a : b) + 15 + kReach >= 4 * k) hi[a][b] = Mfma<T>::run(w[a], w[b], hi[a][b]);
        }
        MV_STAMP(2)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();  // barrier B
        MV_STAMP(3)
#pragma unroll
        for (int a = 0; a < NT; a++)
#pragma unroll
            for (int b = 0; b < NT; b++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int row = 16 * a + Mfma<T>::row(g, j), col = 16 * b + c16;
                    if (row < NVV && col < NVV) A[row * WS + col] = hi[a][b][j];
                }
        if (next < n_groups && !MV_ABL(64)) group_copy(rec + next * (size_t)G * n_entries, Rg, (size_t)G * n_entries);
        if (Hinv && live && !MV_ABL(2)) {
            T *hout = Hinv + s * (size_t)nn;
            if constexpr (sizeof(T) == 4) {
                // H^-1 is symmetric: the four values a lane holds of tile (a, b) -- rows 16 a + 4 g + 0..3 of column 16 b + c16 -- are
                // also columns 16 a + 4 g + 0..3 of ROW 16 b + c16: one 16-byte store instead of four 4-byte ones
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
            } else {
                // f64 accumulators hold rows 4 r + g of column c16: consecutive lanes of a group write consecutive columns of one row
#pragma unroll
                for (int a = 0; a < NT; a++)
#pragma unroll
                    for (int b = 0; b < NT; b++)
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int row = 16 * a + Mfma<T>::row(g, j), col = 16 * b + c16;
                            if (row < nv && col < nv) hout[(size_t)row * nv + col] = hi[a][b][j];
                        }
            }
        }
        MV_STAMP(4)
        if (n_mat == 0) continue;
        wave_lds_fence();  // this wavefront's H^-1 is in its tile
        // ---- 3. X = -H^-1 [P1 | P2], taken transposed: (P^T H^-1)^T -- the right-hand-side fragment is the A operand ----
        Acc acc[NT][NCT];
#pragma unroll
        for (int a = 0; a < NT; a++)
#pragma unroll
            for (int t = 0; t < NCT; t++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[a][t][j] = T(0);
#pragma unroll
        for (int k = 0; k < KS; k++) {
            if (MV_ABL(8)) break;
            const int r = 4 * k + g;
            T av[NT], bv[NCT];
#pragma unroll
            for (int a = 0; a < NT; a++) av[a] = A[r * WS + 16 * a + c16];  // H^-1[a][r] = H^-1[r][a]
#pragma unroll
            for (int t = 0; t < NCT; t++)
                if (t < nct) {
                    const int idx = k * NCT + t;
                    const unsigned e = (idx & 1) ? (gpk[idx / 2] >> 16) : (gpk[idx / 2] & 0xffffu);
                    bv[t] = Pg[e];
                }
#pragma unroll
            for (int t = 0; t < NCT; t++)
                if (t < nct) {
#pragma unroll
                    for (int a = 0; a < NT; a++) acc[a][t] = Mfma<T>::run(bv[t], av[a], acc[a][t]);
                }
        }
        MV_STAMP(5)
        // accumulator (a, t): X[row 16 a + c16][columns 16 t + Mfma::row(g, 0..3)] of [X1 | X2]
        if (live && !MV_ABL(1)) {
#pragma unroll
            for (int t = 0; t < NCT; t++)
                if (t < nct) {
#pragma unroll
                    for (int a = 0; a < NT; a++) {
                        const int row = 16 * a + c16;
                        if (row >= nv) continue;
                        if constexpr (sizeof(T) == 4) {
                            const int c0 = 16 * t + 4 * g;  // first of this lane's four columns of [X1 | X2]
                            const f32x4m v = -acc[a][t];
                            const bool second = c0 >= nv;
                            const int cc0 = second ? c0 - nv : c0;
                            T *o0 = (second ? dst1 : dst0) + s * (size_t)nn + (size_t)row * nv + cc0;
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
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const int c = 16 * t + Mfma<T>::row(g, j);
                                if (c < n_cols) {
                                    T *o = (c >= nv ? dst1 : dst0) + s * (size_t)nn + (size_t)row * nv + (c >= nv ? c - nv : c);
                                    *o = -acc[a][t][j];
                                }
                            }
                        }
                    }
                }
        }
        MV_STAMP(6)
    }
#ifdef GRBDA_EXP_MV_PROF
    if (lane == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&mv_prof[i], prof_acc[i]);
#endif
}

template <class T, int NVV>
static hipError_t launch_minv_n(const T *rec, int n_entries, int r_il, const int32_t *coltab, int max_depth, int base_off, int n_max,
                                const T *P1, const T *P2, int p_il, T *Hinv, T *X1, T *X2, const uint64_t *related, int nv, size_t B, int grid,
                                hipStream_t stream)
{
    if (r_il != kDerivGroup || ((P1 || P2) && p_il != kDerivGroup)) return hipErrorInvalidValue;
    const size_t lds = minv_lds_bytes(nv, (P1 ? 1 : 0) + (P2 ? 1 : 0), n_entries, sizeof(T));
    if (n_max <= 1)
        hipLaunchKernelGGL((minv_mfma_kernel<T, NVV, 1>), dim3(grid), dim3(kWave * kDerivGroup), lds, stream, rec, n_entries, r_il, coltab,
                           max_depth, base_off, P1, P2, p_il, Hinv, X1, X2, related, nv, B);
    else if (n_max <= 2)
        hipLaunchKernelGGL((minv_mfma_kernel<T, NVV, 2>), dim3(grid), dim3(kWave * kDerivGroup), lds, stream, rec, n_entries, r_il, coltab,
                           max_depth, base_off, P1, P2, p_il, Hinv, X1, X2, related, nv, B);
    else
        hipLaunchKernelGGL((minv_mfma_kernel<T, NVV, kMaxClusterDof>), dim3(grid), dim3(kWave * kDerivGroup), lds, stream, rec, n_entries, r_il,
                           coltab, max_depth, base_off, P1, P2, p_il, Hinv, X1, X2, related, nv, B);
    return hipGetLastError();
}
// workgroups of minv_mfma_kernel a CU really holds (registers of the instantiation the launch picks AND the LDS granules): a persistent
// grid one workgroup per CU too large leaves that workgroup waiting until another has finished all its groups
template <class T, int NVV>
static int minv_wpc_n(int n_max, size_t lds)
{
    int n = 0;
    hipError_t e;
    if (n_max <= 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, minv_mfma_kernel<T, NVV, 1>, kWave * kDerivGroup, lds);
    else if (n_max <= 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, minv_mfma_kernel<T, NVV, 2>, kWave * kDerivGroup, lds);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, minv_mfma_kernel<T, NVV, kMaxClusterDof>, kWave * kDerivGroup, lds);
    if (e != hipSuccess || n < 1) n = 1;
    const size_t by_lds = lds_workgroups_per_cu(lds);
    if (static_cast<size_t>(n) > by_lds) n = static_cast<int>(by_lds);
    return n < 1 ? 1 : n;
}
template <class T>
int minv_workgroups_per_cu(int nv, int n_max, int n_rhs, int n_entries)
{
    const size_t lds = minv_lds_bytes(nv, n_rhs, n_entries, sizeof(T));
    if (nv <= 16) return minv_wpc_n<T, 16>(n_max, lds);
    if (nv <= 24) return minv_wpc_n<T, 24>(n_max, lds);
    if (nv <= 32) return minv_wpc_n<T, 32>(n_max, lds);
    if (nv <= 40) return minv_wpc_n<T, 40>(n_max, lds);
    if (nv <= 48) return minv_wpc_n<T, 48>(n_max, lds);
    return minv_wpc_n<T, 64>(n_max, lds);
}
template int minv_workgroups_per_cu<float>(int, int, int, int);
template int minv_workgroups_per_cu<double>(int, int, int, int);

template <class T>
hipError_t launch_minv_solve(const T *rec, int n_entries, int r_il, const int32_t *coltab, int max_depth, int base_off, int n_max, const T *P1,
                             const T *P2, int p_il, T *Hinv, T *X1, T *X2, const uint64_t *related, int nv, size_t B, int grid,
                             hipStream_t stream)
{
#define GRBDA_MINV_CASE(N)                                                                                                               \
    if (nv <= N)                                                                                                                        \
        return launch_minv_n<T, N>(rec, n_entries, r_il, coltab, max_depth, base_off, n_max, P1, P2, p_il, Hinv, X1, X2, related, nv, B, grid, \
                                   stream);
    GRBDA_MINV_CASE(16)
    GRBDA_MINV_CASE(24)
    GRBDA_MINV_CASE(32)
    GRBDA_MINV_CASE(40)
    GRBDA_MINV_CASE(48)
    GRBDA_MINV_CASE(64)
#undef GRBDA_MINV_CASE
    return hipErrorInvalidValue;
}
template hipError_t launch_minv_solve<float>(const float *, int, int, const int32_t *, int, int, int, const float *, const float *, int, float *,
                                             float *, float *, const uint64_t *, int, size_t, int, hipStream_t);
template hipError_t launch_minv_solve<double>(const double *, int, int, const int32_t *, int, int, int, const double *, const double *, int,
                                              double *, double *, double *, const uint64_t *, int, size_t, int, hipStream_t);

template <class T>
static hipError_t set_lds_for()
{
    const void *fns[] = {
        reinterpret_cast<const void *>(&minv_mfma_kernel<T, 16, 1>), reinterpret_cast<const void *>(&minv_mfma_kernel<T, 16, 2>),
        reinterpret_cast<const void *>(&minv_mfma_kernel<T, 16, kMaxClusterDof>), reinterpret_cast<const void *>(&minv_mfma_kernel<T, 24, 1>),
        reinterpret_cast<const void *>(&minv_mfma_kernel<T, 24, 2>), reinterpret_cast<const void *>(&minv_mfma_kernel<T, 24, kMaxClusterDof>),
        reinterpret_cast<const void *>(&minv_mfma_kernel<T, 32, 1>), reinterpret_cast<const void *>(&minv_mfma_kernel<T, 32, 2>),
        reinterpret_cast<const void *>(&minv_mfma_kernel<T, 32, kMaxClusterDof>), reinterpret_cast<const void *>(&minv_mfma_kernel<T, 40, 1>),
        reinterpret_cast<const void *>(&minv_mfma_kernel<T, 40, 2>), reinterpret_cast<const void *>(&minv_mfma_kernel<T, 40, kMaxClusterDof>),
        reinterpret_cast<const void *>(&minv_mfma_kernel<T, 48, 1>), reinterpret_cast<const void *>(&minv_mfma_kernel<T, 48, 2>),
        reinterpret_cast<const void *>(&minv_mfma_kernel<T, 48, kMaxClusterDof>), reinterpret_cast<const void *>(&minv_mfma_kernel<T, 64, 1>),
        reinterpret_cast<const void *>(&minv_mfma_kernel<T, 64, 2>), reinterpret_cast<const void *>(&minv_mfma_kernel<T, 64, kMaxClusterDof>)};
    for (const void *f : fns) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
// up to 160 KiB of dynamic LDS: raised per device by capi.cpp's ensure_device, like every other kernel
hipError_t set_max_dynamic_lds_minv()
{
    hipError_t e = set_lds_for<float>();
    if (e != hipSuccess) return e;
    return set_lds_for<double>();
}

}  // namespace grbda_hip
