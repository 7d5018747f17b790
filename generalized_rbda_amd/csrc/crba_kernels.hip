// crba_kernels.hip -- batched joint-space inertia matrix H(q) by the composite-rigid-body algorithm on the cluster
// tree: ClusterTreeModel::getMassMatrix = TreeModel::compositeRigidBodyAlgorithm (src/Dynamics/TreeModel.cpp:115-160,
// ClusterTreeModel.cpp:98-103).  One state per lane like the dynamics kernels.
//
// Structured restatement (the reference works on dense 6k x 6k cluster blocks): per BODY of the spanning tree, in
// reverse topological order,
//   Ic_i   = I_i + sum_children X_c^T Ic_c X_c                         composite inertia
//   f      = Ic_i s_i                                                    force per unit spanning acceleration of joint i
//   walk up the tree: at every ancestor joint j, H_span[i][j] = s_j^T f, then f <- X_j^T f
// and H = G^T H_span G is accumulated on the fly: within a cluster the n x n diagonal block; at the cluster's parent
// body the 6 x n matrix Fp = sum_i X^T f_i G_i (the composite counterpart of the F of the ABA handlers), which then
// walks up the ancestors once for the whole cluster.  Axisymmetric rotors (plan.cpp) contribute the constant X0^T I X0
// to their parent through the parent's precomputed inertia constants, as in the dynamics kernels.
// Explicit (constant G) models only; models with implicit-loop clusters keep the nv + 1 inverse-dynamics evaluations of
// capi.cpp (derived(), DM_MASS).
#include <hip/hip_runtime.h>

#include "devplan.h"

namespace grbda_hip {

#include "devmath.h"

// IL: interleave factor of the PACKED result ([group of IL states][entry][IL], as rnea_deriv_kernel's workspace: with one state per
// lane a store of the state-major layout opens 64 cache lines that do not survive in L2 until their other entries arrive --
// deriv_kernels.hip; the matrix-core solve and unpack_symmetric_kernel read the interleaved form)
template <class T, int IL>
__global__ __launch_bounds__(kWave, 1) void crba_kernel(DevPlan<T> DP, const CrbaBody *__restrict__ cb_, int n_clusters, int n_rows,
                                                        const T *__restrict__ q, T *__restrict__ H, size_t B,
                                                        T *__restrict__ scratch, int packed)
{
    cptr<ClusterRec> clusters = (cptr<ClusterRec>)DP.clusters;
    cptr<BodyRec> bodies = (cptr<BodyRec>)DP.bodies;
    cptr<T> consts = (cptr<T>)DP.consts;
    cptr<CrbaBody> cb = (cptr<CrbaBody>)cb_;
    const int lane = threadIdx.x, nq = DP.nq, nv = DP.nv;
    T *slab = scratch + (size_t)blockIdx.x * (size_t)n_rows * kWave + lane;  // row r of this lane: slab[r * kWave]
    const size_t n_tiles = (B + kWave - 1) / kWave;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t r = tile * kWave + lane;
        const size_t st = r < B ? r : B - 1;  // lanes past the end redo the last state and do not store
        const bool live = r < B;
        const T *qs = q + st * (size_t)nq;
        T *Hs = H + (st / IL) * (size_t)nv * nv * IL + st % IL;
        // packed: the rows of the lower triangle back to back (entry (r, c <= r) at r (r + 1) / 2 + c), every store of a
        // coordinate's pass in that coordinate's own row -- what spd_solve_kernel and unpack_symmetric_kernel read; the
        // transposed stores of the plain layout revisit every ancestor's row from every descendant.  Plain: the full
        // symmetric nv x nv matrix, structural zeros left to the caller.
        auto put = [&](int r, int c, T v) {
            if (!packed) Hs[(size_t)r * nv + c] = v;  // (IL == 1)
            else if (c <= r) Hs[(size_t)(r * (r + 1) / 2 + c) * IL] = v;
        };
        // ---- pass 1: sin / cos of every revolute spanning joint; composite accumulators start at zero ----
        for (int c = 0; c < n_clusters; c++) {
            const ClusterRec cr = load_rec(clusters + c);
            if (cr.kind == CK_FREE) continue;
            for (int i = 0; i < cr.k; i++) {
                const int gb = cr.first_body + i;
                const BodyRec b = load_rec(bodies + gb);
                const CrbaBody x = load_rec(cb + gb);
                cptr<T> C = consts + b.cofs;
                T qi = 0;
                for (int a = 0; a < cr.n; a++) qi += C[kBodyConstFixed + a] * qs[cr.q_index + a];
                T sn, cs;
                if (b.axisym) { sn = 0; cs = 1; }  // rotors are evaluated at q = 0 (plan.cpp)
                else sincos_t(qi, &sn, &cs);
                slab[(size_t)x.sc_row * kWave] = sn;
                slab[(size_t)(x.sc_row + 1) * kWave] = cs;
            }
        }
        for (int c = 0; c < n_clusters; c++) {
            const ClusterRec cr = load_rec(clusters + c);
            for (int i = 0; i < cr.k; i++) {
                const CrbaBody x = load_rec(cb + (cr.first_body + i));
                if (x.acc_row >= 0)
                    for (int j = 0; j < 21; j++) slab[(size_t)(x.acc_row + j) * kWave] = 0;
            }
        }
        // ---- pass 2: clusters leaf side first ----
        for (int c = n_clusters - 1; c >= 0; c--) {
            const ClusterRec cr = load_rec(clusters + c);
            if (cr.kind == CK_FREE) {
                // H[base][base] = Ic of the base (S = 1)
                const BodyRec b = load_rec(bodies + cr.first_body);
                const CrbaBody x = load_rec(cb + cr.first_body);
                cptr<T> Ib = b.xofs >= 0 ? consts + b.xofs : consts + b.cofs + 12;
                for (int i = 0; i < 6; i++)
                    for (int j = 0; j < 6; j++) {
                        T v = Ib[sidx(i, j)];
                        if (x.acc_row >= 0) v += slab[(size_t)(x.acc_row + sidx(i, j)) * kWave];
                        if (live) put(cr.v_index + i, cr.v_index + j, v);
                    }
                continue;
            }
            const int n = cr.n;
            T Fp[kMaxClusterDof][6];      // force at the parent body per unit acceleration of coordinate a
            T Hcc[kMaxClusterDof][kMaxClusterDof];
#pragma unroll
            for (int a = 0; a < kMaxClusterDof; a++) {
#pragma unroll
                for (int j = 0; j < 6; j++) Fp[a][j] = 0;
#pragma unroll
                for (int b2 = 0; b2 < kMaxClusterDof; b2++) Hcc[a][b2] = 0;
            }
            for (int i = cr.k - 1; i >= 0; i--) {
                const int gb = cr.first_body + i;
                const BodyRec b = load_rec(bodies + gb);
                const CrbaBody x = load_rec(cb + gb);
                cptr<T> C = consts + b.cofs;
                cptr<T> Ib = b.xofs >= 0 ? consts + b.xofs : C + 12;
                T Ic[21];
#pragma unroll
                for (int j = 0; j < 21; j++) Ic[j] = Ib[j];
                if (x.acc_row >= 0) {
#pragma unroll
                    for (int j = 0; j < 21; j++) Ic[j] += slab[(size_t)(x.acc_row + j) * kWave];
                }
                T E[9];
                rotate_z(slab[(size_t)x.sc_row * kWave], slab[(size_t)(x.sc_row + 1) * kWave], C, E);
                // composite inertia to the tree parent (axisymmetric leaves are already part of the parent's constants)
                if (b.parent >= 0 && !b.axisym) {
                    const CrbaBody xp = load_rec(cb + b.parent);
                    T Bc[21];
                    congruence(E, C + 9, Ic, Bc);
#pragma unroll
                    for (int j = 0; j < 21; j++) slab[(size_t)(xp.acc_row + j) * kWave] += Bc[j];
                }
                T Gi[kMaxClusterDof];
#pragma unroll
                for (int a = 0; a < kMaxClusterDof; a++) Gi[a] = a < n ? C[kBodyConstFixed + a] : T(0);
                T f[6];
#pragma unroll
                for (int j = 0; j < 6; j++) f[j] = Ic[sidx(j, 2)];
#pragma unroll
                for (int a = 0; a < kMaxClusterDof; a++)
#pragma unroll
                    for (int b2 = 0; b2 < kMaxClusterDof; b2++) Hcc[a][b2] += Gi[a] * Gi[b2] * f[2];
                // up the in-cluster chain
                T f2[6];
                xforce_inv(E, C + 9, f, f2);
                int l = b.lam;
                while (l >= 0) {
                    const BodyRec bl = load_rec(bodies + l);
                    const CrbaBody xl = load_rec(cb + l);
                    cptr<T> Cl = consts + bl.cofs;
                    T Gl[kMaxClusterDof];
#pragma unroll
                    for (int a = 0; a < kMaxClusterDof; a++) Gl[a] = a < n ? Cl[kBodyConstFixed + a] : T(0);
#pragma unroll
                    for (int a = 0; a < kMaxClusterDof; a++)
#pragma unroll
                        for (int b2 = 0; b2 < kMaxClusterDof; b2++) Hcc[a][b2] += f2[2] * (Gi[a] * Gl[b2] + Gl[a] * Gi[b2]);
                    T El[9], f3[6];
                    rotate_z(slab[(size_t)xl.sc_row * kWave], slab[(size_t)(xl.sc_row + 1) * kWave], Cl, El);
                    xforce_inv(El, Cl + 9, f2, f3);
#pragma unroll
                    for (int j = 0; j < 6; j++) f2[j] = f3[j];
                    l = bl.lam;
                }
#pragma unroll
                for (int a = 0; a < kMaxClusterDof; a++)
#pragma unroll
                    for (int j = 0; j < 6; j++) Fp[a][j] += f2[j] * Gi[a];
            }
            if (live) {
#pragma unroll
                for (int a = 0; a < kMaxClusterDof; a++)
#pragma unroll
                    for (int b2 = 0; b2 < kMaxClusterDof; b2++)
                        if (a < n && b2 < n) put(cr.v_index + a, cr.v_index + b2, Hcc[a][b2]);
            }
            // ---- up the ancestors: H[c][d] = Fp^T S_d, block by block ----
            int j = cr.parent_body;
            while (j >= 0) {
                const BodyRec bj = load_rec(bodies + j);
                const CrbaBody xj = load_rec(cb + j);
                const ClusterRec cd = load_rec(clusters + xj.cluster);
                if (cd.kind == CK_FREE) {
                    if (live) {
#pragma unroll
                        for (int a = 0; a < kMaxClusterDof; a++)
#pragma unroll
                            for (int k6 = 0; k6 < 6; k6++)
                                if (a < n) {
                                    put(cr.v_index + a, cd.v_index + k6, Fp[a][k6]);
                                    put(cd.v_index + k6, cr.v_index + a, Fp[a][k6]);
                                }
                    }
                    break;
                }
                // all bodies of cluster d on the path are consecutive ancestors: accumulate the block, then store it
                T Hcd[kMaxClusterDof][kMaxClusterDof];
#pragma unroll
                for (int a = 0; a < kMaxClusterDof; a++)
#pragma unroll
                    for (int b2 = 0; b2 < kMaxClusterDof; b2++) Hcd[a][b2] = 0;
                int jj = j;
                int next = -1;
                for (;;) {
                    const BodyRec bb = load_rec(bodies + jj);
                    const CrbaBody xb = load_rec(cb + jj);
                    cptr<T> Cj = consts + bb.cofs;
                    T Ej[9];
                    rotate_z(slab[(size_t)xb.sc_row * kWave], slab[(size_t)(xb.sc_row + 1) * kWave], Cj, Ej);
#pragma unroll
                    for (int a = 0; a < kMaxClusterDof; a++) {
#pragma unroll
                        for (int b2 = 0; b2 < kMaxClusterDof; b2++)
                            if (b2 < cd.n) Hcd[a][b2] += Fp[a][2] * Cj[kBodyConstFixed + b2];
                        T fn[6];
                        xforce_inv(Ej, Cj + 9, Fp[a], fn);
#pragma unroll
                        for (int k6 = 0; k6 < 6; k6++) Fp[a][k6] = fn[k6];
                    }
                    next = bb.parent;
                    if (bb.lam < 0) break;  // left the cluster
                    jj = bb.lam;
                }
                if (live) {
#pragma unroll
                    for (int a = 0; a < kMaxClusterDof; a++)
#pragma unroll
                        for (int b2 = 0; b2 < kMaxClusterDof; b2++)
                            if (a < n && b2 < cd.n) {
                                put(cr.v_index + a, cd.v_index + b2, Hcd[a][b2]);
                                put(cd.v_index + b2, cr.v_index + a, Hcd[a][b2]);
                            }
                }
                (void)bj;
                j = next;
            }
        }
        // entries between clusters on different branches are structural zeros of H
    }
}

template <class T>
hipError_t launch_crba(const DevPlan<T> &P, const CrbaBody *cb, int n_clusters, int n_rows, const T *q, T *H, size_t B, T *scratch,
                       int grid, hipStream_t stream, bool packed, int interleave)
{
    if (interleave == kDerivGroup && packed)
        hipLaunchKernelGGL((crba_kernel<T, kDerivGroup>), dim3(grid), dim3(kWave), 0, stream, P, cb, n_clusters, n_rows, q, H, B, scratch, 1);
    else if (interleave == 1)
        hipLaunchKernelGGL((crba_kernel<T, 1>), dim3(grid), dim3(kWave), 0, stream, P, cb, n_clusters, n_rows, q, H, B, scratch, packed ? 1 : 0);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}
template hipError_t launch_crba<float>(const DevPlan<float> &, const CrbaBody *, int, int, const float *, float *, size_t, float *, int,
                                       hipStream_t, bool, int);
template hipError_t launch_crba<double>(const DevPlan<double> &, const CrbaBody *, int, int, const double *, double *, size_t,
                                        double *, int, hipStream_t, bool, int);

// Packed lower triangle -> full symmetric matrix, in place (the packed rows occupy the first nv (nv + 1) / 2 entries of each
// state's nv x nv block; interleaved by il: the first il nv (nv + 1) / 2 entries of each GROUP's il nv^2 block -- B a multiple of
// il).  One group per wavefront: the packed block is read into LDS with consecutive lanes on consecutive addresses, then the
// il nv^2 entries are written the same way; entries between coordinates that are not on one root path (DerivProgram::related) are
// written as zeros, whatever the packed block holds there.  nv <= 64.
template <class T>
__global__ __launch_bounds__(kWave) void unpack_symmetric_kernel(T *__restrict__ H, const uint64_t *__restrict__ related, int nv, size_t B, int il)
{
    T *tri = reinterpret_cast<T *>(grbda_smem);  // [il * nt], as it lies in memory
    __shared__ uint64_t rel_rows[kWave];  // the masks, once per workgroup (a global load per entry otherwise)
    const int lane = threadIdx.x, nn = nv * nv, nt = nv * (nv + 1) / 2;
    const int step_r = kWave / nv, step_c = kWave % nv;
    rel_rows[lane] = lane < nv ? related[lane] : 0;
    const size_t n_groups = B / il;
    for (size_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        T *Hg = H + g * (size_t)nn * il;
        __syncthreads();
        for (int i = lane; i < nt * il; i += kWave) tri[i] = Hg[i];
        __syncthreads();
        for (int s = 0; s < il; s++) {
            T *Hs = Hg + (size_t)s * nn;
            int r = lane / nv, c = lane % nv;
            for (int i = lane; i < nn; i += kWave) {
                const int lo = c <= r ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r;
                const bool rel = (rel_rows[r] >> c) & 1;
                Hs[i] = rel ? tri[lo * il + s] : T(0);
                r += step_r;
                c += step_c;
                if (c >= nv) {
                    c -= nv;
                    r++;
                }
            }
        }
    }
}
size_t unpack_symmetric_lds_bytes(int nv, size_t elem, int il) { return static_cast<size_t>(il) * nv * (nv + 1) / 2 * elem; }
template <class T>
hipError_t launch_unpack_symmetric(T *H, const uint64_t *related, int nv, size_t B, int grid, hipStream_t stream, int il)
{
    if (nv > kWave || !related || il < 1 || B % il != 0) return hipErrorInvalidValue;
    const size_t lds = unpack_symmetric_lds_bytes(nv, sizeof(T), il);
    if (lds > 60 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL((unpack_symmetric_kernel<T>), dim3(grid), dim3(kWave), lds, stream, H, related, nv, B, il);
    return hipGetLastError();
}
template hipError_t launch_unpack_symmetric<float>(float *, const uint64_t *, int, size_t, int, hipStream_t, int);
template hipError_t launch_unpack_symmetric<double>(double *, const uint64_t *, int, size_t, int, hipStream_t, int);

}  // namespace grbda_hip
