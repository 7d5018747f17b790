// gen_rnea_segments.h -- generic clusters (plan.h, ChainGen / ChainGenBody; gen_segments.h) in the inverse-dynamics chain program:
// TreeModel::recursiveNewtonEulerAlgorithm (src/Dynamics/TreeModel.cpp:34-57,173-212) inside one cluster of k revolute bodies
// whose spanning rates are qd_s = G yd, qdd_s = G ydd + g (GenericJoint.cpp:380-450).  Included by chain_kernels.hip inside
// namespace grbda_hip, after gen_segments.h and the run / pair / differential segments of the inverse dynamics.
// Field use of the shared records in THIS program: ChainGen::lds_w = work area of the forward segment ([sin, cos] x k | [v 6][a 6] of
// the bodies with in-cluster children -- ChainGenBody::acc_w = offset of a body's own pair, up_w = of its in-cluster parent's; the
// constraint evaluation's scratch aliases them; up_w = -1 and lam = the body right before: the pair comes over in registers), glb_k =
// the blocks [f 6][sin, cos] x k the forward segment
// leaves for the backward one (child segments add their forces into the f part), lds_acc_out = force slot of the parent body,
// lds_pva = its [v 6][a 6]; ChainGenBody::lds_va = [v 6][a 6] of a body with child clusters.
// ChainGen::reserved[1] = 1: the cluster is the whole model (plan.cpp): ONE LDS object [sin, cos 2k][f 6 x k][pairs][kept block], the
// backward segment reads [sin, cos] from the work area and the constraint's scratch lies over the forces and pairs.
#pragma once

// forward segment: kinematics, the constraint of an implicit cluster, body forces f = I a + v x* I v
template <class T, int N, bool LOOP, class TB, class MM>
__device__ __forceinline__ void gen_rnea_fwd(const TB &P, const MM &M, const ChainGen &g)
{
    const int k = g.k;
    const int sc0 = g.lds_w, v0 = g.lds_w + 2 * k;
    T y[N], yd[N], ydd[N];
#pragma unroll
    for (int a = 0; a < N; a++) {
        yd[a] = M.qd(g.v_index + a);
        ydd[a] = M.x(g.v_index + a);
        y[a] = LOOP ? T(0) : M.q(g.q_index + a);
    }
    for (int i = 0; i < k; i++) {
        const ChainGenBody b = load_rec(P.gbodies + (g.first + i));
        T sc[2] = {0, 1};
        if (!b.axisym) {
            T qi;
            if constexpr (LOOP) {
                qi = M.q(g.q_index + i);
            } else {
                cptr<T> C = P.consts + b.cofs;
                qi = 0;
#pragma unroll
                for (int a = 0; a < N; a++) qi += C[kBodyConstFixed + a] * y[a];
            }
            if constexpr (LOOP) gen_sincos(qi, &sc[0], &sc[1]);
            else sincos_t(qi, &sc[0], &sc[1]);
        }
        M.lds_st(sc0 + 2 * i, sc);
    }
    if constexpr (LOOP) gen_constraint<T, N>(P, M, g, sc0, v0, yd);
    const bool solo = g.reserved[1] != 0;
    const int fstride = solo ? 6 : 8;
    T vp[6], ap[6], vl[6], al[6];
#pragma unroll
    for (int j = 0; j < 6; j++) vl[j] = al[j] = 0;
    if (g.lds_pva >= 0) {
        T va[12];
        M.lds_ld(g.lds_pva, va);
#pragma unroll
        for (int j = 0; j < 6; j++) {
            vp[j] = va[j];
            ap[j] = va[6 + j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 6; j++) {
            vp[j] = 0;
            ap[j] = P.a_root[j];
        }
    }
    for (int i = 0; i < k; i++) {
        const ChainGenBody b = load_rec(P.gbodies + (g.first + i));
        cptr<T> C = P.consts + b.cofs;
        T Gr[N], gi, qdi, sc[2], E[9], v[6], a6[6], chat[6];
        gen_coupling<T, N, LOOP>(P, M, g, b, C, yd, Gr, gi, qdi);
        T qddi = gi;
#pragma unroll
        for (int a = 0; a < N; a++) qddi += Gr[a] * ydd[a];
        M.lds_ld(sc0 + 2 * i, sc);
        rotate_z(sc[0], sc[1], C, E);
        if (b.lam >= 0) {
            if (b.up_w >= 0) {  // (else: the body right before, whose pair is in vl / al)
                T val[12];
                M.lds_ld(g.lds_w + b.up_w, val);
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    vl[j] = val[j];
                    al[j] = val[6 + j];
                }
            }
            xmotion(E, C + 9, vl, v);
            xmotion(E, C + 9, al, a6);
        } else {
            xmotion(E, C + 9, vp, v);
            xmotion(E, C + 9, ap, a6);
        }
        v[2] += qdi;
        vxz(v, qdi, chat);
#pragma unroll
        for (int j = 0; j < 6; j++) a6[j] += chat[j];
        a6[2] += qddi;
        if (b.acc_w >= 0) {
            T val[12];
#pragma unroll
            for (int j = 0; j < 6; j++) {
                val[j] = v[j];
                val[6 + j] = a6[j];
            }
            M.lds_st(g.lds_w + b.acc_w, val);
        }
        T f[6];
        body_force_c(C + 12, v, a6, f);
        M.lds_st(g.glb_k + fstride * i, f);
        if (!solo) M.lds_st(g.glb_k + 8 * i + 6, sc);
#pragma unroll
        for (int j = 0; j < 6; j++) {
            vl[j] = v[j];
            al[j] = a6[j];
        }
        if (b.lds_va >= 0) {
            T va[12];
#pragma unroll
            for (int j = 0; j < 6; j++) {
                va[j] = v[j];
                va[6 + j] = a6[j];
            }
            M.lds_st(b.lds_va, va);
        }
    }
}

// backward segment: tau = G^T (S^T f), forces to the tree parents (TreeModel.cpp:196-209)
template <class T, int N, bool LOOP, bool GLB, class TB, class MM>
__device__ __forceinline__ void gen_rnea_bwd(const TB &P, const MM &M, const ChainGen &g)
{
    T yd[N], tau[N];
#pragma unroll
    for (int a = 0; a < N; a++) {
        yd[a] = M.qd(g.v_index + a);
        tau[a] = 0;
    }
    const bool solo = g.reserved[1] != 0;
    const int fstride = solo ? 6 : 8;
    for (int i = g.k - 1; i >= 0; i--) {
        const ChainGenBody b = load_rec(P.gbodies + (g.first + i));
        cptr<T> C = P.consts + b.cofs;
        T Gr[N], gi, qdi, sc[2], f[6], fp[6], E[9];
        gen_coupling<T, N, LOOP>(P, M, g, b, C, yd, Gr, gi, qdi);
        M.lds_ld(g.glb_k + fstride * i, f);
        M.lds_ld(solo ? g.lds_w + 2 * i : g.glb_k + 8 * i + 6, sc);
#pragma unroll
        for (int a = 0; a < N; a++) tau[a] += Gr[a] * f[2];
        rotate_z(sc[0], sc[1], C, E);
        xforce_inv(E, C + 9, f, fp);
        if (b.lam >= 0) {
            T fl[6];
            M.lds_ld(g.glb_k + fstride * b.lam, fl);
#pragma unroll
            for (int j = 0; j < 6; j++) fl[j] += fp[j];
            M.lds_st(g.glb_k + fstride * b.lam, fl);
        } else if (g.lds_acc_out != -1) {
            add6<T, GLB>(M, g.lds_acc_out, fp);
        }
    }
#pragma unroll
    for (int a = 0; a < N; a++) M.put(g.v_index + a, tau[a]);
}

// OP: 0 forward segment, 1 backward segment
template <class T, int OP, bool GLB, class TB, class MM>
__device__ __forceinline__ void gen_rnea_segment(const TB &P, const MM &M, const ChainGen &g)
{
#define GRBDA_GEN_RNEA(NN, LL)                                     \
    do {                                                            \
        if constexpr (OP == 0) gen_rnea_fwd<T, NN, LL>(P, M, g);    \
        else gen_rnea_bwd<T, NN, LL, GLB>(P, M, g);                 \
    } while (0)
    if (g.kind) {
        if (g.n == 1) GRBDA_GEN_RNEA(1, true);
        else if (g.n == 2) GRBDA_GEN_RNEA(2, true);
        else if (g.n == 3) GRBDA_GEN_RNEA(3, true);
        else GRBDA_GEN_RNEA(4, true);
    } else {
        if (g.n == 1) GRBDA_GEN_RNEA(1, false);
        else if (g.n == 2) GRBDA_GEN_RNEA(2, false);
        else if (g.n == 3) GRBDA_GEN_RNEA(3, false);
        else GRBDA_GEN_RNEA(4, false);
    }
#undef GRBDA_GEN_RNEA
}
