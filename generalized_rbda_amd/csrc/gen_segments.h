// gen_segments.h -- generic clusters inside a chain program (plan.h, ChainGen / ChainGenBody).
// Included by chain_kernels.hip inside namespace grbda_hip, after ChainTables / ChainMem and the run / pair / differential code.
//
// What it restates: the cluster-level recursion of ClusterTreeModel::forwardDynamics / updateArticulatedBodies
// (src/Dynamics/ClusterTreeDynamics.cpp:85-191) for a cluster of k revolute bodies with n independent coordinates whose
// spanning rates are qd_span = G yd, qdd_span = G ydd + g:
//   * explicit clusters: constant G (LoopConstraint::Static, LoopConstraint.cpp:38-52), g = 0 -- RevoluteTripleWithRotor,
//     Generic clusters, pairs / rotors in places their own segment types do not cover;
//   * implicit clusters (GenericJoint.cpp:57-90,387-469): K = dphi/dq per state, G = P [1; -Kd^-1 Ki], g = P [0; -Kd^-1 Kdot qd];
//     phi of URDF+ <loop> elements is the distance of two points carried by the sub-chains nearest common ancestor ->
//     predecessor / successor (ClusterTreeParsing.cpp:310-376), so K is the difference of the two point Jacobians
//     a_t x (p - o_t) and Kdot qd the difference of the points' velocity-product accelerations; trig-polynomial phi
//     (LoopConstraint::FourBar, Tello's differentials when they are not in the two-rotor shape of ChainDiff) is
//     differentiated term by term.
// Per body the structured form of kernels.hip (aba_bwd_static) is kept: D = G^T Hc G, F = sum f_i G_i, u = tau - G^T b.
// Differences from the interpreter's handler, all about what a wavefront has to touch:
//   * one downward pass (sin / cos, constraint, velocities) and one upward pass per evaluation; the forward segment exists only
//     when child clusters need the velocities, and then the backward segment reuses its work area;
//   * the constraint is evaluated once, from the sin / cos the bodies use anyway; joint axes / origins of a sub-chain are
//     walked once for K and reused for Kdot qd;
//   * the in-cluster bias accelerations are not stored: h_i . ccl_i = h_i . chat_i + sum over in-cluster ancestors l of
//     (X^T ... X^T h_i)|_l . chat_l, and the force X^T h_i travels up the in-cluster chain anyway (for Hc);
//   * (IA, psi) go from a body to the body right before it in registers (chains inside the cluster), through the work
//     area otherwise; everything lives in LDS at fixed places: no LDS-or-slab test per access.
#pragma once

template <class T>
__device__ __forceinline__ void gen_cross3(const T (&a)[3], const T (&b)[3], T (&o)[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// inverse of an R x R matrix held in a 3 x 3 array, R in {1, 2, 3} (wave-uniform), closed form; true divisions: Kd^-1
// amplifies every rounding error of K near singular poses
template <class T>
__device__ __forceinline__ void gen_inv_small(int R, const T (&A)[3][3], T (&Ai)[3][3])
{
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Ai[i][j] = 0;
    if (R == 1) {
        Ai[0][0] = T(1) / A[0][0];
    } else if (R == 2) {
        const T id = T(1) / (A[0][0] * A[1][1] - A[0][1] * A[1][0]);
        Ai[0][0] = A[1][1] * id; Ai[0][1] = -A[0][1] * id;
        Ai[1][0] = -A[1][0] * id; Ai[1][1] = A[0][0] * id;
    } else {
        const T c00 = A[1][1] * A[2][2] - A[1][2] * A[2][1];
        const T c01 = A[1][2] * A[2][0] - A[1][0] * A[2][2];
        const T c02 = A[1][0] * A[2][1] - A[1][1] * A[2][0];
        const T id = T(1) / (A[0][0] * c00 + A[0][1] * c01 + A[0][2] * c02);
        Ai[0][0] = c00 * id; Ai[1][0] = c01 * id; Ai[2][0] = c02 * id;
        Ai[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) * id;
        Ai[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) * id;
        Ai[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) * id;
        Ai[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) * id;
        Ai[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) * id;
        Ai[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) * id;
    }
}

// single LDS slot
template <class T, class MM>
__device__ __forceinline__ T gen_ld1(const MM &M, int s)
{
    return reinterpret_cast<const T *>(grbda_smem)[s * kWave + M.lane];
}
template <class T, class MM>
__device__ __forceinline__ void gen_st1(const MM &M, int s, T v)
{
    reinterpret_cast<T *>(grbda_smem)[s * kWave + M.lane] = v;
}

// sin / cos of the joint angles of an implicit cluster: K and the body transforms use the SAME values (devmath.h, sincos_cw)
__device__ __forceinline__ void gen_sincos(float x, float *s, float *c)
{
#ifdef GRBDA_GEN_PRECISE
    sincos_precise(x, s, c);
#elif defined(GRBDA_GEN_HW_SINCOS)
    sincos_t(x, s, c);
#else
    sincos_cw(x, s, c);
#endif
}
__device__ __forceinline__ void gen_sincos(double x, double *s, double *c) { sincos(x, s, c); }

// ---------------------------------------------------------------------------------------------------------------
// URDF+ position loops.  Scratch (LDS slots from `scr`): K [rows x k] | per loop side: [a 3][o 3] per joint of its path
// (axis and origin in the coordinates of the nearest common ancestor), then the constraint point [p 3].
// ---------------------------------------------------------------------------------------------------------------
template <class T, class MM, class TB>
__device__ __forceinline__ void gen_loop_K(const TB &P, const MM &M, const ChainGen &g, int sc0, int scr,
                                           cptr<int32_t> loops, int n_loops)
{
    const int k = g.k;
    int rec = scr + g.rows * k;
    int row0 = 0;
    cptr<int32_t> lp = loops;
    for (int l = 0; l < n_loops; l++) {
        const int np = lp[0], ns = lp[1 + np], mask = lp[2 + np + ns];
        cptr<T> org = P.consts + g.dofs + 24 * l;
        for (int side = 0; side < 2; side++) {
            cptr<int32_t> subs = side == 0 ? lp + 1 : lp + 2 + np;
            const int len = side == 0 ? np : ns;
            T E[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, r[3] = {0, 0, 0};
            for (int t = 0; t < len; t++) {
                const int sub = subs[t];
                const ChainGenBody b = load_rec(P.gbodies + (g.first + sub));
                cptr<T> C = P.consts + b.cofs;
                T sc[2], Eb[9], En[9];
                M.lds_ld(sc0 + 2 * sub, sc);
                rotate_z(sc[0], sc[1], C, Eb);
                // X_new = (Eb, r_tree) * (E, r):  E_new = Eb E,  r_new = r + E^T r_tree
#pragma unroll
                for (int i = 0; i < 3; i++) r[i] += E[i] * C[9] + E[3 + i] * C[10] + E[6 + i] * C[11];
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                    for (int j = 0; j < 3; j++) En[3 * i + j] = Eb[3 * i] * E[j] + Eb[3 * i + 1] * E[3 + j] + Eb[3 * i + 2] * E[6 + j];
#pragma unroll
                for (int i = 0; i < 9; i++) E[i] = En[i];
                const T ao[6] = {E[6], E[7], E[8], r[0], r[1], r[2]};  // the joint axis is the body's z axis (canonical joint axes, plan.cpp)
                M.lds_st(rec + 6 * t, ao);
            }
            T p[3];
            cptr<T> og = org + 12 * side;
#pragma unroll
            for (int i = 0; i < 3; i++) p[i] = r[i] + E[i] * og[9] + E[3 + i] * og[10] + E[6 + i] * og[11];
            M.lds_st(rec + 6 * len, p);
            const T sgn = side == 0 ? T(1) : T(-1);
            for (int t = 0; t < len; t++) {
                T ao[6];
                M.lds_ld(rec + 6 * t, ao);
                const T a[3] = {ao[0], ao[1], ao[2]}, d[3] = {p[0] - ao[3], p[1] - ao[4], p[2] - ao[5]};
                T J[3];
                gen_cross3(a, d, J);
                int row = row0;
                const int sub = subs[t];
#pragma unroll
                for (int ax = 0; ax < 3; ax++)
                    if (mask & (1 << ax)) {
                        gen_st1(M, scr + row * k + sub, sgn * J[ax]);
                        row++;
                    }
            }
            rec += 6 * len + 3;
        }
        row0 += ((mask >> 0) & 1) + ((mask >> 1) & 1) + ((mask >> 2) & 1);
        lp += 3 + np + ns;
    }
}

// velocity-product acceleration of the constraint points, (Kdot qd)[row]; qd0: the k spanning rates (LDS scratch)
template <class T, class MM, class TB>
__device__ __forceinline__ void gen_loop_Kdqd(const TB &P, const MM &M, const ChainGen &g, int scr, int qd0,
                                              cptr<int32_t> loops, int n_loops, T (&kdq)[3])
{
    const int k = g.k;
    int rec = scr + g.rows * k;
    int row0 = 0;
    cptr<int32_t> lp = loops;
    for (int l = 0; l < n_loops; l++) {
        const int np = lp[0], ns = lp[1 + np], mask = lp[2 + np + ns];
        T acc[3] = {0, 0, 0};
        for (int side = 0; side < 2; side++) {
            cptr<int32_t> subs = side == 0 ? lp + 1 : lp + 2 + np;
            const int len = side == 0 ? np : ns;
            T w[3] = {0, 0, 0}, al[3] = {0, 0, 0}, ao_[3] = {0, 0, 0}, op[3] = {0, 0, 0};
            for (int t = 0; t <= len; t++) {
                T a[3] = {0, 0, 0}, o[3];
                T qd_t = 0;
                if (t < len) {
                    T r6[6], q1[1];
                    M.lds_ld(rec + 6 * t, r6);
                    a[0] = r6[0]; a[1] = r6[1]; a[2] = r6[2];
                    o[0] = r6[3]; o[1] = r6[4]; o[2] = r6[5];
                    M.lds_ld(qd0 + subs[t], q1);
                    qd_t = q1[0];
                } else {
                    M.lds_ld(rec + 6 * len, o);
                }
                // the point o is carried rigidly by the previous frame (w, al at origin op)
                const T d[3] = {o[0] - op[0], o[1] - op[1], o[2] - op[2]};
                T wd[3], wwd[3], ad[3];
                gen_cross3(w, d, wd);
                gen_cross3(w, wd, wwd);
                gen_cross3(al, d, ad);
#pragma unroll
                for (int i = 0; i < 3; i++) { ao_[i] += ad[i] + wwd[i]; op[i] = o[i]; }
                // then the joint adds its own rate about a (qdd = 0)
                const T aq[3] = {a[0] * qd_t, a[1] * qd_t, a[2] * qd_t};
                T waq[3];
                gen_cross3(w, aq, waq);
#pragma unroll
                for (int i = 0; i < 3; i++) { al[i] += waq[i]; w[i] += aq[i]; }
            }
            const T sgn = side == 0 ? T(1) : T(-1);
#pragma unroll
            for (int i = 0; i < 3; i++) acc[i] += sgn * ao_[i];
            rec += 6 * len + 3;
        }
        int row = row0;
#pragma unroll
        for (int ax = 0; ax < 3; ax++)
            if (mask & (1 << ax)) {
                if (row == 0) kdq[0] = acc[ax];
                else if (row == 1) kdq[1] = acc[ax];
                else kdq[2] = acc[ax];
                row++;
            }
        row0 = row;
        lp += 3 + np + ns;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// trig-polynomial constraints phi_r = sum_t coef prod_f F_f(w_f . q + b_f), F in {id, sin, cos} (the data form of the
// reference's phi lambdas; plan.cpp: ints [n_args, per row: n_terms, per term: n_factors, (type, argument)...], constants
// [per distinct argument w[k], b][per term coef]).  Scratch: K [rows x k] | per argument [a, sin a, cos a, w . qd_span].
// want_K: rows of K to the scratch; otherwise the second directional derivative along qd_span goes to kdq.
// ---------------------------------------------------------------------------------------------------------------
template <class T, class MM, class TB>
__device__ __forceinline__ void gen_trig_eval(const TB &P, const MM &M, const ChainGen &g, int scr, int qd0,
                                              cptr<int32_t> prog, bool want_K, T (&kdq)[3])
{
    const int k = g.k;
    cptr<int32_t> ip = prog;
    const int n_args = *ip++;
    cptr<T> ap = P.consts + g.dofs;      // per distinct argument: w[k], b
    cptr<T> cp = ap + n_args * (k + 1);  // per term: coef
    const int arg0 = scr + g.rows * k;
    T qv[kMaxClusterBodies];
#pragma unroll
    for (int j = 0; j < kMaxClusterBodies; j++) {
        qv[j] = 0;
        if (j < k) {
            if (want_K) {
                qv[j] = M.q(g.q_index + j);
            } else {
                qv[j] = gen_ld1<T>(M, qd0 + j);
            }
        }
    }
    for (int i = 0; i < n_args; i++) {
        cptr<T> w = ap + i * (k + 1);
        T a = want_K ? w[k] : T(0);
#pragma unroll
        for (int j = 0; j < kMaxClusterBodies; j++)
            if (j < k) a += w[j] * qv[j];
        if (want_K) {
            T sn, cs;
            gen_sincos(a, &sn, &cs);
            const T v3[3] = {a, sn, cs};
            M.lds_st(arg0 + 4 * i, v3);
        } else {
            gen_st1(M, arg0 + 4 * i + 3, a);
        }
    }
    for (int r = 0; r < g.rows; r++) {
        const int nt = *ip++;
        T Krow[kMaxClusterBodies];
#pragma unroll
        for (int j = 0; j < kMaxClusterBodies; j++) Krow[j] = 0;
        T kd = 0;
        for (int t = 0; t < nt; t++) {
            const int nf = *ip++;
            const T coef = *cp++;
            T f0[4], f1[4], f2[4], ad[4];
            cptr<T> wv[4];
#pragma unroll
            for (int f = 0; f < 4; f++) {
                f0[f] = 1; f1[f] = 0; f2[f] = 0; ad[f] = 0;
                wv[f] = ap;
                if (f < nf) {
                    const int type = ip[0], arg = ip[1];
                    ip += 2;
                    wv[f] = ap + arg * (k + 1);
                    T v4[4];
                    M.lds_ld(arg0 + 4 * arg, v4);
                    ad[f] = want_K ? T(0) : v4[3];
                    if (type == 1) { f0[f] = v4[1]; f1[f] = v4[2]; f2[f] = -v4[1]; }
                    else if (type == 2) { f0[f] = v4[2]; f1[f] = -v4[1]; f2[f] = -v4[2]; }
                    else { f0[f] = v4[0]; f1[f] = 1; f2[f] = 0; }
                }
            }
#pragma unroll
            for (int f = 0; f < 4; f++) {
                if (f < nf) {
                    T others = coef;
#pragma unroll
                    for (int h = 0; h < 4; h++)
                        if (h != f) others *= f0[h];
                    if (want_K) {
#pragma unroll
                        for (int j = 0; j < kMaxClusterBodies; j++)
                            if (j < k) Krow[j] += others * f1[f] * wv[f][j];
                    } else {
                        kd += others * f2[f] * ad[f] * ad[f];
#pragma unroll
                        for (int h = 0; h < 4; h++)
                            if (h != f && h < nf) {
                                T rest = coef;
#pragma unroll
                                for (int m = 0; m < 4; m++)
                                    if (m != f && m != h) rest *= f0[m];
                                kd += rest * f1[f] * ad[f] * f1[h] * ad[h];
                            }
                    }
                }
            }
        }
        if (want_K) {
#pragma unroll
            for (int j = 0; j < kMaxClusterBodies; j++)
                if (j < k) gen_st1(M, scr + r * k + j, Krow[j]);
        } else {
            if (r == 0) kdq[0] = kd;
            else if (r == 1) kdq[1] = kd;
            else kdq[2] = kd;
        }
    }
}

// G rows, g and the spanning rates of the DEPENDENT bodies of an implicit cluster into the kept block: row r = [G row N][g][qd_span]
// of dependent coordinate r (an independent body's row of G is a unit vector, its g is 0 and its rate is yd: gen_coupling)
template <class T, int N, class MM, class TB>
__device__ __forceinline__ void gen_constraint(const TB &P, const MM &M, const ChainGen &g, int sc0, int scr,
                                               const T (&yd)[N])
{
    constexpr int ks = N + 2;
    const int k = g.k, rows = g.rows;
    cptr<int32_t> ip = P.cints + g.iofs;
    const int hdr0 = ip[0];  // number of loops (position loops)
    const int n_ind = ip[1];
    cptr<int32_t> ind = ip + 2;
    cptr<int32_t> dep = ip + 3 + n_ind;
    cptr<int32_t> payload = dep + rows;
    T kdq[3] = {0, 0, 0};
    for (int i = 0; i < rows * k; i++) gen_st1(M, scr + i, T(0));
    if (g.kind == 1) gen_loop_K<T>(P, M, g, sc0, scr, payload, hdr0);
    else gen_trig_eval(P, M, g, scr, scr, payload, true, kdq);

    T Kd[3][3], Kdi[3][3], X[3][N];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int j = 0; j < 3; j++) Kd[r][j] = (r < rows && j < rows) ? gen_ld1<T>(M, scr + r * k + dep[j]) : T(r == j);
    gen_inv_small(rows, Kd, Kdi);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int a = 0; a < N; a++) {
            T s = 0;
#pragma unroll
            for (int j = 0; j < 3; j++)
                if (r < rows && j < rows) s += Kdi[r][j] * gen_ld1<T>(M, scr + j * k + ind[a]);
            X[r][a] = s;
        }
    // the spanning rates of all k bodies take the place of K's first row (K is not needed any more)
#pragma unroll
    for (int a = 0; a < N; a++) gen_st1(M, scr + ind[a], yd[a]);
#pragma unroll
    for (int r = 0; r < 3; r++)
        if (r < rows) {
            T row[ks];
            T s = 0;
#pragma unroll
            for (int a = 0; a < N; a++) {
                row[a] = -X[r][a];
                s -= X[r][a] * yd[a];
            }
            row[N] = 0;
            row[N + 1] = s;
            gen_st1(M, scr + dep[r], s);
            M.acc_st(g.keep + r * ks, row);
        }
    // k = -Kdot qd ; g = P [0; Kd^-1 k]
    if (g.kind == 1) gen_loop_Kdqd(P, M, g, scr, scr, payload, hdr0, kdq);
    else gen_trig_eval(P, M, g, scr, scr, payload, false, kdq);
#pragma unroll
    for (int r = 0; r < 3; r++)
        if (r < rows) {
            const T gv[1] = {-(Kdi[r][0] * kdq[0] + Kdi[r][1] * kdq[1] + Kdi[r][2] * kdq[2])};
            M.acc_st(g.keep + r * ks + N, gv);
        }
}

// coupling of a body: G row (registers), g_i, qd_i
template <class T, int N, bool LOOP, class MM, class TB>
__device__ __forceinline__ void gen_coupling(const TB &P, const MM &M, const ChainGen &g, const ChainGenBody &b,
                                             cptr<T> C, const T (&yd)[N], T (&Gr)[N], T &gi, T &qdi)
{
    if constexpr (LOOP) {
        if (b.dep_r >= 0) {
            T row[N + 2];
            M.acc_ld(g.keep + b.dep_r * (N + 2), row);
#pragma unroll
            for (int a = 0; a < N; a++) Gr[a] = row[a];
            gi = row[N];
            qdi = row[N + 1];
        } else {
            T s = 0;
#pragma unroll
            for (int a = 0; a < N; a++) {
                Gr[a] = a == b.ind_a ? T(1) : T(0);
                s += Gr[a] * yd[a];
            }
            gi = 0;
            qdi = s;
        }
    } else {
        T s = 0;
#pragma unroll
        for (int a = 0; a < N; a++) {
            Gr[a] = C[kBodyConstFixed + a];
            s += Gr[a] * yd[a];
        }
        gi = 0;
        qdi = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// downward pass: [sin, cos] of every body, the constraint of an implicit cluster, the velocity of every body
// (TreeModel::forwardKinematics, TreeModel.cpp:6-32, inside the cluster)
// ---------------------------------------------------------------------------------------------------------------
// w: work area; constraint: evaluate it (else the kept block is valid already); publish: velocities of the bodies with child clusters
// to their lds_v
template <class T, int N, bool LOOP, class MM, class TB>
__device__ __forceinline__ void gen_down(const TB &P, const MM &M, const ChainGen &g, int w, bool constraint, bool publish)
{
    const int k = g.k;
    const int sc0 = w, v0 = w + 2 * k;
    T y[N], yd[N];
#pragma unroll
    for (int a = 0; a < N; a++) {
        yd[a] = M.qd(g.v_index + a);
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
    if constexpr (LOOP) {
        if (constraint) gen_constraint<T, N>(P, M, g, sc0, v0, yd);
    }
    T vp[6];
    if (g.lds_pv != -1) {
        M.acc_ld(g.lds_pv, vp);
    } else {
#pragma unroll
        for (int j = 0; j < 6; j++) vp[j] = 0;
    }
    for (int i = 0; i < k; i++) {
        const ChainGenBody b = load_rec(P.gbodies + (g.first + i));
        cptr<T> C = P.consts + b.cofs;
        T Gr[N], gi, qdi, sc[2], E[9], v[6];
        gen_coupling<T, N, LOOP>(P, M, g, b, C, yd, Gr, gi, qdi);
        M.lds_ld(sc0 + 2 * i, sc);
        rotate_z(sc[0], sc[1], C, E);
        if (b.lam >= 0) {
            T vl[6];
            M.lds_ld(v0 + 6 * b.lam, vl);
            xmotion(E, C + 9, vl, v);
        } else {
            xmotion(E, C + 9, vp, v);
        }
        v[2] += qdi;
        M.lds_st(v0 + 6 * i, v);
        if (publish && b.lds_v != -1) M.acc_st(b.lds_v, v);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// upward pass: updateArticulatedBodies + bias back-propagation (ClusterTreeDynamics.cpp:94-129,157-191) of one cluster
// ---------------------------------------------------------------------------------------------------------------
// FUSE: the cluster hangs off the ground and has no child clusters (single-cluster programs, aba_gen1_kernel): the acceleration
// pass is two lines -- ydd = y0 - K a_root goes straight to the results, no [K | y0] block leaves the registers
template <class T, int N, bool LOOP, bool FUSE, class MM, class TB>
__device__ __forceinline__ void gen_up(const TB &P, const MM &M, const ChainGen &g)
{
    const int k = g.k;
    const int sc0 = g.lds_w, v0 = g.lds_w + 2 * k;
    T yd[N], u[N], F[6][N], D[N][N];
    T pIA[21], ppsi[6];  // what the bodies on the cluster's parent body hand to it
    T cIA[21], cpsi[6];  // register hand-over from a body to the body right before it
#pragma unroll
    for (int j = 0; j < 21; j++) pIA[j] = cIA[j] = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) ppsi[j] = cpsi[j] = 0;
#pragma unroll
    for (int a = 0; a < N; a++) {
        yd[a] = M.qd(g.v_index + a);
        u[a] = M.x(g.v_index + a);
#pragma unroll
        for (int r = 0; r < 6; r++) F[r][a] = 0;
#pragma unroll
        for (int bb = 0; bb < N; bb++) D[a][bb] = 0;
    }
    for (int i = k - 1; i >= 0; i--) {
        const ChainGenBody b = load_rec(P.gbodies + (g.first + i));
        cptr<T> C = P.consts + b.cofs;
        cptr<T> Ic = C + 12;
        T G[N], gi, qdi, sc[2], E[9], v[6], chat[6];
        gen_coupling<T, N, LOOP>(P, M, g, b, C, yd, G, gi, qdi);
        M.lds_ld(sc0 + 2 * i, sc);
        M.lds_ld(v0 + 6 * i, v);
        rotate_z(sc[0], sc[1], C, E);
        vxz(v, qdi, chat);
        chat[2] += gi;  // S_implicit g (GenericJoint.cpp:449-450)

        T IA[21], psi[6];
        bias_force(Ic, v, psi);  // pA = v x* (I v), ClusterTreeDynamics.cpp:95-98
        cptr<T> Ib = P.consts + b.iofs;
#pragma unroll
        for (int j = 0; j < 21; j++) IA[j] = Ib[j];
        if (b.carry_in) {
#pragma unroll
            for (int j = 0; j < 21; j++) IA[j] += cIA[j];
#pragma unroll
            for (int j = 0; j < 6; j++) psi[j] += cpsi[j];
        }
        if (b.acc_w >= 0) {
            T acc[27];
            M.lds_ld(g.lds_w + b.acc_w, acc);
#pragma unroll
            for (int j = 0; j < 21; j++) IA[j] += acc[j];
#pragma unroll
            for (int j = 0; j < 6; j++) psi[j] += acc[21 + j];
        }
        if (b.lds_acc != -1) {
            T acc[27];
            M.acc_ld(b.lds_acc, acc);
#pragma unroll
            for (int j = 0; j < 21; j++) IA[j] += acc[j];
#pragma unroll
            for (int j = 0; j < 6; j++) psi[j] += acc[21 + j];
        }
        T h[6];
        #pragma unroll
        for (int j = 0; j < 6; j++) h[j] = IA[sidx(j, 2)];
        const T d = h[2];
        T bj = psi[2];
#pragma unroll
        for (int j = 0; j < 6; j++) bj += h[j] * chat[j];

        // hand composite inertia and bias to the tree parent (in-cluster body, or the cluster's parent body)
        if (b.lam >= 0 || g.has_parent) {
            T t[6], tp[6], Bc[21];
            symv(IA, chat, t);
#pragma unroll
            for (int j = 0; j < 6; j++) t[j] += psi[j];
            xforce_inv(E, C + 9, t, tp);
            if (!b.axisym) {
                congruence(E, C + 9, IA, Bc);
            } else {  // X0^T I X0 is already part of the parent's constants
#pragma unroll
                for (int j = 0; j < 21; j++) Bc[j] = 0;
            }
            if (b.lam < 0) {
#pragma unroll
                for (int j = 0; j < 6; j++) ppsi[j] += tp[j];
#pragma unroll
                for (int j = 0; j < 21; j++) pIA[j] += Bc[j];
            } else if (b.carry_up) {
#pragma unroll
                for (int j = 0; j < 6; j++) cpsi[j] = tp[j];
#pragma unroll
                for (int j = 0; j < 21; j++) cIA[j] = Bc[j];
            } else {
                T acc[27];
                if (b.up_first) {
#pragma unroll
                    for (int j = 0; j < 27; j++) acc[j] = 0;
                } else {
                    M.lds_ld(g.lds_w + b.up_w, acc);
                }
#pragma unroll
                for (int j = 0; j < 21; j++) acc[j] += Bc[j];
#pragma unroll
                for (int j = 0; j < 6; j++) acc[21 + j] += tp[j];
                M.lds_st(g.lds_w + b.up_w, acc);
            }
        }

        // joint-space terms: D += d G^T G, push h up the in-cluster chain (Hc couplings and the ancestors' bias accelerations)
#pragma unroll
        for (int a = 0; a < N; a++)
#pragma unroll
            for (int bb = 0; bb < N; bb++) D[a][bb] += d * G[a] * G[bb];
        T f[6];
        xforce_inv(E, C + 9, h, f);
        int l = b.lam;
        while (l >= 0) {
            const ChainGenBody bl = load_rec(P.gbodies + (g.first + l));
            cptr<T> Cl = P.consts + bl.cofs;
            T Gl[N], gl, qdl, scl[2], vl[6], cl[6], El[9], f2[6];
            gen_coupling<T, N, LOOP>(P, M, g, bl, Cl, yd, Gl, gl, qdl);
            M.lds_ld(sc0 + 2 * l, scl);
            M.lds_ld(v0 + 6 * l, vl);
            vxz(vl, qdl, cl);
            cl[2] += gl;
#pragma unroll
            for (int j = 0; j < 6; j++) bj += f[j] * cl[j];
            const T Hc = f[2];
#pragma unroll
            for (int a = 0; a < N; a++)
#pragma unroll
                for (int bb = 0; bb < N; bb++) D[a][bb] += Hc * (Gl[a] * G[bb] + G[a] * Gl[bb]);
            rotate_z(scl[0], scl[1], Cl, El);
            xforce_inv(El, Cl + 9, f, f2);
#pragma unroll
            for (int j = 0; j < 6; j++) f[j] = f2[j];
            l = bl.lam;
        }
#pragma unroll
        for (int a = 0; a < N; a++) u[a] -= G[a] * bj;
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int a = 0; a < N; a++) F[r][a] += f[r] * G[a];
    }

    // D^-1 u', K = D^-1 F^T
    Chol<T, N> ch;
    ch.factor(D);
#pragma unroll
    for (int a = 0; a < N; a++) M.pivot(ch.inv[a] < T(1e30) ? ch.inv[a] : T(0));  // (1 / sqrt(pivot): Inf for 0, NaN below it)
    ch.solve(u);
    T blk[7 * N];
#pragma unroll
    for (int r = 0; r < 6; r++) {
        T col[N];
#pragma unroll
        for (int a = 0; a < N; a++) col[a] = F[r][a];
        ch.solve(col);
#pragma unroll
        for (int a = 0; a < N; a++) blk[a * 6 + r] = col[a];
    }
#pragma unroll
    for (int a = 0; a < N; a++) blk[6 * N + a] = u[a];
    if constexpr (FUSE) {
#pragma unroll
        for (int a = 0; a < N; a++) {
            T ydd = u[a];
#pragma unroll
            for (int r = 0; r < 6; r++) ydd -= blk[a * 6 + r] * P.a_root[r];
            M.put_f(g.v_index + a, ydd);
        }
        return;
    } else {
        M.glb_st(g.glb_k, blk);
    }

    // the parent body receives sum X^T IA X - F D^-1 F^T and sum X^T (pA + IA c) + F D^-1 u'
    if (g.lds_acc_out != -1) {
        T acc[27];
        if (g.acc_first) {
#pragma unroll
            for (int j = 0; j < 27; j++) acc[j] = 0;
        } else {
            M.acc_ld(g.lds_acc_out, acc);
        }
#pragma unroll
        for (int r = 0; r < 6; r++) {
            T s = ppsi[r];
#pragma unroll
            for (int a = 0; a < N; a++) s += F[r][a] * u[a];
            acc[21 + r] += s;
#pragma unroll
            for (int cc = r; cc < 6; cc++) {
                T m = pIA[sidx(r, cc)];
#pragma unroll
                for (int a = 0; a < N; a++) m -= F[r][a] * blk[a * 6 + cc];
                acc[sidx(r, cc)] += m;
            }
        }
        M.acc_st(g.lds_acc_out, acc);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// acceleration pass (ClusterTreeDynamics.cpp:131-152): ydd = y0 - K a_p; (v, a) of the bodies child clusters hang off
// ---------------------------------------------------------------------------------------------------------------
template <class T, int N, bool LOOP, class MM, class TB>
__device__ __forceinline__ void gen_acc(const TB &P, const MM &M, const ChainGen &g)
{
    T blk[7 * N], vp[6], ap[6], ydd[N];
    M.glb_ld(g.glb_k, blk);
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
#pragma unroll
    for (int a = 0; a < N; a++) {
        T s = blk[6 * N + a];
#pragma unroll
        for (int r = 0; r < 6; r++) s -= blk[a * 6 + r] * ap[r];
        ydd[a] = s;
        M.put_f(g.v_index + a, s);
    }
    if (!g.need_acc) return;
    T y[N], yd[N];
#pragma unroll
    for (int a = 0; a < N; a++) {
        yd[a] = M.qd(g.v_index + a);
        y[a] = LOOP ? T(0) : M.q(g.q_index + a);
    }
    for (int i = 0; i < g.k; i++) {
        const ChainGenBody b = load_rec(P.gbodies + (g.first + i));
        if (b.lds_va < 0) continue;
        cptr<T> C = P.consts + b.cofs;
        T G[N], gi, qdi, qi;
        gen_coupling<T, N, LOOP>(P, M, g, b, C, yd, G, gi, qdi);
        if constexpr (LOOP) {
            qi = M.q(g.q_index + i);
        } else {
            qi = 0;
#pragma unroll
            for (int a = 0; a < N; a++) qi += G[a] * y[a];
        }
        T sc[2], E[9], v[6], a6[6], chat[6];
        if constexpr (LOOP) gen_sincos(qi, &sc[0], &sc[1]);
        else sincos_t(qi, &sc[0], &sc[1]);
        rotate_z(sc[0], sc[1], C, E);
        if (b.lam >= 0) {
            T va[12], vl[6], al[6];
            M.lds_ld(b.pva, va);
#pragma unroll
            for (int j = 0; j < 6; j++) {
                vl[j] = va[j];
                al[j] = va[6 + j];
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
        T qddi = gi;
#pragma unroll
        for (int a = 0; a < N; a++) qddi += G[a] * ydd[a];
        a6[2] += qddi;
        T out[12];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            out[j] = v[j];
            out[6 + j] = a6[j];
        }
        M.lds_st(b.lds_va, out);
    }
}

// OP: 0 forward segment, 1 backward segment, 2 acceleration segment
template <class T, int N, bool LOOP, int OP, class MM, class TB>
__device__ __forceinline__ void gen_run(const TB &P, const MM &M, const ChainGen &g)
{
    if constexpr (OP == 0) {
        gen_down<T, N, LOOP>(P, M, g, g.lds_wf, true, true);
    } else if constexpr (OP == 1) {
        gen_down<T, N, LOOP>(P, M, g, g.lds_w, !g.has_fwd, false);
        gen_up<T, N, LOOP, false>(P, M, g);
    } else {
        gen_acc<T, N, LOOP>(P, M, g);
    }
}
template <class T, int OP, class MM, class TB>
__device__ __forceinline__ void gen_segment(const TB &P, const MM &M, const ChainGen &g)
{
    if (g.kind) {
        if (g.n == 1) gen_run<T, 1, true, OP>(P, M, g);
        else if (g.n == 2) gen_run<T, 2, true, OP>(P, M, g);
        else if (g.n == 3) gen_run<T, 3, true, OP>(P, M, g);
        else gen_run<T, 4, true, OP>(P, M, g);
    } else {
        if (g.n == 1) gen_run<T, 1, false, OP>(P, M, g);
        else if (g.n == 2) gen_run<T, 2, false, OP>(P, M, g);
        else if (g.n == 3) gen_run<T, 3, false, OP>(P, M, g);
        else gen_run<T, 4, false, OP>(P, M, g);
    }
}
