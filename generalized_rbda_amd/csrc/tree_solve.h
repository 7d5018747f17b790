// tree_solve.h -- branch-sparse SPD solve of the derivative pipeline: d ydd / d tau = H^-1, d ydd / d q = -H^-1 dID/dq,
// d ydd / d qd = -H^-1 dID/dqd (included by deriv_kernels.hip).
//
// The reference factors the joint-space inertia matrix as H = L^T L over the expanded parent array of the velocity coordinates
// (src/Utils/Factorization.cpp:9-36; Featherstone 2005), so that L keeps the branch sparsity of H -- L[i][j] != 0 only when j is an ancestor
// of i -- and solves with it by two sweeps over the tree (:86-144).  JVRC-1's H is 49 % structural zeros; the dense Cholesky / L^-1 / two
// GEMMs of spd_mfma_kernel factor and multiply them, one state per wavefront, 5.7 k instructions per state.  This kernel does what the
// reference does, mapped for the machine:
//   * EIGHT states per wavefront, eight lanes per state; a lane owns NS columns of every right-hand side (column t, t + 8, ...), so the
//     two sweeps are per-lane arithmetic on its own columns with the factor row broadcast from LDS -- no cross-lane traffic;
//   * the coordinates are walked in depth-first order (plan.h, TreeSolveProgram) with the values of the CURRENT ROOT PATH in a register
//     stack indexed by depth: stack[l][column] is the ancestor at level l -- static register indices, the tree itself on the scalar unit;
//       backward sweep (leaves first), z = L^-T b:  z_i = b_i' / L_ii;  stack[l] -= L[i][anc_l] z_i        (stack[l] = pending b of ancestor l)
//       forward sweep  (root first),   x = L^-1 z:  x_i = (z_i - sum_l L[i][anc_l] stack[l]) / L_ii;  stack[depth_i] = x_i
//     a node costs depth x (1 LDS read + NM NS FMAs): only entries on one root path are ever touched;
//   * the intermediate z makes its round trip through the RESULT arrays (row i of every matrix is written by consecutive instructions,
//     32 bytes per state and instruction, and read back by the same wavefront: L2 / Infinity Cache), so LDS holds the factor alone
//     (JVRC-1: 12 KB per wavefront) and the kernel runs two wavefronts per SIMD;
//   * the factorisation runs in LDS on the path layout (row i = its ancestors by level): per node the scaled row, then the rank-1 update
//     of the ancestors' rows, eight lanes of a state over the columns.
// Inputs are what rnea_deriv_kernel writes: H as packed rows of its lower triangle, dID/dq and dID/dqd as packed runs, state-major or
// interleaved by groups of `il` states; structural zeros are neither written there nor read here (DerivProgram::related).

constexpr int kTsG = 8;  // lanes per state = states per wavefront

template <class T>
__device__ __forceinline__ T ts_sqrt(T x);
template <>
__device__ __forceinline__ float ts_sqrt<float>(float x) { return __builtin_sqrtf(x); }
template <>
__device__ __forceinline__ double ts_sqrt<double>(double x) { return __builtin_sqrt(x); }

struct TsRec { int32_t node, depth, rowofs, init_lo, leaf, hdiag, pad0, pad1; };

// The two sweeps for NM of the right-hand-side matrices (m0 .. m0 + NM - 1 of IO) of one tile.  Both are software pipelines over the
// positions: the row a node needs from memory (its right-hand side going back, its z going forward) is requested while the node before
// it is computed, the node's record one step earlier still; the factor row of a node is ONE batch of LDS reads (levels beyond the depth
// masked to zero), and the register stack is updated in blocks of four levels under one wave-uniform test each.
template <class T, int NM, int NS>
__device__ __forceinline__ void tree_sweeps(const TreeSolveDev &P, const TreeSolveIO<T> &IO, int m0, const T *Lp, const T *invd, size_t st, bool live,
                                            int g, int t)
{
    constexpr int DMAX = kTreeSolveDmax, NC = NM * NS, G = kTsG;
    cptr<TsRec> recs = (cptr<TsRec>)((cptr<int32_t>)P.tab + P.o_rec);
    cptr<uint64_t> related = (cptr<uint64_t>)P.related;
    const int n = P.n, il = IO.il;
    const size_t nn = (size_t)n * n;
    const size_t grp = st / (size_t)il, sub = st % (size_t)il;
    int col[NS];
    bool cok[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        col[s] = t + G * s;
        cok[s] = col[s] < n;
    }
    const T *Ss[NM];
    T *Ds[NM];
    int kind[NM];
#pragma unroll
    for (int m = 0; m < NM; m++) {
        kind[m] = IO.kind[m0 + m];
        Ss[m] = kind[m] ? IO.src[m0 + m] + grp * nn * il + sub : nullptr;
        Ds[m] = IO.dst[m0 + m] + st * nn;
    }
    // the right-hand-side row of coordinate j, this lane's columns: 0 where an entry is a structural zero; the sign of X = -H^-1 B here
    auto rhs_row = [&](int j, T (&b)[NC]) {
        const uint64_t relj = related[j];
#pragma unroll
        for (int m = 0; m < NM; m++)
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const int c = col[s];
                if (kind[m] == 0) {
                    b[m * NS + s] = c == j ? T(1) : T(0);
                } else {
                    // (loads under the lane mask: unconditional loads from a dummy entry were measured SLOWER, 43.6 -> 53.1 ms per million
                    // JVRC-1 states for all three matrices)
                    const bool on = cok[s] && ((relj >> c) & 1);
                    const int e = c <= j ? j * j + c : c * c + c + 1 + j;
                    b[m * NS + s] = on ? -Ss[m][(size_t)e * il] : T(0);
                }
            }
    };
    auto lrow = [&](const TsRec &r, T (&lv)[DMAX]) {
#pragma unroll
        for (int l = 0; l < DMAX; l++) {
            const int lc = l < r.depth ? l : 0;
            lv[l] = Lp[(r.rowofs + lc) * G + g];   // (depth 0: the entry read is some other row's, masked below)
        }
#pragma unroll
        for (int l = 0; l < DMAX; l++) lv[l] = l < r.depth ? lv[l] : T(0);
    };
    T stk[DMAX][NC];
#pragma unroll
    for (int l = 0; l < DMAX; l++)
#pragma unroll
        for (int k = 0; k < NC; k++) stk[l][k] = 0;
    // ---- backward sweep, z = L^-T b: positions descending; stk[l] = what the descendants visited so far take off ancestor l's b ----
    {
        TsRec rc = load_rec(recs + (n - 1)), rn = load_rec(recs + (n > 1 ? n - 2 : 0));
        T bc[NC];
        rhs_row(rc.node, bc);
        for (int p = n - 1; p >= 0; p--) {
            const TsRec rnn = load_rec(recs + (p >= 2 ? p - 2 : 0));
            T bn[NC];
            if (p >= 1) rhs_row(rn.node, bn);
            T lv[DMAX];
            lrow(rc, lv);
            const T inv = invd[p * G + g];
            const int d = rc.depth;
            // ancestors entered anew start from zero
#pragma unroll
            for (int l = 0; l < DMAX; l++)
                if (l >= rc.init_lo && l < d) {
#pragma unroll
                    for (int k = 0; k < NC; k++) stk[l][k] = 0;
                }
            T z[NC];
#pragma unroll
            for (int k = 0; k < NC; k++) z[k] = bc[k];
            if (!rc.leaf) {
#pragma unroll
                for (int l = 0; l < DMAX; l++)
                    if (l == d) {
#pragma unroll
                        for (int k = 0; k < NC; k++) z[k] += stk[l][k];
                    }
            }
#pragma unroll
            for (int k = 0; k < NC; k++) z[k] *= inv;
            // z's round trip through the result rows (read back by the forward sweep, same lane)
#pragma unroll
            for (int m = 0; m < NM; m++)
#pragma unroll
                for (int s = 0; s < NS; s++)
                    if (live && cok[s]) Ds[m][(size_t)rc.node * n + col[s]] = z[m * NS + s];
#pragma unroll
            for (int l0 = 0; l0 < DMAX; l0 += 4) {
                if (l0 < d) {
#pragma unroll
                    for (int l = l0; l < l0 + 4; l++)
#pragma unroll
                        for (int k = 0; k < NC; k++) stk[l][k] -= lv[l] * z[k];
                }
            }
            rc = rn;
            rn = rnn;
#pragma unroll
            for (int k = 0; k < NC; k++) bc[k] = bn[k];
        }
    }
    // ---- forward sweep, x = L^-1 z: positions ascending; stk[l] = x of the ancestor at level l ----
    {
        auto zrow = [&](int i, T (&x)[NC]) {
#pragma unroll
            for (int m = 0; m < NM; m++)
#pragma unroll
                for (int s = 0; s < NS; s++) x[m * NS + s] = (live && cok[s]) ? Ds[m][(size_t)i * n + col[s]] : T(0);
        };
        TsRec rc = load_rec(recs + 0), rn = load_rec(recs + (n > 1 ? 1 : 0));
        T xc[NC];
        zrow(rc.node, xc);
        for (int p = 0; p < n; p++) {
            const TsRec rnn = load_rec(recs + (p + 2 < n ? p + 2 : n - 1));
            T xn[NC];
            if (p + 1 < n) zrow(rn.node, xn);
            T lv[DMAX];
            lrow(rc, lv);
            const T inv = invd[p * G + g];
            const int d = rc.depth;
            T x[NC];
#pragma unroll
            for (int k = 0; k < NC; k++) x[k] = xc[k];
#pragma unroll
            for (int l0 = 0; l0 < DMAX; l0 += 4) {
                if (l0 < d) {
#pragma unroll
                    for (int l = l0; l < l0 + 4; l++)
#pragma unroll
                        for (int k = 0; k < NC; k++) x[k] -= lv[l] * stk[l][k];
                }
            }
#pragma unroll
            for (int k = 0; k < NC; k++) x[k] *= inv;
#pragma unroll
            for (int l = 0; l < DMAX; l++)
                if (l == d) {
#pragma unroll
                    for (int k = 0; k < NC; k++) stk[l][k] = x[k];
                }
#pragma unroll
            for (int m = 0; m < NM; m++)
#pragma unroll
                for (int s = 0; s < NS; s++)
                    if (live && cok[s]) Ds[m][(size_t)rc.node * n + col[s]] = x[m * NS + s];
            rc = rn;
            rn = rnn;
#pragma unroll
            for (int k = 0; k < NC; k++) xc[k] = xn[k];
        }
    }
}

// n_mat right-hand-side matrices (1 .. 3): swept two at a time, so that the register stack (16 levels x 2 NS columns) leaves room for a
// second wavefront per SIMD in fp32
template <class T, int NS, int WPS>
__global__ __launch_bounds__(kWave, WPS) void tree_solve_kernel(TreeSolveDev P, TreeSolveIO<T> IO, int n_mat, size_t B)
{
    constexpr int DMAX = kTreeSolveDmax, G = kTsG;
    extern __shared__ __attribute__((aligned(16))) unsigned char ts_smem[];
    T *Lp = reinterpret_cast<T *>(ts_smem);  // [nl][G] path entries of L, then [n][G] 1 / L_ii, then [n][G] the diagonal
    T *invd = Lp + (size_t)P.nl * G;
    T *diag = invd + (size_t)P.n * G;
    const int lane = threadIdx.x, g = lane >> 3, t = lane & 7;
    cptr<int32_t> tab = (cptr<int32_t>)P.tab;
    const int32_t *tabv = P.tab;  // (per-lane gathers)
    const int n = P.n, il = IO.il;
    const size_t nn = (size_t)n * n;
    const size_t n_tiles = (B + G - 1) / G;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t s0 = tile * G + g;
        const bool live = s0 < B;
        const size_t st = live ? s0 : B - 1;
        const size_t grp = st / (size_t)il, sub = st % (size_t)il;
        const T *Hs = IO.H + grp * nn * il + sub;
        // ---- 1. the path entries of H into LDS ----
        for (int e = t; e < P.nl; e += G) Lp[e * G + g] = Hs[(size_t)tabv[P.o_hidx + e] * il];
        for (int p = t; p < n; p += G) diag[p * G + g] = Hs[(size_t)tabv[P.o_rec + 8 * p + 5] * il];
        __syncthreads();
        // ---- 2. H = L^T L, leaves first (Factorization.cpp:9-36) ----
        bool bad = false;
        for (int p = n - 1; p >= 0; p--) {
            const int d = tab[P.o_rec + 8 * p + 1], ro = tab[P.o_rec + 8 * p + 2];
            const T dk = diag[p * G + g];
            bad = bad || !(dk > T(0)) || !(dk < T(3e38));
            const T sq = ts_sqrt(dk), inv = T(1) / sq;
            if (t == 0) invd[p * G + g] = inv;
            for (int a = t; a < d; a += G) Lp[(ro + a) * G + g] *= inv;
            __syncthreads();
            for (int a = 0; a < d; a++) {
                const int aro = tab[P.o_ancro + DMAX * p + a], apos = tab[P.o_ancp + DMAX * p + a];
                const T la = Lp[(ro + a) * G + g];
                for (int b = t; b <= a; b += G) {
                    const T lb = Lp[(ro + b) * G + g];
                    T *tg = b < a ? &Lp[(aro + b) * G + g] : &diag[apos * G + g];
                    *tg -= la * lb;
                }
            }
            __syncthreads();
        }
        if (bad && t == 0 && live) atomicAdd(&grbda_spd_bad_count, 1ull);
        // ---- 3. the sweeps ----
        if (n_mat >= 2) tree_sweeps<T, 2, NS>(P, IO, 0, Lp, invd, st, live, g, t);
        if (n_mat != 2) tree_sweeps<T, 1, NS>(P, IO, n_mat == 3 ? 2 : 0, Lp, invd, st, live, g, t);
        __syncthreads();  // LDS is free for the next tile
    }
}

// NS: 3 for n <= 24, 5 for n <= 40 (capi.cpp asks tree_solve_covers first)
size_t tree_solve_lds_bytes(int n, int nl, size_t elem) { return (static_cast<size_t>(nl) + 2 * static_cast<size_t>(n)) * kTsG * elem; }

template <class T, int NS>
static hipError_t launch_tree_solve_ns(const TreeSolveDev &P, const TreeSolveIO<T> &IO, int n_mat, size_t B, int grid, size_t lds, hipStream_t stream)
{
    constexpr int WPS = sizeof(T) == 4 ? 2 : 1;
    hipLaunchKernelGGL((tree_solve_kernel<T, NS, WPS>), dim3(grid), dim3(kWave), lds, stream, P, IO, n_mat, B);
    return hipGetLastError();
}
template <class T>
hipError_t launch_tree_solve(const TreeSolveDev &P, const TreeSolveIO<T> &IO, int n_mat, size_t B, int grid, hipStream_t stream)
{
    const size_t lds = tree_solve_lds_bytes(P.n, P.nl, sizeof(T));
    if (P.n <= 3 * kTsG) return launch_tree_solve_ns<T, 3>(P, IO, n_mat, B, grid, lds, stream);
    return launch_tree_solve_ns<T, 5>(P, IO, n_mat, B, grid, lds, stream);
}
template hipError_t launch_tree_solve<float>(const TreeSolveDev &, const TreeSolveIO<float> &, int, size_t, int, hipStream_t);
template hipError_t launch_tree_solve<double>(const TreeSolveDev &, const TreeSolveIO<double> &, int, size_t, int, hipStream_t);
