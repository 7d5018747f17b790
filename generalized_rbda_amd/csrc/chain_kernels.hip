// chain_kernels.hip -- chain-structured cluster ABA for gfx950 (MI355X): the fast path of the f32 forward dynamics.
//
// Same execution model as kernels.hip (one robot state per lane, 64 states per wavefront, one wavefront per
// workgroup, the model read through the scalar unit) and the same algorithm -- the structured restatement of
// ClusterTreeModel::forwardDynamics / updateArticulatedBodies (src/Dynamics/ClusterTreeDynamics.cpp:85-191) described at
// the top of kernels.hip -- but a different program shape.  kernels.hip interprets one (sweep, cluster) step per loop
// iteration; every quantity that crosses a step boundary in registers (the projected inertia handed to the parent, the
// prefetched [K | y0] block) is then loop-carried through a loop with two dozen paths, and the compiler pays for it
// with register copies at the latch, 200+ VGPRs and control-flow scaffolding around every slot access.
// Here the plan compiler cuts the cluster tree into CHAINS (plan.h, ChainProgram) and a sweep over a chain is one
// tight single-path loop: velocities / accelerations going down and the articulated inertia / bias going up stay
// in registers from link to link; what another segment needs goes through LDS; the [K | y0] blocks go to the wave's
// global slab (written once by the backward run, read once by the acceleration run, the next link's block fetched
// while the current link is computed).  The segment loop itself carries no vector state.
//
// Covered cluster types: Free root or links on the ground, Revolute, RevoluteWithRotor (axisymmetric rotor evaluated at
// q = 0, plan.cpp; any other rotor at its own angle), leaf RevolutePairWithRotor
// (src/Dynamics/ClusterJoints/RevolutePairWithRotorJoint.cpp), and the implicit two-rotor differentials of Tello
// (GenericImplicit clusters of src/Robots/Tello.cpp; ChainDiff below).  Everything else runs on the general interpreter
// (kernels.hip); the plan compiler decides (ChainProgram::ok).  The same file holds the inverse dynamics on the chains
// (rnea_chain_kernel) and the force-propagation kernel of the contact side (osim_chain_kernel).
//
// Four translation units are built from this file (Makefile; unit 3: see GRBDA_KLDS below).  GRBDA_CHAIN_UNIT == 2 carries the kernels of programs with
// generic clusters (gen_segments.h: aba_chain_kernel<T, 2, 2>), GRBDA_CHAIN_UNIT == 1 carries the two fp64 kernels with the deepest
// register pressure -- aba_chain_lm_kernel<double, 2> and aba_chain_kernel<double, 2, true> (the program with differentials); unit 0
// is everything else.  Both fp64 kernels spill, and how much depends on what else the compiler sees in the unit: measured in
// one run against the single-unit build (tools/ab3.sh), Mini Cheetah fp64 at 65 536 states 0.0727 -> 0.0664 ms and TelloWithArms
// fp64 3.60 -> 2.61 ms, every other kernel unchanged.  (Scheduler strategies were tried on top -- make variant
// U1FLAGS="-mllvm -amdgpu-sched-strategy=max-ilp": same figures for these two, fp32 kernels as fast or faster with the default.)
#include <hip/hip_runtime.h>

#include "devplan.h"

#ifndef GRBDA_CHAIN_UNIT
#define GRBDA_CHAIN_UNIT 0
#endif
// GRBDA_CHAIN_UNIT == 3 carries the fp32 latency-mode kernels aba_chain_lm_kernel<float, 2 / 4> and the fp64 one with four wavefronts per tile, whose "slab" blocks [K | y0] are LDS
// objects (ChainMem::glb_ld / glb_st below): a unit of its own, so that the slot test exists in no other kernel's text (as a member flag
// that every other kernel sets to false it still moved TelloWithArms' kernel to 68 bytes of scratch).
#define GRBDA_KLDS (GRBDA_CHAIN_UNIT == 3)
// Which code paths use the permutation-structured transforms of devmath.h (rzp_*).  The defaults are what the same-run A/B
// (tools/ab3.sh, library variants of `make variant VFLAGS=-DGRBDA_PERM_...`) kept: links, general rotors and pairs of the fp32
// forward dynamics (JVRC-1 1.234 -> 1.178 ms per 2^20 states, TelloWithArms 0.801 -> 0.789, MIT Humanoid unchanged at 0.1495 --
// its limbs are short and the time is not in the transforms); NOT the acceleration run, the inverse dynamics (JVRC-1 fp32
// 0.665 -> 0.715 ms with them: the switch costs more than the products save) and the fp64 kernels (they spill; Tello fp64
// 2.6 -> 3.7 ms with them).
#ifndef GRBDA_PERM_LINK
#define GRBDA_PERM_LINK 1
#endif
#ifndef GRBDA_PERM_ACC
#define GRBDA_PERM_ACC 0
#endif
#ifndef GRBDA_PERM_ROTOR
#define GRBDA_PERM_ROTOR 1
#endif
#ifndef GRBDA_PERM_PAIR
#define GRBDA_PERM_PAIR 1
#endif
#ifndef GRBDA_PERM_RNEA
#define GRBDA_PERM_RNEA 0
#endif
#ifndef GRBDA_PERM_F64
#define GRBDA_PERM_F64 0
#endif

namespace grbda_hip {

#include "devmath.h"

// optional in-kernel cycle accounting of the chain kernel (make expc NAME=cprof DEFS=-DGRBDA_CHAIN_PROFILE, tools/chain_prof.py;
// never in the shipped library): s_memtime deltas per tile phase and segment type, summed over wavefronts
#ifdef GRBDA_CHAIN_PROFILE
__device__ unsigned long long grbda_chain_prof[128];
#define CPROF_T0() unsigned long long cprof_t = __builtin_amdgcn_s_memtime(), cprof_acc = 0, cprof_cnt = 0
// lane i keeps bucket i (no memory traffic inside the tile loop); CPROF_END adds the lanes' buckets to the global table
#define CPROF_ADD(i, n)                                                                   \
    do {                                                                                  \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                     \
        if (lane == (i)) {                                                                \
            cprof_acc += now_ - cprof_t;                                                  \
            cprof_cnt += (unsigned long long)(n);                                         \
        }                                                                                 \
        cprof_t = now_;                                                                   \
    } while (0)
#define CPROF_END()                                                                       \
    do {                                                                                  \
        if (lane < 32) {                                                                  \
            atomicAdd(&grbda_chain_prof[lane], cprof_acc);                                \
            atomicAdd(&grbda_chain_prof[32 + lane], cprof_cnt);                           \
        }                                                                                 \
        atomicAdd(&grbda_chain_prof[64 + lane], M.pacc);                                  \
    } while (0)
// marks inside a run's link loop: issue-time stamps (ordered behind the value `dep`), summed per interval into lane buckets 32.. of
// a second per-lane accumulator that lives in ChainMem (M.pacc)
#define CMARK(k, dep)                                                                     \
    do {                                                                                  \
        asm volatile("" ::"v"(dep) : "memory");                                           \
        cm_[k] = __builtin_amdgcn_s_memtime();                                            \
    } while (0)
#define CMARK_DECL(n) unsigned long long cm_[n]
#define CMARK_SUM(base, n)                                                                \
    do {                                                                                  \
        _Pragma("unroll") for (int k_ = 0; k_ + 1 < (n); k_++)                            \
            if (M.lane == (base) + k_) M.pacc += cm_[k_ + 1] - cm_[k_];                   \
    } while (0)
#else
#define CPROF_T0()
#define CPROF_ADD(i, n)
#define CPROF_END()
#define CMARK(k, dep)
#define CMARK_DECL(n)
#define CMARK_SUM(base, n)
#endif

// Every load the wavefront has in flight has landed (vmcnt(0) of the gfx9 s_waitcnt encoding, the other counters untouched).  Placed in
// the PREHEADER of the run loops: the first link's inputs are fetched there, and without it the compiler's wait-count pass must
// assume at the top of EVERY iteration that those loads are still in flight with nothing issued behind them -- it then waits
// vmcnt(0) at the loop top, which drains the [K | y0] stores of the link before (backward run) or the block just requested for the
// next link (acceleration run: the prefetch bought nothing).  With the loads of the preheader retired the waits inside the loops
// count exactly (vmcnt(7) behind seven stores).
// which runs use the restructured link loops (A/B switch): 1 forward run in chunks, 2 branch-free backward run, 4 acceleration run in chunks,
// 8 the chunked runs in two phases (parent-independent work of four links first, the recursion on register operands after)
#ifndef GRBDA_CHUNK_MASK
#define GRBDA_CHUNK_MASK 15
#endif
#ifndef GRBDA_DRAIN_MASK
#define GRBDA_DRAIN_MASK 0
#endif
#define PREHEADER_DRAIN(bit)                                                \
    do {                                                                    \
        if constexpr ((GRBDA_DRAIN_MASK) & (bit)) __builtin_amdgcn_s_waitcnt(0x0F70); \
    } while (0)

template <class T>
struct ChainTables {
    cptr<ChainSeg> segs;
    cptr<ChainLink> links;
    cptr<ChainPair> pairs;
    cptr<ChainFree> frees;
    cptr<ChainDiff> diffs;
    cptr<ChainGen> gens;
    cptr<ChainGenBody> gbodies;
    cptr<int32_t> cints;
    cptr<T> consts;
    int n_segs, nq, nv, ori_repr;
    T a_root[6];
};

// LDS slots [slot][lane]; global slab rows [slot][lane]; the tile's inputs as coordinate-major slab rows
// Slab rows through a buffer descriptor instead of pointers: an experiment switch (make variant VFLAGS=-DGRBDA_SLAB_BUFFER=1), OFF in
// the product.  Measured in one run on one box (profiles/r6_slab_buffer_ab.txt): MIT Humanoid fp32 ABA 0.1427 -> 0.1449 ms, inverse
// dynamics 0.0809 -> 0.0815, JVRC-1 even, TelloWithArms 0.848 -> 0.824 (-3 %), Mini Cheetah fp64 in latency mode 0.0657 -> 0.0778 (+18 %).
#ifndef GRBDA_SLAB_BUFFER
#define GRBDA_SLAB_BUFFER 0
#endif
template <class T>
struct ChainMem {
    T *glb_u;           // wave's global slots (after the input rows): wave-uniform base
    const T *in_q_u, *in_qd_u, *in_x_u;   // the tile's input rows in the slab: wave-uniform bases
    T *out_u;
    int lane;
    unsigned lane_b;  // lane * sizeof(T): the byte offset every slab access adds to its wave-uniform row address
    // lanes of this tile whose D = S^T IA S had a pivot that is not positive (or not a number): the reference's ColPivHouseholderQR
    // (ClusterTreeNode.cpp:33-37) returns a least-squares answer there, the Cholesky factorisations here return NaN / Inf -- and the
    // state is COUNTED (grbda_spd_bad_pivots, the counter of the derivative solves).  A wave-uniform mask on the scalar unit: one
    // v_cmp that writes an SGPR pair and one s_or per pivot.
    mutable unsigned long long bad;
#ifdef GRBDA_EXP_NO_PIVOT  // (A/B builds: make variant VFLAGS=-DGRBDA_EXP_NO_PIVOT)
    __device__ __forceinline__ void pivot(T) const {}
#else
    __device__ __forceinline__ void pivot(T d) const { bad |= __builtin_amdgcn_ballot_w64(!(d > T(0))); }
#endif
    // (end of a tile: `valid` = ballot of the lanes that hold a state of the batch)
    __device__ __forceinline__ void flush_bad(unsigned long long *counter, int rows_valid) const
    {
        const unsigned long long live = rows_valid >= kWave ? ~0ull : ((1ull << rows_valid) - 1ull);
        const unsigned long long b = bad & live;
        if (b && counter && lane == 0) atomicAdd(counter, (unsigned long long)__builtin_popcountll(b));
        bad = 0;
    }
    int gmul;  // 1; 0 under GRBDA_CHAIN_DEBUG bit 3: every global slot aliases row 0 (same instructions, no slab traffic)
    int amask; // ~0; kSlotGlobal under GRBDA_CHAIN_DEBUG bit 4: the slots that overflowed the LDS alias row 0, the [K | y0] blocks stay

    template <int N>
    __device__ __forceinline__ void lds_ld(int s, T (&x)[N]) const
    {
        const T *p = reinterpret_cast<const T *>(grbda_smem) + (s * kWave + lane);
#pragma unroll
        for (int i = 0; i < N; i++) x[i] = p[i * kWave];
    }
    template <int N>
    __device__ __forceinline__ void lds_st(int s, const T (&x)[N]) const
    {
        T *p = reinterpret_cast<T *>(grbda_smem) + (s * kWave + lane);
#pragma unroll
        for (int i = 0; i < N; i++) p[i * kWave] = x[i];
    }
    // The wave's slab [input rows | global slots] through ONE buffer descriptor (GRBDA_SLAB_BUFFER=1): a row access is buffer_load /
    // buffer_store with the lane's byte offset in the VGPR that every access of the kernel shares, the row's byte offset in an SGPR
    // (one 32-bit shift) and the element's row inside the block as the instruction's immediate -- 32-bit address arithmetic only,
    // where the pointer form makes the compiler form 64-bit row addresses (s_mov + s_lshl_b64 and, where the row number is a loop
    // variable, a v_lshl_add_u64 per access: 554 -> 363 of them in the headline kernel's text).  It did not pay (see the switch).
    __amdgpu_buffer_rsrc_t rs;
    unsigned glb_b, q_b, qd_b, x_b, out_b;  // byte offsets of the regions inside the slab
    static constexpr unsigned kRowBytes = kWave * (unsigned)sizeof(T);
    __device__ __forceinline__ void set_slab(T *slab, int n_rows_total, int nq, int nv)
    {
        rs = __builtin_amdgcn_make_buffer_rsrc(slab, /*stride*/ 0, (int)((unsigned)n_rows_total * kRowBytes), 0x00020000);
        q_b = 0;
        qd_b = (unsigned)nq * kRowBytes;
        x_b = (unsigned)(nq + nv) * kRowBytes;
        out_b = x_b;
        glb_b = (unsigned)(nq + 2 * nv) * kRowBytes;
    }
    // element at row byte offset `so` (wave-uniform) + `imm` (a constant: folds into the instruction) of this lane
    __device__ __forceinline__ T bld(unsigned so, int imm) const
    {
        if constexpr (sizeof(T) == 4) {
            return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)lane_b + imm, (int)so, 0));
        } else {
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            const u2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)lane_b + imm, (int)so, 0);
            return __builtin_bit_cast(T, v);
        }
    }
    __device__ __forceinline__ void bst(unsigned so, int imm, T v) const
    {
        if constexpr (sizeof(T) == 4) {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, (int)lane_b + imm, (int)so, 0);
        } else {
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), rs, (int)lane_b + imm, (int)so, 0);
        }
    }
    // GRBDA_KLDS (unit 3, aba_chain_lm_kernel<float, NW>): the program's "slab" blocks [K | y0] are LDS objects unless their slot number
    // carries kSlotGlobal (plan.cpp, build_chain's k_lds)
    template <int N>
    __device__ __forceinline__ void glb_ld(int s, T (&x)[N]) const
    {
#if GRBDA_KLDS
        if (!(s & kSlotGlobal)) {
            lds_ld(s, x);
            return;
        }
#endif
#if GRBDA_SLAB_BUFFER
        const unsigned so = glb_b + (unsigned)((s & ~kSlotGlobal) * gmul) * kRowBytes;
#pragma unroll
        for (int i = 0; i < N; i++) x[i] = bld(so, i * (int)kRowBytes);
#else
        const char *p = reinterpret_cast<const char *>(glb_u + (size_t)(unsigned)((s & ~kSlotGlobal) * gmul * kWave));
#pragma unroll
        for (int i = 0; i < N; i++) x[i] = *reinterpret_cast<const T *>(p + i * (kWave * (int)sizeof(T)) + (size_t)lane_b);
#endif
    }
    template <int N>
    __device__ __forceinline__ void glb_st(int s, const T (&x)[N]) const
    {
#if GRBDA_KLDS
        if (!(s & kSlotGlobal)) {
            lds_st(s, x);
            return;
        }
#endif
#if GRBDA_SLAB_BUFFER
        const unsigned so = glb_b + (unsigned)((s & ~kSlotGlobal) * gmul) * kRowBytes;
#pragma unroll
        for (int i = 0; i < N; i++) bst(so, i * (int)kRowBytes, x[i]);
#else
        char *p = reinterpret_cast<char *>(glb_u + (size_t)(unsigned)((s & ~kSlotGlobal) * gmul * kWave));
#pragma unroll
        for (int i = 0; i < N; i++) *reinterpret_cast<T *>(p + i * (kWave * (int)sizeof(T)) + (size_t)lane_b) = x[i];
#endif
    }
    // accumulators of branching bodies: LDS, or the global slab when the plan could not fit them (rare accesses)
    template <int N>
    __device__ __forceinline__ void acc_ld(int s, T (&x)[N]) const
    {
        if (s & kSlotGlobal) glb_ld(s & amask, x);
        else lds_ld(s, x);
    }
    template <int N>
    __device__ __forceinline__ void acc_st(int s, const T (&x)[N]) const
    {
        if (s & kSlotGlobal) glb_st(s & amask, x);
        else lds_st(s, x);
    }
    __device__ __forceinline__ T row_ld(const T *base_u, int j) const
    {
        return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base_u + (size_t)(unsigned)(j * kWave)) + (size_t)lane_b);
    }
#if GRBDA_SLAB_BUFFER
    __device__ __forceinline__ T q(int j) const { return bld(q_b + (unsigned)j * kRowBytes, 0); }
    __device__ __forceinline__ T qd(int j) const { return bld(qd_b + (unsigned)j * kRowBytes, 0); }
    __device__ __forceinline__ T x(int j) const { return bld(x_b + (unsigned)j * kRowBytes, 0); }
#else
    __device__ __forceinline__ T q(int j) const { return row_ld(in_q_u, j); }
    __device__ __forceinline__ T qd(int j) const { return row_ld(in_qd_u, j); }
    __device__ __forceinline__ T x(int j) const { return row_ld(in_x_u, j); }
#endif
#ifdef GRBDA_CHAIN_PROFILE
    mutable unsigned long long pacc;
#endif
    int out_row;  // >= 0: the result rows [coordinate][lane] live in LDS from this row on (ChainProgram::out_lds); -1: in the slab
    // a result row read back (the differential's forward segment leaves a partial torque its backward segment completes)
    __device__ __forceinline__ T got(int j) const
    {
#if GRBDA_SLAB_BUFFER
        return bld(out_b + (unsigned)j * kRowBytes, 0);
#else
        return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(out_u + (size_t)(unsigned)(j * kWave)) + (size_t)lane_b);
#endif
    }
    // (forward dynamics: rows in LDS or in the slab)
    // (forward dynamics: rows in LDS or in the slab; a wave-uniform branch, so that the store is a DS or a GLOBAL instruction: a
    // flat store counts on both memory counters and turns every later wait of the run into vmcnt(0) lgkmcnt(0).  The fp64 kernels
    // that spill keep the flat store: the branch cost them 40 % in round 3.)
    __device__ __forceinline__ void put_f(int j, T v) const
    {
        if constexpr (sizeof(T) == 4) {
            if (out_row >= 0) reinterpret_cast<T *>(grbda_smem)[(out_row + j) * kWave + lane] = v;
            else put(j, v);
        } else {
            out_f[j * kWave + lane] = v;
        }
    }
    // the result rows through ONE generic pointer (flat stores reach LDS and the slab alike): no branch per store
    T *out_f;
    __device__ __forceinline__ void set_out(int row)
    {
        out_row = row;
        out_f = row >= 0 ? reinterpret_cast<T *>(grbda_smem) + row * kWave : out_u;
    }
    __device__ __forceinline__ void put(int j, T v) const
    {
#if GRBDA_SLAB_BUFFER
        bst(out_b + (unsigned)j * kRowBytes, 0, v);
#else
        *reinterpret_cast<T *>(reinterpret_cast<char *>(out_u + (size_t)(unsigned)(j * kWave)) + (size_t)lane_b) = v;
#endif
    }
};

// c = v x (z qd) for a joint about z: (v1, -v0, 0, v4, -v3, 0) qd   (Spatial.h:131-143)
template <class T>
__device__ __forceinline__ void vxz(const T (&v)[6], T qd, T (&c)[6])
{
    c[0] = v[1] * qd; c[1] = -v[0] * qd; c[2] = 0;
    c[3] = v[4] * qd; c[4] = -v[3] * qd; c[5] = 0;
}

// pA = v x* (I v) for constant packed inertia (ClusterTreeDynamics.cpp:95-98)
template <class T, class I21>
__device__ __forceinline__ void bias_force(const I21 &I, const T (&v)[6], T (&p)[6])
{
    T Iv[6];
    symv_c(I, v, Iv);
    crf(v, Iv, p);
}

// Axisymmetric rotor evaluated at q = 0 (plan.cpp): a body whose inertia is invariant under rotation about its joint
// axis z has the spatial inertia  [[A,0,0,0,k,0],[0,A,0,-k,0,0],[0,0,B,0,0,0],[0,-k,0,m,0,0],[k,0,0,0,m,0],[0,0,0,0,0,m]]
// (k = m c_z), so I v and I c cost 10 / 8 operations instead of 36 / 24, h = I[:, z] = B e_z makes the joint-space
// bias b = pA_z, and the force per unit rotor acceleration X0^T h is a plan constant (ChainLink::rpre).
// In: parent velocity vp, rotor rate qdr.  Out: b (bias torque about the rotor axis), tp = X0^T (pA + I c) at the parent.
template <class T>
__device__ __forceinline__ void rotor_terms(cptr<T> Cr, const T (&vp)[6], T qdr, T &b, T (&tp)[6])
{
    cptr<T> Ir = Cr + 12;
    const T A0 = Ir[sidx(0, 0)], A1 = Ir[sidx(1, 1)], Bz = Ir[sidx(2, 2)], k04 = Ir[sidx(0, 4)], k13 = Ir[sidx(1, 3)];
    const T m3 = Ir[sidx(3, 3)], m4 = Ir[sidx(4, 4)], m5 = Ir[sidx(5, 5)];
    T E0[9], vr[6];
#pragma unroll
    for (int j = 0; j < 9; j++) E0[j] = Cr[j];
    xmotion(E0, Cr + 9, vp, vr);
    vr[2] += qdr;
    // I v
    const T Iv[6] = {A0 * vr[0] + k04 * vr[4], A1 * vr[1] + k13 * vr[3], Bz * vr[2],
                     m3 * vr[3] + k13 * vr[1], m4 * vr[4] + k04 * vr[0], m5 * vr[5]};
    T pA[6];
    crf(vr, Iv, pA);
    b = pA[2];
    // c = v x (z qdr) = (v1, -v0, 0, v4, -v3, 0) qdr ;  t = pA + I c
    const T c0 = vr[1] * qdr, c1 = -vr[0] * qdr, c3 = vr[4] * qdr, c4 = -vr[3] * qdr;
    T t[6] = {pA[0] + A0 * c0 + k04 * c4, pA[1] + A1 * c1 + k13 * c3, pA[2],
              pA[3] + m3 * c3 + k13 * c1, pA[4] + m4 * c4 + k04 * c0, pA[5]};
    xforce_inv(E0, Cr + 9, t, tp);
}

// E-dependent part of one link of the backward run: F = X^T h, psic = X^T t, IAc = X^T IA X.  perm >= 0: the tree rotation is the
// cyclic permutation P_perm (devmath.h, rzp_*), wave-uniform switch; otherwise the general rotation E = Rz Et.
template <class T, int K, class A21, class R3>
__device__ __forceinline__ void link_up_p(T s, T c, const R3 &r, const T (&h)[6], const T (&t)[6], const A21 &IA, T (&F)[6],
                                          T (&psic)[6], T (&IAc)[21])
{
    xforce_inv_p<T, K>(s, c, r, h, F);
    xforce_inv_p<T, K>(s, c, r, t, psic);
    congruence_p<T, K>(s, c, r, IA, IAc);
}
template <class T>
__device__ __forceinline__ int perm_if(bool on, int perm)
{
    return on && (GRBDA_PERM_F64 || sizeof(T) == 4) ? perm : -1;
}
template <class T, class A21>
__device__ __forceinline__ void link_up(int perm, T s, T c, cptr<T> C, const T (&h)[6], const T (&t)[6], const A21 &IA,
                                        T (&F)[6], T (&psic)[6], T (&IAc)[21])
{
    if (perm < 0) {
        T E[9];
        rotate_z(s, c, C, E);
        xforce_inv(E, C + 9, h, F);
        xforce_inv(E, C + 9, t, psic);
        congruence(E, C + 9, IA, IAc);
    } else if (perm == 0) {
        link_up_p<T, 0, A21>(s, c, C + 9, h, t, IA, F, psic, IAc);
    } else if (perm == 1) {
        link_up_p<T, 1, A21>(s, c, C + 9, h, t, IA, F, psic, IAc);
    } else {
        link_up_p<T, 2, A21>(s, c, C + 9, h, t, IA, F, psic, IAc);
    }
}
template <class T>
__device__ __forceinline__ void link_down(int perm, T s, T c, cptr<T> C, const T (&vp)[6], T (&v)[6])
{
    if (perm < 0) {
        T E[9];
        rotate_z(s, c, C, E);
        xmotion(E, C + 9, vp, v);
    } else if (perm == 0) {
        xmotion_p<T, 0>(s, c, C + 9, vp, v);
    } else if (perm == 1) {
        xmotion_p<T, 1>(s, c, C + 9, vp, v);
    } else {
        xmotion_p<T, 2>(s, c, C + 9, vp, v);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// forward run: TreeModel::forwardKinematics (TreeModel.cpp:6-32) along a chain, root side first
// ---------------------------------------------------------------------------------------------------------------
// SVG: the program keeps SOME of the [sin, cos, v] blocks of the links in the wave's global slab instead of LDS (ChainProgram::
// sv_global: chains too long for the LDS budget); a compile-time property of the run so that the common case carries
// neither the test nor the prefetch registers.
template <class T>
__device__ __forceinline__ void link_down2(int perm, T s, T c, cptr<T> C, const T (&vp)[6], const T (&ap)[6], T (&v)[6], T (&a)[6])
{
    if (perm < 0) {
        T E[9];
        rotate_z(s, c, C, E);
        xmotion(E, C + 9, vp, v);
        xmotion(E, C + 9, ap, a);
    } else if (perm == 0) {
        xmotion_p<T, 0>(s, c, C + 9, vp, v);
        xmotion_p<T, 0>(s, c, C + 9, ap, a);
    } else if (perm == 1) {
        xmotion_p<T, 1>(s, c, C + 9, vp, v);
        xmotion_p<T, 1>(s, c, C + 9, ap, a);
    } else {
        xmotion_p<T, 2>(s, c, C + 9, vp, v);
        xmotion_p<T, 2>(s, c, C + 9, ap, a);
    }
}
template <class T>
__device__ __forceinline__ void link_force_up(int perm, T s, T c, cptr<T> C, const T (&f)[6], T (&o)[6])
{
    if (perm < 0) {
        T E[9];
        rotate_z(s, c, C, E);
        xforce_inv(E, C + 9, f, o);
    } else if (perm == 0) {
        xforce_inv_p<T, 0>(s, c, C + 9, f, o);
    } else if (perm == 1) {
        xforce_inv_p<T, 1>(s, c, C + 9, f, o);
    } else {
        xforce_inv_p<T, 2>(s, c, C + 9, f, o);
    }
}

template <class T, bool SVG>
__device__ __forceinline__ void run_fwd(const ChainTables<T> &P, const ChainMem<T> &M, const ChainSeg &sg)
{
    T vp[6];
    {
        const ChainLink l0 = load_rec(P.links + sg.first);
        if (l0.lds_pv != -1) {
            if constexpr (SVG) M.acc_ld(l0.lds_pv, vp);
            else M.lds_ld(l0.lds_pv, vp);
        } else {
#pragma unroll
            for (int j = 0; j < 6; j++) vp[j] = 0;
        }
    }
    // the inputs of the next link travel (global slab rows, L2 latency) while the current link is computed
    ChainLink l = load_rec(P.links + sg.first);
    T qi = M.q(l.q_index), qdi_in = M.qd(l.v_index);
    PREHEADER_DRAIN(1);
    for (int i = 0; i < sg.count; i++) {
        CMARK_DECL(6);
        CMARK(0, vp[0]);
        const bool more = i + 1 < sg.count;
        const ChainLink ln = load_rec(P.links + (sg.first + (more ? i + 1 : i)));
        T qn = 0, qdn = 0;
        if (more) {
            qn = M.q(ln.q_index);
            qdn = M.qd(ln.v_index);
        }
        cptr<T> C = P.consts + l.cofs;
        const T g0 = C[kBodyConstFixed];
        T blk[8], v[6];
        const T ang = g0 * qi;
        CMARK(1, ang);
        sincos_t(ang, &blk[0], &blk[1]);
        CMARK(2, blk[0]);
        link_down(perm_if<T>(GRBDA_PERM_LINK, l.perm), blk[0], blk[1], C, vp, v);
        v[2] += g0 * qdi_in;
        CMARK(3, v[2]);
#pragma unroll
        for (int j = 0; j < 6; j++) {
            blk[2 + j] = v[j];
            vp[j] = v[j];
        }
        if constexpr (SVG) M.acc_st(l.lds_sv, blk);
        else M.lds_st(l.lds_sv, blk);
        CMARK(4, v[0]);
        l = ln;
        qi = qn;
        qdi_in = qdn;
        CMARK(5, qi);
        CMARK_SUM(0, 6);
    }
}

// The same run with the slab rows of FOUR links requested at once (kChunk): a link of the forward run is ~60 instructions, far
// shorter than the L2 / Infinity Cache round trip of its two input rows, so the one-link-ahead prefetch of run_fwd exposes most of
// that latency at every link (profiles/r5_chain_phase_profile.txt: ~780 cycles per link at the loop top); here it is exposed once
// per chunk.  Runs of the URDF robots are 3-7 links long.
constexpr int kChunk = 4;
template <class T>
__device__ __forceinline__ void run_fwd_c(const ChainTables<T> &P, const ChainMem<T> &M, const ChainSeg &sg)
{
    // Two phases per chunk (GRBDA_CHUNK_MASK bit 3): A -- everything of a link that does not depend on its parent (constants, sin / cos,
    // E = Rz(q) Et, the offset r, the joint rate), four links side by side: the scalar loads of all four are in flight together and the
    // arithmetic of one link fills the waits of another; B -- the recursion proper, v_i = X_i v_(i-1) + z qd_i, on REGISTER operands only
    // (no constant fetched on the critical path: the phase profile showed ~780 of a link's ~1 500 cycles in serialized scalar waits).
    constexpr bool TWO_PHASE = ((GRBDA_CHUNK_MASK) & 8) != 0;
    T vp[6];
    for (int base = 0; base < sg.count; base += kChunk) {
        const int n = sg.count - base;
        // (ONE basic block: the four records -- positions past the end of the run repeat its last link -- then every slab row; a branch per
        // link made a block of its own of every record, each with its own scalar wait: ~740 cycles per link in the phase profile)
        ChainLink L[kChunk];
        T qv[kChunk], qdv[kChunk];
#pragma unroll
        for (int u = 0; u < kChunk; u++) L[u] = load_rec(P.links + (sg.first + base + (u < n ? u : n - 1)));
#pragma unroll
        for (int u = 0; u < kChunk; u++) {
            qv[u] = M.q(L[u].q_index);
            qdv[u] = M.qd(L[u].v_index);
        }
        if (base == 0) {
            if (L[0].lds_pv != -1) {
                M.lds_ld(L[0].lds_pv, vp);
            } else {
#pragma unroll
                for (int j = 0; j < 6; j++) vp[j] = 0;
            }
        }
        if constexpr (TWO_PHASE) {
            T E[kChunk][9], r[kChunk][3], sc[kChunk][2], gq[kChunk];
#pragma unroll
            for (int u = 0; u < kChunk; u++) {
                if (u < n) {
                    cptr<T> C = P.consts + L[u].cofs;
                    const T g0 = C[kBodyConstFixed];
                    sincos_t(g0 * qv[u], &sc[u][0], &sc[u][1]);
                    rotate_z(sc[u][0], sc[u][1], C, E[u]);
#pragma unroll
                    for (int j = 0; j < 3; j++) r[u][j] = C[9 + j];
                    gq[u] = g0 * qdv[u];
                }
            }
#pragma unroll
            for (int u = 0; u < kChunk; u++) {
                if (u < n) {
                    T blk[8], v[6];
                    xmotion(E[u], r[u], vp, v);
                    v[2] += gq[u];
                    blk[0] = sc[u][0];
                    blk[1] = sc[u][1];
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        blk[2 + j] = v[j];
                        vp[j] = v[j];
                    }
                    M.lds_st(L[u].lds_sv, blk);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < kChunk; u++) {
                if (u < n) {
                    cptr<T> C = P.consts + L[u].cofs;
                    const T g0 = C[kBodyConstFixed];
                    T blk[8], v[6];
                    sincos_t(g0 * qv[u], &blk[0], &blk[1]);
                    link_down(perm_if<T>(GRBDA_PERM_LINK, L[u].perm), blk[0], blk[1], C, vp, v);
                    v[2] += g0 * qdv[u];
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        blk[2 + j] = v[j];
                        vp[j] = v[j];
                    }
                    M.lds_st(L[u].lds_sv, blk);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// leaf RevolutePairWithRotor cluster, backward: bodies link1 (on the parent body P), link2 (on link1), two
// axisymmetric rotors on P evaluated at q = 0; coordinates y = (link1 angle, link2 angle), rotor angles G_r y.
// Per body the composite inertia / bias are pushed to the tree parent; the cluster-level terms
//   D = sum d_i G_i^T G_i + chain terms,  F = sum f_i G_i,  u = tau - sum G_i^T b_i
// give K = D^-1 F^T, y0 = D^-1 u and the correction -F D^-1 F^T, + F D^-1 u on P (kernels.hip, aba_bwd_static).
// Out: (IA, psi) = what P receives (without the rotors' constant X0^T I X0, which is part of P's constants).
// ---------------------------------------------------------------------------------------------------------------
template <class T, int K>
__device__ __forceinline__ void xmotion_k(T s, T c, const T (&E)[9], cptr<T> C, const T (&m)[6], T (&o)[6])
{
    if constexpr (K < 0) xmotion(E, C + 9, m, o);
    else xmotion_p<T, K>(s, c, C + 9, m, o);
}
template <class T, int K>
__device__ __forceinline__ void xforce_inv_k(T s, T c, const T (&E)[9], cptr<T> C, const T (&f)[6], T (&o)[6])
{
    if constexpr (K < 0) xforce_inv(E, C + 9, f, o);
    else xforce_inv_p<T, K>(s, c, C + 9, f, o);
}
template <class T, int K, class A21>
__device__ __forceinline__ void congruence_k(T s, T c, const T (&E)[9], cptr<T> C, const A21 &A, T (&B)[21])
{
    if constexpr (K < 0) congruence(E, C + 9, A, B);
    else congruence_p<T, K>(s, c, C + 9, A, B);
}

// K1, K2: compile-time ChainPair::perm of the two links (-1: general rotation)
template <class T, bool OSIM, int K1, int K2>
__device__ __forceinline__ void pair_bwd_k(const ChainTables<T> &P, const ChainMem<T> &M, const ChainPair &pr, T (&IA)[21],
                                           T (&psi)[6])
{
    cptr<T> C1 = P.consts + pr.cofs[0], C2 = P.consts + pr.cofs[1];
    T vp[6];
    M.acc_ld(pr.lds_pv, vp);  // (LDS, or the global slab in programs with ChainProgram::sv_global)
    const T y1 = M.q(pr.q_index), y2 = M.q(pr.q_index + 1);
    const T yd1 = M.qd(pr.v_index), yd2 = M.qd(pr.v_index + 1);
    T u[2] = {M.x(pr.v_index), M.x(pr.v_index + 1)};
    T D00 = 0, D01 = 0, D11 = 0;
    T F0[6], F1[6];  // columns of F (force at P per unit y1dd / y2dd)

    // ---- kinematics of the two links ----
    T s1, c1, s2, c2, E1[9], E2[9], v1[6], v2[6], ch1[6], ch2[6];
    sincos_t(y1, &s1, &c1);
    if constexpr (K1 < 0) rotate_z(s1, c1, C1, E1);
    xmotion_k<T, K1>(s1, c1, E1, C1, vp, v1);
    v1[2] += yd1;
    vxz(v1, yd1, ch1);
    sincos_t(y2, &s2, &c2);
    if constexpr (K2 < 0) rotate_z(s2, c2, C2, E2);
    xmotion_k<T, K2>(s2, c2, E2, C2, v1, v2);
    v2[2] += yd2;
    vxz(v2, yd2, ch2);
    // in-cluster bias acceleration of link2: ccl2 = ch2 + X2 ch1 (GenericJoint.cpp:430-450)
    T ccl2[6];
    xmotion_k<T, K2>(s2, c2, E2, C2, ch1, ccl2);
#pragma unroll
    for (int j = 0; j < 6; j++) ccl2[j] += ch2[j];

    // ---- link2 (leaf): IA2 = I2 ----
    T IA1[21], psi1[6];
    {
        cptr<T> I2 = C2 + 12;
        T p2[6], h2[6];
        bias_force(I2, v2, p2);
#pragma unroll
        for (int i = 0; i < 6; i++) h2[i] = I2[sidx(i, 2)];
        T bj = p2[2];
#pragma unroll
        for (int j = 0; j < 6; j++) bj += h2[j] * ccl2[j];
        u[1] -= bj;
        D11 += h2[2];
        // composite to link1: X2^T (p2 + I2 ch2), X2^T I2 X2
        T t[6];
        symv_c(I2, ch2, t);
#pragma unroll
        for (int j = 0; j < 6; j++) t[j] += p2[j];
        xforce_inv_k<T, K2>(s2, c2, E2, C2, t, psi1);
        congruence_k<T, K2>(s2, c2, E2, C2, I2, IA1);
        // joint-space coupling through the chain: f = X2^T h2 at link1, then at P
        T f[6];
        xforce_inv_k<T, K2>(s2, c2, E2, C2, h2, f);
        D01 += f[2];  // Hc (G1 G2^T + G2 G1^T) with G1 = (1, 0), G2 = (0, 1)
        xforce_inv_k<T, K1>(s1, c1, E1, C1, f, F1);
    }
    // ---- link1 ----
    {
        cptr<T> I1 = C1 + 12;
        T p1[6];
        bias_force(I1, v1, p1);
#pragma unroll
        for (int j = 0; j < 21; j++) IA1[j] += I1[j];
#pragma unroll
        for (int j = 0; j < 6; j++) psi1[j] += p1[j];
        T h1[6];
#pragma unroll
        for (int i = 0; i < 6; i++) h1[i] = IA1[sidx(i, 2)];
        // ccl1 = ch1 = (., ., 0, ., ., 0)
        const T bj = psi1[2] + h1[0] * ch1[0] + h1[1] * ch1[1] + h1[3] * ch1[3] + h1[4] * ch1[4];
        u[0] -= bj;
        D00 += h1[2];
        T t[6];
        symv_z(IA1, ch1, t);
#pragma unroll
        for (int j = 0; j < 6; j++) t[j] += psi1[j];
        xforce_inv_k<T, K1>(s1, c1, E1, C1, t, psi);
        congruence_k<T, K1>(s1, c1, E1, C1, IA1, IA);
        xforce_inv_k<T, K1>(s1, c1, E1, C1, h1, F0);
    }
    // ---- rotors (q = 0): bias to P, joint-space terms with their G rows ----
#pragma unroll
    for (int r = 0; r < 2; r++) {
        cptr<T> Cr = P.consts + pr.cofs[2 + r];
        cptr<T> Rp = P.consts + pr.rpre[r];
        const T ga = Cr[kBodyConstFixed], gb = Cr[kBodyConstFixed + 1];
        T bj, tp[6];
        rotor_terms(Cr, vp, ga * yd1 + gb * yd2, bj, tp);
        u[0] -= ga * bj;
        u[1] -= gb * bj;
        D00 += Rp[6] * ga * ga;
        D01 += Rp[6] * ga * gb;
        D11 += Rp[6] * gb * gb;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            F0[j] += Rp[j] * ga;
            F1[j] += Rp[j] * gb;
            psi[j] += tp[j];
        }
    }
    // ---- D^-1 (2 x 2, SPD), K = D^-1 F^T, y0 = D^-1 u ----
    // D = L L^T and triangular solves (see diff_bwd: the adjugate / determinant form cancels when D is nearly rank one)
    const T r00 = rsqrt_t(D00), l10 = D01 * r00, r11 = rsqrt_t(D11 - l10 * l10);
    M.pivot(D00);
    M.pivot(D11 - l10 * l10);
    const T i00 = r00 * r00 + (l10 * r00 * r11) * (l10 * r00 * r11), i01 = -(l10 * r00) * r11 * r11, i11 = r11 * r11;
    T blk[14];  // [K row 0 (6)][K row 1 (6)][y0 (2)]
    auto solve2 = [&](T b0, T b1, T &x0, T &x1) {
        const T y0 = b0 * r00, y1 = (b1 - l10 * y0) * r11;
        x1 = y1 * r11;
        x0 = (y0 - l10 * x1) * r00;
    };
#pragma unroll
    for (int j = 0; j < 6; j++) solve2(F0[j], F1[j], blk[j], blk[6 + j]);
    solve2(u[0], u[1], blk[12], blk[13]);
    M.glb_st(pr.glb_k, blk);
    if constexpr (OSIM) {  // what the force-propagator walk of the contact frames needs (osim_chain_kernel)
        const T ex[7] = {i00, i01, i11, s1, c1, s2, c2};
        M.glb_st(pr.glb_k + 14, ex);
    }
    // ---- correction on P: IA -= F K, psi += F y0 ----
#pragma unroll
    for (int r = 0; r < 6; r++) {
        psi[r] += F0[r] * blk[12] + F1[r] * blk[13];
#pragma unroll
        for (int cc = r; cc < 6; cc++) IA[sidx(r, cc)] -= F0[r] * blk[cc] + F1[r] * blk[6 + cc];
    }
}
template <class T, bool OSIM>
__device__ __forceinline__ void pair_bwd(const ChainTables<T> &P, const ChainMem<T> &M, const ChainPair &pr, T (&IA)[21],
                                         T (&psi)[6])
{
    // (the knee-ankle pairs of the MIT Humanoid: both tree rotations the identity)
    if (perm_if<T>(GRBDA_PERM_PAIR, pr.perm[0]) == 0 && pr.perm[1] == 0) pair_bwd_k<T, OSIM, 0, 0>(P, M, pr, IA, psi);
    else pair_bwd_k<T, OSIM, -1, -1>(P, M, pr, IA, psi);
}

template <class T>
__device__ __forceinline__ void pair_acc(const ChainTables<T> &P, const ChainMem<T> &M, const ChainPair &pr)
{
    T blk[14], va[12];
    M.glb_ld(pr.glb_k, blk);
    M.lds_ld(pr.lds_pva, va);
    T y0 = blk[12], y1 = blk[13];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        y0 -= blk[j] * va[6 + j];
        y1 -= blk[6 + j] * va[6 + j];
    }
    M.put_f(pr.v_index, y0);
    M.put_f(pr.v_index + 1, y1);
}


// ---------------------------------------------------------------------------------------------------------------
// Two-rotor differential cluster (plan.h, ChainDiff): the implicit clusters of Tello (src/Robots/Tello.cpp:77-261,
// GenericImplicit constraints; GenericJoint.cpp:380-450 for how G and g enter the recursions).
//
// Constraint: phi_r(q) = sum_t coef_t prod_f F_f(w_f . q + b_f), F in {sin, cos, id}; the reference differentiates its
// phi lambdas with CasADi, here K = dphi/dq and Kdot qd come from the polynomial itself.  The plan compiler
// (plan.cpp, emit_diff_program) lists the distinct arguments a = w . q + b and the distinct ATOMS F(a) the terms are
// products of, and sorts the terms of every row by their number of factors, so that the loops below carry no
// per-factor branches:
//   pass A   per argument: a, sin a, cos a; per atom the pair (F, F') into the work space W[3 atom + {0, 1}]
//   pass K   per term: dterm/da_f = coef F'_f prod_{g != f} F_g, K[r][:] += that * w_f        (reverse mode)
//   X = -Kd^-1 Ki (columns: rotors independent, links dependent), link rates X yd
//   pass A'  per atom: ad = w . qd_span; W <- second-order Taylor coefficients (F, F' ad, F'' ad^2 / 2) along qd_span
//   pass B   per term: t^2 coefficient of the product of its atoms' series; Kdot qd = 2 * sum   (forward mode: with
//            q(t) = q + qd t, phi'' = Kdot qd), g = -Kd^-1 Kdot qd
// ---------------------------------------------------------------------------------------------------------------
template <class T, class TB>
__device__ __forceinline__ void diff_constraint(const TB &P, const ChainMem<T> &M, int tofs_i, int lds_w, const T (&qs)[4], T yd0,
                                                T yd1, T (&X)[4], T (&g)[2], T (&qdl)[2])
{
    cptr<int32_t> ip = P.cints + tofs_i;
    const int n_args = ip[0], n_atoms = ip[1];
    cptr<T> dp = P.consts + ip[2];
    for (int i = 0; i < n_args; i++) {
        cptr<T> w = dp + 5 * i;
        const T a = w[4] + w[0] * qs[0] + w[1] * qs[1] + w[2] * qs[2] + w[3] * qs[3];
        const int as = ip[4 + 3 * i], ac = ip[5 + 3 * i], al = ip[6 + 3 * i];
        if (as >= 0 || ac >= 0) {
            T sn, cs;
#ifdef GRBDA_EXP_DIFF_PRECISE
            sincos_precise(a, &sn, &cs);
#elif defined(GRBDA_EXP_DIFF_HW_SINCOS)
            sincos_t(a, &sn, &cs);
#else
            sincos_cw(a, &sn, &cs);  // (devmath.h: the hardware sine / cosine cost the fp32 differentials a decimal digit)
#endif
            if (as >= 0) {
                const T v[2] = {sn, cs};
                M.lds_st(lds_w + as, v);
            }
            if (ac >= 0) {
                const T v[2] = {cs, -sn};
                M.lds_st(lds_w + ac, v);
            }
        }
        if (al >= 0) {
            const T v[2] = {a, T(1)};
            M.lds_st(lds_w + al, v);
        }
    }
    cptr<int32_t> tp = ip + 4 + 3 * n_args;
    cptr<T> kp = dp + 5 * (n_args + n_atoms);
    T K[2][4];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        T k0 = 0, k1 = 0, k2 = 0, k3 = 0;
        const int n1 = tp[0], n2 = tp[1], n3 = tp[2];
        tp += 3;
        for (int t = 0; t < n1; t++) {
            T f[2];
            M.lds_ld(lds_w + tp[0], f);
            const T g0 = kp[0] * f[1];
            k0 += g0 * kp[1]; k1 += g0 * kp[2]; k2 += g0 * kp[3]; k3 += g0 * kp[4];
            tp += 1;
            kp += 5;
        }
        for (int t = 0; t < n2; t++) {
            T f[2], h[2];
            M.lds_ld(lds_w + tp[0], f);
            M.lds_ld(lds_w + tp[1], h);
            const T c = kp[0];
            const T g0 = c * f[1] * h[0], g1 = c * f[0] * h[1];
            k0 += g0 * kp[1] + g1 * kp[5]; k1 += g0 * kp[2] + g1 * kp[6];
            k2 += g0 * kp[3] + g1 * kp[7]; k3 += g0 * kp[4] + g1 * kp[8];
            tp += 2;
            kp += 9;
        }
        for (int t = 0; t < n3; t++) {
            T f[2], h[2], m[2];
            M.lds_ld(lds_w + tp[0], f);
            M.lds_ld(lds_w + tp[1], h);
            M.lds_ld(lds_w + tp[2], m);
            const T c = kp[0];
            const T g0 = c * f[1] * h[0] * m[0], g1 = c * f[0] * h[1] * m[0], g2 = c * f[0] * h[0] * m[1];
            k0 += g0 * kp[1] + g1 * kp[5] + g2 * kp[9]; k1 += g0 * kp[2] + g1 * kp[6] + g2 * kp[10];
            k2 += g0 * kp[3] + g1 * kp[7] + g2 * kp[11]; k3 += g0 * kp[4] + g1 * kp[8] + g2 * kp[12];
            tp += 3;
            kp += 13;
        }
        K[r][0] = k0; K[r][1] = k1; K[r][2] = k2; K[r][3] = k3;
    }
    // Kd = K[:, links], Ki = K[:, rotors]
#ifdef GRBDA_EXP_DIFF_DIV
    const T idet = T(1) / (K[0][2] * K[1][3] - K[0][3] * K[1][2]);
#elif defined(GRBDA_EXP_DIFF_F64DET)
    const T idet = T(1.0 / ((double)K[0][2] * (double)K[1][3] - (double)K[0][3] * (double)K[1][2]));
#elif defined(GRBDA_EXP_DIFF_DIVDET)
    const T idet = T(1) / __builtin_fmaf(K[0][2], K[1][3], -(K[0][3] * K[1][2]));
#else
    const T idet = rcp_t(K[0][2] * K[1][3] - K[0][3] * K[1][2]);
#endif
    const T i00 = K[1][3] * idet, i01 = -K[0][3] * idet, i10 = -K[1][2] * idet, i11 = K[0][2] * idet;
    X[0] = -(i00 * K[0][0] + i01 * K[1][0]);
    X[1] = -(i00 * K[0][1] + i01 * K[1][1]);
    X[2] = -(i10 * K[0][0] + i11 * K[1][0]);
    X[3] = -(i10 * K[0][1] + i11 * K[1][1]);
    qdl[0] = X[0] * yd0 + X[1] * yd1;
    qdl[1] = X[2] * yd0 + X[3] * yd1;
    cptr<T> ap = dp + 5 * n_args;
    for (int at = 0; at < n_atoms; at++) {
        cptr<T> w = ap + 5 * at;
        const T ad = w[0] * yd0 + w[1] * yd1 + w[2] * qdl[0] + w[3] * qdl[1];
        T f[2];
        M.lds_ld(lds_w + 3 * at, f);
        const T t12[2] = {f[1] * ad, w[4] * f[0] * ad * ad};
        M.lds_st(lds_w + 3 * at + 1, t12);
    }
    tp = ip + 4 + 3 * n_args;
    T kd[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        T acc = 0;
        const int n1 = tp[0], n2 = tp[1], n3 = tp[2];
        tp += 3;
        for (int t = 0; t < n1; t++) {
            T f[3];
            M.lds_ld(lds_w + tp[0], f);
            acc += kp[0] * f[2];
            tp += 1;
            kp += 1;
        }
        for (int t = 0; t < n2; t++) {
            T f[3], h[3];
            M.lds_ld(lds_w + tp[0], f);
            M.lds_ld(lds_w + tp[1], h);
            acc += kp[0] * (f[0] * h[2] + f[1] * h[1] + f[2] * h[0]);
            tp += 2;
            kp += 1;
        }
        for (int t = 0; t < n3; t++) {
            T f[3], h[3], m[3];
            M.lds_ld(lds_w + tp[0], f);
            M.lds_ld(lds_w + tp[1], h);
            M.lds_ld(lds_w + tp[2], m);
            const T p0 = f[0] * h[0], p1 = f[0] * h[1] + f[1] * h[0], p2 = f[0] * h[2] + f[1] * h[1] + f[2] * h[0];
            acc += kp[0] * (p0 * m[2] + p1 * m[1] + p2 * m[0]);
            tp += 3;
            kp += 1;
        }
        kd[r] = acc + acc;
    }
    g[0] = -(i00 * kd[0] + i01 * kd[1]);
    g[1] = -(i10 * kd[0] + i11 * kd[1]);
}

// kinematics of the two links of a differential from the parent velocity: E, v, chat = v x z qd + z g
template <class T>
__device__ __forceinline__ void diff_links(cptr<T> C1, cptr<T> C2, const T (&sc)[4], const T (&vp)[6], T qd1, T qd2, T (&E1)[9],
                                           T (&E2)[9], T (&v1)[6], T (&v2)[6])
{
    rotate_z(sc[0], sc[1], C1, E1);
    xmotion(E1, C1 + 9, vp, v1);
    v1[2] += qd1;
    rotate_z(sc[2], sc[3], C2, E2);
    xmotion(E2, C2 + 9, v1, v2);
    v2[2] += qd2;
}

template <class T>
__device__ __forceinline__ void diff_fwd(const ChainTables<T> &P, const ChainMem<T> &M, const ChainDiff &d)
{
    T qs[4];
#pragma unroll
    for (int i = 0; i < 4; i++) qs[i] = M.q(d.q_index + d.qpos[i]);
    const T yd0 = M.qd(d.v_index), yd1 = M.qd(d.v_index + 1);
    T gx[10], qdl[2];
    if (d.tofs_i >= 0) {
        T X[4], g[2];
        diff_constraint(P, M, d.tofs_i, d.lds_w, qs, yd0, yd1, X, g, qdl);
#pragma unroll
        for (int j = 0; j < 4; j++) gx[j] = X[j];
        gx[4] = g[0];
        gx[5] = g[1];
    } else {  // explicit pair: the link angles are the coordinates
        gx[0] = 1; gx[1] = 0; gx[2] = 0; gx[3] = 1;
        gx[4] = gx[5] = 0;
        qdl[0] = yd0;
        qdl[1] = yd1;
    }
    sincos_cw(qs[2], &gx[6], &gx[7]);
    sincos_cw(qs[3], &gx[8], &gx[9]);
    M.glb_st(d.glb_k + 14, gx);
    if (d.lds_sv != -1) {
        cptr<T> C1 = P.consts + d.cofs[0], C2 = P.consts + d.cofs[1];
        T vp[6], E1[9], E2[9], v1[6], blk[8];
        if (d.lds_pv != -1) {
            M.acc_ld(d.lds_pv, vp);
        } else {  // the cluster hangs off the ground
#pragma unroll
            for (int j = 0; j < 6; j++) vp[j] = 0;
        }
        const T sc[4] = {gx[6], gx[7], gx[8], gx[9]};
        T v2[6];
        diff_links(C1, C2, sc, vp, qdl[0], qdl[1], E1, E2, v1, v2);
        blk[0] = gx[8];
        blk[1] = gx[9];
#pragma unroll
        for (int j = 0; j < 6; j++) blk[2 + j] = v2[j];
        M.acc_st(d.lds_sv, blk);
    }
}

// backward segment: pair_bwd with per-state G rows -- rotors (1, 0), (0, 1); links (X00, X01), (X10, X11) -- the bias
// accelerations g of the links, and child segments on link2.
template <class T, bool OSIM>
__device__ __forceinline__ void diff_bwd(const ChainTables<T> &P, const ChainMem<T> &M, const ChainDiff &d)
{
    cptr<T> C1 = P.consts + d.cofs[0], C2 = P.consts + d.cofs[1];
    T vp[6], gx[10];
    if (d.lds_pv != -1) {
        M.acc_ld(d.lds_pv, vp);
    } else {
#pragma unroll
        for (int j = 0; j < 6; j++) vp[j] = 0;
    }
    M.glb_ld(d.glb_k + 14, gx);
    const T X00 = gx[0], X01 = gx[1], X10 = gx[2], X11 = gx[3];
    const T yd0 = M.qd(d.v_index), yd1 = M.qd(d.v_index + 1);
    T u0 = M.x(d.v_index), u1 = M.x(d.v_index + 1);
    const T qd1 = X00 * yd0 + X01 * yd1, qd2 = X10 * yd0 + X11 * yd1;
    T E1[9], E2[9], v1[6], v2[6], ch1[6], ch2[6], ccl2[6];
    {
        const T sc[4] = {gx[6], gx[7], gx[8], gx[9]};
        diff_links(C1, C2, sc, vp, qd1, qd2, E1, E2, v1, v2);
    }
    vxz(v1, qd1, ch1);
    ch1[2] += gx[4];
    vxz(v2, qd2, ch2);
    ch2[2] += gx[5];
    xmotion(E2, C2 + 9, ch1, ccl2);
#pragma unroll
    for (int j = 0; j < 6; j++) ccl2[j] += ch2[j];
    T D00 = 0, D01 = 0, D11 = 0, F0[6], F1[6], IA[21], psi[6];
    T IA1[21], psi1[6];
    {   // ---- link2 ----
        cptr<T> Ib = P.consts + d.iofs;
        T IA2[21], p2[6], h2[6];
        bias_force(C2 + 12, v2, p2);
        if (d.lds_acc != -1) {
            T acc[27];
            M.acc_ld(d.lds_acc, acc);
#pragma unroll
            for (int j = 0; j < 21; j++) IA2[j] = Ib[j] + acc[j];
#pragma unroll
            for (int j = 0; j < 6; j++) p2[j] += acc[21 + j];
        } else {
#pragma unroll
            for (int j = 0; j < 21; j++) IA2[j] = Ib[j];
        }
#pragma unroll
        for (int i = 0; i < 6; i++) h2[i] = IA2[sidx(i, 2)];
        T bj = p2[2];
#pragma unroll
        for (int j = 0; j < 6; j++) bj += h2[j] * ccl2[j];
        u0 -= X10 * bj;
        u1 -= X11 * bj;
        D00 += h2[2] * X10 * X10;
        D01 += h2[2] * X10 * X11;
        D11 += h2[2] * X11 * X11;
        T t[6];
        symv(IA2, ch2, t);
#pragma unroll
        for (int j = 0; j < 6; j++) t[j] += p2[j];
        xforce_inv(E2, C2 + 9, t, psi1);
        congruence(E2, C2 + 9, IA2, IA1);
        T f[6], Fc[6];
        xforce_inv(E2, C2 + 9, h2, f);
        const T Hc = f[2];  // D += Hc (G1^T G2 + G2^T G1)
        D00 += 2 * Hc * X00 * X10;
        D01 += Hc * (X00 * X11 + X01 * X10);
        D11 += 2 * Hc * X01 * X11;
        xforce_inv(E1, C1 + 9, f, Fc);
#pragma unroll
        for (int j = 0; j < 6; j++) {
            F0[j] = Fc[j] * X10;
            F1[j] = Fc[j] * X11;
        }
    }
    {   // ---- link1 ----
        cptr<T> I1 = C1 + 12;
        T p1[6], h1[6];
        bias_force(I1, v1, p1);
#pragma unroll
        for (int j = 0; j < 21; j++) IA1[j] += I1[j];
#pragma unroll
        for (int j = 0; j < 6; j++) psi1[j] += p1[j];
#pragma unroll
        for (int i = 0; i < 6; i++) h1[i] = IA1[sidx(i, 2)];
        T bj = psi1[2];
#pragma unroll
        for (int j = 0; j < 6; j++) bj += h1[j] * ch1[j];
        u0 -= X00 * bj;
        u1 -= X01 * bj;
        D00 += h1[2] * X00 * X00;
        D01 += h1[2] * X00 * X01;
        D11 += h1[2] * X01 * X01;
        T t[6], Fc[6];
        symv(IA1, ch1, t);
#pragma unroll
        for (int j = 0; j < 6; j++) t[j] += psi1[j];
        xforce_inv(E1, C1 + 9, t, psi);
        congruence(E1, C1 + 9, IA1, IA);
        xforce_inv(E1, C1 + 9, h1, Fc);
#pragma unroll
        for (int j = 0; j < 6; j++) {
            F0[j] += Fc[j] * X00;
            F1[j] += Fc[j] * X01;
        }
    }
    // ---- rotors (q = 0); their G rows are (1, 0), (0, 1) in a differential, the gear / belt products in an explicit pair ----
#pragma unroll
    for (int r = 0; r < 2; r++) {
        T bj, tp[6];
        cptr<T> Rp = P.consts + d.rpre[r];
        const T ga = P.consts[d.gofs + 2 * r], gb = P.consts[d.gofs + 2 * r + 1];
        rotor_terms(P.consts + d.cofs[2 + r], vp, ga * yd0 + gb * yd1, bj, tp);
        u0 -= ga * bj;
        u1 -= gb * bj;
        D00 += Rp[6] * ga * ga;
        D01 += Rp[6] * ga * gb;
        D11 += Rp[6] * gb * gb;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            F0[j] += Rp[j] * ga;
            F1[j] += Rp[j] * gb;
            psi[j] += tp[j];
        }
    }
#if defined(GRBDA_EXP_DIFF_F64D)
    const double idet_d = 1.0 / ((double)D00 * (double)D11 - (double)D01 * (double)D01);
    const double j00 = (double)D11 * idet_d, j01 = -(double)D01 * idet_d, j11 = (double)D00 * idet_d;
    const T i00 = (T)j00, i01 = (T)j01, i11 = (T)j11;
    T blk[14];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        blk[j] = (T)(j00 * (double)F0[j] + j01 * (double)F1[j]);
        blk[6 + j] = (T)(j01 * (double)F0[j] + j11 * (double)F1[j]);
    }
    blk[12] = (T)(j00 * (double)u0 + j01 * (double)u1);
    blk[13] = (T)(j01 * (double)u0 + j11 * (double)u1);
#elif !defined(GRBDA_EXP_DIFF_DET)
    // L L^T = D: l00 = sqrt(D00), l10 = D01 / l00, l11 = sqrt(D11 - l10^2) and triangular solves.  The closed-form inverse
    // (adjugate / determinant, round 2) lost a decimal digit on the differentials: D = G^T Hc G is nearly rank one when the
    // transmission X is large, the determinant cancels, and every entry of D^-1 F^T inherits its error; measured against the oracle
    // compiled in float on 60 000 gated TelloWithArms states (tools/tello_acc2.py, profiles/r4_tello_acc_variants.txt): max error of
    // ydd 1.07e-4 with the determinant, 2.1e-5 with the factorisation, 1.9e-5 for the dense float restatement itself.
    const T r00 = rsqrt_t(D00), l10 = D01 * r00, r11 = rsqrt_t(D11 - l10 * l10);
    M.pivot(D00);
    M.pivot(D11 - l10 * l10);
    const T i00 = r00 * r00 + (l10 * r00 * r11) * (l10 * r00 * r11), i01 = -(l10 * r00) * r11 * r11, i11 = r11 * r11;
    T blk[14];
    auto solve2 = [&](T b0, T b1, T &x0, T &x1) {
        const T y0 = b0 * r00, y1 = (b1 - l10 * y0) * r11;
        x1 = y1 * r11;
        x0 = (y0 - l10 * x1) * r00;
    };
#pragma unroll
    for (int j = 0; j < 6; j++) solve2(F0[j], F1[j], blk[j], blk[6 + j]);
    solve2(u0, u1, blk[12], blk[13]);
#else
    const T idet = rcp_t(D00 * D11 - D01 * D01);
    const T i00 = D11 * idet, i01 = -D01 * idet, i11 = D00 * idet;
    T blk[14];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        blk[j] = i00 * F0[j] + i01 * F1[j];
        blk[6 + j] = i01 * F0[j] + i11 * F1[j];
    }
    blk[12] = i00 * u0 + i01 * u1;
    blk[13] = i01 * u0 + i11 * u1;
#endif
    M.glb_st(d.glb_k, blk);
    if constexpr (OSIM) {  // D^-1 for the force-propagator walk of the contact frames (osim_chain_kernel)
        const T ex[3] = {i00, i01, i11};
        M.glb_st(d.glb_k + 24, ex);
    }
    if (d.lds_acc_out == -1) return;  // on the ground: nobody to hand the projected inertia to
    T acc[27];
    if (!d.acc_first) M.acc_ld(d.lds_acc_out, acc);
    else {
#pragma unroll
        for (int j = 0; j < 27; j++) acc[j] = 0;
    }
#pragma unroll
    for (int r = 0; r < 6; r++) {
        acc[21 + r] += psi[r] + F0[r] * blk[12] + F1[r] * blk[13];
#pragma unroll
        for (int cc = r; cc < 6; cc++) acc[sidx(r, cc)] += IA[sidx(r, cc)] - (F0[r] * blk[cc] + F1[r] * blk[6 + cc]);
    }
    M.acc_st(d.lds_acc_out, acc);
}

template <class T>
__device__ __forceinline__ void diff_acc(const ChainTables<T> &P, const ChainMem<T> &M, const ChainDiff &d)
{
    T blk[24], vp[6], ap[6];
    M.glb_ld(d.glb_k, blk);
    if (d.lds_pva >= 0) {
        T va[12];
        M.lds_ld(d.lds_pva, va);
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
    T ydd0 = blk[12], ydd1 = blk[13];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        ydd0 -= blk[j] * ap[j];
        ydd1 -= blk[6 + j] * ap[j];
    }
    M.put_f(d.v_index, ydd0);
    M.put_f(d.v_index + 1, ydd1);
    if (d.lds_va >= 0) {
        cptr<T> C1 = P.consts + d.cofs[0], C2 = P.consts + d.cofs[1];
        const T yd0 = M.qd(d.v_index), yd1 = M.qd(d.v_index + 1);
        const T qd1 = blk[14] * yd0 + blk[15] * yd1, qd2 = blk[16] * yd0 + blk[17] * yd1;
        const T qdd1 = blk[14] * ydd0 + blk[15] * ydd1 + blk[18], qdd2 = blk[16] * ydd0 + blk[17] * ydd1 + blk[19];
        T E1[9], E2[9], v1[6], v2[6], a1[6], a2[6], c[6];
        const T sc[4] = {blk[20], blk[21], blk[22], blk[23]};
        diff_links(C1, C2, sc, vp, qd1, qd2, E1, E2, v1, v2);
        xmotion(E1, C1 + 9, ap, a1);
        vxz(v1, qd1, c);
#pragma unroll
        for (int j = 0; j < 6; j++) a1[j] += c[j];
        a1[2] += qdd1;
        xmotion(E2, C2 + 9, a1, a2);
        vxz(v2, qd2, c);
#pragma unroll
        for (int j = 0; j < 6; j++) a2[j] += c[j];
        a2[2] += qdd2;
        T out[12];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            out[j] = v2[j];
            out[6 + j] = a2[j];
        }
        M.lds_st(d.lds_va, out);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward run: updateArticulatedBodies + bias back-propagation (ClusterTreeDynamics.cpp:94-129,157-191) along a
// chain, leaf side first.  Per link (n = 1): D = g0^2 h_z (+ rotor), F = X^T h g0 (+ rotor), u = tau - g0 b (- rotor),
// K = F / D, y0 = u / D, and one combined hand-over X^T IA X - F K, X^T (pA + IA c) + F y0 to the parent.
// ---------------------------------------------------------------------------------------------------------------
// Per link the rotor comes in three kinds (a wave-uniform branch on the link record, so one run may mix them): none;
// axisymmetric (q = 0, plan constants); general leaf rotor evaluated at its own angle, whose X^T I X joins the hand-over
// like a second body of the cluster.
// ROT (ChainSeg::rot_kind, plan.cpp): 1 -- every link of the run carries an axisymmetric rotor, 2 -- none carries a rotor: the link body
// is then ONE basic block (no wave-uniform branch on the rotor kind), so the rotor's constants and the parent velocity are fetched with
// the link's own at the top of the iteration and its arithmetic is scheduled between the link's; 0 -- mixed, the branches stay.
template <class T, bool OSIM, bool SVG, int ROT = 0>
__device__ __forceinline__ void run_bwd(const ChainTables<T> &P, const ChainMem<T> &M, const ChainSeg &sg)
{
    T IAc[21], psic[6];  // what the link below handed up (register hand-over inside the run)
    if (sg.head == HEAD_PAIR) {
        const ChainPair pr = load_rec(P.pairs + sg.head_arg);
        pair_bwd<T, OSIM>(P, M, pr, IAc, psic);
    } else if (sg.head == HEAD_SLOT) {
        T acc[27];
        M.acc_ld(sg.head_arg, acc);
#pragma unroll
        for (int j = 0; j < 21; j++) IAc[j] = acc[j];
#pragma unroll
        for (int j = 0; j < 6; j++) psic[j] = acc[21 + j];
    } else {
#pragma unroll
        for (int j = 0; j < 21; j++) IAc[j] = 0;
#pragma unroll
        for (int j = 0; j < 6; j++) psic[j] = 0;
    }
    // the inputs of the next link travel (global slab rows, L2 latency) while the current link is computed
    ChainLink l = load_rec(P.links + sg.first);
    T yd = M.qd(l.v_index), tau_in = M.x(l.v_index), y_in = M.q(l.q_index);
    T blk[8];  // [sin, cos, v 6] of the current link; SVG: in the global slab, fetched one link ahead
    if constexpr (SVG) M.acc_ld(l.lds_sv, blk);
    PREHEADER_DRAIN(2);
    for (int i = 0; i < sg.count; i++) {
        CMARK_DECL(8);
        CMARK(0, IAc[0]);
        const bool more = i + 1 < sg.count;
        const ChainLink ln = load_rec(P.links + (sg.first + (more ? i + 1 : i)));
        T ydn = 0, taun = 0, yn = 0;
        if (more) {
            ydn = M.qd(ln.v_index);
            taun = M.x(ln.v_index);
            yn = M.q(ln.q_index);
        }
        T blkn[8];
        if constexpr (SVG) {
            if (more) M.acc_ld(ln.lds_sv, blkn);
        } else {
            M.lds_ld(l.lds_sv, blk);
        }
        cptr<T> C = P.consts + l.cofs;
        cptr<T> Ic = C + 12;
        cptr<T> Ib = P.consts + l.iofs;
        const T g0 = C[kBodyConstFixed];
        const T qdi = g0 * yd;
        T v[6];
#pragma unroll
        for (int j = 0; j < 6; j++) v[j] = blk[2 + j];
        T chat[6];
        vxz(v, qdi, chat);
        CMARK(1, chat[0]);

        T IA[21], psi[6];
        bias_force(Ic, v, psi);
#pragma unroll
        for (int j = 0; j < 21; j++) IA[j] = Ib[j] + IAc[j];
#pragma unroll
        for (int j = 0; j < 6; j++) psi[j] += psic[j];
        T h[6];
#pragma unroll
        for (int k = 0; k < 6; k++) h[k] = IA[sidx(k, 2)];
        const T bj = psi[2] + h[0] * chat[0] + h[1] * chat[1] + h[3] * chat[3] + h[4] * chat[4];
        T u = tau_in - g0 * bj;
        T D = h[2] * g0 * g0;
        CMARK(2, u);
        T F[6];
        {
            T t[6];
            symv_z(IA, chat, t);
#pragma unroll
            for (int j = 0; j < 6; j++) t[j] += psi[j];
            link_up(perm_if<T>(GRBDA_PERM_LINK, l.perm), blk[0], blk[1], C, h, t, IA, F, psic, IAc);
        }
#pragma unroll
        for (int r = 0; r < 6; r++) F[r] *= g0;
        CMARK(3, F[0]);
        if (ROT == 0 && l.rofs >= 0 && l.rpre < 0) {
            cptr<T> Cr = P.consts + l.rofs;
            cptr<T> Ir = Cr + 12;
            const T gr = Cr[kBodyConstFixed];
            const T qdr = gr * yd;
            T vp[6];
            if (l.lds_pv != -1) {
                if constexpr (SVG) M.acc_ld(l.lds_pv, vp);
                else M.lds_ld(l.lds_pv, vp);
            } else {
#pragma unroll
                for (int j = 0; j < 6; j++) vp[j] = 0;
            }
            T sr, cr_, vr[6], crv[6], prr[6], hr[6];
            sincos_t(gr * y_in, &sr, &cr_);
            link_down(perm_if<T>(GRBDA_PERM_ROTOR, l.rperm), sr, cr_, Cr, vp, vr);
            vr[2] += qdr;
            vxz(vr, qdr, crv);
            bias_force(Ir, vr, prr);
#pragma unroll
            for (int k = 0; k < 6; k++) hr[k] = Ir[sidx(k, 2)];
            const T bjr = prr[2] + hr[0] * crv[0] + hr[1] * crv[1] + hr[3] * crv[3] + hr[4] * crv[4];
            u -= gr * bjr;
            D += hr[2] * gr * gr;
            T fr[6], t[6], tp[6], Br[21];
            symv_z(Ir, crv, t);
#pragma unroll
            for (int j = 0; j < 6; j++) t[j] += prr[j];
            link_up(perm_if<T>(GRBDA_PERM_ROTOR, l.rperm), sr, cr_, Cr, hr, t, Ir, fr, tp, Br);
#pragma unroll
            for (int r = 0; r < 6; r++) F[r] += fr[r] * gr;
#pragma unroll
            for (int j = 0; j < 6; j++) psic[j] += tp[j];
#pragma unroll
            for (int j = 0; j < 21; j++) IAc[j] += Br[j];
        }
        if (ROT == 1 || (ROT == 0 && l.rofs >= 0 && l.rpre >= 0)) {
            cptr<T> Cr = P.consts + l.rofs;
            cptr<T> Rp = P.consts + l.rpre;  // [X0^T h (6)][h_z]
            const T gr = Cr[kBodyConstFixed];
            T vp[6];
            if constexpr (ROT == 1 && !SVG) {  // branch-free: a run that starts on the ground reads row 0 and discards it
                const bool gnd = l.lds_pv == -1;
                M.lds_ld(gnd ? 0 : l.lds_pv, vp);
#pragma unroll
                for (int j = 0; j < 6; j++) vp[j] = gnd ? T(0) : vp[j];
            } else if (l.lds_pv != -1) {
                if constexpr (SVG) M.acc_ld(l.lds_pv, vp);
                else M.lds_ld(l.lds_pv, vp);
            } else {
#pragma unroll
                for (int j = 0; j < 6; j++) vp[j] = 0;
            }
            T bjr, tp[6];
            rotor_terms(Cr, vp, gr * yd, bjr, tp);
            u -= gr * bjr;
            D += Rp[6] * gr * gr;
#pragma unroll
            for (int r = 0; r < 6; r++) F[r] += Rp[r] * gr;
#pragma unroll
            for (int j = 0; j < 6; j++) psic[j] += tp[j];
        }
        CMARK(4, psic[0]);
        const T Dinv = rcp_t(D);
        M.pivot(D);
        T kb[7];  // [K 6][y0]: what the acceleration run needs (it recomputes sin / cos from q: two slab rows less per link)
#pragma unroll
        for (int r = 0; r < 6; r++) kb[r] = F[r] * Dinv;
        kb[6] = u * Dinv;
        M.glb_st(l.glb_k, kb);
        if constexpr (OSIM) {  // the force-propagator walk (osim_chain_kernel) reads [sin][cos][1 / D] behind the block
            const T ex[3] = {blk[0], blk[1], Dinv};
            M.glb_st(l.glb_k + 7, ex);
        }
        CMARK(5, kb[6]);
#pragma unroll
        for (int r = 0; r < 6; r++) {
            psic[r] += F[r] * kb[6];
#pragma unroll
            for (int cc = r; cc < 6; cc++) IAc[sidx(r, cc)] -= F[r] * kb[cc];
        }
        CMARK(6, IAc[0]);
        l = ln;
        yd = ydn;
        tau_in = taun;
        y_in = yn;
        if constexpr (SVG) {
            if (more) {
#pragma unroll
                for (int j = 0; j < 8; j++) blk[j] = blkn[j];
            }
        }
        CMARK(7, yd);
        CMARK_SUM(8, 8);
    }
    // hand the chain's projected inertia / bias to the body it hangs off
    if (sg.lds_acc_out != -1) {
        T acc[27];
        if (sg.acc_first) {
#pragma unroll
            for (int j = 0; j < 21; j++) acc[j] = IAc[j];
#pragma unroll
            for (int j = 0; j < 6; j++) acc[21 + j] = psic[j];
        } else {
            M.acc_ld(sg.lds_acc_out, acc);
#pragma unroll
            for (int j = 0; j < 21; j++) acc[j] += IAc[j];
#pragma unroll
            for (int j = 0; j < 6; j++) acc[21 + j] += psic[j];
        }
        M.acc_st(sg.lds_acc_out, acc);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// acceleration run (ClusterTreeDynamics.cpp:131-152), root side first: ydd = y0 - K a_p; a = X a_p + c + z g0 ydd
// ---------------------------------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ void run_acc(const ChainTables<T> &P, const ChainMem<T> &M, const ChainSeg &sg)
{
    T vp[6], ap[6];
    if (sg.lds_pva >= 0) {
        T va[12];
        M.lds_ld(sg.lds_pva, va);
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
    ChainLink l = load_rec(P.links + sg.first);
    T kb[7];
    M.glb_ld(l.glb_k, kb);
    T yd = M.qd(l.v_index), yq = l.has_child ? M.q(l.q_index) : T(0);
    PREHEADER_DRAIN(4);
    for (int i = 0; i < sg.count; i++) {
        // the next link's record, [K | y0] block and inputs travel while this link is computed (the joint angle only for
        // links with children: the others need no transform here)
        CMARK_DECL(5);
        CMARK(0, ap[0]);
        const bool more = i + 1 < sg.count;
        const ChainLink ln = load_rec(P.links + (sg.first + (more ? i + 1 : i)));
        T kn[7], ydn = 0, yqn = 0;
        if (more) {
            M.glb_ld(ln.glb_k, kn);
            ydn = M.qd(ln.v_index);
            if (ln.has_child) yqn = M.q(ln.q_index);
        }
        T ydd = kb[6];
#pragma unroll
        for (int r = 0; r < 6; r++) ydd -= kb[r] * ap[r];
        CMARK(1, ydd);
        M.put_f(l.v_index, ydd);
        CMARK(2, ydd);
        if (l.has_child) {
            cptr<T> C = P.consts + l.cofs;
            const T g0 = C[kBodyConstFixed];
            const T qdi = g0 * yd;
            T v[6], a[6], chat[6], sn, cs;
            sincos_t(g0 * yq, &sn, &cs);
            link_down2(perm_if<T>(GRBDA_PERM_ACC, l.perm), sn, cs, C, vp, ap, v, a);
            v[2] += qdi;
            vxz(v, qdi, chat);
#pragma unroll
            for (int j = 0; j < 6; j++) a[j] += chat[j];
            a[2] += g0 * ydd;
            if (l.lds_va >= 0) {
                T va[12];
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    va[j] = v[j];
                    va[6 + j] = a[j];
                }
                M.lds_st(l.lds_va, va);
            }
#pragma unroll
            for (int j = 0; j < 6; j++) {
                vp[j] = v[j];
                ap[j] = a[j];
            }
        }
        CMARK(3, ap[0]);
        if (more) {
            l = ln;
            yd = ydn;
            yq = yqn;
#pragma unroll
            for (int j = 0; j < 7; j++) kb[j] = kn[j];
        }
        CMARK(4, kb[0]);
        CMARK_SUM(20, 5);
    }
}

// The acceleration run with the [K | y0] blocks and input rows of FOUR links requested at once (see run_fwd_c): the blocks were
// written by the backward sweep tens of microseconds earlier and come back from the Infinity Cache (~1 300 cycles measured at the loop
// top of run_acc, where the block of the NEXT link was requested and -- the compiler's wait-count pass merges the loop's entry states --
// waited for at once); a link of this run is 20-150 instructions.
template <class T>
__device__ __forceinline__ void run_acc_c(const ChainTables<T> &P, const ChainMem<T> &M, const ChainSeg &sg)
{
    // (two phases per chunk as in run_fwd_c: the recursion a_i = X_i a_(i-1) + c_i + z g0 ydd_i with ydd_i = y0_i - K_i a_(i-1) runs on
    // register operands prepared for all four links beforehand)
    constexpr bool TWO_PHASE = ((GRBDA_CHUNK_MASK) & 8) != 0;
    T vp[6], ap[6];
    for (int base = 0; base < sg.count; base += kChunk) {
        const int n = sg.count - base;
        CMARK_DECL(5);
        CMARK(0, ap[0]);
        // (one basic block, see run_fwd_c; the input rows are requested BEFORE the [K | y0] blocks, so that phase A -- which needs only
        // them -- runs while the blocks are still on their way from the Infinity Cache)
        ChainLink L[kChunk];
        T kb[kChunk][7], yd[kChunk], yq[kChunk];
#pragma unroll
        for (int u = 0; u < kChunk; u++) L[u] = load_rec(P.links + (sg.first + base + (u < n ? u : n - 1)));
#pragma unroll
        for (int u = 0; u < kChunk; u++) {
            yd[u] = M.qd(L[u].v_index);
            yq[u] = M.q(L[u].q_index);
        }
#pragma unroll
        for (int u = 0; u < kChunk; u++) M.glb_ld(L[u].glb_k, kb[u]);
        if (base == 0) {
            if (sg.lds_pva >= 0) {
                T va[12];
                M.lds_ld(sg.lds_pva, va);
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
        }
        CMARK(1, ap[0]);
        CMARK(2, ap[0]);
        T E[kChunk][9], r[kChunk][3], g0v[kChunk], qdi[kChunk];
        if constexpr (TWO_PHASE) {
#pragma unroll
            for (int u = 0; u < kChunk; u++) {
                if (u < n && L[u].has_child) {
                    cptr<T> C = P.consts + L[u].cofs;
                    const T g0 = C[kBodyConstFixed];
                    T sn, cs;
                    sincos_t(g0 * yq[u], &sn, &cs);
                    rotate_z(sn, cs, C, E[u]);
#pragma unroll
                    for (int j = 0; j < 3; j++) r[u][j] = C[9 + j];
                    g0v[u] = g0;
                    qdi[u] = g0 * yd[u];
                }
            }
        }
        CMARK(3, ap[1]);
#pragma unroll
        for (int u = 0; u < kChunk; u++) {
            if (u < n) {
                const ChainLink &l = L[u];
                T ydd = kb[u][6];
#pragma unroll
                for (int rr = 0; rr < 6; rr++) ydd -= kb[u][rr] * ap[rr];
                M.put_f(l.v_index, ydd);
                if (l.has_child) {
                    T v[6], a[6], chat[6];
                    if constexpr (TWO_PHASE) {
                        xmotion(E[u], r[u], vp, v);
                        xmotion(E[u], r[u], ap, a);
                        v[2] += qdi[u];
                        vxz(v, qdi[u], chat);
#pragma unroll
                        for (int j = 0; j < 6; j++) a[j] += chat[j];
                        a[2] += g0v[u] * ydd;
                    } else {
                        cptr<T> C = P.consts + l.cofs;
                        const T g0 = C[kBodyConstFixed];
                        const T qd1 = g0 * yd[u];
                        T sn, cs;
                        sincos_t(g0 * yq[u], &sn, &cs);
                        link_down2(perm_if<T>(GRBDA_PERM_ACC, l.perm), sn, cs, C, vp, ap, v, a);
                        v[2] += qd1;
                        vxz(v, qd1, chat);
#pragma unroll
                        for (int j = 0; j < 6; j++) a[j] += chat[j];
                        a[2] += g0 * ydd;
                    }
                    if (l.lds_va >= 0) {
                        T va[12];
#pragma unroll
                        for (int j = 0; j < 6; j++) {
                            va[j] = v[j];
                            va[6 + j] = a[j];
                        }
                        M.lds_st(l.lds_va, va);
                    }
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        vp[j] = v[j];
                        ap[j] = a[j];
                    }
                }
            }
        }
        CMARK(4, ap[0]);
        CMARK_SUM(20, 5);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// floating base (FreeJoint.cpp:10-46): S = 1, D = IA, c = 0
// ---------------------------------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ void free_fwd(const ChainTables<T> &P, const ChainMem<T> &M, const ChainFree &f)
{
    if (f.lds_v < 0) return;
    T v[6];
#pragma unroll
    for (int j = 0; j < 6; j++) v[j] = M.qd(f.v_index + j);
    M.lds_st(f.lds_v, v);
}

template <class T>
__device__ __forceinline__ void free_acc_inputs(const ChainTables<T> &P, const ChainMem<T> &M, const ChainFree &f, T (&o)[4], T (&r)[3]);
template <class T>
__device__ __forceinline__ void free_acc_core(const ChainTables<T> &P, const ChainMem<T> &M, const ChainFree &f, const T (&y0)[6],
                                              const T (&o)[4], const T (&r)[3], const T (&vb)[6]);
template <class T, bool OSIM>
__device__ __forceinline__ void free_bwd(const ChainTables<T> &P, const ChainMem<T> &M, const ChainFree &f, bool fuse_acc = false)
{
    cptr<T> Ic = P.consts + f.cofs + 12;
    cptr<T> Ib = P.consts + f.iofs;
    T v[6], psi[6], IA[21];
#pragma unroll
    for (int j = 0; j < 6; j++) v[j] = M.qd(f.v_index + j);
    T o_[4] = {0, 0, 0, 0}, r_[3] = {0, 0, 0};
    if constexpr (!OSIM) {
        if (fuse_acc) free_acc_inputs(P, M, f, o_, r_);
    }
    bias_force(Ic, v, psi);
    if (f.lds_acc != -1) {
        T acc[27];
        M.acc_ld(f.lds_acc, acc);
#pragma unroll
        for (int j = 0; j < 21; j++) IA[j] = Ib[j] + acc[j];
#pragma unroll
        for (int j = 0; j < 6; j++) psi[j] += acc[21 + j];
        if (f.lds_acc2 != -1) {  // latency-mode programs: what the other wavefronts' limbs handed up (dealt out in order: 2, 3, 4)
            auto add = [&](int slot) {
                M.acc_ld(slot, acc);
#pragma unroll
                for (int j = 0; j < 21; j++) IA[j] += acc[j];
#pragma unroll
                for (int j = 0; j < 6; j++) psi[j] += acc[21 + j];
            };
            add(f.lds_acc2);
            if (f.lds_acc3 != -1) {
                add(f.lds_acc3);
                if (f.lds_acc4 != -1) add(f.lds_acc4);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 21; j++) IA[j] = Ib[j];
    }
    T D[6][6], u[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        u[i] = M.x(f.v_index + i) - psi[i];
#pragma unroll
        for (int j = 0; j < 6; j++) D[i][j] = IA[sidx(i, j)];
    }
    Chol<T, 6> ch;
    ch.factor(D);
#pragma unroll
    for (int i = 0; i < 6; i++) M.pivot(ch.inv[i] < T(1e30) ? ch.inv[i] : T(0));  // (1 / sqrt(pivot): Inf for 0, NaN below it)
    ch.solve(u);
    if constexpr (!OSIM) {
        if (fuse_acc) {  // (every LDS read of this segment is behind us: the rows free_acc writes may alias the accumulators)
            free_acc_core(P, M, f, u, o_, r_, v);
            return;
        }
    }
    M.glb_st(f.glb_y0, u);
    if constexpr (OSIM) {  // Cholesky factor of the base's articulated inertia: [L lower triangle 21][1 / diag 6]
        T ex[27];
#pragma unroll
        for (int i = 0; i < 6; i++) {
#pragma unroll
            for (int j = 0; j <= i; j++) ex[i * (i + 1) / 2 + j] = ch.L[i][j];
            ex[21 + i] = ch.inv[i];
        }
        M.glb_st(f.glb_y0 + 6, ex);
    }
}

// (y0 in registers: the tail of free_bwd when the two segments are fused, ChainDev::fuse; the base's seven positions are requested by
// free_acc_inputs -- in the fused segment at its very start, next to the backward segment's own rows: one wait instead of two)
template <class T>
__device__ __forceinline__ void free_acc_inputs(const ChainTables<T> &P, const ChainMem<T> &M, const ChainFree &f, T (&o)[4], T (&r)[3])
{
    const int nori = P.ori_repr == 0 ? 4 : 3;
#pragma unroll
    for (int j = 0; j < 4; j++) o[j] = j < nori ? M.q(f.q_index + 3 + j) : T(0);
#pragma unroll
    for (int j = 0; j < 3; j++) r[j] = M.q(f.q_index + j);
}
template <class T>
__device__ __forceinline__ void free_acc_core(const ChainTables<T> &P, const ChainMem<T> &M, const ChainFree &f, const T (&y0)[6],
                                              const T (&o)[4], const T (&r)[3], const T (&vb)[6])
{
    T E[9], g[6], ag[6];
    free_rotation(P.ori_repr, o, E);
#pragma unroll
    for (int j = 0; j < 6; j++) g[j] = P.a_root[j];
    xmotion(E, r, g, ag);
    // ydd = D^-1 u - D^-1 U^T a' with U = D = IA  =>  ydd = y0 - a' ;  a = a' + ydd = y0
#pragma unroll
    for (int j = 0; j < 6; j++) M.put_f(f.v_index + j, y0[j] - ag[j]);
    if (f.lds_va >= 0) {
        T va[12];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            va[j] = vb[j];
            va[6 + j] = y0[j];
        }
        M.lds_st(f.lds_va, va);
    }
}
template <class T>
__device__ __forceinline__ void free_acc(const ChainTables<T> &P, const ChainMem<T> &M, const ChainFree &f)
{
    T y0[6], o[4], r[3], vb[6];
    M.glb_ld(f.glb_y0, y0);
    free_acc_inputs(P, M, f, o, r);
#pragma unroll
    for (int j = 0; j < 6; j++) vb[j] = M.qd(f.v_index + j);
    free_acc_core(P, M, f, y0, o, r, vb);
}

#include "gen_segments.h"

#if GRBDA_CHAIN_UNIT == 2
// ---------------------------------------------------------------------------------------------------------------
// Single-cluster programs (ChainProgram::single_gen): ONE generic cluster on the ground and nothing else -- the URDF+ loop
// mechanisms of BASELINE config 5 (four_bar.urdf, six_bar.urdf), a triple cluster on a bench.  The whole forward dynamics is the
// cluster's downward pass, its upward pass and ydd = y0 - K a_root, so the kernel is specialised on (n, implicit?) at compile time
// and touches no slab at all; results go straight to the caller's array.
// ---------------------------------------------------------------------------------------------------------------
// The tile is bound by the latency of its dependent LDS round trips and by instruction issue, so what counts is how many wavefronts
// the LDS of a CU holds: the inputs of the NEXT tile travel into registers while this one computes (every lane loads its own row:
// the tile's rows are one contiguous block, every byte of every cache line is used) and are written to ONE set of LDS rows
// [column][lane] at the top of their tile -- half the staging area of a double-buffered asynchronous copy, which is a wavefront
// more per CU on six_bar (devplan.h, lds_workgroups_per_cu).
template <class T>
struct ChainMemC : ChainMem<T> {
    int in_q, in_qd, in_x;  // first LDS rows of the staged blocks ([column][lane])
    T *out_g;               // this lane's result row (nullptr: a lane beyond the batch)
    __device__ __forceinline__ T q(int j) const { return reinterpret_cast<const T *>(grbda_smem)[(in_q + j) * kWave + this->lane]; }
    __device__ __forceinline__ T qd(int j) const { return reinterpret_cast<const T *>(grbda_smem)[(in_qd + j) * kWave + this->lane]; }
    __device__ __forceinline__ T x(int j) const { return reinterpret_cast<const T *>(grbda_smem)[(in_x + j) * kWave + this->lane]; }
    __device__ __forceinline__ void put(int j, T v) const
    {
        if (out_g) out_g[j] = v;
    }
    __device__ __forceinline__ void put_f(int j, T v) const { put(j, v); }
};

// the tile loop of the single-cluster kernels: BODY(M) is the tile's work (forward or inverse dynamics of the cluster)
template <class T, int N, bool LOOP, class BODY>
__device__ __forceinline__ void gen1_tiles(int work_bytes, int nq, const T *__restrict__ q, const T *__restrict__ qd, const T *__restrict__ x,
                                           T *__restrict__ out, size_t B, unsigned long long *bad_count, BODY body)
{
    const int lane = threadIdx.x;
    ChainMemC<T> M;
    M.bad = 0;
    M.lane = lane;
    M.lane_b = (unsigned)lane * (unsigned)sizeof(T);
    M.gmul = 1;
    M.amask = ~0;
    M.glb_u = nullptr;
    M.in_q_u = M.in_qd_u = M.in_x_u = nullptr;
    M.out_u = nullptr;
    M.set_slab(nullptr, 0, 0, 0);  // (the single-cluster kernels touch no slab)
    M.out_row = -1;
    M.out_f = nullptr;
    // LDS: [work area][q rows | qd rows | tau / ydd rows]   (nv = N: the cluster is the whole model)
    M.in_q = work_bytes / (int)(kWave * sizeof(T));
    M.in_qd = M.in_q + nq;
    M.in_x = M.in_qd + N;
    // implicit clusters: spanning positions, one per body; k = n + rows with at most three constraint rows (capi.cpp checks nq)
    constexpr int NQ = LOOP ? (N + 3 < kMaxClusterBodies ? N + 3 : kMaxClusterBodies) : N;
    T rq[NQ], rv[N], rx[N];
    const size_t n_tiles = (B + kWave - 1) / kWave;
    auto fetch = [&](size_t tile) {
        size_t row = tile * kWave + lane;
        if (row >= B) row = B - 1;  // (lanes beyond the batch work on the last state; nothing of theirs is stored)
#pragma unroll
        for (int j = 0; j < NQ; j++)
            if (j < nq) rq[j] = q[row * (size_t)nq + j];
#pragma unroll
        for (int a = 0; a < N; a++) {
            rv[a] = qd[row * (size_t)N + a];
            rx[a] = x[row * (size_t)N + a];
        }
    };
#pragma unroll
    for (int j = 0; j < NQ; j++) rq[j] = 0;
    if ((size_t)blockIdx.x < n_tiles) fetch(blockIdx.x);
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t left = B - tile * kWave;
        const int rows_valid = left < (size_t)kWave ? (int)left : kWave;
#pragma unroll
        for (int j = 0; j < NQ; j++)
            if (j < nq) reinterpret_cast<T *>(grbda_smem)[(M.in_q + j) * kWave + lane] = rq[j];
#pragma unroll
        for (int a = 0; a < N; a++) {
            reinterpret_cast<T *>(grbda_smem)[(M.in_qd + a) * kWave + lane] = rv[a];
            reinterpret_cast<T *>(grbda_smem)[(M.in_x + a) * kWave + lane] = rx[a];
        }
        if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x);
        M.out_g = lane < rows_valid ? out + (tile * kWave + lane) * (size_t)N : nullptr;
        body(M);
        M.flush_bad(bad_count, rows_valid);
    }
}

template <class T, int N, bool LOOP, int WPS>
__global__ __launch_bounds__(kWave, WPS) void aba_gen1_kernel(ChainDev<T> DP, const T *__restrict__ q, const T *__restrict__ qd,
                                                            const T *__restrict__ tau, T *__restrict__ ydd, size_t B)
{
    ChainTables<T> P;
    P.segs = nullptr;
    P.links = nullptr;
    P.pairs = nullptr;
    P.frees = nullptr;
    P.diffs = nullptr;
    P.gens = (cptr<ChainGen>)DP.gens;
    P.gbodies = (cptr<ChainGenBody>)DP.gbodies;
    P.cints = (cptr<int32_t>)DP.cints;
    P.consts = (cptr<T>)DP.consts;
    P.n_segs = 0;
    P.nq = DP.nq;
    P.nv = DP.nv;
    P.ori_repr = DP.ori_repr;
#pragma unroll
    for (int i = 0; i < 6; i++) P.a_root[i] = DP.a_root[i];
    const ChainGen g = load_rec(P.gens);
    gen1_tiles<T, N, LOOP>(DP.lds_bytes, P.nq, q, qd, tau, ydd, B, DP.bad_count, [&](const ChainMemC<T> &M) {
        gen_down<T, N, LOOP>(P, M, g, g.lds_w, true, false);
        gen_up<T, N, LOOP, true>(P, M, g);
    });
}

template <class T, int N, bool LOOP>
static hipError_t launch_gen1(const ChainDev<T> &P, const T *q, const T *qd, const T *tau, T *ydd, size_t B, int grid, size_t lds_bytes,
                              hipStream_t stream)
{
    // wavefronts per SIMD the register footprint allows (checked against the code object: no scratch)
    constexpr int WPS = sizeof(T) == 4 ? (N <= 2 ? 3 : 2) : (N == 1 ? 2 : 1);
    hipLaunchKernelGGL((aba_gen1_kernel<T, N, LOOP, WPS>), dim3(grid), dim3(kWave), lds_bytes, stream, P, q, qd, tau, ydd, B);
    return hipGetLastError();
}
template <class T>
int gen1_waves_per_simd(int n)
{
    return sizeof(T) == 4 ? (n <= 2 ? 3 : 2) : (n == 1 ? 2 : 1);
}
template int gen1_waves_per_simd<float>(int);
template int gen1_waves_per_simd<double>(int);
template <class T>
hipError_t launch_aba_gen1(const ChainDev<T> &P, int n, int implicit, const T *q, const T *qd, const T *tau, T *ydd, size_t B, int grid,
                           size_t lds_bytes, hipStream_t stream)
{
    if (implicit) {
        if (n == 1) return launch_gen1<T, 1, true>(P, q, qd, tau, ydd, B, grid, lds_bytes, stream);
        if (n == 2) return launch_gen1<T, 2, true>(P, q, qd, tau, ydd, B, grid, lds_bytes, stream);
        if (n == 3) return launch_gen1<T, 3, true>(P, q, qd, tau, ydd, B, grid, lds_bytes, stream);
        return launch_gen1<T, 4, true>(P, q, qd, tau, ydd, B, grid, lds_bytes, stream);
    }
    if (n == 1) return launch_gen1<T, 1, false>(P, q, qd, tau, ydd, B, grid, lds_bytes, stream);
    if (n == 2) return launch_gen1<T, 2, false>(P, q, qd, tau, ydd, B, grid, lds_bytes, stream);
    if (n == 3) return launch_gen1<T, 3, false>(P, q, qd, tau, ydd, B, grid, lds_bytes, stream);
    return launch_gen1<T, 4, false>(P, q, qd, tau, ydd, B, grid, lds_bytes, stream);
}
template hipError_t launch_aba_gen1<float>(const ChainDev<float> &, int, int, const float *, const float *, const float *, float *, size_t, int,
                                           size_t, hipStream_t);
template hipError_t launch_aba_gen1<double>(const ChainDev<double> &, int, int, const double *, const double *, const double *, double *,
                                            size_t, int, size_t, hipStream_t);
#endif

// ---------------------------------------------------------------------------------------------------------------
// MODE 1: the program has differential clusters (ChainDiff).  A kernel variant of its own, so that the models without them
// (every URDF robot of the reference) keep the register allocation of the plain run / pair / free code.  MODE 2: the program
// has generic clusters (ChainGen; gen_segments.h) and possibly differentials: translation unit 2.
template <class T, int WPS, int MODE>
__global__ __launch_bounds__(kWave, WPS) void aba_chain_kernel(ChainDev<T> DP, const T *__restrict__ q, const T *__restrict__ qd,
                                                             const T *__restrict__ tau, T *__restrict__ ydd, size_t B,
                                                             T *__restrict__ scratch)
{
    constexpr bool DIFF = MODE >= 1, GEN = MODE == 2;
    ChainTables<T> P;
    P.segs = (cptr<ChainSeg>)DP.segs;
    P.links = (cptr<ChainLink>)DP.links;
    P.pairs = (cptr<ChainPair>)DP.pairs;
    P.frees = (cptr<ChainFree>)DP.frees;
    P.diffs = DIFF ? (cptr<ChainDiff>)DP.diffs : nullptr;
    P.gens = GEN ? (cptr<ChainGen>)DP.gens : nullptr;
    P.gbodies = GEN ? (cptr<ChainGenBody>)DP.gbodies : nullptr;
    P.cints = DIFF ? (cptr<int32_t>)DP.cints : nullptr;
    P.consts = (cptr<T>)DP.consts;
    P.n_segs = DP.n_segs;
    P.nq = DP.nq;
    P.nv = DP.nv;
    P.ori_repr = DP.ori_repr;
#pragma unroll
    for (int i = 0; i < 6; i++) P.a_root[i] = DP.a_root[i];
    const int lane = threadIdx.x;
    // wave slab: [nq + 2 nv input rows][n_glb_slots rows], 64 scalars per row
    T *slab = scratch + (size_t)blockIdx.x * (size_t)(DP.n_glb_slots + P.nq + 2 * P.nv) * kWave;
    ChainMem<T> M;
    M.lane = lane;
    M.lane_b = (unsigned)lane * (unsigned)sizeof(T);
#ifdef GRBDA_EXP  // ablation builds only (make variant VFLAGS=-DGRBDA_EXP): the product kernels carry no wrong-result switches
    const int dbg = DP.debug;
#else
    constexpr int dbg = 0;
#endif
    M.gmul = (dbg & 8) ? 0 : 1;
    M.amask = (dbg & 16) ? kSlotGlobal : ~0;
    M.glb_u = slab + (size_t)(P.nq + 2 * P.nv) * kWave;
    M.in_q_u = slab;
    M.in_qd_u = slab + (size_t)P.nq * kWave;
    M.in_x_u = slab + (size_t)(P.nq + P.nv) * kWave;
    M.out_u = slab + (size_t)(P.nq + P.nv) * kWave;
    M.set_slab(slab, DP.n_glb_slots + P.nq + 2 * P.nv, P.nq, P.nv);
    M.set_out(DP.out_lds);
    M.bad = 0;
#ifdef GRBDA_CHAIN_PROFILE
    M.pacc = 0;
#endif

    const size_t n_tiles = (B + kWave - 1) / kWave;
    CPROF_T0();
#ifdef GRBDA_EXP
    // experiment: the wavefronts in the odd slots of their SIMD start late by (debug >> 8) x 127 x 64 clocks, so that the two wavefronts of a
    // SIMD do not stage their inputs (an HBM burst of the whole grid) and run their latency-bound phases at the same moment
    if ((dbg >> 8) && (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1)) {  // HW_REG_HW_ID (4), offset 0, size 4: WAVE_ID
        for (int i = 0; i < (dbg >> 8); i++) __builtin_amdgcn_s_sleep(127);
    }
#endif
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t left = B - tile * kWave;
        const int rows_valid = left < (size_t)kWave ? (int)left : kWave;
        // (DP.debug: profiling aid of tools/chain_ablate.py -- bit 0 skips the prologue, bit 1 the segments, bit 2 the
        // epilogue; results are then meaningless)
        if (!(dbg & 1)) stage_inputs(q, qd, tau, tile, rows_valid, P.nq, P.nv, slab, lane, DP.lds_bytes);
        if (DP.fuse & 1) {  // the floating base's velocity: from the lane's row of the staged block (still there) to its LDS slot
            const T *mine = reinterpret_cast<const T *>(grbda_smem) + (kWave * P.nq + lane * P.nv + DP.stage_v_index);
            T v[6];
#pragma unroll
            for (int j = 0; j < 6; j++) v[j] = mine[j];
            wave_lds_fence();
            M.lds_st(DP.stage_lds_v, v);
        }
        CPROF_ADD(0, 1);
        for (int s = 0; s < ((dbg & 2) ? 0 : P.n_segs); s++) {
            const ChainSeg sg = load_rec(P.segs + s);
            // (ablation builds: bits 5 / 6 / 7 skip the forward / backward / acceleration segments of every kind)
            if ((dbg & 32) && (sg.op == SEG_RUN_FWD || sg.op == SEG_FREE_FWD)) continue;
            if ((dbg & 64) && (sg.op == SEG_RUN_BWD || sg.op == SEG_FREE_BWD)) continue;
            if ((dbg & 128) && (sg.op == SEG_RUN_ACC || sg.op == SEG_FREE_ACC || sg.op == SEG_PAIR_ACC)) continue;
            switch (sg.op) {
                case SEG_RUN_FWD:
                    if (DP.sv_global) run_fwd<T, true>(P, M, sg);
                    else if constexpr ((GRBDA_CHUNK_MASK) & 1) run_fwd_c(P, M, sg);
                    else run_fwd<T, false>(P, M, sg);
                    break;
                case SEG_RUN_BWD: {
                    if (DP.sv_global) run_bwd<T, false, true>(P, M, sg);
                    else if (((GRBDA_CHUNK_MASK) & 2) && sg.rot_kind == 1) run_bwd<T, false, false, 1>(P, M, sg);
                    else run_bwd<T, false, false>(P, M, sg);
                    break;
                }
                case SEG_RUN_ACC:
                    if constexpr ((GRBDA_CHUNK_MASK) & 4) run_acc_c(P, M, sg);
                    else run_acc(P, M, sg);
                    break;
                case SEG_PAIR_ACC: pair_acc(P, M, load_rec(P.pairs + sg.first)); break;
                case SEG_FREE_FWD:
                    if (!((DP.fuse & 1) && s == 0)) free_fwd(P, M, load_rec(P.frees + sg.first));
                    break;
                case SEG_FREE_BWD: free_bwd<T, false>(P, M, load_rec(P.frees + sg.first), (DP.fuse & 2) != 0); break;
                case SEG_DIFF_FWD:
                    if constexpr (DIFF) diff_fwd(P, M, load_rec(P.diffs + sg.first));
                    break;
                case SEG_DIFF_BWD:
                    if constexpr (DIFF) diff_bwd<T, false>(P, M, load_rec(P.diffs + sg.first));
                    break;
                case SEG_DIFF_ACC:
                    if constexpr (DIFF) diff_acc(P, M, load_rec(P.diffs + sg.first));
                    break;
                case SEG_GEN_FWD:
                    if constexpr (GEN) gen_segment<T, 0>(P, M, load_rec(P.gens + sg.first));
                    break;
                case SEG_GEN_BWD:
                    if constexpr (GEN) gen_segment<T, 1>(P, M, load_rec(P.gens + sg.first));
                    break;
                case SEG_GEN_ACC:
                    if constexpr (GEN) gen_segment<T, 2>(P, M, load_rec(P.gens + sg.first));
                    break;
                default:
                    if (!(DP.fuse & 2)) free_acc(P, M, load_rec(P.frees + sg.first));
                    break;
            }
            CPROF_ADD((sg.op == SEG_RUN_BWD && sg.head == HEAD_PAIR) ? 20 : 2 + sg.op,
                      sg.op == SEG_RUN_FWD || sg.op == SEG_RUN_BWD || sg.op == SEG_RUN_ACC ? sg.count : 1);
        }
        if (!(dbg & 4)) {
            if (M.out_row >= 0) write_outputs_lds<T>(M.out_row, ydd, tile, rows_valid, P.nv, lane);
            else write_outputs(slab + (size_t)(P.nq + P.nv) * kWave, ydd, tile, rows_valid, P.nv, lane);
        }
        M.flush_bad(DP.bad_count, rows_valid);
        CPROF_ADD(1, 1);
    }
    CPROF_END();
}


// ---------------------------------------------------------------------------------------------------------------
// Latency mode (plan.h, ChainProgram::n_waves = 2): a tile of 64 states is run by a WORKGROUP of two wavefronts.  The limbs
// below the floating base are dealt to the two wavefronts (ChainSeg::owner), the base's own segments run on wavefront 0,
// SEG_BARRIER segments order the hand-overs: base velocity (LDS) -> limbs, limb accumulators (global slab) -> base, base
// acceleration (LDS) -> limbs.  The tile prologue is shared too: wavefront 0 stages q, wavefront 1 stages qd and tau.
// For batches that do not fill the chip -- fewer tiles than SIMDs, BASELINE config 2: 65 536 Mini-Cheetah states = 1 024 tiles
// on 1 024 SIMDs -- the ordinary kernel leaves every SIMD with ONE wavefront, which issues an instruction every ~4.1 ns
// whatever it depends on; two wavefronts with half the stream each issue at 2.1 ns per SIMD.  This is
// north_star's "robot per several waves" where it pays: not to parallelise a 6 x 6 product, but to shorten the stream.
// Same device functions and operations per state as aba_chain_kernel; the limbs' inertias reach the base as one partial sum per
// wavefront, so results agree with it to rounding (tests/test_gpu_parity.py, test_latency_mode_matches_the_one_wavefront_kernel).
// ---------------------------------------------------------------------------------------------------------------
// NW = 4 (fp32, batches of at most two tiles per CU -- where two wavefronts per tile leave every SIMD with one): the limbs go to four
// wavefronts, the base keeps one accumulator per wavefront (ChainFree::lds_acc .. lds_acc4), q / qd / tau are staged by wavefronts 0 / 1 / 2.
// DIFF: the program has differential segments (TelloWithArms; fp32 only).
template <class T, int NW, bool DIFF = false>
__global__ __launch_bounds__(NW * kWave) __attribute__((amdgpu_waves_per_eu(2, 2)))
void aba_chain_lm_kernel(ChainDev<T> DP, const T *__restrict__ q, const T *__restrict__ qd, const T *__restrict__ tau,
                         T *__restrict__ ydd, size_t B, T *__restrict__ scratch)
{
    ChainTables<T> P;
    P.segs = (cptr<ChainSeg>)DP.segs;
    P.links = (cptr<ChainLink>)DP.links;
    P.pairs = (cptr<ChainPair>)DP.pairs;
    P.frees = (cptr<ChainFree>)DP.frees;
    P.diffs = DIFF ? (cptr<ChainDiff>)DP.diffs : nullptr;
    P.gens = nullptr;
    P.gbodies = nullptr;
    P.cints = DIFF ? (cptr<int32_t>)DP.cints : nullptr;
    P.consts = (cptr<T>)DP.consts;
    P.n_segs = DP.n_segs;
    P.nq = DP.nq;
    P.nv = DP.nv;
    P.ori_repr = DP.ori_repr;
#pragma unroll
    for (int i = 0; i < 6; i++) P.a_root[i] = DP.a_root[i];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    T *slab = scratch + (size_t)blockIdx.x * (size_t)(DP.n_glb_slots + P.nq + 2 * P.nv) * kWave;
    // (the LAST slab row -- the launch adds one to the program's rows -- carries wavefront 1's mask of bad pivots to wavefront 0: the
    // tile's 40 KiB of LDS are four workgroups per CU exactly, there is no word to spare)
    unsigned long long *lm_bad = reinterpret_cast<unsigned long long *>(slab + (size_t)(DP.n_glb_slots - 1 + P.nq + 2 * P.nv) * kWave);
    ChainMem<T> M;
    M.bad = 0;
    M.lane = lane;
    M.lane_b = (unsigned)lane * (unsigned)sizeof(T);
    M.gmul = 1;
    M.amask = ~0;
    M.glb_u = slab + (size_t)(P.nq + 2 * P.nv) * kWave;
    M.in_q_u = slab;
    M.in_qd_u = slab + (size_t)P.nq * kWave;
    M.in_x_u = slab + (size_t)(P.nq + P.nv) * kWave;
    M.out_u = slab + (size_t)(P.nq + P.nv) * kWave;
    M.set_slab(slab, DP.n_glb_slots + P.nq + 2 * P.nv, P.nq, P.nv);
    M.set_out(DP.out_lds);
    const unsigned bq = (unsigned)(kWave * P.nq) * (unsigned)sizeof(T), bv = (unsigned)(kWave * P.nv) * (unsigned)sizeof(T);

    const size_t n_tiles = (B + kWave - 1) / kWave;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t left = B - tile * kWave;
        const int rows_valid = left < (size_t)kWave ? (int)left : kWave;
        // prologue: each wavefront stages its arrays through its own part of LDS (capi.cpp sizes LDS for all three at once)
        if (wave == 0) {
            stage_issue(q, tile, rows_valid, P.nq, 0u, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wave_lds_fence();
            stage_transpose(P.nq, 0u, slab, lane);
        } else if (NW == 2) {
            stage_issue(qd, tile, rows_valid, P.nv, bq, lane);
            stage_issue(tau, tile, rows_valid, P.nv, bq + bv, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wave_lds_fence();
            stage_transpose(P.nv, bq, slab + (size_t)P.nq * kWave, lane);
            stage_transpose(P.nv, bq + bv, slab + (size_t)(P.nq + P.nv) * kWave, lane);
        } else if (wave < 3) {  // wavefront 1: qd, wavefront 2: tau
            const unsigned at = wave == 1 ? bq : bq + bv;
            stage_issue(wave == 1 ? qd : tau, tile, rows_valid, P.nv, at, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wave_lds_fence();
            stage_transpose(P.nv, at, slab + (size_t)(wave == 1 ? P.nq : P.nq + P.nv) * kWave, lane);
        }
        __syncthreads();  // (drains the slab stores of every wavefront: the rows are visible to all)
        for (int s = 0; s < P.n_segs; s++) {
            const ChainSeg sg = load_rec(P.segs + s);
            if (sg.op == SEG_BARRIER) {
                if (sg.head) {
                    __syncthreads();  // drains the global stores: the limbs' accumulators went through the slab
                } else {  // LDS hand-over only: the [K | y0] stores in flight need not land first
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                continue;
            }
            if (sg.owner != wave) continue;
            switch (sg.op) {
                case SEG_RUN_FWD: run_fwd<T, false>(P, M, sg); break;
                case SEG_RUN_BWD: run_bwd<T, false, false>(P, M, sg); break;
                case SEG_RUN_ACC: run_acc(P, M, sg); break;
                case SEG_PAIR_ACC: pair_acc(P, M, load_rec(P.pairs + sg.first)); break;
                case SEG_FREE_FWD: free_fwd(P, M, load_rec(P.frees + sg.first)); break;
                case SEG_FREE_BWD: free_bwd<T, false>(P, M, load_rec(P.frees + sg.first)); break;
                case SEG_DIFF_FWD:
                    if constexpr (DIFF) diff_fwd(P, M, load_rec(P.diffs + sg.first));
                    break;
                case SEG_DIFF_BWD:
                    if constexpr (DIFF) diff_bwd<T, false>(P, M, load_rec(P.diffs + sg.first));
                    break;
                case SEG_DIFF_ACC:
                    if constexpr (DIFF) diff_acc(P, M, load_rec(P.diffs + sg.first));
                    break;
                default: free_acc(P, M, load_rec(P.frees + sg.first)); break;
            }
        }
        // (a state with a bad pivot is counted once: the other wavefronts hand their masks to wavefront 0)
        if (wave != 0 && lane == 0) lm_bad[wave - 1] = M.bad;
        __syncthreads();  // every result row is in the slab (or in LDS: ChainProgram::out_lds)
        if (wave == 0) {
            if (M.out_row >= 0) write_outputs_lds<T>(M.out_row, ydd, tile, rows_valid, P.nv, lane);
            else write_outputs(slab + (size_t)(P.nq + P.nv) * kWave, ydd, tile, rows_valid, P.nv, lane);
#pragma unroll
            for (int w2 = 1; w2 < NW; w2++) M.bad |= lm_bad[w2 - 1];
            M.flush_bad(DP.bad_count, rows_valid);
        } else {
            M.bad = 0;
        }
        // LDS and the slab are free for the next tile once wavefront 0 has READ the result rows (its own output stores may still
        // be in flight)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
}

hipError_t launch_aba_chain_lm4(const ChainDev<float> &P, const float *q, const float *qd, const float *tau, float *ydd, size_t B, float *scratch,
                                int grid, size_t lds_bytes, hipStream_t stream, int n_waves);
hipError_t launch_aba_chain_lm4_f64(const ChainDev<double> &P, const double *q, const double *qd, const double *tau, double *ydd, size_t B,
                                    double *scratch, int grid, size_t lds_bytes, hipStream_t stream);
template <class T>
hipError_t launch_aba_chain_lm(const ChainDev<T> &P, const T *q, const T *qd, const T *tau, T *ydd, size_t B, T *scratch, int grid,
                               size_t lds_bytes, hipStream_t stream, int n_waves)
{
    if constexpr (sizeof(T) == 4) {
        return launch_aba_chain_lm4(P, q, qd, tau, ydd, B, scratch, grid, lds_bytes, stream, n_waves);  // (unit 3)
    } else {
        if (n_waves == 4) return launch_aba_chain_lm4_f64(P, q, qd, tau, ydd, B, scratch, grid, lds_bytes, stream);  // (unit 3: blocks in LDS)
        if (n_waves != 2) return hipErrorInvalidValue;
        hipLaunchKernelGGL((aba_chain_lm_kernel<T, 2>), dim3(grid), dim3(2 * kWave), lds_bytes, stream, P, q, qd, tau, ydd, B, scratch);
        return hipGetLastError();
    }
}
#if GRBDA_CHAIN_UNIT == 3
hipError_t launch_aba_chain_lm4(const ChainDev<float> &P, const float *q, const float *qd, const float *tau, float *ydd, size_t B, float *scratch,
                                int grid, size_t lds_bytes, hipStream_t stream, int n_waves)
{
    const bool diff = P.n_diffs > 0;
    if (n_waves == 4 && diff) hipLaunchKernelGGL((aba_chain_lm_kernel<float, 4, true>), dim3(grid), dim3(4 * kWave), lds_bytes, stream, P, q, qd, tau, ydd, B, scratch);
    else if (n_waves == 4) hipLaunchKernelGGL((aba_chain_lm_kernel<float, 4>), dim3(grid), dim3(4 * kWave), lds_bytes, stream, P, q, qd, tau, ydd, B, scratch);
    else if (n_waves == 2 && diff) hipLaunchKernelGGL((aba_chain_lm_kernel<float, 2, true>), dim3(grid), dim3(2 * kWave), lds_bytes, stream, P, q, qd, tau, ydd, B, scratch);
    else if (n_waves == 2) hipLaunchKernelGGL((aba_chain_lm_kernel<float, 2>), dim3(grid), dim3(2 * kWave), lds_bytes, stream, P, q, qd, tau, ydd, B, scratch);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
hipError_t launch_aba_chain_lm4_f64(const ChainDev<double> &P, const double *q, const double *qd, const double *tau, double *ydd, size_t B,
                                    double *scratch, int grid, size_t lds_bytes, hipStream_t stream)
{
    hipLaunchKernelGGL((aba_chain_lm_kernel<double, 4>), dim3(grid), dim3(4 * kWave), lds_bytes, stream, P, q, qd, tau, ydd, B, scratch);
    return hipGetLastError();
}
#endif
#if GRBDA_CHAIN_UNIT == 0
template hipError_t launch_aba_chain_lm<float>(const ChainDev<float> &, const float *, const float *, const float *, float *, size_t,
                                               float *, int, size_t, hipStream_t, int);
#elif GRBDA_CHAIN_UNIT == 1
template hipError_t launch_aba_chain_lm<double>(const ChainDev<double> &, const double *, const double *, const double *, double *,
                                                size_t, double *, int, size_t, hipStream_t, int);
#endif

// ---------------------------------------------------------------------------------------------------------------
// Inverse operational-space inertia J H^-1 J^T of contact frames by force propagation -- the recursion behind
// ClusterTreeModel::inverseOperationalSpaceInertiaMatrix / applyTestForce (ClusterTreeDynamics.cpp:194-233,295-435:
// ChiUp = Xup (1 - S D^-1 U^T), lambda_inv += (S^T f)^T D^-1 (S^T f), f <- ChiUp^T f).
// After the forward and backward runs (articulated inertias, K = D^-1 F^T, D^-1 per cluster) every contact frame e walks
// from its body to the root with the 6 x 6 matrix K_e = wrench on the current body per unit contact wrench:
//   s = S^T K_e (n x 6),   W_e[rows of the cluster] = D^-1/2 s,   K_e <- X^T K_e - F D^-1 s
// and Lambda^-1[e1][e2] = sum over the clusters BOTH paths visit of s1^T D^-1 s2 = W_e1^T W_e2 over the shared rows
// (paths merge towards the root, so the shared rows are a common tail).  The Jacobians come from the same walk without
// the articulated correction: J_e[:, cluster] = (S^T X^T...X^T K0)^T.
// ---------------------------------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ void k_up(const T (&E)[9], cptr<T> r, T (&K)[36])
{  // every column through inverseTransformForceVector
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const T f[6] = {K[j], K[6 + j], K[12 + j], K[18 + j], K[24 + j], K[30 + j]};
        T o[6];
        xforce_inv(E, r, f, o);
#pragma unroll
        for (int i = 0; i < 6; i++) K[6 * i + j] = o[i];
    }
}

template <class T>
__device__ __forceinline__ T dot6w(const T (&a)[6], const T (&b)[6])
{
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}

template <class T>
__global__ __launch_bounds__(kWave, 1) void osim_chain_kernel(ChainDev<T> DP, OsimArgs<T> A, const T *__restrict__ q,
                                                              const T *__restrict__ zeros, T *__restrict__ Linv,
                                                              T *__restrict__ Jout, size_t B, T *__restrict__ scratch)
{
    ChainTables<T> P;
    P.segs = (cptr<ChainSeg>)DP.segs;
    P.links = (cptr<ChainLink>)DP.links;
    P.pairs = (cptr<ChainPair>)DP.pairs;
    P.frees = (cptr<ChainFree>)DP.frees;
    P.diffs = (cptr<ChainDiff>)DP.diffs;
    P.gens = nullptr;
    P.gbodies = nullptr;
    P.cints = (cptr<int32_t>)DP.cints;
    P.consts = (cptr<T>)DP.consts;
    P.n_segs = DP.n_segs;
    P.nq = DP.nq;
    P.nv = DP.nv;
    P.ori_repr = DP.ori_repr;
#pragma unroll
    for (int i = 0; i < 6; i++) P.a_root[i] = DP.a_root[i];
    const int lane = threadIdx.x;
    const int n_rows_wave = DP.n_glb_slots + P.nq + 2 * P.nv;  // DP.n_glb_slots includes the W blocks (capi.cpp)
    T *slab = scratch + (size_t)blockIdx.x * (size_t)n_rows_wave * kWave;
    ChainMem<T> M;
    M.bad = 0;
    M.lane = lane;
    M.lane_b = (unsigned)lane * (unsigned)sizeof(T);
    M.gmul = 1;
    M.amask = ~0;
    M.glb_u = slab + (size_t)(P.nq + 2 * P.nv) * kWave;
    M.in_q_u = slab;
    M.in_qd_u = slab + (size_t)P.nq * kWave;
    M.in_x_u = slab + (size_t)(P.nq + P.nv) * kWave;
    M.out_u = slab + (size_t)(P.nq + P.nv) * kWave;
    M.set_slab(slab, DP.n_glb_slots + P.nq + 2 * P.nv, P.nq, P.nv);
    M.set_out(-1);
    const int m = A.n_contacts, nv = P.nv;
    const size_t n_tiles = (B + kWave - 1) / kWave;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t left = B - tile * kWave;
        const int rows_valid = left < (size_t)kWave ? (int)left : kWave;
        const size_t st = tile * kWave + lane;
        const bool live = st < B;
        // velocities and torques do not enter the articulated inertias: zeros
        stage_inputs(q, zeros, zeros, tile, rows_valid, P.nq, P.nv, slab, lane, DP.lds_bytes);
        for (int s = 0; s < P.n_segs; s++) {
            const ChainSeg sg = load_rec(P.segs + s);
            switch (sg.op) {
                case SEG_RUN_FWD:
                    if (DP.sv_global) run_fwd<T, true>(P, M, sg);
                    else run_fwd<T, false>(P, M, sg);
                    break;
                case SEG_RUN_BWD: {
                    if (DP.sv_global) run_bwd<T, true, true>(P, M, sg);
                    else run_bwd<T, true, false>(P, M, sg);
                    break;
                }
                case SEG_FREE_FWD: free_fwd(P, M, load_rec(P.frees + sg.first)); break;
                case SEG_FREE_BWD: free_bwd<T, true>(P, M, load_rec(P.frees + sg.first)); break;
                case SEG_DIFF_FWD: diff_fwd(P, M, load_rec(P.diffs + sg.first)); break;
                case SEG_DIFF_BWD: diff_bwd<T, true>(P, M, load_rec(P.diffs + sg.first)); break;
                default: break;  // no acceleration sweep
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the blocks written above are read back below
        T *Js = Jout ? Jout + (live ? st : B - 1) * (size_t)(6 * m) * nv : nullptr;
        if (Js && live)
            for (int i = 0; i < 6 * m * nv; i++) Js[i] = 0;
        // ---- walks ----
        T Etot[9];  // applyTestForce mode: rotation world -> contact body (plan frames), built up along the path
#pragma unroll
        for (int i = 0; i < 9; i++) Etot[i] = (i % 4 == 0) ? T(1) : T(0);
        auto turn = [&](const T(&E)[9]) {  // Etot <- Etot E
            T R[9];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) R[3 * i + j] = Etot[3 * i] * E[j] + Etot[3 * i + 1] * E[3 + j] + Etot[3 * i + 2] * E[6 + j];
#pragma unroll
            for (int i = 0; i < 9; i++) Etot[i] = R[i];
        };
        for (int e = 0; e < m; e++) {
            T K[36], Kp[36];
#pragma unroll
            for (int i = 0; i < 36; i++) K[i] = Kp[i] = A.K0[e][i];
            const int wbase = A.w_base + e * A.w_stride;
            for (int t = 0; t < A.path_len[e]; t++) {
                const OsimStep stp = A.path[e][t];
                if (stp.kind == OSIM_LINK) {
                    const ChainLink l = load_rec(P.links + stp.rec);
                    cptr<T> C = P.consts + l.cofs;
                    const T g0 = C[kBodyConstFixed];
                    T kb[10], E[9];
                    M.glb_ld(l.glb_k, kb);
                    rotate_z(kb[7], kb[8], C, E);
                    if (A.test_force) turn(E);
                    const T sq = sqrt(kb[9]);
                    T srow[6], w[6];
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        srow[j] = g0 * K[12 + j];
                        w[j] = srow[j] * sq;
                    }
                    M.glb_st(wbase + stp.w_row * 6, w);
                    if (Js && live) {
#pragma unroll
                        for (int j = 0; j < 6; j++) Js[(size_t)(6 * e + j) * nv + stp.v_index] = g0 * Kp[12 + j];
                    }
                    k_up(E, C + 9, K);
                    if (Js) k_up(E, C + 9, Kp);
#pragma unroll
                    for (int i = 0; i < 6; i++)
#pragma unroll
                        for (int j = 0; j < 6; j++) K[6 * i + j] -= kb[i] * srow[j];
                } else if (stp.kind == OSIM_FREE) {
                    const ChainFree f = load_rec(P.frees + stp.rec);
                    T ex[27];
                    M.glb_ld(f.glb_y0 + 6, ex);
                    if (A.test_force) {
                        T o[4], Eb[9];
                        const int nori = P.ori_repr == 0 ? 4 : 3;
#pragma unroll
                        for (int j = 0; j < 4; j++) o[j] = j < nori ? M.q(f.q_index + 3 + j) : T(0);
                        free_rotation(P.ori_repr, o, Eb);
                        turn(Eb);
                    }
                    // W = L^-1 K (forward substitution per column), S = 1
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        T y[6];
#pragma unroll
                        for (int i = 0; i < 6; i++) {
                            T sacc = K[6 * i + j];
#pragma unroll
                            for (int k2 = 0; k2 < i; k2++) sacc -= ex[i * (i + 1) / 2 + k2] * y[k2];
                            y[i] = sacc * ex[21 + i];
                        }
#pragma unroll
                        for (int i = 0; i < 6; i++) M.glb_u[(size_t)(wbase + (stp.w_row + i) * 6 + j) * kWave + lane] = y[i];
                    }
                    if (Js && live) {
#pragma unroll
                        for (int i = 0; i < 6; i++)
#pragma unroll
                            for (int j = 0; j < 6; j++) Js[(size_t)(6 * e + j) * nv + stp.v_index + i] = Kp[6 * i + j];
                    }
                } else if (stp.kind == OSIM_DIFF_LINK1 || stp.kind == OSIM_DIFF_LINK2) {
                    // two-rotor differential (ChainDiff): the path enters at link1 (a contact on it) or at link2 (a contact
                    // on it or anything below it).  S^T K = G_l1^T (z^T K at link1) + G_l2^T (z^T K at link2), G rows (X00,
                    // X01) and (X10, X11) of the state's constraint Jacobian; the rotors see nothing of the wrench
                    const ChainDiff df = load_rec(P.diffs + stp.rec);
                    cptr<T> C1 = P.consts + df.cofs[0], C2 = P.consts + df.cofs[1];
                    T blk[27], E1[9], E2[9];
                    M.glb_ld(df.glb_k, blk);
                    rotate_z(blk[20], blk[21], C1, E1);
                    rotate_z(blk[22], blk[23], C2, E2);
                    if (A.test_force) {
                        if (stp.kind == OSIM_DIFF_LINK2) turn(E2);
                        turn(E1);
                    }
                    T z1[6], z2[6], y1[6], y2[6];
                    if (stp.kind == OSIM_DIFF_LINK2) {
#pragma unroll
                        for (int j = 0; j < 6; j++) { z2[j] = K[12 + j]; y2[j] = Kp[12 + j]; }
                        k_up(E2, C2 + 9, K);
                        if (Js) k_up(E2, C2 + 9, Kp);
                    } else {
#pragma unroll
                        for (int j = 0; j < 6; j++) z2[j] = y2[j] = 0;
                    }
#pragma unroll
                    for (int j = 0; j < 6; j++) { z1[j] = K[12 + j]; y1[j] = Kp[12 + j]; }
                    T s0[6], s1[6], p0[6], p1[6];
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        s0[j] = blk[14] * z1[j] + blk[16] * z2[j];
                        s1[j] = blk[15] * z1[j] + blk[17] * z2[j];
                        p0[j] = blk[14] * y1[j] + blk[16] * y2[j];
                        p1[j] = blk[15] * y1[j] + blk[17] * y2[j];
                    }
                    const T r00 = sqrt(blk[24]), r01 = blk[25] / r00, r11 = sqrt(blk[26] - r01 * r01);
                    T w1[6], w2[6];
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        w1[j] = r00 * s0[j] + r01 * s1[j];
                        w2[j] = r11 * s1[j];
                    }
                    M.glb_st(wbase + stp.w_row * 6, w1);
                    M.glb_st(wbase + (stp.w_row + 1) * 6, w2);
                    if (Js && live) {
#pragma unroll
                        for (int j = 0; j < 6; j++) {
                            Js[(size_t)(6 * e + j) * nv + stp.v_index] = p0[j];
                            Js[(size_t)(6 * e + j) * nv + stp.v_index + 1] = p1[j];
                        }
                    }
                    k_up(E1, C1 + 9, K);
                    if (Js) k_up(E1, C1 + 9, Kp);
#pragma unroll
                    for (int i = 0; i < 6; i++)
#pragma unroll
                        for (int j = 0; j < 6; j++) K[6 * i + j] -= blk[i] * s0[j] + blk[6 + i] * s1[j];
                } else {
                    // leaf pair cluster: the contact sits on link1 or on link2 (child of link1)
                    const ChainPair pr = load_rec(P.pairs + stp.rec);
                    cptr<T> C1 = P.consts + pr.cofs[0], C2 = P.consts + pr.cofs[1];
                    T blk[21], E1[9], E2[9];
                    M.glb_ld(pr.glb_k, blk);
                    rotate_z(blk[17], blk[18], C1, E1);
                    rotate_z(blk[19], blk[20], C2, E2);
                    if (A.test_force) {
                        if (stp.kind == OSIM_PAIR_LINK2) turn(E2);
                        turn(E1);
                    }
                    T s1[6], s2[6], p1[6], p2[6];
                    if (stp.kind == OSIM_PAIR_LINK2) {
#pragma unroll
                        for (int j = 0; j < 6; j++) { s2[j] = K[12 + j]; p2[j] = Kp[12 + j]; }
                        k_up(E2, C2 + 9, K);
                        if (Js) k_up(E2, C2 + 9, Kp);
                    } else {
#pragma unroll
                        for (int j = 0; j < 6; j++) s2[j] = p2[j] = 0;
                    }
#pragma unroll
                    for (int j = 0; j < 6; j++) { s1[j] = K[12 + j]; p1[j] = Kp[12 + j]; }
                    // W = R [s1; s2] with R^T R = D^-1 (upper Cholesky factor of D^-1)
                    const T r00 = sqrt(blk[14]), r01 = blk[15] / r00, r11 = sqrt(blk[16] - r01 * r01);
                    T w1[6], w2[6];
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        w1[j] = r00 * s1[j] + r01 * s2[j];
                        w2[j] = r11 * s2[j];
                    }
                    M.glb_st(wbase + stp.w_row * 6, w1);
                    M.glb_st(wbase + (stp.w_row + 1) * 6, w2);
                    if (Js && live) {
#pragma unroll
                        for (int j = 0; j < 6; j++) {
                            Js[(size_t)(6 * e + j) * nv + stp.v_index] = p1[j];
                            Js[(size_t)(6 * e + j) * nv + stp.v_index + 1] = p2[j];
                        }
                    }
                    k_up(E1, C1 + 9, K);
                    if (Js) k_up(E1, C1 + 9, Kp);
                    // K <- K - F D^-1 s = K - (K_blk row 0)^T s1 - (K_blk row 1)^T s2
#pragma unroll
                    for (int i = 0; i < 6; i++)
#pragma unroll
                        for (int j = 0; j < 6; j++) K[6 * i + j] -= blk[i] * s1[j] + blk[6 + i] * s2[j];
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (A.test_force) {
            // ---- applyTestForce (ClusterTreeDynamics.cpp:194-233): the contact force in the contact body's axes, then per
            // cluster of the path y0 = D^-1 S^T K_e f (from the W rows: W = D^-1/2 S^T K_e), lambda_inv = |W f|^2, and the
            // acceleration sweep of the whole model with these y0 and no gravity gives dstate = H^-1 J^T f
            const size_t sf = live ? st : B - 1;
            const T fw[3] = {A.force[sf * 3], A.force[sf * 3 + 1], A.force[sf * 3 + 2]};
            T f6[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const T fp = Etot[3 * i] * fw[0] + Etot[3 * i + 1] * fw[1] + Etot[3 * i + 2] * fw[2];
                if (A.perm[i] == 0) f6[3] = fp;
                else if (A.perm[i] == 1) f6[4] = fp;
                else f6[5] = fp;
            }
            T lam = 0;
            const int wbase = A.w_base;
            for (int t = 0; t < A.path_len[0]; t++) {
                const OsimStep stp = A.path[0][t];
                if (stp.kind == OSIM_LINK) {
                    const ChainLink l = load_rec(P.links + stp.rec);
                    T w[6], dv[1];
                    M.glb_ld(wbase + stp.w_row * 6, w);
                    M.glb_ld(l.glb_k + 9, dv);
                    const T a = dot6w(w, f6);
                    lam += a * a;
                    const T y0[1] = {sqrt(dv[0]) * a};
                    M.glb_st(l.glb_k + 6, y0);
                } else if (stp.kind == OSIM_FREE) {
                    const ChainFree f = load_rec(P.frees + stp.rec);
                    T ex[27], y[6];
                    M.glb_ld(f.glb_y0 + 6, ex);
#pragma unroll
                    for (int i = 0; i < 6; i++) {
                        T w[6];
                        M.glb_ld(wbase + (stp.w_row + i) * 6, w);
                        y[i] = dot6w(w, f6);
                        lam += y[i] * y[i];
                    }
                    // L^T x = y
#pragma unroll
                    for (int i = 5; i >= 0; i--) {
                        T sacc = y[i];
#pragma unroll
                        for (int k2 = i + 1; k2 < 6; k2++) sacc -= ex[k2 * (k2 + 1) / 2 + i] * y[k2];
                        y[i] = sacc * ex[21 + i];
                    }
                    M.glb_st(f.glb_y0, y);
                } else {
                    // pair / differential: two rows, R^T R = D^-1 with R = [[r00, r01], [0, r11]]
                    int gk;
                    T dinv[3];
                    if (stp.kind == OSIM_DIFF_LINK1 || stp.kind == OSIM_DIFF_LINK2) {
                        const ChainDiff df = load_rec(P.diffs + stp.rec);
                        gk = df.glb_k;
                        M.glb_ld(gk + 24, dinv);
                    } else {
                        const ChainPair pr = load_rec(P.pairs + stp.rec);
                        gk = pr.glb_k;
                        M.glb_ld(gk + 14, dinv);
                    }
                    T w1[6], w2[6];
                    M.glb_ld(wbase + stp.w_row * 6, w1);
                    M.glb_ld(wbase + (stp.w_row + 1) * 6, w2);
                    const T a = dot6w(w1, f6), b = dot6w(w2, f6);
                    lam += a * a + b * b;
                    const T r00 = sqrt(dinv[0]), r01 = dinv[1] / r00, r11 = sqrt(dinv[2] - r01 * r01);
                    const T y0[2] = {r00 * a, r01 * a + r11 * b};
                    M.glb_st(gk + 12, y0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int s2 = 0; s2 < P.n_segs; s2++) {
                const ChainSeg sg = load_rec(P.segs + s2);
                switch (sg.op) {
                    case SEG_RUN_ACC: run_acc(P, M, sg); break;
                    case SEG_PAIR_ACC: pair_acc(P, M, load_rec(P.pairs + sg.first)); break;
                    case SEG_DIFF_ACC: diff_acc(P, M, load_rec(P.diffs + sg.first)); break;
                    case SEG_FREE_ACC: free_acc(P, M, load_rec(P.frees + sg.first)); break;
                    default: break;
                }
            }
            if (live) A.lambda_inv[st] = lam;
            write_outputs(slab + (size_t)(P.nq + P.nv) * kWave, A.dstate, tile, rows_valid, P.nv, lane);
            continue;
        }
        // ---- Lambda^-1 blocks: W_e1^T W_e2 over the rows the two paths share ----
        T *Ls = Linv + (live ? st : B - 1) * (size_t)(36 * m * m);
        for (int e1 = 0; e1 < m; e1++)
            for (int e2 = e1; e2 < m; e2++) {
                const int nc = A.common[e1][e2];
                const int o1 = A.w_base + e1 * A.w_stride + (A.n_rows[e1] - nc) * 6;
                const int o2 = A.w_base + e2 * A.w_stride + (A.n_rows[e2] - nc) * 6;
                T acc[36];
#pragma unroll
                for (int i = 0; i < 36; i++) acc[i] = 0;
                for (int t = 0; t < nc; t++) {
                    T a[6], b[6];
                    M.glb_ld(o1 + 6 * t, a);
                    M.glb_ld(o2 + 6 * t, b);
#pragma unroll
                    for (int i = 0; i < 6; i++)
#pragma unroll
                        for (int j = 0; j < 6; j++) acc[6 * i + j] += a[i] * b[j];
                }
                if (live) {
#pragma unroll
                    for (int i = 0; i < 6; i++)
#pragma unroll
                        for (int j = 0; j < 6; j++) {
                            Ls[(size_t)(6 * e1 + i) * (6 * m) + 6 * e2 + j] = acc[6 * i + j];
                            Ls[(size_t)(6 * e2 + j) * (6 * m) + 6 * e1 + i] = acc[6 * i + j];
                        }
                }
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

template <class T>
hipError_t launch_osim_chain(const ChainDev<T> &P, const OsimArgs<T> &A, const T *q, const T *zeros, T *Linv, T *J, size_t B,
                             T *scratch, int grid, size_t lds_bytes, hipStream_t stream)
{
    hipLaunchKernelGGL((osim_chain_kernel<T>), dim3(grid), dim3(kWave), lds_bytes, stream, P, A, q, zeros, Linv, J, B, scratch);
    return hipGetLastError();
}
#if GRBDA_CHAIN_UNIT == 0
template hipError_t launch_osim_chain<float>(const ChainDev<float> &, const OsimArgs<float> &, const float *, const float *, float *,
                                             float *, size_t, float *, int, size_t, hipStream_t);
template hipError_t launch_osim_chain<double>(const ChainDev<double> &, const OsimArgs<double> &, const double *, const double *,
                                              double *, double *, size_t, double *, int, size_t, hipStream_t);
#endif

hipError_t launch_aba_chain_diff_f64(const ChainDev<double> &P, const double *q, const double *qd, const double *tau, double *ydd, size_t B,
                                     double *scratch, int grid, size_t lds_bytes, hipStream_t stream);
template <class T>
hipError_t launch_aba_chain_gen(const ChainDev<T> &P, const T *q, const T *qd, const T *tau, T *ydd, size_t B, T *scratch, int grid,
                                size_t lds_bytes, hipStream_t stream);
template <class T>
hipError_t launch_aba_chain(const ChainDev<T> &P, const T *q, const T *qd, const T *tau, T *ydd, size_t B, T *scratch, int grid,
                            size_t lds_bytes, hipStream_t stream, bool four_waves_per_simd)
{
    if (P.n_gens > 0) {
        if (four_waves_per_simd) return hipErrorInvalidValue;
        return launch_aba_chain_gen<T>(P, q, qd, tau, ydd, B, scratch, grid, lds_bytes, stream);  // (unit 2)
    }
    if (P.n_diffs > 0) {
        if (four_waves_per_simd) return hipErrorInvalidValue;  // (capi.cpp keeps such programs at two wavefronts per SIMD)
        if constexpr (sizeof(T) == 8) {
            return launch_aba_chain_diff_f64(P, q, qd, tau, ydd, B, scratch, grid, lds_bytes, stream);  // (unit 1, below)
        } else {
            hipLaunchKernelGGL((aba_chain_kernel<T, 2, 1>), dim3(grid), dim3(kWave), lds_bytes, stream, P, q, qd, tau, ydd, B, scratch);
            return hipGetLastError();
        }
    }
    if constexpr (sizeof(T) == 4) {
        if (four_waves_per_simd) {
            hipLaunchKernelGGL((aba_chain_kernel<T, kChainWideWps, 0>), dim3(grid), dim3(kWave), lds_bytes, stream, P, q, qd, tau, ydd, B, scratch);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((aba_chain_kernel<T, 2, 0>), dim3(grid), dim3(kWave), lds_bytes, stream, P, q, qd, tau, ydd, B, scratch);
    return hipGetLastError();
}
#if GRBDA_CHAIN_UNIT == 0
template hipError_t launch_aba_chain<float>(const ChainDev<float> &, const float *, const float *, const float *, float *, size_t,
                                            float *, int, size_t, hipStream_t, bool);
template hipError_t launch_aba_chain<double>(const ChainDev<double> &, const double *, const double *, const double *, double *,
                                             size_t, double *, int, size_t, hipStream_t, bool);
#elif GRBDA_CHAIN_UNIT == 1
hipError_t launch_aba_chain_diff_f64(const ChainDev<double> &P, const double *q, const double *qd, const double *tau, double *ydd, size_t B,
                                     double *scratch, int grid, size_t lds_bytes, hipStream_t stream)
{
    hipLaunchKernelGGL((aba_chain_kernel<double, 2, 1>), dim3(grid), dim3(kWave), lds_bytes, stream, P, q, qd, tau, ydd, B, scratch);
    return hipGetLastError();
}
#elif GRBDA_CHAIN_UNIT == 2
template <class T>
hipError_t launch_aba_chain_gen(const ChainDev<T> &P, const T *q, const T *qd, const T *tau, T *ydd, size_t B, T *scratch, int grid,
                                size_t lds_bytes, hipStream_t stream)
{
    // fp64: ONE wavefront per SIMD.  At two the kernel spills 1 377 registers (768 B of scratch per lane); with 512 registers it needs 428 and
    // none -- teleop_arm 0.526 -> 0.247 ms, planar leg linkage 0.199 -> 0.108, a tree of triple clusters 0.703 -> 0.307 per 262 144 states
    // (tools/ab_gen64.py; -DGRBDA_EXP_GEN64_WPS=2 rebuilds the old shape)
#ifndef GRBDA_EXP_GEN64_WPS
#define GRBDA_EXP_GEN64_WPS 1
#endif
    if constexpr (sizeof(T) == 8)
        hipLaunchKernelGGL((aba_chain_kernel<T, GRBDA_EXP_GEN64_WPS, 2>), dim3(grid), dim3(kWave), lds_bytes, stream, P, q, qd, tau, ydd, B, scratch);
    else
        hipLaunchKernelGGL((aba_chain_kernel<T, 2, 2>), dim3(grid), dim3(kWave), lds_bytes, stream, P, q, qd, tau, ydd, B, scratch);
    return hipGetLastError();
}
template hipError_t launch_aba_chain_gen<float>(const ChainDev<float> &, const float *, const float *, const float *, float *, size_t,
                                                float *, int, size_t, hipStream_t);
template hipError_t launch_aba_chain_gen<double>(const ChainDev<double> &, const double *, const double *, const double *, double *,
                                                 size_t, double *, int, size_t, hipStream_t);
#endif


// ===============================================================================================================
// Inverse dynamics on the chains: TreeModel::recursiveNewtonEulerAlgorithm (src/Dynamics/TreeModel.cpp:34-57,173-212)
// ===============================================================================================================
template <class T>
struct RneaTables {
    cptr<RneaSeg> segs;
    cptr<RneaLink> links;
    cptr<RneaPair> pairs;
    cptr<RneaFree> frees;
    cptr<RneaDiff> diffs;
    cptr<ChainGen> gens;          // generic clusters (gen_rnea_segments.h)
    cptr<ChainGenBody> gbodies;
    cptr<int32_t> cints;
    cptr<T> consts;
    int n_segs, nq, nv, ori_repr;
    T a_root[6];
};

// f = I a + v x* (I v)  (TreeModel.cpp:185-189), general constant inertia
template <class T>
__device__ __forceinline__ void body_force_c(cptr<T> I, const T (&v)[6], const T (&a)[6], T (&f)[6])
{
    T Ia[6], Iv[6];
    symv_c(I, a, Ia);
    symv_c(I, v, Iv);
    crf(v, Iv, f);
#pragma unroll
    for (int j = 0; j < 6; j++) f[j] += Ia[j];
}

// axisymmetric rotor at q = 0 (see rotor_terms): torque about its axis and its force on the parent body
template <class T>
__device__ __forceinline__ void rotor_rnea(cptr<T> Cr, const T (&vp)[6], const T (&ap)[6], T qdr, T qddr, T &tau_z, T (&fp)[6])
{
    cptr<T> Ir = Cr + 12;
    const T A0 = Ir[sidx(0, 0)], A1 = Ir[sidx(1, 1)], Bz = Ir[sidx(2, 2)], k04 = Ir[sidx(0, 4)], k13 = Ir[sidx(1, 3)];
    const T m3 = Ir[sidx(3, 3)], m4 = Ir[sidx(4, 4)], m5 = Ir[sidx(5, 5)];
    T E0[9], vr[6], ar[6];
#pragma unroll
    for (int j = 0; j < 9; j++) E0[j] = Cr[j];
    xmotion(E0, Cr + 9, vp, vr);
    xmotion(E0, Cr + 9, ap, ar);
    vr[2] += qdr;
    ar[0] += vr[1] * qdr;
    ar[1] -= vr[0] * qdr;
    ar[3] += vr[4] * qdr;
    ar[4] -= vr[3] * qdr;
    ar[2] += qddr;
    const T Iv[6] = {A0 * vr[0] + k04 * vr[4], A1 * vr[1] + k13 * vr[3], Bz * vr[2],
                     m3 * vr[3] + k13 * vr[1], m4 * vr[4] + k04 * vr[0], m5 * vr[5]};
    T f[6];
    crf(vr, Iv, f);
    f[0] += A0 * ar[0] + k04 * ar[4];
    f[1] += A1 * ar[1] + k13 * ar[3];
    f[2] += Bz * ar[2];
    f[3] += m3 * ar[3] + k13 * ar[1];
    f[4] += m4 * ar[4] + k04 * ar[0];
    f[5] += m5 * ar[5];
    tau_z = f[2];
    xforce_inv(E0, Cr + 9, f, fp);
}

// GLB: the program keeps some of the links' [f | sin, cos | rotor torque] blocks in the wave's global slab (RneaChainProgram::
// n_glb > 0: chains too long for the LDS budget, JVRC-1); a compile-time property of the kernel so that the common case
// carries no test of where a block lives.
template <class T, bool GLB>
__device__ __forceinline__ void add6(const ChainMem<T> &M, int slot, const T (&x)[6])
{
    T y[6];
    if constexpr (GLB) M.acc_ld(slot, y);
    else M.lds_ld(slot, y);
#pragma unroll
    for (int j = 0; j < 6; j++) y[j] += x[j];
    if constexpr (GLB) M.acc_st(slot, y);
    else M.lds_st(slot, y);
}

template <class T, bool GLB>
__device__ __forceinline__ void rnea_run_fwd(const RneaTables<T> &P, const ChainMem<T> &M, const RneaSeg &sg)
{
    T vp[6], ap[6];
    if (sg.lds_pva >= 0) {
        T va[12];
        M.lds_ld(sg.lds_pva, va);
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
    RneaLink l = load_rec(P.links + sg.first);
    T qi = M.q(l.q_index), ydi = M.qd(l.v_index), yddi = M.x(l.v_index);
    for (int i = 0; i < sg.count; i++) {
        const bool more = i + 1 < sg.count;
        const RneaLink ln = load_rec(P.links + (sg.first + (more ? i + 1 : i)));
        T qn = 0, ydn = 0, yddn = 0;
        if (more) {
            qn = M.q(ln.q_index);
            ydn = M.qd(ln.v_index);
            yddn = M.x(ln.v_index);
        }
        cptr<T> C = P.consts + l.cofs;
        const T g0 = C[kBodyConstFixed];
        const T qdi = g0 * ydi;
        T blk[9], v[6], a[6], f[6];
        sincos_t(g0 * qi, &blk[6], &blk[7]);
        link_down2(perm_if<T>(GRBDA_PERM_RNEA, l.perm), blk[6], blk[7], C, vp, ap, v, a);
        v[2] += qdi;
        a[0] += v[1] * qdi;
        a[1] -= v[0] * qdi;
        a[3] += v[4] * qdi;
        a[4] -= v[3] * qdi;
        a[2] += g0 * yddi;
        body_force_c(C + 12, v, a, f);
        blk[8] = 0;
        if (l.rofs >= 0 && !l.general_rotor) {
            cptr<T> Cr = P.consts + l.rofs;
            const T gr = Cr[kBodyConstFixed];
            T tz, fpr[6];
            rotor_rnea(Cr, vp, ap, gr * ydi, gr * yddi, tz, fpr);
            blk[8] = gr * tz;
            if (l.lds_pf >= 0) add6<T, GLB>(M, l.lds_pf, fpr);  // the rotor hangs off the parent body
        }
        if (l.rofs >= 0 && l.general_rotor) {  // general rotor: a second full body of the cluster, at its own angle
            cptr<T> Cr = P.consts + l.rofs;
            const T gr = Cr[kBodyConstFixed];
            const T qdr = gr * ydi;
            T sr, cr_, vr[6], ar[6], fr[6], fpr[6];
            sincos_t(gr * qi, &sr, &cr_);
            link_down2(perm_if<T>(GRBDA_PERM_RNEA, l.rperm), sr, cr_, Cr, vp, ap, vr, ar);
            vr[2] += qdr;
            ar[0] += vr[1] * qdr;
            ar[1] -= vr[0] * qdr;
            ar[3] += vr[4] * qdr;
            ar[4] -= vr[3] * qdr;
            ar[2] += gr * yddi;
            body_force_c(Cr + 12, vr, ar, fr);
            blk[8] = gr * fr[2];
            link_force_up(perm_if<T>(GRBDA_PERM_RNEA, l.rperm), sr, cr_, Cr, fr, fpr);
            if (l.lds_pf >= 0) add6<T, GLB>(M, l.lds_pf, fpr);
        }
#pragma unroll
        for (int j = 0; j < 6; j++) blk[j] = f[j];
        if constexpr (GLB) M.acc_st(l.lds_blk, blk);
        else M.lds_st(l.lds_blk, blk);
        if (l.lds_va >= 0) {
            T va[12];
#pragma unroll
            for (int j = 0; j < 6; j++) {
                va[j] = v[j];
                va[6 + j] = a[j];
            }
            M.lds_st(l.lds_va, va);
        }
#pragma unroll
        for (int j = 0; j < 6; j++) {
            vp[j] = v[j];
            ap[j] = a[j];
        }
        l = ln;
        qi = qn;
        ydi = ydn;
        yddi = yddn;
    }
}

template <class T, bool GLB>
__device__ __forceinline__ void rnea_run_bwd(const RneaTables<T> &P, const ChainMem<T> &M, const RneaSeg &sg)
{
    T fc[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < sg.count; i++) {
        const RneaLink l = load_rec(P.links + (sg.first + i));
        cptr<T> C = P.consts + l.cofs;
        T blk[9], ft[6];
        if constexpr (GLB) M.acc_ld(l.lds_blk, blk);
        else M.lds_ld(l.lds_blk, blk);
#pragma unroll
        for (int j = 0; j < 6; j++) ft[j] = blk[j] + fc[j];
        M.put(l.v_index, C[kBodyConstFixed] * ft[2] + blk[8]);
        link_force_up(perm_if<T>(GRBDA_PERM_RNEA, l.perm), blk[6], blk[7], C, ft, fc);
    }
    if (sg.lds_pf >= 0) add6<T, GLB>(M, sg.lds_pf, fc);
}

// leaf pair cluster: link1 (on P), link2 (on link1), two axisymmetric rotors on P (see pair_bwd)
template <class T, bool GLB>
__device__ __forceinline__ void rnea_pair(const RneaTables<T> &P, const ChainMem<T> &M, const RneaPair &pr)
{
    cptr<T> C1 = P.consts + pr.cofs[0], C2 = P.consts + pr.cofs[1];
    T va[12], vp[6], ap[6];
    M.lds_ld(pr.lds_pva, va);
#pragma unroll
    for (int j = 0; j < 6; j++) {
        vp[j] = va[j];
        ap[j] = va[6 + j];
    }
    const T y1 = M.q(pr.q_index), y2 = M.q(pr.q_index + 1);
    const T yd1 = M.qd(pr.v_index), yd2 = M.qd(pr.v_index + 1);
    const T ydd1 = M.x(pr.v_index), ydd2 = M.x(pr.v_index + 1);
    T s1, c1, s2, c2, E1[9], E2[9], v1[6], a1[6], v2[6], a2[6], f1[6], f2[6];
    sincos_t(y1, &s1, &c1);
    rotate_z(s1, c1, C1, E1);
    xmotion(E1, C1 + 9, vp, v1);
    xmotion(E1, C1 + 9, ap, a1);
    v1[2] += yd1;
    a1[0] += v1[1] * yd1; a1[1] -= v1[0] * yd1; a1[3] += v1[4] * yd1; a1[4] -= v1[3] * yd1;
    a1[2] += ydd1;
    sincos_t(y2, &s2, &c2);
    rotate_z(s2, c2, C2, E2);
    xmotion(E2, C2 + 9, v1, v2);
    xmotion(E2, C2 + 9, a1, a2);
    v2[2] += yd2;
    a2[0] += v2[1] * yd2; a2[1] -= v2[0] * yd2; a2[3] += v2[4] * yd2; a2[4] -= v2[3] * yd2;
    a2[2] += ydd2;
    body_force_c(C1 + 12, v1, a1, f1);
    body_force_c(C2 + 12, v2, a2, f2);
    T tau1, tau2 = f2[2], f21[6], fp[6];
    xforce_inv(E2, C2 + 9, f2, f21);
#pragma unroll
    for (int j = 0; j < 6; j++) f1[j] += f21[j];
    tau1 = f1[2];
    xforce_inv(E1, C1 + 9, f1, fp);
#pragma unroll
    for (int r = 0; r < 2; r++) {
        cptr<T> Cr = P.consts + pr.cofs[2 + r];
        const T ga = Cr[kBodyConstFixed], gb = Cr[kBodyConstFixed + 1];
        T tz, fpr[6];
        rotor_rnea(Cr, vp, ap, ga * yd1 + gb * yd2, ga * ydd1 + gb * ydd2, tz, fpr);
        tau1 += ga * tz;
        tau2 += gb * tz;
#pragma unroll
        for (int j = 0; j < 6; j++) fp[j] += fpr[j];
    }
    M.put(pr.v_index, tau1);
    M.put(pr.v_index + 1, tau2);
    add6<T, GLB>(M, pr.lds_pf, fp);
}


// two-rotor differential cluster (plan.h, RneaDiff; see diff_constraint): forward segment -- G and g from the constraint,
// v, a and the body forces of the two links, the rotors' torques and their forces on the parent body; backward segment
// (after the child segments have added their forces to link2's) -- tau = G^T (S^T f) and the links' force to the parent.
template <class T, bool GLB>
__device__ __forceinline__ void rnea_diff_fwd(const RneaTables<T> &P, const ChainMem<T> &M, const RneaDiff &d)
{
    cptr<T> C1 = P.consts + d.cofs[0], C2 = P.consts + d.cofs[1];
    T qs[4];
#pragma unroll
    for (int i = 0; i < 4; i++) qs[i] = M.q(d.q_index + d.qpos[i]);
    const T yd0 = M.qd(d.v_index), yd1 = M.qd(d.v_index + 1);
    const T ydd0 = M.x(d.v_index), ydd1 = M.x(d.v_index + 1);
    T X[4], g[2], qdl[2];
    if (d.tofs_i >= 0) {
        diff_constraint(P, M, d.tofs_i, d.lds_w, qs, yd0, yd1, X, g, qdl);
    } else {  // explicit pair: the link angles are the coordinates
        X[0] = 1; X[1] = 0; X[2] = 0; X[3] = 1;
        g[0] = g[1] = 0;
        qdl[0] = yd0;
        qdl[1] = yd1;
    }
    const T qdd1 = X[0] * ydd0 + X[1] * ydd1 + g[0], qdd2 = X[2] * ydd0 + X[3] * ydd1 + g[1];
    T vp[6], ap[6];
    if (d.lds_pva >= 0) {
        T va[12];
        M.lds_ld(d.lds_pva, va);
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
    T sc[4], E1[9], E2[9], v1[6], v2[6], a1[6], a2[6], c[6];
    sincos_cw(qs[2], &sc[0], &sc[1]);
    sincos_cw(qs[3], &sc[2], &sc[3]);
    diff_links(C1, C2, sc, vp, qdl[0], qdl[1], E1, E2, v1, v2);
    xmotion(E1, C1 + 9, ap, a1);
    vxz(v1, qdl[0], c);
#pragma unroll
    for (int j = 0; j < 6; j++) a1[j] += c[j];
    a1[2] += qdd1;
    xmotion(E2, C2 + 9, a1, a2);
    vxz(v2, qdl[1], c);
#pragma unroll
    for (int j = 0; j < 6; j++) a2[j] += c[j];
    a2[2] += qdd2;
    T f1[6], f2[6], blk[14];
    body_force_c(C1 + 12, v1, a1, f1);
    body_force_c(C2 + 12, v2, a2, f2);
    // what does not wait for the child segments goes out now: link1's own force and the rotors' to the parent body,
    // their share of tau to the result rows (the backward segment adds link2's)
    T tz0, tz1, fp[6], fp0[6], fp1[6];
    const T g00 = P.consts[d.gofs], g01 = P.consts[d.gofs + 1], g10 = P.consts[d.gofs + 2], g11 = P.consts[d.gofs + 3];
    rotor_rnea(P.consts + d.cofs[2], vp, ap, g00 * yd0 + g01 * yd1, g00 * ydd0 + g01 * ydd1, tz0, fp0);
    rotor_rnea(P.consts + d.cofs[3], vp, ap, g10 * yd0 + g11 * yd1, g10 * ydd0 + g11 * ydd1, tz1, fp1);
    xforce_inv(E1, C1 + 9, f1, fp);
#pragma unroll
    for (int j = 0; j < 6; j++) fp[j] += fp0[j] + fp1[j];
    if (d.lds_pf >= 0) add6<T, GLB>(M, d.lds_pf, fp);
    M.put(d.v_index, g00 * tz0 + g10 * tz1 + X[0] * f1[2]);
    M.put(d.v_index + 1, g01 * tz0 + g11 * tz1 + X[1] * f1[2]);
#pragma unroll
    for (int j = 0; j < 6; j++) blk[j] = f2[j];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        blk[6 + j] = sc[j];
        blk[10 + j] = X[j];
    }
    M.lds_st(d.lds_blk, blk);
    if (d.lds_va >= 0) {
        T out[12];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            out[j] = v2[j];
            out[6 + j] = a2[j];
        }
        M.lds_st(d.lds_va, out);
    }
}

template <class T, bool GLB>
__device__ __forceinline__ void rnea_diff_bwd(const RneaTables<T> &P, const ChainMem<T> &M, const RneaDiff &d)
{
    cptr<T> C1 = P.consts + d.cofs[0], C2 = P.consts + d.cofs[1];
    T blk[14], f2[6], f21[6], fp[6], E1[9], E2[9];
    M.lds_ld(d.lds_blk, blk);
#pragma unroll
    for (int j = 0; j < 6; j++) f2[j] = blk[j];
    rotate_z(blk[6], blk[7], C1, E1);
    rotate_z(blk[8], blk[9], C2, E2);
    const T tl2 = f2[2];
    xforce_inv(E2, C2 + 9, f2, f21);
    const T tl1 = f21[2];
    xforce_inv(E1, C1 + 9, f21, fp);
    if (d.lds_pf >= 0) add6<T, GLB>(M, d.lds_pf, fp);
    // (the forward segment put the partial torques into the result rows)
    M.put(d.v_index, M.got(d.v_index) + blk[10] * tl1 + blk[12] * tl2);
    M.put(d.v_index + 1, M.got(d.v_index + 1) + blk[11] * tl1 + blk[13] * tl2);
}

template <class T>
__device__ __forceinline__ void rnea_free_fwd(const RneaTables<T> &P, const ChainMem<T> &M, const RneaFree &f)
{
    T o[4], E[9], r[3], g[6], a[6], v[6], fo[6];
    const int nori = P.ori_repr == 0 ? 4 : 3;
#pragma unroll
    for (int j = 0; j < 4; j++) o[j] = j < nori ? M.q(f.q_index + 3 + j) : T(0);
    free_rotation(P.ori_repr, o, E);
#pragma unroll
    for (int j = 0; j < 3; j++) r[j] = M.q(f.q_index + j);
#pragma unroll
    for (int j = 0; j < 6; j++) g[j] = P.a_root[j];
    xmotion(E, r, g, a);
#pragma unroll
    for (int j = 0; j < 6; j++) {
        v[j] = M.qd(f.v_index + j);
        a[j] += M.x(f.v_index + j);
    }
    body_force_c(P.consts + f.cofs + 12, v, a, fo);
    M.lds_st(f.lds_f, fo);
    if (f.lds_va >= 0) {
        T va[12];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            va[j] = v[j];
            va[6 + j] = a[j];
        }
        M.lds_st(f.lds_va, va);
    }
}

template <class T>
__device__ __forceinline__ void rnea_free_bwd(const RneaTables<T> &P, const ChainMem<T> &M, const RneaFree &f)
{
    T fo[6];
    M.lds_ld(f.lds_f, fo);
#pragma unroll
    for (int j = 0; j < 6; j++) M.put(f.v_index + j, fo[j]);
}

#include "gen_rnea_segments.h"

// MODE 0: runs, pairs, bases; 1: + differential clusters; 2: + generic clusters (translation unit 2)
template <class T, int MODE, bool GLB>
__global__ __launch_bounds__(kWave, 2) void rnea_chain_kernel(RneaChainDev<T> DP, const T *__restrict__ q, const T *__restrict__ qd,
                                                              const T *__restrict__ ydd, T *__restrict__ tau, size_t B,
                                                              T *__restrict__ scratch)
{
    constexpr bool DIFF = MODE >= 1, GEN = MODE == 2;
    RneaTables<T> P;
    P.segs = (cptr<RneaSeg>)DP.segs;
    P.links = (cptr<RneaLink>)DP.links;
    P.pairs = (cptr<RneaPair>)DP.pairs;
    P.frees = (cptr<RneaFree>)DP.frees;
    P.diffs = DIFF ? (cptr<RneaDiff>)DP.diffs : nullptr;
    P.gens = GEN ? (cptr<ChainGen>)DP.gens : nullptr;
    P.gbodies = GEN ? (cptr<ChainGenBody>)DP.gbodies : nullptr;
    P.cints = DIFF ? (cptr<int32_t>)DP.cints : nullptr;
    P.consts = (cptr<T>)DP.consts;
    P.n_segs = DP.n_segs;
    P.nq = DP.nq;
    P.nv = DP.nv;
    P.ori_repr = DP.ori_repr;
#pragma unroll
    for (int i = 0; i < 6; i++) P.a_root[i] = DP.a_root[i];
    const int lane = threadIdx.x;
    // wave slab: [nq + 2 nv input / result rows][n_glb_slots rows]
    T *slab = scratch + (size_t)blockIdx.x * (size_t)(P.nq + 2 * P.nv + DP.n_glb_slots) * kWave;
    ChainMem<T> M;
    M.bad = 0;
    M.lane = lane;
    M.lane_b = (unsigned)lane * (unsigned)sizeof(T);
    M.gmul = 1;
    M.amask = ~0;
    // (the torque rows stay in the slab: LDS rows measured no faster here)
    M.glb_u = slab + (size_t)(P.nq + 2 * P.nv) * kWave;
    M.in_q_u = slab;
    M.in_qd_u = slab + (size_t)P.nq * kWave;
    M.in_x_u = slab + (size_t)(P.nq + P.nv) * kWave;
    M.out_u = slab + (size_t)(P.nq + P.nv) * kWave;
    M.set_slab(slab, DP.n_glb_slots + P.nq + 2 * P.nv, P.nq, P.nv);
    M.set_out(-1);
    const size_t n_tiles = (B + kWave - 1) / kWave;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t left = B - tile * kWave;
        const int rows_valid = left < (size_t)kWave ? (int)left : kWave;
        stage_inputs(q, qd, ydd, tile, rows_valid, P.nq, P.nv, slab, lane, DP.lds_bytes);
#ifdef GRBDA_EXP_RNEA_NO_SEGS  // (ablation build, results wrong: the tile's staging and epilogue alone -- profiles/r6_rnea_occupancy.txt)
        for (int s = 0; s < 0; s++) {
#else
        for (int s = 0; s < P.n_segs; s++) {
#endif
            const RneaSeg sg = load_rec(P.segs + s);
            switch (sg.op) {
                case RSEG_RUN_FWD: rnea_run_fwd<T, GLB>(P, M, sg); break;
                case RSEG_RUN_BWD: rnea_run_bwd<T, GLB>(P, M, sg); break;
                case RSEG_PAIR: rnea_pair<T, GLB>(P, M, load_rec(P.pairs + sg.first)); break;
                case RSEG_FREE_FWD: rnea_free_fwd(P, M, load_rec(P.frees + sg.first)); break;
                case RSEG_DIFF_FWD:
                    if constexpr (DIFF) rnea_diff_fwd<T, GLB>(P, M, load_rec(P.diffs + sg.first));
                    break;
                case RSEG_DIFF_BWD:
                    if constexpr (DIFF) rnea_diff_bwd<T, GLB>(P, M, load_rec(P.diffs + sg.first));
                    break;
                case RSEG_GEN_FWD:
                    if constexpr (GEN) gen_rnea_segment<T, 0, GLB>(P, M, load_rec(P.gens + sg.first));
                    break;
                case RSEG_GEN_BWD:
                    if constexpr (GEN) gen_rnea_segment<T, 1, GLB>(P, M, load_rec(P.gens + sg.first));
                    break;
                default: rnea_free_bwd(P, M, load_rec(P.frees + sg.first)); break;
            }
        }
        write_outputs(slab + (size_t)(P.nq + P.nv) * kWave, tau, tile, rows_valid, P.nv, lane);
    }
}

#if GRBDA_CHAIN_UNIT == 3
// ---------------------------------------------------------------------------------------------------------------
// Latency mode of the inverse dynamics (plan.h, RneaChainProgram::n_waves = 2 or 4): as aba_chain_lm_kernel, a tile of 64 states on a workgroup of NW
// wavefronts.  The base's forward segment runs on wavefront 0, a barrier, the limbs on their wavefronts (RneaSeg::owner; every wavefront adds its limbs'
// forces into a block of its own, RneaFree::lds_f .. lds_f4), a barrier, the base's backward segment on wavefront 0.  Links and leaf pairs; every block
// in LDS (40 / 80 KiB per tile).  Same device functions and operations per state as rnea_chain_kernel; the limbs' forces reach the base as one partial sum
// per wavefront.
// ---------------------------------------------------------------------------------------------------------------
template <class T, int NW, bool DIFF = false>
__global__ __launch_bounds__(NW * kWave) __attribute__((amdgpu_waves_per_eu(2, 2)))
void rnea_chain_lm_kernel(RneaChainDev<T> DP, const T *__restrict__ q, const T *__restrict__ qd, const T *__restrict__ ydd, T *__restrict__ tau, size_t B,
                          T *__restrict__ scratch)
{
    RneaTables<T> P;
    P.segs = (cptr<RneaSeg>)DP.segs;
    P.links = (cptr<RneaLink>)DP.links;
    P.pairs = (cptr<RneaPair>)DP.pairs;
    P.frees = (cptr<RneaFree>)DP.frees;
    P.diffs = DIFF ? (cptr<RneaDiff>)DP.diffs : nullptr;
    P.gens = nullptr;
    P.gbodies = nullptr;
    P.cints = DIFF ? (cptr<int32_t>)DP.cints : nullptr;
    P.consts = (cptr<T>)DP.consts;
    P.n_segs = DP.n_segs;
    P.nq = DP.nq;
    P.nv = DP.nv;
    P.ori_repr = DP.ori_repr;
#pragma unroll
    for (int i = 0; i < 6; i++) P.a_root[i] = DP.a_root[i];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    T *slab = scratch + (size_t)blockIdx.x * (size_t)(P.nq + 2 * P.nv) * kWave;
    ChainMem<T> M;
    M.bad = 0;
    M.lane = lane;
    M.lane_b = (unsigned)lane * (unsigned)sizeof(T);
    M.gmul = 1;
    M.amask = ~0;
    M.glb_u = slab + (size_t)(P.nq + 2 * P.nv) * kWave;
    M.in_q_u = slab;
    M.in_qd_u = slab + (size_t)P.nq * kWave;
    M.in_x_u = slab + (size_t)(P.nq + P.nv) * kWave;
    M.out_u = slab + (size_t)(P.nq + P.nv) * kWave;
    M.set_slab(slab, P.nq + 2 * P.nv, P.nq, P.nv);
    M.set_out(-1);
    const unsigned bq = (unsigned)(kWave * P.nq) * (unsigned)sizeof(T), bv = (unsigned)(kWave * P.nv) * (unsigned)sizeof(T);
    const size_t n_tiles = (B + kWave - 1) / kWave;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t left = B - tile * kWave;
        const int rows_valid = left < (size_t)kWave ? (int)left : kWave;
        // prologue: the arrays are staged by different wavefronts through their own parts of LDS (capi.cpp sizes LDS for all three at once)
        if (wave == 0) {
            stage_issue(q, tile, rows_valid, P.nq, 0u, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wave_lds_fence();
            stage_transpose(P.nq, 0u, slab, lane);
        } else if (NW == 2) {
            stage_issue(qd, tile, rows_valid, P.nv, bq, lane);
            stage_issue(ydd, tile, rows_valid, P.nv, bq + bv, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wave_lds_fence();
            stage_transpose(P.nv, bq, slab + (size_t)P.nq * kWave, lane);
            stage_transpose(P.nv, bq + bv, slab + (size_t)(P.nq + P.nv) * kWave, lane);
        } else if (wave < 3) {
            const unsigned at = wave == 1 ? bq : bq + bv;
            stage_issue(wave == 1 ? qd : ydd, tile, rows_valid, P.nv, at, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wave_lds_fence();
            stage_transpose(P.nv, at, slab + (size_t)(wave == 1 ? P.nq : P.nq + P.nv) * kWave, lane);
        }
        __syncthreads();  // (drains the slab stores: the rows are visible to every wavefront; the staging area is free)
        for (int s = 0; s < P.n_segs; s++) {
            const RneaSeg sg = load_rec(P.segs + s);
            if (sg.op == RSEG_BARRIER) {  // LDS hand-over
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                continue;
            }
            if (sg.owner != wave) continue;
            switch (sg.op) {
                case RSEG_RUN_FWD: rnea_run_fwd<T, false>(P, M, sg); break;
                case RSEG_RUN_BWD: rnea_run_bwd<T, false>(P, M, sg); break;
                case RSEG_PAIR: rnea_pair<T, false>(P, M, load_rec(P.pairs + sg.first)); break;
                case RSEG_DIFF_FWD:
                    if constexpr (DIFF) rnea_diff_fwd<T, false>(P, M, load_rec(P.diffs + sg.first));
                    break;
                case RSEG_DIFF_BWD:
                    if constexpr (DIFF) rnea_diff_bwd<T, false>(P, M, load_rec(P.diffs + sg.first));
                    break;
                case RSEG_FREE_FWD: {
                    const RneaFree f = load_rec(P.frees + sg.first);
                    rnea_free_fwd(P, M, f);
                    const T zero[6] = {0, 0, 0, 0, 0, 0};  // the force blocks the other wavefronts' limbs add into
                    if (f.lds_f2 != -1) M.lds_st(f.lds_f2, zero);
                    if (f.lds_f3 != -1) M.lds_st(f.lds_f3, zero);
                    if (f.lds_f4 != -1) M.lds_st(f.lds_f4, zero);
                    break;
                }
                default: {
                    const RneaFree f = load_rec(P.frees + sg.first);
                    if (f.lds_f2 != -1) {
                        T x[6];
                        M.lds_ld(f.lds_f2, x);
                        add6<T, false>(M, f.lds_f, x);
                        if (f.lds_f3 != -1) {
                            M.lds_ld(f.lds_f3, x);
                            add6<T, false>(M, f.lds_f, x);
                            if (f.lds_f4 != -1) {
                                M.lds_ld(f.lds_f4, x);
                                add6<T, false>(M, f.lds_f, x);
                            }
                        }
                    }
                    rnea_free_bwd(P, M, f);
                    break;
                }
            }
        }
        __syncthreads();  // every torque row is in the slab
        if (wave == 0) write_outputs(slab + (size_t)(P.nq + P.nv) * kWave, tau, tile, rows_valid, P.nv, lane);
        // LDS and the slab are free for the next tile once wavefront 0 has read the torque rows
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
}
template <class T>
hipError_t launch_rnea_chain_lm(const RneaChainDev<T> &P, const T *q, const T *qd, const T *ydd, T *tau, size_t B, T *scratch, int grid, size_t lds_bytes,
                                hipStream_t stream, int n_waves)
{
    if constexpr (sizeof(T) == 4) {
        if (P.n_diffs > 0) {  // (fp32 programs only: plan.cpp)
            if (n_waves == 4) hipLaunchKernelGGL((rnea_chain_lm_kernel<T, 4, true>), dim3(grid), dim3(4 * kWave), lds_bytes, stream, P, q, qd, ydd, tau, B, scratch);
            else if (n_waves == 2) hipLaunchKernelGGL((rnea_chain_lm_kernel<T, 2, true>), dim3(grid), dim3(2 * kWave), lds_bytes, stream, P, q, qd, ydd, tau, B, scratch);
            else return hipErrorInvalidValue;
            return hipGetLastError();
        }
    }
    if (P.n_diffs > 0) return hipErrorInvalidValue;
    if (n_waves == 4) hipLaunchKernelGGL((rnea_chain_lm_kernel<T, 4>), dim3(grid), dim3(4 * kWave), lds_bytes, stream, P, q, qd, ydd, tau, B, scratch);
    else if (n_waves == 2) hipLaunchKernelGGL((rnea_chain_lm_kernel<T, 2>), dim3(grid), dim3(2 * kWave), lds_bytes, stream, P, q, qd, ydd, tau, B, scratch);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
template hipError_t launch_rnea_chain_lm<float>(const RneaChainDev<float> &, const float *, const float *, const float *, float *, size_t, float *, int, size_t,
                                                hipStream_t, int);
template hipError_t launch_rnea_chain_lm<double>(const RneaChainDev<double> &, const double *, const double *, const double *, double *, size_t, double *, int,
                                                 size_t, hipStream_t, int);
#endif

template <class T>
hipError_t launch_rnea_chain_gen(const RneaChainDev<T> &P, const T *q, const T *qd, const T *ydd, T *tau, size_t B, T *scratch, int grid,
                                 size_t lds_bytes, hipStream_t stream);
#if GRBDA_CHAIN_UNIT == 2
template <class T>
hipError_t launch_rnea_chain_gen(const RneaChainDev<T> &P, const T *q, const T *qd, const T *ydd, T *tau, size_t B, T *scratch, int grid,
                                 size_t lds_bytes, hipStream_t stream)
{
    if (P.n_glb_slots > 0) hipLaunchKernelGGL((rnea_chain_kernel<T, 2, true>), dim3(grid), dim3(kWave), lds_bytes, stream, P, q, qd, ydd, tau, B, scratch);
    else hipLaunchKernelGGL((rnea_chain_kernel<T, 2, false>), dim3(grid), dim3(kWave), lds_bytes, stream, P, q, qd, ydd, tau, B, scratch);
    return hipGetLastError();
}
template hipError_t launch_rnea_chain_gen<float>(const RneaChainDev<float> &, const float *, const float *, const float *, float *, size_t,
                                                 float *, int, size_t, hipStream_t);
template hipError_t launch_rnea_chain_gen<double>(const RneaChainDev<double> &, const double *, const double *, const double *, double *,
                                                  size_t, double *, int, size_t, hipStream_t);

// Single-cluster programs (RneaChainProgram::single_gen): the inverse dynamics of ONE generic cluster on the ground, the counterpart of
// aba_gen1_kernel -- TreeModel::recursiveNewtonEulerAlgorithm (src/Dynamics/TreeModel.cpp:173-212) of a model whose only cluster is a
// loop mechanism (four_bar.urdf, six_bar.urdf).  Specialised on (n, implicit?) at compile time; no slab, no segment loop; the tile loop
// and the input staging of aba_gen1_kernel (gen1_tiles).  ONE LDS object (plan.cpp: [sin, cos][forces | the constraint's scratch]
// [kept block]) keeps the work area at the forward dynamics' size.
template <class T, int N, bool LOOP, int WPS>
__global__ __launch_bounds__(kWave, WPS) void rnea_gen1_kernel(RneaChainDev<T> DP, const T *__restrict__ q, const T *__restrict__ qd,
                                                             const T *__restrict__ ydd, T *__restrict__ tau, size_t B)
{
    RneaTables<T> P;
    P.segs = nullptr;
    P.links = nullptr;
    P.pairs = nullptr;
    P.frees = nullptr;
    P.diffs = nullptr;
    P.gens = (cptr<ChainGen>)DP.gens;
    P.gbodies = (cptr<ChainGenBody>)DP.gbodies;
    P.cints = (cptr<int32_t>)DP.cints;
    P.consts = (cptr<T>)DP.consts;
    P.n_segs = 0;
    P.nq = DP.nq;
    P.nv = DP.nv;
    P.ori_repr = DP.ori_repr;
#pragma unroll
    for (int i = 0; i < 6; i++) P.a_root[i] = DP.a_root[i];
    const ChainGen g = load_rec(P.gens);
    gen1_tiles<T, N, LOOP>(DP.lds_bytes, P.nq, q, qd, ydd, tau, B, nullptr, [&](const ChainMemC<T> &M) {
        gen_rnea_fwd<T, N, LOOP>(P, M, g);
        gen_rnea_bwd<T, N, LOOP, false>(P, M, g);
    });
}

template <class T>
int rnea_gen1_waves_per_simd(int n)
{
    (void)n;
    return sizeof(T) == 4 ? 4 : 2;  // (the LDS of a CU holds fewer than that for every cluster with a constraint)
}
template int rnea_gen1_waves_per_simd<float>(int);
template int rnea_gen1_waves_per_simd<double>(int);
template <class T, int N, bool LOOP>
static hipError_t launch_rgen1(const RneaChainDev<T> &P, const T *q, const T *qd, const T *ydd, T *tau, size_t B, int grid, size_t lds_bytes,
                               hipStream_t stream)
{
    // wavefronts per SIMD the register footprint allows (checked against the code object: no scratch)
    constexpr int WPS = sizeof(T) == 4 ? 4 : 2;
    hipLaunchKernelGGL((rnea_gen1_kernel<T, N, LOOP, WPS>), dim3(grid), dim3(kWave), lds_bytes, stream, P, q, qd, ydd, tau, B);
    return hipGetLastError();
}
template <class T>
hipError_t launch_rnea_gen1(const RneaChainDev<T> &P, int n, int implicit, const T *q, const T *qd, const T *ydd, T *tau, size_t B, int grid,
                            size_t lds_bytes, hipStream_t stream)
{
    if (implicit) {
        if (n == 1) return launch_rgen1<T, 1, true>(P, q, qd, ydd, tau, B, grid, lds_bytes, stream);
        if (n == 2) return launch_rgen1<T, 2, true>(P, q, qd, ydd, tau, B, grid, lds_bytes, stream);
        if (n == 3) return launch_rgen1<T, 3, true>(P, q, qd, ydd, tau, B, grid, lds_bytes, stream);
        return launch_rgen1<T, 4, true>(P, q, qd, ydd, tau, B, grid, lds_bytes, stream);
    }
    if (n == 1) return launch_rgen1<T, 1, false>(P, q, qd, ydd, tau, B, grid, lds_bytes, stream);
    if (n == 2) return launch_rgen1<T, 2, false>(P, q, qd, ydd, tau, B, grid, lds_bytes, stream);
    if (n == 3) return launch_rgen1<T, 3, false>(P, q, qd, ydd, tau, B, grid, lds_bytes, stream);
    return launch_rgen1<T, 4, false>(P, q, qd, ydd, tau, B, grid, lds_bytes, stream);
}
template hipError_t launch_rnea_gen1<float>(const RneaChainDev<float> &, int, int, const float *, const float *, const float *, float *, size_t,
                                            int, size_t, hipStream_t);
template hipError_t launch_rnea_gen1<double>(const RneaChainDev<double> &, int, int, const double *, const double *, const double *, double *,
                                             size_t, int, size_t, hipStream_t);
#endif
template <class T>
hipError_t launch_rnea_chain(const RneaChainDev<T> &P, const T *q, const T *qd, const T *ydd, T *tau, size_t B, T *scratch, int grid,
                             size_t lds_bytes, hipStream_t stream)
{
    const bool glb = P.n_glb_slots > 0;
    if (P.n_gens > 0) return launch_rnea_chain_gen<T>(P, q, qd, ydd, tau, B, scratch, grid, lds_bytes, stream);  // (unit 2)
    if (P.n_diffs > 0 && glb) hipLaunchKernelGGL((rnea_chain_kernel<T, 1, true>), dim3(grid), dim3(kWave), lds_bytes, stream, P, q, qd, ydd, tau, B, scratch);
    else if (P.n_diffs > 0) hipLaunchKernelGGL((rnea_chain_kernel<T, 1, false>), dim3(grid), dim3(kWave), lds_bytes, stream, P, q, qd, ydd, tau, B, scratch);
    else if (glb) hipLaunchKernelGGL((rnea_chain_kernel<T, 0, true>), dim3(grid), dim3(kWave), lds_bytes, stream, P, q, qd, ydd, tau, B, scratch);
    else hipLaunchKernelGGL((rnea_chain_kernel<T, 0, false>), dim3(grid), dim3(kWave), lds_bytes, stream, P, q, qd, ydd, tau, B, scratch);
    return hipGetLastError();
}
#if GRBDA_CHAIN_UNIT == 0
template hipError_t launch_rnea_chain<float>(const RneaChainDev<float> &, const float *, const float *, const float *, float *, size_t,
                                             float *, int, size_t, hipStream_t);
template hipError_t launch_rnea_chain<double>(const RneaChainDev<double> &, const double *, const double *, const double *, double *,
                                              size_t, double *, int, size_t, hipStream_t);
#endif

#if defined(GRBDA_CHAIN_PROFILE) && GRBDA_CHAIN_UNIT == 0
extern "C" int grbda_debug_chain_profile(unsigned long long *out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(grbda_chain_prof), sizeof(unsigned long long) * 128) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[128] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(grbda_chain_prof), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif

static hipError_t set_max_dynamic_lds(const void *const *kernels, int n)
{
    for (int i = 0; i < n; i++) {
        const hipError_t e = hipFuncSetAttribute(kernels[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
hipError_t set_max_dynamic_lds_chain_unit1();
hipError_t set_max_dynamic_lds_chain_unit2();
hipError_t set_max_dynamic_lds_chain_unit3();
#if GRBDA_CHAIN_UNIT == 0
hipError_t set_max_dynamic_lds_chain()
{
    const void *const kernels[] = {
        reinterpret_cast<const void *>(&aba_chain_kernel<float, 2, 0>), reinterpret_cast<const void *>(&aba_chain_kernel<float, 2, 1>),
        reinterpret_cast<const void *>(&aba_chain_kernel<float, 4, 0>),
        reinterpret_cast<const void *>(&aba_chain_kernel<double, 2, 0>),
        reinterpret_cast<const void *>(&rnea_chain_kernel<float, 0, false>), reinterpret_cast<const void *>(&rnea_chain_kernel<float, 1, false>),
        reinterpret_cast<const void *>(&rnea_chain_kernel<double, 0, false>), reinterpret_cast<const void *>(&rnea_chain_kernel<double, 1, false>),
        reinterpret_cast<const void *>(&rnea_chain_kernel<float, 0, true>), reinterpret_cast<const void *>(&rnea_chain_kernel<float, 1, true>),
        reinterpret_cast<const void *>(&rnea_chain_kernel<double, 0, true>), reinterpret_cast<const void *>(&rnea_chain_kernel<double, 1, true>),
        reinterpret_cast<const void *>(&osim_chain_kernel<float>), reinterpret_cast<const void *>(&osim_chain_kernel<double>)};
    const hipError_t e = set_max_dynamic_lds(kernels, static_cast<int>(sizeof(kernels) / sizeof(kernels[0])));
    if (e != hipSuccess) return e;
    const hipError_t e1 = set_max_dynamic_lds_chain_unit1();
    if (e1 != hipSuccess) return e1;
    const hipError_t e2 = set_max_dynamic_lds_chain_unit2();
    return e2 != hipSuccess ? e2 : set_max_dynamic_lds_chain_unit3();
}
#elif GRBDA_CHAIN_UNIT == 1
hipError_t set_max_dynamic_lds_chain_unit1()
{
    const void *const kernels[] = {reinterpret_cast<const void *>(&aba_chain_kernel<double, 2, 1>),
                                   reinterpret_cast<const void *>(&aba_chain_lm_kernel<double, 2>)};
    return set_max_dynamic_lds(kernels, 2);
}
#elif GRBDA_CHAIN_UNIT == 3
hipError_t set_max_dynamic_lds_chain_unit3()
{
    const void *const kernels[] = {reinterpret_cast<const void *>(&aba_chain_lm_kernel<float, 2>), reinterpret_cast<const void *>(&aba_chain_lm_kernel<float, 4>),
                                   reinterpret_cast<const void *>(&aba_chain_lm_kernel<float, 2, true>), reinterpret_cast<const void *>(&aba_chain_lm_kernel<float, 4, true>),
                                   reinterpret_cast<const void *>(&aba_chain_lm_kernel<double, 4>),
                                   reinterpret_cast<const void *>(&rnea_chain_lm_kernel<float, 2>), reinterpret_cast<const void *>(&rnea_chain_lm_kernel<float, 4>),
                                   reinterpret_cast<const void *>(&rnea_chain_lm_kernel<double, 2>), reinterpret_cast<const void *>(&rnea_chain_lm_kernel<double, 4>),
                                   reinterpret_cast<const void *>(&rnea_chain_lm_kernel<float, 2, true>), reinterpret_cast<const void *>(&rnea_chain_lm_kernel<float, 4, true>)};
    return set_max_dynamic_lds(kernels, 11);
}
#else
hipError_t set_max_dynamic_lds_chain_unit2()
{
    const void *const kernels[] = {reinterpret_cast<const void *>(&aba_chain_kernel<float, 2, 2>),
                                   reinterpret_cast<const void *>(&aba_chain_kernel<double, 2, 2>),
                                   reinterpret_cast<const void *>(&rnea_chain_kernel<float, 2, false>),
                                   reinterpret_cast<const void *>(&rnea_chain_kernel<float, 2, true>),
                                   reinterpret_cast<const void *>(&rnea_chain_kernel<double, 2, false>),
                                   reinterpret_cast<const void *>(&rnea_chain_kernel<double, 2, true>)};
    return set_max_dynamic_lds(kernels, 6);  // (the single-cluster kernels stay below the 64 KiB default: capi.cpp)
}
#endif

}  // namespace grbda_hip
