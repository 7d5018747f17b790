// devplan.h -- kernel argument block and launcher declarations shared by kernels.hip and capi.cpp
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>

#include "plan.h"

namespace grbda_hip {

// wavefronts per SIMD of the "wide" fp32 chain kernel (GRBDA_CHAIN_WIDE=1; experiment builds: -DGRBDA_EXP_WIDE_WPS=3)
#ifndef GRBDA_EXP_WIDE_WPS
#define GRBDA_EXP_WIDE_WPS 4
#endif
constexpr int kChainWideWps = GRBDA_EXP_WIDE_WPS;
// wavefronts (one-wavefront workgroups) whose dynamic LDS fits a CU: gfx950 hands out its 160 KiB in granules of 1 280 bytes, so a
// kernel asking for 13 312 bytes holds 11 workgroups per CU, not 12 -- and a persistent grid sized for 12 leaves every twelfth
// workgroup waiting for a slot until another has finished ALL its tiles (measured: four_bar forward dynamics 0.103 ms at 12 per CU,
// 0.074 ms at 10; profiles/r5_single_cluster_kernels.txt)
inline size_t lds_workgroups_per_cu(size_t lds_bytes)
{
    if (lds_bytes == 0) return 32;
    static const size_t granule = [] {  // (GRBDA_LDS_GRANULE=1: the naive quotient, for A/B runs)
        const char *e = std::getenv("GRBDA_LDS_GRANULE");
        const long v = e ? std::atol(e) : 0;
        return static_cast<size_t>(v > 0 ? v : 1280);
    }();
    return (160u * 1024u) / ((lds_bytes + granule - 1) / granule * granule);
}
constexpr int kChainWideLdsBytes = (160 * 1024 / (4 * kChainWideWps)) / 256 * 256;  // per wavefront, whole rows of 64 floats


template <class T>
struct DevPlan {
    const Step *steps;
    int n_steps;
    const ClusterRec *clusters;
    const BodyRec *bodies;
    const T *consts;
    const int32_t *cints;  // integer payload of implicit constraints
    const int32_t *acc_k;  // ABA: per step, K block to prefetch (Layout::acc_k)
    int nq, nv;
    int n_lds_slots, n_glb_slots;
    int lds_bytes;  // dynamic LDS actually allocated per wave (slot store / input staging area)
    int ori_repr;
    int general;   // launch the general kernel variant: implicit-loop clusters and / or external forces
    int split;     // the layout keeps [K | y0] blocks in the global slab and everything else in LDS (Slots<T, true>)
    int n_bodies;
    const T *fext; // [B][n_bodies][6] world-frame spatial forces or nullptr (TreeModel::setExternalForces)
    unsigned long long *bad_count;  // aba_kernel: per-device counter of states with a pivot of D that is not positive (grbda_spd_bad_pivots)
    T a_root[6];  // -gravity (ClusterTreeDynamics.cpp:147)
};

template <class T>
hipError_t launch_aba(const DevPlan<T> &P, const T *q, const T *qd, const T *tau, T *ydd, size_t B, T *scratch,
                      int grid, size_t lds_bytes, hipStream_t stream, bool two_waves_per_simd);
template <class T>
hipError_t launch_rnea(const DevPlan<T> &P, const T *q, const T *qd, const T *ydd, T *tau, size_t B, T *scratch,
                       int grid, size_t lds_bytes, hipStream_t stream);
template <class T>
hipError_t launch_project(const DevPlan<T> &P, int n_clusters, T *q, int32_t *ok, size_t B, int max_iter, T tol,
                          T *scratch, int grid, size_t lds_bytes, hipStream_t stream);
// bit c of pos / vel: cluster c's positions / velocities are given as SPANNING coordinates (state_kernel)
constexpr int kStateFlagWords = 4;
struct StateFlags {
    uint64_t pos[kStateFlagWords], vel[kStateFlagWords];
};
template <class T>
hipError_t launch_state(const DevPlan<T> &P, int n_clusters, const StateFlags &F, const T *q_in, const T *qd_in, int in_nq, int in_nv,
                        T *q_out, T *qd_out, int32_t *status, T *gmax, size_t B, T tol, T *scratch, int grid, size_t lds_bytes,
                        hipStream_t stream);
template <class T>
hipError_t launch_spanning(const DevPlan<T> &P, int n_clusters, int n_span, const T *q, const T *qd, const T *ydd,
                           T *qd_span, T *qdd_span, size_t B, T *scratch, int grid, size_t lds_bytes,
                           hipStream_t stream);
template <class T>
hipError_t launch_poses(const DevPlan<T> &P, int n_clusters, const T *q, T *Xa, size_t B, int grid, hipStream_t stream);
template <class T>
hipError_t launch_twists(const DevPlan<T> &P, int n_clusters, int n_span, const T *q, const T *qd_span, const T *qdd_span, T *V, size_t B,
                         int grid, hipStream_t stream);
hipError_t set_max_dynamic_lds();

// kernel argument block of the chain-structured fast path (chain_kernels.hip; plan.h, ChainProgram)
template <class T>
struct ChainDev {
    const ChainSeg *segs;
    const ChainLink *links;
    const ChainPair *pairs;
    const ChainFree *frees;
    const ChainDiff *diffs;
    int n_diffs;
    const ChainGen *gens;          // generic clusters (plan.h, ChainGen; gen_segments.h)
    const ChainGenBody *gbodies;
    int n_gens;
    const int32_t *cints;
    const T *consts;
    int n_segs;
    int nq, nv;
    int n_glb_slots;
    int lds_bytes;
    int ori_repr;
    int debug;   // GRBDA_CHAIN_DEBUG: phase ablation for profiling (chain_kernels.hip)
    int sv_global;  // ChainProgram::sv_global
    int out_lds;    // ChainProgram::out_lds
    // aba_chain_kernel only (capi.cpp, run_chain): bit 0 -- the program starts with the floating base's forward segment, whose whole work is
    // "own velocity to LDS slot stage_lds_v": the tile prologue does it from the staged block (no slab round trip) and the segment is
    // skipped; bit 1 -- the base's acceleration segment follows its backward segment directly: the backward segment finishes it with y0
    // in registers (no slab round trip) and the acceleration segment is skipped
    int fuse, stage_lds_v, stage_v_index;
    // per-device counter of states whose D = S^T IA S had a pivot that is not positive (grbda_spd_bad_pivots; deriv_kernels.hip owns the word)
    unsigned long long *bad_count;
    T a_root[6];
};
template <class T>
hipError_t launch_aba_chain(const ChainDev<T> &P, const T *q, const T *qd, const T *tau, T *ydd, size_t B, T *scratch, int grid,
                            size_t lds_bytes, hipStream_t stream, bool four_waves_per_simd);
// latency mode: a tile per workgroup of n_waves = 2 (or, fp32, 4) wavefronts (chain_kernels.hip, aba_chain_lm_kernel)
template <class T>
hipError_t launch_aba_chain_lm(const ChainDev<T> &P, const T *q, const T *qd, const T *tau, T *ydd, size_t B, T *scratch, int grid,
                               size_t lds_bytes, hipStream_t stream, int n_waves);
hipError_t set_max_dynamic_lds_chain();
// single-cluster programs (ChainProgram::single_gen; chain_kernels.hip, aba_gen1_kernel): P.lds_bytes = the work area, lds_bytes = work
// area + nq + 2 nv rows of staged inputs
template <class T>
hipError_t launch_aba_gen1(const ChainDev<T> &P, int n, int implicit, const T *q, const T *qd, const T *tau, T *ydd, size_t B, int grid,
                           size_t lds_bytes, hipStream_t stream);
template <class T>
int gen1_waves_per_simd(int n);

template <class T>
struct RneaChainDev {
    const RneaSeg *segs;
    const RneaLink *links;
    const RneaPair *pairs;
    const RneaFree *frees;
    const RneaDiff *diffs;
    int n_diffs;
    const ChainGen *gens;          // generic clusters (gen_rnea_segments.h)
    const ChainGenBody *gbodies;
    int n_gens;
    const int32_t *cints;
    const T *consts;
    int n_segs;
    int nq, nv;
    int n_glb_slots;  // RneaChainProgram::n_glb
    int lds_bytes;
    int ori_repr;
    T a_root[6];
};
template <class T>
hipError_t launch_rnea_chain(const RneaChainDev<T> &P, const T *q, const T *qd, const T *ydd, T *tau, size_t B, T *scratch, int grid,
                             size_t lds_bytes, hipStream_t stream);
// latency mode: a tile per workgroup of n_waves = 2 or 4 wavefronts (chain_kernels.hip, rnea_chain_lm_kernel)
template <class T>
hipError_t launch_rnea_chain_lm(const RneaChainDev<T> &P, const T *q, const T *qd, const T *ydd, T *tau, size_t B, T *scratch, int grid, size_t lds_bytes,
                                hipStream_t stream, int n_waves);

// single-cluster programs (RneaChainProgram::single_gen; chain_kernels.hip, rnea_gen1_kernel): P.lds_bytes = the work area, lds_bytes =
// work area + nq + 2 nv rows of staged inputs
template <class T>
hipError_t launch_rnea_gen1(const RneaChainDev<T> &P, int n, int implicit, const T *q, const T *qd, const T *ydd, T *tau, size_t B, int grid,
                            size_t lds_bytes, hipStream_t stream);
template <class T>
int rnea_gen1_waves_per_simd(int n);

// inverse operational-space inertia by force propagation along the contacts' ancestor paths (chain_kernels.hip,
// osim_chain_kernel).  Built per call on the host (capi.cpp) and passed by value.
constexpr int kOsimMaxContacts = 8;
constexpr int kOsimMaxPath = 12;  // clusters between a contact body and the root
enum OsimStepKind : int32_t { OSIM_LINK = 0, OSIM_PAIR_LINK1 = 1, OSIM_PAIR_LINK2 = 2, OSIM_FREE = 3, OSIM_DIFF_LINK1 = 4, OSIM_DIFF_LINK2 = 5 };
struct OsimStep {
    int16_t kind;
    int16_t rec;      // index into links[] / pairs[] / frees[]
    int16_t v_index;  // first velocity coordinate of the cluster
    int16_t w_row;    // first of the cluster's rows in the contact's W block (n rows)
};
template <class T>
struct OsimArgs {
    int n_contacts;
    int want_J;
    int path_len[kOsimMaxContacts];
    int n_rows[kOsimMaxContacts];                        // rows of W of each contact (sum of n over its path)
    int common[kOsimMaxContacts][kOsimMaxContacts];      // number of trailing W rows two contacts share (their common ancestors)
    OsimStep path[kOsimMaxContacts][kOsimMaxPath];       // leaf side first
    T K0[kOsimMaxContacts][36];                          // wrench on the contact body (plan frame) per unit contact wrench, row-major
    int w_base;                                          // first slab row (after the chain program's rows) of the W blocks
    int w_stride;                                        // rows per contact
    // applyTestForce mode (one contact): a force per state, given in world coordinates at the contact point; results
    // lambda_inv[B] = f^T (J H^-1 J^T) f and dstate[B][nv] = H^-1 J^T f.  perm: plan-frame axis i of the contact body is the
    // reference's axis perm[i] (canonical joint axes, plan.cpp)
    int test_force;
    int perm[3];
    const T *force;
    T *lambda_inv;
    T *dstate;
};
template <class T>
hipError_t launch_osim_chain(const ChainDev<T> &P, const OsimArgs<T> &A, const T *q, const T *zeros, T *Linv, T *J, size_t B,
                             T *scratch, int grid, size_t lds_bytes, hipStream_t stream);

// composite-rigid-body algorithm (crba_kernels.hip)
template <class T>
hipError_t launch_crba(const DevPlan<T> &P, const CrbaBody *cb, int n_clusters, int n_rows, const T *q, T *H, size_t B, T *scratch,
                       int grid, hipStream_t stream, bool packed, int interleave);
// (in place; B a multiple of il; il nv (nv + 1) / 2 elements of LDS: unpack_symmetric_lds_bytes <= 60 KiB)
template <class T>
hipError_t launch_unpack_symmetric(T *H, const uint64_t *related, int nv, size_t B, int grid, hipStream_t stream, int il);
size_t unpack_symmetric_lds_bytes(int nv, size_t elem, int il);

// inverse-dynamics derivatives and the batched SPD solve behind d ydd / d (q, qd, tau) (deriv_kernels.hip)
// states per group of the interleaved derivative workspace ([group][entry][kDerivGroup]) = wavefronts per workgroup of the
// matrix-core solve (deriv_kernels.hip)
constexpr int kDerivGroup = 4;
template <class T>
hipError_t launch_rnea_deriv(const DevPlan<T> &P, const DerivBody *db, int n_clusters, int n_rows, int n_max, const T *q, const T *qd,
                             const T *ydd, T *Dq, T *Dqd, T *H, size_t B, T *scratch, int grid, hipStream_t stream, int interleave);
template <class TIO, class TC>
hipError_t launch_spd_solve(const TIO *H, int h_packed, const TIO *P1, const TIO *P2, TIO *Hinv, TIO *X1, TIO *X2, const uint64_t *related,
                            int nv, size_t B, int grid, hipStream_t stream, int interleave);
size_t spd_solve_lds_bytes(int nv, size_t elem, int n_rhs);

// analytic derivatives of models with implicit clusters through the spanning tree (manifold_kernels.hip)
template <class T>
hipError_t launch_manifold_constraint(const DevPlan<T> &P, int n_clusters, const int32_t *span_q, const int32_t *span_v, const int32_t *crow,
                                      int nq_s, int nv_s, int n_cpl_rows, int want_d, const T *q, const T *qd, const T *ydd, T *q_s, T *qd_s,
                                      T *qdd_s, T *cpl, size_t B, int grid, hipStream_t stream, int shape = false, bool trig = true);
template <class T>
hipError_t launch_manifold_apply(const DevPlan<T> &P, int n_clusters, const int32_t *span_v, const int32_t *crow, int nv_s, int n_cpl_rows, int mode,
                                 const T *x_s, const T *tau, const T *Hinv, const T *cpl, T *out, size_t B, int grid, hipStream_t stream,
                                 bool big = false);
template <class T>
hipError_t launch_manifold_project(const DevPlan<T> &P, int n_clusters, const int32_t *span_v, const int32_t *crow, const uint64_t *rel,
                                   const uint64_t *rel_s, int nv_s, int n_cpl_rows, int mode, const T *Aq, const T *Av, const T *Hs,
                                   const T *tau_s, const T *cpl, T *Dq, T *Dqd, T *H, size_t B, int grid, hipStream_t stream, int interleave,
                                   bool big = false);
// (plans with clusters beyond the structured limits: H only, one-word masks or nv x nv tables; the solve for up to 128 velocities)
template <class T>
hipError_t launch_manifold_project_wide(const DevPlan<T> &P, int n_clusters, const int32_t *span_v, const int32_t *crow, const uint64_t *rel,
                                        const uint64_t *rel_s, const int32_t *relt, const int32_t *relt_s, int nv_s, int n_cpl_rows, const T *Hs,
                                        const T *cpl, T *H, size_t B, int grid, hipStream_t stream);
template <class T>
hipError_t launch_manifold_state(const DevPlan<T> &P, int n_clusters, const StateFlags &F, const T *q_in, const T *qd_in, int in_nq, int in_nv,
                                 T *q_out, T *qd_out, int32_t *status, T *cond, size_t B, T tol, int grid, hipStream_t stream);
template <class T>
hipError_t launch_manifold_newton(const DevPlan<T> &P, int n_clusters, T *q, int32_t *ok, size_t B, int max_iter, T tol, int grid, hipStream_t stream);
template <class T>
hipError_t launch_spd_wide_solve(const T *H, const int32_t *relt, const T *rhs, T *out, int nv, size_t B, int n_cu, hipStream_t stream,
                                 unsigned long long *bad_count);
// device address of the bad-pivot counter of the SPD solves (deriv_kernels.hip; nullptr when it cannot be resolved): the solve of the wide
// route (manifold_kernels.hip, another translation unit) counts into the same word
unsigned long long *spd_bad_count_address();
int spd_mfma_workgroups_per_cu(int nv);
bool spd_solve_on_mfma(size_t elem, int nv, int n_rhs);
hipError_t set_max_dynamic_lds_deriv();
// H^-1 = W^T W from the articulated-body quantities (minv_kernels.hip; plan.h, MinvProgram): the record blocks (one state per lane, the
// slab rows and processing order of rnea_deriv_kernel), then the walk + the two products on the matrix cores (one state per wavefront;
// records and right-hand sides interleaved by groups of kDerivGroup states)
template <class T>
hipError_t launch_abi_factor(const DevPlan<T> &P, const DerivBody *db, const MinvBody *mb, int n_clusters, int n_rows, int n_max, int n_entries,
                             const T *q, T *rec, size_t B, T *scratch, int grid, hipStream_t stream, int interleave,
                             unsigned long long *bad_count);
template <class T>
hipError_t launch_minv_solve(const T *rec, int n_entries, int r_il, const int32_t *coltab, int max_depth, int base_off, int n_max, const T *P1,
                             const T *P2, int p_il, T *Hinv, T *X1, T *X2, const uint64_t *related, int nv, size_t B, int grid,
                             hipStream_t stream);
size_t minv_solve_lds_bytes(int nv, int n_rhs, int n_entries, size_t elem);
template <class T>
int minv_workgroups_per_cu(int nv, int n_max, int n_rhs, int n_entries);  // (registers of the instantiation and LDS granules; needs the current device)
hipError_t set_max_dynamic_lds_minv();
hipError_t spd_bad_pivots(unsigned long long *count, int reset);

}  // namespace grbda_hip
