// capi.cpp -- implementation of include/grbda_hip.h on top of plan.cpp and kernels.hip.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/grbda_hip.h"
#include "../../include/grbda_model_desc.h"
#include "devplan.h"

namespace grbda_hip {
int urdf_to_blob(const char *const *paths, int n_paths, int ori_repr, std::vector<unsigned char> &blob,
                 std::string &err);

}  // namespace grbda_hip

using namespace grbda_hip;

namespace {

thread_local std::string g_last_error;

int set_err(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}
int hip_err(hipError_t e, const char *what)
{
    return set_err(GRBDA_EHIP, std::string(what) + ": " + hipGetErrorString(e));
}

// per-(device) copies of the plan tables; per-(device, stream) scratch slabs
struct DeviceTables {
    Step *aba_steps = nullptr, *rnea_steps = nullptr;
    // [0] f32, [1] f64, [2] f32 + external forces, [3] f64 + external forces, [4] f32 split layout
    ClusterRec *clusters[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    ClusterRec *rnea_clusters[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int32_t *cints = nullptr;
    int32_t *acc_k[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int32_t *dq_map = nullptr;  // per velocity index: (kind, position index, component), see grbda_fd_dq_*
    BodyRec *bodies[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};       // ABA slots
    BodyRec *rnea_bodies[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // RNEA slots
    double *consts64 = nullptr;
    float *consts32 = nullptr;
    // chain program of the f32 fast path (plan.h, ChainProgram)
    // [0] f32, two wavefronts per SIMD (HostPlan::chain32), [1] f32, four (chain32w), [2] f64 (chain64)
    // chain programs: 0 f32, 1 f32 at four wavefronts per SIMD, 2 f64, 3 / 4 latency mode f32 / f64 (ChainProgram::n_waves = 2),
    // 5 / 6 latency mode f32 / f64 with four wavefronts per tile
    ChainSeg *chain_segs[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    ChainLink *chain_links[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    ChainPair *chain_pairs[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    ChainFree *chain_frees[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    ChainDiff *chain_diffs[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    ChainGen *chain_gens[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    ChainGenBody *chain_gbodies[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    CrbaBody *crba_bodies = nullptr;
    DerivBody *deriv_bodies = nullptr;
    uint64_t *deriv_related = nullptr;  // DerivProgram::related
    MinvBody *minv_bodies = nullptr;    // DerivProgram::minv (plan.h, MinvProgram): record offsets per body ...
    int32_t *minv_coltab = nullptr;     // ... and the column programs of minv_mfma_kernel
    int32_t *related_table = nullptr;   // HostPlan::related_table (plans of the wide route with more than 64 velocities)
    int32_t *span_q = nullptr, *span_v = nullptr, *crow = nullptr;  // grbda_plan::span_q / span_v / crow
    // inverse dynamics on the chains: [0] f32 (HostPlan::rchain32), [1] f64, [2] f32 at four wavefronts per SIMD (rchain32w),
    // [3] / [4] latency mode f32 / f64 with two wavefronts per tile (rchain32p / rchain64p), [5] / [6] with four (rchain32q / rchain64q)
    RneaSeg *rchain_segs[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    RneaLink *rchain_links[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    RneaPair *rchain_pairs[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    RneaFree *rchain_frees[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    RneaDiff *rchain_diffs[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    ChainGen *rchain_gens[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    ChainGenBody *rchain_gbodies[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int n_cu = 0;
    unsigned long long *bad_count = nullptr;  // this device's counter of states with a pivot that is not positive (deriv_kernels.hip)
};
struct Scratch {
    void *ptr = nullptr;
    size_t bytes = 0;
};

// every entry point that may switch the calling thread's HIP device puts it back on return (torch and other users of the
// runtime in the same thread keep their current device)
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define GRBDA_CALL_SCOPE(p) DeviceGuard device_guard_; std::unique_lock<std::recursive_mutex> plan_lock_((p)->mu)

int env_int(const char *name, int dflt)
{
    const char *v = std::getenv(name);
    return v && *v ? std::atoi(v) : dflt;
}

constexpr int kLayouts = 5;
const Layout &layout_of(const HostPlan &h, int w)
{
    return w == 0 ? h.lay32 : (w == 1 ? h.lay64 : (w == 2 ? h.lay32x : (w == 3 ? h.lay64x : h.lay32s)));
}

}  // namespace

struct grbda_plan {
    HostPlan host;
    std::vector<unsigned char> blob;
    // Held from the moment a call sizes its scratch / work buffers until its last kernel is ENQUEUED: a second thread
    // that needs a bigger buffer for the same (device, stream) can then only free the old one after the first thread's
    // launches are in the stream (hipFree waits for them).  Recursive: the derived entry points call run().
    mutable std::recursive_mutex mu;
    mutable std::map<int, DeviceTables> dev;
    mutable std::map<std::pair<int, void *>, Scratch> scratch;
    mutable std::map<std::pair<int, void *>, Scratch> work;  // expanded batches of the derived quantities
    mutable std::map<std::pair<int, void *>, Scratch> work_cvt;  // fp64 copies of fp32 inputs (grbda_fd_dq_f32)
    mutable std::map<std::pair<int, void *>, Scratch> work_proj; // projection_run (called from inside the users of `work`)
    // launch shape per kernel, index = (rnea ? 2 : 0) + (f64 ? 1 : 0): LDS budget per wavefront for the
    // slot store, and wavefronts launched per CU (the grid is persistent)
    // (defaults from sweeps on MI355X over the MIT humanoid, Mini Cheetah and JVRC-1 at 4096 tiles: the f32
    // RNEA kernel needs ~100 VGPRs and gains from 16 wavefronts per CU with 10 KiB each; the f64 RNEA kernel is
    // register-bound to 8 per CU and prefers 6 with more LDS)
    int lds_bytes_per_wave[4] = {20480, 20480, 10240, 26624};
    int waves_per_cu[4] = {8, 8, 16, 6};
    // fp64 forward dynamics OFF the plain chain kernels -- the cluster interpreter, and chain programs with generic segments (one wavefront
    // per SIMD, 428 registers, chain_kernels.hip) -- runs best at one wavefront per SIMD (measured on the zoo, 131 072 states: interpreter
    // 0.71-0.94 -> 0.46-0.66 ms; generic chain kernel even); GRBDA_WAVES_PER_CU / _ABA64 set it like the others
    int waves_per_cu_f64_wide_regs = 4;
    bool no_split = false;
    bool no_chain = false;  // GRBDA_NO_CHAIN=1: keep the general interpreter (A/B runs, tests of the general kernels)
    int chain_debug = 0;
    bool no_crba = false;
    bool rnea_narrow = false;  // GRBDA_RNEA_NARROW=1: the inverse-dynamics chain kernel stays at two wavefronts per SIMD
    bool no_analytic = false;  // GRBDA_NO_ANALYTIC=1: derivatives by the unit-vector / central-difference batches only
    bool solve_f64 = false;    // GRBDA_SOLVE_F64=1: the SPD solve of the f32 derivative entry points computes in f64
    int gen1_tiles_per_wave = 0;   // GRBDA_GEN1_TILES_PER_WAVE > 0: grid = tiles / this (the dispatcher balances the workgroups)
    int gen1_waves_cap = 0;        // GRBDA_GEN1_WAVES_PER_CU > 0: wavefronts per CU of the single-cluster kernels (experiments)
    bool no_latency_mode = false;  // GRBDA_NO_LATENCY_MODE=1: small batches keep the one-wavefront-per-tile kernel
    int lm_waves = 0;              // GRBDA_LM_WAVES=2: latency mode never takes four wavefronts per tile (A/B runs)
    int crba_waves = 16;       // GRBDA_CRBA_WAVES_PER_CU: grid of the composite-rigid-body kernel (fp32: 99 registers, four wavefronts per SIMD:
                               // JVRC-1 mass matrix 1.95 -> 1.81 ms per 262 144 states against eight per CU; fp64 is capped at eight)
    int deriv_waves = 0;       // GRBDA_DERIV_WAVES_PER_CU: grid of the inverse-dynamics derivative kernel (0: 3)
    bool no_minv = false;      // GRBDA_NO_MINV=1: the derivative pipeline keeps the dense factorisation of H (A/B runs)
    int minv_wpc = 0;          // GRBDA_MINV_WPC: upper limit of the workgroups per CU of minv_mfma_kernel (0: what registers and LDS hold)
    bool no_efpa = false;  // GRBDA_NO_EFPA=1: inverse OSIM through unit wrenches and the ABA / RNEA kernels  // GRBDA_NO_CRBA=1: mass matrix through nv + 1 inverse-dynamics evaluations (the path of loop models)
    bool chain_wide = false;  // GRBDA_CHAIN_WIDE=1: chain kernel at four wavefronts per SIMD for batches that fill them
    // Models with implicit clusters: the spanning-tree model as a plan of its own (plan.cpp, make_spanning_blob) -- the analytic
    // derivatives and the mass matrix are taken on it and projected with the per-state G (manifold_kernels.hip).  span_q / span_v:
    // spanning position / velocity index of every body; crow: first row of every implicit cluster in the coupling slab.
    grbda_plan *span = nullptr;
    ~grbda_plan() { if (span) grbda_plan_free(span); }
    std::vector<int32_t> span_q, span_v, crow;
    int n_cpl_rows = 0;
    bool has_trig = false;     // some implicit cluster is a trig-polynomial constraint (its sine / cosine cache takes dynamic LDS of the manifold constraint kernel)
    int constraint_shape = 0;  // manifold_kernels.hip, launch_manifold_constraint: 0 structured, 1 beyond the limits, 2 at most 4 bodies / 2 coordinates
    bool no_manifold = false;  // GRBDA_NO_MANIFOLD=1: implicit models keep the difference batches (A/B runs)
};

namespace {

int ensure_device(const grbda_plan *p, int device, DeviceTables **out)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return set_err(GRBDA_ENODEVICE, "no HIP device available (there is no CPU fallback)");
    if (device < 0 || device >= count) return set_err(GRBDA_EINVAL, "device index out of range");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_err(e, "hipSetDevice");
    std::lock_guard<std::recursive_mutex> lk(p->mu);
    auto it = p->dev.find(device);
    if (it != p->dev.end()) {
        *out = &it->second;
        return 0;
    }
    DeviceTables t;
    const HostPlan &h = p->host;
    auto up = [&](const void *src, size_t bytes, void **dst) -> hipError_t {
        hipError_t e2 = hipMalloc(dst, bytes ? bytes : 16);
        if (e2 != hipSuccess) return e2;
        return bytes ? hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
    };
    std::vector<float> c32(h.consts.begin(), h.consts.end());
    if ((e = up(h.aba_steps.data(), h.aba_steps.size() * sizeof(Step), (void **)&t.aba_steps)) != hipSuccess ||
        (e = up(h.rnea_steps.data(), h.rnea_steps.size() * sizeof(Step), (void **)&t.rnea_steps)) != hipSuccess ||
        (e = up(h.consts.data(), h.consts.size() * sizeof(double), (void **)&t.consts64)) != hipSuccess ||
        (e = up(c32.data(), c32.size() * sizeof(float), (void **)&t.consts32)) != hipSuccess ||
        (e = up(h.cints.data(), h.cints.size() * sizeof(int32_t), (void **)&t.cints)) != hipSuccess)
        return hip_err(e, "plan upload");
    {
        // tangent-space perturbation of the positions (UnitTests/testHelpers.hpp:50-112): kind 0 q[i] += d,
        // 1 free-base rotation (quat += quat x (0, d e_a) / 2), 2 free-base translation (pos += R^T d e_a),
        // (implicit-loop clusters: kind 0 on the independent spanning position + re-projection of the dependent ones)
        std::vector<int32_t> map(static_cast<size_t>(h.nv) * 3, 0);
        for (const ClusterRec &c : h.lay64.clusters)
            for (int a = 0; a < c.n; a++) {
                int32_t *e = &map[static_cast<size_t>(c.v_index + a) * 3];
                if (c.kind == CK_FREE && h.ori_repr == 0) {
                    e[0] = a < 3 ? 1 : 2;
                    e[1] = c.q_index;
                    e[2] = a % 3;
                } else if (c.kind == CK_FREE) {
                    // roll-pitch-yaw base: the reference's plus() only special-cases the quaternion (testHelpers.hpp:49-74);
                    // every other position vector is q + dq
                    e[0] = 0;
                    e[1] = c.q_index + a;
                } else if (c.kind == CK_LOOP) {
                    // implicit cluster: the a-th INDEPENDENT spanning position moves, the dependent ones are put back on
                    // phi(q) = 0 by the Newton projection (derived(), DM_DQ): the derivative on the constraint manifold
                    e[0] = 0;
                    e[1] = c.q_index + h.cints[c.iofs + 2 + a];
                } else {
                    e[0] = 0;
                    e[1] = c.q_index + a;
                }
            }
        if ((e = up(map.data(), map.size() * sizeof(int32_t), (void **)&t.dq_map)) != hipSuccess) return hip_err(e, "plan upload");
    }
    for (int w = 0; w < kLayouts; w++) {
        const Layout &L = layout_of(h, w);
        if ((e = up(L.clusters.data(), L.clusters.size() * sizeof(ClusterRec), (void **)&t.clusters[w])) != hipSuccess ||
            (e = up(L.rnea_clusters.data(), L.rnea_clusters.size() * sizeof(ClusterRec), (void **)&t.rnea_clusters[w])) != hipSuccess ||
            (e = up(L.bodies.data(), L.bodies.size() * sizeof(BodyRec), (void **)&t.bodies[w])) != hipSuccess ||
            (e = up(L.rnea_bodies.data(), L.rnea_bodies.size() * sizeof(BodyRec), (void **)&t.rnea_bodies[w])) != hipSuccess ||
            (e = up(L.acc_k.data(), L.acc_k.size() * sizeof(int32_t), (void **)&t.acc_k[w])) != hipSuccess)
            return hip_err(e, "plan upload");
    }
    for (int w = 0; w < 7; w++) {
        const RneaChainProgram &rp = w == 0 ? h.rchain32 : (w == 1 ? h.rchain64 : (w == 2 ? h.rchain32w : (w == 3 ? h.rchain32p : (w == 4 ? h.rchain64p : (w == 5 ? h.rchain32q : h.rchain64q)))));
        if (!rp.ok) continue;
        if ((e = up(rp.segs.data(), rp.segs.size() * sizeof(RneaSeg), (void **)&t.rchain_segs[w])) != hipSuccess ||
            (e = up(rp.links.data(), rp.links.size() * sizeof(RneaLink), (void **)&t.rchain_links[w])) != hipSuccess ||
            (e = up(rp.pairs.data(), rp.pairs.size() * sizeof(RneaPair), (void **)&t.rchain_pairs[w])) != hipSuccess ||
            (e = up(rp.frees.data(), rp.frees.size() * sizeof(RneaFree), (void **)&t.rchain_frees[w])) != hipSuccess ||
            (e = up(rp.diffs.data(), rp.diffs.size() * sizeof(RneaDiff), (void **)&t.rchain_diffs[w])) != hipSuccess ||
            (e = up(rp.gens.data(), rp.gens.size() * sizeof(ChainGen), (void **)&t.rchain_gens[w])) != hipSuccess ||
            (e = up(rp.gbodies.data(), rp.gbodies.size() * sizeof(ChainGenBody), (void **)&t.rchain_gbodies[w])) != hipSuccess)
            return hip_err(e, "plan upload");
        if ((e = set_max_dynamic_lds_chain()) != hipSuccess) return hip_err(e, "hipFuncSetAttribute");
    }
    if (h.crba.ok && (e = up(h.crba.bodies.data(), h.crba.bodies.size() * sizeof(CrbaBody), (void **)&t.crba_bodies)) != hipSuccess)
        return hip_err(e, "plan upload");
    if (h.deriv.ok) {
        if ((e = up(h.deriv.bodies.data(), h.deriv.bodies.size() * sizeof(DerivBody), (void **)&t.deriv_bodies)) != hipSuccess)
            return hip_err(e, "plan upload");
    }
    if (!h.deriv.related.empty() &&
        (e = up(h.deriv.related.data(), h.deriv.related.size() * sizeof(uint64_t), (void **)&t.deriv_related)) != hipSuccess)
        return hip_err(e, "plan upload");
    if (!h.related_table.empty() &&
        (e = up(h.related_table.data(), h.related_table.size() * sizeof(int32_t), (void **)&t.related_table)) != hipSuccess)
        return hip_err(e, "plan upload");
    if (h.deriv.ok && h.deriv.minv.ok &&
        ((e = up(h.deriv.minv.bodies.data(), h.deriv.minv.bodies.size() * sizeof(MinvBody), (void **)&t.minv_bodies)) != hipSuccess ||
         (e = up(h.deriv.minv.coltab.data(), h.deriv.minv.coltab.size() * sizeof(int32_t), (void **)&t.minv_coltab)) != hipSuccess))
        return hip_err(e, "plan upload");
    for (int w = 0; w < 7; w++) {
        const ChainProgram &cp = w == 0 ? h.chain32 : (w == 1 ? h.chain32w : (w == 2 ? h.chain64 : (w == 3 ? h.chain32p : (w == 4 ? h.chain64p : (w == 5 ? h.chain32q : h.chain64q)))));
        if (!cp.ok) continue;
        if ((e = up(cp.segs.data(), cp.segs.size() * sizeof(ChainSeg), (void **)&t.chain_segs[w])) != hipSuccess ||
            (e = up(cp.links.data(), cp.links.size() * sizeof(ChainLink), (void **)&t.chain_links[w])) != hipSuccess ||
            (e = up(cp.pairs.data(), cp.pairs.size() * sizeof(ChainPair), (void **)&t.chain_pairs[w])) != hipSuccess ||
            (e = up(cp.frees.data(), cp.frees.size() * sizeof(ChainFree), (void **)&t.chain_frees[w])) != hipSuccess ||
            (e = up(cp.diffs.data(), cp.diffs.size() * sizeof(ChainDiff), (void **)&t.chain_diffs[w])) != hipSuccess ||
            (e = up(cp.gens.data(), cp.gens.size() * sizeof(ChainGen), (void **)&t.chain_gens[w])) != hipSuccess ||
            (e = up(cp.gbodies.data(), cp.gbodies.size() * sizeof(ChainGenBody), (void **)&t.chain_gbodies[w])) != hipSuccess)
            return hip_err(e, "plan upload");
        if ((e = set_max_dynamic_lds_chain()) != hipSuccess) return hip_err(e, "hipFuncSetAttribute");
    }
    if (p->span) {
        if ((e = up(p->span_q.data(), p->span_q.size() * sizeof(int32_t), (void **)&t.span_q)) != hipSuccess ||
            (e = up(p->span_v.data(), p->span_v.size() * sizeof(int32_t), (void **)&t.span_v)) != hipSuccess ||
            (e = up(p->crow.data(), p->crow.size() * sizeof(int32_t), (void **)&t.crow)) != hipSuccess)
            return hip_err(e, "plan upload");
    }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return hip_err(e, "hipGetDeviceProperties");
    t.n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if ((e = set_max_dynamic_lds()) != hipSuccess) return hip_err(e, "hipFuncSetAttribute");
    if ((e = set_max_dynamic_lds_deriv()) != hipSuccess) return hip_err(e, "hipFuncSetAttribute");
    if ((e = set_max_dynamic_lds_minv()) != hipSuccess) return hip_err(e, "hipFuncSetAttribute");
    t.bad_count = spd_bad_count_address();
    auto ins = p->dev.emplace(device, t);
    *out = &ins.first->second;
    return 0;
}

int ensure_scratch(const grbda_plan *p, int device, void *stream, size_t bytes, void **out)
{
    std::lock_guard<std::recursive_mutex> lk(p->mu);
    Scratch &s = p->scratch[{device, stream}];
    if (s.bytes < bytes) {
        // A stream under capture must not see hipFree / hipMalloc, and a graph captured earlier on this (device, stream)
        // holds the slab's address: growing is refused while the stream captures (reserve with one eager call of the
        // largest batch first, include/grbda_hip.h "Graph capture")
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (stream && hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return set_err(GRBDA_EINVAL, "the per-stream scratch slab would have to grow during stream capture: run the largest "
                                         "batch once on this stream before capturing");
        if (s.ptr) {
            hipError_t e = hipFree(s.ptr);
            if (e != hipSuccess) return hip_err(e, "hipFree");
            s.ptr = nullptr;
            s.bytes = 0;
        }
        hipError_t e = hipMalloc(&s.ptr, bytes);
        if (e != hipSuccess) return hip_err(e, "hipMalloc(scratch)");
        s.bytes = bytes;
    }
    *out = s.ptr;
    return 0;
}

// The per-(device, stream) work slab of the derived quantities, grown on demand under the same rule as ensure_scratch: never
// while the stream captures (a graph captured earlier holds the old address).
// Upper bound of a work-slab request of the chunked pipelines (derivatives, projection): what the call site asks for (1-16 GiB, sized so that a
// million states go through in a few chunks), cut to GRBDA_WORK_MAX_MB when that is set and to
//     max(the slab this (device, stream) already holds, a quarter of the memory that is FREE on the device right now, 256 MiB)
// so that (a) a call never shrinks below what it already owns -- the chunk, and with it the timing, of a repeated call is reproducible whatever
// else has been allocated since -- and (b) a stream that is being CAPTURED derives its chunk from the held slab alone (no hipMemGetInfo, no
// growth: INTEGRATION.md's rule "run the largest batch once on the stream before capturing" then always suffices).  The slabs are kept per
// (device, stream) and never shrink by themselves; grbda_plan_release_work() hands them back.
static size_t work_budget(const grbda_plan *p, const std::map<std::pair<int, void *>, Scratch> &pool, int device, void *stream, size_t want)
{
    const int want_mb = env_int("GRBDA_WORK_WANT_MB", 0);  // (experiments: another chunk size)
    size_t cap = want_mb > 0 ? static_cast<size_t>(want_mb) << 20 : want;
    size_t held = 0;
    {
        std::lock_guard<std::recursive_mutex> lk(p->mu);
        const auto it = pool.find({device, stream});
        if (it != pool.end()) held = it->second.bytes > 256 ? it->second.bytes - 256 : 0;
    }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = stream && hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    if (capturing && held > 0) return cap < held ? cap : held;
    size_t free_b = 0, total_b = 0;
    if (!capturing && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0) {
        size_t limit = free_b / 4;
        if (limit < (256ull << 20)) limit = 256ull << 20;
        if (limit < held) limit = held;
        if (cap > limit) cap = limit;
    }
    const int mb = env_int("GRBDA_WORK_MAX_MB", 0);
    if (mb > 0 && cap > (static_cast<size_t>(mb) << 20)) cap = static_cast<size_t>(mb) << 20;
    return cap;
}

int ensure_work(const grbda_plan *p, std::map<std::pair<int, void *>, Scratch> &pool, int device, void *stream, size_t bytes, void **out)
{
    std::lock_guard<std::recursive_mutex> lk(p->mu);
    Scratch &s = pool[{device, stream}];
    if (s.bytes < bytes) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (stream && hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return set_err(GRBDA_EINVAL, "a per-stream work buffer would have to grow during stream capture: run the largest batch once on "
                                         "this stream before capturing");
        hipError_t e;
        if (s.ptr && (e = hipFree(s.ptr)) != hipSuccess) return hip_err(e, "hipFree");
        s.ptr = nullptr;
        s.bytes = 0;
        if ((e = hipMalloc(&s.ptr, bytes)) != hipSuccess) return hip_err(e, "hipMalloc(work)");
        s.bytes = bytes;
    }
    *out = s.ptr;
    return 0;
}

template <class T>
DevPlan<T> make_dev_plan(const grbda_plan *p, const DeviceTables &t, bool rnea, bool fext)
{
    DevPlan<T> d;
    const HostPlan &h = p->host;
    d.bad_count = t.bad_count;
    d.steps = rnea ? t.rnea_steps : t.aba_steps;
    d.n_steps = static_cast<int>(rnea ? h.rnea_steps.size() : h.aba_steps.size());
    int w = (sizeof(T) == 4 ? 0 : 1) + (fext ? 2 : 0);
    // f32 fast path: the split layout when it exists for this kernel (GRBDA_NO_SPLIT=1 keeps the mixed one, for A/B runs)
    const bool split = w == 0 && (rnea ? h.lay32s.split_rnea : h.lay32s.split_aba) && !p->no_split;
    if (split) w = 4;
    const Layout &L = layout_of(h, w);
    d.clusters = rnea ? t.rnea_clusters[w] : t.clusters[w];
    d.cints = t.cints;
    d.acc_k = t.acc_k[w];
    d.bodies = rnea ? t.rnea_bodies[w] : t.bodies[w];
    d.consts = sizeof(T) == 4 ? reinterpret_cast<const T *>(t.consts32) : reinterpret_cast<const T *>(t.consts64);
    d.nq = h.nq;
    d.nv = h.nv;
    d.n_lds_slots = rnea ? L.n_lds_rnea : L.n_lds_aba;
    d.n_glb_slots = rnea ? L.n_glb_rnea : L.n_glb_aba;
    d.ori_repr = h.ori_repr;
    d.general = fext ? 1 : 0;
    for (const ClusterRec &cr : L.clusters) d.general |= cr.kind == CK_LOOP;
    d.split = split ? 1 : 0;
    d.n_bodies = h.n_bodies;
    for (int i = 0; i < 6; i++) d.a_root[i] = static_cast<T>(-h.gravity[i]);
    return d;
}

template <class T>
bool chain_covers(const grbda_plan *p)
{
    if (p->no_chain) return false;
    if (sizeof(T) == 8) return p->host.chain64.ok;
    return p->host.chain32.ok || (p->host.chain32w.ok && p->host.chain32w.diffs.empty() && p->host.chain32w.gens.empty() && p->chain_wide);
}

// which forward-dynamics kernel a batch of B states runs on a device with n_cu compute units: ONE definition, used by the launch
// path below and by grbda_kernel_name (bench.py prints the name next to the roofline figures)
enum AbaPath { ABA_GEN1, ABA_LM, ABA_LM4, ABA_CHAIN_WIDE, ABA_CHAIN, ABA_INTERPRETER };
template <class T>
size_t lm_lds_bytes(const grbda_plan *p, int n_waves = 2)
{
    const HostPlan &h = p->host;
    const ChainProgram &lp = n_waves == 4 ? (sizeof(T) == 8 ? h.chain64q : h.chain32q) : (sizeof(T) == 8 ? h.chain64p : h.chain32p);
    const size_t stage_all = static_cast<size_t>(kWave) * static_cast<size_t>(h.nq + 2 * h.nv) * sizeof(T);
    const size_t lds_lm = static_cast<size_t>(lp.n_lds) * kWave * sizeof(T);
    return lds_lm < stage_all ? stage_all : lds_lm;
}
// the single-cluster kernels prefetch a state's positions into min(n + 3, 8) registers (implicit cluster: one per body) or n (explicit)
static bool gen1_positions_fit(int nq, const ChainGen &g)
{
    const int room = g.kind ? std::min(g.n + 3, kMaxClusterBodies) : g.n;
    return nq <= room && g.n <= 4;
}
template <class T>
size_t gen1_lds_bytes(const grbda_plan *p)
{
    const HostPlan &h = p->host;
    const ChainProgram &sp = sizeof(T) == 8 ? h.chain64 : h.chain32;
    return static_cast<size_t>(sp.n_lds) * kWave * sizeof(T) + static_cast<size_t>(kWave) * static_cast<size_t>(h.nq + 2 * h.nv) * sizeof(T);
}
template <class T>
AbaPath choose_aba(const grbda_plan *p, int n_cu, size_t B, bool f_ext)
{
    const HostPlan &h = p->host;
    if (f_ext || !chain_covers<T>(p)) return ABA_INTERPRETER;
    const size_t n_tiles = (B + kWave - 1) / kWave;
    const ChainProgram &sp = sizeof(T) == 8 ? h.chain64 : h.chain32;
    if (sp.ok && sp.single_gen && !p->chain_debug && gen1_lds_bytes<T>(p) <= 65536 && gen1_positions_fit(h.nq, sp.gens[0])) return ABA_GEN1;
    const ChainProgram &lp = sizeof(T) == 8 ? h.chain64p : h.chain32p;
    // four wavefronts per tile while that still leaves at most two wavefronts per SIMD (two tiles per CU); GRBDA_LM_WAVES=2 keeps two
    // (only the fp32 latency-mode kernels carry the differential segments: plan.cpp builds no fp64 program with them)
    const ChainProgram &lq = sizeof(T) == 8 ? h.chain64q : h.chain32q;
    if (lq.ok && (sizeof(T) == 4 || lq.diffs.empty()) && !p->no_latency_mode && !p->chain_debug && p->lm_waves != 2 && n_tiles > 0 &&
        n_tiles <= static_cast<size_t>(n_cu) * 2 && lm_lds_bytes<T>(p, 4) <= 81920)
        return ABA_LM4;
    if (lp.ok && (sizeof(T) == 4 || lp.diffs.empty()) && !p->no_latency_mode && !p->chain_debug && n_tiles <= static_cast<size_t>(n_cu) * 4 && n_tiles > 0 &&
        lm_lds_bytes<T>(p) <= 40960)
        return ABA_LM;
    const bool wide = sizeof(T) == 4 && h.chain32w.ok && h.chain32w.diffs.empty() && h.chain32w.gens.empty() && p->chain_wide &&
                      (!h.chain32.ok || n_tiles > static_cast<size_t>(n_cu) * 8);
    return wide ? ABA_CHAIN_WIDE : ABA_CHAIN;
}

template <class T>
int run_chain(const grbda_plan *p, const DeviceTables &t, const T *q, const T *qd, const T *tau, T *ydd, size_t B,
              int device, void *stream)
{
    const HostPlan &h = p->host;
    const size_t n_tiles0 = (B + kWave - 1) / kWave;
    const AbaPath path = choose_aba<T>(p, t.n_cu, B, false);
    {   // single-cluster programs: the fused, slab-free kernel (chain_kernels.hip, aba_gen1_kernel)
        const int w1 = sizeof(T) == 8 ? 2 : 0;
        const ChainProgram &sp = sizeof(T) == 8 ? h.chain64 : h.chain32;
        const size_t work = static_cast<size_t>(sp.n_lds) * kWave * sizeof(T);
        const size_t lds_total = gen1_lds_bytes<T>(p);  // work area + the staged input rows
        if (path == ABA_GEN1) {
            ChainDev<T> d;
            std::memset(&d, 0, sizeof d);
            d.bad_count = t.bad_count;
            d.gens = t.chain_gens[w1];
            d.gbodies = t.chain_gbodies[w1];
            d.n_gens = 1;
            d.cints = t.cints;
            d.consts = sizeof(T) == 4 ? reinterpret_cast<const T *>(t.consts32) : reinterpret_cast<const T *>(t.consts64);
            d.nq = h.nq;
            d.nv = h.nv;
            d.ori_repr = h.ori_repr;
            d.out_lds = -1;
            d.lds_bytes = static_cast<int>(work);
            for (int i = 0; i < 6; i++) d.a_root[i] = static_cast<T>(-h.gravity[i]);
            size_t per_cu = static_cast<size_t>(gen1_waves_per_simd<T>(sp.gens[0].n)) * 4;
            const size_t fit = lds_workgroups_per_cu(lds_total);
            if (fit < per_cu) per_cu = fit;
            if (p->gen1_waves_cap > 0 && static_cast<size_t>(p->gen1_waves_cap) < per_cu) per_cu = static_cast<size_t>(p->gen1_waves_cap);
            size_t grid = static_cast<size_t>(t.n_cu) * per_cu;
            if (p->gen1_tiles_per_wave > 0) grid = (n_tiles0 + p->gen1_tiles_per_wave - 1) / p->gen1_tiles_per_wave;
            if (grid > n_tiles0) grid = n_tiles0;
            hipError_t e = launch_aba_gen1<T>(d, sp.gens[0].n, sp.gens[0].kind != 0, q, qd, tau, ydd, B, static_cast<int>(grid), lds_total,
                                              static_cast<hipStream_t>(stream));
            return e == hipSuccess ? GRBDA_OK : hip_err(e, "aba single-cluster launch");
        }
    }
    // Latency mode: a batch of at most one tile per SIMD would leave every SIMD with a single wavefront; a tile then goes
    // to a workgroup of two wavefronts that split its limbs (chain_kernels.hip, aba_chain_lm_kernel).  GRBDA_NO_LATENCY_MODE=1
    // keeps the ordinary kernel (A/B runs); results agree to rounding (the base sums one partial inertia per wavefront).
    {
        const int lm_waves = path == ABA_LM4 ? 4 : 2;
        const ChainProgram &lp = lm_waves == 4 ? (sizeof(T) == 8 ? h.chain64q : h.chain32q) : (sizeof(T) == 8 ? h.chain64p : h.chain32p);
        const size_t lds_lm = lm_lds_bytes<T>(p, lm_waves);
        if (path == ABA_LM || path == ABA_LM4) {
            const int w = lm_waves == 4 ? (sizeof(T) == 8 ? 6 : 5) : (sizeof(T) == 8 ? 4 : 3);
            ChainDev<T> d;
            d.bad_count = t.bad_count;
            d.segs = t.chain_segs[w];
            d.links = t.chain_links[w];
            d.pairs = t.chain_pairs[w];
            d.frees = t.chain_frees[w];
            d.diffs = lp.diffs.empty() ? nullptr : t.chain_diffs[w];  // (fp32 programs only: plan.cpp)
            d.n_diffs = static_cast<int>(lp.diffs.size());
            d.gens = nullptr;
            d.gbodies = nullptr;
            d.n_gens = 0;
            d.cints = t.cints;
            d.consts = sizeof(T) == 4 ? reinterpret_cast<const T *>(t.consts32) : reinterpret_cast<const T *>(t.consts64);
            d.n_segs = static_cast<int>(lp.segs.size());
            d.nq = h.nq;
            d.nv = h.nv;
            d.n_glb_slots = lp.n_glb + 1;  // (+ the row that carries the other wavefronts' bad-pivot masks to wavefront 0, aba_chain_lm_kernel)
            d.ori_repr = h.ori_repr;
            d.debug = 0;
            d.sv_global = 0;
            d.fuse = d.stage_lds_v = d.stage_v_index = 0;
            d.out_lds = lp.out_lds;
            d.lds_bytes = static_cast<int>(lds_lm);
            for (int i = 0; i < 6; i++) d.a_root[i] = static_cast<T>(-h.gravity[i]);
            const size_t grid = n_tiles0;  // (<= 4 workgroups per CU: all resident)
            const size_t n_rows = static_cast<size_t>(d.n_glb_slots) + static_cast<size_t>(d.nq + 2 * d.nv);
            void *scratch = nullptr;
            if (int rc = ensure_scratch(p, device, stream, grid * n_rows * kWave * sizeof(T) + 256, &scratch)) return rc;
            hipError_t e = launch_aba_chain_lm<T>(d, q, qd, tau, ydd, B, static_cast<T *>(scratch), static_cast<int>(grid), lds_lm,
                                                  static_cast<hipStream_t>(stream), lm_waves);
            return e == hipSuccess ? GRBDA_OK : hip_err(e, "aba chain launch (latency mode)");
        }
    }
    // f32: four wavefronts per SIMD (16 per CU, half the LDS each) once the batch fills them, when that layout exists
    const bool wide = path == ABA_CHAIN_WIDE;
    const int w = sizeof(T) == 8 ? 2 : (wide ? 1 : 0);
    const int kid = sizeof(T) == 8 ? 1 : 0;
    const ChainProgram &cp = w == 2 ? h.chain64 : (wide ? h.chain32w : h.chain32);
    const size_t waves_per_cu = wide ? static_cast<size_t>(4 * kChainWideWps)
                                     : static_cast<size_t>((sizeof(T) == 8 && !cp.gens.empty()) ? p->waves_per_cu_f64_wide_regs : p->waves_per_cu[kid]);
    const size_t lds_budget = wide ? static_cast<size_t>(kChainWideLdsBytes) : static_cast<size_t>(p->lds_bytes_per_wave[kid]);
    ChainDev<T> d;
    d.bad_count = t.bad_count;
    d.segs = t.chain_segs[w];
    d.links = t.chain_links[w];
    d.pairs = t.chain_pairs[w];
    d.frees = t.chain_frees[w];
    d.diffs = t.chain_diffs[w];
    d.n_diffs = static_cast<int>(cp.diffs.size());
    d.gens = t.chain_gens[w];
    d.gbodies = t.chain_gbodies[w];
    d.n_gens = static_cast<int>(cp.gens.size());
    d.cints = t.cints;
    d.consts = sizeof(T) == 4 ? reinterpret_cast<const T *>(t.consts32) : reinterpret_cast<const T *>(t.consts64);
    d.n_segs = static_cast<int>(cp.segs.size());
    d.nq = h.nq;
    d.nv = h.nv;
    d.n_glb_slots = cp.n_glb;
    d.ori_repr = h.ori_repr;
    d.debug = p->chain_debug;
    d.sv_global = cp.sv_global ? 1 : 0;
    d.out_lds = cp.out_lds;
    d.fuse = d.stage_lds_v = d.stage_v_index = 0;
    {   // the floating base's segments that only move data through the slab (devplan.h, ChainDev::fuse)
        int n_free_bwd = 0, at_bwd = -1;
        for (size_t s = 0; s < cp.segs.size(); s++)
            if (cp.segs[s].op == SEG_FREE_BWD) { n_free_bwd++; at_bwd = static_cast<int>(s); }
        if (!cp.segs.empty() && cp.segs[0].op == SEG_FREE_FWD && n_free_bwd == 1 && cp.frees[cp.segs[0].first].lds_v >= 0) {
            d.fuse |= 1;
            d.stage_lds_v = cp.frees[cp.segs[0].first].lds_v;
            d.stage_v_index = cp.frees[cp.segs[0].first].v_index;
        }
        if (n_free_bwd == 1 && at_bwd + 1 < static_cast<int>(cp.segs.size()) && cp.segs[at_bwd + 1].op == SEG_FREE_ACC &&
            cp.segs[at_bwd + 1].first == cp.segs[at_bwd].first)
            d.fuse |= 2;
        if (env_int("GRBDA_NO_FREE_FUSE", 0)) d.fuse = 0;
    }
    for (int i = 0; i < 6; i++) d.a_root[i] = static_cast<T>(-h.gravity[i]);
    const size_t n_tiles = (B + kWave - 1) / kWave;
    size_t grid = static_cast<size_t>(t.n_cu) * waves_per_cu;
    if (grid > n_tiles) grid = n_tiles;
    // LDS: the slot store, and at tile boundaries the staging area of the input transposition
    size_t lds_bytes = static_cast<size_t>(cp.n_lds) * kWave * sizeof(T);
    const size_t stage_one = static_cast<size_t>(kWave) * static_cast<size_t>(d.nq > d.nv ? d.nq : d.nv) * sizeof(T);
    const size_t stage_all = static_cast<size_t>(kWave) * static_cast<size_t>(d.nq + 2 * d.nv) * sizeof(T);
    if (lds_bytes < stage_one) lds_bytes = stage_one;
    if (lds_bytes < stage_all && stage_all <= lds_budget) lds_bytes = stage_all;
    d.lds_bytes = static_cast<int>(lds_bytes);
    if (lds_bytes < stage_all) d.fuse &= ~1;  // (the prologue stages one array at a time: the velocities are gone when it returns)
    const size_t fit = lds_workgroups_per_cu(lds_bytes);
    if (fit >= 1 && fit < waves_per_cu) {
        const size_t g2 = static_cast<size_t>(t.n_cu) * fit;
        if (grid > g2) grid = g2;
    }
    const size_t n_rows = static_cast<size_t>(cp.n_glb) + static_cast<size_t>(d.nq + 2 * d.nv);
    void *scratch = nullptr;
    if (int rc = ensure_scratch(p, device, stream, grid * n_rows * kWave * sizeof(T) + 256, &scratch)) return rc;
    hipError_t e = launch_aba_chain<T>(d, q, qd, tau, ydd, B, static_cast<T *>(scratch), static_cast<int>(grid), lds_bytes,
                                           static_cast<hipStream_t>(stream), wide);
    return e == hipSuccess ? GRBDA_OK : hip_err(e, "aba chain launch");
}

template <class T>
size_t rnea_gen1_lds_bytes(const grbda_plan *p)
{
    const HostPlan &h = p->host;
    const RneaChainProgram &rp = sizeof(T) == 8 ? h.rchain64 : h.rchain32;
    return static_cast<size_t>(rp.n_lds) * kWave * sizeof(T) + static_cast<size_t>(kWave) * static_cast<size_t>(h.nq + 2 * h.nv) * sizeof(T);
}
template <class T>
bool rnea_gen1_usable(const grbda_plan *p)
{
    const RneaChainProgram &rp = sizeof(T) == 8 ? p->host.rchain64 : p->host.rchain32;
    return rp.ok && rp.single_gen && !p->chain_debug && rnea_gen1_lds_bytes<T>(p) <= 65536 && gen1_positions_fit(p->host.nq, rp.gens[0]);
}

// latency mode of the inverse dynamics: wavefronts per tile for a batch of B states (0: the one-wavefront kernels); ONE definition, used by the launch path
// and by grbda_kernel_name
template <class T>
int choose_rnea_lm(const grbda_plan *p, int n_cu, size_t B)
{
    const HostPlan &h = p->host;
    const size_t n_tiles = (B + kWave - 1) / kWave;
    if (p->no_latency_mode || p->no_chain || n_tiles == 0) return 0;
    const bool kid = sizeof(T) == 8;
    const RneaChainProgram &r4 = kid ? h.rchain64q : h.rchain32q, &r2 = kid ? h.rchain64p : h.rchain32p;
    const size_t stage_all = static_cast<size_t>(kWave) * static_cast<size_t>(h.nq + 2 * h.nv) * sizeof(T);
    if (r4.ok && r4.n_waves == 4 && (!kid || r4.diffs.empty()) && p->lm_waves != 2 && n_tiles <= static_cast<size_t>(n_cu) * 2 &&
        std::max(static_cast<size_t>(r4.n_lds) * kWave * sizeof(T), stage_all) <= 81920)
        return 4;
    if (r2.ok && r2.n_waves == 2 && (!kid || r2.diffs.empty()) && n_tiles <= static_cast<size_t>(n_cu) * 4 && std::max(static_cast<size_t>(r2.n_lds) * kWave * sizeof(T), stage_all) <= 40960)
        return 2;
    return 0;
}

template <class T>
int run_rnea_chain(const grbda_plan *p, const DeviceTables &t, const T *q, const T *qd, const T *ydd, T *tau, size_t B, int device,
                   void *stream)
{
    const HostPlan &h = p->host;
    const int kid = sizeof(T) == 8 ? 1 : 0;
    const size_t n_tiles = (B + kWave - 1) / kWave;
    // f32: the inverse-dynamics kernel needs 109 VGPRs, so four wavefronts per SIMD fit; models whose blocks all fit the
    // LDS of the two-per-SIMD shape (no global-slab fallback) run the program laid out for half the LDS per wavefront
    // once the batch fills 16 wavefronts per CU (MIT humanoid 0.105 -> 0.095 ms, Mini Cheetah 0.079 -> 0.070 ms; JVRC-1,
    // whose blocks spill already, loses)
    const bool wide = sizeof(T) == 4 && !p->rnea_narrow && h.rchain32w.ok && h.rchain32.ok && h.rchain32.n_glb == 0 && h.rchain32.gens.empty() &&
                      n_tiles > static_cast<size_t>(t.n_cu) * 8;
    const int w = wide ? 2 : kid;
    const RneaChainProgram &rp = wide ? h.rchain32w : (kid ? h.rchain64 : h.rchain32);
    // Latency mode (as the forward dynamics': batches of at most one tile per SIMD go to workgroups of two wavefronts per tile, of at most two tiles per CU to
    // workgroups of four when the base carries four limbs; chain_kernels.hip, rnea_chain_lm_kernel).  GRBDA_NO_LATENCY_MODE=1 / GRBDA_LM_WAVES=2 as there.
    {
        const RneaChainProgram &r4 = kid ? h.rchain64q : h.rchain32q, &r2 = kid ? h.rchain64p : h.rchain32p;
        const size_t stage_all = static_cast<size_t>(kWave) * static_cast<size_t>(h.nq + 2 * h.nv) * sizeof(T);
        const int lm_waves = choose_rnea_lm<T>(p, t.n_cu, B);
        const bool use4 = lm_waves == 4, use2 = lm_waves == 2;
        if (use4 || use2) {
            const RneaChainProgram &lp = use4 ? r4 : r2;
            const int wi = use4 ? (kid ? 6 : 5) : (kid ? 4 : 3);
            RneaChainDev<T> d;
            std::memset(&d, 0, sizeof d);
            d.segs = t.rchain_segs[wi];
            d.links = t.rchain_links[wi];
            d.pairs = t.rchain_pairs[wi];
            d.frees = t.rchain_frees[wi];
            d.diffs = lp.diffs.empty() ? nullptr : t.rchain_diffs[wi];  // (fp32 programs only)
            d.n_diffs = static_cast<int>(lp.diffs.size());
            d.cints = t.cints;
            d.consts = sizeof(T) == 4 ? reinterpret_cast<const T *>(t.consts32) : reinterpret_cast<const T *>(t.consts64);
            d.n_segs = static_cast<int>(lp.segs.size());
            d.nq = h.nq;
            d.nv = h.nv;
            d.ori_repr = h.ori_repr;
            for (int i = 0; i < 6; i++) d.a_root[i] = static_cast<T>(-h.gravity[i]);
            const size_t lds_bytes = std::max(static_cast<size_t>(lp.n_lds) * kWave * sizeof(T), stage_all);
            d.lds_bytes = static_cast<int>(lds_bytes);
            const size_t grid = n_tiles;  // (all resident)
            void *scratch = nullptr;
            if (int rc = ensure_scratch(p, device, stream, grid * static_cast<size_t>(h.nq + 2 * h.nv) * kWave * sizeof(T) + 256, &scratch)) return rc;
            hipError_t e = launch_rnea_chain_lm<T>(d, q, qd, ydd, tau, B, static_cast<T *>(scratch), static_cast<int>(grid), lds_bytes,
                                                   static_cast<hipStream_t>(stream), use4 ? 4 : 2);
            return e == hipSuccess ? GRBDA_OK : hip_err(e, "rnea chain launch (latency mode)");
        }
    }
    if (rnea_gen1_usable<T>(p)) {  // single-cluster programs: the fused, slab-free kernel (chain_kernels.hip, rnea_gen1_kernel)
        RneaChainDev<T> d;
        std::memset(&d, 0, sizeof d);
        d.gens = t.rchain_gens[kid];
        d.gbodies = t.rchain_gbodies[kid];
        d.n_gens = 1;
        d.cints = t.cints;
        d.consts = sizeof(T) == 4 ? reinterpret_cast<const T *>(t.consts32) : reinterpret_cast<const T *>(t.consts64);
        d.nq = h.nq;
        d.nv = h.nv;
        d.ori_repr = h.ori_repr;
        d.lds_bytes = static_cast<int>(static_cast<size_t>(rp.n_lds) * kWave * sizeof(T));
        for (int i = 0; i < 6; i++) d.a_root[i] = static_cast<T>(-h.gravity[i]);
        const size_t lds_total = rnea_gen1_lds_bytes<T>(p);  // work area + the staged input rows
        size_t per_cu = static_cast<size_t>(rnea_gen1_waves_per_simd<T>(rp.gens[0].n)) * 4;
        const size_t fit = lds_workgroups_per_cu(lds_total);
        if (fit < per_cu) per_cu = fit;
        if (p->gen1_waves_cap > 0 && static_cast<size_t>(p->gen1_waves_cap) < per_cu) per_cu = static_cast<size_t>(p->gen1_waves_cap);
        size_t grid = static_cast<size_t>(t.n_cu) * per_cu;
        if (p->gen1_tiles_per_wave > 0) grid = (n_tiles + p->gen1_tiles_per_wave - 1) / p->gen1_tiles_per_wave;
        if (grid > n_tiles) grid = n_tiles;
        hipError_t e = launch_rnea_gen1<T>(d, rp.gens[0].n, rp.gens[0].kind != 0, q, qd, ydd, tau, B, static_cast<int>(grid), lds_total,
                                           static_cast<hipStream_t>(stream));
        return e == hipSuccess ? GRBDA_OK : hip_err(e, "rnea single-cluster launch");
    }
    RneaChainDev<T> d;
    d.segs = t.rchain_segs[w];
    d.links = t.rchain_links[w];
    d.pairs = t.rchain_pairs[w];
    d.frees = t.rchain_frees[w];
    d.diffs = t.rchain_diffs[w];
    d.n_diffs = static_cast<int>(rp.diffs.size());
    d.gens = t.rchain_gens[w];
    d.gbodies = t.rchain_gbodies[w];
    d.n_gens = static_cast<int>(rp.gens.size());
    d.cints = t.cints;
    d.consts = sizeof(T) == 4 ? reinterpret_cast<const T *>(t.consts32) : reinterpret_cast<const T *>(t.consts64);
    d.n_segs = static_cast<int>(rp.segs.size());
    d.nq = h.nq;
    d.nv = h.nv;
    d.n_glb_slots = rp.n_glb;
    d.ori_repr = h.ori_repr;
    for (int i = 0; i < 6; i++) d.a_root[i] = static_cast<T>(-h.gravity[i]);
    const size_t waves_per_cu = wide ? static_cast<size_t>(4 * kChainWideWps) : static_cast<size_t>(p->waves_per_cu[kid]);  // the ABA launch shape: 8 wavefronts per CU
    const size_t lds_budget = wide ? static_cast<size_t>(kChainWideLdsBytes) : static_cast<size_t>(p->lds_bytes_per_wave[kid]);
    size_t grid = static_cast<size_t>(t.n_cu) * waves_per_cu;
    if (grid > n_tiles) grid = n_tiles;
    size_t lds_bytes = static_cast<size_t>(rp.n_lds) * kWave * sizeof(T);
    const size_t stage_one = static_cast<size_t>(kWave) * static_cast<size_t>(d.nq > d.nv ? d.nq : d.nv) * sizeof(T);
    const size_t stage_all = static_cast<size_t>(kWave) * static_cast<size_t>(d.nq + 2 * d.nv) * sizeof(T);
    if (lds_bytes < stage_one) lds_bytes = stage_one;
    if (lds_bytes < stage_all && stage_all <= lds_budget) lds_bytes = stage_all;
    d.lds_bytes = static_cast<int>(lds_bytes);
    const size_t fit = lds_workgroups_per_cu(lds_bytes);
    if (fit >= 1 && fit < waves_per_cu) {
        const size_t g2 = static_cast<size_t>(t.n_cu) * fit;
        if (grid > g2) grid = g2;
    }
    const size_t n_rows = static_cast<size_t>(d.nq + 2 * d.nv) + static_cast<size_t>(rp.n_glb);
    void *scratch = nullptr;
    if (int rc = ensure_scratch(p, device, stream, grid * n_rows * kWave * sizeof(T) + 256, &scratch)) return rc;
    hipError_t e = launch_rnea_chain<T>(d, q, qd, ydd, tau, B, static_cast<T *>(scratch), static_cast<int>(grid), lds_bytes,
                                        static_cast<hipStream_t>(stream));
    return e == hipSuccess ? GRBDA_OK : hip_err(e, "rnea chain launch");
}

template <class T>
int projection_run(const grbda_plan *p, bool rnea, const T *q, const T *qd, const T *x, const T *f_ext, T *out, size_t B, int device, void *stream);
int projection_run_f32_through_f64(const grbda_plan *p, bool rnea, const float *q, const float *qd, const float *x, const float *f_ext, float *out,
                                   size_t B, int device, void *stream);

template <class T>
int run(const grbda_plan *p, bool rnea, const T *q, const T *qd, const T *x, const T *f_ext, T *out, size_t B,
        int device, void *stream)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    GRBDA_CALL_SCOPE(p);
    if (!p || !q || !qd || !x || !out) return set_err(GRBDA_EINVAL, "null argument");
    if (B == 0) return GRBDA_OK;
    if (p->host.projection_only) {
        // (external forces: world-frame wrenches on the BODIES, which the spanning model shares with this one -- they enter its inverse dynamics)
        if constexpr (sizeof(T) == 4) {
            // clusters beyond the structured limits: a dense solve over chains tens of links long loses cond(H) x 6e-8 in fp32 (measured
            // 1e-2 on the reference's depth-10 parallel chains); the route is slow anyway, so fp32 callers get the fp64 route's result
            if (p->host.big_clusters && !rnea) return projection_run_f32_through_f64(p, rnea, q, qd, x, f_ext, out, B, device, stream);
        }
        return projection_run<T>(p, rnea, q, qd, x, f_ext, out, B, device, stream);
    }
    DeviceTables *t = nullptr;
    if (int rc = ensure_device(p, device, &t)) return rc;
    // chain-structured fast path (chain_kernels.hip): forward dynamics of models the chain program covers
    if (!rnea && !f_ext && chain_covers<T>(p)) return run_chain<T>(p, *t, q, qd, x, out, B, device, stream);
    if (rnea && !f_ext && !p->no_chain && (sizeof(T) == 8 ? p->host.rchain64.ok : p->host.rchain32.ok))
        return run_rnea_chain<T>(p, *t, q, qd, x, out, B, device, stream);
    DevPlan<T> d = make_dev_plan<T>(p, *t, rnea, f_ext != nullptr);
    d.fext = f_ext;
    const size_t n_tiles = (B + kWave - 1) / kWave;
    const int kid = (rnea ? 2 : 0) + (sizeof(T) == 8 ? 1 : 0);
    const size_t waves_per_cu = static_cast<size_t>(kid == 1 ? p->waves_per_cu_f64_wide_regs : p->waves_per_cu[kid]);
    size_t grid = static_cast<size_t>(t->n_cu) * waves_per_cu;
    if (grid > n_tiles) grid = n_tiles;
    const size_t n_glb = static_cast<size_t>(d.n_glb_slots) + static_cast<size_t>(d.nq + 2 * d.nv);  // + staged inputs
    const size_t scratch_bytes = grid * n_glb * kWave * sizeof(T) + 256;
    void *scratch = nullptr;
    if (int rc = ensure_scratch(p, device, stream, scratch_bytes, &scratch)) return rc;
    // LDS: the slot store, and at tile boundaries the staging area of the input transposition
    size_t lds_bytes = static_cast<size_t>(d.n_lds_slots) * kWave * sizeof(T);
    const size_t stage_one = static_cast<size_t>(kWave) * static_cast<size_t>(d.nq > d.nv ? d.nq : d.nv) * sizeof(T);
    const size_t stage_all = static_cast<size_t>(kWave) * static_cast<size_t>(d.nq + 2 * d.nv) * sizeof(T);
    if (lds_bytes < stage_one) lds_bytes = stage_one;
    if (lds_bytes < stage_all && stage_all <= static_cast<size_t>(p->lds_bytes_per_wave[kid])) lds_bytes = stage_all;
    d.lds_bytes = static_cast<int>(lds_bytes);
    // a CU holds 160 KiB of LDS: never launch more persistent wavefronts than can be resident at once
    const size_t fit = lds_workgroups_per_cu(lds_bytes);
    if (fit >= 1 && fit < waves_per_cu) {
        const size_t g2 = static_cast<size_t>(t->n_cu) * fit;
        if (grid > g2) grid = g2;
    }
    hipError_t e;
    if (rnea)
        e = launch_rnea<T>(d, q, qd, x, out, B, static_cast<T *>(scratch), static_cast<int>(grid), lds_bytes,
                           static_cast<hipStream_t>(stream));
    else
        // (f64 only) the two-wavefronts-per-SIMD build pays for its spills only when the grid fills them
        e = launch_aba<T>(d, q, qd, x, out, B, static_cast<T *>(scratch), static_cast<int>(grid), lds_bytes,
                          static_cast<hipStream_t>(stream), grid > static_cast<size_t>(t->n_cu) * 4);
    if (e != hipSuccess) return hip_err(e, rnea ? "rnea launch" : "aba launch");
    return GRBDA_OK;
}

int run_host_f64(const grbda_plan *p, bool rnea, const double *q, const double *qd, const double *x,
                 const double *f_ext, double *out, size_t B, int device)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    GRBDA_CALL_SCOPE(p);
    if (!p || !q || !qd || !x || !out) return set_err(GRBDA_EINVAL, "null argument");
    if (B == 0) return GRBDA_OK;
    DeviceTables *t = nullptr;
    if (int rc = ensure_device(p, device, &t)) return rc;
    const size_t nq = p->host.nq, nv = p->host.nv;
    double *dq = nullptr, *dqd = nullptr, *dx = nullptr, *dout = nullptr, *dfe = nullptr;
    const size_t nfe = static_cast<size_t>(p->host.n_bodies) * 6;
    hipError_t e;
    int rc = GRBDA_OK;
    if (f_ext) {
        if ((e = hipMalloc((void **)&dfe, B * nfe * 8)) != hipSuccess) return hip_err(e, "hipMalloc");
        if ((e = hipMemcpy(dfe, f_ext, B * nfe * 8, hipMemcpyHostToDevice)) != hipSuccess) { (void)hipFree(dfe); return hip_err(e, "hipMemcpy H2D"); }
    }
    if ((e = hipMalloc((void **)&dq, B * nq * 8)) != hipSuccess || (e = hipMalloc((void **)&dqd, B * nv * 8)) != hipSuccess ||
        (e = hipMalloc((void **)&dx, B * nv * 8)) != hipSuccess || (e = hipMalloc((void **)&dout, B * nv * 8)) != hipSuccess) {
        rc = hip_err(e, "hipMalloc");
    } else if ((e = hipMemcpy(dq, q, B * nq * 8, hipMemcpyHostToDevice)) != hipSuccess ||
               (e = hipMemcpy(dqd, qd, B * nv * 8, hipMemcpyHostToDevice)) != hipSuccess ||
               (e = hipMemcpy(dx, x, B * nv * 8, hipMemcpyHostToDevice)) != hipSuccess) {
        rc = hip_err(e, "hipMemcpy H2D");
    } else {
        rc = run<double>(p, rnea, dq, dqd, dx, dfe, dout, B, device, nullptr);
        if (rc == GRBDA_OK) {
            if ((e = hipDeviceSynchronize()) != hipSuccess) rc = hip_err(e, "kernel execution");
            else if ((e = hipMemcpy(out, dout, B * nv * 8, hipMemcpyDeviceToHost)) != hipSuccess) rc = hip_err(e, "hipMemcpy D2H");
        }
    }
    if (dq) (void)hipFree(dq);
    if (dqd) (void)hipFree(dqd);
    if (dx) (void)hipFree(dx);
    if (dout) (void)hipFree(dout);
    if (dfe) (void)hipFree(dfe);
    return rc;
}

// ---- steps either side of the path: Newton projection, spanning recovery (kernels.hip) -----------------------
int span_count(const grbda_plan *p)
{
    int n = 0;
    for (const ClusterRec &c : p->host.lay64.clusters) n += c.kind == CK_FREE ? 6 : c.k;
    return n;
}

// launch shape of the ABA kernel of the same precision: the auxiliary kernels use its slot layout
template <class T>
int aux_setup(const grbda_plan *p, size_t B, int device, void *stream, DevPlan<T> &d, T **scratch, int *grid,
              size_t *lds_bytes)
{
    // (the auxiliary kernels walk the cluster tables with per-lane arrays sized by kMaxClusterBodies / kMaxClusterDof)
    if (p->host.big_clusters)
        return set_err(GRBDA_EUNSUPPORTED, "only forward / inverse dynamics and the mass matrix are covered for clusters beyond the structured kernels' limits");
    DeviceTables *t = nullptr;
    if (int rc = ensure_device(p, device, &t)) return rc;
    d = make_dev_plan<T>(p, *t, false, false);
    const int kid = sizeof(T) == 8 ? 1 : 0;
    const size_t n_tiles = (B + kWave - 1) / kWave;
    size_t g = static_cast<size_t>(t->n_cu) * 4;  // one wavefront per SIMD: these kernels are not register-tuned
    if (g > n_tiles) g = n_tiles;
    const size_t n_glb = static_cast<size_t>(d.n_glb_slots) + static_cast<size_t>(d.nq + 2 * d.nv);
    void *sp = nullptr;
    if (int rc = ensure_scratch(p, device, stream, g * n_glb * kWave * sizeof(T) + 256, &sp)) return rc;
    size_t lb = static_cast<size_t>(d.n_lds_slots) * kWave * sizeof(T);
    const size_t stage_one = static_cast<size_t>(kWave) * static_cast<size_t>(d.nq > d.nv ? d.nq : d.nv) * sizeof(T);
    const size_t stage_all = static_cast<size_t>(kWave) * static_cast<size_t>(d.nq + 2 * d.nv) * sizeof(T);
    if (lb < stage_one) lb = stage_one;
    if (lb < stage_all && stage_all <= static_cast<size_t>(p->lds_bytes_per_wave[kid])) lb = stage_all;
    d.lds_bytes = static_cast<int>(lb);
    *scratch = static_cast<T *>(sp);
    *grid = static_cast<int>(g);
    *lds_bytes = lb;
    return GRBDA_OK;
}

template <class T>
int project(const grbda_plan *p, T *q, int32_t *ok, size_t B, int max_iter, double tol, int device, void *stream)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    GRBDA_CALL_SCOPE(p);
    if (!p || !q || max_iter < 0) return set_err(GRBDA_EINVAL, "bad argument");
    if (B == 0) return GRBDA_OK;
    if (p->host.big_clusters) {  // (clusters beyond the structured limits: the wide Newton kernel of manifold_kernels.hip)
        DeviceTables *t = nullptr;
        if (int rc = ensure_device(p, device, &t)) return rc;
        DevPlan<T> dp = make_dev_plan<T>(p, *t, false, false);
        size_t g = static_cast<size_t>(t->n_cu) * 4;
        if (g > (B + kWave - 1) / kWave) g = (B + kWave - 1) / kWave;
        hipError_t e = launch_manifold_newton<T>(dp, p->host.n_clusters, q, ok, B, max_iter, static_cast<T>(tol), static_cast<int>(g),
                                                 static_cast<hipStream_t>(stream));
        return e == hipSuccess ? GRBDA_OK : hip_err(e, "projection launch");
    }
    DevPlan<T> d;
    T *scratch = nullptr;
    int grid = 0;
    size_t lds = 0;
    if (int rc = aux_setup<T>(p, B, device, stream, d, &scratch, &grid, &lds)) return rc;
    hipError_t e = launch_project<T>(d, p->host.n_clusters, q, ok, B, max_iter, static_cast<T>(tol), scratch, grid, lds,
                                     static_cast<hipStream_t>(stream));
    return e == hipSuccess ? GRBDA_OK : hip_err(e, "projection launch");
}

template <class T>
int spanning(const grbda_plan *p, const T *q, const T *qd, const T *ydd, T *qd_span, T *qdd_span, size_t B, int device,
             void *stream)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    GRBDA_CALL_SCOPE(p);
    if (!p || !q || !qd || !ydd || !qdd_span) return set_err(GRBDA_EINVAL, "null argument");
    if (B == 0) return GRBDA_OK;
    if (p->host.big_clusters) {
        // clusters beyond the structured limits: qd_s = G yd, qdd_s = G ydd + g from the wide constraint kernel of the spanning-tree route
        // (manifold_kernels.hip), whose spanning coordinates are the bodies in order -- the layout of this entry point
        if (!p->span) return set_err(GRBDA_EUNSUPPORTED, "the model needs the spanning-tree route, which covers at most 128 velocities");
        DeviceTables *t = nullptr;
        if (int rc = ensure_device(p, device, &t)) return rc;
        const size_t nq = p->host.nq, nv = p->host.nv, nq_s = p->span->host.nq, nv_s = p->span->host.nv;
        if (static_cast<size_t>(span_count(p)) != nv_s) return set_err(GRBDA_EUNSUPPORTED, "spanning layout mismatch");
        const size_t per_state = nq_s + nv_s + static_cast<size_t>(p->n_cpl_rows);
        size_t chunk = work_budget(p, p->work_proj, device, stream, 1024ull << 20) / (per_state * sizeof(T));
        chunk &= ~static_cast<size_t>(kWave - 1);
        if (chunk < static_cast<size_t>(kWave)) chunk = kWave;
        const size_t b_round = (B + kWave - 1) / kWave * kWave;
        if (chunk > b_round) chunk = b_round;
        void *wptr = nullptr;
        if (int rc = ensure_work(p, p->work_proj, device, stream, chunk * per_state * sizeof(T) + 256, &wptr)) return rc;
        T *q_s = static_cast<T *>(wptr), *v_tmp = q_s + chunk * nq_s, *cpl = v_tmp + chunk * nv_s;
        DevPlan<T> dp = make_dev_plan<T>(p, *t, false, false);
        for (size_t b0 = 0; b0 < B; b0 += chunk) {
            const size_t nb = B - b0 < chunk ? B - b0 : chunk;
            size_t g = static_cast<size_t>(t->n_cu) * 4;
            if (g > (nb + kWave - 1) / kWave) g = (nb + kWave - 1) / kWave;
            hipError_t e = launch_manifold_constraint<T>(dp, p->host.n_clusters, t->span_q, t->span_v, t->crow, static_cast<int>(nq_s),
                                                         static_cast<int>(nv_s), p->n_cpl_rows, 0, q + b0 * nq, qd + b0 * nv, ydd + b0 * nv, q_s,
                                                         qd_span ? qd_span + b0 * nv_s : v_tmp, qdd_span + b0 * nv_s, cpl, nb, static_cast<int>(g),
                                                         static_cast<hipStream_t>(stream), 1, p->has_trig);
            if (e != hipSuccess) return hip_err(e, "manifold constraint launch");
        }
        return GRBDA_OK;
    }
    DevPlan<T> d;
    T *scratch = nullptr;
    int grid = 0;
    size_t lds = 0;
    if (int rc = aux_setup<T>(p, B, device, stream, d, &scratch, &grid, &lds)) return rc;
    hipError_t e = launch_spanning<T>(d, p->host.n_clusters, span_count(p), q, qd, ydd, qd_span, qdd_span, B, scratch,
                                      grid, lds, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? GRBDA_OK : hip_err(e, "spanning launch");
}

// ---- state input in the reference's conventions (kernels.hip, state_kernel) ----------------------------------
// widths of the caller's rows for the given per-cluster flags; GRBDA_ESTATE for independent positions of an implicit cluster
int state_widths(const grbda_plan *p, const uint8_t *pos_sp, const uint8_t *vel_sp, StateFlags *F, int *in_nq, int *in_nv)
{
    const auto &cl = p->host.lay64.clusters;
    if (cl.size() > static_cast<size_t>(64 * kStateFlagWords)) return set_err(GRBDA_EUNSUPPORTED, "more than 256 clusters");
    StateFlags f{};
    int wq = 0, wv = 0;
    const int npos_free = p->host.ori_repr == GRBDA_ORI_QUATERNION ? 7 : 6;
    for (size_t c = 0; c < cl.size(); c++) {
        const ClusterRec &r = cl[c];
        const bool ps = pos_sp ? pos_sp[c] != 0 : r.kind == CK_LOOP, vs = vel_sp ? vel_sp[c] != 0 : false;
        if (r.kind == CK_LOOP && !ps)
            return set_err(GRBDA_ESTATE, "cluster " + std::to_string(c) +
                                             ": Independent positions cannot be converted to spanning positions when the constraint is implicit.");
        if (ps) f.pos[c >> 6] |= 1ull << (c & 63);
        if (vs) f.vel[c >> 6] |= 1ull << (c & 63);
        wq += r.kind == CK_FREE ? npos_free : (ps ? r.k : r.n);
        wv += r.kind == CK_FREE ? 6 : (vs ? r.k : r.n);
    }
    if (F) *F = f;
    if (in_nq) *in_nq = wq;
    if (in_nv) *in_nv = wv;
    return GRBDA_OK;
}

template <class T>
int state_convert(const grbda_plan *p, const uint8_t *pos_sp, const uint8_t *vel_sp, const T *q_in, const T *qd_in, T *q, T *qd,
                  int32_t *status, T *cond, size_t B, double tol, int device, void *stream)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    GRBDA_CALL_SCOPE(p);
    if (!q_in || (qd && !qd_in)) return set_err(GRBDA_EINVAL, "null argument");
    StateFlags F;
    int in_nq = 0, in_nv = 0;
    if (int rc = state_widths(p, pos_sp, vel_sp, &F, &in_nq, &in_nv)) return rc;
    if (B == 0) return GRBDA_OK;
    if (p->host.big_clusters) {  // (clusters beyond the structured limits: the same rules in manifold_kernels.hip's wide state kernel)
        DeviceTables *t = nullptr;
        if (int rc = ensure_device(p, device, &t)) return rc;
        DevPlan<T> dp = make_dev_plan<T>(p, *t, false, false);
        size_t g = static_cast<size_t>(t->n_cu) * 4;
        if (g > (B + kWave - 1) / kWave) g = (B + kWave - 1) / kWave;
        hipError_t e = launch_manifold_state<T>(dp, p->host.n_clusters, F, q_in, qd_in, in_nq, in_nv, q, qd, status, cond, B, static_cast<T>(tol),
                                                static_cast<int>(g), static_cast<hipStream_t>(stream));
        return e == hipSuccess ? GRBDA_OK : hip_err(e, "state conversion launch");
    }
    DevPlan<T> d;
    T *scratch = nullptr;
    int grid = 0;
    size_t lds = 0;
    if (int rc = aux_setup<T>(p, B, device, stream, d, &scratch, &grid, &lds)) return rc;
    hipError_t e = launch_state<T>(d, p->host.n_clusters, F, q_in, qd_in, in_nq, in_nv, q, qd, status, cond, B, static_cast<T>(tol),
                                   scratch, grid, lds, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? GRBDA_OK : hip_err(e, "state conversion launch");
}

// ---- contact side: body poses, test force (include/grbda_hip.h) ---------------------------------------------
template <class T>
int poses(const grbda_plan *p, const T *q, T *Xa, size_t B, int device, void *stream)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    GRBDA_CALL_SCOPE(p);
    if (!p || !q || !Xa) return set_err(GRBDA_EINVAL, "null argument");
    if (B == 0) return GRBDA_OK;
    DeviceTables *t = nullptr;
    if (int rc = ensure_device(p, device, &t)) return rc;
    DevPlan<T> d = make_dev_plan<T>(p, *t, false, false);
    const size_t n_tiles = (B + kWave - 1) / kWave;
    const size_t g = n_tiles < static_cast<size_t>(t->n_cu) * 8 ? n_tiles : static_cast<size_t>(t->n_cu) * 8;
    hipError_t e = launch_poses<T>(d, p->host.n_clusters, q, Xa, B, static_cast<int>(g), static_cast<hipStream_t>(stream));
    return e == hipSuccess ? GRBDA_OK : hip_err(e, "poses launch");
}

// spatial velocity / acceleration of every body: the spanning rates (spanning_kernel) into the plan's per-(device, stream)
// workspace, then the tree walk (twists_kernel)
template <class T>
int twists(const grbda_plan *p, const T *q, const T *qd, const T *ydd, T *V, size_t B, int device, void *stream)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    GRBDA_CALL_SCOPE(p);
    if (!q || !qd || !ydd || !V) return set_err(GRBDA_EINVAL, "null argument");
    if (B == 0) return GRBDA_OK;
    DeviceTables *t = nullptr;
    if (int rc = ensure_device(p, device, &t)) return rc;
    const size_t ns = static_cast<size_t>(span_count(p));
    void *wptr = nullptr;
    if (int rc = ensure_work(p, p->work, device, stream, 2 * B * ns * sizeof(T) + 256, &wptr)) return rc;
    T *vs = static_cast<T *>(wptr), *as = vs + B * ns;
    if (int rc = spanning<T>(p, q, qd, ydd, vs, as, B, device, stream)) return rc;
    DevPlan<T> d = make_dev_plan<T>(p, *t, false, false);
    const size_t n_tiles = (B + kWave - 1) / kWave;
    const size_t g = n_tiles < static_cast<size_t>(t->n_cu) * 8 ? n_tiles : static_cast<size_t>(t->n_cu) * 8;
    hipError_t e = launch_twists<T>(d, p->host.n_clusters, static_cast<int>(ns), q, vs, as, V, B, static_cast<int>(g),
                                    static_cast<hipStream_t>(stream));
    return e == hipSuccess ? GRBDA_OK : hip_err(e, "twists launch");
}

// world wrench (about the world origin) of a Cartesian force at a point fixed in body `body`
template <class T>
__global__ void wrench_kernel(const T *__restrict__ Xa, const T *__restrict__ force, int n_bodies, int body, T ox, T oy,
                              T oz, size_t nb, T *__restrict__ fext)
{
    for (size_t b = blockIdx.x * (size_t)blockDim.x + threadIdx.x; b < nb; b += (size_t)gridDim.x * blockDim.x) {
        const T *X = Xa + (b * n_bodies + body) * 12;
        // p = r + E^T offset (E: world -> body)
        const T px = X[9] + X[0] * ox + X[3] * oy + X[6] * oz;
        const T py = X[10] + X[1] * ox + X[4] * oy + X[7] * oz;
        const T pz = X[11] + X[2] * ox + X[5] * oy + X[8] * oz;
        const T fx = force[3 * b], fy = force[3 * b + 1], fz = force[3 * b + 2];
        T *w = fext + b * (size_t)n_bodies * 6;
        for (int i = 0; i < n_bodies * 6; i++) w[i] = 0;
        w += (size_t)body * 6;
        w[0] = py * fz - pz * fy;
        w[1] = pz * fx - px * fz;
        w[2] = px * fy - py * fx;
        w[3] = fx;
        w[4] = fy;
        w[5] = fz;
    }
}
// dstate = a1 - a0 ;  lambda_inv = (t0 - t1) . dstate
template <class T>
__global__ void test_force_finish(const T *__restrict__ a1, const T *__restrict__ a0, const T *__restrict__ t1,
                                  const T *__restrict__ t0, int nv, size_t nb, T *__restrict__ dstate,
                                  T *__restrict__ lambda_inv)
{
    for (size_t b = blockIdx.x * (size_t)blockDim.x + threadIdx.x; b < nb; b += (size_t)gridDim.x * blockDim.x) {
        T s = 0;
        for (int i = 0; i < nv; i++) {
            const T d = a1[b * nv + i] - a0[b * nv + i];
            dstate[b * nv + i] = d;
            s += (t0[b * nv + i] - t1[b * nv + i]) * d;
        }
        lambda_inv[b] = s;
    }
}

template <class T>
int inv_osim_chain(const grbda_plan *p, const T *q, int n_contacts, const int *bodies, const double *offsets, T *Linv, T *J,
                   size_t B, int device, void *stream, const T *tf_force, T *tf_lambda, T *tf_dstate);

template <class T>
int test_force(const grbda_plan *p, const T *q, int body, const double *offset, const T *force, T *lambda_inv, T *dstate,
               size_t B, int device, void *stream)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    GRBDA_CALL_SCOPE(p);
    if (!p || !q || !offset || !force || !lambda_inv || !dstate) return set_err(GRBDA_EINVAL, "null argument");
    if (body < 0 || body >= p->host.n_bodies) return set_err(GRBDA_EINVAL, "body index out of range");
    if (B == 0) return GRBDA_OK;
    {   // models the chain program covers: one launch of the force-propagation kernel (osim_chain_kernel, applyTestForce mode)
        const int rc = inv_osim_chain<T>(p, q, 1, &body, offset, nullptr, nullptr, B, device, stream, force, lambda_inv, dstate);
        if (rc != 1) return rc;
    }
    DeviceTables *t = nullptr;
    if (int rc = ensure_device(p, device, &t)) return rc;
    const size_t nq = p->host.nq, nv = p->host.nv, nbod = p->host.n_bodies;
    const size_t per_state = nbod * 18 + 5 * nv;  // poses, wrenches, zeros, four results
    size_t chunk = (256u << 20) / (per_state * sizeof(T));
    if (chunk < 1) chunk = 1;
    if (chunk > B) chunk = B;
    void *wptr = nullptr;
    if (int rc = ensure_work(p, p->work, device, stream, chunk * per_state * sizeof(T) + 256, &wptr)) return rc;
    T *Xa = static_cast<T *>(wptr);
    T *fext = Xa + chunk * nbod * 12;
    T *zero = fext + chunk * nbod * 6;
    T *a1 = zero + chunk * nv, *a0 = a1 + chunk * nv, *t1 = a0 + chunk * nv, *t0 = t1 + chunk * nv;
    hipStream_t hs = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(zero, 0, chunk * nv * sizeof(T), hs);
    if (e != hipSuccess) return hip_err(e, "hipMemsetAsync");
    for (size_t b0 = 0; b0 < B; b0 += chunk) {
        const size_t nb = B - b0 < chunk ? B - b0 : chunk;
        const T *qc = q + b0 * nq;
        if (int rc = poses<T>(p, qc, Xa, nb, device, stream)) return rc;
        const int blocks = static_cast<int>((nb + 255) / 256 < 65535 ? (nb + 255) / 256 : 65535);
        hipLaunchKernelGGL((wrench_kernel<T>), dim3(blocks), dim3(256), 0, hs, Xa, force + 3 * b0, static_cast<int>(nbod),
                           body, static_cast<T>(offset[0]), static_cast<T>(offset[1]), static_cast<T>(offset[2]), nb, fext);
        if ((e = hipGetLastError()) != hipSuccess) return hip_err(e, "wrench launch");
        int rc;
        if ((rc = run<T>(p, false, qc, zero, zero, fext, a1, nb, device, stream)) ||
            (rc = run<T>(p, false, qc, zero, zero, nullptr, a0, nb, device, stream)) ||
            (rc = run<T>(p, true, qc, zero, zero, fext, t1, nb, device, stream)) ||
            (rc = run<T>(p, true, qc, zero, zero, nullptr, t0, nb, device, stream)))
            return rc;
        hipLaunchKernelGGL((test_force_finish<T>), dim3(blocks), dim3(256), 0, hs, a1, a0, t1, t0, static_cast<int>(nv), nb,
                           dstate + b0 * nv, lambda_inv + b0);
        if ((e = hipGetLastError()) != hipSuccess) return hip_err(e, "finish launch");
    }
    return GRBDA_OK;
}

// ---- inverse operational-space inertia of a set of contact frames (include/grbda_hip.h) -----------------------
constexpr int kMaxContacts = 8;
template <class T>
struct ContactSet {
    int n;
    int body[kMaxContacts];
    T off[kMaxContacts][3];
};

// rows (b, j), j < 6 n: unit spatial force e_{j % 6} in contact frame j / 6 (body axes, origin at the contact
// point) as a world wrench on that body; row 6 n: no force
template <class T>
__global__ void osim_expand_kernel(ContactSet<T> cs, const T *__restrict__ q, const T *__restrict__ Xa, int nq,
                                   int n_bodies, size_t nb, T *__restrict__ qx, T *__restrict__ fext)
{
    const int R = 6 * cs.n + 1;
    const size_t rows = nb * (size_t)R;
    for (size_t row = blockIdx.x * (size_t)blockDim.x + threadIdx.x; row < rows; row += (size_t)gridDim.x * blockDim.x) {
        const size_t b = row / R;
        const int j = (int)(row % R);
        for (int i = 0; i < nq; i++) qx[row * nq + i] = q[b * nq + i];
        T *w = fext + row * (size_t)n_bodies * 6;
        for (int i = 0; i < n_bodies * 6; i++) w[i] = 0;
        if (j == R - 1) continue;
        const int c = j / 6, k = j % 6;
        const T *X = Xa + (b * n_bodies + cs.body[c]) * 12;
        const T ox = cs.off[c][0], oy = cs.off[c][1], oz = cs.off[c][2];
        const T p[3] = {X[9] + X[0] * ox + X[3] * oy + X[6] * oz, X[10] + X[1] * ox + X[4] * oy + X[7] * oz,
                        X[11] + X[2] * ox + X[5] * oy + X[8] * oz};
        const int a = k % 3;
        const T e[3] = {X[3 * a], X[3 * a + 1], X[3 * a + 2]};  // E^T e_a: body axis a in world coordinates
        w += (size_t)cs.body[c] * 6;
        if (k < 3) {  // unit moment
            w[0] = e[0]; w[1] = e[1]; w[2] = e[2];
        } else {      // unit force at p
            w[0] = p[1] * e[2] - p[2] * e[1];
            w[1] = p[2] * e[0] - p[0] * e[2];
            w[2] = p[0] * e[1] - p[1] * e[0];
            w[3] = e[0]; w[4] = e[1]; w[5] = e[2];
        }
    }
}
// g_i = tau(no force) - tau(w_i) = J^T e_i,  a_j = ydd(w_j) - ydd(no force) = H^-1 J^T e_j,  Linv[i][j] = g_i . a_j
template <class T>
__global__ void osim_combine_kernel(const T *__restrict__ acc, const T *__restrict__ tau, int nv, int m, size_t nb,
                                    T *__restrict__ Linv, T *__restrict__ J)
{
    const size_t total = nb * (size_t)m * m;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t b = t / ((size_t)m * m);
        const int i = (int)((t / m) % m), j = (int)(t % m);
        const T *ab = acc + b * (size_t)(m + 1) * nv, *tb = tau + b * (size_t)(m + 1) * nv;
        T s = 0;
        for (int v = 0; v < nv; v++) {
            const T g = tb[(size_t)m * nv + v] - tb[(size_t)i * nv + v];
            s += g * (ab[(size_t)j * nv + v] - ab[(size_t)m * nv + v]);
            if (J && j == 0) J[(b * m + i) * nv + v] = g;
        }
        Linv[t] = s;
    }
}

// Inverse OSIM by force propagation (chain_kernels.hip, osim_chain_kernel) for models the chain program covers and contact
// frames on link / base bodies.  Returns 1 when the fast path does not apply (the caller then takes the unit-wrench path).
template <class T>
int inv_osim_chain(const grbda_plan *p, const T *q, int n_contacts, const int *bodies, const double *offsets, T *Linv, T *J,
                   size_t B, int device, void *stream, const T *tf_force, T *tf_lambda, T *tf_dstate)
{
    const HostPlan &h = p->host;
    const ChainProgram &cp = sizeof(T) == 8 ? h.chain64 : h.chain32;
    // (programs with generic clusters -- plan.h, ChainGen -- have no walk steps in the force-propagation kernel: unit-wrench path)
    if (p->no_chain || p->no_efpa || !cp.ok || !cp.gens.empty() || n_contacts > kOsimMaxContacts) return 1;
    const Layout &L = h.lay64;
    OsimArgs<T> A;
    std::memset(&A, 0, sizeof A);
    A.n_contacts = n_contacts;
    A.want_J = J ? 1 : 0;
    A.test_force = tf_force ? 1 : 0;
    A.force = tf_force;
    A.lambda_inv = tf_lambda;
    A.dstate = tf_dstate;
    if (tf_force && n_contacts != 1) return 1;
    std::vector<std::vector<int>> path_clusters(n_contacts);
    int max_rows = 0;
    for (int e = 0; e < n_contacts; e++) {
        const int b = bodies[e];
        int c = h.crba.bodies[b].cluster;
        int rows = 0, len = 0;
        bool first = true;
        while (c >= 0) {
            if (len >= kOsimMaxPath) return 1;
            const ClusterRec &cr = L.clusters[c];
            OsimStep st;
            st.v_index = static_cast<int16_t>(cr.v_index);
            st.w_row = static_cast<int16_t>(rows);
            int found = -1;
            if (cr.kind == CK_FREE) {
                for (size_t i = 0; i < cp.frees.size(); i++)
                    if (cp.frees[i].v_index == cr.v_index) found = static_cast<int>(i);
                st.kind = OSIM_FREE;
                rows += 6;
            } else if (cr.shape != SHAPE_GENERIC) {
                if (first && b != cr.link_body) return 1;  // a contact on a rotor
                for (size_t i = 0; i < cp.links.size(); i++)
                    if (cp.links[i].v_index == cr.v_index) found = static_cast<int>(i);
                st.kind = OSIM_LINK;
                rows += 1;
            } else if (cr.kind == CK_LOOP || [&] {
                           for (const ChainDiff &df : cp.diffs)
                               if (df.v_index == cr.v_index) return true;
                           return false;
                       }()) {
                // two-rotor differential, or an explicit pair that runs through its segments: the path enters at link1 (a
                // contact on it) or at link2 (a contact on it or below)
                for (size_t i = 0; i < cp.diffs.size(); i++)
                    if (cp.diffs[i].v_index == cr.v_index) found = static_cast<int>(i);
                if (found < 0) return 1;
                if (!first || L.bodies[b].cofs == cp.diffs[found].cofs[1]) st.kind = OSIM_DIFF_LINK2;
                else if (L.bodies[b].cofs == cp.diffs[found].cofs[0]) st.kind = OSIM_DIFF_LINK1;
                else return 1;  // a contact on a rotor
                rows += 2;
            } else {
                if (!first) return 1;  // pair clusters are leaves of the chain program
                for (size_t i = 0; i < cp.pairs.size(); i++)
                    if (cp.pairs[i].v_index == cr.v_index) found = static_cast<int>(i);
                if (found < 0) return 1;
                if (L.bodies[b].cofs == cp.pairs[found].cofs[0]) st.kind = OSIM_PAIR_LINK1;
                else if (L.bodies[b].cofs == cp.pairs[found].cofs[1]) st.kind = OSIM_PAIR_LINK2;
                else return 1;
                rows += 2;
            }
            if (found < 0) return 1;
            st.rec = static_cast<int16_t>(found);
            A.path[e][len++] = st;
            path_clusters[e].push_back(c);
            first = false;
            c = cr.parent_body >= 0 ? h.crba.bodies[cr.parent_body].cluster : -1;
        }
        A.path_len[e] = len;
        A.n_rows[e] = rows;
        if (rows > max_rows) max_rows = rows;
        // K0: wrench on the contact body, in the PLAN's body frame (canonical joint axes, plan.cpp), per unit contact wrench
        // given in the reference's body axes at the contact point: [n; f] -> [n + o x f; f], then the cyclic permutation
        const double *o = offsets + 3 * e;
        double W[36] = {0};
        for (int j = 0; j < 3; j++) {
            W[6 * j + j] = 1.0;            // unit moment e_j
            W[6 * (3 + j) + 3 + j] = 1.0;  // unit force e_j ...
        }
        // ... and its moment about the body origin o x e_j (column 3 + j, rows 0..2)
        W[6 * 1 + 3] = o[2];  W[6 * 2 + 3] = -o[1];   // o x e_x = (0, o_z, -o_y)
        W[6 * 0 + 4] = -o[2]; W[6 * 2 + 4] = o[0];    // o x e_y = (-o_z, 0, o_x)
        W[6 * 0 + 5] = o[1];  W[6 * 1 + 5] = -o[0];   // o x e_z = (o_y, -o_x, 0)
        const int ca = L.bodies[b].canon_axis;  // v_plan = Rc v_ref: x -> z: rows (y, z, x); y -> z: rows (z, x, y)
        const int perm[3][3] = {{1, 2, 0}, {2, 0, 1}, {0, 1, 2}};
        if (e == 0)
            for (int i = 0; i < 3; i++) A.perm[i] = perm[ca][i];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 6; j++) {
                A.K0[e][6 * i + j] = static_cast<T>(W[6 * perm[ca][i] + j]);
                A.K0[e][6 * (3 + i) + j] = static_cast<T>(W[6 * (3 + perm[ca][i]) + j]);
            }
    }
    for (int e1 = 0; e1 < n_contacts; e1++)
        for (int e2 = 0; e2 < n_contacts; e2++) {
            const std::vector<int> &a = path_clusters[e1], &b2 = path_clusters[e2];
            int rows = 0;
            size_t i = a.size(), j = b2.size();
            while (i > 0 && j > 0 && a[i - 1] == b2[j - 1]) {
                const ClusterRec &cr = L.clusters[a[i - 1]];
                rows += cr.kind == CK_FREE ? 6 : cr.n;
                i--;
                j--;
            }
            A.common[e1][e2] = rows;
        }
    DeviceTables *t = nullptr;
    if (int rc = ensure_device(p, device, &t)) return rc;
    const int w = sizeof(T) == 8 ? 2 : 0;
    const int kid = sizeof(T) == 8 ? 1 : 0;
    A.w_base = cp.n_glb;
    A.w_stride = 6 * max_rows;
    ChainDev<T> d;
    d.bad_count = t->bad_count;
    d.segs = t->chain_segs[w];
    d.links = t->chain_links[w];
    d.pairs = t->chain_pairs[w];
    d.frees = t->chain_frees[w];
    d.diffs = t->chain_diffs[w];
    d.n_diffs = static_cast<int>(cp.diffs.size());
    d.gens = nullptr;
    d.gbodies = nullptr;
    d.n_gens = 0;
    d.cints = t->cints;
    d.consts = sizeof(T) == 4 ? reinterpret_cast<const T *>(t->consts32) : reinterpret_cast<const T *>(t->consts64);
    d.n_segs = static_cast<int>(cp.segs.size());
    d.nq = h.nq;
    d.nv = h.nv;
    d.n_glb_slots = cp.n_glb + n_contacts * A.w_stride;
    d.ori_repr = h.ori_repr;
    d.debug = 0;
    d.sv_global = cp.sv_global ? 1 : 0;
    d.out_lds = -1;  // (the force-propagation kernel keeps its result rows in the slab)
    d.fuse = d.stage_lds_v = d.stage_v_index = 0;
    // (gravity enters the acceleration sweep only, which runs in applyTestForce mode alone -- there without it)
    for (int i = 0; i < 6; i++) d.a_root[i] = tf_force ? T(0) : static_cast<T>(-h.gravity[i]);
    const size_t n_tiles = (B + kWave - 1) / kWave;
    size_t grid = static_cast<size_t>(t->n_cu) * 4;  // one wavefront per SIMD: the walk kernel is not register-tuned
    if (grid > n_tiles) grid = n_tiles;
    size_t lds_bytes = static_cast<size_t>(cp.n_lds) * kWave * sizeof(T);
    const size_t stage_one = static_cast<size_t>(kWave) * static_cast<size_t>(d.nq > d.nv ? d.nq : d.nv) * sizeof(T);
    const size_t stage_all = static_cast<size_t>(kWave) * static_cast<size_t>(d.nq + 2 * d.nv) * sizeof(T);
    if (lds_bytes < stage_one) lds_bytes = stage_one;
    if (lds_bytes < stage_all && stage_all <= static_cast<size_t>(p->lds_bytes_per_wave[kid])) lds_bytes = stage_all;
    d.lds_bytes = static_cast<int>(lds_bytes);
    const size_t n_rows = static_cast<size_t>(d.n_glb_slots) + static_cast<size_t>(d.nq + 2 * d.nv);
    void *scratch = nullptr;
    if (int rc = ensure_scratch(p, device, stream, grid * n_rows * kWave * sizeof(T) + 256, &scratch)) return rc;
    // a block of zeros stands in for the velocities and torques of every tile
    Scratch &zs = p->work[{device, stream}];
    const size_t zneed = B * static_cast<size_t>(h.nv) * sizeof(T) + 256;
    hipError_t e;
    if (zs.bytes < zneed) {
        if (zs.ptr && (e = hipFree(zs.ptr)) != hipSuccess) return hip_err(e, "hipFree");
        zs.ptr = nullptr;
        zs.bytes = 0;
        if ((e = hipMalloc(&zs.ptr, zneed)) != hipSuccess) return hip_err(e, "hipMalloc(work)");
        zs.bytes = zneed;
    }
    hipStream_t hs = static_cast<hipStream_t>(stream);
    if ((e = hipMemsetAsync(zs.ptr, 0, zneed, hs)) != hipSuccess) return hip_err(e, "hipMemsetAsync");
    e = launch_osim_chain<T>(d, A, q, static_cast<const T *>(zs.ptr), Linv, J, B, static_cast<T *>(scratch), static_cast<int>(grid),
                             lds_bytes, hs);
    return e == hipSuccess ? GRBDA_OK : hip_err(e, "osim chain launch");
}

template <class T>
int inv_osim(const grbda_plan *p, const T *q, int n_contacts, const int *bodies, const double *offsets, T *Linv, T *J,
             size_t B, int device, void *stream)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    GRBDA_CALL_SCOPE(p);
    if (!p || !q || !bodies || !offsets || !Linv) return set_err(GRBDA_EINVAL, "null argument");
    if (n_contacts < 1 || n_contacts > kMaxContacts) return set_err(GRBDA_EINVAL, "1..8 contact frames per call");
    ContactSet<T> cs;
    cs.n = n_contacts;
    for (int c = 0; c < n_contacts; c++) {
        if (bodies[c] < 0 || bodies[c] >= p->host.n_bodies) return set_err(GRBDA_EINVAL, "body index out of range");
        cs.body[c] = bodies[c];
        for (int i = 0; i < 3; i++) cs.off[c][i] = static_cast<T>(offsets[3 * c + i]);
    }
    if (B == 0) return GRBDA_OK;
    {
        const int rc = inv_osim_chain<T>(p, q, n_contacts, bodies, offsets, Linv, J, B, device, stream, nullptr, nullptr, nullptr);
        if (rc != 1) return rc;
    }
    DeviceTables *t = nullptr;
    if (int rc = ensure_device(p, device, &t)) return rc;
    const size_t nq = p->host.nq, nv = p->host.nv, nbod = p->host.n_bodies;
    const size_t m = 6 * static_cast<size_t>(n_contacts), R = m + 1;
    const size_t per_state = nbod * 12 + R * (nq + nbod * 6 + 3 * nv);  // poses; per row q, wrenches, zeros, 2 results
    size_t chunk = (256u << 20) / (per_state * sizeof(T));
    if (chunk < 1) chunk = 1;
    if (chunk > B) chunk = B;
    void *wptr = nullptr;
    if (int rc = ensure_work(p, p->work, device, stream, chunk * per_state * sizeof(T) + 256, &wptr)) return rc;
    const size_t rows = chunk * R;
    T *Xa = static_cast<T *>(wptr);
    T *qx = Xa + chunk * nbod * 12;
    T *fext = qx + rows * nq;
    T *zero = fext + rows * nbod * 6;
    T *acc = zero + rows * nv, *tau = acc + rows * nv;
    hipStream_t hs = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(zero, 0, rows * nv * sizeof(T), hs);
    if (e != hipSuccess) return hip_err(e, "hipMemsetAsync");
    for (size_t b0 = 0; b0 < B; b0 += chunk) {
        const size_t nb = B - b0 < chunk ? B - b0 : chunk;
        const size_t nrows = nb * R;
        if (int rc = poses<T>(p, q + b0 * nq, Xa, nb, device, stream)) return rc;
        int blocks = static_cast<int>((nrows + 255) / 256 < 65535 ? (nrows + 255) / 256 : 65535);
        hipLaunchKernelGGL((osim_expand_kernel<T>), dim3(blocks), dim3(256), 0, hs, cs, q + b0 * nq, Xa, static_cast<int>(nq),
                           static_cast<int>(nbod), nb, qx, fext);
        if ((e = hipGetLastError()) != hipSuccess) return hip_err(e, "expand launch");
        int rc;
        if ((rc = run<T>(p, false, qx, zero, zero, fext, acc, nrows, device, stream)) ||
            (rc = run<T>(p, true, qx, zero, zero, fext, tau, nrows, device, stream)))
            return rc;
        const size_t tot = nb * m * m;
        blocks = static_cast<int>((tot + 255) / 256 < 65535 ? (tot + 255) / 256 : 65535);
        hipLaunchKernelGGL((osim_combine_kernel<T>), dim3(blocks), dim3(256), 0, hs, acc, tau, static_cast<int>(nv),
                           static_cast<int>(m), nb, Linv + b0 * m * m, J ? J + b0 * m * nv : nullptr);
        if ((e = hipGetLastError()) != hipSuccess) return hip_err(e, "combine launch");
    }
    return GRBDA_OK;
}

// ---- derived quantities: expanded batches over the two kernels (include/grbda_hip.h) ---------------------
enum DerivedMode { DM_BIAS = 0, DM_MASS = 1, DM_DTAU = 2, DM_DQD = 3, DM_DQ = 4 };

// row (b, j) of the expanded batch: state b with the j-th unit vector (or none) applied
// position col of state q0 after the tangent step `d` along velocity coordinate k (testHelpers.hpp:50-112)
template <class T>
__device__ T perturbed_position(const T *q0, int col, const int32_t *map, int k, T d)
{
    const int kind = map[3 * k], qi = map[3 * k + 1], a = map[3 * k + 2];
    const T x = q0[col];
    if (kind == 0) return col == qi ? x + d : x;
    if (kind == 1) {  // quat (scalar first, positions qi+3 .. qi+6) += quat x (0, d e_a) / 2
        if (col < qi + 3 || col > qi + 6) return x;
        const T w = q0[qi + 3], v[3] = {q0[qi + 4], q0[qi + 5], q0[qi + 6]};
        const int i = col - qi - 3;
        if (i == 0) return x - T(0.5) * d * v[a];
        const int j = i - 1;  // vector component: w e_a + v x e_a
        T p = j == a ? w : T(0);
        if (j == (a + 1) % 3) p += v[(a + 2) % 3];   // (v x e_a)_{a+1} = v_{a+2}
        if (j == (a + 2) % 3) p -= v[(a + 1) % 3];   // (v x e_a)_{a+2} = -v_{a+1}
        return x + T(0.5) * d * p;
    }
    if (kind == 2) {  // pos += R(quat)^T d e_a: component i gets R[a][i] (OrientationTools.h:251-269)
        if (col < qi || col > qi + 2) return x;
        const T e0 = q0[qi + 3], e1 = q0[qi + 4], e2 = q0[qi + 5], e3 = q0[qi + 6];
        const T M[9] = {1 - 2 * (e2 * e2 + e3 * e3), 2 * (e1 * e2 - e0 * e3), 2 * (e1 * e3 + e0 * e2),
                        2 * (e1 * e2 + e0 * e3), 1 - 2 * (e1 * e1 + e3 * e3), 2 * (e2 * e3 - e0 * e1),
                        2 * (e1 * e3 - e0 * e2), 2 * (e2 * e3 + e0 * e1), 1 - 2 * (e1 * e1 + e2 * e2)};
        const int i = col - qi;
        return x + d * M[3 * i + a];  // R = M^T: R[a][i] = M[i][a]
    }
    return x;
}

template <class T>
__global__ void expand_kernel(int mode, const T *__restrict__ q, const T *__restrict__ qd, const T *__restrict__ tau,
                              int nq, int nv, int R, size_t nb, T *__restrict__ qx, T *__restrict__ qdx,
                              T *__restrict__ xx, const int32_t *__restrict__ dq_map, T step)
{
    const size_t rows = nb * (size_t)R;
    const int w = nq + 2 * nv;
    const size_t total = rows * (size_t)w;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / w;
        const int col = (int)(i % w);
        const size_t b = row / R;
        const int j = (int)(row % R);
        if (col < nq) {
            qx[row * nq + col] = mode == DM_DQ ? perturbed_position(q + b * nq, col, dq_map, j >> 1, (j & 1) ? -step : step)
                                               : q[b * nq + col];
        } else if (col < nq + nv) {
            const int k = col - nq;
            T v = 0;
            if (mode == DM_BIAS || mode == DM_DQ) v = qd[b * nv + k];
            else if (mode == DM_DQD) v = qd[b * nv + k] + ((j >> 1) == k ? ((j & 1) ? T(-1) : T(1)) : T(0));
            qdx[row * nv + k] = v;
        } else {
            const int k = col - nq - nv;
            T v = 0;
            if (mode == DM_MASS || mode == DM_DTAU) v = (j == k) ? T(1) : T(0);
            else if (mode == DM_DQD || mode == DM_DQ) v = tau[b * nv + k];
            xx[row * nv + k] = v;
        }
    }
}

template <class A, class Bt>
__global__ void convert_kernel(const A *__restrict__ src, Bt *__restrict__ dst, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = static_cast<Bt>(src[i]);
}

// out[b][i][j] from the kernel results r[(b, j)][i]
template <class T>
__global__ void combine_kernel(int mode, const T *__restrict__ r, int nv, int R, size_t nb, T *__restrict__ out,
                               T step)
{
    const size_t total = nb * (size_t)nv * nv;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const size_t b = t / ((size_t)nv * nv);
        const int i = (int)((t / nv) % nv), j = (int)(t % nv);
        const T *rb = r + b * (size_t)R * nv;
        T v;
        if (mode == DM_DQD || mode == DM_DQ)
            v = (rb[(size_t)(2 * j) * nv + i] - rb[(size_t)(2 * j + 1) * nv + i]) / (T(2) * (mode == DM_DQ ? step : T(1)));
        else v = rb[(size_t)j * nv + i] - rb[(size_t)nv * nv + i];
        out[t] = v;
    }
}

template <class T>
int derived(const grbda_plan *p, int mode, const T *q, const T *qd, const T *tau, const T *f_ext, T *out, size_t B,
            int device, void *stream, double step = 1.0)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    GRBDA_CALL_SCOPE(p);
    if (!p || !q || !out) return set_err(GRBDA_EINVAL, "null argument");
    if ((mode == DM_BIAS || mode == DM_DQD || mode == DM_DQ) && !qd) return set_err(GRBDA_EINVAL, "null argument");
    if ((mode == DM_DQD || mode == DM_DQ) && !tau) return set_err(GRBDA_EINVAL, "null argument");
    bool reproject = false;
    if (mode == DM_DQ) {
        if (!(step > 0)) return set_err(GRBDA_EINVAL, "step must be positive");
        for (const ClusterRec &c : p->host.lay64.clusters) reproject |= c.kind == CK_LOOP;
    }
    if (B == 0) return GRBDA_OK;
    DeviceTables *t = nullptr;
    if (int rc = ensure_device(p, device, &t)) return rc;
    const int nq = p->host.nq, nv = p->host.nv;
    if (mode == DM_MASS && p->host.crba.ok && !p->no_crba) {
        // composite-rigid-body kernel (crba_kernels.hip): one launch instead of nv + 1 inverse-dynamics evaluations
        DevPlan<T> d = make_dev_plan<T>(p, *t, false, false);
        const size_t n_tiles = (B + kWave - 1) / kWave;
        const size_t crba_waves = static_cast<size_t>(sizeof(T) == 4 ? p->crba_waves : std::min(p->crba_waves, 8));  // (fp64: 256 registers, two per SIMD)
        size_t grid = static_cast<size_t>(t->n_cu) * crba_waves;
        if (grid > n_tiles) grid = n_tiles;
        void *scratch = nullptr;
        if (int rc = ensure_scratch(p, device, stream, grid * static_cast<size_t>(p->host.crba.n_rows) * kWave * sizeof(T) + 256, &scratch))
            return rc;
        hipStream_t hs = static_cast<hipStream_t>(stream);
        hipError_t e;
        if (t->deriv_related) {
            // the lower triangle in packed rows (row-local stores), then unpacked in place, one wavefront per state
            // (JVRC-1, 131 072 states, f32: 1.10 against 1.65 ms for the plain layout; f64 about even)
            // whole groups of kDerivGroup states interleaved (crba_kernels.hip: a quarter of the open cache lines per store), the tail
            // of the batch state-major
            const int il = unpack_symmetric_lds_bytes(nv, sizeof(T), kDerivGroup) <= 60 * 1024 && !env_int("GRBDA_CRBA_STATE_MAJOR", 0) ? kDerivGroup : 1;
            const size_t Bg = il > 1 ? B / il * il : 0;
            for (int part = 0; part < 2; part++) {
                const size_t b0 = part == 0 ? 0 : Bg, nbp = part == 0 ? Bg : B - Bg;
                if (nbp == 0) continue;
                const int ilp = part == 0 ? il : 1;
                size_t gp = static_cast<size_t>(t->n_cu) * crba_waves;
                if (gp > (nbp + kWave - 1) / kWave) gp = (nbp + kWave - 1) / kWave;
                e = launch_crba<T>(d, t->crba_bodies, p->host.n_clusters, p->host.crba.n_rows, q + b0 * nq, out + b0 * static_cast<size_t>(nv) * nv, nbp,
                                   static_cast<T *>(scratch), static_cast<int>(gp), hs, true, ilp);
                if (e != hipSuccess) return hip_err(e, "crba launch");
                // (a persistent grid: no more workgroups than the LDS of a CU holds at once -- 12 of JVRC-1's 12.4 KB blocks, not 16)
                size_t g2 = static_cast<size_t>(t->n_cu) * std::min<size_t>(16, lds_workgroups_per_cu(unpack_symmetric_lds_bytes(nv, sizeof(T), ilp) + 512));
                if (g2 > nbp / ilp) g2 = nbp / ilp;
                e = launch_unpack_symmetric<T>(out + b0 * static_cast<size_t>(nv) * nv, t->deriv_related, nv, nbp, static_cast<int>(g2), hs, ilp);
                if (e != hipSuccess) return hip_err(e, "unpack launch");
            }
            return GRBDA_OK;
        }
        e = hipMemsetAsync(out, 0, B * static_cast<size_t>(nv) * nv * sizeof(T), hs);
        if (e != hipSuccess) return hip_err(e, "hipMemsetAsync");
        e = launch_crba<T>(d, t->crba_bodies, p->host.n_clusters, p->host.crba.n_rows, q, out, B, static_cast<T *>(scratch),
                           static_cast<int>(grid), hs, false, 1);
        return e == hipSuccess ? GRBDA_OK : hip_err(e, "crba launch");
    }
    const int R = mode == DM_BIAS ? 1 : ((mode == DM_DQD || mode == DM_DQ) ? 2 * nv : nv + 1);
    const size_t row_scalars = static_cast<size_t>(nq) + 3 * static_cast<size_t>(nv);  // q, qd, x, result
    size_t chunk = (256u << 20) / (row_scalars * sizeof(T) * static_cast<size_t>(R));
    if (chunk < 1) chunk = 1;
    if (chunk > B) chunk = B;
    const size_t rows = chunk * static_cast<size_t>(R);
    void *wptr = nullptr;
    if (int rc = ensure_work(p, p->work, device, stream, rows * row_scalars * sizeof(T) + 256, &wptr)) return rc;
    T *qx = static_cast<T *>(wptr);
    T *qdx = qx + rows * nq;
    T *xx = qdx + rows * nv;
    T *res = xx + rows * nv;
    hipStream_t hs = static_cast<hipStream_t>(stream);
    const bool via_rnea = mode == DM_BIAS || mode == DM_MASS;
    for (size_t b0 = 0; b0 < B; b0 += chunk) {
        const size_t nb = B - b0 < chunk ? B - b0 : chunk;
        const size_t nrows = nb * static_cast<size_t>(R);
        const size_t total = nrows * static_cast<size_t>(nq + 2 * nv);
        int blocks = static_cast<int>((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
        hipLaunchKernelGGL((expand_kernel<T>), dim3(blocks), dim3(256), 0, hs, mode, q + b0 * nq,
                           qd ? qd + b0 * nv : nullptr, tau ? tau + b0 * nv : nullptr, nq, nv, R, nb, qx, qdx, xx, t->dq_map,
                           static_cast<T>(step));
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_err(e, "expand launch");
        if (reproject) {
            // the perturbed states leave the constraint manifold of the implicit clusters by O(step): Newton puts the
            // dependent coordinates back (GenericJoint.cpp:289-385), starting one step away from the solution
            if (int rc = project<T>(p, qx, nullptr, nrows, 25, sizeof(T) == 8 ? 1e-13 : 1e-6, device, stream)) return rc;
        }
        const T *fe = (mode == DM_BIAS && f_ext) ? f_ext + b0 * static_cast<size_t>(p->host.n_bodies) * 6 : nullptr;
        T *dst = mode == DM_BIAS ? out + b0 * nv : res;
        if (int rc = run<T>(p, via_rnea, qx, qdx, xx, fe, dst, nrows, device, stream)) return rc;
        if (mode != DM_BIAS) {
            const size_t tot2 = nb * static_cast<size_t>(nv) * nv;
            blocks = static_cast<int>((tot2 + 255) / 256 < 65535 ? (tot2 + 255) / 256 : 65535);
            hipLaunchKernelGGL((combine_kernel<T>), dim3(blocks), dim3(256), 0, hs, mode, res, nv, R, nb,
                               out + b0 * static_cast<size_t>(nv) * nv, static_cast<T>(step));
            if ((e = hipGetLastError()) != hipSuccess) return hip_err(e, "combine launch");
        }
    }
    return GRBDA_OK;
}

// ---- analytic first-order derivatives of the forward dynamics (deriv_kernels.hip) ---------------------------------
// d ydd / d tau = H^-1, d ydd / d q = -H^-1 dID/dq, d ydd / d qd = -H^-1 dID/dqd at ydd = FD(q, qd, tau); any of the three
// outputs may be null.  Returns 1 when the model is not covered (implicit loops, nv > 64): the
// callers then fall back to the unit-vector / central-difference batches of derived().
template <class T>
bool analytic_covers(const grbda_plan *p)
{
    return p->host.deriv.ok && p->host.crba.ok && !p->no_analytic && !p->no_crba && p->host.nv <= kWave;
}
// Forward / inverse dynamics through the spanning tree (HostPlan::projection_only; the reference's Projection-method cross-check,
// RigidBodyTreeDynamics.cpp:86-97):  tau = G^T ID_s(q_s, G yd, G ydd + g);  ydd = (G^T H_s G)^-1 (tau - G^T ID_s(q_s, G yd, g)).
template <class T>
int projection_run(const grbda_plan *p, bool rnea, const T *q, const T *qd, const T *x, const T *f_ext, T *out, size_t B, int device, void *stream)
{
    if (!p->span)
        return set_err(GRBDA_EUNSUPPORTED, "the model needs the spanning-tree route, which covers at most 64 spanning velocities (128 for plans with big clusters)");
    DeviceTables *t = nullptr, *ts = nullptr;
    if (int rc = ensure_device(p, device, &t)) return rc;
    const grbda_plan *sp = p->span;
    if (int rc = ensure_device(sp, device, &ts)) return rc;
    const bool big = p->host.big_clusters;
    const size_t nq = p->host.nq, nv = p->host.nv, nn = nv * nv;
    const size_t nq_s = sp->host.nq, nv_s = sp->host.nv, nn_s = nv_s * nv_s;
    // (forward dynamics: H_s alone of the spanning recursion's three matrices is stored)
    const size_t per_state = nq_s + 3 * nv_s + static_cast<size_t>(p->n_cpl_rows) + (rnea ? 0 : nn_s + 2 * nn);  // (wide: nn + nv would do)
    // (plans with big clusters: 40-50 KB per state; a chunk that leaves most SIMDs without a tile costs more than the memory)
    size_t chunk = work_budget(p, p->work_proj, device, stream, (big ? 4096ull : 1024ull) << 20) / (per_state * sizeof(T));
    chunk &= ~static_cast<size_t>(kWave - 1);
    if (chunk < static_cast<size_t>(kWave)) chunk = kWave;
    const size_t b_round = (B + kWave - 1) / kWave * kWave;
    if (chunk > b_round) chunk = b_round;
    void *wptr = nullptr;
    if (int rc = ensure_work(p, p->work_proj, device, stream, chunk * per_state * sizeof(T) + 256, &wptr)) return rc;
    T *w = static_cast<T *>(wptr);
    auto take = [&](size_t per) { T *r = w; w += chunk * per; return r; };
    T *q_s = take(nq_s), *qd_s = take(nv_s), *qdd_s = take(nv_s), *x_s = take(nv_s), *cpl = take(p->n_cpl_rows);
    T *Aq = nullptr, *Av = nullptr, *Hs = rnea ? nullptr : take(nn_s);
    // (more than 64 velocities: related-coordinate TABLES instead of one-word masks, and the workgroup-per-state solve on the one right-hand
    // side instead of H^-1 -- manifold_kernels.hip, kernels 2w and 4)
    const bool wide_nv = nv > static_cast<size_t>(kWave) || nv_s > static_cast<size_t>(kWave);
    T *Hw = rnea ? nullptr : take(nn), *Hinv = rnea ? nullptr : take(wide_nv ? nv : nn);
    hipStream_t hs = static_cast<hipStream_t>(stream);
    DevPlan<T> d = make_dev_plan<T>(p, *t, false, false);
    DevPlan<T> ds = make_dev_plan<T>(sp, *ts, false, false);
    hipError_t e;
    for (size_t b0 = 0; b0 < B; b0 += chunk) {
        const size_t nb = B - b0 < chunk ? B - b0 : chunk;
        const size_t n_tiles = (nb + kWave - 1) / kWave;
        size_t grid = static_cast<size_t>(t->n_cu) * 4;
        if (grid > n_tiles) grid = n_tiles;
        // inverse dynamics: qdd_s = G ydd + g; forward dynamics: qdd_s = g (the bias of the spanning tree with the constraint's own acceleration)
        e = launch_manifold_constraint<T>(d, p->host.n_clusters, t->span_q, t->span_v, t->crow, static_cast<int>(nq_s), static_cast<int>(nv_s),
                                          p->n_cpl_rows, 0, q + b0 * nq, qd + b0 * nv, rnea ? x + b0 * nv : nullptr, q_s, qd_s, qdd_s, cpl, nb,
                                          static_cast<int>(grid), hs, p->constraint_shape, p->has_trig);
        if (e != hipSuccess) return hip_err(e, "manifold constraint launch");
        if (int rc = run<T>(sp, true, q_s, qd_s, qdd_s, f_ext ? f_ext + b0 * static_cast<size_t>(p->host.n_bodies) * 6 : nullptr, x_s, nb, device, stream))
            return rc;
        if (rnea) {
            e = launch_manifold_apply<T>(d, p->host.n_clusters, t->span_v, t->crow, static_cast<int>(nv_s), p->n_cpl_rows, 0, x_s, nullptr, nullptr,
                                         cpl, out + b0 * nv, nb, static_cast<int>(grid), hs, big);
            if (e != hipSuccess) return hip_err(e, "manifold apply launch");
            continue;
        }
        size_t g2 = static_cast<size_t>(ts->n_cu) * 4;
        if (g2 > n_tiles) g2 = n_tiles;
        void *scratch = nullptr;
        if (int rc = ensure_scratch(sp, device, stream, g2 * static_cast<size_t>(sp->host.deriv.n_rows) * kWave * sizeof(T) + 256, &scratch)) return rc;
        e = launch_rnea_deriv<T>(ds, ts->deriv_bodies, sp->host.n_clusters, sp->host.deriv.n_rows, sp->host.deriv.n_max, q_s, qd_s, qdd_s, Aq, Av, Hs,
                                 nb, static_cast<T *>(scratch), static_cast<int>(g2), hs, kWave);
        if (e != hipSuccess) return hip_err(e, "spanning derivative launch");
        if (wide_nv) {
            if (!t->related_table || !ts->related_table) return set_err(GRBDA_EUNSUPPORTED, "more than 64 velocities on a plan without big clusters");
            e = launch_manifold_project_wide<T>(d, p->host.n_clusters, t->span_v, t->crow, nullptr, nullptr, t->related_table, ts->related_table,
                                                static_cast<int>(nv_s), p->n_cpl_rows, Hs, cpl, Hw, nb, static_cast<int>(grid), hs);
            if (e != hipSuccess) return hip_err(e, "manifold projection launch");
            e = launch_manifold_apply<T>(d, p->host.n_clusters, t->span_v, t->crow, static_cast<int>(nv_s), p->n_cpl_rows, 2, x_s, x + b0 * nv, nullptr,
                                         cpl, Hinv, nb, static_cast<int>(grid), hs, big);
            if (e != hipSuccess) return hip_err(e, "manifold apply launch");
            e = launch_spd_wide_solve<T>(Hw, t->related_table, Hinv, out + b0 * nv, static_cast<int>(nv), nb, t->n_cu, hs, spd_bad_count_address());
            if (e != hipSuccess) return hip_err(e, "wide solve launch");
            continue;
        }
        e = launch_manifold_project<T>(d, p->host.n_clusters, t->span_v, t->crow, t->deriv_related, ts->deriv_related, static_cast<int>(nv_s),
                                       p->n_cpl_rows, 1, nullptr, nullptr, Hs, nullptr, cpl, nullptr, nullptr, Hw, nb, static_cast<int>(grid), hs, 1, big);
        if (e != hipSuccess) return hip_err(e, "manifold projection launch");
        const size_t lds = spd_solve_lds_bytes(static_cast<int>(nv), sizeof(T), 0);
        size_t per_cu = lds ? lds_workgroups_per_cu(lds) : 16;
        if (per_cu > 16) per_cu = 16;
        if (per_cu < 1) per_cu = 1;
        size_t g3 = static_cast<size_t>(t->n_cu) * per_cu;
        if (g3 > nb) g3 = nb;
        if constexpr (sizeof(T) == 4)
            e = launch_spd_solve<float, float>(Hw, 1, nullptr, nullptr, Hinv, nullptr, nullptr, t->deriv_related, static_cast<int>(nv), nb,
                                               static_cast<int>(g3), hs, 1);
        else
            e = launch_spd_solve<double, double>(Hw, 1, nullptr, nullptr, Hinv, nullptr, nullptr, t->deriv_related, static_cast<int>(nv), nb,
                                                 static_cast<int>(g3), hs, 1);
        if (e != hipSuccess) return hip_err(e, "spd solve launch");
        e = launch_manifold_apply<T>(d, p->host.n_clusters, t->span_v, t->crow, static_cast<int>(nv_s), p->n_cpl_rows, 1, x_s, x + b0 * nv, Hinv, cpl,
                                     out + b0 * nv, nb, static_cast<int>(grid), hs, big);
        if (e != hipSuccess) return hip_err(e, "manifold apply launch");
    }
    return GRBDA_OK;
}

int projection_run_f32_through_f64(const grbda_plan *p, bool rnea, const float *q, const float *qd, const float *x, const float *f_ext, float *out,
                                   size_t B, int device, void *stream)
{
    const size_t nq = p->host.nq, nv = p->host.nv, nfe = f_ext ? static_cast<size_t>(p->host.n_bodies) * 6 : 0;
    const size_t per_state = nq + 3 * nv + nfe;
    size_t chunk = (256u << 20) / (per_state * sizeof(double));
    if (chunk < 1) chunk = 1;
    if (chunk > B) chunk = B;
    void *cvt = nullptr;
    if (int rc = ensure_work(p, p->work_cvt, device, stream, chunk * per_state * sizeof(double) + 256, &cvt)) return rc;
    double *q64 = static_cast<double *>(cvt), *qd64 = q64 + chunk * nq, *x64 = qd64 + chunk * nv, *o64 = x64 + chunk * nv, *fe64 = o64 + chunk * nv;
    hipStream_t hs = static_cast<hipStream_t>(stream);
    auto blocks = [](size_t n) { return static_cast<int>((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535); };
    for (size_t b0 = 0; b0 < B; b0 += chunk) {
        const size_t nb = B - b0 < chunk ? B - b0 : chunk;
        hipLaunchKernelGGL((convert_kernel<float, double>), dim3(blocks(nb * nq)), dim3(256), 0, hs, q + b0 * nq, q64, nb * nq);
        hipLaunchKernelGGL((convert_kernel<float, double>), dim3(blocks(nb * nv)), dim3(256), 0, hs, qd + b0 * nv, qd64, nb * nv);
        hipLaunchKernelGGL((convert_kernel<float, double>), dim3(blocks(nb * nv)), dim3(256), 0, hs, x + b0 * nv, x64, nb * nv);
        if (f_ext) hipLaunchKernelGGL((convert_kernel<float, double>), dim3(blocks(nb * nfe)), dim3(256), 0, hs, f_ext + b0 * nfe, fe64, nb * nfe);
        if (int rc = projection_run<double>(p, rnea, q64, qd64, x64, f_ext ? fe64 : nullptr, o64, nb, device, stream)) return rc;
        hipLaunchKernelGGL((convert_kernel<double, float>), dim3(blocks(nb * nv)), dim3(256), 0, hs, o64, out + b0 * nv, nb * nv);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_err(e, "convert launch");
    }
    return GRBDA_OK;
}

// Models with implicit clusters (manifold_kernels.hip): ydd = FD; spanning state and the first-order parts of G, g per state;
// tau_s and (A_q, A_v, H_s) of the spanning tree from its own plan; projection with the per-state G; the same SPD solve.
// H only (dq == dqd == nullptr): qd and tau may be null.
template <class T>
bool manifold_covers(const grbda_plan *p)
{
    return p->span && !p->no_manifold && !p->no_analytic && p->host.nv <= kWave && p->span->host.nv <= kWave &&
           p->host.deriv.related.size() == static_cast<size_t>(p->host.nv);
}
template <class T>
int manifold_derivs(const grbda_plan *p, const T *q, const T *qd, const T *tau, T *dq, T *dqd, T *dtau, T *Hout, size_t B, int device,
                    void *stream)
{
    if (!q || ((dq || dqd) && (!qd || !tau))) return set_err(GRBDA_EINVAL, "null argument");
    if (B == 0 || (!dq && !dqd && !dtau && !Hout)) return GRBDA_OK;
    DeviceTables *t = nullptr, *ts = nullptr;
    if (int rc = ensure_device(p, device, &t)) return rc;
    const grbda_plan *sp = p->span;
    if (int rc = ensure_device(sp, device, &ts)) return rc;
    const size_t nq = p->host.nq, nv = p->host.nv, nn = nv * nv;
    const size_t nq_s = sp->host.nq, nv_s = sp->host.nv, nn_s = nv_s * nv_s;
    const bool need_d = dq || dqd;
    const bool big = p->host.big_clusters;
    // (clusters beyond the structured limits: no analytic d/dq, d/dqd -- the callers' difference batches take over: forward dynamics
    // differences for explicit clusters; implicit ones would need the Newton re-projection, which refuses such plans)
    if (big && need_d) return 1;
    const int n_rhs = (dq ? 1 : 0) + (dqd ? 1 : 0);
    const bool solve = need_d || dtau;
    const int il = (need_d && spd_solve_on_mfma(sizeof(T), static_cast<int>(nv), n_rhs)) ? kDerivGroup : 1;
    // workspace per state: spanning state (q_s, qd_s, qdd_s, tau_s), zeros for a missing qd, coupling rows, the three spanning
    // matrices, the three projected matrices, ydd
    const size_t per_state = nq_s + 3 * nv_s + nv + static_cast<size_t>(p->n_cpl_rows) + 3 * nn_s + 3 * nn + nv;
    // (16 GiB, cut to a quarter of the free memory: TelloWithArms takes 33 KB per state; a 4 GiB chunk -- 131 072 states, 2 048 tiles -- left half
    // of the projection kernel's wavefront slots empty: 13.3 -> 10.8 ms per 262 144 states, 55 -> 44 ms per 1 048 576)
    // (up to 16 GiB -- but no more than the batch itself needs: a small batch does not pin a large slab)
    size_t chunk = work_budget(p, p->work, device, stream, std::min<size_t>(16384ull << 20, ((B + kWave - 1) / kWave * kWave) * per_state * sizeof(T) + (1u << 20))) / (per_state * sizeof(T));
    chunk &= ~static_cast<size_t>(kWave - 1);
    if (chunk < static_cast<size_t>(kWave)) chunk = kWave;
    const size_t b_round = (B + kWave - 1) / kWave * kWave;
    if (chunk > b_round) chunk = b_round;
    void *wptr = nullptr;
    if (int rc = ensure_work(p, p->work, device, stream, chunk * per_state * sizeof(T) + 256, &wptr)) return rc;
    T *w = static_cast<T *>(wptr);
    auto take = [&](size_t per) { T *r = w; w += chunk * per; return r; };
    T *q_s = take(nq_s), *qd_s = take(nv_s), *qdd_s = take(nv_s), *tau_s = take(nv_s), *zeros = take(nv), *cpl = take(p->n_cpl_rows);
    T *Aq = take(nn_s), *Av = take(nn_s), *Hs = take(nn_s), *Dq = take(nn), *Dqd = take(nn), *Hw = take(nn), *ydd = take(nv);
    hipStream_t hs = static_cast<hipStream_t>(stream);
    DevPlan<T> d = make_dev_plan<T>(p, *t, false, false);
    DevPlan<T> ds = make_dev_plan<T>(sp, *ts, false, false);
    hipError_t e;
    for (size_t b0 = 0; b0 < B; b0 += chunk) {
        const size_t nb = B - b0 < chunk ? B - b0 : chunk;
        const size_t n_tiles = (nb + kWave - 1) / kWave;
        const T *qc = q + b0 * nq, *qdc = qd ? qd + b0 * nv : zeros, *yddc = nullptr;
        if (!qd && (e = hipMemsetAsync(zeros, 0, nb * nv * sizeof(T), hs)) != hipSuccess) return hip_err(e, "hipMemsetAsync");
        if (need_d) {
            if (int rc = run<T>(p, false, qc, qdc, tau + b0 * nv, nullptr, ydd, nb, device, stream)) return rc;
            yddc = ydd;
        } else {
            // H only: the recursion runs at zero velocity and acceleration (its H does not depend on them)
            if ((e = hipMemsetAsync(qd_s, 0, 2 * chunk * nv_s * sizeof(T), hs)) != hipSuccess) return hip_err(e, "hipMemsetAsync");
        }
        size_t grid = static_cast<size_t>(t->n_cu) * 4;
        if (grid > n_tiles) grid = n_tiles;
        e = launch_manifold_constraint<T>(d, p->host.n_clusters, t->span_q, t->span_v, t->crow, static_cast<int>(nq_s), static_cast<int>(nv_s),
                                          p->n_cpl_rows, need_d ? 1 : 0, qc, qdc, yddc, q_s, qd_s, need_d ? qdd_s : nullptr, cpl, nb,
                                          static_cast<int>(grid), hs, p->constraint_shape, p->has_trig);
        if (e != hipSuccess) return hip_err(e, "manifold constraint launch");
        if (!need_d && (e = hipMemsetAsync(qd_s, 0, chunk * nv_s * sizeof(T), hs)) != hipSuccess) return hip_err(e, "hipMemsetAsync");
        if (need_d)
            if (int rc = run<T>(sp, true, q_s, qd_s, qdd_s, nullptr, tau_s, nb, device, stream)) return rc;
        {
            const size_t deriv_waves = 4;
            size_t g2 = static_cast<size_t>(ts->n_cu) * deriv_waves;
            if (g2 > n_tiles) g2 = n_tiles;
            void *scratch = nullptr;
            if (int rc = ensure_scratch(sp, device, stream, g2 * static_cast<size_t>(sp->host.deriv.n_rows) * kWave * sizeof(T) + 256, &scratch))
                return rc;
            e = launch_rnea_deriv<T>(ds, ts->deriv_bodies, sp->host.n_clusters, sp->host.deriv.n_rows, sp->host.deriv.n_max, q_s, qd_s, qdd_s,
                                     need_d ? Aq : nullptr, need_d ? Av : nullptr, Hs, nb, static_cast<T *>(scratch), static_cast<int>(g2), hs, kWave);
            if (e != hipSuccess) return hip_err(e, "spanning derivative launch");
        }
        // the projected H goes to the caller's array when no solve follows, or (state-major layouts, whole groups) to d/dtau
        T *H = !solve ? Hout + b0 * nn : ((dtau && !(il > 1 && (B % kDerivGroup) != 0)) ? dtau + b0 * nn : Hw);
        e = launch_manifold_project<T>(d, p->host.n_clusters, t->span_v, t->crow, t->deriv_related, ts->deriv_related, static_cast<int>(nv_s),
                                       p->n_cpl_rows, need_d ? 0 : 1, Aq, Av, Hs, tau_s, cpl, need_d ? Dq : nullptr, need_d ? Dqd : nullptr, H,
                                       nb, static_cast<int>(grid), hs, solve ? il : 1, big);
        if (e != hipSuccess) return hip_err(e, "manifold projection launch");
        if (!solve) {
            // packed lower rows -> the full symmetric matrix, in place
            size_t g4 = static_cast<size_t>(t->n_cu) * std::min<size_t>(16, lds_workgroups_per_cu(unpack_symmetric_lds_bytes(static_cast<int>(nv), sizeof(T), 1) + 512));
            if (g4 > nb) g4 = nb;
            e = launch_unpack_symmetric<T>(H, t->deriv_related, static_cast<int>(nv), nb, static_cast<int>(g4), hs, 1);
            if (e != hipSuccess) return hip_err(e, "unpack launch");
            continue;
        }
        const size_t lds = spd_solve_lds_bytes(static_cast<int>(nv), sizeof(T), n_rhs);
        size_t per_cu = lds ? lds_workgroups_per_cu(lds) : 16;
        const bool mfma = need_d && spd_solve_on_mfma(sizeof(T), static_cast<int>(nv), n_rhs);
        if (per_cu > (mfma ? static_cast<size_t>(spd_mfma_workgroups_per_cu(static_cast<int>(nv))) : 16u)) per_cu = mfma ? spd_mfma_workgroups_per_cu(static_cast<int>(nv)) : 16;
        if (per_cu < 1) per_cu = 1;
        size_t g3 = static_cast<size_t>(t->n_cu) * per_cu;
        const size_t units = mfma ? (nb + kDerivGroup - 1) / kDerivGroup : nb;
        if (g3 > units) g3 = units;
        T *o1 = dq ? dq + b0 * nn : nullptr, *o2 = dqd ? dqd + b0 * nn : nullptr, *o3 = dtau ? dtau + b0 * nn : nullptr;
        const T *r1 = dq ? Dq : nullptr, *r2 = dqd ? Dqd : nullptr;
        if constexpr (sizeof(T) == 4)
            e = launch_spd_solve<float, float>(H, 1, r1, r2, o3, o1, o2, t->deriv_related, static_cast<int>(nv), nb, static_cast<int>(g3), hs, il);
        else
            e = launch_spd_solve<double, double>(H, 1, r1, r2, o3, o1, o2, t->deriv_related, static_cast<int>(nv), nb, static_cast<int>(g3), hs, 1);
        if (e != hipSuccess) return hip_err(e, "spd solve launch");
    }
    return GRBDA_OK;
}

template <class T>
int analytic_derivs(const grbda_plan *p, const T *q, const T *qd, const T *tau, T *dq, T *dqd, T *dtau, size_t B, int device,
                    void *stream)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    GRBDA_CALL_SCOPE(p);
    if (!analytic_covers<T>(p)) {
        if (manifold_covers<T>(p)) return manifold_derivs<T>(p, q, qd, tau, dq, dqd, dtau, nullptr, B, device, stream);
        return 1;
    }
    if (!q || ((dq || dqd) && (!qd || !tau))) return set_err(GRBDA_EINVAL, "null argument");
    if (B == 0 || (!dq && !dqd && !dtau)) return GRBDA_OK;
    DeviceTables *t = nullptr;
    if (int rc = ensure_device(p, device, &t)) return rc;
    const size_t nq = p->host.nq, nv = p->host.nv, nn = nv * nv;
    const bool need_d = dq || dqd;
    // H is built in the caller's d/dtau array when that is wanted (the factor is out of it before H^-1 goes in); dID/dq and
    // dID/dqd in rnea_deriv_kernel's packed layout, the H nobody asked for, and ydd take workspace
    const bool wide0 = sizeof(T) == 4 && p->solve_f64;
    const int n_rhs = (dq ? 1 : 0) + (dqd ? 1 : 0);
    // (d / d tau alone: the CRBA kernel writes the same interleaved H and the matrix-core solve inverts it)
    // (fp64 stays state-major: its interleaved workspace was built and measured in round 4 -- MIT Humanoid 8 % faster, JVRC-1 36 % slower, the
    // row-per-lane fp64 solve reads an interleaved block strided; profiles/r4_derivative_recursion_experiments.txt -- and removed again)
    // H^-1 = W^T W from the articulated-body quantities (minv_kernels.hip): no H, no dense factorisation; f32 and f64 alike on the
    // matrix cores, both workspaces interleaved by groups of kDerivGroup states.  GRBDA_NO_MINV=1 keeps the factorisation route (A/B runs)
    const MinvProgram &mv = p->host.deriv.minv;
    const bool minv = mv.ok && t->minv_bodies && t->minv_coltab && !wide0 && !p->no_minv &&
                      minv_solve_lds_bytes(static_cast<int>(nv), n_rhs, mv.n_entries, sizeof(T)) <= 160u * 1024u;
    const int il = minv ? kDerivGroup : ((!wide0 && spd_solve_on_mfma(sizeof(T), static_cast<int>(nv), n_rhs)) ? kDerivGroup : 1);
    // (only an INTERLEAVED H block can reach past the caller's array: the state-major layouts always build H in place)
    const bool h_in_place = !minv && dtau && (il == 1 || (B % kDerivGroup) == 0);
    const size_t per_state = minv ? static_cast<size_t>(mv.n_entries) + (need_d ? 2 * nn + nv : 0)
                                  : (h_in_place ? 0 : nn) + (need_d ? 2 * nn + nv : 0);
    size_t budget = work_budget(p, p->work, device, stream, (4096ull << 20) + (need_d ? B * nv * sizeof(T) : 0));
    if (need_d && budget > 2 * B * nv * sizeof(T)) budget -= B * nv * sizeof(T);  // (room for the whole batch's ydd, below)
    size_t chunk = budget / (per_state ? per_state * sizeof(T) : 1);
    chunk &= ~static_cast<size_t>(kWave - 1);  // whole tiles, whole groups of the interleaved workspace
    if (chunk < static_cast<size_t>(kWave)) chunk = kWave;
    {
        // whole ROUNDS of the one-state-per-lane kernels: a chunk of 2.4 rounds of wavefront slots takes as long as 3 (measured: 159 488-state
        // chunks of JVRC-1, 2 492 tiles on 1 024 slots of the recursion and 2 048 of the factor kernel / the ABA: 19 % and 40 % of the
        // slots idle in the last round).  n_cu * 8 wavefronts = one round at two per SIMD, two rounds of the recursion's four per CU.
        const size_t round = static_cast<size_t>(t->n_cu) * 8 * kWave;
        if (chunk >= round) chunk = chunk / round * round;
    }
    if (chunk > B) chunk = (B + kDerivGroup - 1) / kDerivGroup * kDerivGroup;  // (the last group of the workspace is allocated whole)
    // f32 with the matrix-core solve: the recursion writes H, dID/dq, dID/dqd interleaved by groups of kDerivGroup states
    // (deriv_kernels.hip); every other combination keeps the state-major layout (il, above)
    // several chunks: the forward dynamics of the WHOLE batch in one launch up front (B nv scalars more of workspace: 160 MB for a million
    // JVRC-1 states in fp32) instead of one launch per chunk -- a quarter-million-state launch runs at 0.35 ms, a quarter of the
    // million-state launch at 0.29
    const bool ydd_all = need_d && B > chunk;
    void *wptr = nullptr;
    if (int rc = ensure_work(p, p->work, device, stream, (chunk * per_state + (ydd_all ? B * nv : 0)) * sizeof(T) + 256, &wptr)) return rc;
    T *wnext = static_cast<T *>(wptr);
    auto take = [&](bool wanted) -> T * {
        if (!wanted) return nullptr;
        T *r = wnext;
        wnext += chunk * nn;
        return r;
    };
    T *wH = take(!minv && !h_in_place), *Dq = take(need_d), *Dqd = take(need_d);
    T *ydd_chunk = wnext;
    T *recs = minv ? ydd_chunk + (need_d ? chunk * nv : 0) : nullptr;
    T *ydd_whole = ydd_all ? static_cast<T *>(wptr) + chunk * per_state : nullptr;
    hipStream_t hs = static_cast<hipStream_t>(stream);
    DevPlan<T> d = make_dev_plan<T>(p, *t, false, false);
    if (ydd_all)
        if (int rc = run<T>(p, false, q, qd, tau, nullptr, ydd_whole, B, device, stream)) return rc;
    for (size_t b0 = 0; b0 < B; b0 += chunk) {
        T *ydd = ydd_all ? ydd_whole + b0 * nv : ydd_chunk;
        const size_t nb = B - b0 < chunk ? B - b0 : chunk;
        const size_t n_tiles = (nb + kWave - 1) / kWave;
        // (an interleaved H block spans the slots of a whole group: when the batch does not end on a group boundary the last
        // group would reach past the caller's d/dtau array, so that H goes to the workspace)
        T *H = h_in_place ? dtau + b0 * nn : wH;
        hipError_t e = hipSuccess;
        // (both kernels write H as packed rows of its lower triangle; the solve reads it through DerivProgram::related, so
        // nothing is cleared)
        if (need_d && !ydd_all)
            if (int rc = run<T>(p, false, q + b0 * nq, qd + b0 * nv, tau + b0 * nv, nullptr, ydd, nb, device, stream)) return rc;
        size_t grid = static_cast<size_t>(t->n_cu) * 8;
        if (grid > n_tiles) grid = n_tiles;
        const size_t rows = std::max(p->host.crba.n_rows, (need_d || minv) ? p->host.deriv.n_rows : 0);  // (the factor kernel of the minv route uses the recursion's rows)
        // (state-major results: three wavefronts per CU -- a fourth only adds open cache lines; interleaved: one per SIMD)
        const size_t deriv_waves = p->deriv_waves ? static_cast<size_t>(p->deriv_waves) : (il > 1 ? 4 : 3);
        const size_t slabs = std::max(grid, static_cast<size_t>(t->n_cu) * deriv_waves);  // (the factor kernel of the minv route runs n_cu * 8 wavefronts, as `grid`)
        void *scratch = nullptr;
        if (int rc = ensure_scratch(p, device, stream, slabs * rows * kWave * sizeof(T) + 256, &scratch)) return rc;
        if (need_d) {
            // (the derivative recursion carries the composite inertias in a common frame: H comes out of the same launch)
            size_t g2 = static_cast<size_t>(t->n_cu) * deriv_waves;
            if (g2 > n_tiles) g2 = n_tiles;
            e = launch_rnea_deriv<T>(d, t->deriv_bodies, p->host.n_clusters, p->host.deriv.n_rows, p->host.deriv.n_max, q + b0 * nq,
                                     qd + b0 * nv, ydd, Dq, Dqd, minv ? nullptr : H, nb, static_cast<T *>(scratch), static_cast<int>(g2), hs, il);
            if (e != hipSuccess) return hip_err(e, "rnea derivative launch");
        } else if (!minv) {
            e = launch_crba<T>(d, t->crba_bodies, p->host.n_clusters, p->host.crba.n_rows, q + b0 * nq, H, nb, static_cast<T *>(scratch),
                               static_cast<int>(grid), hs, true, il);
            if (e != hipSuccess) return hip_err(e, "crba launch");
        }
        T *o1 = dq ? dq + b0 * nn : nullptr, *o2 = dqd ? dqd + b0 * nn : nullptr, *o3 = dtau ? dtau + b0 * nn : nullptr;
        const T *r1 = dq ? Dq : nullptr, *r2 = dqd ? Dqd : nullptr;
        if (minv) {
            // articulated-inertia recursion -> record blocks (one state per lane, two wavefronts per SIMD), then the walk and the two
            // products on the matrix cores (one state per wavefront)
            e = launch_abi_factor<T>(d, t->deriv_bodies, t->minv_bodies, p->host.n_clusters, p->host.deriv.n_rows, p->host.deriv.n_max,
                                     mv.n_entries, q + b0 * nq, recs, nb, static_cast<T *>(scratch), static_cast<int>(grid), hs, kDerivGroup,
                                     t->bad_count);
            if (e != hipSuccess) return hip_err(e, "articulated-inertia factor launch");
            size_t per_cu = static_cast<size_t>(minv_workgroups_per_cu<T>(static_cast<int>(nv), p->host.deriv.n_max, n_rhs, mv.n_entries));
            if (p->minv_wpc > 0 && per_cu > static_cast<size_t>(p->minv_wpc)) per_cu = static_cast<size_t>(p->minv_wpc);
            size_t g3 = static_cast<size_t>(t->n_cu) * per_cu;
            const size_t units = (nb + kDerivGroup - 1) / kDerivGroup;
            if (g3 > units) g3 = units;
            e = launch_minv_solve<T>(recs, mv.n_entries, kDerivGroup, t->minv_coltab, mv.max_depth, mv.base_off, p->host.deriv.n_max, r1, r2,
                                     kDerivGroup, o3, o1, o2, t->deriv_related, static_cast<int>(nv), nb, static_cast<int>(g3), hs);
            if (e != hipSuccess) return hip_err(e, "minv solve launch");
            continue;
        }
        // one wavefront per state; as many as the LDS of a CU holds
        const bool wide = sizeof(T) == 4 && p->solve_f64;
        const size_t lds = spd_solve_lds_bytes(static_cast<int>(nv), wide ? 8 : sizeof(T), (dq ? 1 : 0) + (dqd ? 1 : 0));
        size_t per_cu = lds ? lds_workgroups_per_cu(lds) : 16;
        const bool mfma = !wide && spd_solve_on_mfma(sizeof(T), static_cast<int>(nv), n_rhs);
        if (per_cu > (mfma ? static_cast<size_t>(spd_mfma_workgroups_per_cu(static_cast<int>(nv))) : 16u)) per_cu = mfma ? spd_mfma_workgroups_per_cu(static_cast<int>(nv)) : 16;  // (matrix-core kernel: workgroups of four wavefronts)
        if (per_cu < 1) per_cu = 1;
        size_t g3 = static_cast<size_t>(t->n_cu) * per_cu;
        const size_t units = mfma ? (nb + kDerivGroup - 1) / kDerivGroup : nb;
        if (g3 > units) g3 = units;
        const uint64_t *rel = t->deriv_related;
        const int nvi = static_cast<int>(nv), g3i = static_cast<int>(g3), hp = 1;
        const int sil = il;
        if constexpr (sizeof(T) == 4) {
            if (wide) e = launch_spd_solve<float, double>(H, hp, r1, r2, o3, o1, o2, rel, nvi, nb, g3i, hs, 1);
            else e = launch_spd_solve<float, float>(H, hp, r1, r2, o3, o1, o2, rel, nvi, nb, g3i, hs, sil);
        } else {
            e = launch_spd_solve<double, double>(H, hp, r1, r2, o3, o1, o2, rel, nvi, nb, g3i, hs, 1);
        }
        if (e != hipSuccess) return hip_err(e, "spd solve launch");
    }
    return GRBDA_OK;
}

template <class T>
int manifold_mass(const grbda_plan *p, const T *q, T *H, size_t B, int device, void *stream)
{
    // models with implicit clusters: H = G^T H_s G through the spanning tree (manifold_kernels.hip) instead of nv + 1 inverse dynamics
    if (!p || !q || !H || p->no_crba) return 1;
    GRBDA_CALL_SCOPE(p);
    if (!manifold_covers<T>(p)) return 1;
    return manifold_derivs<T>(p, q, nullptr, nullptr, nullptr, nullptr, nullptr, H, B, device, stream);
}

template <class T>
static std::string kernel_name_of(const grbda_plan *p, int kind, int n_cu, size_t B)
{
    const HostPlan &h = p->host;
    const char *tn = sizeof(T) == 4 ? "float" : "double";
    char buf[160];
    if (kind == 0) {
        const ChainProgram &cp = sizeof(T) == 8 ? h.chain64 : h.chain32;
        switch (choose_aba<T>(p, n_cu, B, false)) {
            case ABA_GEN1:
                std::snprintf(buf, sizeof buf, "grbda_hip::aba_gen1_kernel<%s, %d, %s, %d>", tn, cp.gens[0].n, cp.gens[0].kind ? "true" : "false",
                              gen1_waves_per_simd<T>(cp.gens[0].n));
                return buf;
            case ABA_LM: {
                const ChainProgram &lp = sizeof(T) == 8 ? h.chain64p : h.chain32p;
                std::snprintf(buf, sizeof buf, "grbda_hip::aba_chain_lm_kernel<%s, 2%s>", tn, lp.diffs.empty() ? "" : ", true");
                return buf;
            }
            case ABA_LM4:
                std::snprintf(buf, sizeof buf, "grbda_hip::aba_chain_lm_kernel<%s, 4%s>", tn, (sizeof(T) == 8 ? h.chain64q : h.chain32q).diffs.empty() ? "" : ", true");
                return buf;
            case ABA_CHAIN_WIDE: std::snprintf(buf, sizeof buf, "grbda_hip::aba_chain_kernel<%s, %d, 0>", tn, kChainWideWps); return buf;
            case ABA_CHAIN:
                std::snprintf(buf, sizeof buf, "grbda_hip::aba_chain_kernel<%s, %d, %d>", tn, (sizeof(T) == 8 && !cp.gens.empty()) ? 1 : 2,
                              !cp.gens.empty() ? 2 : (!cp.diffs.empty() ? 1 : 0));
                return buf;
            default: break;
        }
        bool loop = false;
        for (const ClusterRec &cr : h.lay64.clusters) loop = loop || cr.kind == CK_LOOP;
        std::snprintf(buf, sizeof buf, "grbda_hip::aba_kernel<%s, %s>", tn, loop ? "true" : "false");
        return buf;
    }
    const bool chain = !p->no_chain && (sizeof(T) == 8 ? h.rchain64.ok : h.rchain32.ok);
    if (chain) {
        const RneaChainProgram &rp = sizeof(T) == 8 ? h.rchain64 : h.rchain32;
        if (rnea_gen1_usable<T>(p)) {
            std::snprintf(buf, sizeof buf, "grbda_hip::rnea_gen1_kernel<%s, %d, %s, %d>", tn, rp.gens[0].n, rp.gens[0].kind ? "true" : "false",
                          rnea_gen1_waves_per_simd<T>(rp.gens[0].n));
            return buf;
        }
        if (const int lmw = choose_rnea_lm<T>(p, n_cu, B)) {
            const RneaChainProgram &lr = lmw == 4 ? (sizeof(T) == 8 ? h.rchain64q : h.rchain32q) : (sizeof(T) == 8 ? h.rchain64p : h.rchain32p);
            std::snprintf(buf, sizeof buf, "grbda_hip::rnea_chain_lm_kernel<%s, %d%s>", tn, lmw, lr.diffs.empty() ? "" : ", true");
            return buf;
        }
        std::snprintf(buf, sizeof buf, "grbda_hip::rnea_chain_kernel<%s, %d, %s>", tn, !rp.gens.empty() ? 2 : (rp.diffs.empty() ? 0 : 1), rp.n_glb > 0 ? "true" : "false");
        return buf;
    }
    bool loop = false;
    for (const ClusterRec &cr : h.lay64.clusters) loop = loop || cr.kind == CK_LOOP;
    std::snprintf(buf, sizeof buf, "grbda_hip::rnea_kernel<%s, %s>", tn, loop ? "true" : "false");
    return buf;
}

// ---- one process, several devices: contiguous batch shards, plan replicated ------------------------
template <class T>
int run_sharded(const grbda_plan *p, bool rnea, const T *q, const T *qd, const T *x, T *out, size_t B, int n_gpus)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    GRBDA_CALL_SCOPE(p);
    if (!p || !q || !qd || !x || !out) return set_err(GRBDA_EINVAL, "null argument");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return set_err(GRBDA_ENODEVICE, "no HIP device available (there is no CPU fallback)");
    if (n_gpus < 1 || n_gpus > count) return set_err(GRBDA_EINVAL, "n_gpus out of range");
    if (B == 0) return GRBDA_OK;
    const size_t nq = p->host.nq, nv = p->host.nv;
    struct Shard {
        size_t b0 = 0, nb = 0;
        T *dq = nullptr, *dqd = nullptr, *dx = nullptr, *dout = nullptr;
        hipStream_t s = nullptr;
    };
    std::vector<Shard> sh(n_gpus);
    int rc = GRBDA_OK;
    hipError_t e = hipSuccess;
    for (int g = 0; g < n_gpus && rc == GRBDA_OK; g++) {
        Shard &S = sh[g];
        S.b0 = B * g / n_gpus;
        S.nb = B * (g + 1) / n_gpus - S.b0;
        if (S.nb == 0) continue;
        if ((e = hipSetDevice(g)) != hipSuccess || (e = hipStreamCreate(&S.s)) != hipSuccess ||
            (e = hipMalloc((void **)&S.dq, S.nb * nq * sizeof(T))) != hipSuccess ||
            (e = hipMalloc((void **)&S.dqd, S.nb * nv * sizeof(T))) != hipSuccess ||
            (e = hipMalloc((void **)&S.dx, S.nb * nv * sizeof(T))) != hipSuccess ||
            (e = hipMalloc((void **)&S.dout, S.nb * nv * sizeof(T))) != hipSuccess ||
            (e = hipMemcpyAsync(S.dq, q + S.b0 * nq, S.nb * nq * sizeof(T), hipMemcpyHostToDevice, S.s)) != hipSuccess ||
            (e = hipMemcpyAsync(S.dqd, qd + S.b0 * nv, S.nb * nv * sizeof(T), hipMemcpyHostToDevice, S.s)) != hipSuccess ||
            (e = hipMemcpyAsync(S.dx, x + S.b0 * nv, S.nb * nv * sizeof(T), hipMemcpyHostToDevice, S.s)) != hipSuccess) {
            rc = hip_err(e, "shard setup");
            break;
        }
        rc = run<T>(p, rnea, S.dq, S.dqd, S.dx, nullptr, S.dout, S.nb, g, S.s);
        if (rc == GRBDA_OK &&
            (e = hipMemcpyAsync(out + S.b0 * nv, S.dout, S.nb * nv * sizeof(T), hipMemcpyDeviceToHost, S.s)) != hipSuccess)
            rc = hip_err(e, "shard copy back");
    }
    for (int g = 0; g < n_gpus; g++) {
        Shard &S = sh[g];
        if (!S.s) continue;
        (void)hipSetDevice(g);
        if ((e = hipStreamSynchronize(S.s)) != hipSuccess && rc == GRBDA_OK) rc = hip_err(e, "shard execution");
        if (S.dq) (void)hipFree(S.dq);
        if (S.dqd) (void)hipFree(S.dqd);
        if (S.dx) (void)hipFree(S.dx);
        if (S.dout) (void)hipFree(S.dout);
        {   // the scratch slab this call made for its private stream goes with the stream
            std::lock_guard<std::recursive_mutex> lk(p->mu);
            auto it = p->scratch.find({g, S.s});
            if (it != p->scratch.end()) {
                if (it->second.ptr) (void)hipFree(it->second.ptr);
                p->scratch.erase(it);
            }
        }
        (void)hipStreamDestroy(S.s);
    }
    return rc;
}

// ---- one process, several devices, DEVICE-resident shards: launch every shard on its own device / stream, optionally gather the
// result slabs on the first device over the peer links (hipMemcpyPeerAsync: xGMI, no host hop).  Enqueues only.
template <class T>
int run_sharded_dev(const grbda_plan *p, bool rnea, int n_gpus, const int *devices, const T *const *q, const T *const *qd, const T *const *x,
                    T *const *out, const size_t *B, void *const *streams, T *gathered)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    if (n_gpus < 1 || !q || !qd || !x || !B || (!out && !gathered)) return set_err(GRBDA_EINVAL, "null argument or n_gpus < 1");
    GRBDA_CALL_SCOPE(p);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return set_err(GRBDA_ENODEVICE, "no HIP device available (there is no CPU fallback)");
    const size_t nv = p->host.nv;
    for (int g = 0; g < n_gpus; g++) {
        const int dev = devices ? devices[g] : g;
        if (dev < 0 || dev >= count) return set_err(GRBDA_EINVAL, "device index out of range");
        for (int h = 0; h < g; h++)
            if ((devices ? devices[h] : h) == dev && (streams ? streams[h] : nullptr) == (streams ? streams[g] : nullptr) && B[g] && B[h])
                return set_err(GRBDA_EINVAL, "two shards on one (device, stream): they would share one scratch slab -- give them distinct streams");
        if (B[g] && (!q[g] || !qd[g] || !x[g] || (!gathered && (!out || !out[g])))) return set_err(GRBDA_EINVAL, "null shard pointer");
    }
    const int dev0 = devices ? devices[0] : 0;
    void *const s0 = streams ? streams[0] : nullptr;
    size_t off = 0;
    hipError_t e = hipSuccess;
    for (int g = 0; g < n_gpus; g++) {
        const int dev = devices ? devices[g] : g;
        void *const sg = streams ? streams[g] : nullptr;
        const size_t nb = B[g];
        if (nb == 0) continue;
        // a shard on the gather device computes straight into its place of the gathered array
        T *dst = gathered ? gathered + off * nv : nullptr;
        T *o = (out && out[g]) ? out[g] : nullptr;
        if (gathered && dev == dev0 && !o) o = dst;
        if (!o) return set_err(GRBDA_EINVAL, "a shard on another device than the gather device needs its own output slab (out[g])");
        if (const int rc = run<T>(p, rnea, q[g], qd[g], x[g], nullptr, o, nb, dev, sg)) return rc;
        if (gathered && o != dst) {
            if ((e = hipSetDevice(dev)) != hipSuccess) return hip_err(e, "hipSetDevice");
            if (dev != dev0) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, dev, dev0) == hipSuccess && can) {
                    e = hipDeviceEnablePeerAccess(dev0, 0);  // (direct over the link; without it the runtime stages the copy)
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                    (void)hipGetLastError();
                }
                e = hipMemcpyPeerAsync(dst, dev0, o, dev, nb * nv * sizeof(T), static_cast<hipStream_t>(sg));
            } else {
                e = hipMemcpyAsync(dst, o, nb * nv * sizeof(T), hipMemcpyDeviceToDevice, static_cast<hipStream_t>(sg));
            }
            if (e != hipSuccess) return hip_err(e, "gather copy");
        }
        // the first shard's stream is the join point: work enqueued on it after this call sees every slab of `gathered`
        if (gathered && (dev != dev0 || sg != s0)) {
            hipEvent_t ev = nullptr;
            if ((e = hipSetDevice(dev)) != hipSuccess || (e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess ||
                (e = hipEventRecord(ev, static_cast<hipStream_t>(sg))) != hipSuccess)
                return hip_err(e, "gather event");
            if ((e = hipSetDevice(dev0)) != hipSuccess || (e = hipStreamWaitEvent(static_cast<hipStream_t>(s0), ev, 0)) != hipSuccess) {
                (void)hipEventDestroy(ev);
                return hip_err(e, "gather join");
            }
            (void)hipEventDestroy(ev);  // (released by the runtime once the recorded work has completed)
        }
        off += nb;
    }
    return GRBDA_OK;
}

// ---- host-pointer convenience for the contact-side entry points (single-state facade calls) ----------------------
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { hipError_t e = hipMalloc(&p, bytes ? bytes : 16); return e == hipSuccess ? 0 : hip_err(e, "hipMalloc"); }
    int put(const void *src, size_t bytes) { hipError_t e = hipMemcpy(p, src, bytes, hipMemcpyHostToDevice); return e == hipSuccess ? 0 : hip_err(e, "hipMemcpy H2D"); }
    int get(void *dst, size_t bytes) const { hipError_t e = hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost); return e == hipSuccess ? 0 : hip_err(e, "hipMemcpy D2H"); }
};

}  // namespace

extern "C" {

const char *grbda_strerror(int code)
{
    switch (code) {
        case GRBDA_OK: return "ok";
        case GRBDA_EINVAL: return "invalid argument or malformed model description";
        case GRBDA_EUNSUPPORTED: return "model feature not supported by the HIP kernels";
        case GRBDA_ENODEVICE: return "no usable HIP device";
        case GRBDA_EHIP: return "HIP runtime error";
        case GRBDA_ENOMEM: return "out of memory";
        case GRBDA_EPARSE: return "URDF parse error";
        case GRBDA_ESTATE: return "invalid spanning state";
        default: return "unknown error";
    }
}
const char *grbda_last_error(void) { return g_last_error.c_str(); }

// nv x nv, 1 where two velocity coordinates lie on one root path (same cluster, or one cluster an ancestor of the other): what
// DerivProgram::related holds as one 64-bit word per coordinate, for plans beyond 64 coordinates (manifold_kernels.hip, kernels 2w and 4)
static void build_related_table(HostPlan &h)
{
    const std::vector<ClusterRec> &cl = h.lay64.clusters;
    const int nc = static_cast<int>(cl.size()), nv = h.nv;
    std::vector<int> body_cluster(h.n_bodies, 0), parent(nc, -1);
    for (int c = 0; c < nc; c++)
        for (int i = 0; i < cl[c].k; i++) body_cluster[cl[c].first_body + i] = c;
    for (int c = 0; c < nc; c++) parent[c] = cl[c].parent_body >= 0 ? body_cluster[cl[c].parent_body] : -1;
    auto dof = [&](int c) { return cl[c].kind == CK_FREE ? 6 : cl[c].n; };
    h.related_table.assign(static_cast<size_t>(nv) * nv, 0);
    for (int c = 0; c < nc; c++)
        for (int a = c; a >= 0; a = parent[a])
            for (int i = 0; i < dof(c); i++)
                for (int j = 0; j < dof(a); j++) {
                    h.related_table[static_cast<size_t>(cl[c].v_index + i) * nv + cl[a].v_index + j] = 1;
                    h.related_table[static_cast<size_t>(cl[a].v_index + j) * nv + cl[c].v_index + i] = 1;
                }
}

int grbda_plan_from_blob(const void *blob, size_t bytes, grbda_plan **out)
{
    if (!out) return set_err(GRBDA_EINVAL, "null out pointer");
    *out = nullptr;
    std::unique_ptr<grbda_plan> p(new (std::nothrow) grbda_plan());
    if (!p) return set_err(GRBDA_ENOMEM, "allocation failed");
    char msg[256] = {0};
    // tuning knobs: GRBDA_LDS_BYTES_PER_WAVE / GRBDA_WAVES_PER_CU set all four kernels, the suffixed
    // forms (_ABA32, _ABA64, _RNEA32, _RNEA64) one of them
    static const char *const suffix[4] = {"_ABA32", "_ABA64", "_RNEA32", "_RNEA64"};
    for (int k = 0; k < 4; k++) {
        int v = env_int("GRBDA_LDS_BYTES_PER_WAVE", p->lds_bytes_per_wave[k]);
        v = env_int((std::string("GRBDA_LDS_BYTES_PER_WAVE") + suffix[k]).c_str(), v);
        p->lds_bytes_per_wave[k] = v < 0 ? 0 : (v > 160 * 1024 ? 160 * 1024 : v);
        int w = env_int("GRBDA_WAVES_PER_CU", p->waves_per_cu[k]);
        w = env_int((std::string("GRBDA_WAVES_PER_CU") + suffix[k]).c_str(), w);
        p->waves_per_cu[k] = w < 1 ? 1 : (w > 32 ? 32 : w);
    }
    {
        int w = env_int("GRBDA_WAVES_PER_CU", p->waves_per_cu_f64_wide_regs);
        w = env_int("GRBDA_WAVES_PER_CU_ABA64", w);
        p->waves_per_cu_f64_wide_regs = w < 1 ? 1 : (w > 32 ? 32 : w);
    }
    p->no_split = env_int("GRBDA_NO_SPLIT", 0) != 0;
    p->no_minv = env_int("GRBDA_NO_MINV", 0) != 0;
    p->minv_wpc = env_int("GRBDA_MINV_WPC", 0);
    p->no_chain = env_int("GRBDA_NO_CHAIN", 0) != 0;
    p->no_latency_mode = env_int("GRBDA_NO_LATENCY_MODE", 0) != 0;
    p->lm_waves = env_int("GRBDA_LM_WAVES", 0);
    p->gen1_waves_cap = env_int("GRBDA_GEN1_WAVES_PER_CU", 0);
    p->gen1_tiles_per_wave = env_int("GRBDA_GEN1_TILES_PER_WAVE", 0);
    p->chain_wide = env_int("GRBDA_CHAIN_WIDE", 0) != 0;
#ifdef GRBDA_EXP
    // ablation switches of tools/chain_ablate.py: results are WRONG when set, so the product library does not read them -- only
    // the experiment builds do (make variant VFLAGS=-DGRBDA_EXP)
    p->chain_debug = env_int("GRBDA_CHAIN_DEBUG", 0);
#else
    p->chain_debug = 0;
#endif
    p->no_crba = env_int("GRBDA_NO_CRBA", 0) != 0;
    p->rnea_narrow = env_int("GRBDA_RNEA_NARROW", 0) != 0;
    p->no_analytic = env_int("GRBDA_NO_ANALYTIC", 0) != 0;
    p->solve_f64 = env_int("GRBDA_SOLVE_F64", 0) != 0;
    p->crba_waves = env_int("GRBDA_CRBA_WAVES_PER_CU", 16);
    if (p->crba_waves < 1 || p->crba_waves > 16) p->crba_waves = 16;
    p->deriv_waves = env_int("GRBDA_DERIV_WAVES_PER_CU", 0);
    if (p->deriv_waves < 0) p->deriv_waves = 0;
    p->no_efpa = env_int("GRBDA_NO_EFPA", 0) != 0;
    LdsBudget lds;
    lds.aba32 = p->lds_bytes_per_wave[0] / (4 * kWave);
    lds.aba64 = p->lds_bytes_per_wave[1] / (8 * kWave);
    lds.rnea32 = p->lds_bytes_per_wave[2] / (4 * kWave);
    lds.rnea64 = p->lds_bytes_per_wave[3] / (8 * kWave);
    lds.chain32w = kChainWideLdsBytes / (4 * kWave);
    // profiling aid (results are wrong when set): GRBDA_DEBUG_SWEEPS is a bit mask of the ABA sweeps to
    // keep -- 1 forward, 2 backward, 4 acceleration -- so that the cost of each sweep can be ablated
#ifdef GRBDA_EXP
    const int sweeps = env_int("GRBDA_DEBUG_SWEEPS", 7);
#else
    const int sweeps = 7;
#endif
    int rc = compile_plan(blob, bytes, lds, sweeps, p->host, msg, sizeof msg);
    if (rc) return set_err(rc, msg);
    p->blob.assign(static_cast<const unsigned char *>(blob), static_cast<const unsigned char *>(blob) + bytes);
    p->no_manifold = env_int("GRBDA_NO_MANIFOLD", 0) != 0;
    bool implicit = false;
    for (const ClusterRec &cr : p->host.lay64.clusters) implicit = implicit || cr.kind == CK_LOOP;
    // (plans with big clusters: up to 128 velocities -- tables instead of the one-word masks, build_related_table)
    if ((implicit || p->host.projection_only) && p->host.nv <= (p->host.big_clusters ? 2 * kWave : kWave)) {
        // the spanning-tree model for the derivatives on the constraint manifold (at most 64 spanning velocities: the masks of
        // DerivProgram::related)
        std::vector<unsigned char> sb;
        if (make_spanning_blob(blob, bytes, sb, p->span_q, p->span_v, msg, sizeof msg) == 0) {
            grbda_plan *sp = nullptr;
            if (grbda_plan_from_blob(sb.data(), sb.size(), &sp) == GRBDA_OK) {
                if (sp->host.nv <= (p->host.big_clusters ? 2 * kWave : kWave) && sp->host.deriv.ok) {
                    p->span = sp;
                    if (p->host.nv > kWave || sp->host.nv > kWave) {
                        build_related_table(p->host);
                        build_related_table(sp->host);
                    }
                    std::memcpy(sp->host.gravity, p->host.gravity, sizeof sp->host.gravity);
                    p->crow.assign(p->host.n_clusters, 0);
                    int rows = 0;
                    for (int c = 0; c < p->host.n_clusters; c++) {
                        const ClusterRec &cr = p->host.lay64.clusters[c];
                        p->crow[c] = rows;
                        if (cr.kind == CK_LOOP) rows += cr.k * (p->host.big_clusters ? cr.n : cr.n * (4 + cr.n));  // (manifold_kernels.hip, cpl_stride)
                        else if (cr.kind == CK_STATIC && p->host.big_clusters) rows += cr.k * cr.n;  // (wide plans keep every cluster's G rows in the slab)
                    }
                    p->n_cpl_rows = rows;
                    // (every implicit cluster within 4 bodies / 2 independent coordinates: the constraint kernel's half-size build)
                    bool small = !p->host.big_clusters;
                    for (int c = 0; c < p->host.n_clusters; c++) {
                        const ClusterRec &cr = p->host.lay64.clusters[c];
                        if (cr.kind == CK_LOOP && (cr.k > 4 || cr.n > 2)) small = false;
                        if (cr.kind == CK_LOOP && cr.cons_type != 0) p->has_trig = true;
                    }
                    p->constraint_shape = p->host.big_clusters ? 1 : (small && !env_int("GRBDA_NO_SMALL_CONSTRAINT", 0) ? 2 : 0);
                } else {
                    grbda_plan_free(sp);
                }
            }
        }
    }
    *out = p.release();
    return GRBDA_OK;
}

int grbda_urdf_to_blob(const char *const *paths, int n_paths, int ori_repr, void *buf, size_t cap, size_t *needed)
{
    if (!paths || n_paths <= 0) return set_err(GRBDA_EINVAL, "no URDF path");
    std::vector<unsigned char> blob;
    std::string err;
    int rc = urdf_to_blob(paths, n_paths, ori_repr, blob, err);
    if (rc) return set_err(rc, err);
    if (needed) *needed = blob.size();
    if (buf) {
        if (cap < blob.size()) return set_err(GRBDA_EINVAL, "buffer too small");
        std::memcpy(buf, blob.data(), blob.size());
    }
    return GRBDA_OK;
}

int grbda_plan_from_urdf(const char *path, int ori_repr, grbda_plan **out)
{
    if (!path || !out) return set_err(GRBDA_EINVAL, "null argument");
    std::vector<unsigned char> blob;
    std::string err;
    const char *paths[1] = {path};
    int rc = urdf_to_blob(paths, 1, ori_repr, blob, err);
    if (rc) return set_err(rc, err);
    return grbda_plan_from_blob(blob.data(), blob.size(), out);
}

void grbda_plan_free(grbda_plan *p)
{
    if (!p) return;
    DeviceGuard device_guard_;
    for (auto &kv : p->dev) {
        if (hipSetDevice(kv.first) != hipSuccess) continue;
        DeviceTables &t = kv.second;
        (void)hipFree(t.aba_steps); (void)hipFree(t.rnea_steps); (void)hipFree(t.consts64); (void)hipFree(t.consts32);
        (void)hipFree(t.cints); (void)hipFree(t.dq_map); (void)hipFree(t.crba_bodies); (void)hipFree(t.deriv_bodies); (void)hipFree(t.deriv_related); (void)hipFree(t.related_table); (void)hipFree(t.minv_bodies); (void)hipFree(t.minv_coltab);
        (void)hipFree(t.span_q); (void)hipFree(t.span_v); (void)hipFree(t.crow);
        for (int w = 0; w < 7; w++) { (void)hipFree(t.rchain_segs[w]); (void)hipFree(t.rchain_links[w]); (void)hipFree(t.rchain_pairs[w]); (void)hipFree(t.rchain_frees[w]); (void)hipFree(t.rchain_diffs[w]); (void)hipFree(t.rchain_gens[w]); (void)hipFree(t.rchain_gbodies[w]); }
        for (int w = 0; w < 7; w++) { (void)hipFree(t.chain_segs[w]); (void)hipFree(t.chain_links[w]); (void)hipFree(t.chain_pairs[w]); (void)hipFree(t.chain_frees[w]); (void)hipFree(t.chain_diffs[w]); (void)hipFree(t.chain_gens[w]); (void)hipFree(t.chain_gbodies[w]); }
        for (int w = 0; w < kLayouts; w++) { (void)hipFree(t.acc_k[w]); (void)hipFree(t.clusters[w]); (void)hipFree(t.rnea_clusters[w]); (void)hipFree(t.bodies[w]); (void)hipFree(t.rnea_bodies[w]); }
    }
    for (auto *m : {&p->scratch, &p->work, &p->work_cvt, &p->work_proj})
        for (auto &kv : *m) {
            if (hipSetDevice(kv.first.first) != hipSuccess) continue;
            if (kv.second.ptr) (void)hipFree(kv.second.ptr);
        }
    delete p;
}

int grbda_plan_release_work(grbda_plan *p, unsigned long long *bytes_released)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    GRBDA_CALL_SCOPE(p);
    unsigned long long total = 0;
    for (auto *m : {&p->work, &p->work_cvt, &p->work_proj})
        for (auto &kv : *m) {
            if (!kv.second.ptr) continue;
            hipError_t e = hipSetDevice(kv.first.first);
            if (e != hipSuccess) return hip_err(e, "hipSetDevice");
            // a stream that is capturing must not lose a buffer its graph refers to, and hipFree would wait for the capture
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (kv.first.second && hipStreamIsCapturing(static_cast<hipStream_t>(kv.first.second), &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
                return set_err(GRBDA_EINVAL, "a stream of this plan is capturing: its work buffer cannot be released now");
            if ((e = hipFree(kv.second.ptr)) != hipSuccess) return hip_err(e, "hipFree");  // (waits for the work enqueued on it)
            total += kv.second.bytes;
            kv.second.ptr = nullptr;
            kv.second.bytes = 0;
        }
    if (p->span) {
        unsigned long long more = 0;
        if (const int rc = grbda_plan_release_work(p->span, &more)) return rc;
        total += more;
    }
    if (bytes_released) *bytes_released = total;
    return GRBDA_OK;
}

int grbda_plan_dims(const grbda_plan *p, int *nq, int *nv, int *n_bodies, int *n_clusters)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    if (nq) *nq = p->host.nq;
    if (nv) *nv = p->host.nv;
    if (n_bodies) *n_bodies = p->host.n_bodies;
    if (n_clusters) *n_clusters = p->host.n_clusters;
    return GRBDA_OK;
}

int grbda_plan_set_gravity(grbda_plan *p, const double g[3])
{
    if (!p || !g) return set_err(GRBDA_EINVAL, "null argument");
    for (int i = 0; i < 3; i++) p->host.gravity[3 + i] = g[i];
    if (p->span) grbda_plan_set_gravity(p->span, g);
    // keep the stored description in sync so that grbda_plan_blob() round-trips
    std::memcpy(reinterpret_cast<grbda_desc_header *>(p->blob.data())->gravity, p->host.gravity, sizeof(double) * 6);
    return GRBDA_OK;
}
int grbda_plan_get_gravity(const grbda_plan *p, double g[3])
{
    if (!p || !g) return set_err(GRBDA_EINVAL, "null argument");
    for (int i = 0; i < 3; i++) g[i] = p->host.gravity[3 + i];
    return GRBDA_OK;
}

int grbda_plan_blob(const grbda_plan *p, const void **blob, size_t *bytes)
{
    if (!p || !blob || !bytes) return set_err(GRBDA_EINVAL, "null argument");
    *blob = p->blob.data();
    *bytes = p->blob.size();
    return GRBDA_OK;
}

int grbda_plan_info(const grbda_plan *p, grbda_plan_info_t *info)
{
    if (!p || !info) return set_err(GRBDA_EINVAL, "null argument");
    std::memset(info, 0, sizeof *info);
    info->n_slots = p->host.lay32.n_lds_aba + p->host.lay32.n_glb_aba;
    info->n_lds_slots_f32 = p->host.lay32.n_lds_aba;
    info->n_lds_slots_f64 = p->host.lay64.n_lds_aba;
    info->lds_bytes_f32 = static_cast<size_t>(info->n_lds_slots_f32) * kWave * 4;
    info->lds_bytes_f64 = static_cast<size_t>(info->n_lds_slots_f64) * kWave * 8;
    info->scratch_bytes_per_wave_f32 = static_cast<size_t>(p->host.lay32.n_glb_aba) * kWave * 4;
    info->scratch_bytes_per_wave_f64 = static_cast<size_t>(p->host.lay64.n_glb_aba) * kWave * 8;
    info->flops_aba = p->host.flops_aba;
    info->flops_rnea = p->host.flops_rnea;
    info->bytes_aba_f32 = (p->host.nq + 3.0 * p->host.nv) * 4;
    info->bytes_aba_f64 = (p->host.nq + 3.0 * p->host.nv) * 8;
    for (const BodyRec &b : p->host.lay32.bodies) info->n_axisym_bodies += b.axisym;
    for (const ClusterRec &c : p->host.lay32.clusters) info->n_carry_clusters += c.carry_out;
    info->split_aba_f32 = p->host.lay32s.split_aba && !p->no_split;
    info->split_rnea_f32 = p->host.lay32s.split_rnea && !p->no_split;
    info->n_lds_slots_split_f32 = p->host.lay32s.n_lds_aba;
    info->chain_aba_f32 = p->host.chain32.ok && !p->no_chain;
    info->n_lds_slots_chain_f32 = p->host.chain32.n_lds;
    info->n_chain_segments = static_cast<int>(p->host.chain32.segs.size());
    info->chain_aba_f64 = p->host.chain64.ok && !p->no_chain;
    info->chain_rnea_f32 = p->host.rchain32.ok && !p->no_chain;
    info->chain_rnea_f64 = p->host.rchain64.ok && !p->no_chain;
    info->analytic_derivatives = (analytic_covers<double>(p) || manifold_covers<double>(p)) ? 1 : 0;
    info->n_chain_differentials = p->no_chain ? 0 : static_cast<int>(p->host.chain32.diffs.size());
    info->latency_mode_f32 = p->host.chain32p.ok && !p->no_chain && !p->no_latency_mode;
    info->latency_mode_f64 = p->host.chain64p.ok && !p->no_chain && !p->no_latency_mode;
    info->n_chain_generic = p->no_chain ? 0 : static_cast<int>(p->host.chain32.gens.size());
    info->spanning_tree_route = p->host.projection_only ? 1 : 0;
    return GRBDA_OK;
}

int grbda_aba_f64(const grbda_plan *p, const double *q, const double *qd, const double *tau, const double *f_ext,
                  double *ydd, size_t B, int device, void *stream)
{
    return run<double>(p, false, q, qd, tau, f_ext, ydd, B, device, stream);
}
int grbda_aba_f32(const grbda_plan *p, const float *q, const float *qd, const float *tau, const float *f_ext,
                  float *ydd, size_t B, int device, void *stream)
{
    return run<float>(p, false, q, qd, tau, f_ext, ydd, B, device, stream);
}
int grbda_rnea_f64(const grbda_plan *p, const double *q, const double *qd, const double *ydd, const double *f_ext,
                   double *tau, size_t B, int device, void *stream)
{
    return run<double>(p, true, q, qd, ydd, f_ext, tau, B, device, stream);
}
int grbda_rnea_f32(const grbda_plan *p, const float *q, const float *qd, const float *ydd, const float *f_ext,
                   float *tau, size_t B, int device, void *stream)
{
    return run<float>(p, true, q, qd, ydd, f_ext, tau, B, device, stream);
}

int grbda_bias_f64(const grbda_plan *p, const double *q, const double *qd, const double *f_ext, double *out, size_t B,
                   int device, void *stream)
{
    return derived<double>(p, DM_BIAS, q, qd, nullptr, f_ext, out, B, device, stream);
}
int grbda_bias_f32(const grbda_plan *p, const float *q, const float *qd, const float *f_ext, float *out, size_t B,
                   int device, void *stream)
{
    return derived<float>(p, DM_BIAS, q, qd, nullptr, f_ext, out, B, device, stream);
}
int grbda_mass_matrix_f64(const grbda_plan *p, const double *q, double *H, size_t B, int device, void *stream)
{
    if (const int rc = manifold_mass<double>(p, q, H, B, device, stream); rc != 1) return rc;
    return derived<double>(p, DM_MASS, q, nullptr, nullptr, nullptr, H, B, device, stream);
}
int grbda_mass_matrix_f32(const grbda_plan *p, const float *q, float *H, size_t B, int device, void *stream)
{
    if (const int rc = manifold_mass<float>(p, q, H, B, device, stream); rc != 1) return rc;
    return derived<float>(p, DM_MASS, q, nullptr, nullptr, nullptr, H, B, device, stream);
}
int grbda_fd_dtau_f64(const grbda_plan *p, const double *q, double *Hinv, size_t B, int device, void *stream)
{
    if (p && q && Hinv)
        if (const int rc = analytic_derivs<double>(p, q, nullptr, nullptr, nullptr, nullptr, Hinv, B, device, stream); rc != 1) return rc;
    return derived<double>(p, DM_DTAU, q, nullptr, nullptr, nullptr, Hinv, B, device, stream);
}
int grbda_fd_dtau_f32(const grbda_plan *p, const float *q, float *Hinv, size_t B, int device, void *stream)
{
    if (p && q && Hinv)
        if (const int rc = analytic_derivs<float>(p, q, nullptr, nullptr, nullptr, nullptr, Hinv, B, device, stream); rc != 1) return rc;
    return derived<float>(p, DM_DTAU, q, nullptr, nullptr, nullptr, Hinv, B, device, stream);
}
int grbda_fd_dqd_f64(const grbda_plan *p, const double *q, const double *qd, const double *tau, double *J, size_t B,
                     int device, void *stream)
{
    if (p && q && qd && tau && J)
        if (const int rc = analytic_derivs<double>(p, q, qd, tau, nullptr, J, nullptr, B, device, stream); rc != 1) return rc;
    return derived<double>(p, DM_DQD, q, qd, tau, nullptr, J, B, device, stream);
}
int grbda_fd_dqd_f32(const grbda_plan *p, const float *q, const float *qd, const float *tau, float *J, size_t B,
                     int device, void *stream)
{
    if (p && q && qd && tau && J)
        if (const int rc = analytic_derivs<float>(p, q, qd, tau, nullptr, J, nullptr, B, device, stream); rc != 1) return rc;
    return derived<float>(p, DM_DQD, q, qd, tau, nullptr, J, B, device, stream);
}

int grbda_fd_dq_f64(const grbda_plan *p, const double *q, const double *qd, const double *tau, double step, double *J,
                    size_t B, int device, void *stream)
{
    if (p && q && qd && tau && J && step > 0)
        if (const int rc = analytic_derivs<double>(p, q, qd, tau, J, nullptr, nullptr, B, device, stream); rc != 1) return rc;
    return derived<double>(p, DM_DQ, q, qd, tau, nullptr, J, B, device, stream, step);
}
int grbda_fd_dq_f32(const grbda_plan *p, const float *q, const float *qd, const float *tau, double step, float *J,
                    size_t B, int device, void *stream)
{
    // A central difference in fp32 has no usable step (eps / h + h^2 bottoms out near 1e-2 relative): the differences
    // are taken in fp64 on the converted inputs and the matrices converted back.
    if (!p || !q || !qd || !tau || !J) return set_err(GRBDA_EINVAL, "null argument");
    if (B == 0) return GRBDA_OK;
    if (step > 0)
        if (const int rc = analytic_derivs<float>(p, q, qd, tau, J, nullptr, nullptr, B, device, stream); rc != 1) return rc;
    GRBDA_CALL_SCOPE(p);
    DeviceTables *t = nullptr;
    if (int rc = ensure_device(p, device, &t)) return rc;
    const size_t nq = p->host.nq, nv = p->host.nv;
    const size_t per_state = nq + 2 * nv + nv * nv;
    size_t chunk = (64u << 20) / (per_state * sizeof(double));
    if (chunk < 1) chunk = 1;
    if (chunk > B) chunk = B;
    void *cvt = nullptr;
    if (int rc = ensure_work(p, p->work_cvt, device, stream, chunk * per_state * sizeof(double) + 256, &cvt)) return rc;
    double *q64 = static_cast<double *>(cvt), *qd64 = q64 + chunk * nq, *tau64 = qd64 + chunk * nv, *J64 = tau64 + chunk * nv;
    hipStream_t hs = static_cast<hipStream_t>(stream);
    auto blocks = [](size_t n) { return static_cast<int>((n + 255) / 256 < 65535 ? (n + 255) / 256 : 65535); };
    for (size_t b0 = 0; b0 < B; b0 += chunk) {
        const size_t nb = B - b0 < chunk ? B - b0 : chunk;
        hipLaunchKernelGGL((convert_kernel<float, double>), dim3(blocks(nb * nq)), dim3(256), 0, hs, q + b0 * nq, q64, nb * nq);
        hipLaunchKernelGGL((convert_kernel<float, double>), dim3(blocks(nb * nv)), dim3(256), 0, hs, qd + b0 * nv, qd64, nb * nv);
        hipLaunchKernelGGL((convert_kernel<float, double>), dim3(blocks(nb * nv)), dim3(256), 0, hs, tau + b0 * nv, tau64, nb * nv);
        if (int rc = derived<double>(p, DM_DQ, q64, qd64, tau64, nullptr, J64, nb, device, stream, step)) return rc;
        hipLaunchKernelGGL((convert_kernel<double, float>), dim3(blocks(nb * nv * nv)), dim3(256), 0, hs, J64, J + b0 * nv * nv, nb * nv * nv);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_err(e, "convert launch");
    }
    return GRBDA_OK;
}
int grbda_fd_derivatives_f64(const grbda_plan *p, const double *q, const double *qd, const double *tau, double *dq, double *dqd,
                             double *dtau, size_t B, int device, void *stream)
{
    if (!p || !q || !qd || !tau) return set_err(GRBDA_EINVAL, "null argument");
    const int rc = analytic_derivs<double>(p, q, qd, tau, dq, dqd, dtau, B, device, stream);
    if (rc != 1) return rc;
    if (dtau) if (const int r2 = grbda_fd_dtau_f64(p, q, dtau, B, device, stream)) return r2;
    if (dqd) if (const int r2 = grbda_fd_dqd_f64(p, q, qd, tau, dqd, B, device, stream)) return r2;
    if (dq) if (const int r2 = grbda_fd_dq_f64(p, q, qd, tau, 1e-6, dq, B, device, stream)) return r2;
    return GRBDA_OK;
}
int grbda_fd_derivatives_f32(const grbda_plan *p, const float *q, const float *qd, const float *tau, float *dq, float *dqd, float *dtau,
                             size_t B, int device, void *stream)
{
    if (!p || !q || !qd || !tau) return set_err(GRBDA_EINVAL, "null argument");
    const int rc = analytic_derivs<float>(p, q, qd, tau, dq, dqd, dtau, B, device, stream);
    if (rc != 1) return rc;
    if (dtau) if (const int r2 = grbda_fd_dtau_f32(p, q, dtau, B, device, stream)) return r2;
    if (dqd) if (const int r2 = grbda_fd_dqd_f32(p, q, qd, tau, dqd, B, device, stream)) return r2;
    if (dq) if (const int r2 = grbda_fd_dq_f32(p, q, qd, tau, 1e-6, dq, B, device, stream)) return r2;
    return GRBDA_OK;
}
int grbda_kernel_name(const grbda_plan *p, int kind, int precision, size_t B, int device, char *buf, size_t cap)
{
    if (!p || !buf || cap == 0 || (kind != 0 && kind != 1) || (precision != 32 && precision != 64)) return set_err(GRBDA_EINVAL, "bad argument");
    GRBDA_CALL_SCOPE(p);
    DeviceTables *t = nullptr;
    if (int rc = ensure_device(p, device, &t)) return rc;
    const std::string name = precision == 32 ? kernel_name_of<float>(p, kind, t->n_cu, B) : kernel_name_of<double>(p, kind, t->n_cu, B);
    std::snprintf(buf, cap, "%s", name.c_str());
    return GRBDA_OK;
}
int grbda_spd_bad_pivots(int device, unsigned long long *count, int reset)
{
    if (!count) return set_err(GRBDA_EINVAL, "null argument");
    DeviceGuard device_guard_;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return set_err(GRBDA_ENODEVICE, "no HIP device available");
    if (device < 0 || device >= n) return set_err(GRBDA_EINVAL, "device index out of range");
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = spd_bad_pivots(count, reset);
    return e == hipSuccess ? GRBDA_OK : hip_err(e, "grbda_spd_bad_pivots");
}
int grbda_body_twists_f64(const grbda_plan *p, const double *q, const double *qd, const double *ydd, double *V, size_t B, int device,
                          void *stream)
{
    return twists<double>(p, q, qd, ydd, V, B, device, stream);
}
int grbda_body_twists_f32(const grbda_plan *p, const float *q, const float *qd, const float *ydd, float *V, size_t B, int device,
                          void *stream)
{
    return twists<float>(p, q, qd, ydd, V, B, device, stream);
}
int grbda_body_poses_f64(const grbda_plan *p, const double *q, double *Xa, size_t B, int device, void *stream)
{
    return poses<double>(p, q, Xa, B, device, stream);
}
int grbda_body_poses_f32(const grbda_plan *p, const float *q, float *Xa, size_t B, int device, void *stream)
{
    return poses<float>(p, q, Xa, B, device, stream);
}
int grbda_apply_test_force_f64(const grbda_plan *p, const double *q, int body, const double offset[3], const double *force,
                               double *lambda_inv, double *dstate, size_t B, int device, void *stream)
{
    return test_force<double>(p, q, body, offset, force, lambda_inv, dstate, B, device, stream);
}
int grbda_apply_test_force_f32(const grbda_plan *p, const float *q, int body, const double offset[3], const float *force,
                               float *lambda_inv, float *dstate, size_t B, int device, void *stream)
{
    return test_force<float>(p, q, body, offset, force, lambda_inv, dstate, B, device, stream);
}
int grbda_inv_osim_f64(const grbda_plan *p, const double *q, int n_contacts, const int *bodies, const double *offsets,
                       double *Linv, double *J, size_t B, int device, void *stream)
{
    return inv_osim<double>(p, q, n_contacts, bodies, offsets, Linv, J, B, device, stream);
}
int grbda_inv_osim_f32(const grbda_plan *p, const float *q, int n_contacts, const int *bodies, const double *offsets,
                       float *Linv, float *J, size_t B, int device, void *stream)
{
    return inv_osim<float>(p, q, n_contacts, bodies, offsets, Linv, J, B, device, stream);
}
int grbda_aba_sharded_f32(const grbda_plan *p, const float *q, const float *qd, const float *tau, float *ydd, size_t B,
                          int n_gpus)
{
    return run_sharded<float>(p, false, q, qd, tau, ydd, B, n_gpus);
}
int grbda_aba_sharded_f64(const grbda_plan *p, const double *q, const double *qd, const double *tau, double *ydd, size_t B,
                          int n_gpus)
{
    return run_sharded<double>(p, false, q, qd, tau, ydd, B, n_gpus);
}
int grbda_rnea_sharded_f32(const grbda_plan *p, const float *q, const float *qd, const float *ydd, float *tau, size_t B,
                           int n_gpus)
{
    return run_sharded<float>(p, true, q, qd, ydd, tau, B, n_gpus);
}
int grbda_aba_sharded_dev_f32(const grbda_plan *p, int n_gpus, const int *devices, const float *const *q, const float *const *qd,
                              const float *const *tau, float *const *ydd, const size_t *B, void *const *streams, float *gathered)
{
    return run_sharded_dev<float>(p, false, n_gpus, devices, q, qd, tau, ydd, B, streams, gathered);
}
int grbda_aba_sharded_dev_f64(const grbda_plan *p, int n_gpus, const int *devices, const double *const *q, const double *const *qd,
                              const double *const *tau, double *const *ydd, const size_t *B, void *const *streams, double *gathered)
{
    return run_sharded_dev<double>(p, false, n_gpus, devices, q, qd, tau, ydd, B, streams, gathered);
}
int grbda_rnea_sharded_dev_f32(const grbda_plan *p, int n_gpus, const int *devices, const float *const *q, const float *const *qd,
                               const float *const *ydd, float *const *tau, const size_t *B, void *const *streams, float *gathered)
{
    return run_sharded_dev<float>(p, true, n_gpus, devices, q, qd, ydd, tau, B, streams, gathered);
}
int grbda_rnea_sharded_dev_f64(const grbda_plan *p, int n_gpus, const int *devices, const double *const *q, const double *const *qd,
                               const double *const *ydd, double *const *tau, const size_t *B, void *const *streams, double *gathered)
{
    return run_sharded_dev<double>(p, true, n_gpus, devices, q, qd, ydd, tau, B, streams, gathered);
}
int grbda_rnea_sharded_f64(const grbda_plan *p, const double *q, const double *qd, const double *ydd, double *tau, size_t B,
                           int n_gpus)
{
    return run_sharded<double>(p, true, q, qd, ydd, tau, B, n_gpus);
}
int grbda_body_poses_host_f64(const grbda_plan *p, const double *q, double *Xa, size_t B, int device)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    GRBDA_CALL_SCOPE(p);
    if (!p || !q || !Xa) return set_err(GRBDA_EINVAL, "null argument");
    DeviceTables *t = nullptr;
    if (int rc = ensure_device(p, device, &t)) return rc;
    const size_t nq = p->host.nq, nb = p->host.n_bodies;
    DevBuf dq, dX;
    int rc;
    if ((rc = dq.alloc(B * nq * 8)) || (rc = dX.alloc(B * nb * 12 * 8)) || (rc = dq.put(q, B * nq * 8))) return rc;
    if ((rc = poses<double>(p, static_cast<double *>(dq.p), static_cast<double *>(dX.p), B, device, nullptr))) return rc;
    if (hipDeviceSynchronize() != hipSuccess) return set_err(GRBDA_EHIP, "kernel execution");
    return dX.get(Xa, B * nb * 12 * 8);
}
int grbda_body_twists_host_f64(const grbda_plan *p, const double *q, const double *qd, const double *ydd, double *V, size_t B, int device)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    GRBDA_CALL_SCOPE(p);
    if (!q || !qd || !ydd || !V) return set_err(GRBDA_EINVAL, "null argument");
    DeviceTables *t = nullptr;
    if (int rc = ensure_device(p, device, &t)) return rc;
    const size_t nq = p->host.nq, nv = p->host.nv, nb = p->host.n_bodies;
    DevBuf dq, dqd, dy, dV;
    int rc;
    if ((rc = dq.alloc(B * nq * 8)) || (rc = dqd.alloc(B * nv * 8)) || (rc = dy.alloc(B * nv * 8)) || (rc = dV.alloc(B * nb * 12 * 8)) ||
        (rc = dq.put(q, B * nq * 8)) || (rc = dqd.put(qd, B * nv * 8)) || (rc = dy.put(ydd, B * nv * 8)))
        return rc;
    if ((rc = twists<double>(p, static_cast<double *>(dq.p), static_cast<double *>(dqd.p), static_cast<double *>(dy.p),
                             static_cast<double *>(dV.p), B, device, nullptr)))
        return rc;
    if (hipDeviceSynchronize() != hipSuccess) return set_err(GRBDA_EHIP, "kernel execution");
    return dV.get(V, B * nb * 12 * 8);
}
int grbda_apply_test_force_host_f64(const grbda_plan *p, const double *q, int body, const double offset[3],
                                    const double *force, double *lambda_inv, double *dstate, size_t B, int device)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    GRBDA_CALL_SCOPE(p);
    if (!p || !q || !force || !lambda_inv || !dstate) return set_err(GRBDA_EINVAL, "null argument");
    DeviceTables *t = nullptr;
    if (int rc = ensure_device(p, device, &t)) return rc;
    const size_t nq = p->host.nq, nv = p->host.nv;
    DevBuf dq, df, dl, dd;
    int rc;
    if ((rc = dq.alloc(B * nq * 8)) || (rc = df.alloc(B * 3 * 8)) || (rc = dl.alloc(B * 8)) || (rc = dd.alloc(B * nv * 8)) ||
        (rc = dq.put(q, B * nq * 8)) || (rc = df.put(force, B * 3 * 8)))
        return rc;
    if ((rc = test_force<double>(p, static_cast<double *>(dq.p), body, offset, static_cast<double *>(df.p),
                                 static_cast<double *>(dl.p), static_cast<double *>(dd.p), B, device, nullptr)))
        return rc;
    if (hipDeviceSynchronize() != hipSuccess) return set_err(GRBDA_EHIP, "kernel execution");
    if ((rc = dl.get(lambda_inv, B * 8))) return rc;
    return dd.get(dstate, B * nv * 8);
}
int grbda_inv_osim_host_f64(const grbda_plan *p, const double *q, int n_contacts, const int *bodies, const double *offsets,
                            double *Linv, double *J, size_t B, int device)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    GRBDA_CALL_SCOPE(p);
    if (!p || !q || !Linv) return set_err(GRBDA_EINVAL, "null argument");
    if (n_contacts < 1 || n_contacts > kMaxContacts) return set_err(GRBDA_EINVAL, "1..8 contact frames per call");
    DeviceTables *t = nullptr;
    if (int rc = ensure_device(p, device, &t)) return rc;
    const size_t nq = p->host.nq, nv = p->host.nv, m = 6 * static_cast<size_t>(n_contacts);
    DevBuf dq, dL, dJ;
    int rc;
    if ((rc = dq.alloc(B * nq * 8)) || (rc = dL.alloc(B * m * m * 8)) || (J && (rc = dJ.alloc(B * m * nv * 8))) ||
        (rc = dq.put(q, B * nq * 8)))
        return rc;
    if ((rc = inv_osim<double>(p, static_cast<double *>(dq.p), n_contacts, bodies, offsets, static_cast<double *>(dL.p),
                               J ? static_cast<double *>(dJ.p) : nullptr, B, device, nullptr)))
        return rc;
    if (hipDeviceSynchronize() != hipSuccess) return set_err(GRBDA_EHIP, "kernel execution");
    if ((rc = dL.get(Linv, B * m * m * 8))) return rc;
    return J ? dJ.get(J, B * m * nv * 8) : GRBDA_OK;
}
// host arrays: mass matrix and the derivatives of the forward dynamics (facade: getMassMatrix, forwardDynamicsDerivativesBatch)
int grbda_mass_matrix_host_f64(const grbda_plan *p, const double *q, double *H, size_t B, int device)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    GRBDA_CALL_SCOPE(p);
    if (!q || !H) return set_err(GRBDA_EINVAL, "null argument");
    DeviceTables *t = nullptr;
    if (int rc0 = ensure_device(p, device, &t)) return rc0;  // `device` current before anything is allocated on it
    const size_t nq = p->host.nq, nv = p->host.nv;
    DevBuf dq, dH;
    int rc;
    if ((rc = dq.alloc(B * nq * 8)) || (rc = dH.alloc(B * nv * nv * 8)) || (rc = dq.put(q, B * nq * 8))) return rc;
    if ((rc = grbda_mass_matrix_f64(p, static_cast<double *>(dq.p), static_cast<double *>(dH.p), B, device, nullptr))) return rc;
    if (hipDeviceSynchronize() != hipSuccess) return set_err(GRBDA_EHIP, "kernel execution");
    return dH.get(H, B * nv * nv * 8);
}
int grbda_fd_derivatives_host_f64(const grbda_plan *p, const double *q, const double *qd, const double *tau, double *dq, double *dqd,
                                  double *dtau, size_t B, int device)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    GRBDA_CALL_SCOPE(p);
    if (!q || !qd || !tau) return set_err(GRBDA_EINVAL, "null argument");
    DeviceTables *t = nullptr;
    if (int rc0 = ensure_device(p, device, &t)) return rc0;  // `device` current before anything is allocated on it
    const size_t nq = p->host.nq, nv = p->host.nv, nn = nv * nv;
    DevBuf bq, bqd, bt, b1, b2, b3;
    int rc;
    if ((rc = bq.alloc(B * nq * 8)) || (rc = bqd.alloc(B * nv * 8)) || (rc = bt.alloc(B * nv * 8)) || (dq && (rc = b1.alloc(B * nn * 8))) ||
        (dqd && (rc = b2.alloc(B * nn * 8))) || (dtau && (rc = b3.alloc(B * nn * 8))) || (rc = bq.put(q, B * nq * 8)) ||
        (rc = bqd.put(qd, B * nv * 8)) || (rc = bt.put(tau, B * nv * 8)))
        return rc;
    if ((rc = grbda_fd_derivatives_f64(p, static_cast<double *>(bq.p), static_cast<double *>(bqd.p), static_cast<double *>(bt.p),
                                       dq ? static_cast<double *>(b1.p) : nullptr, dqd ? static_cast<double *>(b2.p) : nullptr,
                                       dtau ? static_cast<double *>(b3.p) : nullptr, B, device, nullptr)))
        return rc;
    if (hipDeviceSynchronize() != hipSuccess) return set_err(GRBDA_EHIP, "kernel execution");
    if (dq && (rc = b1.get(dq, B * nn * 8))) return rc;
    if (dqd && (rc = b2.get(dqd, B * nn * 8))) return rc;
    return dtau ? b3.get(dtau, B * nn * 8) : GRBDA_OK;
}
int grbda_plan_span_dims(const grbda_plan *p, int *n_span_vel)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    if (n_span_vel) *n_span_vel = span_count(p);
    return GRBDA_OK;
}
int grbda_project_positions_f64(const grbda_plan *p, double *q, int32_t *ok, size_t B, int max_iter, double tol,
                                int device, void *stream)
{
    return project<double>(p, q, ok, B, max_iter, tol, device, stream);
}
int grbda_project_positions_f32(const grbda_plan *p, float *q, int32_t *ok, size_t B, int max_iter, double tol,
                                int device, void *stream)
{
    return project<float>(p, q, ok, B, max_iter, tol, device, stream);
}
int grbda_project_positions_host_f64(const grbda_plan *p, double *q, int32_t *ok, size_t B, int max_iter, double tol, int device)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    GRBDA_CALL_SCOPE(p);
    if (!q) return set_err(GRBDA_EINVAL, "null argument");
    if (B == 0) return GRBDA_OK;
    DeviceTables *t = nullptr;
    if (int rc = ensure_device(p, device, &t)) return rc;
    const size_t nq = p->host.nq;
    DevBuf dq, dok;
    int rc;
    if ((rc = dq.alloc(B * nq * 8)) || (rc = dok.alloc(B * sizeof(int32_t))) || (rc = dq.put(q, B * nq * 8))) return rc;
    if ((rc = project<double>(p, static_cast<double *>(dq.p), static_cast<int32_t *>(dok.p), B, max_iter, tol, device, nullptr))) return rc;
    if (hipDeviceSynchronize() != hipSuccess) return set_err(GRBDA_EHIP, "kernel execution");
    if ((rc = dq.get(q, B * nq * 8))) return rc;
    return ok ? dok.get(ok, B * sizeof(int32_t)) : GRBDA_OK;
}
int grbda_state_input_dims(const grbda_plan *p, const uint8_t *pos_is_spanning, const uint8_t *vel_is_spanning, int *in_nq, int *in_nv)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    GRBDA_CALL_SCOPE(p);
    return state_widths(p, pos_is_spanning, vel_is_spanning, nullptr, in_nq, in_nv);
}
int grbda_state_to_independent_f64(const grbda_plan *p, const uint8_t *pos_is_spanning, const uint8_t *vel_is_spanning, const double *q_in,
                                   const double *qd_in, double *q, double *qd, int32_t *status, double *cond, size_t B, double tol,
                                   int device, void *stream)
{
    return state_convert<double>(p, pos_is_spanning, vel_is_spanning, q_in, qd_in, q, qd, status, cond, B, tol, device, stream);
}
int grbda_state_to_independent_f32(const grbda_plan *p, const uint8_t *pos_is_spanning, const uint8_t *vel_is_spanning, const float *q_in,
                                   const float *qd_in, float *q, float *qd, int32_t *status, float *cond, size_t B, double tol,
                                   int device, void *stream)
{
    return state_convert<float>(p, pos_is_spanning, vel_is_spanning, q_in, qd_in, q, qd, status, cond, B, tol, device, stream);
}
int grbda_state_to_independent_host_f64(const grbda_plan *p, const uint8_t *pos_is_spanning, const uint8_t *vel_is_spanning,
                                        const double *q_in, const double *qd_in, double *q, double *qd, size_t B, double tol, int device)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    GRBDA_CALL_SCOPE(p);
    if (!q_in || !qd_in || !q || !qd) return set_err(GRBDA_EINVAL, "null argument");
    int in_nq = 0, in_nv = 0;
    if (int rc = state_widths(p, pos_is_spanning, vel_is_spanning, nullptr, &in_nq, &in_nv)) return rc;
    if (B == 0) return GRBDA_OK;
    DeviceTables *t = nullptr;
    if (int rc = ensure_device(p, device, &t)) return rc;
    const size_t nq = p->host.nq, nv = p->host.nv;
    DevBuf bqi, bvi, bq, bv, bs;
    int rc;
    if ((rc = bqi.alloc(B * in_nq * 8)) || (rc = bvi.alloc(B * in_nv * 8)) || (rc = bq.alloc(B * nq * 8)) || (rc = bv.alloc(B * nv * 8)) ||
        (rc = bs.alloc(B * 4)) || (rc = bqi.put(q_in, B * in_nq * 8)) || (rc = bvi.put(qd_in, B * in_nv * 8)))
        return rc;
    if ((rc = state_convert<double>(p, pos_is_spanning, vel_is_spanning, static_cast<const double *>(bqi.p), static_cast<const double *>(bvi.p),
                                    static_cast<double *>(bq.p), static_cast<double *>(bv.p), static_cast<int32_t *>(bs.p), nullptr, B, tol,
                                    device, nullptr)))
        return rc;
    if (hipDeviceSynchronize() != hipSuccess) return set_err(GRBDA_EHIP, "kernel execution");
    std::vector<int32_t> st(B);
    if ((rc = bs.get(st.data(), B * 4)) || (rc = bq.get(q, B * nq * 8)) || (rc = bv.get(qd, B * nv * 8))) return rc;
    for (size_t b = 0; b < B; b++)
        if (st[b]) {
            const int code = st[b] & 255, cluster = st[b] >> 8;
            return set_err(GRBDA_ESTATE, "state " + std::to_string(b) + ", cluster " + std::to_string(cluster) + ": " +
                                             (code == 1 ? "Spanning position is not valid" : "Spanning velocity is not valid"));
        }
    return GRBDA_OK;
}
int grbda_spanning_f64(const grbda_plan *p, const double *q, const double *qd, const double *ydd, double *qd_span,
                       double *qdd_span, size_t B, int device, void *stream)
{
    return spanning<double>(p, q, qd, ydd, qd_span, qdd_span, B, device, stream);
}
int grbda_spanning_f32(const grbda_plan *p, const float *q, const float *qd, const float *ydd, float *qd_span,
                       float *qdd_span, size_t B, int device, void *stream)
{
    return spanning<float>(p, q, qd, ydd, qd_span, qdd_span, B, device, stream);
}

int grbda_aba_host_f64(const grbda_plan *p, const double *q, const double *qd, const double *tau,
                       const double *f_ext, double *ydd, size_t B, int device)
{
    return run_host_f64(p, false, q, qd, tau, f_ext, ydd, B, device);
}
int grbda_rnea_host_f64(const grbda_plan *p, const double *q, const double *qd, const double *ydd,
                        const double *f_ext, double *tau, size_t B, int device)
{
    return run_host_f64(p, true, q, qd, ydd, f_ext, tau, B, device);
}

int grbda_time_kernel(const grbda_plan *p, int kind, int precision, const void *q, const void *qd, const void *x,
                      void *out, size_t B, int device, void *stream, int iters, float *avg_ms)
{
    if (!p) return set_err(GRBDA_EINVAL, "null plan");
    GRBDA_CALL_SCOPE(p);
    if (!avg_ms || iters <= 0 || (precision != 32 && precision != 64) || (kind != 0 && kind != 1))
        return set_err(GRBDA_EINVAL, "bad timing arguments");
    DeviceTables *t = nullptr;
    if (int rc = ensure_device(p, device, &t)) return rc;
    hipEvent_t e0, e1;
    hipError_t e;
    if ((e = hipEventCreate(&e0)) != hipSuccess || (e = hipEventCreate(&e1)) != hipSuccess) return hip_err(e, "hipEventCreate");
    auto once = [&]() -> int {
        if (precision == 32)
            return run<float>(p, kind == 1, static_cast<const float *>(q), static_cast<const float *>(qd),
                              static_cast<const float *>(x), nullptr, static_cast<float *>(out), B, device, stream);
        return run<double>(p, kind == 1, static_cast<const double *>(q), static_cast<const double *>(qd),
                           static_cast<const double *>(x), nullptr, static_cast<double *>(out), B, device, stream);
    };
    int rc = once();  // warm-up: uploads tables, sizes scratch
    if (rc == GRBDA_OK) {
        hipStream_t s = static_cast<hipStream_t>(stream);
        if ((e = hipEventRecord(e0, s)) != hipSuccess) rc = hip_err(e, "hipEventRecord");
        for (int i = 0; i < iters && rc == GRBDA_OK; i++) rc = once();
        if (rc == GRBDA_OK && (e = hipEventRecord(e1, s)) != hipSuccess) rc = hip_err(e, "hipEventRecord");
        if (rc == GRBDA_OK && (e = hipEventSynchronize(e1)) != hipSuccess) rc = hip_err(e, "hipEventSynchronize");
        float ms = 0;
        if (rc == GRBDA_OK && (e = hipEventElapsedTime(&ms, e0, e1)) != hipSuccess) rc = hip_err(e, "hipEventElapsedTime");
        if (rc == GRBDA_OK) *avg_ms = ms / static_cast<float>(iters);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

// Experiment support (tools/spec_experiment.py): the f32 fast-path tables of a plan as C initialisers.
int grbda_debug_dump_plan(const grbda_plan *p, const char *path)
{
    if (!p || !path) return set_err(GRBDA_EINVAL, "null argument");
    FILE *f = std::fopen(path, "w");
    if (!f) return set_err(GRBDA_EINVAL, "cannot open the dump file");
    const HostPlan &h = p->host;
    const Layout &L = h.lay32;
    auto ints = [&](const char *type, const char *name, const void *data, size_t count, size_t per) {
        const int32_t *v = static_cast<const int32_t *>(data);
        std::fprintf(f, "__constant__ const %s %s[] = {\n", type, name);
        for (size_t i = 0; i < count; i++) {
            std::fprintf(f, "  {");
            for (size_t j = 0; j < per; j++) std::fprintf(f, "%d,", v[i * per + j]);
            std::fprintf(f, "},\n");
        }
        std::fprintf(f, "};\n");
    };
    std::fprintf(f, "constexpr int kSpecNq = %d, kSpecNv = %d, kSpecSteps = %d, kSpecLds = %d, kSpecGlb = %d, kSpecOri = %d;\n",
                 h.nq, h.nv, (int)h.aba_steps.size(), L.n_lds_aba, L.n_glb_aba, h.ori_repr);
    std::fprintf(f, "constexpr float kSpecARoot[6] = {");
    for (int i = 0; i < 6; i++) std::fprintf(f, "%.9ef,", (float)-h.gravity[i]);
    std::fprintf(f, "};\n");
    // records as flat int initialisers (the structs are all-int32 PODs; nested arrays written as scalars)
    std::fprintf(f, "__constant__ const int32_t kSpecStepsRaw[] = {");
    for (const Step &st : h.aba_steps) std::fprintf(f, "%d,%d,0,0,", st.op, st.cluster);
    std::fprintf(f, "};\n");
    auto raw = [&](const char *name, const void *data, size_t n_ints) {
        const int32_t *v = static_cast<const int32_t *>(data);
        std::fprintf(f, "__constant__ const int32_t %s[] = {", name);
        for (size_t i = 0; i < n_ints; i++) std::fprintf(f, "%d,", v[i]);
        std::fprintf(f, "};\n");
    };
    raw("kSpecClustersRaw", L.clusters.data(), L.clusters.size() * sizeof(ClusterRec) / 4);
    raw("kSpecBodiesRaw", L.bodies.data(), L.bodies.size() * sizeof(BodyRec) / 4);
    raw("kSpecAccK", L.acc_k.data(), L.acc_k.size());
    std::fprintf(f, "__constant__ const float kSpecConsts[] = {");
    for (double c : h.consts) std::fprintf(f, "%.9ef,", (float)c);
    std::fprintf(f, "};\n");
    (void)ints;
    std::fclose(f);
    return GRBDA_OK;
}

int grbda_device_count(void)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count < 0 ? 0 : count;
}

}  // extern "C"
