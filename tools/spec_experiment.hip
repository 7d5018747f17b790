// Experiment (tools/spec_experiment.py; build: dump tables with grbda_debug_dump_plan into tools/spec_tables.inc,
// then hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -shared tools/spec_experiment.hip
// -o build/spec/libspec.so -- about 25 minutes): the ABA kernel specialised for ONE plan -- tables are compile-time constants, the step loop is
// fully unrolled, so every record field, slot number and model constant folds into the instruction stream.
#include "../generalized_rbda_amd/csrc/kernels.hip"
namespace grbda_hip {
#include "spec_tables.inc"

template <int S>
__device__ __forceinline__ void spec_step(const Tables<float> &P, const Slots<float> &Sl, Lane<float> &L, Carry<float> &carry,
                                          float (&kpre)[7], bool &kpre_valid)
{
    using T = float;
    constexpr bool HAS_LOOP = false;
    const bool use_shapes = true;
    const Step st = load_rec(P.steps + S);
    if (st.op & kOpSkipFast) return;
    const int op = st.op & kOpMask;
    const ClusterRec c = load_rec(P.clusters + st.cluster);
    const int s = S;
    if (op == OP_ABA_FWD) {
        if (c.kind == CK_FREE) aba_fwd_free(P, Sl, c, L);
        else if (c.shape) aba_fwd_rev<T>(P, Sl, c, L);
        else { GRBDA_DISPATCH_N(c, aba_fwd_static, P, Sl, c, L) }
    } else if (op == OP_ABA_BWD) {
        if (c.kind == CK_FREE) aba_bwd_free(P, Sl, c, L, carry);
        else if (c.shape == SHAPE_REV) aba_bwd_rev<T, false>(P, Sl, c, L, carry);
        else if (c.shape == SHAPE_REV_ROTOR) aba_bwd_rev<T, true>(P, Sl, c, L, carry);
        else { GRBDA_DISPATCH_N(c, aba_bwd_static, P, Sl, c, L, carry) }
    } else {
        if (!c.shape) {
            const int knext = P.acc_k[s + 1];
            kpre_valid = knext != -1;
            if (kpre_valid) Sl.ld(knext, kpre);
        }
        if (c.kind == CK_FREE) aba_acc_free(P, Sl, c, L);
        else if (c.shape) {
            T kblk[7];
#pragma unroll
            for (int j = 0; j < 7; j++) kblk[j] = kpre[j];
            const bool have = kpre_valid;
            const int knext = P.acc_k[s + 1];
            kpre_valid = knext != -1;
            if (kpre_valid) Sl.ld(knext, kpre);
            aba_acc_rev<T>(P, Sl, c, L, kblk, have);
        } else { GRBDA_DISPATCH_N(c, aba_acc_static, P, Sl, c, L) }
    }
    (void)use_shapes;
}
template <int S0, int S1>
__device__ __forceinline__ void spec_steps(const Tables<float> &P, const Slots<float> &Sl, Lane<float> &L, Carry<float> &carry,
                                           float (&kpre)[7], bool &kv)
{
    if constexpr (S0 < S1) {
        spec_step<S0>(P, Sl, L, carry, kpre, kv);
        spec_steps<S0 + 1, S1>(P, Sl, L, carry, kpre, kv);
    }
}

__global__ __launch_bounds__(kWave, 2) void aba_kernel_spec(const float *__restrict__ q, const float *__restrict__ qd,
                                                            const float *__restrict__ tau, float *__restrict__ ydd,
                                                            size_t B, float *__restrict__ scratch, int lds_bytes)
{
    using T = float;
    Tables<T> P;
    P.steps = (cptr<Step>)kSpecStepsRaw;
    P.clusters = (cptr<ClusterRec>)kSpecClustersRaw;
    P.bodies = (cptr<BodyRec>)kSpecBodiesRaw;
    P.consts = (cptr<T>)kSpecConsts;
    P.cints = nullptr;
    P.acc_k = (cptr<int32_t>)kSpecAccK;
    P.n_steps = kSpecSteps;
    P.nq = kSpecNq;
    P.nv = kSpecNv;
    P.ori_repr = kSpecOri;
    for (int i = 0; i < 6; i++) P.a_root[i] = kSpecARoot[i];
    const int lane = threadIdx.x;
    Slots<T> S;
    S.lane = lane;
    T *slab = scratch + (size_t)blockIdx.x * (size_t)(kSpecGlb + kSpecNq + 2 * kSpecNv) * kWave;
    S.glb = slab + (size_t)(kSpecNq + 2 * kSpecNv) * kWave;
    const size_t n_tiles = (B + kWave - 1) / kWave;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t r = tile * kWave + lane;
        const size_t left = B - tile * kWave;
        const int rows_valid = left < (size_t)kWave ? (int)left : kWave;
        stage_inputs(q, qd, tau, tile, rows_valid, P.nq, P.nv, slab, lane, lds_bytes);
        Lane<T> L;
        L.active = r < B;
        L.in_q = slab + lane;
        L.in_qd = slab + (size_t)P.nq * kWave + lane;
        L.in_x = slab + (size_t)(P.nq + P.nv) * kWave + lane;
        L.out_rows = slab + (size_t)(P.nq + P.nv) * kWave + lane;
        L.fext = nullptr;
        L.lane = lane;
        Carry<T> carry;
#pragma unroll
        for (int j = 0; j < 21; j++) carry.IA[j] = 0;
#pragma unroll
        for (int j = 0; j < 6; j++) carry.psi[j] = 0;
        T kpre[7] = {0, 0, 0, 0, 0, 0, 0};
        bool kv = false;
        spec_steps<0, kSpecSteps>(P, S, L, carry, kpre, kv);
        write_outputs(slab + (size_t)(P.nq + P.nv) * kWave, ydd, tile, rows_valid, P.nv, lane);
    }
}
}  // namespace grbda_hip

extern "C" int spec_launch_aba_f32(const float *q, const float *qd, const float *tau, float *ydd, size_t B, float *scratch,
                                   int grid, int lds_bytes, void *stream)
{
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&grbda_hip::aba_kernel_spec),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once = true;
    }
    hipLaunchKernelGGL(grbda_hip::aba_kernel_spec, dim3(grid), dim3(64), lds_bytes, (hipStream_t)stream, q, qd, tau, ydd, B,
                       scratch, lds_bytes);
    return (int)hipGetLastError();
}
extern "C" int spec_scratch_rows(void) { return grbda_hip::kSpecGlb + grbda_hip::kSpecNq + 2 * grbda_hip::kSpecNv; }
extern "C" int spec_lds_slots(void) { return grbda_hip::kSpecLds; }
