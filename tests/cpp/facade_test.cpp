// facade_test.cpp -- the reference's own construction code, compiled against the facade.
//
// The builder below is written the way src/Robots/SerialChains/RevoluteChainWithRotor.cpp:45-109 and
// RevolutePairChainWithRotor.cpp:62-138 use the reference API (registerBody /
// appendRegisteredBodiesAsCluster<JointT> / setState / forwardDynamics / inverseDynamics).
//   facade_test --dump <dir>   : write model-description blobs (CPU only)
//   facade_test --run <urdf>   : run dynamics through the HIP path and check ID(FD(tau)) == tau
#include <cmath>
#include <cstdio>
#include <fstream>
#include <string>

#include "grbda/Dynamics/ClusterTreeModel.h"

// (device-array mode only: the HIP runtime API for allocation and copies, compiled by g++ -- no device code in this file)
#ifdef FACADE_TEST_WITH_HIP
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#endif

using namespace grbda;

template <size_t N>
void buildRevoluteChainWithRotor(ClusterTreeModel<double> &model)
{
    using RevoluteWithRotor = ClusterJoints::RevoluteWithRotor<double>;
    using TransmissionModule = ClusterJoints::GearedTransmissionModule<double>;
    const Mat3<double> I3 = Mat3<double>::Identity();
    const Vec3<double> z3 = Vec3<double>::Zero();
    model.setGravity(Vec3<double>{9.81, 0., 0.});
    const double I = 1., Irot = 1e-4, m = 1., l = 1., c = 0.5, gr = 2., br = 3.;
    const ori::CoordinateAxis axis = ori::CoordinateAxis::Z;
    Mat3<double> link_inertia, rotor_inertia;
    link_inertia(2, 2) = I;
    rotor_inertia(2, 2) = Irot;
    const SpatialInertia<double> link_spatial_inertia(m, Vec3<double>{c, 0., 0.}, link_inertia);
    const SpatialInertia<double> rotor_spatial_inertia(0., Vec3<double>::Zero(), rotor_inertia);
    std::string prev_link_name = "ground";
    for (size_t i = 0; i < N; i++) {
        const spatial::Transform<double> Xtree = i == 0 ? spatial::Transform<double>(I3, z3)
                                                        : spatial::Transform<double>(I3, Vec3<double>{l, 0., 0.});
        const std::string link_name = "link-" + std::to_string(i);
        auto link = model.registerBody(link_name, link_spatial_inertia, prev_link_name, Xtree);
        const std::string rotor_name = "rotor-" + std::to_string(i);
        auto rotor = model.registerBody(rotor_name, rotor_spatial_inertia, prev_link_name, Xtree);
        TransmissionModule module{link, rotor, "link-joint-" + std::to_string(i), "rotor-joint-" + std::to_string(i),
                                  axis, axis, gr * br};
        model.appendRegisteredBodiesAsCluster<RevoluteWithRotor>("cluster-" + std::to_string(i), module);
        prev_link_name = link_name;
    }
}

template <size_t N>
void buildRevolutePairChainWithRotor(ClusterTreeModel<double> &model)
{
    using RevPairRotor = ClusterJoints::RevolutePairWithRotor<double>;
    using ProxTransModule = ClusterJoints::ParallelBeltTransmissionModule<1, double>;
    using DistTransModule = ClusterJoints::ParallelBeltTransmissionModule<2, double>;
    const Mat3<double> I3 = Mat3<double>::Identity();
    model.setGravity(Vec3<double>{9.81, 0., 0.});
    const double gr = 2., br = 3.;
    const ori::CoordinateAxis axis = ori::CoordinateAxis::Z;
    const spatial::Transform<double> Xtree2(I3, Vec3<double>{1., 0., 0.});
    Mat3<double> link_inertia, rotor_inertia;
    link_inertia(2, 2) = 1.;
    rotor_inertia(2, 2) = 1e-4;
    const SpatialInertia<double> link_si(1., Vec3<double>{0.5, 0., 0.}, link_inertia);
    const SpatialInertia<double> rotor_si(0., Vec3<double>::Zero(), rotor_inertia);
    std::string parent_name = "ground";
    for (size_t i = 0; i < N / 2; i++) {
        const spatial::Transform<double> Xtree1 = i == 0 ? spatial::Transform<double>(I3, Vec3<double>::Zero()) : Xtree2;
        const std::string s = std::to_string(i);
        auto linkA = model.registerBody("link-A-" + s, link_si, parent_name, Xtree1);
        auto rotorA = model.registerBody("rotor-A-" + s, rotor_si, parent_name, Xtree1);
        auto rotorB = model.registerBody("rotor-B-" + s, rotor_si, parent_name, Xtree1);
        auto linkB = model.registerBody("link-B-" + s, link_si, "link-A-" + s, Xtree2);
        ProxTransModule moduleA{linkA, rotorA, axis, axis, gr, Vec1<double>{br}};
        DistTransModule moduleB{linkB, rotorB, axis, axis, gr, Vec2<double>{br, 1.}};
        model.appendRegisteredBodiesAsCluster<RevPairRotor>("cluster-" + s, moduleA, moduleB);
        parent_name = "link-B-" + s;
    }
}

#ifdef FACADE_TEST_WITH_HIP
// ClusterTreeModel<Scalar>::forwardDynamicsBatch / inverseDynamicsBatch on DEVICE arrays of the model's Scalar: the float model runs
// grbda_aba_f32 / grbda_rnea_f32, the double model the _f64 entry points; float against double to 1e-3 (BASELINE's fp32 tolerance),
// double device arrays against the host-array overload exactly, ID(FD(tau)) = tau on the device in both precisions.
template <class S>
static int deviceBatch(const std::string &urdf, const std::vector<double> &q, const std::vector<double> &qd, const std::vector<double> &tau,
                       size_t B, std::vector<double> &ydd_out, double id_tol)
{
    ClusterTreeModel<S> m(urdf);
    const size_t nq = m.getNumPositions(), nv = m.getNumDegreesOfFreedom();
    std::vector<S> hq(q.begin(), q.end()), hqd(qd.begin(), qd.end()), htau(tau.begin(), tau.end()), hydd(B * nv), hback(B * nv);
    S *dq = nullptr, *dqd = nullptr, *dtau = nullptr, *dydd = nullptr, *dback = nullptr;
    hipStream_t stream = nullptr;
    if (hipSetDevice(0) != hipSuccess || hipStreamCreate(&stream) != hipSuccess) return 1;
    if (hipMalloc((void **)&dq, B * nq * sizeof(S)) != hipSuccess || hipMalloc((void **)&dqd, B * nv * sizeof(S)) != hipSuccess ||
        hipMalloc((void **)&dtau, B * nv * sizeof(S)) != hipSuccess || hipMalloc((void **)&dydd, B * nv * sizeof(S)) != hipSuccess ||
        hipMalloc((void **)&dback, B * nv * sizeof(S)) != hipSuccess)
        return 1;
    (void)hipMemcpy(dq, hq.data(), B * nq * sizeof(S), hipMemcpyHostToDevice);
    (void)hipMemcpy(dqd, hqd.data(), B * nv * sizeof(S), hipMemcpyHostToDevice);
    (void)hipMemcpy(dtau, htau.data(), B * nv * sizeof(S), hipMemcpyHostToDevice);
    m.forwardDynamicsBatch(dq, dqd, dtau, dydd, B, 0, stream);
    m.inverseDynamicsBatch(dq, dqd, dydd, dback, B, 0, stream);
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    (void)hipMemcpy(hydd.data(), dydd, B * nv * sizeof(S), hipMemcpyDeviceToHost);
    (void)hipMemcpy(hback.data(), dback, B * nv * sizeof(S), hipMemcpyDeviceToHost);
    (void)hipFree(dq); (void)hipFree(dqd); (void)hipFree(dtau); (void)hipFree(dydd); (void)hipFree(dback);
    (void)hipStreamDestroy(stream);
    ydd_out.assign(hydd.begin(), hydd.end());
    double worst = 0;
    for (size_t i = 0; i < B * nv; i++) worst = std::max(worst, std::fabs(static_cast<double>(hback[i]) - tau[i]) / (1.0 + std::fabs(tau[i])));
    std::printf("  %s device batch of %zu states: max |ID(FD(tau)) - tau| / (1 + |tau|) = %.3g\n", sizeof(S) == 4 ? "float " : "double", B, worst);
    return worst < id_tol ? 0 : 1;
}

static int deviceArrays(const std::string &urdf)
{
    ClusterTreeModel<double> md(urdf);
    const size_t B = 4099, nq = md.getNumPositions(), nv = md.getNumDegreesOfFreedom();
    std::vector<double> q(B * nq), qd(B * nv), tau(B * nv);
    unsigned long long sd = 0x9E3779B97F4A7C15ull;
    auto uni = [&]() { sd = sd * 6364136223846793005ull + 1442695040888963407ull; return (double)(sd >> 11) / 9007199254740992.0 * 2.0 - 1.0; };
    for (size_t s = 0; s < B; s++) {
        for (size_t j = 0; j < nq; j++) q[s * nq + j] = uni();
        if (nq == nv + 1) {  // floating base: a unit quaternion in the last four of its seven positions (Joint.h:61-68)
            double nrm = 0;
            for (int j = 3; j < 7; j++) nrm += q[s * nq + j] * q[s * nq + j];
            for (int j = 3; j < 7; j++) q[s * nq + j] /= std::sqrt(nrm);
        }
        for (size_t j = 0; j < nv; j++) { qd[s * nv + j] = uni(); tau[s * nv + j] = uni(); }
    }
    std::vector<double> y32, y64, yhost(B * nv);
    int rc = deviceBatch<float>(urdf, q, qd, tau, B, y32, 5e-3) | deviceBatch<double>(urdf, q, qd, tau, B, y64, 1e-8);
    md.forwardDynamicsBatch(q.data(), qd.data(), tau.data(), yhost.data(), B);  // host-array overload of the same model
    double d_host = 0, d_f32 = 0, scale = 0;
    for (size_t i = 0; i < B * nv; i++) scale = std::max(scale, std::fabs(y64[i]));
    for (size_t i = 0; i < B * nv; i++) {
        d_host = std::max(d_host, std::fabs(y64[i] - yhost[i]));
        d_f32 = std::max(d_f32, std::fabs(y32[i] - y64[i]) / (1.0 + std::fabs(y64[i])));
    }
    std::printf("  double device vs host overload: %.3g;  float vs double: %.3g relative (1 + |ydd|), max |ydd| %.3g\n", d_host, d_f32, scale);
    return rc | (d_host == 0.0 ? 0 : 1) | (d_f32 < 1e-3 ? 0 : 1);
}
#endif

static void dump(const std::string &path, const std::vector<unsigned char> &blob)
{
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char *>(blob.data()), static_cast<std::streamsize>(blob.size()));
}

template <class Model>
static int roundtrip(Model &model, const char *what, const std::string &cp_body_name = "")
{
    const int nq = model.getNumPositions(), nv = model.getNumDegreesOfFreedom();
    double worst = 0;
    DVec<double> last_q, last_qd;
    for (int trial = 0; trial < 5; trial++) {
        DVec<double> q = DVec<double>::Random(nq), qd = DVec<double>::Random(nv), tau = DVec<double>::Random(nv);
        if (nq == nv + 1) {  // floating base: a valid unit quaternion in the last 4 of the first 7 positions
            Quat<double> quat = ori::rpyToQuat(Vec3<double>{q[3], q[4], q[5]});
            for (int i = 0; i < 4; i++) q[3 + i] = quat[i];
        }
        model.setState(std::make_pair(q, qd));
        last_q = q;
        last_qd = qd;
        const DVec<double> ydd = model.forwardDynamics(tau);
        const DVec<double> back = model.inverseDynamics(ydd);
        worst = std::fmax(worst, (back - tau).norm());
        // H ydd + C = tau (getMassMatrix / getBiasForceVector, testRigidBodyDynamicsAlgos.cpp:208-232)
        const DMat<double> H = model.getMassMatrix();
        const DVec<double> C = model.getBiasForceVector();
        double res = 0;
        for (int i = 0; i < nv; i++) {
            double s = C[i] - tau[i];
            for (int j = 0; j < nv; j++) s += H(i, j) * ydd[j];
            res += s * s;
        }
        worst = std::fmax(worst, std::sqrt(res));
        // derivatives of the forward dynamics (forwardDynamicsDerivativesBatch): H (d ydd / d tau) = 1, and
        // d ydd / d qd against the exact unit central difference (the forward dynamics is quadratic in qd)
        std::vector<double> qv(q.data(), q.data() + nq), qdv(qd.data(), qd.data() + nv), tv(tau.data(), tau.data() + nv);
        std::vector<double> dtau(static_cast<size_t>(nv) * nv), dqd(static_cast<size_t>(nv) * nv);
        model.forwardDynamicsDerivativesBatch(qv.data(), qdv.data(), tv.data(), nullptr, dqd.data(), dtau.data(), 1);
        for (int i = 0; i < nv; i++)
            for (int j = 0; j < nv; j++) {
                double s = i == j ? -1.0 : 0.0;
                for (int k = 0; k < nv; k++) s += H(i, k) * dtau[static_cast<size_t>(k) * nv + j];
                worst = std::fmax(worst, std::fabs(s));
            }
        {
            DVec<double> qp = qd, qm = qd;
            qp[nv - 1] += 1.0;
            qm[nv - 1] -= 1.0;
            model.setState(std::make_pair(q, qp));
            const DVec<double> yp = model.forwardDynamics(tau);
            model.setState(std::make_pair(q, qm));
            const DVec<double> ym = model.forwardDynamics(tau);
            model.setState(std::make_pair(q, qd));
            for (int i = 0; i < nv; i++)
                worst = std::fmax(worst, std::fabs(0.5 * (yp[i] - ym[i]) - dqd[static_cast<size_t>(i) * nv + nv - 1]) / (1 + std::fabs(yp[i])));
        }
    }
    // contact side: applyTestForce against H^-1 (dstate = H^-1 J^T f is what a unit impulse does), and the
    // end-effector operational-space inertia against it (testRigidBodyDynamicsAlgos.cpp:241-335)
    if (!model.contactPoints().empty()) {
        const auto &cp = model.contactPoints().front();
        model.forwardKinematicsIncludingContactPoints();
        const Vec3<double> f{0.3, -0.7, 0.5};
        DVec<double> ds;
        const double lam = model.applyTestForce(cp.name_, f, ds);
        const DMat<double> H = model.getMassMatrix();
        // J^T f = H dstate ; lambda = (J^T f) . dstate
        double lam2 = 0;
        for (int i = 0; i < nv; i++) {
            double jtf = 0;
            for (int j = 0; j < nv; j++) jtf += H(i, j) * ds[j];
            lam2 += jtf * ds[i];
        }
        worst = std::fmax(worst, std::fabs(lam - lam2) / (1 + std::fabs(lam)));
        const DMat<double> Linv = model.inverseOperationalSpaceInertiaMatrix();
        if (Linv.rows() != 6 * model.getNumEndEffectors()) return 1;
        for (int i = 0; i < Linv.rows(); i++) {
            if (!(Linv(i, i) > -1e-12)) return 1;  // positive semi-definite: a planar chain cannot move out of its plane
            for (int j = 0; j < Linv.cols(); j++) worst = std::fmax(worst, std::fabs(Linv(i, j) - Linv(j, i)) / (1 + std::fabs(Linv(i, i))));
        }
        std::printf("%s: contact point at (%.3f, %.3f, %.3f), lambda_inv = %.6f\n", what, cp.position_[0], cp.position_[1],
                    cp.position_[2], lam);
        // contact Jacobians (ClusterTreeDynamics.cpp:10-79): J_world = diag(R, R) J_body with R = getOrientation; its linear rows
        // give J^T f = H dstate of applyTestForce and the point's velocity; on fixed-base models the velocity is also the
        // central difference of the point's position along qd
        const DMat<double> Jw = model.contactJacobianWorldFrame(cp.name_);
        const DMat<double> Jb = model.contactJacobianBodyFrame(cp.name_);
        if (Jw.rows() != 6 || Jw.cols() != nv || Jb.rows() != 6 || Jb.cols() != nv) return 1;
        const DVec<double> H_ds = [&] { DVec<double> r = DVec<double>::Zero(nv); for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) r[i] += H(i, j) * ds[j]; return r; }();
        for (int k = 0; k < nv; k++) {
            double jtf = 0;
            for (int i = 0; i < 3; i++) jtf += Jw(3 + i, k) * f[i];
            worst = std::fmax(worst, std::fabs(jtf - H_ds[k]) / (1 + std::fabs(jtf)));
        }
        model.updateContactPointJacobians();
        const auto &cp2 = model.contactPoint(cp.name_);
        double v_from_J[3] = {0, 0, 0}, w_from_J[3] = {0, 0, 0};
        for (int i = 0; i < 3; i++)
            for (int k = 0; k < nv; k++) {
                v_from_J[i] += cp2.jacobian_(3 + i, k) * last_qd[k];
                w_from_J[i] += cp2.jacobian_(i, k) * last_qd[k];
            }
        for (int i = 0; i < 3; i++) worst = std::fmax(worst, std::fabs(v_from_J[i] - cp2.velocity_[i]));
        if (nq == nv && !cp_body_name.empty()) {  // (hand-built fixed-base chains: the test knows the body's name)
            const Mat3<double> R = model.getOrientation(cp_body_name);
            for (int half = 0; half < 2; half++)
                for (int i = 0; i < 3; i++)
                    for (int k = 0; k < nv; k++) {
                        double s = 0;
                        for (int m2 = 0; m2 < 3; m2++) s += R(i, m2) * Jb(3 * half + m2, k);
                        worst = std::fmax(worst, std::fabs(s - Jw(3 * half + i, k)));
                    }
            const Vec3<double> vl = model.getLinearVelocity(cp_body_name, cp.local_offset_), wl = model.getAngularVelocity(cp_body_name);
            for (int i = 0; i < 3; i++) worst = std::fmax(worst, std::fmax(std::fabs(vl[i] - v_from_J[i]), std::fabs(wl[i] - w_from_J[i])));
            const double h = 1e-6;
            DVec<double> qp = last_q, qm = last_q;
            for (int k = 0; k < nv; k++) { qp[k] += h * last_qd[k]; qm[k] -= h * last_qd[k]; }
            model.setState(std::make_pair(qp, last_qd));
            const Vec3<double> pp = model.getPosition(cp_body_name, cp.local_offset_);
            model.setState(std::make_pair(qm, last_qd));
            const Vec3<double> pm = model.getPosition(cp_body_name, cp.local_offset_);
            model.setState(std::make_pair(last_q, last_qd));
            for (int i = 0; i < 3; i++) worst = std::fmax(worst, std::fabs((pp[i] - pm[i]) / (2 * h) - v_from_J[i]) * 1e-2);  // O(h^2) + rounding / h
            std::printf("%s: contact velocity (%.4f, %.4f, %.4f) = J qd = d position / dt\n", what, v_from_J[0], v_from_J[1], v_from_J[2]);
            // contact acceleration (TreeModel.cpp:78-99): the second central difference of the point's position along
            // q(t) = q + qd t + qdd t^2 / 2 is its classical acceleration; getLinearAcceleration is the same without gravity
            const DVec<double> qdd = DVec<double>::Random(nv);
            model.forwardAccelerationKinematicsIncludingContactPoints(qdd);
            const Vec3<double> acc = model.contactPoint(cp.name_).acceleration_;
            const Vec3<double> lin = model.getLinearAcceleration(qdd, cp_body_name, cp.local_offset_);
            const double h2 = 1e-4;
            DVec<double> q2p = last_q, q2m = last_q;
            for (int k = 0; k < nv; k++) {
                q2p[k] += h2 * last_qd[k] + 0.5 * h2 * h2 * qdd[k];
                q2m[k] += -h2 * last_qd[k] + 0.5 * h2 * h2 * qdd[k];
            }
            const Vec3<double> p0 = model.getPosition(cp_body_name, cp.local_offset_);
            model.setState(std::make_pair(q2p, last_qd));
            const Vec3<double> p2p = model.getPosition(cp_body_name, cp.local_offset_);
            model.setState(std::make_pair(q2m, last_qd));
            const Vec3<double> p2m = model.getPosition(cp_body_name, cp.local_offset_);
            model.setState(std::make_pair(last_q, last_qd));
            const SVec<double> grav = model.getGravity();
            for (int i = 0; i < 3; i++) {
                const double fd = (p2p[i] - 2 * p0[i] + p2m[i]) / (h2 * h2);
                worst = std::fmax(worst, std::fabs(fd - acc[i]) * 1e-3);  // (second difference: rounding 1e-16 / h^2 = 1e-8, O(h^2) truncation)
                worst = std::fmax(worst, std::fabs(lin[i] + grav[3 + i] - acc[i]));
            }
            std::printf("%s: contact acceleration (%.4f, %.4f, %.4f) = d^2 position / dt^2\n", what, acc[0], acc[1], acc[2]);
        }
    }
    std::printf("%s: nq=%d nv=%d max(|ID(FD(tau)) - tau|, |H ydd + C - tau|) = %.3e\n", what, nq, nv, worst);
    return worst < 5e-8 ? 0 : 1;  // tol of UnitTests/testRigidBodyDynamicsAlgos.cpp:9
}

// A planar parallelogram four-bar through ClusterJoints::FourBar (FourBarJoint.h): cranks of length a on the
// ground pivots (0,0) and (d,0), coupler of length d.  Every q = (t, t, -t) satisfies phi = 0.
static int fourBar()
{
    using namespace ClusterJoints;
    const double a = 0.4, d = 0.7;
    const Mat3<double> I3 = Mat3<double>::Identity();
    const SpatialInertia<double> crank(0.8, Vec3<double>{a / 2, 0., 0.}, I3 * 0.01), coupler(1.1, Vec3<double>{d / 2, 0., 0.}, I3 * 0.02);
    ClusterTreeModel<double> m;
    Body<double> b0 = m.registerBody("crank-1", crank, "ground", spatial::Transform<double>(I3, Vec3<double>::Zero()));
    Body<double> b1 = m.registerBody("crank-2", crank, "ground", spatial::Transform<double>(I3, Vec3<double>{d, 0., 0.}));
    Body<double> b2 = m.registerBody("coupler", coupler, "crank-1", spatial::Transform<double>(I3, Vec3<double>{a, 0., 0.}));
    std::vector<JointPtr<double>> joints;
    for (const char *n : {"j0", "j1", "j2"}) joints.emplace_back(new Joints::Revolute<double>(ori::CoordinateAxis::Z, n));
    auto phi = std::make_shared<LoopConstraint::FourBar<double>>(std::vector<double>{a, d}, std::vector<double>{a},
                                                                  Vec2<double>{d, 0.}, 0);
    m.appendRegisteredBodiesAsCluster<FourBar<double>>("four-bar", std::vector<Body<double>>{b0, b1, b2}, joints, phi);
    if (m.getNumPositions() != 3 || m.getNumDegreesOfFreedom() != 1) return 1;
    double worst = 0;
    for (double t : {0.3, -0.9, 1.4}) {
        DVec<double> q(3), qd(1), tau(1);
        q[0] = t; q[1] = t; q[2] = -t;
        qd[0] = 0.7;
        tau[0] = -0.4;
        m.setState(std::make_pair(q, qd));
        const DVec<double> ydd = m.forwardDynamics(tau);
        worst = std::fmax(worst, (m.inverseDynamics(ydd) - tau).norm());
        // one degree of freedom: H ydd + C = tau with the scalar mass "matrix"
        worst = std::fmax(worst, std::fabs(m.getMassMatrix()(0, 0) * ydd[0] + m.getBiasForceVector()[0] - tau[0]));
    }
    // random joint states of the implicit cluster (GenericJoint.cpp:289-361): spanning positions on the constraint manifold, accepted by
    // setState(ModelState) (an invalid spanning position would throw)
    for (int rep = 0; rep < 3; rep++) {
        const JointState<double> js = m.cluster(0)->joint_->randomJointState();
        if (!js.position.isSpanning() || js.position.size() != 3 || js.velocity.size() != 1) return 1;
        m.setState(ModelState<double>{js});
        DVec<double> tau(1);
        tau[0] = 0.3;
        worst = std::fmax(worst, (m.inverseDynamics(m.forwardDynamics(tau)) - tau).norm());
    }
    std::printf("FourBar (parallelogram): nq=3 nv=1 |ID(FD(tau)) - tau| = %.3e\n", worst);
    return worst < 5e-8 ? 0 : 1;
}

// setState(ModelState) with spanning joint states (testRigidBodyDynamicsAlgos.cpp:45-72: use_spanning_state): every
// cluster hands over q_span = G y, qd_span = G yd flagged as spanning; the dynamics must not change.  An invalid spanning
// velocity throws "Spanning velocity is not valid" (ClusterJoint.cpp:62-65).
static int spanningStates()
{
    ClusterTreeModel<double> m;
    buildRevolutePairChainWithRotor<4>(m);
    const int nv = m.getNumDegreesOfFreedom();
    const DVec<double> tau = DVec<double>::Random(nv);
    ModelState<double> independent, spanning, broken;
    for (const auto &cluster : m.clusters()) {
        const auto &joint = cluster->joint_;
        const DVec<double> y = DVec<double>::Random(joint->numPositions()), yd = DVec<double>::Random(joint->numVelocities());
        const DMat<double> &G = joint->G();
        DVec<double> qs = DVec<double>::Zero(G.rows()), vs = DVec<double>::Zero(G.rows());
        for (int i = 0; i < G.rows(); i++)
            for (int j = 0; j < G.cols(); j++) {
                qs[i] += G(i, j) * y[j];
                vs[i] += G(i, j) * yd[j];
            }
        independent.emplace_back(JointCoordinate<double>(y, false), JointCoordinate<double>(yd, false));
        spanning.emplace_back(JointCoordinate<double>(qs, true), JointCoordinate<double>(vs, true));
        DVec<double> bad = vs;
        bad[0] += 1e-3;
        broken.emplace_back(JointCoordinate<double>(qs, true), JointCoordinate<double>(bad, true));
    }
    m.setState(independent);
    const DVec<double> a = m.forwardDynamics(tau);
    m.setState(spanning);
    const DVec<double> b = m.forwardDynamics(tau);
    const double diff = (a - b).norm();
    bool threw = false;
    try {
        m.setState(broken);
    } catch (const std::runtime_error &e) {
        threw = std::string(e.what()).find("Spanning velocity is not valid") != std::string::npos;
    }
    std::printf("spanning vs independent joint states: |dydd| = %.3e, invalid spanning velocity throws: %d\n", diff, threw ? 1 : 0);
    return diff < 1e-9 && threw ? 0 : 1;
}

// The same through a model of the reference's parallel-chain benchmark family (Benchmarking/urdfs/parallel_chains; here one cluster of 12
// bodies / 11 independent coordinates): the class API, setState(ModelState) with spanning joint states -- the call the reference's
// benchmark makes (pinocchioBenchmark.cpp:160-168) --, forward dynamics, ID(FD(tau)) = tau.  Such clusters run through the spanning tree
// (DESIGN 7c); the facade must not notice.
static int bigClusterStates(const std::string &urdf)
{
    // the explicit parallel chain of depth 6 and loop size 12, hand-built as the reference's URDF+ parser would cluster it: twelve bodies in
    // ONE Generic cluster (joint_2_6 coupled to joint_1_6 with ratio 1: eleven independent coordinates), through the class API; `urdf`
    // (the reference's own depth-10 file) is only read to check that the URDF route gives a model of the expected size
    using namespace ClusterJoints;
    {
        // the URDF route: clusters() of the model as read -- sizes, G of the 16-body cluster, random joint states as the reference's
        // benchmark draws them, spanning against independent joint states
        ClusterTreeModel<double> u;
        u.buildModelFromURDF(urdf);
        if (u.getNumBodies() != 20 || u.getNumDegreesOfFreedom() != 19 || u.clusters().size() != 5 || u.bodies().size() != 20) return 1;
        int np = 0, nvv = 0, big = 0;
        ModelState<double> ind, span;
        for (const auto &cluster : u.clusters()) {
            const auto &joint = cluster->joint_;
            np += joint->numPositions();
            nvv += joint->numVelocities();
            big = std::max(big, static_cast<int>(cluster->bodies_.size()));
            JointState<double> js = joint->randomJointState();
            for (int j = 0; j < static_cast<int>(js.position.size()); j++) js.position[j] *= 0.5;
            const DMat<double> &G = joint->G();
            if (G.rows() != static_cast<int>(cluster->bodies_.size()) || G.cols() != joint->numVelocities()) return 1;
            DVec<double> qs = DVec<double>::Zero(G.rows()), vs = DVec<double>::Zero(G.rows());
            for (int i = 0; i < G.rows(); i++)
                for (int j = 0; j < G.cols(); j++) {
                    qs[i] += G(i, j) * js.position[j];
                    vs[i] += G(i, j) * js.velocity[j];
                }
            ind.push_back(js);
            span.emplace_back(JointCoordinate<double>(qs, true), JointCoordinate<double>(vs, true));
        }
        if (np != u.getNumPositions() || nvv != 19 || big != 16 || u.body("link_2_8").parent_index_ != u.body("link_2_7").index_) return 1;
        const DVec<double> tau_u = DVec<double>::Random(19);
        u.setState(ind);
        const DVec<double> a_u = u.forwardDynamics(tau_u);
        u.setState(span);
        const DVec<double> b_u = u.forwardDynamics(tau_u);
        std::printf("URDF-built parallel chain: clusters() %zu, biggest %d bodies, spanning vs independent |dydd| = %.3e\n", u.clusters().size(), big,
                    (a_u - b_u).norm());
        if (!((a_u - b_u).norm() < 1e-9)) return 1;
    }
    const int depth = 6, k = 2 * depth, n = k - 1;
    const Mat3<double> I3 = Mat3<double>::Identity();
    Mat3<double> Ic = Mat3<double>::Zero();
    Ic(0, 0) = 0.01; Ic(1, 1) = 0.1; Ic(2, 2) = 0.01;
    const SpatialInertia<double> link(0.25, Vec3<double>{0., 0.5, 0.}, Ic);
    ClusterTreeModel<double> m;
    std::vector<Body<double>> bodies;
    std::vector<JointPtr<double>> joints;
    for (int i = 1; i <= depth; i++)
        for (int chain = 1; chain <= 2; chain++) {
            const std::string name = "link_" + std::to_string(chain) + "_" + std::to_string(i);
            const std::string parent = i == 1 ? "ground" : "link_" + std::to_string(chain) + "_" + std::to_string(i - 1);
            const Vec3<double> r = i == 1 ? Vec3<double>::Zero() : Vec3<double>{0., 1., 0.};
            bodies.push_back(m.registerBody(name, link, parent, spatial::Transform<double>(I3, r)));
            joints.emplace_back(new Joints::Revolute<double>(ori::CoordinateAxis::Z, "joint_" + std::to_string(chain) + "_" + std::to_string(i)));
        }
    // independent: every joint but the last one of chain 2 (body index k - 1), which follows the last one of chain 1 (k - 2)
    DMat<double> G = DMat<double>::Zero(k, n), K = DMat<double>::Zero(1, k);
    for (int i = 0; i < n; i++) G(i, i) = 1.0;
    G(k - 1, k - 2) = 1.0;
    K(0, k - 2) = 1.0;
    K(0, k - 1) = -1.0;
    m.appendRegisteredBodiesAsCluster<Generic<double>>("loop", bodies, joints, std::make_shared<LoopConstraint::Static<double>>(G, K));
    const int nv = m.getNumDegreesOfFreedom();
    int biggest = 0;
    for (const auto &cluster : m.clusters()) biggest = std::max(biggest, static_cast<int>(cluster->bodies_.size()));
    const DVec<double> tau = DVec<double>::Random(nv);
    ModelState<double> independent, spanning;
    for (const auto &cluster : m.clusters()) {
        const auto &joint = cluster->joint_;
        DVec<double> y = DVec<double>::Random(joint->numPositions());
        const DVec<double> yd = DVec<double>::Random(joint->numVelocities());
        for (int j = 0; j < static_cast<int>(y.size()); j++) y[j] *= 0.5;
        const DMat<double> &G = joint->G();
        DVec<double> qs = DVec<double>::Zero(G.rows()), vs = DVec<double>::Zero(G.rows());
        for (int i = 0; i < G.rows(); i++)
            for (int j = 0; j < G.cols(); j++) {
                qs[i] += G(i, j) * y[j];
                vs[i] += G(i, j) * yd[j];
            }
        independent.emplace_back(JointCoordinate<double>(y, false), JointCoordinate<double>(yd, false));
        spanning.emplace_back(JointCoordinate<double>(qs, true), JointCoordinate<double>(vs, true));
    }
    m.setState(independent);
    const DVec<double> a = m.forwardDynamics(tau);
    const DVec<double> back = m.inverseDynamics(a);
    m.setState(spanning);
    const DVec<double> b = m.forwardDynamics(tau);
    const double diff = (a - b).norm(), rt = (back - tau).norm();
    std::printf("big cluster (%d bodies, model nv %d): spanning vs independent |dydd| = %.3e, |ID(FD(tau)) - tau| = %.3e\n", biggest, nv, diff, rt);
    return biggest > 8 && diff < 1e-9 && rt < 1e-8 ? 0 : 1;
}

int main(int argc, char **argv)
{
    const std::string mode = argc > 1 ? argv[1] : "";
    try {
        if (mode == "--big" && argc > 2) {
            int rc = bigClusterStates(argv[2]);
            if (argc > 3) {
                // the implicit family: random joint states of every cluster as the reference's benchmark draws them (the 17-body loop
                // cluster: spanning positions on the constraint manifold through the library's Newton projection), setState, dynamics
                ClusterTreeModel<double> u;
                u.buildModelFromURDF(argv[3]);
                double worst = 0;
                for (int rep = 0; rep < 3; rep++) {
                    ModelState<double> state;
                    int spanning = 0;
                    for (const auto &cluster : u.clusters()) {
                        state.push_back(cluster->joint_->randomJointState());
                        if (state.back().position.isSpanning()) {
                            spanning++;
                            for (int j = 0; j < static_cast<int>(state.back().position.size()); j++)
                                if (!std::isfinite(state.back().position[j])) rc |= 1;
                        }
                    }
                    if (spanning != 1) rc |= 1;
                    u.setState(state);
                    const DVec<double> tau = DVec<double>::Random(u.getNumDegreesOfFreedom());
                    const DVec<double> ydd = u.forwardDynamics(tau);
                    worst = std::fmax(worst, (u.inverseDynamics(ydd) - tau).norm() / (1.0 + ydd.norm()));
                }
                std::printf("URDF-built implicit parallel chain: random model states, |ID(FD(tau)) - tau| = %.3e\n", worst);
                if (!(worst < 1e-8)) rc |= 1;
            }
            std::printf(rc ? "FAILED\n" : "OK\n");
            return rc;
        }
        if (mode == "--dump" && argc > 2) {
            const std::string dir = argv[2];
            { ClusterTreeModel<double> m; buildRevoluteChainWithRotor<2>(m); dump(dir + "/rev2.grbd", m.serialize()); }
            { ClusterTreeModel<double> m; buildRevoluteChainWithRotor<4>(m); dump(dir + "/rev4.grbd", m.serialize()); }
            { ClusterTreeModel<double> m; buildRevolutePairChainWithRotor<2>(m); dump(dir + "/pair2.grbd", m.serialize()); }
            { ClusterTreeModel<double> m; buildRevolutePairChainWithRotor<4>(m); dump(dir + "/pair4.grbd", m.serialize()); }
            // reference error behaviour: a free joint cannot have a parent body (FreeJoint.cpp:14-15)
            bool threw = false;
            try {
                ClusterTreeModel<double> m;
                buildRevoluteChainWithRotor<2>(m);
                m.appendBody<ClusterJoints::Free<double>>("floating", SpatialInertia<double>(), "link-1", spatial::Transform<double>());
            } catch (const std::runtime_error &) { threw = true; }
            std::printf("free-joint-with-parent throws: %d\n", threw ? 1 : 0);
            return threw ? 0 : 1;
        }
        if (mode == "--run" && argc > 2) {
            int rc = 0;
            {
                ClusterTreeModel<double> m;
                buildRevoluteChainWithRotor<4>(m);
                m.appendEndEffector("link-3", Vec3<double>{1.0, 0., 0.}, "tip");             // ClusterTreeModel.cpp:216-221
                m.appendContactPoint("link-1", Vec3<double>{0.5, 0.1, 0.}, "mid-contact");    // :165-187
                rc |= roundtrip(m, "RevoluteChainWithRotor<4>", "link-3");
                // cluster bookkeeping queries (ClusterTreeModel.h:97-121)
                if (m.getIndexOfClusterContainingBody("link-3") != 3 || m.getNumBodiesInCluster(3) != 2 ||
                    m.getSubIndexWithinClusterForBody("link-3") != 0 || m.getClusterContainingBody("link-3")->bodies_.size() != 2 ||
                    m.getBody(m.body("link-3").index_).name_ != "link-3" || m.getClusterAncestorIndexFromParent(m.body("link-3").index_) != m.body("link-3").index_ ||
                    m.stateVectorToModelState(std::make_pair(DVec<double>::Zero(4), DVec<double>::Zero(4))).size() != 4)
                    rc |= 1;
                // appendContactBox (ClusterTreeModel.cpp:200-213): eight corners, the reference's names and order
                const size_t before = m.contactPoints().size();
                m.appendContactBox("link-2", Vec3<double>{0.2, 0.4, 0.6});
                if (m.contactPoints().size() != before + 8 || m.contactPoint("torso-contact-2").local_offset_[0] != -0.1 ||
                    m.contactPoint("torso-contact-3").local_offset_[1] != -0.2 || m.contactPoint("torso-contact-5").local_offset_[2] != -0.3 ||
                    m.contactPoint(static_cast<int>(before)).name_ != "torso-contact-1")
                    rc |= 1;
            }
            { ClusterTreeModel<double> m; buildRevolutePairChainWithRotor<4>(m); rc |= roundtrip(m, "RevolutePairChainWithRotor<4>"); }
            {
                ClusterTreeModel<double> m(argv[2]);
                if (argc > 3) m.appendEndEffector(argv[3], Vec3<double>{0., 0., -0.1}, "urdf-ee");  // a link name of the URDF
                rc |= roundtrip(m, argv[2]);
            }
            rc |= fourBar();
            rc |= spanningStates();
            std::printf(rc ? "FAILED\n" : "OK\n");
            return rc;
        }
#ifdef FACADE_TEST_WITH_HIP
        if (mode == "--device" && argc > 2) {
            const int rc = deviceArrays(argv[2]);
            std::printf(rc ? "FAILED\n" : "OK\n");
            return rc;
        }
#endif
    } catch (const std::exception &e) {
        std::printf("exception: %s\n", e.what());
        return 2;
    }
    std::printf("usage: facade_test --dump <dir> | --run <urdf> | --device <urdf>\n");
    return 2;
}
