// grbda.h -- C++17 facade with the reference's class names over the MI355X engine.
//
// A user of ROAM-Lab-ND/generalized_rbda builds a model with
//   model.registerBody(...); model.appendRegisteredBodiesAsCluster<ClusterJoints::RevoluteWithRotor<>>(...)
//   model.setState(...); model.forwardDynamics(tau);
// (include/grbda/Dynamics/ClusterTreeModel.h:24-165, TreeModel.h:15-143).  This header keeps those
// names, argument meanings and error behaviour (std::runtime_error with the reference's messages) for
// the forward / inverse dynamics path; the model is serialised to a model description
// (ModelDescription.h) and every dynamics call goes through the C ABI (include/grbda_hip.h) to the
// HIP kernels -- there is no CPU implementation behind this class.
//
// Eigen is not available on the target image, so the few value types the API needs
// (Vec3 / Mat3 / Mat6 / SVec / DVec / DMat) are minimal stand-ins with Eigen-like accessors.
//
// Not carried over (out of the hot path): contact points, Jacobians, mass matrix,
// EFPA, and LoopConstraint::GenericImplicit built from a CasADi lambda -- use
// LoopConstraint::LoopPosition / LoopConstraint::TrigPolynomial, the data-driven equivalents.
#pragma once

#include <array>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../../include/grbda_hip.h"
#include "ModelDescription.h"

namespace grbda {

// ------------------------------------------------------------------------------------------------
// value types (include/grbda/Utils/cppTypes.h:17-83)
// ------------------------------------------------------------------------------------------------
template <typename T, int N>
struct Vec : std::array<T, N> {
    Vec() { this->fill(T(0)); }
    Vec(std::initializer_list<T> l)
    {
        this->fill(T(0));
        int i = 0;
        for (const T &x : l)
            if (i < N) (*this)[i++] = x;
    }
    T &operator()(int i) { return (*this)[i]; }
    const T &operator()(int i) const { return (*this)[i]; }
    static Vec Zero() { return Vec(); }
    static Vec Random()
    {
        Vec v;
        for (auto &x : v) x = T(2.0 * std::rand() / RAND_MAX - 1.0);
        return v;
    }
};
template <typename T> using Vec1 = Vec<T, 1>;
template <typename T> using Vec2 = Vec<T, 2>;
template <typename T> using Vec3 = Vec<T, 3>;
template <typename T> using Quat = Vec<T, 4>;
template <typename T> using SVec = Vec<T, 6>;

template <typename T, int R, int C>
struct Mat {
    std::array<T, R * C> d{};  // row-major
    T &operator()(int i, int j) { return d[i * C + j]; }
    const T &operator()(int i, int j) const { return d[i * C + j]; }
    static Mat Zero() { return Mat(); }
    static Mat Identity()
    {
        Mat m;
        for (int i = 0; i < (R < C ? R : C); i++) m(i, i) = T(1);
        return m;
    }
    Mat<T, C, R> transpose() const
    {
        Mat<T, C, R> t;
        for (int i = 0; i < R; i++)
            for (int j = 0; j < C; j++) t(j, i) = (*this)(i, j);
        return t;
    }
    template <int K>
    Mat<T, R, K> operator*(const Mat<T, C, K> &o) const
    {
        Mat<T, R, K> m;
        for (int i = 0; i < R; i++)
            for (int j = 0; j < K; j++) {
                T s = 0;
                for (int k = 0; k < C; k++) s += (*this)(i, k) * o(k, j);
                m(i, j) = s;
            }
        return m;
    }
    Mat operator+(const Mat &o) const { Mat m; for (size_t i = 0; i < d.size(); i++) m.d[i] = d[i] + o.d[i]; return m; }
    Mat operator*(T s) const { Mat m; for (size_t i = 0; i < d.size(); i++) m.d[i] = d[i] * s; return m; }
};
template <typename T> using Mat3 = Mat<T, 3, 3>;
template <typename T> using Mat6 = Mat<T, 6, 6>;
template <typename T> using RotMat = Mat3<T>;

template <typename T>
struct DVec : std::vector<T> {
    using std::vector<T>::vector;
    DVec() = default;
    DVec(const std::vector<T> &v) : std::vector<T>(v) {}
    T &operator()(int i) { return (*this)[i]; }
    const T &operator()(int i) const { return (*this)[i]; }
    int rows() const { return static_cast<int>(this->size()); }
    DVec segment(int i, int n) const { return DVec(this->begin() + i, this->begin() + i + n); }
    static DVec Zero(int n) { return DVec(static_cast<size_t>(n), T(0)); }
    static DVec Random(int n)
    {
        DVec v(static_cast<size_t>(n));
        for (auto &x : v) x = T(2.0 * std::rand() / RAND_MAX - 1.0);
        return v;
    }
    DVec operator-(const DVec &o) const { DVec r(*this); for (size_t i = 0; i < r.size(); i++) r[i] -= o[i]; return r; }
    DVec operator+(const DVec &o) const { DVec r(*this); for (size_t i = 0; i < r.size(); i++) r[i] += o[i]; return r; }
    T norm() const { T s = 0; for (const T &x : *this) s += x * x; return std::sqrt(s); }
};

template <typename T>
struct DMat {
    int r = 0, c = 0;
    std::vector<T> d;  // row-major
    DMat() = default;
    DMat(int rows, int cols) : r(rows), c(cols), d(static_cast<size_t>(rows) * cols, T(0)) {}
    int rows() const { return r; }
    int cols() const { return c; }
    T &operator()(int i, int j) { return d[static_cast<size_t>(i) * c + j]; }
    const T &operator()(int i, int j) const { return d[static_cast<size_t>(i) * c + j]; }
    static DMat Zero(int rows, int cols) { return DMat(rows, cols); }
    static DMat Identity(int rows, int cols)
    {
        DMat m(rows, cols);
        for (int i = 0; i < (rows < cols ? rows : cols); i++) m(i, i) = T(1);
        return m;
    }
    DMat operator*(const DMat &o) const
    {
        DMat m(r, o.c);
        for (int i = 0; i < r; i++)
            for (int j = 0; j < o.c; j++) {
                T s = 0;
                for (int k = 0; k < c; k++) s += (*this)(i, k) * o(k, j);
                m(i, j) = s;
            }
        return m;
    }
    T maxAbs() const { T s = 0; for (const T &x : d) s = std::fabs(x) > s ? std::fabs(x) : s; return s; }
};

template <typename T> using D6Mat = DMat<T>;  // 6 x n (cppTypes.h)

// ------------------------------------------------------------------------------------------------
// ori:: (include/grbda/Utils/OrientationTools.h)
// ------------------------------------------------------------------------------------------------
namespace ori {
enum class CoordinateAxis { X, Y, Z };

template <typename T>
Mat3<T> vectorToSkewMat(const Vec3<T> &v)
{
    Mat3<T> m;
    m(0, 1) = -v[2]; m(0, 2) = v[1]; m(1, 0) = v[2]; m(1, 2) = -v[0]; m(2, 0) = -v[1]; m(2, 1) = v[0];
    return m;
}
// OrientationTools.h:46-68
template <typename T>
Mat3<T> coordinateRotation(CoordinateAxis axis, T theta)
{
    const T s = std::sin(theta), c = std::cos(theta);
    Mat3<T> R = Mat3<T>::Identity();
    if (axis == CoordinateAxis::X) { R(1, 1) = c; R(1, 2) = s; R(2, 1) = -s; R(2, 2) = c; }
    else if (axis == CoordinateAxis::Y) { R(0, 0) = c; R(0, 2) = -s; R(2, 0) = s; R(2, 2) = c; }
    else { R(0, 0) = c; R(0, 1) = s; R(1, 0) = -s; R(1, 1) = c; }
    return R;
}
// OrientationTools.h:121-130
template <typename T>
Mat3<T> rpyToRotMat(const Vec3<T> &v)
{
    return coordinateRotation(CoordinateAxis::X, v[0]) * coordinateRotation(CoordinateAxis::Y, v[1]) *
           coordinateRotation(CoordinateAxis::Z, v[2]);
}
// OrientationTools.h:251-269 (scalar first, transposed)
template <typename T>
Mat3<T> quaternionToRotationMatrix(const Quat<T> &q)
{
    const T e0 = q[0], e1 = q[1], e2 = q[2], e3 = q[3];
    Mat3<T> R;
    R(0, 0) = 1 - 2 * (e2 * e2 + e3 * e3); R(0, 1) = 2 * (e1 * e2 - e0 * e3);     R(0, 2) = 2 * (e1 * e3 + e0 * e2);
    R(1, 0) = 2 * (e1 * e2 + e0 * e3);     R(1, 1) = 1 - 2 * (e1 * e1 + e3 * e3); R(1, 2) = 2 * (e2 * e3 - e0 * e1);
    R(2, 0) = 2 * (e1 * e3 - e0 * e2);     R(2, 1) = 2 * (e2 * e3 + e0 * e1);     R(2, 2) = 1 - 2 * (e1 * e1 + e2 * e2);
    return R.transpose();
}
// OrientationTools.h:160-200
template <typename T>
Quat<T> rotationMatrixToQuaternion(const Mat3<T> &r1)
{
    Quat<T> q;
    const Mat3<T> r = r1.transpose();
    const T tr = r(0, 0) + r(1, 1) + r(2, 2);
    if (tr > 0.0) {
        const T S = std::sqrt(tr + 1.0) * 2.0;
        q[0] = 0.25 * S; q[1] = (r(2, 1) - r(1, 2)) / S; q[2] = (r(0, 2) - r(2, 0)) / S; q[3] = (r(1, 0) - r(0, 1)) / S;
    } else if (r(0, 0) > r(1, 1) && r(0, 0) > r(2, 2)) {
        const T S = std::sqrt(1.0 + r(0, 0) - r(1, 1) - r(2, 2)) * 2.0;
        q[0] = (r(2, 1) - r(1, 2)) / S; q[1] = 0.25 * S; q[2] = (r(0, 1) + r(1, 0)) / S; q[3] = (r(0, 2) + r(2, 0)) / S;
    } else if (r(1, 1) > r(2, 2)) {
        const T S = std::sqrt(1.0 + r(1, 1) - r(0, 0) - r(2, 2)) * 2.0;
        q[0] = (r(0, 2) - r(2, 0)) / S; q[1] = (r(0, 1) + r(1, 0)) / S; q[2] = 0.25 * S; q[3] = (r(1, 2) + r(2, 1)) / S;
    } else {
        const T S = std::sqrt(1.0 + r(2, 2) - r(0, 0) - r(1, 1)) * 2.0;
        q[0] = (r(1, 0) - r(0, 1)) / S; q[1] = (r(0, 2) + r(2, 0)) / S; q[2] = (r(1, 2) + r(2, 1)) / S; q[3] = 0.25 * S;
    }
    return q;
}
template <typename T>
Quat<T> rpyToQuat(const Vec3<T> &rpy) { return rotationMatrixToQuaternion(rpyToRotMat(rpy)); }
}  // namespace ori

// OrientationRepresentation.h:11-49
namespace ori_representation {
struct Quaternion {
    static const int num_ori_parameter = 4;
    static constexpr int desc_id = GRBDA_ORI_QUATERNION;
};
struct RollPitchYaw {
    static const int num_ori_parameter = 3;
    static constexpr int desc_id = GRBDA_ORI_RPY;
};
}  // namespace ori_representation

// ------------------------------------------------------------------------------------------------
// spatial::Transform (include/grbda/Utils/SpatialTransforms.h:13-60), SpatialInertia (SpatialInertia.h:74-116)
// ------------------------------------------------------------------------------------------------
namespace spatial {
template <typename Scalar = double>
class Transform {
public:
    Transform(const Mat3<Scalar> &E = Mat3<Scalar>::Identity(), const Vec3<Scalar> &r = Vec3<Scalar>::Zero()) : E_(E), r_(r) {}
    const Mat3<Scalar> &getRotation() const { return E_; }
    const Vec3<Scalar> &getTranslation() const { return r_; }
    // SpatialTransforms.cpp:149-157: E = E1 E2, r = r2 + E2^T r1
    Transform operator*(const Transform &X_in) const
    {
        const Mat3<Scalar> E = E_ * X_in.E_;
        Vec3<Scalar> r = X_in.r_;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) r[i] += X_in.E_(j, i) * r_[j];
        return Transform(E, r);
    }
private:
    Mat3<Scalar> E_;
    Vec3<Scalar> r_;
};
}  // namespace spatial

template <typename T>
class SpatialInertia {
public:
    SpatialInertia() = default;
    SpatialInertia(T mass, const Vec3<T> &com, const Mat3<T> &inertia)
    {
        const Mat3<T> c = ori::vectorToSkewMat(com);
        const Mat3<T> cct = c * c.transpose();
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                I_(i, j) = inertia(i, j) + mass * cct(i, j);
                I_(i, j + 3) = mass * c(i, j);
                I_(i + 3, j) = mass * c(j, i);
                I_(i + 3, j + 3) = i == j ? mass : T(0);
            }
    }
    explicit SpatialInertia(const Mat6<T> &inertia) : I_(inertia) {}
    const Mat6<T> &getMatrix() const { return I_; }
private:
    Mat6<T> I_;
};

// Body (include/grbda/Dynamics/Body.h:16-43)
template <typename Scalar = double>
struct Body {
    int index_ = -1;
    std::string name_;
    int parent_index_ = -1;
    spatial::Transform<Scalar> Xtree_;
    SpatialInertia<Scalar> inertia_;
    int sub_index_within_cluster_ = 0;
    int cluster_ancestor_index_ = -1;
    int cluster_ancestor_sub_index_within_cluster_ = 0;
};

// StateRepresentation.h:9-80
template <typename Scalar = double>
class JointCoordinate : public DVec<Scalar> {
public:
    JointCoordinate() = default;
    JointCoordinate(const DVec<Scalar> &vec, bool is_spanning) : DVec<Scalar>(vec), is_spanning_(is_spanning) {}
    bool isSpanning() const { return is_spanning_; }
private:
    bool is_spanning_ = false;
};
template <typename Scalar = double>
struct JointState {
    JointState() = default;
    JointState(const JointCoordinate<Scalar> &pos, const JointCoordinate<Scalar> &vel) : position(pos), velocity(vel) {}
    JointCoordinate<Scalar> position, velocity;
};
template <typename Scalar = double> using ModelState = std::vector<JointState<Scalar>>;
template <typename Scalar = double>
struct ExternalForceAndBodyIndexPair {
    ExternalForceAndBodyIndexPair(int index, const SVec<Scalar> &force) : index_(index), force_(force) {}
    int index_;
    SVec<Scalar> force_;
};

// include/grbda/Dynamics/ContactPoint.h: a point fixed in a body (an end effector when flagged)
template <typename Scalar = double>
struct ContactPoint {
    ContactPoint(int body_index, const Vec3<Scalar> &local_offset, const std::string &name, bool is_end_effector = false)
        : body_index_(body_index), local_offset_(local_offset), name_(name), is_end_effector_(is_end_effector) {}
    int body_index_;
    Vec3<Scalar> local_offset_;
    std::string name_;
    bool is_end_effector_;
    int end_effector_index_ = -1;
    Vec3<Scalar> position_;  // world position after forwardKinematicsIncludingContactPoints()
    Vec3<Scalar> velocity_;  // world-axes linear velocity of the point, the same call
    Vec3<Scalar> acceleration_;  // classical acceleration of the point in world axes (forwardAccelerationKinematicsIncludingContactPoints)
    DMat<Scalar> jacobian_;  // 6 x nv, world axes at the point [angular; linear] (contactJacobianWorldFrame)
};

// ------------------------------------------------------------------------------------------------
// loop constraints (include/grbda/Dynamics/ClusterJoints/LoopConstraint.h:14-59)
// ------------------------------------------------------------------------------------------------
namespace LoopConstraint {
template <typename Scalar = double>
struct Base {
    virtual ~Base() {}
    int kind = GRBDA_CONSTRAINT_STATIC;
    DMat<Scalar> G_, K_;
    std::vector<int32_t> ints;
    std::vector<double> dbls;
    std::vector<bool> independent;
    int rows = 0;
    const DMat<Scalar> &G() const { return G_; }
    const DMat<Scalar> &K() const { return K_; }
    bool isExplicit() const { return kind == GRBDA_CONSTRAINT_STATIC || kind == GRBDA_CONSTRAINT_FREE; }
};
// LoopConstraint.cpp:38-52
template <typename Scalar = double>
struct Static : Base<Scalar> {
    Static(const DMat<Scalar> &G, const DMat<Scalar> &K)
    {
        this->kind = GRBDA_CONSTRAINT_STATIC;
        this->G_ = G;
        this->K_ = K;
        this->rows = K.rows();
    }
};
template <typename Scalar = double>
struct Free : Base<Scalar> {
    Free() { this->kind = GRBDA_CONSTRAINT_FREE; }
};
// data-driven replacement of GenericImplicit for URDF+ <loop> constraints
// (ClusterTreeParsing.cpp:310-376); see include/grbda_model_desc.h for the payload
template <typename Scalar = double>
struct LoopPosition : Base<Scalar> {
    struct Loop {
        std::vector<int> nca_to_predecessor, nca_to_successor;  // sub-indices within the cluster
        spatial::Transform<Scalar> predecessor_origin, successor_origin;
        int axis_mask = 7;
    };
    LoopPosition(const std::vector<bool> &is_coordinate_independent, const std::vector<Loop> &loops)
    {
        this->kind = GRBDA_CONSTRAINT_LOOP_POSITION;
        this->independent = is_coordinate_independent;
        this->ints.push_back(static_cast<int32_t>(loops.size()));
        for (bool b : is_coordinate_independent) this->ints.push_back(b ? 1 : 0);
        for (const Loop &l : loops) {
            this->ints.push_back(static_cast<int32_t>(l.nca_to_predecessor.size()));
            for (int s : l.nca_to_predecessor) this->ints.push_back(s);
            this->ints.push_back(static_cast<int32_t>(l.nca_to_successor.size()));
            for (int s : l.nca_to_successor) this->ints.push_back(s);
            this->ints.push_back(l.axis_mask);
            for (const auto *X : {&l.predecessor_origin, &l.successor_origin}) {
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++) this->dbls.push_back(static_cast<double>(X->getRotation()(i, j)));
                for (int i = 0; i < 3; i++) this->dbls.push_back(static_cast<double>(X->getTranslation()[i]));
            }
            for (int a = 0; a < 3; a++) this->rows += (l.axis_mask >> a) & 1;
        }
    }
};
// data-driven replacement of GenericImplicit for hand-written trig-polynomial phi lambdas
// (src/Robots/Tello.cpp:139-163,237-261)
template <typename Scalar = double>
struct TrigPolynomial : Base<Scalar> {
    enum Fn { Linear = 0, Sin = 1, Cos = 2 };
    struct Factor { Fn fn; std::vector<double> w; double b; };
    struct Term { double coef; std::vector<Factor> factors; };
    TrigPolynomial(const std::vector<bool> &is_coordinate_independent, const std::vector<std::vector<Term>> &phi_rows)
    {
        this->kind = GRBDA_CONSTRAINT_TRIG_POLY;
        this->independent = is_coordinate_independent;
        for (bool b : is_coordinate_independent) this->ints.push_back(b ? 1 : 0);
        for (const auto &row : phi_rows) {
            this->ints.push_back(static_cast<int32_t>(row.size()));
            for (const Term &t : row) {
                this->ints.push_back(static_cast<int32_t>(t.factors.size()));
                this->dbls.push_back(t.coef);
                for (const Factor &f : t.factors) {
                    this->ints.push_back(static_cast<int32_t>(f.fn));
                    for (double w : f.w) this->dbls.push_back(w);
                    this->dbls.push_back(f.b);
                }
            }
        }
        this->rows = static_cast<int>(phi_rows.size());
    }
};
// LoopConstraint::FourBar (include/grbda/Dynamics/ClusterJoints/FourBarJoint.h:12-54, FourBarJoint.cpp:7-79): the
// planar loop closure  phi = sum_path1 l_i [cos, sin](cumulative angle) - offset - sum_path2 l_i [cos, sin](...)
// over spanning coordinates (q0, q2) on path 1 and (q1) on path 2 -- a trigonometric polynomial, so it is
// handed to the engine as one (the reference differentiates the same expression by hand, :80-199).
template <typename Scalar = double>
struct FourBar : TrigPolynomial<Scalar> {
    using typename TrigPolynomial<Scalar>::Term;
    using typename TrigPolynomial<Scalar>::Factor;
    using Fn = typename TrigPolynomial<Scalar>::Fn;
    FourBar(std::vector<Scalar> path1_link_lengths, std::vector<Scalar> path2_link_lengths, Vec2<Scalar> offset,
            int independent_coordinate)
        : TrigPolynomial<Scalar>(independent_flags(path1_link_lengths, path2_link_lengths, independent_coordinate),
                                 rows_of(path1_link_lengths, path2_link_lengths, offset)),
          independent_coordinate_(independent_coordinate) {}
    const int &independent_coordinate() const { return independent_coordinate_; }

private:
    static std::vector<bool> independent_flags(const std::vector<Scalar> &p1, const std::vector<Scalar> &p2, int ind)
    {
        if (p1.size() + p2.size() != 3) throw std::runtime_error("FourBar: Must contain 3 links");
        if (p1.size() != 2) throw std::runtime_error("FourBar: path 1 carries spanning coordinates 0 and 2, path 2 coordinate 1");
        if (ind < 0 || ind > 2) throw std::runtime_error("FourBar: Invalid independent coordinate");
        std::vector<bool> f(3, false);
        f[ind] = true;
        return f;
    }
    static std::vector<std::vector<Term>> rows_of(const std::vector<Scalar> &p1, const std::vector<Scalar> &p2,
                                                  const Vec2<Scalar> &offset)
    {
        std::vector<std::vector<Term>> rows(2);
        for (int r = 0; r < 2; r++) {
            const Fn fn = r == 0 ? TrigPolynomial<Scalar>::Cos : TrigPolynomial<Scalar>::Sin;
            const int path1_coords[2] = {0, 2};
            std::vector<double> w(3, 0.0);
            for (size_t i = 0; i < p1.size(); i++) {  // cumulative angle along path 1
                w[path1_coords[i]] = 1.0;
                rows[r].push_back(Term{static_cast<double>(p1[i]), {Factor{fn, w, 0.0}}});
            }
            rows[r].push_back(Term{-static_cast<double>(offset[r]), {}});
            std::vector<double> w2(3, 0.0);
            w2[1] = 1.0;
            rows[r].push_back(Term{-static_cast<double>(p2[0]), {Factor{fn, w2, 0.0}}});
        }
        return rows;
    }
    int independent_coordinate_;
};
}  // namespace LoopConstraint

// single joints (include/grbda/Dynamics/Joints/Joint.h:43-102)
namespace Joints {
template <typename Scalar = double>
struct Base {
    virtual ~Base() {}
    int type = GRBDA_JOINT_REVOLUTE;
    ori::CoordinateAxis axis = ori::CoordinateAxis::Z;
    std::string name;
};
template <typename Scalar = double>
struct Revolute : Base<Scalar> {
    Revolute(ori::CoordinateAxis a, std::string n = "unnamed_revolute_joint") { this->type = GRBDA_JOINT_REVOLUTE; this->axis = a; this->name = n; }
};
template <typename Scalar = double, typename Ori = ori_representation::Quaternion>
struct Free : Base<Scalar> {
    Free(std::string n = "unnamed_free_joint") { this->type = GRBDA_JOINT_FREE; this->name = n; }
};
}  // namespace Joints
template <typename Scalar> using JointPtr = std::shared_ptr<Joints::Base<Scalar>>;

// ------------------------------------------------------------------------------------------------
// cluster joints (include/grbda/Dynamics/ClusterJoints/*.h, Transmissions.h:11-43)
// ------------------------------------------------------------------------------------------------
namespace ClusterJoints {
template <typename Scalar = double>
struct GearedTransmissionModule {
    Body<Scalar> body_, rotor_;
    std::string body_joint_name_, rotor_joint_name_;
    ori::CoordinateAxis joint_axis_, rotor_axis_;
    Scalar gear_ratio_;
};
template <size_t N_belts, typename Scalar = double>
struct ParallelBeltTransmissionModule {
    Body<Scalar> body_, rotor_;
    ori::CoordinateAxis joint_axis_, rotor_axis_;
    Scalar gear_ratio_;
    Vec<Scalar, static_cast<int>(N_belts)> belt_ratios_;
};
// Transmissions.h:34-43
template <typename Scalar, int N>
Vec<Scalar, N> beltMatrixRowFromBeltRatios(Vec<Scalar, N> ratios)
{
    for (int i = 1; i < N; i++) ratios[i] = ratios[i - 1] * ratios[i];
    return ratios;
}

template <typename Scalar = double>
class Base {
public:
    virtual ~Base() {}
    int numBodies() const { return num_bodies_; }
    int numPositions() const { return num_positions_; }
    int numVelocities() const { return num_velocities_; }
    const DMat<Scalar> &G() const { return loop_constraint_->G(); }
    const DMat<Scalar> &K() const { return loop_constraint_->K(); }
    std::shared_ptr<LoopConstraint::Base<Scalar>> cloneLoopConstraint() const { return loop_constraint_; }
    const std::vector<JointPtr<Scalar>> &singleJoints() const { return single_joints_; }
    // ClusterJoint.cpp:73-80; implicit clusters (GenericJoint.cpp:289-361): spanning positions on the constraint manifold -- independent
    // coordinates U(-1, 1), a dependent guess U(-0.1, 0.1), Newton, up to 45 draws; the model the cluster belongs to installs the root finder
    std::function<JointCoordinate<double>()> find_roots_for_phi_;
    virtual JointState<double> randomJointState() const
    {
        if (find_roots_for_phi_)
            return JointState<double>(find_roots_for_phi_(), JointCoordinate<double>(DVec<double>::Random(num_velocities_), false));
        return JointState<double>(JointCoordinate<double>(DVec<double>::Random(num_positions_), false),
                                  JointCoordinate<double>(DVec<double>::Random(num_velocities_), false));
    }
    // single joints in sub-index order (body i of the cluster <-> ordered_joints()[i])
    std::vector<JointPtr<Scalar>> ordered_joints_;
protected:
    Base(int nb, int np, int nv) : num_bodies_(nb), num_positions_(np), num_velocities_(nv) {}
    int num_bodies_, num_positions_, num_velocities_;
    std::vector<JointPtr<Scalar>> single_joints_;
    std::shared_ptr<LoopConstraint::Base<Scalar>> loop_constraint_;
};

// FreeJoint.cpp:10-26
template <typename Scalar = double, typename Ori = ori_representation::Quaternion>
class Free : public Base<Scalar> {
public:
    Free(const Body<Scalar> &body, std::string name = "free") : Base<Scalar>(1, Ori::num_ori_parameter + 3, 6)
    {
        if (body.parent_index_ >= 0)
            throw std::runtime_error("Free joint is only valid as the first joint in a tree and thus cannot have a parent body");
        this->single_joints_.emplace_back(new Joints::Free<Scalar, Ori>(name));
        this->ordered_joints_ = this->single_joints_;
        this->loop_constraint_ = std::make_shared<LoopConstraint::Free<Scalar>>();
    }
};
// RevoluteJoint.cpp:9-23
template <typename Scalar = double>
class Revolute : public Base<Scalar> {
public:
    Revolute(const Body<Scalar> &, ori::CoordinateAxis joint_axis, std::string name = "revolute") : Base<Scalar>(1, 1, 1)
    {
        this->single_joints_.emplace_back(new Joints::Revolute<Scalar>(joint_axis, name));
        this->ordered_joints_ = this->single_joints_;
        this->loop_constraint_ = std::make_shared<LoopConstraint::Static<Scalar>>(DMat<Scalar>::Identity(1, 1), DMat<Scalar>::Zero(0, 1));
    }
};
// RevoluteWithRotorJoint.cpp:9-31: bodies [link, rotor], G = [1; N], K = [N, -1]
template <typename Scalar = double>
class RevoluteWithRotor : public Base<Scalar> {
public:
    RevoluteWithRotor(GearedTransmissionModule<Scalar> &module) : Base<Scalar>(2, 1, 1)
    {
        this->single_joints_.emplace_back(new Joints::Revolute<Scalar>(module.joint_axis_, module.body_joint_name_));
        this->single_joints_.emplace_back(new Joints::Revolute<Scalar>(module.rotor_axis_, module.rotor_joint_name_));
        this->ordered_joints_ = this->single_joints_;
        DMat<Scalar> G(2, 1), K(1, 2);
        G(0, 0) = 1; G(1, 0) = module.gear_ratio_;
        K(0, 0) = module.gear_ratio_; K(0, 1) = -1;
        this->loop_constraint_ = std::make_shared<LoopConstraint::Static<Scalar>>(G, K);
    }
};
// RevolutePairJoint.cpp: two links in series, no constraint rows (G = identity)
template <typename Scalar = double>
class RevolutePair : public Base<Scalar> {
public:
    RevolutePair(Body<Scalar> &, Body<Scalar> &, ori::CoordinateAxis a1, ori::CoordinateAxis a2) : Base<Scalar>(2, 2, 2)
    {
        this->single_joints_.emplace_back(new Joints::Revolute<Scalar>(a1));
        this->single_joints_.emplace_back(new Joints::Revolute<Scalar>(a2));
        this->ordered_joints_ = this->single_joints_;
        this->loop_constraint_ = std::make_shared<LoopConstraint::Static<Scalar>>(DMat<Scalar>::Identity(2, 2), DMat<Scalar>::Zero(0, 2));
    }
};
// RevolutePairWithRotorJoint.cpp:10-69
template <typename Scalar = double>
class RevolutePairWithRotor : public Base<Scalar> {
public:
    using ProximalTransmission = ParallelBeltTransmissionModule<1, Scalar>;
    using DistalTransmission = ParallelBeltTransmissionModule<2, Scalar>;
    RevolutePairWithRotor(ProximalTransmission &m1, DistalTransmission &m2) : Base<Scalar>(4, 2, 2)
    {
        const int l1 = m1.body_.sub_index_within_cluster_, l2 = m2.body_.sub_index_within_cluster_;
        const int r1 = m1.rotor_.sub_index_within_cluster_, r2 = m2.rotor_.sub_index_within_cluster_;
        this->ordered_joints_.resize(4);
        this->ordered_joints_[l1] = std::make_shared<Joints::Revolute<Scalar>>(m1.joint_axis_);
        this->ordered_joints_[r1] = std::make_shared<Joints::Revolute<Scalar>>(m1.rotor_axis_);
        this->ordered_joints_[r2] = std::make_shared<Joints::Revolute<Scalar>>(m2.rotor_axis_);
        this->ordered_joints_[l2] = std::make_shared<Joints::Revolute<Scalar>>(m2.joint_axis_);
        this->single_joints_ = this->ordered_joints_;
        const auto b1 = beltMatrixRowFromBeltRatios(m1.belt_ratios_);
        const auto b2 = beltMatrixRowFromBeltRatios(m2.belt_ratios_);
        const Scalar rp00 = m1.gear_ratio_ * b1[0], rp10 = m2.gear_ratio_ * b2[0], rp11 = m2.gear_ratio_ * b2[1];
        DMat<Scalar> G(4, 2), K(2, 4);
        G(l1, 0) = 1; G(r1, 0) = rp00; G(r2, 0) = rp10; G(r2, 1) = rp11; G(l2, 1) = 1;
        const int c1 = r1 > r2, c2 = r2 > r1;
        K(c1, r1) = -1; K(c1, l1) = G(r1, 0);
        K(c2, r2) = -1; K(c2, l1) = G(r2, 0); K(c2, l2) = G(r2, 1);
        this->loop_constraint_ = std::make_shared<LoopConstraint::Static<Scalar>>(G, K);
    }
};
// RevoluteTripleWithRotorJoint.cpp:10-60: bodies [link1, link2, link3, rotor1, rotor2, rotor3]
template <typename Scalar = double>
class RevoluteTripleWithRotor : public Base<Scalar> {
public:
    using ProximalTransmission = ParallelBeltTransmissionModule<1, Scalar>;
    using IntermediateTransmission = ParallelBeltTransmissionModule<2, Scalar>;
    using DistalTransmission = ParallelBeltTransmissionModule<3, Scalar>;
    RevoluteTripleWithRotor(const ProximalTransmission &m1, const IntermediateTransmission &m2, const DistalTransmission &m3)
        : Base<Scalar>(6, 3, 3)
    {
        for (auto ax : {m1.joint_axis_, m2.joint_axis_, m3.joint_axis_}) this->single_joints_.emplace_back(new Joints::Revolute<Scalar>(ax));
        for (auto ax : {m1.rotor_axis_, m2.rotor_axis_, m3.rotor_axis_}) this->single_joints_.emplace_back(new Joints::Revolute<Scalar>(ax));
        this->ordered_joints_ = this->single_joints_;
        const auto b1 = beltMatrixRowFromBeltRatios(m1.belt_ratios_);
        const auto b2 = beltMatrixRowFromBeltRatios(m2.belt_ratios_);
        const auto b3 = beltMatrixRowFromBeltRatios(m3.belt_ratios_);
        DMat<Scalar> G(6, 3), K(3, 6);
        for (int i = 0; i < 3; i++) G(i, i) = 1;
        G(3, 0) = m1.gear_ratio_ * b1[0];
        G(4, 0) = m2.gear_ratio_ * b2[0]; G(4, 1) = m2.gear_ratio_ * b2[1];
        G(5, 0) = m3.gear_ratio_ * b3[0]; G(5, 1) = m3.gear_ratio_ * b3[1]; G(5, 2) = m3.gear_ratio_ * b3[2];
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) K(i, j) = -G(3 + i, j);
            K(i, 3 + i) = 1;
        }
        this->loop_constraint_ = std::make_shared<LoopConstraint::Static<Scalar>>(G, K);
    }
};
// GenericJoint.cpp:243-287
template <typename Scalar = double>
class Generic : public Base<Scalar> {
public:
    Generic(const std::vector<Body<Scalar>> &bodies, const std::vector<JointPtr<Scalar>> &joints,
            std::shared_ptr<LoopConstraint::Base<Scalar>> loop_constraint)
        : Base<Scalar>(static_cast<int>(bodies.size()), 0, 0)
    {
        if (bodies.size() != joints.size()) throw std::runtime_error("Generic cluster: one joint per body is required");
        this->single_joints_ = joints;
        this->ordered_joints_ = joints;
        this->loop_constraint_ = loop_constraint;
        const int k = static_cast<int>(bodies.size());
        if (loop_constraint->isExplicit()) {
            this->num_positions_ = this->num_velocities_ = loop_constraint->G().cols();
        } else {
            int n_ind = 0;
            for (bool b : loop_constraint->independent) n_ind += b ? 1 : 0;
            this->num_positions_ = k;  // spanning positions (GenericJoint.cpp:246-249)
            this->num_velocities_ = n_ind;
        }
    }
};
// A cluster joint of a model read from URDF+: sizes, single joints and loop constraint as the reader formed them (ClusterTreeParsing.cpp:
// 232-440) -- what clusters()[i]->joint_ is for such a model (numPositions(), numVelocities(), G(), K(), randomJointState()).
template <typename Scalar = double>
class Described : public Base<Scalar> {
public:
    Described(int n_bodies, int n_positions, int n_velocities, const std::vector<JointPtr<Scalar>> &joints,
              std::shared_ptr<LoopConstraint::Base<Scalar>> loop_constraint)
        : Base<Scalar>(n_bodies, n_positions, n_velocities)
    {
        this->single_joints_ = joints;
        this->ordered_joints_ = joints;
        this->loop_constraint_ = loop_constraint;
    }
};
// ClusterJoints::FourBar (FourBarJoint.h:57-77)
template <typename Scalar = double>
class FourBar : public Generic<Scalar> {
public:
    FourBar(const std::vector<Body<Scalar>> &bodies, const std::vector<JointPtr<Scalar>> &joints,
            std::shared_ptr<LoopConstraint::FourBar<Scalar>> loop_constraint)
        : Generic<Scalar>(bodies, joints, loop_constraint) {}
};
}  // namespace ClusterJoints

// ------------------------------------------------------------------------------------------------
// TreeModel / ClusterTreeModel (TreeModel.h:15-143, ClusterTreeModel.h:24-165)
// ------------------------------------------------------------------------------------------------
template <typename Scalar = double>
struct ClusterTreeNode {
    int index_ = 0;
    std::string name_;
    std::vector<Body<Scalar>> bodies_;
    std::shared_ptr<ClusterJoints::Base<Scalar>> joint_;
    int parent_index_ = -1;
    int position_index_ = 0, num_positions_ = 0, velocity_index_ = 0, num_velocities_ = 0;
    JointState<Scalar> joint_state_;
};
template <typename Scalar> using ClusterTreeNodePtr = std::shared_ptr<ClusterTreeNode<Scalar>>;

template <typename Scalar = double>
class TreeModel {
public:
    TreeModel() { gravity_[5] = Scalar(-9.81); }
    virtual ~TreeModel() {}
    const int &getNumPositions() const { return position_index_; }
    const int &getNumDegreesOfFreedom() const { return velocity_index_; }
    void setGravity(const Vec3<Scalar> &g) { for (int i = 0; i < 3; i++) gravity_[3 + i] = g[i]; plan_dirty_ = true; }
    SVec<Scalar> getGravity() const { return gravity_; }
    virtual DVec<Scalar> forwardDynamics(const DVec<Scalar> &tau) = 0;
    virtual DVec<Scalar> inverseDynamics(const DVec<Scalar> &qdd) = 0;
protected:
    SVec<Scalar> gravity_;
    int position_index_ = 0, velocity_index_ = 0;
    bool plan_dirty_ = true;
};

template <typename Scalar = double, typename OriTpl = ori_representation::Quaternion>
class ClusterTreeModel : public TreeModel<Scalar> {
public:
    ClusterTreeModel() { body_name_to_body_index_["ground"] = -1; }
    explicit ClusterTreeModel(const std::string &urdf_filename) : ClusterTreeModel() { buildModelFromURDF(urdf_filename); }
    ~ClusterTreeModel() { if (plan_) grbda_plan_free(plan_); }
    ClusterTreeModel(const ClusterTreeModel &) = delete;
    ClusterTreeModel &operator=(const ClusterTreeModel &) = delete;

    // ---- construction -------------------------------------------------------------------------
    void buildModelFromURDF(const std::string &urdf_filename) { buildModelFromURDF(std::vector<std::string>{urdf_filename}); }
    void buildModelFromURDF(const std::vector<std::string> &urdf_filenames)
    {
        std::vector<const char *> paths;
        for (const auto &s : urdf_filenames) paths.push_back(s.c_str());
        size_t need = 0;
        check(grbda_urdf_to_blob(paths.data(), static_cast<int>(paths.size()), OriTpl::desc_id, nullptr, 0, &need));
        urdf_blob_.resize(need);
        check(grbda_urdf_to_blob(paths.data(), static_cast<int>(paths.size()), OriTpl::desc_id, urdf_blob_.data(), need, &need));
        from_urdf_ = true;
        const auto *h = reinterpret_cast<const grbda_desc_header *>(urdf_blob_.data());
        this->position_index_ = h->nq;
        this->velocity_index_ = h->nv;
        n_bodies_urdf_ = h->n_bodies;
        this->plan_dirty_ = true;
        // body names (for appendContactPoint / appendEndEffector on a model read from URDF): the blob's name pool
        // holds n_bodies + n_clusters NUL-terminated strings, bodies first (include/grbda_model_desc.h)
        const size_t off = sizeof(grbda_desc_header) + sizeof(grbda_desc_body) * static_cast<size_t>(h->n_bodies) +
                           sizeof(grbda_desc_cluster) * static_cast<size_t>(h->n_clusters) +
                           4u * static_cast<size_t>(h->n_ints + (h->n_ints & 1)) + 8u * static_cast<size_t>(h->n_doubles);
        const char *names = reinterpret_cast<const char *>(urdf_blob_.data()) + off;
        const char *end = names + h->n_name_bytes;
        std::vector<std::string> body_names, cluster_names;
        for (int b = 0; b < h->n_bodies && names < end; b++) {
            const std::string nm(names);
            body_name_to_body_index_[nm] = b;
            body_names.push_back(nm);
            names += nm.size() + 1;
        }
        for (int c = 0; c < h->n_clusters && names < end; c++) {
            cluster_names.emplace_back(names);
            names += cluster_names.back().size() + 1;
        }
        // bodies() / clusters() of the model as read (the reference's callers walk them: joint sizes, G, random joint states --
        // Benchmarking/src/pinocchioBenchmark.cpp:150-168).  The dynamics keep running on the description itself.
        const auto *db = reinterpret_cast<const grbda_desc_body *>(urdf_blob_.data() + sizeof(grbda_desc_header));
        const auto *dc = reinterpret_cast<const grbda_desc_cluster *>(db + h->n_bodies);
        const auto *ints = reinterpret_cast<const int32_t *>(dc + h->n_clusters);
        const auto *dbls = reinterpret_cast<const double *>(ints + (h->n_ints + (h->n_ints & 1)));
        bodies_.clear();
        cluster_nodes_.clear();
        body_index_to_cluster_index_.clear();
        for (int b = 0; b < h->n_bodies; b++) {
            Body<Scalar> body;
            body.index_ = b;
            body.name_ = b < static_cast<int>(body_names.size()) ? body_names[b] : "body_" + std::to_string(b);
            body.parent_index_ = db[b].parent;
            Mat3<Scalar> E;
            Vec3<Scalar> r;
            for (int i = 0; i < 3; i++) {
                r[i] = static_cast<Scalar>(db[b].Xtree_r[i]);
                for (int j = 0; j < 3; j++) E(i, j) = static_cast<Scalar>(db[b].Xtree_E[3 * i + j]);
            }
            body.Xtree_ = spatial::Transform<Scalar>(E, r);
            Mat6<Scalar> I;
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 6; j++) I(i, j) = static_cast<Scalar>(db[b].inertia[6 * i + j]);
            body.inertia_ = SpatialInertia<Scalar>(I);
            body.sub_index_within_cluster_ = db[b].sub_index;
            int anc = db[b].parent;
            const int first = dc[db[b].cluster].first_body;
            while (anc >= first) anc = db[anc].parent;
            body.cluster_ancestor_index_ = anc;
            body.cluster_ancestor_sub_index_within_cluster_ = anc >= 0 ? db[anc].sub_index : 0;
            bodies_.push_back(body);
            body_index_to_cluster_index_[b] = db[b].cluster;
        }
        for (int c = 0; c < h->n_clusters; c++) {
            const grbda_desc_cluster &cl = dc[c];
            auto node = std::make_shared<ClusterTreeNode<Scalar>>();
            node->index_ = c;
            node->name_ = c < static_cast<int>(cluster_names.size()) ? cluster_names[c] : "cluster_" + std::to_string(c);
            std::vector<JointPtr<Scalar>> joints;
            for (int i = 0; i < cl.n_bodies; i++) {
                const grbda_desc_body &bd = db[cl.first_body + i];
                node->bodies_.push_back(bodies_[cl.first_body + i]);
                if (bd.joint_type == GRBDA_JOINT_FREE) joints.emplace_back(new Joints::Free<Scalar, OriTpl>());
                else joints.emplace_back(new Joints::Revolute<Scalar>(static_cast<ori::CoordinateAxis>(bd.axis)));
            }
            std::shared_ptr<LoopConstraint::Base<Scalar>> lc;
            if (cl.constraint_type == GRBDA_CONSTRAINT_FREE) {
                lc = std::make_shared<LoopConstraint::Free<Scalar>>();
            } else if (cl.constraint_type == GRBDA_CONSTRAINT_STATIC) {
                const int k = cl.n_span_vel, n = cl.n_vel, rows = cl.n_constraint_rows;
                DMat<Scalar> Gm = DMat<Scalar>::Zero(k, n), Km = DMat<Scalar>::Zero(rows, k);
                for (int i = 0; i < k; i++)
                    for (int j = 0; j < n; j++) Gm(i, j) = static_cast<Scalar>(dbls[cl.dbl_offset + i * n + j]);
                if (cl.n_dbl >= k * n + rows * k)
                    for (int i = 0; i < rows; i++)
                        for (int j = 0; j < k; j++) Km(i, j) = static_cast<Scalar>(dbls[cl.dbl_offset + k * n + i * k + j]);
                lc = std::make_shared<LoopConstraint::Static<Scalar>>(Gm, Km);
            } else {  // implicit: the description's payload (G and K depend on the state)
                lc = std::make_shared<LoopConstraint::Base<Scalar>>();
                lc->kind = cl.constraint_type;
                lc->rows = cl.n_constraint_rows;
                lc->ints.assign(ints + cl.int_offset, ints + cl.int_offset + cl.n_int);
                lc->dbls.assign(dbls + cl.dbl_offset, dbls + cl.dbl_offset + cl.n_dbl);
                const int32_t *flags = cl.constraint_type == GRBDA_CONSTRAINT_LOOP_POSITION ? lc->ints.data() + 1 : lc->ints.data();
                for (int i = 0; i < cl.n_bodies; i++) lc->independent.push_back(flags[i] != 0);
            }
            auto described = std::make_shared<ClusterJoints::Described<Scalar>>(cl.n_bodies, cl.n_pos, cl.n_vel, joints, lc);
            if (cl.constraint_type != GRBDA_CONSTRAINT_FREE && cl.constraint_type != GRBDA_CONSTRAINT_STATIC)
                described->find_roots_for_phi_ = rootFinder(cl.q_index, cl.n_bodies, lc->independent);
            node->joint_ = described;
            node->parent_index_ = cl.parent_cluster;
            node->position_index_ = cl.q_index;
            node->num_positions_ = cl.n_pos;
            node->velocity_index_ = cl.v_index;
            node->num_velocities_ = cl.n_vel;
            cluster_nodes_.push_back(node);
        }
    }

    // The root finder behind randomJointState() of an implicit cluster (GenericJoint.cpp:289-361): the cluster's spanning positions at
    // [q0, q0 + k) of the model's position vector.  The REQUESTED cluster is drawn (independent coordinates U(-1, 1), dependent guess
    // U(-0.1, 0.1)) and projected by the library's Newton kernel; the other implicit clusters rest at the last state the kernel accepted (its
    // verdict covers the whole state, so a cluster that is hard to converge must not spend the budget of the others -- the reference solves
    // per cluster) and are drawn only while no such state exists; the explicit ones rest at zero.  Up to 45 draws, then the reference's error.
    std::function<JointCoordinate<double>()> rootFinder(int q0, int k, const std::vector<bool> &)
    {
        return [this, q0, k]() {
            const int nq_all = this->position_index_;
            // (the cache is dropped with the plan on every model mutation; should the cached state still fail -- it is re-projected with the
            // new draw -- the first failed attempt forgets it and every implicit cluster is drawn afresh from then on)
            bool cached = static_cast<int>(valid_q_cache_.size()) == nq_all;
            for (int attempt = 0; attempt < 45; attempt++) {
                std::vector<double> q(nq_all, 0.0);
                for (const auto &node : cluster_nodes_) {
                    const auto lc = node->joint_->cloneLoopConstraint();
                    if (lc->kind == GRBDA_CONSTRAINT_FREE) {
                        if (node->num_positions_ == 7) q[node->position_index_ + 3] = 1.0;  // the identity quaternion
                    } else if (!lc->isExplicit()) {
                        if (cached && node->position_index_ != q0) {
                            for (int i = 0; i < node->num_positions_; i++) q[node->position_index_ + i] = valid_q_cache_[node->position_index_ + i];
                            continue;
                        }
                        const DVec<double> r = DVec<double>::Random(node->num_positions_);
                        for (int i = 0; i < node->num_positions_; i++)
                            q[node->position_index_ + i] = (i < static_cast<int>(lc->independent.size()) && lc->independent[i]) ? r[i] : 0.1 * r[i];
                    }
                }
                int32_t ok = 0;
                check(grbda_project_positions_host_f64(plan(), q.data(), &ok, 1, 50, 1e-8, 0));
                if (ok) {
                    valid_q_cache_ = q;
                    return JointCoordinate<double>(DVec<double>(q.begin() + q0, q.begin() + q0 + k), true);
                }
                if (cached && attempt == 0) {
                    cached = false;
                    valid_q_cache_.clear();
                }
            }
            throw std::runtime_error("Failed to find valid roots for implicit loop constraint");
        };
    }

    // ClusterTreeModel.cpp:10-32
    Body<Scalar> registerBody(const std::string name, const SpatialInertia<Scalar> inertia, const std::string parent_name,
                              const spatial::Transform<Scalar> Xtree)
    {
        if (body_name_to_body_index_.count(name)) throw std::runtime_error("Body " + name + " is already registered");
        auto it = body_name_to_body_index_.find(parent_name);
        if (it == body_name_to_body_index_.end()) throw std::out_of_range("map::at");  // the reference uses map::at
        Body<Scalar> b;
        b.index_ = static_cast<int>(bodies_.size());
        b.name_ = name;
        b.parent_index_ = it->second;
        b.Xtree_ = Xtree;
        b.inertia_ = inertia;
        b.sub_index_within_cluster_ = static_cast<int>(bodies_in_current_cluster_.size());
        int anc = b.parent_index_;
        const int first = bodies_in_current_cluster_.empty() ? b.index_ : bodies_in_current_cluster_.front().index_;
        while (anc >= first) anc = bodies_[anc].parent_index_;
        b.cluster_ancestor_index_ = anc;
        b.cluster_ancestor_sub_index_within_cluster_ = anc >= 0 ? bodies_[anc].sub_index_within_cluster_ : 0;
        body_name_to_body_index_[name] = b.index_;
        bodies_.push_back(b);
        bodies_in_current_cluster_.push_back(b);
        return b;
    }

    // ClusterTreeModel.h:61-66
    template <typename ClusterJointType, typename... Args>
    void appendRegisteredBodiesAsCluster(const std::string name, Args &&...args)
    {
        auto joint = std::make_shared<ClusterJointType>(args...);
        appendRegisteredBodiesAsCluster(name, std::static_pointer_cast<ClusterJoints::Base<Scalar>>(joint));
    }
    // ClusterTreeModel.h:69-78
    template <typename ClusterJointType, typename... Args>
    void appendBody(const std::string name, const SpatialInertia<Scalar> inertia, const std::string parent_name,
                    const spatial::Transform<Scalar> Xtree, Args &&...args)
    {
        Body<Scalar> body = registerBody(name, inertia, parent_name, Xtree);
        auto joint = std::make_shared<ClusterJointType>(body, args...);
        appendRegisteredBodiesAsCluster(name, std::static_pointer_cast<ClusterJoints::Base<Scalar>>(joint));
    }
    // ClusterTreeModel.cpp:34-67
    void appendRegisteredBodiesAsCluster(const std::string name, std::shared_ptr<ClusterJoints::Base<Scalar>> joint)
    {
        if (bodies_in_current_cluster_.empty()) throw std::runtime_error("Cluster is empty");
        if (static_cast<int>(bodies_in_current_cluster_.size()) != joint->numBodies())
            throw std::runtime_error("number of registered bodies does not match the cluster joint");
        auto node = std::make_shared<ClusterTreeNode<Scalar>>();
        node->index_ = static_cast<int>(cluster_nodes_.size());
        node->name_ = name;
        node->bodies_ = bodies_in_current_cluster_;
        node->joint_ = joint;
        node->position_index_ = this->position_index_;
        node->num_positions_ = joint->numPositions();
        node->velocity_index_ = this->velocity_index_;
        node->num_velocities_ = joint->numVelocities();
        // checkValidParentClusterForBodiesInCluster (ClusterTreeModel.cpp:112-126)
        const int first = node->bodies_.front().index_;
        int parent_cluster = -2;
        for (const auto &b : node->bodies_) {
            if (b.parent_index_ >= first) continue;
            const int pc = b.parent_index_ < 0 ? -1 : body_index_to_cluster_index_.at(b.parent_index_);
            if (parent_cluster == -2) parent_cluster = pc;
            else if (parent_cluster != pc)
                throw std::runtime_error("The parents of all bodies in a cluster must have parents in the current cluster OR in the same parent cluster");
        }
        node->parent_index_ = parent_cluster;
        for (const auto &b : node->bodies_) body_index_to_cluster_index_[b.index_] = node->index_;
        {
            const auto lc = joint->cloneLoopConstraint();
            if (lc && !lc->isExplicit()) joint->find_roots_for_phi_ = rootFinder(node->position_index_, joint->numPositions(), lc->independent);
        }
        cluster_nodes_.push_back(node);
        this->position_index_ += joint->numPositions();
        this->velocity_index_ += joint->numVelocities();
        bodies_in_current_cluster_.clear();
        this->plan_dirty_ = true;
    }

    // ---- accessors ------------------------------------------------------------------------------
    int getNumBodies() const { return from_urdf_ ? n_bodies_urdf_ : static_cast<int>(bodies_.size()); }
    const std::vector<Body<Scalar>> &bodies() const { return bodies_; }
    const std::vector<ClusterTreeNodePtr<Scalar>> &clusters() const { return cluster_nodes_; }
    ClusterTreeNodePtr<Scalar> cluster(int i) const { return cluster_nodes_.at(i); }
    const Body<Scalar> &body(const std::string &name) const { return bodies_.at(body_name_to_body_index_.at(name)); }
    const Body<Scalar> &body(int index) const { return bodies_.at(index); }
    const Body<Scalar> &getBody(int index) const { return bodies_.at(index); }
    // cluster bookkeeping queries (ClusterTreeModel.h:97-121, ClusterTreeModel.cpp:408-470)
    int getNumClusters() const { return static_cast<int>(cluster_nodes_.size()); }
    int getSubIndexWithinClusterForBody(int body_index) const { return body_index >= 0 ? bodies_.at(body_index).sub_index_within_cluster_ : 0; }
    int getSubIndexWithinClusterForBody(const Body<Scalar> &b) const { return getSubIndexWithinClusterForBody(b.index_); }
    int getSubIndexWithinClusterForBody(const std::string &body_name) const { return body(body_name).sub_index_within_cluster_; }
    int getNumBodiesInCluster(int cluster_index) const
    {
        return cluster_index >= 0 ? static_cast<int>(cluster_nodes_.at(cluster_index)->bodies_.size()) : 0;
    }
    int getNumBodiesInCluster(const ClusterTreeNodePtr<Scalar> &cluster) const { return static_cast<int>(cluster->bodies_.size()); }
    int getNumBodiesInCluster(const std::string &cluster_name) const
    {
        for (const auto &c : cluster_nodes_)
            if (c->name_ == cluster_name) return static_cast<int>(c->bodies_.size());
        throw std::runtime_error("no cluster named " + cluster_name);
    }
    int getIndexOfClusterContainingBody(int body_index) const { return body_index_to_cluster_index_.at(body_index); }
    int getIndexOfClusterContainingBody(const Body<Scalar> &b) const { return body_index_to_cluster_index_.at(b.index_); }
    int getIndexOfClusterContainingBody(const std::string &body_name) const { return body_index_to_cluster_index_.at(body(body_name).index_); }
    ClusterTreeNodePtr<Scalar> getClusterContainingBody(int body_index) const { return cluster_nodes_.at(getIndexOfClusterContainingBody(body_index)); }
    ClusterTreeNodePtr<Scalar> getClusterContainingBody(const Body<Scalar> &b) const { return getClusterContainingBody(b.index_); }
    ClusterTreeNodePtr<Scalar> getClusterContainingBody(const std::string &body_name) const { return getClusterContainingBody(body(body_name).index_); }
    // the nearest body at or above `body_index` that already belongs to a cluster (-1: the ground)
    int getClusterAncestorIndexFromParent(int body_index) const
    {
        int j = body_index;
        while (j != -1 && !body_index_to_cluster_index_.count(j)) j = bodies_.at(j).parent_index_;
        return j;
    }

    // ---- state (ClusterTreeModel.cpp:256-308) -----------------------------------------------------
    typedef std::pair<DVec<Scalar>, DVec<Scalar>> StatePair;
    // independent coordinates only (ClusterTreeModel.cpp:287-308)
    ModelState<Scalar> stateVectorToModelState(const StatePair &q_qd_pair) const
    {
        ModelState<Scalar> state;
        for (const auto &c : cluster_nodes_)
            state.push_back(JointState<Scalar>(
                JointCoordinate<Scalar>(q_qd_pair.first.segment(c->position_index_, c->num_positions_), false),
                JointCoordinate<Scalar>(q_qd_pair.second.segment(c->velocity_index_, c->num_velocities_), false)));
        return state;
    }
    // One JointState per cluster, each position / velocity flagged independent or spanning
    // (JointCoordinate::isSpanning()).  ClusterJoints::Base::toSpanningTreeState (ClusterJoint.cpp:22-71) decides per
    // cluster what is accepted; the same rules run in the library (grbda_state_to_independent_host_f64) and invalid
    // spanning positions / velocities throw std::runtime_error with the reference's messages.
    void setState(const ModelState<Scalar> &model_state)
    {
        const grbda_plan *pl = plan();
        int nq = 0, nv = 0, nb = 0, nc = 0;
        check(grbda_plan_dims(pl, &nq, &nv, &nb, &nc));
        if (static_cast<int>(model_state.size()) != nc) throw std::runtime_error("model state must hold one joint state per cluster");
        std::vector<uint8_t> ps(nc), vs(nc);
        std::vector<double> q, qd;
        for (int c = 0; c < nc; c++) {
            const auto &js = model_state[c];
            ps[c] = js.position.isSpanning();
            vs[c] = js.velocity.isSpanning();
            q.insert(q.end(), js.position.begin(), js.position.end());
            qd.insert(qd.end(), js.velocity.begin(), js.velocity.end());
        }
        int in_nq = 0, in_nv = 0;
        check(grbda_state_input_dims(pl, ps.data(), vs.data(), &in_nq, &in_nv));
        if (static_cast<int>(q.size()) != in_nq || static_cast<int>(qd.size()) != in_nv)
            throw std::runtime_error("state has the wrong dimension");
        std::vector<double> qo(nq), vo(nv);
        check(grbda_state_to_independent_host_f64(pl, ps.data(), vs.data(), q.data(), qd.data(), qo.data(), vo.data(), 1, 1e-8, 0));
        setState(StatePair{DVec<Scalar>(qo.begin(), qo.end()), DVec<Scalar>(vo.begin(), vo.end())});
    }
    void setState(const StatePair &q_qd_pair)
    {
        if (static_cast<int>(q_qd_pair.first.size()) != this->position_index_ ||
            static_cast<int>(q_qd_pair.second.size()) != this->velocity_index_)
            throw std::runtime_error("state has the wrong dimension");
        q_ = q_qd_pair.first;
        qd_ = q_qd_pair.second;
        f_ext_.clear();  // setState -> setExternalForces({}) (ClusterTreeModel.cpp:266)
    }
    void setState(const DVec<Scalar> &q_qd_vec)
    {
        setState(StatePair{q_qd_vec.segment(0, this->position_index_), q_qd_vec.segment(this->position_index_, this->velocity_index_)});
    }
    // TreeModel.cpp:214-239
    void setExternalForces(const std::vector<ExternalForceAndBodyIndexPair<Scalar>> &force_and_body_index_pairs = {})
    {
        f_ext_ = force_and_body_index_pairs;
    }

    // ---- dynamics: the accelerated path ------------------------------------------------------------
    DVec<Scalar> forwardDynamics(const DVec<Scalar> &tau) override { return single(false, tau); }
    DVec<Scalar> inverseDynamics(const DVec<Scalar> &qdd) override { return single(true, qdd); }

    // ---- contact side (ClusterTreeModel.cpp:165-221, TreeModel.cpp:59-76, ClusterTreeDynamics.cpp:194-435) --------
    void appendContactPoint(const std::string body_name, const Vec3<Scalar> &local_offset,
                            const std::string contact_point_name, const bool is_end_effector = false)
    {
        contact_name_to_contact_index_[contact_point_name] = static_cast<int>(contact_points_.size());
        contact_points_.emplace_back(body_name_to_body_index_.at(body_name), local_offset, contact_point_name, is_end_effector);
        if (is_end_effector) contact_points_.back().end_effector_index_ = num_end_effectors_++;
    }
    void appendEndEffector(const std::string body_name, const Vec3<Scalar> &local_offset, const std::string end_effector_name)
    {
        appendContactPoint(body_name, local_offset, end_effector_name, true);
    }
    // the eight corners of a box centred on the body frame, named as the reference names them (ClusterTreeModel.cpp:200-213)
    void appendContactBox(const std::string body_name, const Vec3<Scalar> &box_dimensions)
    {
        int n = 0;
        for (int sz = 1; sz >= -1; sz -= 2)
            for (int sy = 1; sy >= -1; sy -= 2)
                for (int sx = 1; sx >= -1; sx -= 2)
                    appendContactPoint(body_name,
                                       Vec3<Scalar>{Scalar(sx) * box_dimensions[0] / 2, Scalar(sy) * box_dimensions[1] / 2,
                                                    Scalar(sz) * box_dimensions[2] / 2},
                                       "torso-contact-" + std::to_string(++n));
    }
    const ContactPoint<Scalar> &contactPoint(int index) const { return contact_points_.at(index); }
    const std::vector<ContactPoint<Scalar>> &contactPoints() const { return contact_points_; }
    const ContactPoint<Scalar> &contactPoint(const std::string &name) const
    {
        return contact_points_[contact_name_to_contact_index_.at(name)];
    }
    int getNumEndEffectors() const { return num_end_effectors_; }
    // TreeModel::contactPointForwardKinematics (TreeModel.cpp:59-76): position_ = Xa.inverseTransformPoint(local_offset_),
    // velocity_ = the point's linear velocity in world axes (= the linear rows of the world-frame Jacobian times qd)
    void forwardKinematicsIncludingContactPoints()
    {
        const std::vector<double> Xa = bodyPoses();
        for (auto &cp : contact_points_) {
            cp.position_ = pointInWorld(Xa, cp.body_index_, cp.local_offset_);
            const DMat<Scalar> J = jacobianAt(cp.body_index_, cp.local_offset_, &Xa);
            for (int i = 0; i < 3; i++) {
                Scalar s = 0;
                for (int k = 0; k < this->velocity_index_; k++) s += J(3 + i, k) * qd_[k];
                cp.velocity_[i] = s;
            }
        }
    }
    // ClusterTreeModel::contactJacobianBodyFrame (ClusterTreeDynamics.cpp:47-79): 6 x nv, [angular; linear] in the body's axes at
    // the contact point (the frame of the end-effector force propagators)
    D6Mat<Scalar> contactJacobianBodyFrame(const std::string &cp_name)
    {
        const ContactPoint<Scalar> &cp = contactPoint(cp_name);
        return jacobianAt(cp.body_index_, cp.local_offset_, nullptr);
    }
    // ClusterTreeModel::contactJacobianWorldFrame (ClusterTreeDynamics.cpp:10-45): the same rows turned into world axes; kept in
    // ContactPoint::jacobian_ as the reference does
    const D6Mat<Scalar> &contactJacobianWorldFrame(const std::string &cp_name)
    {
        ContactPoint<Scalar> &cp = contact_points_[contact_name_to_contact_index_.at(cp_name)];
        const std::vector<double> Xa = bodyPoses();
        cp.jacobian_ = jacobianAt(cp.body_index_, cp.local_offset_, &Xa);
        return cp.jacobian_;
    }
    // TreeModel::updateContactPointJacobians (TreeModel.cpp:101-112)
    void updateContactPointJacobians()
    {
        for (auto &cp : contact_points_) contactJacobianWorldFrame(cp.name_);
    }
    // body kinematics in the world (ClusterTreeModel.cpp:320-375)
    Vec3<Scalar> getPosition(const std::string &body_name, const Vec3<Scalar> &offset = Vec3<Scalar>::Zero())
    {
        return pointInWorld(bodyPoses(), body_name_to_body_index_.at(body_name), offset);
    }
    Mat3<Scalar> getOrientation(const std::string &body_name)  // body axes -> world axes
    {
        const std::vector<double> Xa = bodyPoses();
        const double *X = &Xa[static_cast<size_t>(body_name_to_body_index_.at(body_name)) * 12];
        Mat3<Scalar> R;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) R(i, j) = static_cast<Scalar>(X[3 * j + i]);
        return R;
    }
    Vec3<Scalar> getLinearVelocity(const std::string &body_name, const Vec3<Scalar> &offset = Vec3<Scalar>::Zero())
    {
        return twistRows(body_name_to_body_index_.at(body_name), offset, 3);
    }
    Vec3<Scalar> getAngularVelocity(const std::string &body_name)
    {
        return twistRows(body_name_to_body_index_.at(body_name), Vec3<Scalar>::Zero(), 0);
    }
    // ClusterTreeModel::getLinearAcceleration / getAngularAcceleration (ClusterTreeModel.cpp:376-404): the body's spatial
    // acceleration after forwardAccelerationKinematics(qdd) -- the recursion's, with the base's -gravity -- at the point, in
    // world axes: Rai * spatialToLinearAcceleration(a, v, offset) (Spatial.h:420-437)
    Vec3<Scalar> getLinearAcceleration(const DVec<Scalar> &qdd, const std::string &body_name,
                                       const Vec3<Scalar> &offset = Vec3<Scalar>::Zero())
    {
        const int b = body_name_to_body_index_.at(body_name);
        const std::vector<double> Xa = bodyPoses(), V = bodyTwists(qdd);
        return toWorld(Xa, b, pointAcceleration(&V[static_cast<size_t>(b) * 12], offset));
    }
    Vec3<Scalar> getAngularAcceleration(const DVec<Scalar> &qdd, const std::string &body_name)
    {
        const int b = body_name_to_body_index_.at(body_name);
        const std::vector<double> Xa = bodyPoses(), V = bodyTwists(qdd);
        const double *a = &V[static_cast<size_t>(b) * 12 + 6];
        return toWorld(Xa, b, Vec3<Scalar>{Scalar(a[0]), Scalar(a[1]), Scalar(a[2])});
    }
    // TreeModel::forwardAccelerationKinematicsIncludingContactPoints (TreeModel.h:70-74, TreeModel.cpp:78-99): positions,
    // velocities and the classical accelerations of the contact points (gravity added back to the recursion's acceleration)
    void forwardAccelerationKinematicsIncludingContactPoints(const DVec<Scalar> &qdd)
    {
        forwardKinematicsIncludingContactPoints();
        const std::vector<double> Xa = bodyPoses(), V = bodyTwists(qdd);
        for (auto &cp : contact_points_) {
            const Vec3<Scalar> a = toWorld(Xa, cp.body_index_, pointAcceleration(&V[static_cast<size_t>(cp.body_index_) * 12], cp.local_offset_));
            for (int i = 0; i < 3; i++) cp.acceleration_[i] = a[i] + this->gravity_[3 + i];
        }
    }
    // ClusterTreeModel::applyTestForce: returns f^T J H^-1 J^T f, dstate_out = H^-1 J^T f (force in world axes)
    Scalar applyTestForce(const std::string &contact_point_name, const Vec3<Scalar> &force, DVec<Scalar> &dstate_out)
    {
        const ContactPoint<Scalar> &cp = contactPoint(contact_point_name);
        const std::vector<double> q = state_q();
        const double off[3] = {double(cp.local_offset_[0]), double(cp.local_offset_[1]), double(cp.local_offset_[2])};
        const double f[3] = {double(force[0]), double(force[1]), double(force[2])};
        double lam = 0;
        std::vector<double> ds(static_cast<size_t>(this->velocity_index_));
        check(grbda_apply_test_force_host_f64(plan(), q.data(), cp.body_index_, off, f, &lam, ds.data(), 1, 0));
        dstate_out = DVec<Scalar>(ds.begin(), ds.end());
        return static_cast<Scalar>(lam);
    }
    // ClusterTreeModel::inverseOperationalSpaceInertiaMatrix: 6 x 6 blocks of the end effectors, in the order
    // of their end_effector_index_, each in the frame (body axes, origin at the point) of its force propagator
    DMat<Scalar> inverseOperationalSpaceInertiaMatrix()
    {
        std::vector<int> bodies;
        std::vector<double> offsets;
        for (const auto &cp : contact_points_)
            if (cp.is_end_effector_) {
                bodies.push_back(cp.body_index_);
                for (int i = 0; i < 3; i++) offsets.push_back(static_cast<double>(cp.local_offset_[i]));
            }
        const int m = 6 * static_cast<int>(bodies.size());
        DMat<Scalar> L(m, m);
        if (m == 0) return L;
        const std::vector<double> q = state_q();
        std::vector<double> out(static_cast<size_t>(m) * m);
        check(grbda_inv_osim_host_f64(plan(), q.data(), static_cast<int>(bodies.size()), bodies.data(), offsets.data(),
                                      out.data(), nullptr, 1, 0));
        for (int i = 0; i < m; i++)
            for (int j = 0; j < m; j++) L(i, j) = static_cast<Scalar>(out[static_cast<size_t>(i) * m + j]);
        return L;
    }

    // TreeModel::updateBiasForceVector (TreeModel.cpp:162-171): C = RNEA(0), external forces included
    DVec<Scalar> getBiasForceVector() { return single(true, DVec<Scalar>::Zero(this->velocity_index_)); }
    // ClusterTreeModel::getMassMatrix (ClusterTreeModel.cpp:99-104): the batched cluster CRBA kernel (crba_kernels.hip;
    // models with implicit-loop clusters: unit-acceleration inverse-dynamics columns), one state
    DMat<Scalar> getMassMatrix()
    {
        const int nv = this->velocity_index_;
        const std::vector<double> q = state_q();
        std::vector<double> out(static_cast<size_t>(nv) * nv);
        check(grbda_mass_matrix_host_f64(plan(), q.data(), out.data(), 1, 0));
        DMat<Scalar> H(nv, nv);
        for (int i = 0; i < nv; i++)
            for (int j = 0; j < nv; j++) H(i, j) = static_cast<Scalar>(out[static_cast<size_t>(i) * nv + j]);
        return H;
    }

    // d ydd / d q, d ydd / d qd, d ydd / d tau of forwardDynamics for B states on HOST arrays, each [B][nv][nv] row-major, any
    // of them may be null (grbda_fd_derivatives_*: analytic for explicit models).  The reference obtains these matrices by
    // instantiating its templated algorithms with casadi::SX (UnitTests/testRigidBodyDynamicsAlgosDerivatives.cpp:271-383).
    void forwardDynamicsDerivativesBatch(const double *q, const double *qd, const double *tau, double *dydd_dq, double *dydd_dqd,
                                         double *dydd_dtau, size_t B, int device = 0)
    {
        check(grbda_fd_derivatives_host_f64(plan(), q, qd, tau, dydd_dq, dydd_dqd, dydd_dtau, B, device));
    }

    // batched entry points on HOST arrays (row-major q[B][nq], qd[B][nv], tau[B][nv] -> ydd[B][nv])
    void forwardDynamicsBatch(const double *q, const double *qd, const double *tau, double *ydd, size_t B, int device = 0)
    {
        check(grbda_aba_host_f64(plan(), q, qd, tau, nullptr, ydd, B, device));
    }
    void inverseDynamicsBatch(const double *q, const double *qd, const double *ydd, double *tau, size_t B, int device = 0)
    {
        check(grbda_rnea_host_f64(plan(), q, qd, ydd, nullptr, tau, B, device));
    }
    // The same on DEVICE arrays of the model's Scalar (SURVEY 8b's batched surface: ClusterTreeModel<float> runs grbda_aba_f32 /
    // grbda_rnea_f32 -- BASELINE's headline precision -- ClusterTreeModel<double> the _f64 entry points): row-major q[B][nq], qd[B][nv],
    // tau[B][nv] -> ydd[B][nv], all resident on `device`; the call only enqueues on `stream` (a hipStream_t, passed as void * so that
    // this header needs no HIP header; nullptr = the default stream) and returns.  f_ext: nullptr or [B][n_bodies][6] world-frame
    // forces (TreeModel::setExternalForces).  Reference members: ClusterTreeModel.h:158-159 over ClusterTreeDynamics.cpp:85-191 /
    // TreeModel.cpp:173-212.
    void forwardDynamicsBatch(const Scalar *q, const Scalar *qd, const Scalar *tau, Scalar *ydd, size_t B, int device, void *stream,
                              const Scalar *f_ext = nullptr)
    {
        static_assert(std::is_same<Scalar, float>::value || std::is_same<Scalar, double>::value, "device batches are float or double");
        if constexpr (std::is_same<Scalar, float>::value) check(grbda_aba_f32(plan(), q, qd, tau, f_ext, ydd, B, device, stream));
        else check(grbda_aba_f64(plan(), q, qd, tau, f_ext, ydd, B, device, stream));
    }
    void inverseDynamicsBatch(const Scalar *q, const Scalar *qd, const Scalar *ydd, Scalar *tau, size_t B, int device, void *stream,
                              const Scalar *f_ext = nullptr)
    {
        static_assert(std::is_same<Scalar, float>::value || std::is_same<Scalar, double>::value, "device batches are float or double");
        if constexpr (std::is_same<Scalar, float>::value) check(grbda_rnea_f32(plan(), q, qd, ydd, f_ext, tau, B, device, stream));
        else check(grbda_rnea_f64(plan(), q, qd, ydd, f_ext, tau, B, device, stream));
    }
    // d ydd / d q, d qd, d tau on device arrays, each [B][nv][nv] row-major, any of them may be null (grbda_fd_derivatives_f32 / _f64)
    void forwardDynamicsDerivativesBatch(const Scalar *q, const Scalar *qd, const Scalar *tau, Scalar *dydd_dq, Scalar *dydd_dqd,
                                         Scalar *dydd_dtau, size_t B, int device, void *stream)
    {
        static_assert(std::is_same<Scalar, float>::value || std::is_same<Scalar, double>::value, "device batches are float or double");
        if constexpr (std::is_same<Scalar, float>::value)
            check(grbda_fd_derivatives_f32(plan(), q, qd, tau, dydd_dq, dydd_dqd, dydd_dtau, B, device, stream));
        else check(grbda_fd_derivatives_f64(plan(), q, qd, tau, dydd_dq, dydd_dqd, dydd_dtau, B, device, stream));
    }
    // the immutable compiled plan, for the device-pointer C ABI (grbda_aba_f32 / _f64, grbda_rnea_*)
    const grbda_plan *plan()
    {
        if (this->plan_dirty_ || !plan_) {
            if (plan_) { grbda_plan_free(plan_); plan_ = nullptr; }
            valid_q_cache_.clear();  // (positions accepted for the model as it was)
            std::vector<unsigned char> blob = serialize();
            check(grbda_plan_from_blob(blob.data(), blob.size(), &plan_));
            this->plan_dirty_ = false;
        }
        return plan_;
    }

    // model description blob (include/grbda_model_desc.h)
    std::vector<unsigned char> serialize() const
    {
        if (!bodies_in_current_cluster_.empty()) throw std::runtime_error("registered bodies have not been appended as a cluster");
        if (from_urdf_) {
            std::vector<unsigned char> b = urdf_blob_;
            auto *h = reinterpret_cast<grbda_desc_header *>(b.data());
            for (int i = 0; i < 6; i++) h->gravity[i] = static_cast<double>(this->gravity_[i]);
            return b;
        }
        desc::ModelDescription md;
        md.ori_repr = OriTpl::desc_id;
        for (int i = 0; i < 6; i++) md.gravity[i] = static_cast<double>(this->gravity_[i]);
        for (const auto &node : cluster_nodes_) {
            desc::ClusterDesc cd;
            cd.name = node->name_;
            const auto lc = node->joint_->cloneLoopConstraint();
            const auto &joints = node->joint_->ordered_joints_;
            int nsv = 0, nsp = 0;
            for (size_t i = 0; i < node->bodies_.size(); i++) {
                const Body<Scalar> &b = node->bodies_[i];
                desc::BodyDesc bd;
                bd.name = b.name_;
                bd.parent = b.parent_index_;
                bd.joint_type = joints[i]->type;
                bd.axis = static_cast<int>(joints[i]->axis);
                for (int r = 0; r < 3; r++)
                    for (int c = 0; c < 3; c++) bd.E[r * 3 + c] = static_cast<double>(b.Xtree_.getRotation()(r, c));
                for (int r = 0; r < 3; r++) bd.r[r] = static_cast<double>(b.Xtree_.getTranslation()[r]);
                for (int r = 0; r < 6; r++)
                    for (int c = 0; c < 6; c++) bd.inertia[r * 6 + c] = static_cast<double>(b.inertia_.getMatrix()(r, c));
                cd.bodies.push_back(bd);
                nsv += bd.joint_type == GRBDA_JOINT_FREE ? 6 : 1;
                nsp += bd.joint_type == GRBDA_JOINT_FREE ? 3 + OriTpl::num_ori_parameter : 1;
            }
            cd.n_pos = node->num_positions_;
            cd.n_vel = node->num_velocities_;
            cd.n_span_pos = nsp;
            cd.n_span_vel = nsv;
            cd.constraint_type = lc->kind;
            cd.n_rows = lc->rows;
            if (lc->kind == GRBDA_CONSTRAINT_STATIC) {
                const auto &G = lc->G();
                const auto &K = lc->K();
                if (G.rows() != nsv || G.cols() != cd.n_vel) throw std::runtime_error("loop constraint G has the wrong shape");
                for (int r = 0; r < G.rows(); r++)
                    for (int c = 0; c < G.cols(); c++) cd.dbls.push_back(static_cast<double>(G(r, c)));
                for (int r = 0; r < K.rows(); r++)
                    for (int c = 0; c < K.cols(); c++) cd.dbls.push_back(static_cast<double>(K(r, c)));
            } else {
                cd.ints = lc->ints;
                cd.dbls = lc->dbls;
            }
            md.appendCluster(cd);
        }
        return md.serialize();
    }

private:
    static void check(int rc)
    {
        if (rc != GRBDA_OK) throw std::runtime_error(std::string(grbda_strerror(rc)) + ": " + grbda_last_error());
    }
    std::vector<double> state_q() const
    {
        if (static_cast<int>(q_.size()) != this->position_index_) throw std::runtime_error("state has not been set");
        return std::vector<double>(q_.begin(), q_.end());
    }
    // Xa[n_bodies][12] of the current state: rotation world -> body (row-major), origin in the world (grbda_body_poses_*)
    std::vector<double> bodyPoses()
    {
        const std::vector<double> q = state_q();
        std::vector<double> Xa(static_cast<size_t>(getNumBodies()) * 12);
        check(grbda_body_poses_host_f64(plan(), q.data(), Xa.data(), 1, 0));
        return Xa;
    }
    // [v 6 | a 6] of every body in its own coordinates for the current state and the given accelerations (grbda_body_twists_*)
    std::vector<double> bodyTwists(const DVec<Scalar> &qdd)
    {
        if (static_cast<int>(qdd.size()) != this->velocity_index_) throw std::runtime_error("input has the wrong dimension");
        const std::vector<double> q = state_q(), qd(qd_.begin(), qd_.end()), ydd(qdd.begin(), qdd.end());
        std::vector<double> V(static_cast<size_t>(getNumBodies()) * 12);
        check(grbda_body_twists_host_f64(plan(), q.data(), qd.data(), ydd.data(), V.data(), 1, 0));
        return V;
    }
    // spatialToLinearAcceleration(a, v, x): (a_lin + alpha x x) + omega x (v_lin + omega x x), body coordinates
    static Vec3<Scalar> pointAcceleration(const double *va, const Vec3<Scalar> &x)
    {
        const double *v = va, *a = va + 6;
        const double xs[3] = {double(x[0]), double(x[1]), double(x[2])};
        auto cross = [](const double *p, const double *r, double *o) {
            o[0] = p[1] * r[2] - p[2] * r[1];
            o[1] = p[2] * r[0] - p[0] * r[2];
            o[2] = p[0] * r[1] - p[1] * r[0];
        };
        double ax[3], wx[3], vl[3], wv[3];
        cross(a, xs, ax);
        cross(v, xs, wx);
        for (int i = 0; i < 3; i++) vl[i] = v[3 + i] + wx[i];
        cross(v, vl, wv);
        return Vec3<Scalar>{Scalar(a[3] + ax[0] + wv[0]), Scalar(a[4] + ax[1] + wv[1]), Scalar(a[5] + ax[2] + wv[2])};
    }
    static Vec3<Scalar> toWorld(const std::vector<double> &Xa, int body_index, const Vec3<Scalar> &x)  // E^T x
    {
        const double *X = &Xa[static_cast<size_t>(body_index) * 12];
        Vec3<Scalar> w;
        for (int i = 0; i < 3; i++) w[i] = static_cast<Scalar>(X[i] * x[0] + X[3 + i] * x[1] + X[6 + i] * x[2]);
        return w;
    }
    static Vec3<Scalar> pointInWorld(const std::vector<double> &Xa, int body_index, const Vec3<Scalar> &offset)
    {
        const double *X = &Xa[static_cast<size_t>(body_index) * 12];
        Vec3<Scalar> p;
        for (int i = 0; i < 3; i++)
            p[i] = static_cast<Scalar>(X[9 + i] + X[i] * offset[0] + X[3 + i] * offset[1] + X[6 + i] * offset[2]);
        return p;
    }
    // 6 x nv Jacobian of the frame (body axes, origin at `offset`) on a body, from the force-propagation kernel
    // (grbda_inv_osim_*: J next to the operational-space inertia); with poses: rows turned into world axes
    DMat<Scalar> jacobianAt(int body_index, const Vec3<Scalar> &offset, const std::vector<double> *Xa)
    {
        const int nv = this->velocity_index_;
        const std::vector<double> q = state_q();
        const double off[3] = {double(offset[0]), double(offset[1]), double(offset[2])};
        std::vector<double> L(36), Jb(static_cast<size_t>(6) * nv);
        check(grbda_inv_osim_host_f64(plan(), q.data(), 1, &body_index, off, L.data(), Jb.data(), 1, 0));
        DMat<Scalar> J(6, nv);
        for (int half = 0; half < 2; half++)
            for (int i = 0; i < 3; i++)
                for (int k = 0; k < nv; k++) {
                    double s = 0;
                    if (Xa) {  // world axes: E^T on each 3-row half
                        const double *X = &(*Xa)[static_cast<size_t>(body_index) * 12];
                        for (int m = 0; m < 3; m++) s += X[3 * m + i] * Jb[static_cast<size_t>(3 * half + m) * nv + k];
                    } else {
                        s = Jb[static_cast<size_t>(3 * half + i) * nv + k];
                    }
                    J(3 * half + i, k) = static_cast<Scalar>(s);
                }
        return J;
    }
    // rows first .. first + 2 of (world-frame Jacobian at the point) x qd: 0 -> angular velocity, 3 -> linear velocity
    Vec3<Scalar> twistRows(int body_index, const Vec3<Scalar> &offset, int first)
    {
        const std::vector<double> Xa = bodyPoses();
        const DMat<Scalar> J = jacobianAt(body_index, offset, &Xa);
        Vec3<Scalar> v;
        for (int i = 0; i < 3; i++) {
            Scalar s = 0;
            for (int k = 0; k < this->velocity_index_; k++) s += J(first + i, k) * qd_[k];
            v[i] = s;
        }
        return v;
    }
    DVec<Scalar> single(bool inverse, const DVec<Scalar> &x)
    {
        if (static_cast<int>(x.size()) != this->velocity_index_) throw std::runtime_error("input has the wrong dimension");
        if (static_cast<int>(q_.size()) != this->position_index_) throw std::runtime_error("state has not been set");
        std::vector<double> q(q_.begin(), q_.end()), qd(qd_.begin(), qd_.end()), in(x.begin(), x.end()), out(x.size());
        // TreeModel::setExternalForces (TreeModel.cpp:214-239): forces on the same body add up
        std::vector<double> fe;
        if (!f_ext_.empty()) {
            fe.assign(static_cast<size_t>(getNumBodies()) * 6, 0.0);
            for (const auto &fb : f_ext_) {
                if (fb.index_ < 0 || fb.index_ >= getNumBodies()) throw std::runtime_error("external force on an unknown body");
                for (int i = 0; i < 6; i++) fe[static_cast<size_t>(fb.index_) * 6 + i] += static_cast<double>(fb.force_[i]);
            }
        }
        const double *fp = fe.empty() ? nullptr : fe.data();
        check(inverse ? grbda_rnea_host_f64(plan(), q.data(), qd.data(), in.data(), fp, out.data(), 1, 0)
                      : grbda_aba_host_f64(plan(), q.data(), qd.data(), in.data(), fp, out.data(), 1, 0));
        return DVec<Scalar>(out.begin(), out.end());
    }

    std::vector<Body<Scalar>> bodies_, bodies_in_current_cluster_;
    std::vector<ClusterTreeNodePtr<Scalar>> cluster_nodes_;
    std::map<std::string, int> body_name_to_body_index_;
    std::map<int, int> body_index_to_cluster_index_;
    DVec<Scalar> q_, qd_;
    std::vector<ExternalForceAndBodyIndexPair<Scalar>> f_ext_;
    std::vector<ContactPoint<Scalar>> contact_points_;
    std::map<std::string, int> contact_name_to_contact_index_;
    int num_end_effectors_ = 0;
    std::vector<unsigned char> urdf_blob_;
    bool from_urdf_ = false;
    int n_bodies_urdf_ = 0;
    grbda_plan *plan_ = nullptr;
    std::vector<double> valid_q_cache_;  // the last full position vector the Newton projection accepted (rootFinder)
};

}  // namespace grbda
