// ModelDescription.h -- header-only writer of the model-description blob
// (include/grbda_model_desc.h).  Used by the URDF+ reader (csrc/urdf.cpp) and by the C++ facade
// (grbda/Dynamics/ClusterTreeModel.h): both describe a model as the reference's
// ClusterTreeModel holds it (bodies_ + cluster_nodes_, ClusterTreeModel.cpp:10-67).
#pragma once

#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/grbda_model_desc.h"

namespace grbda {
namespace desc {

struct BodyDesc {
    std::string name;
    int parent = -1;  // global body index, -1 ground
    int joint_type = GRBDA_JOINT_REVOLUTE;
    int axis = 2;
    std::array<double, 9> E{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
    std::array<double, 3> r{{0, 0, 0}};
    std::array<double, 36> inertia{};
};

struct ClusterDesc {
    std::string name;
    std::vector<BodyDesc> bodies;
    int n_pos = 0, n_vel = 0, n_span_pos = 0, n_span_vel = 0;
    int constraint_type = GRBDA_CONSTRAINT_STATIC;
    int n_rows = 0;
    std::vector<int32_t> ints;
    std::vector<double> dbls;
};

class ModelDescription {
public:
    int ori_repr = GRBDA_ORI_QUATERNION;
    double gravity[6] = {0, 0, 0, 0, 0, -9.81};  // TreeModel.h:19-22

    int numBodies() const { return n_bodies_; }
    int numClusters() const { return static_cast<int>(clusters_.size()); }
    int nq() const { return nq_; }
    int nv() const { return nv_; }
    const std::vector<ClusterDesc> &clusters() const { return clusters_; }
    int clusterOfBody(int body) const { return body < 0 ? -1 : body_cluster_.at(body); }

    // Appends a cluster; its bodies receive the next global indices in order.  Enforces the
    // reference's rule that every body's parent is in this cluster or in one single parent cluster
    // (ClusterTreeModel.cpp:112-126).  Returns the cluster index.
    int appendCluster(const ClusterDesc &c)
    {
        if (c.bodies.empty()) throw std::runtime_error("Cluster is empty");
        const int first = n_bodies_;
        int parent_cluster = -2;
        for (size_t i = 0; i < c.bodies.size(); i++) {
            const int p = c.bodies[i].parent;
            if (p >= first + static_cast<int>(i)) throw std::runtime_error("body parent must be registered before the body");
            if (p >= first) continue;
            const int pc = clusterOfBody(p);
            if (parent_cluster == -2) parent_cluster = pc;
            else if (parent_cluster != pc)
                throw std::runtime_error("The parents of all bodies in a cluster must have parents in the current cluster OR in the same parent cluster");
        }
        if (parent_cluster == -2) throw std::runtime_error("cluster has no body attached to a parent cluster");
        clusters_.push_back(c);
        parent_cluster_.push_back(parent_cluster);
        first_body_.push_back(first);
        q_index_.push_back(nq_);
        v_index_.push_back(nv_);
        for (size_t i = 0; i < c.bodies.size(); i++) body_cluster_.push_back(static_cast<int>(clusters_.size()) - 1);
        n_bodies_ += static_cast<int>(c.bodies.size());
        nq_ += c.n_pos;
        nv_ += c.n_vel;
        return static_cast<int>(clusters_.size()) - 1;
    }

    std::vector<unsigned char> serialize() const
    {
        std::vector<int32_t> ints;
        std::vector<double> dbls;
        std::vector<grbda_desc_cluster> crecs;
        std::vector<grbda_desc_body> brecs;
        std::string names;
        for (size_t c = 0; c < clusters_.size(); c++) {
            const ClusterDesc &cl = clusters_[c];
            grbda_desc_cluster r;
            std::memset(&r, 0, sizeof r);
            r.parent_cluster = parent_cluster_[c];
            r.first_body = first_body_[c];
            r.n_bodies = static_cast<int32_t>(cl.bodies.size());
            r.q_index = q_index_[c];
            r.n_pos = cl.n_pos;
            r.v_index = v_index_[c];
            r.n_vel = cl.n_vel;
            r.n_span_pos = cl.n_span_pos;
            r.n_span_vel = cl.n_span_vel;
            r.constraint_type = cl.constraint_type;
            r.n_constraint_rows = cl.n_rows;
            r.int_offset = static_cast<int32_t>(ints.size());
            r.n_int = static_cast<int32_t>(cl.ints.size());
            r.dbl_offset = static_cast<int32_t>(dbls.size());
            r.n_dbl = static_cast<int32_t>(cl.dbls.size());
            ints.insert(ints.end(), cl.ints.begin(), cl.ints.end());
            dbls.insert(dbls.end(), cl.dbls.begin(), cl.dbls.end());
            crecs.push_back(r);
            for (size_t i = 0; i < cl.bodies.size(); i++) {
                const BodyDesc &b = cl.bodies[i];
                grbda_desc_body br;
                std::memset(&br, 0, sizeof br);
                br.parent = b.parent;
                br.cluster = static_cast<int32_t>(c);
                br.sub_index = static_cast<int32_t>(i);
                br.joint_type = b.joint_type;
                br.axis = b.axis;
                std::memcpy(br.Xtree_E, b.E.data(), sizeof br.Xtree_E);
                std::memcpy(br.Xtree_r, b.r.data(), sizeof br.Xtree_r);
                std::memcpy(br.inertia, b.inertia.data(), sizeof br.inertia);
                brecs.push_back(br);
            }
        }
        for (const auto &cl : clusters_)
            for (const auto &b : cl.bodies) names += b.name + '\0';
        for (const auto &cl : clusters_) names += cl.name + '\0';
        while (names.size() % 8) names += '\0';
        if (ints.size() % 2) ints.push_back(0);

        grbda_desc_header h;
        std::memset(&h, 0, sizeof h);
        h.magic = GRBDA_DESC_MAGIC;
        h.version = GRBDA_DESC_VERSION;
        h.n_bodies = n_bodies_;
        h.n_clusters = static_cast<int32_t>(clusters_.size());
        h.nq = nq_;
        h.nv = nv_;
        h.ori_repr = ori_repr;
        int n_ints = 0;
        for (const auto &cl : clusters_) n_ints += static_cast<int>(cl.ints.size());
        h.n_ints = n_ints;
        h.n_doubles = static_cast<int32_t>(dbls.size());
        h.n_name_bytes = static_cast<int32_t>(names.size());
        std::memcpy(h.gravity, gravity, sizeof h.gravity);

        std::vector<unsigned char> out;
        auto put = [&out](const void *p, size_t n) {
            const unsigned char *b = static_cast<const unsigned char *>(p);
            out.insert(out.end(), b, b + n);
        };
        put(&h, sizeof h);
        if (!brecs.empty()) put(brecs.data(), brecs.size() * sizeof(grbda_desc_body));
        if (!crecs.empty()) put(crecs.data(), crecs.size() * sizeof(grbda_desc_cluster));
        if (!ints.empty()) put(ints.data(), ints.size() * sizeof(int32_t));
        if (!dbls.empty()) put(dbls.data(), dbls.size() * sizeof(double));
        put(names.data(), names.size());
        return out;
    }

private:
    std::vector<ClusterDesc> clusters_;
    std::vector<int> parent_cluster_, first_body_, q_index_, v_index_, body_cluster_;
    int n_bodies_ = 0, nq_ = 0, nv_ = 0;
};

}  // namespace desc
}  // namespace grbda
