// urdf.cpp -- URDF+ reader: file(s) -> model description blob.
//
// Replaces the reference's ClusterTreeModel::buildModelFromURDF
// (include/grbda/Dynamics/ClusterTreeModel.h:41-53) = urdf::parseURDFFile(s) of the
// mit-biomimetics urdfdom fork (NOT vendored in the reference) followed by
// ClusterTreeModel::buildFromUrdfModelInterface (src/Dynamics/ClusterTreeParsing.cpp:5-440).
//
// The element set handled is the one the reference's robot-models/*.urdf use:
//   <link name><inertial><mass value/><origin xyz rpy/><inertia ixx ixy ixz iyy iyz izz/></inertial></link>
//   <joint name type=revolute|continuous|floating independent=true|false>
//       <parent link/><child link/><origin xyz rpy/><axis xyz/></joint>
//   <coupling name><predecessor link/><successor link/><ratio value/></coupling>
//   <loop name><predecessor link><origin xyz rpy/></predecessor><successor link><origin/></successor></loop>
//
// Because the fork that forms clusters is absent, the rules below are this build's own; they are
// pinned by the reference's hand-built robots (UnitTests/testClusterTreeModel.cpp:100-114 requires
// URDF-built == hand-built, cluster by cluster):
//   * clusters: links joined by a constraint (every link on the NCA->predecessor and
//     NCA->successor sub-chains) are one cluster; an unconstrained link is its own cluster;
//   * bodies inside a cluster: ascending link name, a body only after its parent
//     (registerBodiesInUrdfCluster, ClusterTreeParsing.cpp:260-307) -- gives the MIT-Humanoid
//     knee/ankle order [ankle_rotor, knee_link, knee_rotor, ankle_link] (MIT_Humanoid.cpp:172-179);
//   * child clusters: depth-first (ClusterTreeParsing.cpp:29-43), the children of a cluster in REVERSE order of
//     their first discovery by a link-level walk that follows child joints in ascending joint name and then
//     loop links (see "cluster order" below) -- reproduces MiniCheetah.cpp:29 ({HR, HL, FR, FL}) and
//     MIT_Humanoid.cpp:351-354 (right arm, right leg, left arm, left leg); tests/test_urdf_vs_manual.py.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/grbda_hip.h"
#include "../include/grbda/ModelDescription.h"

namespace grbda_hip {

namespace {

// ---------------------------------------------------------------------------------------------
// a very small XML reader (elements + attributes; text, comments, PIs and DOCTYPE are skipped)
// ---------------------------------------------------------------------------------------------
struct XmlNode {
    std::string tag;
    std::map<std::string, std::string> attr;
    std::vector<XmlNode> kids;
    const XmlNode *child(const std::string &t) const
    {
        for (const auto &k : kids)
            if (k.tag == t) return &k;
        return nullptr;
    }
    bool has(const std::string &a) const { return attr.count(a) != 0; }
    std::string get(const std::string &a, const std::string &d = "") const
    {
        auto it = attr.find(a);
        return it == attr.end() ? d : it->second;
    }
};

struct XmlParser {
    const std::string &s;
    size_t p = 0;
    std::string err;
    explicit XmlParser(const std::string &src) : s(src) {}

    void skip_ws() { while (p < s.size() && std::isspace(static_cast<unsigned char>(s[p]))) p++; }
    bool starts(const char *lit) const { return s.compare(p, std::strlen(lit), lit) == 0; }
    bool skip_misc()
    {  // whitespace, comments, processing instructions, doctype, text
        for (;;) {
            while (p < s.size() && s[p] != '<') p++;
            if (p >= s.size()) return true;
            if (starts("<!--")) {
                size_t e = s.find("-->", p + 4);
                if (e == std::string::npos) { err = "unterminated comment"; return false; }
                p = e + 3;
            } else if (starts("<?")) {
                size_t e = s.find("?>", p + 2);
                if (e == std::string::npos) { err = "unterminated processing instruction"; return false; }
                p = e + 2;
            } else if (starts("<!")) {
                size_t e = s.find('>', p);
                if (e == std::string::npos) { err = "unterminated declaration"; return false; }
                p = e + 1;
            } else
                return true;
        }
    }
    static bool name_char(char c) { return std::isalnum(static_cast<unsigned char>(c)) || c == '_' || c == '-' || c == ':' || c == '.'; }

    bool parse_element(XmlNode &n)
    {
        if (p >= s.size() || s[p] != '<') { err = "expected '<'"; return false; }
        p++;
        size_t b = p;
        while (p < s.size() && name_char(s[p])) p++;
        n.tag = s.substr(b, p - b);
        if (n.tag.empty()) { err = "empty tag name"; return false; }
        for (;;) {
            skip_ws();
            if (p >= s.size()) { err = "unexpected end inside <" + n.tag + ">"; return false; }
            if (s[p] == '/') {
                if (p + 1 < s.size() && s[p + 1] == '>') { p += 2; return true; }
                err = "malformed tag <" + n.tag + ">";
                return false;
            }
            if (s[p] == '>') { p++; break; }
            size_t ab = p;
            while (p < s.size() && name_char(s[p])) p++;
            std::string an = s.substr(ab, p - ab);
            skip_ws();
            if (an.empty() || p >= s.size() || s[p] != '=') { err = "malformed attribute in <" + n.tag + ">"; return false; }
            p++;
            skip_ws();
            if (p >= s.size() || (s[p] != '"' && s[p] != '\'')) { err = "unquoted attribute in <" + n.tag + ">"; return false; }
            const char qc = s[p++];
            size_t vb = p;
            while (p < s.size() && s[p] != qc) p++;
            if (p >= s.size()) { err = "unterminated attribute value"; return false; }
            n.attr[an] = s.substr(vb, p - vb);
            p++;
        }
        // children until </tag>
        for (;;) {
            if (!skip_misc()) return false;
            if (p >= s.size()) { err = "missing </" + n.tag + ">"; return false; }
            if (starts("</")) {
                size_t e = s.find('>', p);
                if (e == std::string::npos) { err = "unterminated closing tag"; return false; }
                std::string ct = s.substr(p + 2, e - p - 2);
                while (!ct.empty() && std::isspace(static_cast<unsigned char>(ct.back()))) ct.pop_back();
                if (ct != n.tag) { err = "mismatched </" + ct + "> for <" + n.tag + ">"; return false; }
                p = e + 1;
                return true;
            }
            XmlNode k;
            if (!parse_element(k)) return false;
            n.kids.push_back(std::move(k));
        }
    }
    bool parse_document(XmlNode &root)
    {
        if (!skip_misc()) return false;
        if (p >= s.size()) { err = "no root element"; return false; }
        return parse_element(root);
    }
};

// ---------------------------------------------------------------------------------------------
// URDF+ model
// ---------------------------------------------------------------------------------------------
struct Pose {
    double xyz[3] = {0, 0, 0};
    double rpy[3] = {0, 0, 0};
};
struct ULink {
    std::string name;
    bool has_inertial = false;
    double mass = 0, com[3] = {0, 0, 0}, I[6] = {0, 0, 0, 0, 0, 0};  // ixx ixy ixz iyy iyz izz
    std::string parent_joint;  // empty for the root
};
struct UJoint {
    std::string name, type, parent, child;
    Pose origin;
    double axis[3] = {1, 0, 0};
    bool independent = true;
};
struct UConstraint {
    std::string name;
    bool is_loop = false;
    std::string pred, succ;
    double ratio = 1.0;
    Pose pred_origin, succ_origin;
};
struct UModel {
    std::map<std::string, ULink> links;
    std::map<std::string, UJoint> joints;
    std::vector<UConstraint> constraints;
};

bool parse_vec3(const std::string &txt, double *v)
{
    std::istringstream is(txt);
    return static_cast<bool>(is >> v[0] >> v[1] >> v[2]);
}
bool parse_double(const std::string &txt, double &v)
{
    char *e = nullptr;
    v = std::strtod(txt.c_str(), &e);
    return e != txt.c_str();
}
bool parse_pose(const XmlNode *n, Pose &p, std::string &err)
{
    if (!n) return true;
    if (n->has("xyz") && !parse_vec3(n->get("xyz"), p.xyz)) { err = "bad origin xyz"; return false; }
    if (n->has("rpy") && !parse_vec3(n->get("rpy"), p.rpy)) { err = "bad origin rpy"; return false; }
    return true;
}

bool load_robot(const XmlNode &robot, UModel &m, std::string &err)
{
    if (robot.tag != "robot") { err = "root element is <" + robot.tag + ">, expected <robot>"; return false; }
    for (const XmlNode &e : robot.kids) {
        if (e.tag == "link") {
            ULink l;
            l.name = e.get("name");
            if (l.name.empty()) { err = "<link> without a name"; return false; }
            if (const XmlNode *in = e.child("inertial")) {
                l.has_inertial = true;
                const XmlNode *ms = in->child("mass"), *ine = in->child("inertia");
                if (!ms || !parse_double(ms->get("value"), l.mass)) { err = "link " + l.name + ": bad <mass>"; return false; }
                Pose o;
                if (!parse_pose(in->child("origin"), o, err)) return false;
                std::memcpy(l.com, o.xyz, sizeof l.com);  // rotation of the inertial frame is ignored (SpatialInertia.h:105-116)
                static const char *keys[6] = {"ixx", "ixy", "ixz", "iyy", "iyz", "izz"};
                if (!ine) { err = "link " + l.name + ": missing <inertia>"; return false; }
                for (int i = 0; i < 6; i++)
                    if (!parse_double(ine->get(keys[i]), l.I[i])) { err = "link " + l.name + ": bad <inertia>"; return false; }
            }
            auto it = m.links.find(l.name);
            if (it == m.links.end()) m.links[l.name] = l;
            else if (l.has_inertial && !it->second.has_inertial) it->second = l;  // multi-file merge
        } else if (e.tag == "joint") {
            UJoint j;
            j.name = e.get("name");
            j.type = e.get("type");
            if (j.name.empty()) { err = "<joint> without a name"; return false; }
            const XmlNode *pa = e.child("parent"), *ch = e.child("child");
            if (!pa || !ch) { err = "joint " + j.name + ": missing parent/child"; return false; }
            j.parent = pa->get("link");
            j.child = ch->get("link");
            if (!parse_pose(e.child("origin"), j.origin, err)) return false;
            if (const XmlNode *ax = e.child("axis"))
                if (!parse_vec3(ax->get("xyz"), j.axis)) { err = "joint " + j.name + ": bad axis"; return false; }
            const std::string ind = e.get("independent", "true");
            j.independent = !(ind == "false" || ind == "0" || ind == "False");
            if (m.joints.count(j.name)) { err = "joint " + j.name + " defined twice"; return false; }
            m.joints[j.name] = j;
        } else if (e.tag == "coupling" || e.tag == "loop") {
            UConstraint c;
            c.name = e.get("name");
            c.is_loop = e.tag == "loop";
            const XmlNode *pr = e.child("predecessor"), *su = e.child("successor");
            if (!pr || !su) { err = "constraint " + c.name + ": missing predecessor/successor"; return false; }
            c.pred = pr->get("link");
            c.succ = su->get("link");
            if (c.is_loop) {
                if (!parse_pose(pr->child("origin"), c.pred_origin, err) || !parse_pose(su->child("origin"), c.succ_origin, err))
                    return false;
            } else {
                const XmlNode *ra = e.child("ratio");
                if (!ra || !parse_double(ra->get("value"), c.ratio)) { err = "coupling " + c.name + ": bad <ratio>"; return false; }
            }
            m.constraints.push_back(c);
        }
        // everything else (material, transmission, gazebo, ...) is irrelevant to the dynamics
    }
    return true;
}

// urdf::Rotation::setFromRPY -> quaternion -> ori::quaternionToRotationMatrix
// (spatial::Transform(urdf::Pose), SpatialTransforms.cpp:16-23)
void pose_to_transform(const Pose &p, double *E, double *r)
{
    const double phi = p.rpy[0] / 2, the = p.rpy[1] / 2, psi = p.rpy[2] / 2;
    const double x = std::sin(phi) * std::cos(the) * std::cos(psi) - std::cos(phi) * std::sin(the) * std::sin(psi);
    const double y = std::cos(phi) * std::sin(the) * std::cos(psi) + std::sin(phi) * std::cos(the) * std::sin(psi);
    const double z = std::cos(phi) * std::cos(the) * std::sin(psi) - std::sin(phi) * std::sin(the) * std::cos(psi);
    const double w = std::cos(phi) * std::cos(the) * std::cos(psi) + std::sin(phi) * std::sin(the) * std::sin(psi);
    const double e0 = w, e1 = x, e2 = y, e3 = z;
    const double R[9] = {1 - 2 * (e2 * e2 + e3 * e3), 2 * (e1 * e2 - e0 * e3), 2 * (e1 * e3 + e0 * e2),
                         2 * (e1 * e2 + e0 * e3), 1 - 2 * (e1 * e1 + e3 * e3), 2 * (e2 * e3 - e0 * e1),
                         2 * (e1 * e3 - e0 * e2), 2 * (e2 * e3 + e0 * e1), 1 - 2 * (e1 * e1 + e2 * e2)};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) E[i * 3 + j] = R[j * 3 + i];
    for (int i = 0; i < 3; i++) r[i] = p.xyz[i];
}

// SpatialInertia(m, com, I) (SpatialInertia.h:74-82,105-116)
void spatial_inertia(const ULink &l, double *M)
{
    const double *c = l.com;
    const double cs[9] = {0, -c[2], c[1], c[2], 0, -c[0], -c[1], c[0], 0};
    const double I3[9] = {l.I[0], l.I[1], l.I[2], l.I[1], l.I[3], l.I[4], l.I[2], l.I[4], l.I[5]};
    std::memset(M, 0, sizeof(double) * 36);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double cct = 0;
            for (int k = 0; k < 3; k++) cct += cs[i * 3 + k] * cs[j * 3 + k];
            M[i * 6 + j] = I3[i * 3 + j] + l.mass * cct;
            M[i * 6 + 3 + j] = l.mass * cs[i * 3 + j];
            M[(3 + i) * 6 + j] = l.mass * cs[j * 3 + i];
        }
    for (int i = 0; i < 3; i++) M[(3 + i) * 6 + 3 + i] = l.mass;
}

// ori::urdfAxisToCoordinateAxis (OrientationTools.h:70-93): negative axes map to the positive axis
int coordinate_axis(const double *a, std::string &err)
{
    if (std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) != 1.0) { err = "Error: Joint axis must be a unit vector"; return -1; }
    if (a[0] == 1 || a[0] == -1) return 0;
    if (a[1] == 1 || a[1] == -1) return 1;
    if (a[2] == 1 || a[2] == -1) return 2;
    err = "Error: Joint axis not defined";
    return -1;
}

struct Mat3 {
    double m[9];
};
Mat3 mul3(const double *A, const double *B)
{
    Mat3 C;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C.m[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
    return C;
}
void coord_rot(int axis, double th, double *R)
{
    const double s = std::sin(th), c = std::cos(th);
    const double X[9] = {1, 0, 0, 0, c, s, 0, -s, c}, Y[9] = {c, 0, -s, 0, 1, 0, s, 0, c}, Z[9] = {c, s, 0, -s, c, 0, 0, 0, 1};
    std::memcpy(R, axis == 0 ? X : (axis == 1 ? Y : Z), sizeof X);
}

}  // namespace

int urdf_to_blob(const char *const *paths, int n_paths, int ori_repr, std::vector<unsigned char> &blob, std::string &err)
{
    UModel um;
    for (int f = 0; f < n_paths; f++) {
        std::ifstream in(paths[f], std::ios::binary);
        if (!in) { err = std::string("Could not parse URDF file: cannot open ") + paths[f]; return GRBDA_EPARSE; }
        std::stringstream ss;
        ss << in.rdbuf();
        const std::string src = ss.str();
        XmlParser xp(src);
        XmlNode root;
        if (!xp.parse_document(root)) { err = std::string(paths[f]) + ": " + xp.err; return GRBDA_EPARSE; }
        if (!load_robot(root, um, err)) { err = std::string(paths[f]) + ": " + err; return GRBDA_EPARSE; }
    }
    // tree
    for (auto &kv : um.joints) {
        UJoint &j = kv.second;
        if (!um.links.count(j.parent)) { err = "joint " + j.name + ": unknown parent link " + j.parent; return GRBDA_EPARSE; }
        if (!um.links.count(j.child)) { err = "joint " + j.name + ": unknown child link " + j.child; return GRBDA_EPARSE; }
        ULink &c = um.links[j.child];
        if (!c.parent_joint.empty()) { err = "link " + c.name + " has two parent joints"; return GRBDA_EPARSE; }
        c.parent_joint = j.name;
    }
    std::string root;
    for (auto &kv : um.links)
        if (kv.second.parent_joint.empty()) {
            if (!root.empty()) { err = "URDF has more than one root link (" + root + ", " + kv.first + ")"; return GRBDA_EPARSE; }
            root = kv.first;
        }
    if (root.empty()) { err = "URDF has no root link"; return GRBDA_EPARSE; }
    auto parent_of = [&](const std::string &l) -> std::string {
        const ULink &lk = um.links.at(l);
        return lk.parent_joint.empty() ? std::string() : um.joints.at(lk.parent_joint).parent;
    };
    auto chain_to_root = [&](const std::string &l) {
        std::vector<std::string> c;
        for (std::string x = l; !x.empty(); x = parent_of(x)) {
            c.push_back(x);
            if (c.size() > um.links.size()) break;  // cycle guard
        }
        return c;
    };
    for (auto &kv : um.links)
        if (chain_to_root(kv.first).size() > um.links.size()) { err = "kinematic cycle in the joint tree"; return GRBDA_EPARSE; }

    // constraints: nearest common ancestor + sub-chains; union-find over links
    std::map<std::string, std::string> uf;
    for (auto &kv : um.links) uf[kv.first] = kv.first;
    std::function<std::string(const std::string &)> find = [&](const std::string &x) -> std::string {
        std::string r = x;
        while (uf[r] != r) r = uf[r];
        std::string c = x;
        while (uf[c] != r) { std::string n = uf[c]; uf[c] = r; c = n; }
        return r;
    };
    struct CInfo { std::string nca; std::vector<std::string> pred_chain, succ_chain; /* nca -> link, excluding nca */ };
    std::vector<CInfo> cinfo(um.constraints.size());
    for (size_t i = 0; i < um.constraints.size(); i++) {
        const UConstraint &c = um.constraints[i];
        if (!um.links.count(c.pred) || !um.links.count(c.succ)) { err = "constraint " + c.name + ": unknown link"; return GRBDA_EPARSE; }
        std::vector<std::string> pc = chain_to_root(c.pred), sc = chain_to_root(c.succ);
        std::set<std::string> pset(pc.begin(), pc.end());
        std::string nca;
        for (const auto &x : sc) if (pset.count(x)) { nca = x; break; }
        if (nca.empty()) { err = "constraint " + c.name + ": no common ancestor"; return GRBDA_EPARSE; }
        CInfo &ci = cinfo[i];
        ci.nca = nca;
        for (const auto &x : pc) { if (x == nca) break; ci.pred_chain.push_back(x); }
        for (const auto &x : sc) { if (x == nca) break; ci.succ_chain.push_back(x); }
        std::reverse(ci.pred_chain.begin(), ci.pred_chain.end());
        std::reverse(ci.succ_chain.begin(), ci.succ_chain.end());
        std::vector<std::string> all = ci.pred_chain;
        all.insert(all.end(), ci.succ_chain.begin(), ci.succ_chain.end());
        if (all.empty()) { err = "constraint " + c.name + " constrains nothing"; return GRBDA_EPARSE; }
        for (size_t k = 1; k < all.size(); k++) uf[find(all[k])] = find(all[0]);
    }
    std::map<std::string, std::vector<std::string>> members;  // cluster representative -> links (ascending name)
    for (auto &kv : um.links) members[find(kv.first)].push_back(kv.first);
    if (members[find(root)].size() != 1) { err = "The root cluster may only contain one body"; return GRBDA_EPARSE; }

    // cluster order.  The reference walks urdf::Cluster::child_clusters depth-first (ClusterTreeParsing.cpp:20-43); the
    // fork fills them from a strongly-connected-components pass over the link graph whose neighbour lists are the
    // child links in ASCENDING parent-joint name followed by the loop links (UnitTests/testUrdfParser.cpp:268-337,
    // "the order of the neighbors matters"): components come out leaves first, the model lists them root first.
    // Net effect: pre-order over the cluster tree, the children of a cluster in REVERSE order of their first
    // discovery by that link-level walk.  Pinned by the hand-built robots (UnitTests/testClusterTreeModel.cpp:100-114
    // compares cluster j of the URDF model with cluster j of MiniCheetah.cpp:29 -- {HR, HL, FR, FL} -- and
    // MIT_Humanoid.cpp:351-354 -- right arm, right leg, left arm, left leg).
    std::vector<std::string> order;  // representatives, root cluster excluded
    {
        std::map<std::string, std::vector<const UJoint *>> child_joints;  // parent link -> joints, ascending name
        for (auto &kv : um.joints) child_joints[kv.second.parent].push_back(&kv.second);  // std::map iterates by name
        std::map<std::string, std::vector<std::string>> loop_links;
        for (size_t i = 0; i < um.constraints.size(); i++) {
            const UConstraint &c = um.constraints[i];
            loop_links[c.pred].push_back(c.succ);
            loop_links[c.succ].push_back(cinfo[i].pred_chain.empty() ? c.pred : cinfo[i].pred_chain.front());
        }
        std::map<std::string, int> discovered;  // cluster representative -> time of first visit
        std::set<std::string> seen;
        int clock = 0;
        std::function<void(const std::string &)> walk = [&](const std::string &l) {
            seen.insert(l);
            if (!discovered.count(find(l))) discovered[find(l)] = clock++;
            for (const UJoint *j : child_joints[l])
                if (!seen.count(j->child)) walk(j->child);
            for (const std::string &n : loop_links[l])
                if (!seen.count(n)) walk(n);
        };
        walk(root);
        if (discovered.size() != members.size()) { err = "URDF contains links that are not connected to the root"; return GRBDA_EPARSE; }
        std::map<std::string, std::vector<std::string>> kids;  // parent cluster -> child clusters
        for (auto &kv : members) {
            if (kv.first == find(root)) continue;
            std::string parent_cluster;
            for (const std::string &l : kv.second) {
                const std::string pc = find(parent_of(l));
                if (pc != kv.first) parent_cluster = pc;
            }
            kids[parent_cluster].push_back(kv.first);
        }
        std::function<void(const std::string &)> dfs = [&](const std::string &rep) {
            std::vector<std::string> &k = kids[rep];
            std::sort(k.begin(), k.end(), [&](const std::string &a, const std::string &b) { return discovered[a] > discovered[b]; });
            for (const std::string &c : k) {
                order.push_back(c);
                dfs(c);
            }
        };
        dfs(find(root));
        if (order.size() + 1 != members.size()) { err = "URDF contains links that are not connected to the root"; return GRBDA_EPARSE; }
    }

    grbda::desc::ModelDescription md;
    md.ori_repr = ori_repr == 1 ? GRBDA_ORI_RPY : GRBDA_ORI_QUATERNION;
    std::map<std::string, int> body_index;  // link -> global body index
    body_index[root] = -1;
    int next_body = 0;
    try {
        for (const std::string &rep : order) {
            const std::vector<std::string> &links = members[rep];
            grbda::desc::ClusterDesc cd;
            cd.name = links.size() == 1 ? links[0] : "cluster-" + std::to_string(md.numClusters());
            // body registration: multi-pass, parent first (ClusterTreeParsing.cpp:260-307)
            std::vector<std::string> pending = links, reg;
            std::map<std::string, int> sub;
            while (!pending.empty()) {
                std::vector<std::string> next;
                for (const auto &l : pending) {
                    if (!body_index.count(parent_of(l))) { next.push_back(l); continue; }
                    body_index[l] = next_body++;
                    sub[l] = static_cast<int>(reg.size());
                    reg.push_back(l);
                }
                if (next.size() == pending.size()) { err = "cluster " + cd.name + ": parent link outside the registered tree"; return GRBDA_EPARSE; }
                pending = next;
            }
            std::vector<bool> independent;
            for (const auto &l : reg) {
                const ULink &lk = um.links.at(l);
                const UJoint &j = um.joints.at(lk.parent_joint);
                grbda::desc::BodyDesc b;
                b.name = l;
                b.parent = body_index.at(j.parent);
                pose_to_transform(j.origin, b.E.data(), b.r.data());
                spatial_inertia(lk, b.inertia.data());
                if (j.type == "floating") {
                    b.joint_type = GRBDA_JOINT_FREE;
                } else if (j.type == "revolute" || j.type == "continuous") {
                    b.joint_type = GRBDA_JOINT_REVOLUTE;
                    b.axis = coordinate_axis(j.axis, err);
                    if (b.axis < 0) { err = "joint " + j.name + ": " + err; return GRBDA_EPARSE; }
                } else {
                    err = "joint " + j.name + ": type '" + j.type + "' is not supported (revolute, continuous or floating)";
                    return GRBDA_EPARSE;
                }
                independent.push_back(j.independent);
                cd.bodies.push_back(b);
            }
            const int k = static_cast<int>(reg.size());
            if (k == 1) {
                // ClusterTreeParsing.cpp:56-76
                if (cd.bodies[0].joint_type == GRBDA_JOINT_FREE) {
                    if (md.numClusters() > 0) { err = "Floating joint must be the first joint in the system"; return GRBDA_EPARSE; }
                    const int npos = md.ori_repr == GRBDA_ORI_QUATERNION ? 7 : 6;
                    cd.n_pos = cd.n_span_pos = npos;
                    cd.n_vel = cd.n_span_vel = 6;
                    cd.constraint_type = GRBDA_CONSTRAINT_FREE;
                } else {
                    cd.n_pos = cd.n_vel = cd.n_span_pos = cd.n_span_vel = 1;
                    cd.constraint_type = GRBDA_CONSTRAINT_STATIC;
                    cd.dbls = {1.0};  // G = [1], K empty (RevoluteJoint.cpp:19-22)
                }
                md.appendCluster(cd);
                continue;
            }
            for (const auto &b : cd.bodies)
                if (b.joint_type != GRBDA_JOINT_REVOLUTE) { err = "cluster " + cd.name + ": only revolute joints may be constrained"; return GRBDA_EPARSE; }
            // constraints of this cluster (document order)
            std::vector<size_t> cons;
            for (size_t i = 0; i < um.constraints.size(); i++) {
                const CInfo &ci = cinfo[i];
                const std::string &any = !ci.pred_chain.empty() ? ci.pred_chain[0] : ci.succ_chain[0];
                if (find(any) == rep) cons.push_back(i);
            }
            if (cons.empty()) { err = "Cluster must have at least one constraint"; return GRBDA_EPARSE; }
            for (size_t i : cons)
                if (um.constraints[i].is_loop != um.constraints[cons[0]].is_loop) { err = "All constraints in cluster must be of same class type"; return GRBDA_EPARSE; }
            int n_ind = 0;
            for (bool b : independent) n_ind += b ? 1 : 0;
            cd.n_span_pos = cd.n_span_vel = k;
            cd.n_vel = n_ind;

            if (!um.constraints[cons[0]].is_loop) {
                // explicitRollingConstraint (ClusterTreeParsing.cpp:378-440)
                const int rows = static_cast<int>(cons.size());
                if (rows != k - n_ind) { err = "cluster " + cd.name + ": number of couplings != number of dependent joints"; return GRBDA_EPARSE; }
                std::vector<double> K(static_cast<size_t>(rows) * k, 0.0);
                for (int r = 0; r < rows; r++) {
                    const CInfo &ci = cinfo[cons[r]];
                    for (const auto &l : ci.pred_chain) K[r * k + sub.at(l)] = um.constraints[cons[r]].ratio;
                    for (const auto &l : ci.succ_chain) K[r * k + sub.at(l)] = -1.0;
                }
                std::vector<int> ind, dep;
                for (int i = 0; i < k; i++) (independent[i] ? ind : dep).push_back(i);
                const int nd = static_cast<int>(dep.size());
                // X = Kd^-1 Ki by Gauss-Jordan with partial pivoting
                std::vector<double> A(static_cast<size_t>(nd) * (nd + n_ind));
                for (int r = 0; r < nd; r++) {
                    for (int j = 0; j < nd; j++) A[r * (nd + n_ind) + j] = K[r * k + dep[j]];
                    for (int j = 0; j < n_ind; j++) A[r * (nd + n_ind) + nd + j] = K[r * k + ind[j]];
                }
                const int W = nd + n_ind;
                for (int c = 0; c < nd; c++) {
                    int p = c;
                    for (int r = c + 1; r < nd; r++) if (std::fabs(A[r * W + c]) > std::fabs(A[p * W + c])) p = r;
                    if (std::fabs(A[p * W + c]) < 1e-14) { err = "cluster " + cd.name + ": dependent-coordinate block of K is singular"; return GRBDA_EPARSE; }
                    if (p != c) for (int j = 0; j < W; j++) std::swap(A[c * W + j], A[p * W + j]);
                    const double piv = A[c * W + c];
                    for (int j = 0; j < W; j++) A[c * W + j] /= piv;
                    for (int r = 0; r < nd; r++) {
                        if (r == c) continue;
                        const double f = A[r * W + c];
                        if (f != 0.0) for (int j = 0; j < W; j++) A[r * W + j] -= f * A[c * W + j];
                    }
                }
                std::vector<double> G(static_cast<size_t>(k) * n_ind, 0.0);
                for (int j = 0; j < n_ind; j++) G[ind[j] * n_ind + j] = 1.0;
                for (int r = 0; r < nd; r++)
                    for (int j = 0; j < n_ind; j++) G[dep[r] * n_ind + j] = -A[r * W + nd + j];
                cd.constraint_type = GRBDA_CONSTRAINT_STATIC;
                cd.n_pos = n_ind;
                cd.n_rows = rows;
                cd.dbls = G;
                cd.dbls.insert(cd.dbls.end(), K.begin(), K.end());
            } else {
                // implicitPositionConstraint (ClusterTreeParsing.cpp:310-376)
                cd.constraint_type = GRBDA_CONSTRAINT_LOOP_POSITION;
                cd.n_pos = k;  // spanning positions (GenericJoint.cpp:246-249)
                cd.ints.push_back(static_cast<int32_t>(cons.size()));
                for (bool b : independent) cd.ints.push_back(b ? 1 : 0);
                int rows = 0;
                for (size_t ci_idx : cons) {
                    const CInfo &ci = cinfo[ci_idx];
                    const UConstraint &uc = um.constraints[ci_idx];
                    double oE[2][9], orr[2][3];
                    pose_to_transform(uc.pred_origin, oE[0], orr[0]);
                    pose_to_transform(uc.succ_origin, oE[1], orr[1]);
                    // axes of r_pred - r_succ that depend on q (which_depends in the reference):
                    // probe the translation at a few joint configurations
                    auto endpoint = [&](const std::vector<std::string> &chain, const double *E0, const double *r0,
                                        const std::vector<double> &q, double *out) {
                        double E[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, r[3] = {0, 0, 0};
                        auto compose = [&](const double *Ea, const double *ra) {  // X = A * X : E = Ea E, r = r + E^T ra
                            double t[3];
                            for (int i = 0; i < 3; i++) t[i] = E[i] * ra[0] + E[3 + i] * ra[1] + E[6 + i] * ra[2];
                            Mat3 En = mul3(Ea, E);
                            std::memcpy(E, En.m, sizeof E);
                            for (int i = 0; i < 3; i++) r[i] += t[i];
                        };
                        for (const auto &l : chain) {
                            const grbda::desc::BodyDesc &b = cd.bodies[sub.at(l)];
                            double R[9];
                            coord_rot(b.axis, q[sub.at(l)], R);
                            Mat3 EJ = mul3(R, b.E.data());
                            compose(EJ.m, b.r.data());
                        }
                        compose(E0, r0);
                        std::memcpy(out, r, sizeof(double) * 3);
                    };
                    int mask = 0;
                    std::vector<double> q0(k, 0.0);
                    double base[3], pb[3], sb[3];
                    endpoint(ci.pred_chain, oE[0], orr[0], q0, pb);
                    endpoint(ci.succ_chain, oE[1], orr[1], q0, sb);
                    for (int a = 0; a < 3; a++) base[a] = pb[a] - sb[a];
                    unsigned lcg = 12345u;
                    for (int trial = 0; trial < 8; trial++) {
                        std::vector<double> q(k);
                        for (int i = 0; i < k; i++) { lcg = lcg * 1664525u + 1013904223u; q[i] = (static_cast<double>(lcg >> 8) / 8388608.0 - 1.0) * 1.3; }
                        endpoint(ci.pred_chain, oE[0], orr[0], q, pb);
                        endpoint(ci.succ_chain, oE[1], orr[1], q, sb);
                        for (int a = 0; a < 3; a++)
                            if (std::fabs((pb[a] - sb[a]) - base[a]) > 1e-9) mask |= 1 << a;
                    }
                    cd.ints.push_back(static_cast<int32_t>(ci.pred_chain.size()));
                    for (const auto &l : ci.pred_chain) cd.ints.push_back(sub.at(l));
                    cd.ints.push_back(static_cast<int32_t>(ci.succ_chain.size()));
                    for (const auto &l : ci.succ_chain) cd.ints.push_back(sub.at(l));
                    cd.ints.push_back(mask);
                    for (int s = 0; s < 2; s++) {
                        cd.dbls.insert(cd.dbls.end(), oE[s], oE[s] + 9);
                        cd.dbls.insert(cd.dbls.end(), orr[s], orr[s] + 3);
                    }
                    for (int a = 0; a < 3; a++) rows += (mask >> a) & 1;
                }
                if (rows != k - n_ind) { err = "cluster " + cd.name + ": number of loop-constraint rows != number of dependent joints"; return GRBDA_EPARSE; }
                cd.n_rows = rows;
            }
            md.appendCluster(cd);
        }
    } catch (const std::exception &e) {
        err = e.what();
        return GRBDA_EPARSE;
    }
    if (md.numClusters() == 0) { err = "URDF has no movable link"; return GRBDA_EPARSE; }
    blob = md.serialize();
    return GRBDA_OK;
}

}  // namespace grbda_hip
