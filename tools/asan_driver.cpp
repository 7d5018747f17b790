// AddressSanitizer / UBSan driver of the host-side code (make asan): the URDF+ reader (csrc/urdf.cpp), the plan compiler
// (csrc/plan.cpp) at the launch shapes capi.cpp uses, and the oracle (oracle/grbda_oracle.c) on a few states -- for every
// URDF given on the command line, both base orientations.  Also a truncated and a corrupted blob: the plan compiler and the
// oracle must reject them without reading out of bounds.  Exit code 0 = no finding (the sanitizers abort otherwise).
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../generalized_rbda_amd/csrc/plan.h"
#include "../include/grbda_hip.h"
#include "../include/grbda_model_desc.h"
extern "C" {
#include "../oracle/grbda_oracle.h"
}

namespace grbda_hip {
int urdf_to_blob(const char *const *paths, int n_paths, int ori_repr, std::vector<unsigned char> &blob, std::string &err);
}

static int compile(const std::vector<unsigned char> &blob, grbda_hip::HostPlan &hp, char *msg, size_t cap)
{
    grbda_hip::LdsBudget lds;  // the defaults of capi.cpp: 20 KiB per wavefront f32, 40 KiB f64 ABA; 10 / 20 KiB RNEA
    lds.aba32 = 20480 / (4 * 64);
    lds.aba64 = 40960 / (8 * 64);
    lds.rnea32 = 10240 / (4 * 64);
    lds.rnea64 = 20480 / (8 * 64);
    lds.chain32w = 10240 / (4 * 64);
    return grbda_hip::compile_plan(blob.data(), blob.size(), lds, 7, hp, msg, cap);
}

int main(int argc, char **argv)
{
    int bad = 0;
    char msg[512];
    for (int a = 1; a < argc; a++) {
        for (int ori = 0; ori < 2; ori++) {
            std::vector<unsigned char> blob;
            std::string err;
            const char *paths[1] = {argv[a]};
            int rc = 0;
            const std::string name = argv[a];
            if (name.size() > 5 && name.substr(name.size() - 5) == ".grbd") {  // a serialised model description (e.g. TelloWithArms)
                if (ori == 1) continue;
                FILE *f = std::fopen(argv[a], "rb");
                if (!f) { std::printf("%s: cannot open\n", argv[a]); bad++; continue; }
                unsigned char buf[4096];
                size_t n;
                while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) blob.insert(blob.end(), buf, buf + n);
                std::fclose(f);
            } else {
                rc = grbda_hip::urdf_to_blob(paths, 1, ori, blob, err);
            }
            if (rc) { std::printf("%s: reader %d (%s)\n", argv[a], rc, err.c_str()); bad++; continue; }
            grbda_hip::HostPlan hp;
            msg[0] = 0;
            rc = compile(blob, hp, msg, sizeof msg);
            std::printf("%s ori %d: %zu bytes, plan rc %d %s nq %d nv %d chain %d\n", argv[a], ori, blob.size(), rc, msg, hp.nq, hp.nv,
                        hp.chain32.ok ? 1 : 0);
            if (rc) { bad++; continue; }
            // the oracle on a few states (zero positions are on every explicit model's manifold; implicit models: the
            // oracle's own Newton projection first)
            const int B = 3;
            std::vector<double> q(static_cast<size_t>(B) * hp.nq, 0.0), qd(static_cast<size_t>(B) * hp.nv), tau(qd.size()), ydd(qd.size()), back(qd.size());
            std::mt19937 gen(7);
            std::uniform_real_distribution<double> U(-1, 1);
            for (auto &x : q) x = 0.3 * U(gen);
            for (auto &x : qd) x = U(gen);
            for (auto &x : tau) x = U(gen);
            const grbda_desc_header *h = reinterpret_cast<const grbda_desc_header *>(blob.data());
            if (hp.nq == hp.nv + 1 && ori == 0 && h->n_clusters > 0)
                for (int b = 0; b < B; b++) { double *o = &q[static_cast<size_t>(b) * hp.nq + 3]; o[0] = 1; o[1] = o[2] = o[3] = 0; }
            std::vector<int> ok(B, 0);
            grbda_oracle_project_positions(blob.data(), blob.size(), q.data(), B, 50, ok.data());
            rc = grbda_oracle_forward_dynamics(blob.data(), blob.size(), q.data(), qd.data(), tau.data(), nullptr, ydd.data(), B);
            if (!rc) rc = grbda_oracle_inverse_dynamics(blob.data(), blob.size(), q.data(), qd.data(), ydd.data(), nullptr, back.data(), B);
            double err_rt = 0;
            for (size_t i = 0; i < back.size(); i++)
                if (ok[i / hp.nv]) err_rt = std::max(err_rt, std::fabs(back[i] - tau[i]));
            std::printf("   oracle rc %d, ID(FD(tau)) - tau = %.2e\n", rc, err_rt);
            if (ori == 0) {  // malformed inputs: every prefix length class and a few flipped bytes must be rejected or survive cleanly
                for (size_t cut : {size_t(0), size_t(40), size_t(96), blob.size() / 2, blob.size() - 8}) {
                    std::vector<unsigned char> t(blob.begin(), blob.begin() + static_cast<long>(cut));
                    grbda_hip::HostPlan h2;
                    (void)compile(t, h2, msg, sizeof msg);
                    (void)grbda_oracle_forward_dynamics(t.data(), t.size(), q.data(), qd.data(), tau.data(), nullptr, ydd.data(), 1);
                }
                for (int trial = 0; trial < 200; trial++) {
                    std::vector<unsigned char> t(blob);
                    const size_t at = 96 + (gen() % (t.size() - 96));  // past the header: counts and offsets of bodies / clusters
                    t[at] = static_cast<unsigned char>(gen());
                    grbda_hip::HostPlan h2;
                    (void)compile(t, h2, msg, sizeof msg);
                }
            }
        }
    }
    std::printf(bad ? "FAILED (%d)\n" : "OK\n", bad);
    return bad ? 1 : 0;
}
