"""CPU tests of the C ABI: the library loads, exports every symbol include/grbda_hip.h declares,
compiles plans on the host, reports errors by code, and refuses to compute without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import generalized_rbda_amd as G
from generalized_rbda_amd import modeldesc as md
from models import zoo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "grbda_hip.h")).read()
    declared = set(re.findall(r"\b(grbda_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"grbda_plan"}
    assert declared == set(G.C_ABI_SYMBOLS), declared ^ set(G.C_ABI_SYMBOLS)
    L = ctypes.CDLL(G.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name


@pytest.mark.parametrize("name,blob", list(zoo().items()), ids=list(zoo().keys()))
def test_plan_compiles_on_host(name, blob):
    p = G.Plan(blob)
    from generalized_rbda_amd.states import parse_clusters
    m = parse_clusters(blob)
    assert (p.nq, p.nv, p.n_bodies, p.n_clusters) == (m["nq"], m["nv"], m["nb"], m["nc"])
    assert p.blob == blob
    info = p.info()
    assert info.n_slots > 0 and info.flops_aba > 0 and info.bytes_aba_f32 == (p.nq + 3 * p.nv) * 4


def test_gravity_roundtrip():
    p = G.Plan(md.revolute_chain_with_rotor(3).serialize())
    assert p.get_gravity() == [9.81, 0.0, 0.0]
    p.set_gravity([0, 0, -9.81])
    assert p.get_gravity() == [0.0, 0.0, -9.81]


def test_bad_blob_is_rejected():
    with pytest.raises(G.GrbdaError) as e:
        G.Plan(b"\0" * 200)
    assert e.value.code == -1
    blob = bytearray(md.revolute_chain_with_rotor(2).serialize())
    with pytest.raises(G.GrbdaError):
        G.Plan(bytes(blob[:300]))


def two_parent_model():
    """A cluster whose two bodies hang off DIFFERENT bodies of the parent cluster (per-body parent_subindex of the reference's
    GeneralizedTransform, SpatialTransforms.cpp:312-344,415-477): the projected inertia of cluster c couples b1 and b2."""
    m = md.ClusterTreeModel()
    rng = np.random.default_rng(12)
    I = lambda: md.spatial_inertia(rng.uniform(0.5, 2.0), rng.uniform(-0.2, 0.2, 3), np.eye(3) * rng.uniform(0.02, 0.1))
    X = lambda: (md.rpy_to_rotmat(rng.uniform(-1, 1, 3)), rng.uniform(-0.4, 0.4, 3))
    m.appendBody("a", I(), "ground", *X(), joint="revolute", axis="z")
    m.registerBody("b1", I(), "a", *X())
    m.registerBody("b2", I(), "a", *X())
    m.appendRegisteredBodiesAsCluster("b", "Generic", axes="zy", G=[[1.0], [2.0]], K=[[2.0, -1.0]])
    m.registerBody("c1", I(), "b1", *X())
    m.registerBody("c2", I(), "b2", *X())
    m.appendRegisteredBodiesAsCluster("c", "Generic", axes="xz", G=[[1.0], [-1.5]], K=[[1.5, 1.0]])
    m.appendBody("d", I(), "c2", *X(), joint="revolute", axis="y")
    return m


def test_cluster_with_two_parent_bodies_takes_the_spanning_tree_route(monkeypatch):
    """Rounds 1-3 refused such models (GRBDA_EUNSUPPORTED); they now compile to a plan whose entry points run through the
    spanning tree (plan.h, HostPlan::projection_only) -- GPU parity in tests/test_gpu_parity.py.  The oracle's dense 6k x 6k
    formulation covers the shape directly and agrees with its own Projection-method restatement."""
    import oracle_py as O

    blob = two_parent_model().serialize()
    info = G.Plan(blob).info()
    assert info.spanning_tree_route == 1 and info.analytic_derivatives == 1 and info.chain_aba_f32 == 0
    rng = np.random.default_rng(3)
    q, qd, tau = rng.uniform(-1, 1, (20, 4)), rng.uniform(-1, 1, (20, 4)), rng.uniform(-1, 1, (20, 4))
    a = O.forward_dynamics(blob, q, qd, tau)
    b = O.forward_dynamics_projection(blob, q, qd, tau)
    assert np.abs(a - b).max() / (1 + np.abs(a).max()) < 5e-8
    monkeypatch.setenv("GRBDA_NO_PROJECTION", "1")  # (A/B switch: the old refusal)
    with pytest.raises(G.GrbdaError) as e:
        G.Plan(blob)
    assert e.value.code == -2 and "more than one body" in str(e.value)


def test_no_cpu_fallback():
    """Without a HIP device every compute entry point must fail loudly (GRBDA_ENODEVICE)."""
    if G.device_count() > 0:
        pytest.skip("a HIP device is present")
    p = G.Plan(md.revolute_chain_with_rotor(2).serialize())
    z = np.zeros((1, 2))
    with pytest.raises(G.GrbdaError) as e:
        p.forward_dynamics_host(z, z, z)
    assert e.value.code == -3
    with pytest.raises(G.GrbdaError) as e:
        p.inverse_dynamics_host(z, z, z)
    assert e.value.code == -3


def test_device_resident_sharded_entry_points_check_their_arguments():
    """grbda_{aba,rnea}_sharded_dev_*: argument checking happens before anything touches a device (null plan, n_gpus < 1, null
    pointer arrays -> GRBDA_EINVAL = -1); with well-formed arguments and no HIP device the answer is GRBDA_ENODEVICE = -3."""
    import ctypes

    L = G.lib()
    p = G.Plan(md.revolute_chain_with_rotor(2).serialize())
    P = ctypes.c_void_p
    one = (P * 1)(0x1000)   # (never dereferenced on these paths)
    Bs = (ctypes.c_size_t * 1)(4)
    for name in ("grbda_aba_sharded_dev_f32", "grbda_aba_sharded_dev_f64", "grbda_rnea_sharded_dev_f32", "grbda_rnea_sharded_dev_f64"):
        fn = getattr(L, name)
        assert fn(None, 1, None, one, one, one, one, Bs, None, None) == -1            # null plan
        assert fn(p._h, 0, None, one, one, one, one, Bs, None, None) == -1            # n_gpus < 1
        assert fn(p._h, 1, None, None, one, one, one, Bs, None, None) == -1           # null q array
        assert fn(p._h, 1, None, one, one, one, None, Bs, None, None) == -1           # neither per-shard outputs nor a gathered array
        assert fn(p._h, 1, None, one, one, one, one, None, None, None) == -1          # null batch sizes
        if G.device_count() == 0:
            assert fn(p._h, 1, None, one, one, one, one, Bs, None, None) == -3        # no device: no CPU fallback
        else:
            bad_dev = (ctypes.c_int * 1)(G.device_count())
            assert fn(p._h, 1, bad_dev, one, one, one, one, Bs, None, None) == -1     # device index out of range


def test_model_builder_rejects_invalid_topology():
    m = md.ClusterTreeModel()
    I = md.spatial_inertia(1.0, [0, 0, 0], np.eye(3))
    m.appendBody("a", I, "ground", joint="revolute", axis="x")
    m.appendBody("b", I, "ground", joint="revolute", axis="x")
    m.registerBody("c1", I, "a")
    m.registerBody("c2", I, "b")
    with pytest.raises(RuntimeError, match="same parent cluster"):
        m.appendRegisteredBodiesAsCluster("c", "Generic", axes="xx", G=[[1.0], [1.0]], K=[[1.0, -1.0]])
    with pytest.raises(RuntimeError, match="Free joint"):
        m2 = md.ClusterTreeModel()
        m2.appendBody("a", I, "ground", joint="revolute", axis="x")
        m2.appendBody("f", I, "a", joint="free")


def test_chain_program_covers_the_headline_models():
    """The chain-structured fast path (plan.h, ChainProgram) must be what runs the floating-base robots made of
    revolute links, geared rotors and leaf pair clusters -- the BASELINE workloads mini_cheetah / mit_humanoid -- and
    the random models of the zoo that exercise its run / branch / pair / roll-pitch-yaw code -- and TelloWithArms, whose
    implicit differentials are a segment type of their own; every other cluster type is a generic segment (ChainGen)."""
    z = zoo()
    for name in ("urdf_mini_cheetah", "urdf_mit_humanoid", "urdf_jvrc1_humanoid", "urdf_mini_cheetah_rpy", "tree_chain_rotor_float", "tree_chain_rev_float", "tree_rotor_float",
                 "chain_tree_a", "chain_tree_b", "chain_tree_rpy", "chain_tree_norotor", "tello_with_arms"):
        info = G.Plan(z[name]).info()
        assert info.chain_aba_f32 == 1 and info.n_chain_segments > 0, name
    for name in ("urdf_mini_cheetah", "urdf_mit_humanoid", "chain_tree_b", "chain_tree_norotor"):
        assert G.Plan(z[name]).info().chain_aba_f64 == 1, name
    ti = G.Plan(z["tello_with_arms"]).info()
    assert ti.chain_rnea_f32 == 1 and ti.n_chain_differentials == 4
    # analytic derivatives: explicit clusters by the recursion of deriv_kernels.hip, implicit ones on the constraint manifold through
    # the spanning tree (manifold_kernels.hip) -- nothing of the zoo takes differences
    for name in ("urdf_jvrc1_humanoid", "urdf_mit_humanoid", "tree_mixed_fixed", "tree_generic_float", "tello_with_arms", "tello", "urdf_four_bar",
                 "urdf_six_bar", "urdf_planar_leg_linkage", "urdf_mini_cheetah_rpy"):
        assert G.Plan(z[name]).info().analytic_derivatives == 1, name
    # fixed-base chains of links (the reference's RevoluteChainWithRotor family, config 1's URDF) start their runs on the ground
    for name in ("rev_rotor_chain_3", "urdf_revolute_rotor_chain", "tree_rev_fixed", "rev_pair_rotor_chain_4"):
        info = G.Plan(z[name]).info()
        assert info.chain_aba_f32 == 1 and info.chain_rnea_f32 == 1 and info.chain_aba_f64 == 1, name
    # (pairs in series -- the reference's RevolutePairChainWithRotor -- run through the differential's segments with constant G)
    assert G.Plan(z["rev_pair_rotor_chain_4"]).info().n_chain_differentials == 2
    # generic clusters (plan.h, ChainGen): URDF+ position loops, RevoluteTripleWithRotor, Generic + Static clusters run inside
    # chain programs too -- nothing of the zoo is left on the interpreter for the fp32 forward dynamics
    for name in ("urdf_four_bar", "urdf_six_bar", "urdf_planar_leg_linkage", "teleop_arm", "tree_generic_float", "tree_triple_fixed",
                 "tree_mixed_float", "tree_mixed_fixed", "rev_triple_rotor_chain_6"):
        info = G.Plan(z[name]).info()
        assert info.chain_aba_f32 == 1 and info.n_chain_generic >= 1, name
    for name, blob in z.items():
        assert G.Plan(blob).info().chain_aba_f32 == 1, name


def test_parallel_chain_generator_reproduces_the_reference_files(tmp_path):
    """tests/parallel_chains.py writes the reference's parallel-chain family for any (depth, loop size); for the two files of that
    family kept under tests/golden/robot-models (the reference's own depth-10 models) it gives the same model description."""
    from models import ROBOT_MODELS
    from parallel_chains import parallel_chain_urdf

    for implicit, loop, name in ((False, 16, "parallel_chain_exp_d10_l16"), (True, 17, "parallel_chain_imp_d10_l17")):
        path = tmp_path / (name + ".urdf")
        path.write_text(parallel_chain_urdf(10, loop, implicit))
        assert G.urdf_to_blob(str(path)) == G.urdf_to_blob(os.path.join(ROBOT_MODELS, name + ".urdf"))


@pytest.mark.parametrize("implicit,depth,loop", [(False, 10, 16), (True, 10, 17), (False, 20, 30), (True, 20, 31), (False, 40, 40), (True, 40, 41)])
def test_clusters_beyond_the_structured_limits_compile_to_the_spanning_tree_route(tmp_path, monkeypatch, implicit, depth, loop):
    """Clusters of more than 8 bodies / 4 independent coordinates (asked for since round 1; the reference's parallel-chain benchmark
    family reaches 41 bodies) compile to a plan on the spanning-tree route (HostPlan::big_clusters).  The checker for them is the
    oracle built with room for 48 bodies per cluster; its cluster recursion agrees with its own Projection-method restatement and
    inverts its inverse dynamics."""
    import oracle_py as O
    from models import valid_states
    from parallel_chains import parallel_chain_urdf

    path = tmp_path / "pc.urdf"
    path.write_text(parallel_chain_urdf(depth, loop, implicit))
    plan = G.Plan.from_urdf(str(path))
    assert plan.info().spanning_tree_route == 1
    assert plan.n_bodies == 2 * depth + (1 if implicit else 0) and plan.nv == 2 * depth - 1
    blob = plan.blob
    q, qd, tau = valid_states(blob, 12, config_index=5, big=True, scale=0.5 if depth < 40 else 0.25)
    a = O.forward_dynamics(blob, q, qd, tau, big=True)
    b = O.forward_dynamics_projection(blob, q, qd, tau, big=True)
    assert np.abs(a - b).max() / (1 + np.abs(a).max()) < 5e-8
    assert np.abs(O.inverse_dynamics(blob, q, qd, a, big=True) - tau).max() / (1 + np.abs(a).max()) < 5e-8
    with pytest.raises(RuntimeError):  # (the default build of the checker has no room for them)
        O.forward_dynamics(blob, q, qd, tau)
    monkeypatch.setenv("GRBDA_NO_PROJECTION", "1")  # (A/B switch: the old refusal)
    with pytest.raises(G.GrbdaError) as e:
        G.Plan(blob)
    assert e.value.code == -2 and "exceed the kernel limit" in str(e.value)
