"""URDF+ text of the reference's parallel-chain benchmark family (Benchmarking/urdfs/parallel_chains: two chains of `depth`
links from a common base, closed into ONE loop cluster of `loop_size` bodies; pinocchioHelpers.cpp:355-410 lists the sizes).
  explicit: joint_2_m is coupled to joint_1_m with ratio 1, m = loop_size / 2  ->  a cluster of 2 m bodies, 2 m - 1 DoF
  implicit: a connecting rod from link_1_m closes on link_2_m with a planar revolute loop joint, m = (loop_size - 1) / 2
            ->  a cluster of 2 m + 1 bodies, two constraint rows, 2 m - 1 DoF
The text is generated here for any (depth, loop size); tests/golden/robot-models holds the reference's own depth-10 files
(parallel_chain_exp_d10_l16.urdf, parallel_chain_imp_d10_l17.urdf) and test_capi_cpu.py checks that both give the same model."""
import numpy as np

LINK = ('  <link name="{name}">\n    <inertial>\n      <mass value="0.25"/>\n      <origin rpy="0 0 0" xyz="0 0.5 0"/>\n'
        '      <inertia ixx="0.01" ixy="0" ixz="0" iyy="0.1" iyz="0" izz="0.01"/>\n    </inertial>\n  </link>\n')
JOINT = ('  <joint independent="{ind}" name="{name}" type="revolute">\n    <parent link="{parent}"/>\n    <child link="{child}"/>\n'
         '    <origin xyz="{xyz}"/>\n    <axis xyz="0 0 1"/>\n    <limit effort="30" lower="-10" upper="10" velocity="1.0"/>\n  </joint>\n')


def parallel_chain_urdf(depth: int, loop_size: int, implicit: bool) -> str:
    m = (loop_size - 1) // 2 if implicit else loop_size // 2
    if m < 1 or m > depth or loop_size != (2 * m + 1 if implicit else 2 * m):
        raise ValueError("loop size does not fit the family")
    out = [f'<?xml version="1.0" ?>\n<robot name="parallel_chain_{"imp" if implicit else "exp"}_d{depth}_l{loop_size}">\n  <link name="base"/>\n']
    for i in range(1, depth + 1):
        for chain in (1, 2):
            out.append(LINK.format(name=f"link_{chain}_{i}"))
            if implicit:
                ind = not (chain == 2 and i == 1)
            else:
                ind = not (chain == 2 and i == m)
            parent = "base" if i == 1 else f"link_{chain}_{i - 1}"
            xyz = ("0 0 0" if chain == 1 else ("1.0 0 0" if implicit else "0 0 0")) if i == 1 else "0 1.0 0"
            out.append(JOINT.format(ind="true" if ind else "false", name=f"joint_{chain}_{i}", parent=parent, child=f"link_{chain}_{i}", xyz=xyz))
        if i == m:
            if implicit:
                out.append(LINK.format(name="connecting_rod"))
                out.append(JOINT.format(ind="false", name="connecting_rod_joint", parent=f"link_1_{i}", child="connecting_rod", xyz="0 1.0 0"))
                out.append(f'  <loop name="position_{i}" type="revolute">\n    <predecessor link="connecting_rod">\n      <origin xyz="1.0 0.0 0.0"/>\n'
                           f'    </predecessor>\n    <successor link="link_2_{i}">\n      <origin xyz="0.0 1.0 0.0"/>\n    </successor>\n'
                           '    <axis xyz="0 0 1"/>\n  </loop>\n')
            else:
                out.append(f'  <coupling name="coupling_{i}">\n    <predecessor link="link_1_{i}"/>\n    <successor link="link_2_{i}"/>\n'
                           '    <ratio value="1"/>\n  </coupling>\n')
    out.append("</robot>\n")
    return "".join(out)


def parallel_chain_states(plan, B: int, seed: int, implicit: bool):
    """Random states of a parallel-chain plan: independent coordinates drawn near the parallelogram configuration (chain 2 follows
    chain 1 up to a small perturbation), the dependent ones of an implicit cluster by the ORACLE's Newton projection from the
    parallelogram's closed form.  Returns q [B, nq], qd [B, nv], tau [B, nv] (float64) and the rows that converged."""
    from generalized_rbda_amd.states import parse_clusters
    rng = np.random.default_rng(seed)
    nq, nv = plan.nq, plan.nv
    q = np.zeros((B, nq))
    names = plan.body_names() if hasattr(plan, "body_names") else None
    clusters = parse_clusters(plan.blob)
    # angles per chain level: chain 1 free, chain 2 = chain 1 + perturbation
    for c in clusters:
        k = c["n_bodies"]
        if c["n_pos"] == c["n_vel"]:  # explicit or single joint: independent coordinates
            q[:, c["q_index"]:c["q_index"] + c["n_pos"]] = rng.uniform(-0.6, 0.6, (B, c["n_pos"]))
        else:
            q[:, c["q_index"]:c["q_index"] + k] = 0.0
    return q, clusters
