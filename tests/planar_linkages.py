"""Planar linkages with SEVERAL position loops that share links, for the tests of clusters with more than three constraint rows
(the reference's own loop fixtures -- four_bar.urdf, six_bar.urdf, the planar leg linkage -- have one loop, two rows).

A parallelogram four-bar (crank link1, coupler link2, rocker link3 on the ground link) carries `n_dyads` two-link dyads, each from a
point of the coupler to a point of the rocker: one cluster of 3 + 2 n_dyads bodies, ONE independent coordinate (the crank angle) and
2 + 2 n_dyads constraint rows.  The dimensions keep every dyad inside its reach for |crank angle| < 1 rad.

`python tests/planar_linkages.py` rewrites tests/golden/robot-models/watt_six_bar.urdf (one dyad, 4 rows) and
three_loop_linkage.urdf (two dyads, 6 rows)."""
import os

# (attachment on the coupler, first link's length, attachment on the rocker, second link's length)
DYADS = [(0.5, 0.4, 0.25, 0.35), (0.25, 0.45, 0.4, 0.4), (0.75, 0.3, 0.1, 0.25)]


def linkage_urdf(n_dyads, name="planar_linkage"):
    assert 0 <= n_dyads <= len(DYADS)
    links = [("link1", 1.1, 0.25, 0.024), ("link2", 2.3, 0.5, 0.19), ("link3", 1.4, 0.25, 0.03)]
    joints = [("joint1", "base_link", "link1", 0.0, "true"), ("joint2", "link1", "link2", 0.5, "false"),
              ("joint3", "base_link", "link3", 1.0, "false")]
    loops = [("constraint1", "link2", 1.0, "link3", 0.5)]
    for d, (s, la, t, lb) in enumerate(DYADS[:n_dyads]):
        a, b = f"link{4 + 2 * d}", f"link{5 + 2 * d}"
        links += [(a, 0.6 - 0.1 * d, 0.5 * la, 0.009), (b, 0.5 + 0.05 * d, 0.5 * lb, 0.006)]
        joints += [(f"joint{4 + 2 * d}", "link2", a, s, "false"), (f"joint{5 + 2 * d}", "link3", b, t, "false")]
        loops.append((f"constraint{2 + d}", a, la, b, lb))
    out = ['<?xml version="1.0" ?>\n',
           f"<!-- written by tests/planar_linkages.py: a parallelogram four-bar with {n_dyads} dyad(s) between coupler and rocker;\n"
           f"     one cluster of {len(links)} bodies, one independent coordinate, {2 * len(loops)} constraint rows -->\n",
           f'<robot name="{name}">\n\n    <link name="base_link"/>\n']
    for n, m, c, i in links:
        out.append(f'\n    <link name="{n}">\n        <inertial>\n            <mass value="{m:.6g}"/>\n'
                   f'            <origin xyz="{c:.6g} 0.0 0.0"/>\n'
                   f'            <inertia ixx="{0.1 * i:.6g}" ixy="0" ixz="0" iyy="{0.9 * i:.6g}" iyz="0" izz="{i:.6g}"/>\n'
                   f'        </inertial>\n    </link>\n')
    for n, p, c, o, ind in joints:
        out.append(f'\n    <joint name="{n}" type="continuous" independent="{ind}">\n        <parent link="{p}"/>\n'
                   f'        <child link="{c}"/>\n        <origin xyz="{o:.6g} 0.0 0.0"/>\n        <axis xyz="0 0 1"/>\n    </joint>\n')
    for n, p, po, c, co in loops:
        out.append(f'\n    <loop name="{n}" type="revolute">\n        <predecessor link="{p}">\n'
                   f'            <origin xyz="{po:.6g} 0.0 0.0"/>\n        </predecessor>\n        <successor link="{c}">\n'
                   f'            <origin xyz="{co:.6g} 0.0 0.0"/>\n        </successor>\n        <axis xyz="0 0 1"/>\n    </loop>\n')
    out.append("\n</robot>\n")
    return "".join(out)


if __name__ == "__main__":
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "robot-models")
    for n_dyads, name in ((1, "watt_six_bar"), (2, "three_loop_linkage")):
        with open(os.path.join(here, name + ".urdf"), "w") as f:
            f.write(linkage_urdf(n_dyads, name))
