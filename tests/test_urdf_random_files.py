"""Random URDF files -- branched, every joint type, coordinate / negative / scaled / skew / missing axes, missing origins -- through rtbhip.urdf and the
chain kernels (CPU replay) against the reference's lowering restated link by link (tests/test_urdf_skew_axes.py: reference_constant) and its
depth-first joint numbering.  The model files the reference ships do not exercise any of: skew axes, scaled axes, branches listed out of depth-first
order, fixed joints with an axis tag."""
import numpy as np

import cpu_backend
from oracle import chains
from rtbhip import urdf
from test_urdf_skew_axes import reference_constant


def test_random_urdf_files_are_lowered_and_numbered_as_the_reference_does():
    paths = 0
    with cpu_backend.installed():
        for seed in range(40):
            rng = np.random.default_rng(seed)
            nl = int(rng.integers(2, 11))
            xml, joints = ['<robot name="r%d">' % seed, '<link name="l0"/>'], {}
            for j in range(1, nl):
                parent = int(rng.integers(max(0, j - 3), j))
                typ = str(rng.choice(["revolute", "continuous", "prismatic", "fixed"], p=[0.45, 0.15, 0.25, 0.15]))
                ax = (np.eye(3)[rng.integers(3)] * rng.choice([-1, 1]) * rng.choice([1.0, 2.5])) if rng.random() < 0.6 else rng.normal(size=3) * rng.uniform(0.3, 3)
                noaxis, noorigin = rng.random() < 0.1, rng.random() < 0.1
                xyz, rpy = rng.uniform(-0.3, 0.3, 3) * (rng.random() > 0.2), rng.uniform(-3, 3, 3) * (rng.random() > 0.3)
                xml.append('<link name="l%d"/>' % j)
                xml.append('<joint name="j%d" type="%s"><parent link="l%d"/><child link="l%d"/>%s%s%s</joint>' % (
                    j, typ, parent, j,
                    "" if noorigin else '<origin xyz="%r %r %r" rpy="%r %r %r"/>' % (*[float(v) for v in xyz], *[float(v) for v in rpy]),
                    "" if noaxis else '<axis xyz="%r %r %r"/>' % tuple(float(v) for v in ax),
                    '<limit lower="-2" upper="2" effort="1" velocity="1"/>' if typ in ("revolute", "prismatic") else ""))
                joints[j] = dict(parent=parent, typ=typ, ax=np.array([1.0, 0, 0]) if noaxis else ax,
                                 xyz=np.zeros(3) if noorigin else xyz, rpy=np.zeros(3) if noorigin else rpy)
            xml.append("</robot>")
            r = urdf.loadstr("\n".join(xml))
            n = r.n
            assert n == sum(v["typ"] != "fixed" for v in joints.values())
            number = {l.name: l.jindex for l in r.erobot().links}                         # the reference's depth-first numbers (ERobot)
            assert {j.child: r.jindex[j.name] for j in r.joints if j.name in r.jindex} == {k: v for k, v in number.items() if v is not None}
            q = rng.uniform(-2, 2, (2, max(n, 1)))[:, :n]

            def world(j, row):
                if j == 0:
                    return np.eye(4)
                d = joints[j]
                C, ax = reference_constant(d["xyz"], d["rpy"], d["ax"])
                T = world(d["parent"], row) @ C
                if d["typ"] != "fixed":
                    k = int(np.argmax(np.abs(ax)))
                    name = ("Rx", "Ry", "Rz")[k] if d["typ"] in ("revolute", "continuous") else ("tx", "ty", "tz")[k]
                    T = T @ chains.elementary(name, (-1.0 if ax[k] < 0 else 1.0) * row[number["l%d" % j]])
                return T

            for j in range(1, nl):
                got = np.asarray(r.ets(end="l%d" % j, compact=False).eval(q if n else np.zeros((2, 0)))).reshape(-1, 4, 4)
                np.testing.assert_allclose(got, [world(j, row) for row in q], atol=1e-10, err_msg="seed %d link l%d" % (seed, j))
                paths += 1
    assert paths > 150
