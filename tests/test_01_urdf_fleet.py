"""URDF -> ETS lowering (rtbhip/urdf.py, SURVEY 8f-3 / BASELINE config 5).  CPU tests run the lowered
chains through tests/emu (the kernels' own device functions executed on the host) and the C oracle;
the GPU test walks the 16-arm fleet in one launch."""
import os

import numpy as np
import numpy.testing as nt
import pytest

import rtbhip
from rtbhip import urdf
from oracle import oracle
from helpers import chain_from_ets, urdf_fk_numpy, DEV, full_size

TOL = 1e-10


def _rand_q(rng, ets, N):
    ql = np.clip(ets.qlim, -6.3, 6.3)      # some descriptions carry +-1e10-style "no limit" limits
    return rng.uniform(ql[0], ql[1], (N, ets.n))


def test_all_twenty_descriptions_load_and_lower():
    names = urdf.available()
    assert len(names) == 20 and set(urdf.FLEET16) <= set(names) and len(set(urdf.FLEET16)) == 16
    dof = {}
    for name in names:
        r = urdf.load(name)
        e = r.ets()
        assert e.n == r.njoints(r.ee) >= 4
        assert e.qlim.shape == (2, e.n) and np.all(e.qlim[0] <= e.qlim[1])
        dof[name] = e.n
    assert dof["AL5D"] == 4 and dof["Panda"] == 7 and dof["UR5"] == 6 and dof["Fetch"] == 10
    assert urdf.load("YuMi").n == 18 and urdf.load("Panda").n == 9       # arm joints + gripper fingers


def test_G12_readme_panda_fkine_qr():
    """reference README.md:218-226: URDF Panda fkine(qr), gripper tool included, 4 significant figures."""
    import emu_harness as emu
    p = urdf.load("Panda")
    qr = np.array([0, -0.3, 0, -2.2, 0, 2.0, np.pi / 4])
    T, _, _ = emu.kin(p.ets(), qr, want=("T",))            # ets() carries the gripper tool (BaseRobot.ets, robot/BaseRobot.py:1610-1616)
    want = np.array([[0.995, 0, 0.09983, 0.484], [0, -1, 0, 0], [0.09983, 0, -0.995, 0.4126], [0, 0, 0, 1]])
    nt.assert_allclose(T[0], want, atol=6e-4)


@pytest.mark.parametrize("name", urdf.FLEET16)
def test_lowering_equals_independent_urdf_fk_and_oracle(name):
    """The lowered chain (kernel device code via tests/emu) == FK computed straight from the XML
    (tests/helpers.urdf_fk_numpy) == the C oracle on the same op-table; J0 == numerical Jacobian."""
    import emu_harness as emu
    r = urdf.load(name)
    e = r.ets()
    rng = np.random.default_rng(abs(hash(name)) % 1000)
    q = _rand_q(rng, e, 6)
    T, J, _ = emu.kin(e, q, want=("T", "J"), reg=e.n <= 8)
    path = os.path.join(urdf.DATA_DIR, name + ".urdf")
    tool = np.eye(4) if r.tool is None else r.tool          # ets() ends with the model's gripper tool (Panda), the XML walk does not
    for i in range(len(q)):
        nt.assert_allclose(T[i], urdf_fk_numpy(path, r.ee, q[i]) @ tool, atol=1e-12)
    oc = chain_from_ets(e)
    nt.assert_allclose(T, oracle.fkine(oc, q), atol=TOL)
    nt.assert_allclose(J, oracle.jacob0(oc, q), atol=TOL)
    h = 1e-6
    for k in range(e.n):
        dq = np.zeros(e.n); dq[k] = h
        Tp, _, _ = emu.kin(e, q[0] + dq, want=("T",), reg=e.n <= 8)
        Tm, _, _ = emu.kin(e, q[0] - dq, want=("T",), reg=e.n <= 8)
        nt.assert_allclose(J[0][:3, k], (Tp[0][:3, 3] - Tm[0][:3, 3]) / (2 * h), atol=1e-6)


def test_skew_axis_quirk_and_robotwide_jindex_and_errors():
    xml = """<robot name="t"><link name="a"/><link name="b"/><link name="c"/><link name="d"/>
      <joint name="j1" type="revolute"><parent link="a"/><child link="b"/><origin xyz="0 0 0.1" rpy="0.1 0.2 0.3"/>
        <axis xyz="0 -1 0"/><limit lower="-1" upper="2"/></joint>
      <joint name="j2" type="prismatic"><parent link="b"/><child link="c"/><origin xyz="0.2 0 0"/>
        <axis xyz="1 0 0"/><limit lower="0" upper="0.5"/></joint>
      <joint name="j3" type="continuous"><parent link="a"/><child link="d"/><axis xyz="0 0.6 0.8"/></joint></robot>"""
    r = urdf.loadstr(xml)
    e = r.ets(end="c")
    assert [x.axis for x in e.joints()] == ["Ry", "tx"] and e.joints()[0].isflip and not e.joints()[1].isflip
    nt.assert_allclose(e.qlim, [[-1, 0], [2, 0.5]])
    # skew axis: the reference folds angvec2r(|v|, v/|v|) into the constant and turns about z (urdf.py:1710-1722)
    e3 = r.ets(end="d")
    assert e3.joints()[0].axis == "Rz"
    nt.assert_allclose(e3[0].T[:3, :3], urdf.angvec_matrix(1.0, np.array([0, 0.6, 0.8])), atol=1e-15)
    nt.assert_allclose(e3.qlim, [[-np.pi], [np.pi]])
    # robot-wide numbering keeps URDF joint order; compact numbering is per path
    assert list(r.ets(end="d", compact=False).jindices) == [2] and list(r.ets(end="d").jindices) == [0]
    with pytest.raises(ValueError):
        urdf.loadstr(xml.replace('name="d"/>', 'name="c"/>', 1))
    with pytest.raises(ValueError):
        urdf.loadstr("<notrobot/>")
    with pytest.raises(ValueError):
        urdf.load("NoSuchRobot")


@pytest.mark.gpu
def test_gpu_fleet16_urdf_arms_one_launch_vs_oracle():
    """BASELINE config 5 shape: the 16 URDF arms (4..10 joints on the path), each with its own batch,
    through ONE rtbhip_fleet_fkine_jacob launch; every arm against the C oracle."""
    rng = np.random.default_rng(5)
    robots = [urdf.load(n) for n in urdf.FLEET16]
    chs = [r.ets() for r in robots]
    assert sorted({c.n for c in chs}) == [4, 5, 6, 7, 8, 9, 10]
    sizes = [1000, 65, 1, 129, 64, 63, 500, 7, 2048, 100, 300, 77, 1025, 640, 33, 999]
    qs = [_rand_q(rng, c, N) for c, N in zip(chs, sizes)]
    Ts, Js = rtbhip.fleet_fkine_jacob(chs, qs)
    for c, q, T, J in zip(chs, qs, Ts, Js):
        oc = chain_from_ets(c)
        nt.assert_allclose(T, oracle.fkine(oc, q), atol=TOL)
        nt.assert_allclose(J, oracle.jacob0(oc, q), atol=TOL)
        Tb, Jb = c.fkine_jacob0(q)                       # the per-chain kernels agree with the fleet kernel
        nt.assert_allclose(T, np.reshape(Tb, T.shape), atol=1e-13)
        nt.assert_allclose(J, np.reshape(Jb, J.shape), atol=1e-13)
    # out=: a second call writes into the buffers of the first (host arrays and device tensors)
    want = [T.copy() for T in Ts], [J.copy() for J in Js]
    for T, J in zip(Ts, Js):
        T[...] = 0; J[...] = 0
    T2, J2 = rtbhip.fleet_fkine_jacob(chs, qs, out=(Ts, Js))
    assert all(a is b for a, b in zip(T2, Ts)) and all(a is b for a, b in zip(J2, Js))
    for a, b in zip(Ts + Js, want[0] + want[1]):
        nt.assert_array_equal(a, b)
    import torch
    qd = [torch.from_numpy(q).cuda() for q in qs]
    Td, Jd = rtbhip.fleet_fkine_jacob(chs, qd)
    ptrs = [t.data_ptr() for t in Td + Jd]
    for t in Td + Jd:
        t.zero_()
    Td2, Jd2 = rtbhip.fleet_fkine_jacob(chs, qd, out=(Td, Jd))
    assert [t.data_ptr() for t in Td2 + Jd2] == ptrs
    for a, b in zip(Td + Jd, want[0] + want[1]):
        nt.assert_array_equal(a.cpu().numpy(), b)
    with pytest.raises(ValueError):
        rtbhip.fleet_fkine_jacob(chs, qd, out=(Td[:-1], Jd))


@pytest.mark.gpu
def test_gpu_full_size_fleet16_1e6_each_properties_and_sampled_parity():
    """BASELINE configs[4] at full size: 16 URDF arms x 1e6 configurations, device resident, one fleet call.
    Size-independent properties over every row (rotation blocks orthonormal with det +1, bottom row 0 0 0 1,
    angular Jacobian columns of revolute joints unit length, prismatic ones zero) + oracle parity on a sample."""
    import torch
    N = full_size(1000000, 50)
    robots = [urdf.load(n) for n in urdf.FLEET16]
    chs = [r.ets() for r in robots]
    qs = []
    for i, c in enumerate(chs):
        ql = np.clip(c.qlim, -2 * np.pi, 2 * np.pi)
        g = torch.Generator(device=DEV()).manual_seed(4 + i)
        lo, hi = (torch.from_numpy(x).cuda() for x in (ql[0], ql[1]))
        qs.append(lo + (hi - lo) * torch.rand((N, c.n), dtype=torch.float64, device=DEV(), generator=g))
    Ts, Js = rtbhip.fleet_fkine_jacob(chs, qs)
    torch.cuda.synchronize()
    eye = torch.eye(3, dtype=torch.float64, device=DEV())
    for c, q, T, J in zip(chs, qs, Ts, Js):
        R = T[:, :3, :3]
        assert float((R @ R.transpose(1, 2) - eye).abs().max()) < 1e-12
        assert float((torch.linalg.det(R) - 1).abs().max()) < 1e-12
        assert bool((T[:, 3, :] == torch.tensor([0.0, 0, 0, 1], dtype=torch.float64, device=DEV())).all())
        wn = torch.linalg.norm(J[:, 3:, :], dim=1)                       # (N, n)
        rev = torch.tensor([e.isrotation for e in c.joints()], device=DEV())
        assert float((wn[:, rev] - 1).abs().max()) < 1e-12
        if (~rev).any():
            assert float(wn[:, ~rev].abs().max()) == 0.0
        idx = torch.randint(0, N, (64,), device=DEV())
        oc = chain_from_ets(c)
        qh = q[idx].cpu().numpy()
        nt.assert_allclose(T[idx].cpu().numpy(), oracle.fkine(oc, qh), atol=TOL)
        nt.assert_allclose(J[idx].cpu().numpy(), oracle.jacob0(oc, qh), atol=TOL)
        del T, J


@pytest.mark.gpu
def test_gpu_fleet_more_chains_than_one_launch_table_holds():
    """40 chains (> the 32-entry kernel-argument table): the call is split into several launches per class."""
    rng = np.random.default_rng(8)
    robots = [urdf.load(urdf.FLEET16[i % 16]) for i in range(40)]
    chs = [r.ets() for r in robots]
    qs = [_rand_q(rng, c, 50 + 3 * i) for i, c in enumerate(chs)]
    Ts, Js = rtbhip.fleet_fkine_jacob(chs, qs)
    for c, q, T, J in zip(chs, qs, Ts, Js):
        oc = chain_from_ets(c)
        nt.assert_allclose(T, oracle.fkine(oc, q), atol=TOL)
        nt.assert_allclose(J, oracle.jacob0(oc, q), atol=TOL)
    assert rtbhip.fleet_fkine_jacob([], []) == ([], [])


def test_Robot_URDF_reads_a_file_and_folds_the_gripper(tmp_path):
    """Robot.URDF(file_path, gripper=) (robot/Robot.py:288-330).  The reference's own pins (tests/test_Robot.py:618-628): the Fetch with
    gripper = link 6 has 5 joints, with gripper = "forearm_roll_link" 7.  A path of the reference's data package resolves to the shipped
    description; a file on disk is read as it is (xacro through rtbhip.xacro); unknown paths are refused."""
    r = rtbhip.ERobot.URDF("fetch_description/robots/fetch.urdf", gripper=6)
    assert r.n == 5
    r = rtbhip.ERobot.URDF("fetch_description/robots/fetch.urdf", gripper="forearm_roll_link")
    assert r.n == 7 and "forearm_roll_link" not in [l.name for l in r.links]
    whole = rtbhip.ERobot.URDF("fetch_description/robots/fetch.urdf")
    assert whole.n == urdf.load("Fetch").n and whole.urdf_string.lstrip().startswith("<")
    # a plain file on disk
    src = open(os.path.join(os.path.dirname(urdf.__file__), "data", "urdf", "UR5.urdf")).read()
    path = tmp_path / "my_arm.urdf"
    path.write_text(src)
    mine = rtbhip.ERobot.URDF(str(path))
    ref = urdf.load("UR5").erobot()
    assert mine.n == ref.n and [l.name for l in mine.links] == [l.name for l in ref.links]
    u = urdf.read(path)
    assert u.n == urdf.load("UR5").n
    with pytest.raises(ValueError):
        rtbhip.ERobot.URDF("fetch_description/robots/fetch.urdf", gripper="no_such_link")
    with pytest.raises(TypeError):
        rtbhip.ERobot.URDF("fetch_description/robots/fetch.urdf", gripper=1.5)
    with pytest.raises(FileNotFoundError):
        rtbhip.ERobot.URDF("nowhere/robot.urdf")
    # a subclass is constructed through its own __init__ (robot/Robot.py:325-331: `return cls(links, name=..., urdf_string=..., urdf_filepath=...)`)
    class MyArm(rtbhip.ERobot):
        def __init__(self, links, **kw):
            super().__init__(links, **kw)
            self.qr = np.zeros(self.n)
    arm = MyArm.URDF(str(path))
    assert type(arm) is MyArm and arm.qr.shape == (ref.n,) and arm.urdf_filepath == str(path) and arm.name == mine.name
    x = tmp_path / "arm.urdf.xacro"                      # an xacro file goes through rtbhip.xacro (tests/test_xacro.py); plain URDF is valid xacro
    x.write_text(src)
    assert urdf.read(x).n == u.n


@pytest.mark.gpu
def test_gpu_Robot_URDF_kinematics_equal_the_loaded_model():
    rng = np.random.default_rng(17)
    r = rtbhip.ERobot.URDF("ur_description/urdf/ur5_joint_limited_robot.urdf.xacro")
    u = urdf.load("UR5")
    q = rng.uniform(-2, 2, (40, r.n))
    end = u.ee
    nt.assert_allclose(r.fkine(q, end=end), u.fkine(q, end=end), atol=1e-12)
    nt.assert_allclose(r.jacob0(q, end=end), u.jacob0(q, end=end), atol=1e-12)
