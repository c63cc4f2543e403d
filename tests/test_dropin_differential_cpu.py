"""Differential test of the drop-in surface where a user of the reference stands, without a GPU.

Left: the reference's OWN classes (`ET`, `ETS` of robot/ET.py, robot/ETS.py, loaded unmodified by oracle/ref_classes.py) on the reference's
compiled extension.  Right: `rtbhip.ET` / `rtbhip.ETS` on tests/cpu_backend.py (the product's entry-point validation + the kernel bodies on the
CPU).  The same randomly generated chains are spelled with both sets of constructors and the same calls are made on both, in every argument
form the reference accepts: what comes back must have the same type of container, the same shape and the same numbers -- and where the
reference raises, this backend must raise too.  (The GPU run of the same surface: tests/test_05_reference_classes.py, tests/test_reference_suite.py.)"""
import numpy as np
import numpy.testing as nt
import pytest

import cpu_backend
from oracle import chains, ref_classes, ref_harness
from test_random_chains import random_spec

pytestmark = pytest.mark.skipif(not (ref_classes.available() and ref_harness.available()),
                                reason="needs oracle/_ref (the reference's compiled extension and byte-compiled classes)")


def build(ET, ETS, spec, SE3=None, limits=False):
    """The chain of `spec` with the given constructors; joints numbered in order of appearance (as both libraries do for an ETS);
    limits: prismatic joints get [-1.5, 1.5] (the Python solvers refuse a chain with an unlimited prismatic joint)."""
    out = []
    for item in spec:
        if isinstance(item, np.ndarray):
            out.append(ET.SE3(SE3(item) if SE3 is not None else item))
            continue
        axis, eta = item[0], item[1]
        flip = bool(item[2]) if len(item) > 2 else False
        ctor = getattr(ET, axis)
        if eta is not None:
            out.append(ctor(eta))
        elif limits and axis[0] == "t":
            out.append(ctor(flip=flip, qlim=[-1.5, 1.5]))
        else:
            out.append(ctor(flip=flip))
    return ETS(out)


def both(seed, count, nmax=8, limits=False):
    import rtbhip
    ns = ref_classes.load_reference()
    rng = np.random.default_rng(seed)
    for k in range(count):
        n = 1 + k % nmax
        spec = random_spec(rng, n)
        ref = build(ns.ET, ns.ETS, spec, SE3=ns.SE3, limits=limits)
        mine = build(rtbhip.ET, rtbhip.ETS, spec, limits=limits)
        yield n, spec, ref, mine, rng


def A(x):
    """Plain ndarray of whatever a pose-returning call handed back (SE3 stand-in, SE3Array, list of poses)."""
    if hasattr(x, "A") and not isinstance(x, np.ndarray):
        x = x.A
    if isinstance(x, (list, tuple)):
        return np.array([A(v) for v in x])
    return np.asarray(x)


def test_structure_and_string_forms_agree():
    with cpu_backend.installed():
        for n, spec, ref, mine, rng in both(3, 40):
            assert (mine.n, mine.m) == (ref.n, ref.m)
            assert str(mine) == str(ref)
            assert mine.structure == ref.structure
            nt.assert_array_equal(mine.jindices, ref.jindices)
            assert [e.isjoint for e in mine] == [e.isjoint for e in ref]
            try:
                rq = ref.qlim
            except Exception as ex:                                       # an unlimited prismatic joint: ValueError in both (robot/ETS.py:335-337)
                with pytest.raises(type(ex)):
                    mine.qlim
            else:
                nt.assert_allclose(mine.qlim, rq)
            for k, (a, b) in enumerate(zip(mine, ref)):
                # (the reference writes the automatic joint number into its COPY of the ET, robot/ETS.py:835-840; this backend keeps the
                # caller's ET untouched and numbers inside the ETS: `ets.jindices` agrees, `ets[k].jindex` stays None for automatic numbering)
                assert (a.axis, a.isflip) == (b.axis, b.isflip) and (a.jindex is None or a.jindex == b.jindex)
                if not a.isjoint:
                    nt.assert_allclose(A(a.A()), A(b.A()), atol=1e-15)
            c_m, c_r = mine.compile(), ref.compile()
            assert (c_m.n, c_m.m) == (c_r.n, c_r.m) and str(c_m) == str(c_r)
            # the inverse chain keeps every joint's number (found late in round 3: automatic numbers were re-dealt in the reversed order,
            # so ets.inv().eval(q) was not the inverse of ets.eval(q) for the same q)
            q = rng.uniform(-2, 2, n)
            i_m, i_r = mine.inv(), ref.inv()
            assert str(i_m) == str(i_r)
            nt.assert_allclose(i_m.eval(q), i_r.eval(q), atol=1e-12)
            nt.assert_allclose(i_m.eval(q) @ mine.eval(q), np.eye(4), atol=1e-12)
            nt.assert_allclose(c_m.eval(q), c_r.eval(q), atol=1e-12)


def test_kinematics_calls_agree_in_every_argument_form():
    with cpu_backend.installed():
        for n, spec, ref, mine, rng in both(5, 40):
            q = rng.uniform(-2.5, 2.5, n)
            tool = chains.elementary("tx", rng.uniform(-0.2, 0.2)) @ chains.elementary("Rx", rng.uniform(-1, 1))
            base = chains.elementary("Rz", rng.uniform(-1, 1)) @ chains.elementary("ty", rng.uniform(-0.2, 0.2))
            # one configuration: array, list, (1,n), (n,1) -- the reference takes all four as ONE q (tests/test_ETS.py:359-362)
            forms = [q, list(q), q.reshape(1, n)] + ([q.reshape(n, 1)] if n > 1 else [])
            for qq in forms:
                for kw in ({}, {"tool": tool}, {"base": base}, {"base": base, "tool": tool}, {"base": base, "include_base": False}):
                    r, m = ref.eval(qq, **kw), mine.eval(qq, **kw)
                    assert np.shape(m) == np.shape(r), (np.shape(qq), kw)
                    nt.assert_allclose(m, r, atol=1e-12)
                    nt.assert_allclose(A(mine.fkine(qq, **kw)).reshape(-1, 4, 4), A(ref.fkine(qq, **kw)).reshape(-1, 4, 4), atol=1e-12)
                for name in ("jacob0", "jacobe"):
                    for kw in ({}, {"tool": tool}):
                        r, m = getattr(ref, name)(qq, **kw), getattr(mine, name)(qq, **kw)
                        assert np.shape(m) == np.shape(r) == (6, n)
                        nt.assert_allclose(m, r, atol=1e-12)
                for name in ("hessian0", "hessiane"):
                    r, m = getattr(ref, name)(qq), getattr(mine, name)(qq)
                    assert np.shape(m) == np.shape(r) == (n, 6, n)
                    nt.assert_allclose(m, r, atol=1e-12)
            # the Hessian from a supplied Jacobian (ETS.py:1384-1392)
            nt.assert_allclose(mine.hessian0(J0=mine.jacob0(q)), ref.hessian0(J0=ref.jacob0(q)), atol=1e-12)
            nt.assert_allclose(mine.hessiane(Je=mine.jacobe(q)), ref.hessiane(Je=ref.jacobe(q)), atol=1e-12)
            # a trajectory: (N,n) -> (N,4,4) from eval, N poses from fkine; the reference has no batched Jacobian
            Q = rng.uniform(-2.5, 2.5, (7, n))
            if n > 1:                                                     # (N,1) on a one-joint chain is ONE vector in the reference
                r, m = ref.eval(Q, tool=tool), mine.eval(Q, tool=tool)
                assert np.shape(m) == np.shape(r) == (7, 4, 4)
                nt.assert_allclose(m, r, atol=1e-12)
                assert len(mine.fkine(Q)) == len(ref.fkine(Q)) == 7
                nt.assert_allclose(A(mine.fkine(Q)), A(ref.fkine(Q)), atol=1e-12)
                nt.assert_allclose(mine.jacob0(Q), np.array([ref.jacob0(row) for row in Q]), atol=1e-12)


def test_both_refuse_the_same_inputs():
    with cpu_backend.installed():
        for n, spec, ref, mine, rng in both(9, 12, nmax=6):
            sym = np.array([object()] * n, dtype=object)
            for obj in (ref, mine):
                for call in (lambda o: o.eval(sym), lambda o: o.jacob0(sym), lambda o: o.hessian0(sym)):
                    with pytest.raises(Exception):                        # the extension's TypeError("Symbolic value") starts the Python fall-back,
                        call(obj)                                         # which then fails on these objects in both libraries
            # a q row WIDER than the chain needs is read by joint number in both libraries (the C entry points take the row length from the
            # array: core/fknm.cpp:964-988 -- it is how robot.ets(end=...) is handed the whole robot's q); a narrower one the reference
            # reads out of bounds -- rtbhip refuses it
            wide1, wide2 = rng.uniform(-1, 1, n + 1), rng.uniform(-1, 1, (3, n + 2))
            nt.assert_allclose(A(mine.eval(wide1)), A(ref.eval(wide1)), atol=1e-12)
            nt.assert_allclose(A(mine.eval(wide2)), A(ref.eval(wide2)), atol=1e-12)
            nt.assert_allclose(mine.jacob0(wide1), ref.jacob0(wide1), atol=1e-12)
            if n > 2:                                                     # (a (k, 1) array is ONE configuration of k values: fknm.cpp:976-981)
                for bad in (np.zeros(n - 1), np.zeros((4, n - 1))):
                    with pytest.raises(Exception):
                        mine.eval(bad)


def test_c_solver_tuples_agree_for_a_supplied_start():
    """ik_LM / ik_GN / ik_NR with q0 given (no random restart involved on the first search): the reference's five-tuple, value for value.
    Chains whose Jacobian is rank deficient at the target (two collinear joint axes in a row: random chains produce them) are where the two
    pseudo-inverse formulations part: the reference solves with an SVD (core/ik.cpp:103-108, core/linalg.cpp `_pseudo_inverse`), which has a
    rank threshold; the kernel's `J^T (J J^T + d^2 1)^-1 e` through LDL^T has none and abandons the search on a non-finite step (the next
    restart takes over).  There only the claim of success is checked."""
    with cpu_backend.installed():
        done = degenerate = 0
        for n, spec, ref, mine, rng in both(13, 30):
            if n < 6:
                continue
            qs = rng.uniform(-1.0, 1.0, n)
            Tep = ref.eval(qs)
            q0 = qs + rng.uniform(-0.05, 0.05, n)
            sv = np.linalg.svd(ref.jacob0(qs), compute_uv=False)
            regular = sv[min(n, 6) - 1] > 1e-6 * sv[0]
            degenerate += not regular
            for name, kw in (("ik_LM", {"method": "chan"}), ("ik_LM", {"method": "wampler", "k": 0.01}), ("ik_LM", {"method": "sugihara", "k": 0.01}),
                             ("ik_GN", {"pinv": True}), ("ik_NR", {"pinv": True})):
                r = getattr(ref, name)(Tep, q0=q0, ilimit=30, slimit=1, tol=1e-6, joint_limits=False, **kw)
                m = getattr(mine, name)(Tep, q0=q0, ilimit=30, slimit=1, tol=1e-6, joint_limits=False, **kw)
                assert len(m) == len(r) == 5
                if m[1]:
                    assert m[4] < 1e-6 and np.abs(mine.eval(m[0]) - Tep).max() < 5e-3       # E = e.e / 2 < tol
                if regular or name == "ik_LM":
                    assert (int(m[1]), int(m[2]), int(m[3])) == (int(r[1]), int(r[2]), int(r[3])), (name, kw, r, m)
                    if r[1]:
                        nt.assert_allclose(m[0], r[0], atol=1e-6)
                    done += 1
        assert done >= 30 and degenerate >= 1


# ------------------------------------------------------------------------------------------------ DH robots
def random_dh(rng, n, mdh, rich=True):
    """Constructor keywords for n links: revolute / prismatic, offsets, flips, and (rich) full inertial, motor and friction parameters."""
    links = []
    for j in range(n):
        kw = {"a": float(rng.uniform(-0.4, 0.4)) if rng.random() < 0.7 else 0.0, "alpha": float(rng.choice([0.0, np.pi / 2, -np.pi / 2, 0.3])),
              "offset": float(rng.uniform(-0.5, 0.5)) if rng.random() < 0.3 else 0.0, "flip": bool(rng.random() < 0.2)}
        prismatic = rng.random() < 0.25
        if prismatic:
            kw["theta"] = float(rng.uniform(-1, 1))
            kw["qlim"] = [0.0, 0.8]
        else:
            kw["d"] = float(rng.uniform(-0.3, 0.3)) if rng.random() < 0.6 else 0.0
        if rich:
            B = rng.uniform(-0.2, 0.2, (3, 3))
            kw.update(m=float(rng.uniform(0.2, 5)), r=rng.uniform(-0.2, 0.2, 3) * (rng.random() < 0.7), I=B @ B.T + 0.01 * np.eye(3),
                      Jm=float(rng.uniform(0, 4e-4)), G=float(rng.choice([1.0, -62.6, 107.8])), B=float(rng.uniform(0, 2e-3)),
                      Tc=[float(rng.uniform(0, 0.5)), float(-rng.uniform(0, 0.5))])
        kw["flip"] = False          # (flipped DH joints: test_flipped_dh_joints_where_the_reference_is_not_self_consistent)
        links.append((prismatic, kw))
    return links


def build_dh(ns, links, mdh, **robot_kw):
    cls = {(False, False): ns.RevoluteDH, (True, False): ns.PrismaticDH, (False, True): ns.RevoluteMDH, (True, True): ns.PrismaticMDH}
    return ns.DHRobot([cls[(p, mdh)](**kw) for p, kw in links], **robot_kw)


def dh_both(seed, count):
    import rtbhip
    from test_06_reference_dh_classes import ref_dh
    ns = ref_dh()
    rng = np.random.default_rng(seed)
    for k in range(count):
        n, mdh = 2 + k % 7, bool(k % 2)
        links = random_dh(rng, n, mdh)
        yield n, mdh, build_dh(ns, links, mdh, name="r%d" % k), build_dh(rtbhip, links, mdh, name="r%d" % k), rng


@pytest.mark.skipif(not ref_classes.dh_available(), reason="needs the reference's byte-compiled DH classes")
def test_dh_robots_agree_kinematics_and_dynamics_in_every_call_form():
    symmetric = lopsided = 0
    with cpu_backend.installed() as be:
        for n, mdh, ref, mine, rng in dh_both(21, 28):
            assert (mine.n, bool(mine.mdh)) == (ref.n, bool(ref.mdh)) == (n, mdh)
            nt.assert_allclose(mine.qlim, ref.qlim)
            for a, b in zip(mine.links, ref.links):
                assert str(a) == str(b) and a.dyn() == b.dyn()
                assert bool(a.isrevolute) == bool(b.isrevolute)
                qj = float(rng.uniform(-1, 1))
                nt.assert_allclose(A(a.A(qj)), A(b.A(qj)), atol=1e-13)                     # DHLink.A, the closed form (robot/DHLink.py:633-673)
                nt.assert_allclose(a.friction(0.7), b.friction(0.7), rtol=1e-14)
            q, qd, qdd = rng.uniform(-1.5, 1.5, (3, n))
            Q, QD, QDD = rng.uniform(-1.5, 1.5, (3, 5, n))
            nt.assert_allclose(A(mine.fkine(q)), A(ref.fkine(q)), atol=1e-12)
            nt.assert_allclose(A(mine.fkine(Q)), A(ref.fkine(Q)), atol=1e-12)
            nt.assert_allclose(mine.jacob0(q), ref.jacob0(q), atol=1e-12)
            nt.assert_allclose(mine.jacobe(q), ref.jacobe(q), atol=1e-12)
            Tall_m, Tall_r = mine.fkine_all(q), ref.fkine_all(q)
            nt.assert_allclose(A(Tall_m), A(Tall_r), atol=1e-12)
            # inverse dynamics in the reference's call forms: one row, a trajectory, gravity, a tip wrench (robot/DHRobot.py:1373-1456)
            for args, kw in (((q, qd, qdd), {}), ((Q, QD, QDD), {}), ((q, qd, qdd), {"gravity": [0, 0, 0]}), ((q, qd, qdd), {"gravity": [1.0, -2.0, 9.0]}),
                             ((q, qd, qdd), {"fext": [1, 2, 3, 0.1, 0.2, 0.3]}), ((Q, QD, QDD), {"fext": [1, 2, 3, 0.1, 0.2, 0.3], "gravity": [0, 0, 3.0]})):
                r, m = ref.rne(*args, **kw), mine.rne(*args, **kw)
                assert np.shape(m) == np.shape(r)
                nt.assert_allclose(m, r, rtol=1e-9, atol=1e-10)
            scale = 1.0 + np.abs(ref.inertia(q)).max()
            for name, args in (("gravload", (q,)), ("gravload", (Q,)), ("inertia", (q,)), ("inertia", (Q,)), ("coriolis", (q, qd)), ("coriolis", (Q, QD)),
                               ("itorque", (q, qdd)), ("itorque", (Q, QDD))):
                r, m = getattr(ref, name)(*args), getattr(mine, name)(*args)
                assert np.shape(m) == np.shape(r), name
                nt.assert_allclose(m, r, rtol=1e-8, atol=1e-9 * scale, err_msg=name)
            # forward dynamics: conditioned by M(q) (motor inertias referred through G^2 dominate it for geared joints)
            Mr = ref.inertia(q)
            cond = np.linalg.cond(Mr)
            r, m = ref.accel(q, qd, qdd), mine.accel(q, qd, qdd)
            assert np.shape(m) == np.shape(r) == (n,)
            if np.abs(Mr - Mr.T).max() < 1e-9 * scale:
                nt.assert_allclose(m, r, rtol=1e-9 * cond, atol=1e-9 * cond)
                symmetric += 1
            else:
                # The reference's own recursion (core/ne.c) returns a NON-symmetric "inertia matrix" for a modified-DH chain whose FIRST joint
                # is prismatic (M[0,0] holds the motor term only: the link masses are missing from the first joint's force) -- found by this
                # test, reproduced to the bit by rtbhip's rne and inertia (asserted above).  Its accel solves with that full matrix
                # (Dynamics.py:505).  Rounds 1-3 solved with the lower triangle there (LDL^T) and recorded the deviation; since round 4 the
                # kernel keeps the full matrix for modified-DH chains with prismatic joints and solves it as it stands (csrc/ldl.h
                # lu_solve_mem): the reference's answer, confined to this case:
                assert mdh and ref.links[0].isprismatic
                nt.assert_allclose(m, r, rtol=1e-9 * cond, atol=1e-9 * cond)
                lopsided += 1
            # base / tool assigned AFTER the first calls are honoured (the reference rebuilds ets() per call; here the kept chain is dropped)
            Bm = chains.elementary("tx", 0.4) @ chains.elementary("Rz", 0.7)
            Tm = chains.elementary("tz", 0.1) @ chains.elementary("Rx", -0.3)
            mine.base, ref.base = Bm, ref.base.__class__(Bm)
            mine.tool, ref.tool = Tm, ref.tool.__class__(Tm)
            nt.assert_allclose(A(mine.fkine(q)), A(ref.fkine(q)), atol=1e-12)
            nt.assert_allclose(mine.jacob0(q), ref.jacob0(q), atol=1e-12)
            nt.assert_allclose(mine.rne(q, qd, qdd), ref.rne(q, qd, qdd), rtol=1e-9, atol=1e-10)      # gravity seen from the new base
            mine.base = ref.base = None
            mine.tool = ref.tool = None
            nt.assert_allclose(A(mine.fkine(q)), A(ref.fkine(q)), atol=1e-12)
            # a changed link parameter reaches the device table, as in the reference (Link setters -> dynchanged)
            mine.links[0].m = ref.links[0].m = 9.5
            mine.links[-1].Tc = ref.links[-1].Tc = [0.3, -0.2]
            nt.assert_allclose(mine.rne(q, qd, qdd), ref.rne(q, qd, qdd), rtol=1e-9, atol=1e-10)
            # ... and so does a changed KINEMATIC parameter (fkine and rne of the reference follow it; its jacob0 goes through the ets() built
            # at construction and does not -- not compared), and a gravity vector assigned as a list
            mine.links[-1].a = ref.links[-1].a = 0.77
            mine.gravity = ref.gravity = [0.5, 0.0, -3.0]
            nt.assert_allclose(A(mine.fkine(q)), A(ref.fkine(q)), atol=1e-12)
            nt.assert_allclose(mine.rne(q, qd, qdd), ref.rne(q, qd, qdd), rtol=1e-9, atol=1e-10)
        assert be.calls.get("rtbhip_rne", 0) > 100 and be.calls.get("rtbhip_coriolis", 0) > 20
    assert symmetric >= 20 and lopsided >= 1


@pytest.mark.skipif(not ref_classes.dh_available(), reason="needs the reference's byte-compiled DH classes")
def test_flipped_dh_joints_where_the_reference_is_not_self_consistent():
    """`flip=True` on a DH link is the one place where this backend does not return the reference's numbers, because the reference's
    numbers contradict each other there: DHLink.A (and DHRobot.fkine, a product of A's) honours the flip only when the link's ETS ENDS with
    the joint (robot/DHLink.py:636-639: modified DH, or standard DH with d = a = alpha = 0), DHRobot.jacobe (robot/DHRobot.py:1066-1140, its own
    Paul-style recursion over A) never negates the flipped joint's column, and rne never sees the flag (the 24-number link record has no
    slot for it, robot/DHRobot.py:1340-1361).  Shown here on the reference's own classes: its jacobe is NOT the derivative of its fkine for a
    flipped modified-DH joint.  This backend flips the joint in ets(), so fkine, A and the Jacobians agree with one another (checked by the
    same finite difference); rne ignores the flag as the reference's does."""
    import rtbhip
    from test_06_reference_dh_classes import ref_dh
    ns = ref_dh()
    kws = [dict(a=0.3, alpha=np.pi / 2, d=0.1), dict(a=0.2, alpha=-np.pi / 2, d=0.0, flip=True), dict(a=0.1, alpha=0.3, d=0.2)]
    q = np.array([0.3, -0.7, 0.5])

    def numjac_e(robot):
        T0 = A(robot.fkine(q))
        J = np.zeros((6, 3))
        for j in range(3):
            dq = np.zeros(3); dq[j] = 1e-7
            dT = (A(robot.fkine(q + dq)) - A(robot.fkine(q - dq))) / 2e-7
            J[:3, j] = T0[:3, :3].T @ dT[:3, 3]
            W = T0[:3, :3].T @ dT[:3, :3]
            J[3:, j] = [W[2, 1], W[0, 2], W[1, 0]]
        return J

    ref = ns.DHRobot([ns.RevoluteMDH(**kw) for kw in kws])
    assert np.abs(ref.jacobe(q) - numjac_e(ref)).max() > 0.1                      # the reference contradicts itself on the flipped joint
    with cpu_backend.installed():
        mine = rtbhip.DHRobot([rtbhip.RevoluteMDH(**kw) for kw in kws])
        nt.assert_allclose(mine.jacobe(q), numjac_e(mine), atol=1e-6)             # this backend does not
        nt.assert_allclose(A(mine.fkine(q)), A(ref.fkine(q)), atol=1e-12)         # (modified DH: the reference's fkine does honour the flip)
        z = np.zeros(3)
        nt.assert_allclose(mine.rne(q, z, z), ref.rne(q, z, z), atol=1e-12)


# ------------------------------------------------------------------------------------------------ ETS robots (link trees)
def random_tree(rng, nlinks):
    """[(name, parent name or None, ets spec)]: a tree of links, each a few constants and (mostly) one joint at the end; branching and
    static links included."""
    out = []
    for k in range(nlinks):
        parent = None if k == 0 else "L%d" % int(rng.integers(max(0, k - 3), k))
        spec = []
        for _ in range(int(rng.integers(0, 3))):
            a = ["Rx", "Ry", "Rz", "tx", "ty", "tz"][int(rng.integers(6))]
            spec.append((a, float(rng.uniform(-1.2, 1.2) if a[0] == "R" else rng.uniform(-0.3, 0.3))))
        if k == 0 or rng.random() < 0.8:
            spec.append((["Rx", "Ry", "Rz", "tx", "ty", "tz"][int(rng.integers(6))], None, bool(rng.random() < 0.25)))
        out.append(("L%d" % k, parent, spec))
    return out


def build_tree(ET, ETS, Link, Robot, tree):
    links = {}
    for name, parent, spec in tree:
        ets = ETS([getattr(ET, it[0])(it[1]) if it[1] is not None else getattr(ET, it[0])(flip=it[2]) for it in spec])
        links[name] = Link(ets, name=name, parent=None if parent is None else links[parent])
    return Robot(list(links.values()), name="tree"), links


@pytest.mark.skipif(not ref_classes.dh_available(), reason="needs the reference's byte-compiled Link / Robot classes")
def test_link_trees_agree_paths_and_kinematics():
    import rtbhip
    from test_06_reference_dh_classes import ref_dh
    ns = ref_dh()
    rLink, rRobot = ns.mods["Link"].Link, ns.mods["Robot"].Robot
    rng = np.random.default_rng(33)
    paths = 0
    with cpu_backend.installed():
        for k in range(25):
            tree = random_tree(rng, 3 + k % 7)
            ref, rl = build_tree(ns.ET, ns.ETS, rLink, rRobot, tree)
            mine, ml = build_tree(rtbhip.ET, rtbhip.ETS, rtbhip.Link, rtbhip.ERobot, tree)
            assert mine.n == ref.n and [l.name for l in mine.links] == [l.name for l in ref.links]
            assert [l.jindex for l in mine.links] == [l.jindex for l in ref.links]
            assert [bool(l.isjoint) for l in mine.links] == [bool(l.isjoint) for l in ref.links]
            if ref.n == 0:
                continue
            q = rng.uniform(-1.5, 1.5, ref.n)
            names = [t[0] for t in tree]
            for end in names:
                e_r = ref.ets(end=rl[end])
                e_m = mine.ets(end=ml[end])
                assert str(e_m) == str(e_r), (end, str(e_m), str(e_r))
                if e_r.n == 0:
                    continue
                for form in (rl[end], end):                                   # a Link object or its name
                    fm = ml[end] if form is rl[end] else end
                    nt.assert_allclose(A(mine.fkine(q, end=fm)), A(ref.fkine(q, end=form)), atol=1e-12)
                r, m = ref.jacob0(q, end=rl[end]), mine.jacob0(q, end=ml[end])
                assert np.shape(m) == np.shape(r)
                nt.assert_allclose(m, r, atol=1e-12)
                nt.assert_allclose(mine.jacobe(q, end=ml[end]), ref.jacobe(q, end=rl[end]), atol=1e-12)
                nt.assert_allclose(mine.hessian0(q, end=ml[end]), ref.hessian0(q, end=rl[end]), atol=1e-12)
                paths += 1
            # a path between two arbitrary links: up towards the common ancestor, then down (BaseRobot.py:1426-1467)
            for _ in range(4):
                a, b = names[int(rng.integers(len(names)))], names[int(rng.integers(len(names)))]
                try:
                    e_r = ref.ets(start=rl[a], end=rl[b])
                except Exception as ex:
                    with pytest.raises(type(ex)):
                        mine.ets(start=ml[a], end=ml[b])
                    continue
                e_m = mine.ets(start=ml[a], end=ml[b])
                assert str(e_m) == str(e_r), (a, b, str(e_m), str(e_r))
                if e_r.n:
                    nt.assert_allclose(A(mine.fkine(q, start=ml[a], end=ml[b])), A(ref.fkine(q, start=rl[a], end=rl[b])), atol=1e-12)
                    nt.assert_allclose(mine.jacob0(q, start=ml[a], end=ml[b]), ref.jacob0(q, start=rl[a], end=rl[b]), atol=1e-12)
            nt.assert_allclose(A(mine.fkine_all(q)), A(ref.fkine_all(q)), atol=1e-12)
    assert paths >= 60


def test_python_solvers_agree_for_a_supplied_start():
    """ETS.ikine_LM / ikine_GN / ikine_NR (the Python solvers of robot/IK.py behind them) from a supplied start that converges in the first
    search, on random chains: the IKSolution fields, value for value (success, iterations, searches, residual, q)."""
    with cpu_backend.installed():
        done = 0
        for n, spec, ref, mine, rng in both(17, 56, limits=True):
            if n < 6:
                continue
            qs = rng.uniform(-1.0, 1.0, n)
            Tep = ref.eval(qs)
            q0 = qs + rng.uniform(-0.05, 0.05, n)
            sv = np.linalg.svd(ref.jacob0(qs), compute_uv=False)
            if sv[5] < 1e-3 * sv[0]:
                continue                                                  # (rank-deficient chains: see the C-solver test above)
            for name, kw in (("ikine_LM", {"method": "chan", "k": 1.0}), ("ikine_LM", {"method": "wampler", "k": 0.01}),
                             ("ikine_LM", {"method": "sugihara", "k": 0.01}), ("ikine_GN", {"pinv": True}), ("ikine_NR", {"pinv": True})):
                r = getattr(ref, name)(Tep, q0=q0, ilimit=30, slimit=1, tol=1e-6, joint_limits=False, **kw)
                m = getattr(mine, name)(Tep, q0=q0, ilimit=30, slimit=1, tol=1e-6, joint_limits=False, **kw)
                assert (bool(m.success), int(m.iterations), int(m.searches)) == (bool(r.success), int(r.iterations), int(r.searches)), (name, kw, r, m)
                if r.success:
                    nt.assert_allclose(m.q, r.q, atol=1e-6)
                    assert abs(m.residual - r.residual) <= 1e-9 + 1e-3 * r.residual
                    assert m.reason == r.reason
                done += 1
        assert done >= 40


def test_differential_kinematics_consumers_agree():
    """ETS.manipulability (three methods x three axis selections), ETS.jacobm, ETS.partial_fkine0 (orders 2..4) on random chains.
    The manipulability measures come from the eigenvalues of a Gram matrix on the device (csrc/diff_device.h) where the reference takes an SVD
    of J itself: at a (numerically) singular configuration the device value carries an absolute error of ~1e-8 sigma_max (the square root
    of the rounding noise of the Gram matrix) where the reference returns ~1e-17 -- the tolerance below says so.  An all-zero selected block
    (a chain of prismatic joints asked for axes="rot") is 0 for "invcondition", as the reference's 1 / cond(0) is."""
    with cpu_backend.installed():
        for n, spec, ref, mine, rng in both(41, 40):
            q = rng.uniform(-2, 2, n)
            smax = np.linalg.svd(ref.jacob0(q), compute_uv=False)[0]
            for method in ("yoshikawa", "minsingular", "invcondition"):
                for axes in ("all", "trans", "rot"):
                    r = ref.manipulability(q, method=method, axes=axes)
                    m = mine.manipulability(q, method=method, axes=axes)
                    assert not np.isnan(m), (n, method, axes)
                    scale = max(1.0, smax) ** (min(n, 6 if axes == "all" else 3) if method == "yoshikawa" else 1)
                    assert abs(m - r) <= 1e-7 * scale + 1e-9 * abs(r), (n, method, axes, r, m)
            if n >= 2 and np.linalg.svd(ref.jacob0(q), compute_uv=False)[min(n, 6) - 1] > 1e-3 * smax and n >= 6:
                nt.assert_allclose(np.ravel(mine.jacobm(q)), np.ravel(ref.jacobm(q)), rtol=1e-6, atol=1e-9)
            for order in (2, 3, 4):
                r, m = ref.partial_fkine0(q, n=order), mine.partial_fkine0(q, n=order)
                assert np.shape(m) == np.shape(r)
                nt.assert_allclose(m, r, atol=1e-10)
            with pytest.raises(ValueError):
                mine.manipulability(q, method="nonsense")
            with pytest.raises(ValueError):
                ref.manipulability(q, method="nonsense")


def test_reference_classes_on_the_shim_modules_random_chains():
    """The reference's OWN ET / ETS classes bound to `rtbhip.compat.fknm` (the plug-in module with the extension's function table, here over the
    CPU replay) against the same classes on the reference's compiled extension: random chains, the calls of tests/test_05_reference_classes.py's GPU
    half -- this is the drop-in boundary itself (SURVEY section 8 row b) with nothing of rtbhip's Python mirror in between."""
    import rtbhip.compat
    with cpu_backend.installed() as be:
        before = sum(be.calls.values())
        on_ext = ref_classes.load_reference()
        on_shim = ref_classes.load(rtbhip.compat.fknm, "shim-cpu-replay")
        rng = np.random.default_rng(77)
        for k in range(30):
            n = 1 + k % 8
            spec = random_spec(rng, n)
            a = build(on_ext.ET, on_ext.ETS, spec, SE3=on_ext.SE3)
            b = build(on_shim.ET, on_shim.ETS, spec, SE3=on_shim.SE3)
            assert type(b).__module__ == "roboticstoolbox.robot.ETS" and b is not a
            q = rng.uniform(-2.5, 2.5, n)
            tool = chains.elementary("ty", 0.1) @ chains.elementary("Rz", 0.4)
            for qq in (q, list(q), q.reshape(1, n)):
                nt.assert_allclose(b.eval(qq), a.eval(qq), atol=1e-12)
                nt.assert_allclose(b.eval(qq, tool=tool), a.eval(qq, tool=tool), atol=1e-12)
                nt.assert_allclose(b.jacob0(qq), a.jacob0(qq), atol=1e-12)
                nt.assert_allclose(b.jacobe(qq, tool=tool), a.jacobe(qq, tool=tool), atol=1e-12)
                nt.assert_allclose(b.hessian0(qq), a.hessian0(qq), atol=1e-12)
                nt.assert_allclose(b.hessiane(qq), a.hessiane(qq), atol=1e-12)
            assert b.eval(q).flags["F_CONTIGUOUS"] == a.eval(q).flags["F_CONTIGUOUS"]            # (4,4) comes back in Fortran order (fknm.cpp:993)
            assert b.jacob0(q).flags["F_CONTIGUOUS"] == a.jacob0(q).flags["F_CONTIGUOUS"]
            if n > 1:
                Q = rng.uniform(-2.5, 2.5, (6, n))
                nt.assert_allclose(b.eval(Q), a.eval(Q), atol=1e-12)
            if n >= 6:
                sv = np.linalg.svd(a.jacob0(q), compute_uv=False)
                if sv[5] > 1e-3 * sv[0]:
                    q0 = q + 0.03
                    ra = a.ik_LM(a.eval(q), q0=q0, slimit=1, joint_limits=False)
                    rb = b.ik_LM(b.eval(q), q0=q0, slimit=1, joint_limits=False)
                    assert (ra[1], ra[2], ra[3]) == (rb[1], rb[2], rb[3])
                    nt.assert_allclose(rb[0], ra[0], atol=1e-6)
        assert sum(be.calls.values()) - before > 500


@pytest.mark.skipif(not ref_classes.dh_available(), reason="needs the reference's byte-compiled DH classes")
def test_reference_dh_classes_on_the_shim_modules_random_robots():
    """The reference's OWN DHRobot / DHLink / Dynamics classes bound to `rtbhip.compat.fknm` + `rtbhip.compat.frne` (CPU replay) against the same
    classes on the reference's compiled extensions: random robots (revolute / prismatic, both conventions, full inertial and friction parameters)."""
    import rtbhip.compat
    from test_06_reference_dh_classes import ref_dh
    with cpu_backend.installed() as be:
        on_ext = ref_dh()
        on_shim = ref_classes.load_dh(rtbhip.compat.fknm, rtbhip.compat.frne, "shim-dh-cpu-replay")
        rng = np.random.default_rng(91)
        for k in range(20):
            n, mdh = 2 + k % 7, bool(k % 2)
            links = random_dh(rng, n, mdh)
            a, b = build_dh(on_ext, links, mdh), build_dh(on_shim, links, mdh)
            q, qd, qdd = rng.uniform(-1.5, 1.5, (3, n))
            Q, QD, QDD = rng.uniform(-1.5, 1.5, (3, 4, n))
            nt.assert_allclose(A(b.fkine(q)), A(a.fkine(q)), atol=1e-12)
            nt.assert_allclose(b.jacob0(q), a.jacob0(q), atol=1e-12)
            for args, kw in (((q, qd, qdd), {}), ((Q, QD, QDD), {}), ((q, qd, qdd), {"gravity": [0, 0, 0]}), ((q, qd, qdd), {"fext": [1, 2, 3, 0.1, 0.2, 0.3]})):
                nt.assert_allclose(b.rne(*args, **kw), a.rne(*args, **kw), rtol=1e-9, atol=1e-10)
            scale = 1.0 + np.abs(a.inertia(q)).max()
            nt.assert_allclose(b.inertia(q), a.inertia(q), rtol=1e-8, atol=1e-9 * scale)
            nt.assert_allclose(b.coriolis(q, qd), a.coriolis(q, qd), rtol=1e-8, atol=1e-9 * scale)
            nt.assert_allclose(b.gravload(q), a.gravload(q), rtol=1e-9, atol=1e-10)
            nt.assert_allclose(b.itorque(q, qdd), a.itorque(q, qdd), rtol=1e-8, atol=1e-9 * scale)
            M = a.inertia(q)
            nt.assert_allclose(b.accel(q, qd, qdd), a.accel(q, qd, qdd), rtol=1e-9 * np.linalg.cond(M), atol=1e-9 * np.linalg.cond(M))
        assert be.calls.get("rtbhip_rne", 0) > 500                    # the reference's Dynamics mixin calls rne n or n^2 times per term


def test_solver_classes_agree_with_masks_limits_and_failures():
    """IK_LM / IK_GN / IK_NR objects (robot/IK.py:IKSolver.solve) from a supplied start: weights (`mask`), joint-limit checking, a tight tolerance,
    an iteration limit too small to converge -- success, iterations, searches, q and the `reason` string, on random chains."""
    import rtbhip
    ns = ref_classes.load_reference()
    done = 0
    with cpu_backend.installed():
        for n, spec, ref, mine, rng in both(23, 40, limits=True):
            if n < 6:
                continue
            qs = rng.uniform(-1.0, 1.0, n)
            Tep, q0 = ref.eval(qs), qs + rng.uniform(-0.05, 0.05, n)
            sv = np.linalg.svd(ref.jacob0(qs), compute_uv=False)
            if sv[5] < 1e-3 * sv[0]:
                continue
            for cls, kw in (("IK_LM", dict(method="chan", k=1.0)), ("IK_LM", dict(method="sugihara", k=0.001, mask=[1, 1, 1, 0.5, 0.5, 0.5])),
                            ("IK_LM", dict(method="wampler", k=0.01, mask=[1, 1, 1, 0, 0, 0])), ("IK_GN", dict(pinv=True)),
                            ("IK_NR", dict(pinv=True, mask=[1, 1, 1, 1, 1, 0])), ("IK_LM", dict(joint_limits=True)),
                            ("IK_LM", dict(tol=1e-10, ilimit=50)), ("IK_LM", dict(ilimit=2))):
                r = getattr(ns.IK, cls)(slimit=1, **kw).solve(ref, Tep, q0)
                m = getattr(rtbhip, cls)(slimit=1, **kw).solve(mine, Tep, q0)
                assert (bool(m.success), int(m.iterations), int(m.searches), m.reason) == (bool(r.success), int(r.iterations), int(r.searches), r.reason), (cls, kw, r, m)
                if r.success:
                    nt.assert_allclose(m.q, r.q, atol=1e-6)
                done += 1
    assert done >= 60


def test_copies_do_not_share_device_tables():
    """copy.deepcopy / copy.copy of an ETS, a DHRobot, an ERobot: the copy builds its own device table -- the original keeps working after
    the copy is collected (a shared handle was destroyed with the copy: found at the end of round 3), and an edit of the copy does not
    reach the original."""
    import copy
    import gc
    import rtbhip
    with cpu_backend.installed():
        p = rtbhip.models.DH.Puma560()
        q = np.array([0.1, 0.7, 2.8, 0.2, 0.6, 0.3])
        z = np.zeros(6)
        t0, T0 = p.rne(q, z, z).copy(), np.asarray(p.fkine(q)).copy()
        for make in (copy.deepcopy, copy.copy):
            p2 = make(p)
            nt.assert_array_equal(p2.rne(q, z, z), t0)
            nt.assert_array_equal(np.asarray(p2.fkine(q)), T0)
            del p2
            gc.collect()
            nt.assert_array_equal(p.rne(q, z, z), t0)
            nt.assert_array_equal(np.asarray(p.fkine(q)), T0)
        p3 = copy.deepcopy(p)
        p3.links[1].m = 40.0
        p3.base = chains.elementary("tx", 1.0)
        assert np.abs(p3.rne(q, z, z) - t0).max() > 1.0 and abs(np.asarray(p3.fkine(q))[0, 3] - T0[0, 3] - 1.0) < 1e-12
        nt.assert_array_equal(p.rne(q, z, z), t0)
        nt.assert_array_equal(np.asarray(p.fkine(q)), T0)
        e = rtbhip.models.Panda().ets()
        q7 = np.linspace(-0.5, 0.5, 7)
        E0 = e.eval(q7).copy()
        for make in (copy.deepcopy, copy.copy):
            e2 = make(e)
            nt.assert_array_equal(e2.eval(q7), E0)
            del e2
            gc.collect()
            nt.assert_array_equal(e.eval(q7), E0)
        er = rtbhip.ERobot(e)
        for l in er.links:
            l.m = 1.0
        g0 = er.gravload(q7).copy()
        er2 = copy.deepcopy(er)
        nt.assert_array_equal(er2.gravload(q7), g0)
        er2.links[3].m = 9.0
        assert np.abs(er2.gravload(q7) - g0).max() > 0.1
        del er2
        gc.collect()
        nt.assert_array_equal(er.gravload(q7), g0)


def test_robots_and_chains_survive_pickling():
    """One process per GPU: a robot handed to a worker through pickle (multiprocessing, torch.distributed object collectives) arrives without the
    sender's native handles and builds its own tables on first use -- same numbers, and the sender's object is untouched."""
    import pickle
    import rtbhip
    from rtbhip import urdf
    with cpu_backend.installed():
        q7 = np.linspace(-0.5, 0.5, 7)
        poe = rtbhip.PoERobot([rtbhip.PoERevolute([0, 0, 1], [0, 0, 0]), rtbhip.PoEPrismatic([0, 1, 0])], np.eye(4))
        for obj, call in ((rtbhip.models.Panda().ets(), lambda o: o.eval(q7)), (rtbhip.models.DH.Puma560(), lambda o: o.rne(q7[:6], q7[:6], q7[:6])),
                          (rtbhip.models.Panda(), lambda o: np.asarray(o.fkine(q7))), (urdf.load("Panda"), lambda o: np.asarray(o.fkine(q7))),
                          (rtbhip.ERobot(rtbhip.models.Panda().ets()), lambda o: o.jacob0(q7)), (poe, lambda o: np.asarray(o.fkine(q7[:2])))):
            before = call(obj).copy()
            twin = pickle.loads(pickle.dumps(obj))
            nt.assert_array_equal(call(twin), before)
            del twin
            nt.assert_array_equal(call(obj), before)


@pytest.mark.skipif(not ref_classes.dh_available(), reason="needs the reference's byte-compiled DH classes")
def test_dh_robots_agree_on_the_rest_of_the_surface():
    """What test_dh_robots_agree_kinematics_and_dynamics_in_every_call_form leaves out: jacob0 / jacobe(half=), jacob0(T=), hessian0 (from q and from a
    supplied Jacobian), jacob0_dot, manipulability per axes, jacobm, islimit, isspherical, friction / nofriction, todegrees / toradians, A(j, q) and
    A([j0, j1], q), payload, and the vector accessors -- value for value, and the same exception type for a bad `half`."""
    def same(r, m, tol=1e-9):
        r, m = np.asarray(A(r), dtype=float), np.asarray(A(m), dtype=float)
        assert np.shape(r) == np.shape(m)
        nt.assert_allclose(m, r, atol=tol, rtol=1e-7)
    with cpu_backend.installed():
        for n, mdh, ref, mine, rng in dh_both(55, 21):
            q, qd = rng.uniform(-1.5, 1.5, (2, n))
            for half in (None, "trans", "rot"):
                same(ref.jacob0(q, half=half), mine.jacob0(q, half=half))
                same(ref.jacobe(q, half=half), mine.jacobe(q, half=half))
            for robot in (ref, mine):
                with pytest.raises(ValueError):
                    robot.jacob0(q, half="x")
            T = ref.fkine(q)
            same(ref.jacob0(q, T=T), mine.jacob0(q, T=A(T)))
            same(ref.hessian0(q), mine.hessian0(q))
            same(ref.hessian0(J0=ref.jacob0(q)), mine.hessian0(J0=mine.jacob0(q)))
            same(ref.jacob0_dot(q, qd), mine.jacob0_dot(q, qd), 1e-8)
            for axes in ("all", "trans", "rot"):
                same(ref.manipulability(q, axes=axes), mine.manipulability(q, axes=axes), 1e-7)
            if n >= 6:
                same(ref.jacobm(q), mine.jacobm(q), 1e-6)
            same(ref.islimit(q), mine.islimit(q))
            assert bool(ref.isspherical()) == bool(mine.isspherical())
            same(ref.friction(qd), mine.friction(qd))
            same(ref.todegrees(q), mine.todegrees(q))
            same(ref.toradians(q), mine.toradians(q))
            same(ref.A(n - 1, q), mine.A(n - 1, q))
            same(ref.A([1, n - 1], q), mine.A([1, n - 1], q))
            same(ref.nofriction().rne(q, qd, qd), mine.nofriction().rne(q, qd, qd))
            ref.payload(2.0, [0.1, 0, 0.2]); mine.payload(2.0, [0.1, 0, 0.2])
            same(ref.gravload(q), mine.gravload(q))
            for attr in ("d", "a", "alpha", "theta", "offset", "r", "revolutejoints", "prismaticjoints", "mdh", "n"):
                same(getattr(ref, attr), getattr(mine, attr))


def test_joint_numbering_rules_agree():
    """Automatic numbering, explicit numbering, "numbered in order with the last one open" (robot/ETS.py:820-828: it becomes n-1), and the
    refusal of any other mixture -- the same on both sides, strings and values."""
    import rtbhip
    ns = ref_classes.load_reference()
    out = {}
    with cpu_backend.installed():
        for name, lib in (("ref", ns), ("mine", rtbhip)):
            E = lib.ET
            a = E.Rz(jindex=0) * E.tx(0.3) * E.Ry(jindex=1) * E.tz(0.2) * E.Rx()
            b = E.Rz(jindex=2) * E.tx(0.3) * E.Ry(jindex=0) * E.tz(0.2) * E.Rx(jindex=1)
            with pytest.raises(ValueError):
                (E.Rz(jindex=1) * E.Rz()).eval([0.1, 0.2])
            out[name] = (str(a), list(map(int, a.jindices)), a.eval([0.1, 0.2, 0.3]), str(b), list(map(int, b.jindices)), b.eval([0.1, 0.2, 0.3]))
    for r, m in zip(out["ref"], out["mine"]):
        if isinstance(r, np.ndarray):
            nt.assert_allclose(m, r, atol=1e-12)
        else:
            assert m == r


def test_ets_qlim_setter_takes_two_rows_and_nothing_else():
    """robot/ETS.py:346-358: qlim is (2, n) -- row 0 the lower limits -- or (2,) for a single joint.  (rtbhip used to accept (n, 2) as well and
    transposed it: for a TWO-joint chain that turned the reference's own (2, 2) form into [lo0 == hi0, lo1 == hi1], and every IK solution of a
    two-joint arm was rejected by the limit check -- found by scripts/gpu_fuzz_ik.py in round 4.)"""
    import rtbhip
    ns = ref_classes.load_reference()
    with cpu_backend.installed():
        for lib in (ns, rtbhip):
            E = lib.ET
            two = E.Rz() * E.tx(0.3) * E.Ry() * E.tx(0.3)
            two.qlim = np.array([[-1.0, -2.0], [1.5, 2.5]])
            nt.assert_array_equal(np.asarray(two.qlim), [[-1.0, -2.0], [1.5, 2.5]])
            one = E.Rz() * E.tx(0.3)
            one.qlim = np.array([-0.5, 0.75])
            nt.assert_array_equal(np.asarray(one.qlim), [[-0.5], [0.75]])
            three = E.Rz() * E.tx(0.3) * E.Ry() * E.tx(0.3) * E.Rx()
            with pytest.raises(ValueError):
                three.qlim = np.zeros((3, 2))
            # and the solver honours them: a two-joint arm, target inside the limits, from a nearby start
            two.qlim = np.array([[-2.6, -2.6], [2.6, 2.6]])
            T = two.eval([0.5, -0.7])
            sol = two.ik_LM(T, q0=np.array([0.6, -0.6]), mask=[1, 1, 1, 0, 0, 0], slimit=1)
            assert sol[1] == 1 and sol[3] == 1
            nt.assert_allclose(two.eval(sol[0])[:3, 3], np.asarray(T)[:3, 3], atol=1e-4)          # (tol = 1e-6 on E = e.e / 2)


def test_robot_paths_take_the_whole_robots_q_and_a_chain_of_constants_has_empty_derivatives():
    """robot.ets(end=link) keeps the robot-wide joint numbers and is handed the ROBOT's q (reference Robot.jacob0(q, end=) =
    self.ets(end=).jacob0(q), robot/Robot.py:1974-1981): every link of random branched robots -- numbered automatically and by hand -- against the
    product of the links' own transforms, the Jacobian's translational rows against central differences of it.  A path without a joint (the base
    link's constants) has a (6, 0) Jacobian and an empty Hessian.  Found by a fuzz run in round 4: the paths refused the robot's q, and the
    joint-less Jacobian divided by its zero row width."""
    import rtbhip
    from rtbhip import ERobot
    from oracle import chains
    from test_erobot_rne import random_tree
    from test_erobot_dynamics import renumbered_case

    def link_T(l, q, jidx):
        T = np.eye(4)
        for it in l["ets"]:
            if isinstance(it, np.ndarray):
                T = T @ it
            elif len(it) > 1 and it[1] is not None:
                T = T @ chains.elementary(it[0], it[1])
            else:
                T = T @ chains.elementary(it[0], -q[jidx] if (len(it) > 2 and it[2]) else q[jidx])
        return T

    with cpu_backend.installed():
        e = rtbhip.ET.tx(0.3) * rtbhip.ET.Rz(0.2)
        assert e.jacob0(np.zeros((5, 0))).shape == (5, 6, 0) and e.hessian0(np.zeros((5, 0))).shape == (5, 0, 6, 0)
        nt.assert_allclose(e.eval(np.zeros((2, 3)))[1], chains.elementary("tx", 0.3) @ chains.elementary("Rz", 0.2), atol=1e-15)
        links_checked = 0
        for seed in range(12):
            rng = np.random.default_rng(500 + seed)
            if seed % 2 == 0:
                prod, orc = random_tree(rng, n_links=int(rng.integers(3, 10)))
                rob = ERobot(prod)
            else:
                rob, orc, rng = renumbered_case(500 + seed, 4 + seed % 6)
            n = rob.n
            if n == 0:
                continue
            byname = {l["name"]: l for l in orc}
            jix = {l.name: l.jindex for l in rob.links}

            def world(name, qrow):
                l = byname[name]
                T = link_T(l, qrow, jix[name])
                return (world(l["parent"], qrow) if l["parent"] is not None else np.eye(4)) @ T

            q = rng.uniform(-2, 2, (4, n))
            for l in rob.links:
                path = rob.ets(end=l)
                want = np.array([world(l.name, row) for row in q])
                nt.assert_allclose(path.eval(q), want, atol=1e-12)
                J = path.jacob0(q)
                cols = [int(c) for c in path.jindices]
                assert J.shape == (4, 6, len(cols))
                for c, jq in enumerate(cols):
                    qp, qm = q[0].copy(), q[0].copy()
                    qp[jq] += 1e-6; qm[jq] -= 1e-6
                    nt.assert_allclose(J[0, :3, c], (world(l.name, qp)[:3, 3] - world(l.name, qm)[:3, 3]) / 2e-6, atol=1e-7)
                links_checked += 1
        assert links_checked > 40


@pytest.mark.skipif(not ref_classes.dh_available(), reason="needs the reference's byte-compiled Link / Robot classes")
def test_hand_numbered_link_trees_agree_paths_and_kinematics():
    """The same with the joints numbered BY HAND in a shuffled order (`Link(..., jindex=k)` on every joint link: BaseRobot.py:353-370 keeps the links
    in the order given and their numbers): link order, numbers, every path's ETS string, fkine / jacob0 / jacobe / hessian0 on the robot's q,
    fkine_all, qlim -- the reference's own Link / Robot classes against rtbhip's."""
    import rtbhip
    from test_06_reference_dh_classes import ref_dh
    ns = ref_dh()
    rLink, rRobot = ns.mods["Link"].Link, ns.mods["Robot"].Robot
    rng = np.random.default_rng(77)

    def build(ET, ETS, Link, Robot, tree, numbers):
        links = {}
        for name, parent, spec in tree:
            ets = ETS([getattr(ET, it[0])(it[1]) if it[1] is not None else getattr(ET, it[0])(flip=it[2]) for it in spec])
            kw = {"jindex": numbers[name]} if name in numbers else {}
            links[name] = Link(ets, name=name, parent=None if parent is None else links[parent], **kw)
        return Robot(list(links.values()), name="tree"), links

    paths = 0
    with cpu_backend.installed():
        for k in range(20):
            tree = random_tree(rng, 3 + k % 7)
            joint_links = [name for name, _, spec in tree if spec and spec[-1][1] is None]
            if len(joint_links) < 2:
                continue
            numbers = dict(zip(joint_links, [int(x) for x in rng.permutation(len(joint_links))]))
            ref, rl = build(ns.ET, ns.ETS, rLink, rRobot, tree, numbers)
            mine, ml = build(rtbhip.ET, rtbhip.ETS, rtbhip.Link, rtbhip.ERobot, tree, numbers)
            assert mine.n == ref.n and [l.name for l in mine.links] == [l.name for l in ref.links]
            assert [l.jindex for l in mine.links] == [l.jindex for l in ref.links]
            try:
                want = ref.qlim
            except ValueError as ex:                        # a prismatic joint without limits: "Undefined prismatic joint limit"
                with pytest.raises(ValueError, match=str(ex)):
                    mine.qlim
                lim = np.tile(np.array([[-1.0], [2.0]]), (1, ref.n)) * rng.uniform(0.5, 1.5, (1, ref.n))
                ref.qlim = lim; mine.qlim = lim             # ... set through the robot: column j to the j-th joint LINK (BaseRobot.py:1041-1051)
                want = ref.qlim
            nt.assert_array_equal(mine.qlim, want)
            assert [None if l.qlim is None else list(l.qlim) for l in mine.links] == [None if l.qlim is None else list(l.qlim) for l in ref.links]
            q = rng.uniform(-1.5, 1.5, ref.n)
            for name, _, _ in tree:
                e_r, e_m = ref.ets(end=rl[name]), mine.ets(end=ml[name])
                assert str(e_m) == str(e_r), (name, str(e_m), str(e_r))
                if e_r.n == 0:
                    continue
                nt.assert_allclose(A(mine.fkine(q, end=ml[name])), A(ref.fkine(q, end=rl[name])), atol=1e-12)
                nt.assert_allclose(mine.jacob0(q, end=ml[name]), ref.jacob0(q, end=rl[name]), atol=1e-12)
                nt.assert_allclose(mine.jacobe(q, end=ml[name]), ref.jacobe(q, end=rl[name]), atol=1e-12)
                nt.assert_allclose(mine.hessian0(q, end=ml[name]), ref.hessian0(q, end=rl[name]), atol=1e-12)
                paths += 1
            nt.assert_allclose(A(mine.fkine_all(q)), A(ref.fkine_all(q)), atol=1e-12)
    assert paths >= 40
