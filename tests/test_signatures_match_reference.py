"""Call signatures of the drop-in classes against the reference's OWN classes (loaded unmodified, oracle/ref_classes.py): for every public
method the reference class has and this backend has too -- parameter names, positional order and DEFAULT VALUES.  A different default is the
quietest way for a drop-in to return different numbers for the same call (found this way at the end of round 3: `DHRobot.ikine_LM`, which the
reference overrides with `joint_limits=False` and no start / end; `p_servo`, whose default method is "rpy").  Accepted differences are listed
with their reason; anything else fails."""
import inspect

import numpy as np
import pytest

from oracle import ref_classes, ref_harness

pytestmark = pytest.mark.skipif(not (ref_classes.dh_available() and ref_harness.available()),
                                reason="needs oracle/_ref (the reference's byte-compiled classes)")

ACCEPTED = {
    # list editing of an ETS (collections.UserList in the reference): positional in every use, the parameter name is incidental
    "ETS.append": "item", "ETS.count": "item", "ETS.extend": "other", "ETS.index": "item", "ETS.remove": "item",
    # a superset: the argument the reference requires has a default here
    "ETS.partial_fkine0": "n", "DHRobot.todegrees": "q", "Robot.rne": "qd qdd",
    # the same value spelled differently (None -> zeros(3); 0 -> 0.0; [] -> None -> ETS(); list -> tuple)
    "DHRobot.payload": "p", "DHLink.__init__": "offset", "Link.__init__": "ets", "Robot.__init__": "gravity keywords",
    "Robot~URDFRobot.rne": "qd qdd", "Robot~PoERobot.rne": "qd qdd", "Robot~models.ERobot.rne": "qd qdd",
    # built from other things than a link list: a URDF string, an ETS with limits, twists and a zero pose
    "Robot~URDFRobot.__init__": "*", "Robot~models.ERobot.__init__": "*", "Robot~PoERobot.__init__": "*",
}


def _sig(f):
    try:
        s = inspect.signature(f)
    except (TypeError, ValueError):
        return None
    return [(n, p.default if p.default is not inspect.Parameter.empty else "<required>", p.kind.name) for n, p in s.parameters.items() if n != "self"]


def _same(a, b):
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        return np.array_equal(np.asarray(a), np.asarray(b))
    return repr(a) == repr(b) or (isinstance(a, (int, float)) and isinstance(b, (int, float)) and not isinstance(a, bool) and float(a) == float(b))


def test_public_method_signatures_and_defaults():
    import rtbhip
    from test_06_reference_dh_classes import ref_dh
    ns = ref_dh()
    pairs = [("ET", ns.ET, rtbhip.ET), ("ETS", ns.ETS, rtbhip.ETS), ("DHRobot", ns.DHRobot, rtbhip.DHRobot), ("DHLink", ns.DHLink, rtbhip.DHLink),
             ("RevoluteDH", ns.RevoluteDH, rtbhip.RevoluteDH), ("PrismaticDH", ns.PrismaticDH, rtbhip.PrismaticDH),
             ("RevoluteMDH", ns.RevoluteMDH, rtbhip.RevoluteMDH), ("PrismaticMDH", ns.PrismaticMDH, rtbhip.PrismaticMDH),
             ("Link", ns.mods["Link"].Link, rtbhip.Link), ("Robot", ns.mods["Robot"].Robot, rtbhip.ERobot),
             # the other robot classes of this backend answer to the reference's Robot too (the methods they share with it)
             ("Robot~models.ERobot", ns.mods["Robot"].Robot, rtbhip.models.ERobot), ("Robot~URDFRobot", ns.mods["Robot"].Robot, rtbhip.urdf.URDFRobot),
             ("Robot~PoERobot", ns.mods["Robot"].Robot, rtbhip.PoERobot),
             ("IKSolver", ns.IK.IKSolver, rtbhip.IKSolver), ("IK_LM", ns.IK.IK_LM, rtbhip.IK_LM), ("IK_NR", ns.IK.IK_NR, rtbhip.IK_NR),
             ("IK_GN", ns.IK.IK_GN, rtbhip.IK_GN), ("IK_QP", ns.IK.IK_QP, rtbhip.IK_QP)]
    problems, compared = [], 0
    for cname, R, M in pairs:
        for name in sorted(set(dir(R))):
            if (name.startswith("_") and name != "__init__") or not hasattr(M, name):
                continue
            if isinstance(inspect.getattr_static(R, name, None), property):
                continue
            fr, fm = getattr(R, name), getattr(M, name)
            if not callable(fr) or not callable(fm):
                continue
            sr, sm = _sig(fr), _sig(fm)
            if sr is None or sm is None:
                continue
            compared += 1
            mine = {n: d for n, d, _ in sm}
            takes_kw = any(k == "VAR_KEYWORD" for _, _, k in sm)
            allowed = ACCEPTED.get("%s.%s" % (cname, name), "").split()
            if allowed == ["*"]:
                continue
            for n, d, k in sr:
                if k in ("VAR_KEYWORD", "VAR_POSITIONAL") or n in allowed:
                    continue
                if n not in mine:
                    if not takes_kw:
                        problems.append("%s.%s: no parameter %r" % (cname, name, n))
                elif not _same(d, mine[n]):
                    problems.append("%s.%s: %s defaults to %r in the reference, %r here" % (cname, name, n, d, mine[n]))
            order_r = [n for n, _, k in sr if k == "POSITIONAL_OR_KEYWORD"]
            order_m = [n for n, _, k in sm if k == "POSITIONAL_OR_KEYWORD"]
            common = [n for n in order_r if n in order_m]
            if [n for n in order_m if n in common] != common:
                problems.append("%s.%s: positional order %s in the reference, %s here" % (cname, name, order_r, order_m))
            elif common and order_m[:len(common)] != common and order_r[:len(common)] == common:
                problems.append("%s.%s: a parameter of this backend sits BEFORE the reference's %s: %s" % (cname, name, common, order_m))
    assert compared > 150, compared
    assert not problems, "\n".join(problems)


def test_module_level_functions():
    import rtbhip
    ns = ref_classes.load_reference()
    for name in ("p_servo", "angle_axis", "angle_axis_python"):
        sr, sm = _sig(getattr(ns.p_servo, name)), _sig(getattr(rtbhip, name))
        assert [(n, d) for n, d, _ in sr] == [(n, d) for n, d, _ in sm][:len(sr)], (name, sr, sm)
