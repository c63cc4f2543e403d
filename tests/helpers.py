"""Shared fixtures for the parity tests (TEST INFRASTRUCTURE)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def literals():
    with open(os.path.join(GOLD, "reference_literals.json")) as f:
        return {k: np.array(v["value"]) for k, v in json.load(f).items()}


def ref_outputs():
    return dict(np.load(os.path.join(GOLD, "ref_outputs.npz")))


def ref_python_ik():
    """Outputs of the reference's own Python solvers (robot/IK.py) and of fknm.Angle_Axis -- make_golden.py python_ik."""
    return dict(np.load(os.path.join(GOLD, "ref_python_ik.npz")))


PI_ARRAY = np.array([0.3, 0.2, 0.4, 0.25, 0.3, 0.5, 0.35])     # one influence distance per joint (make_golden.py)

# the cases of ref_python_ik.npz: key -> (problem set, first start from q0?, slimit, step, solver keywords)
#   problem set "ik": random reachable targets (one unreachable), "ikn": near-solution / near-limit starts
PY_IK_CASES = {
    "lm_chan": ("ik", False, 100, "lm", dict(method="chan", k=1.0)),
    "lm_wampler": ("ik", False, 100, "lm", dict(method="wampler", k=0.01)),
    "lm_sugihara": ("ik", False, 100, "lm", dict(method="sugihara", k=0.01)),
    "lm_chan_q0": ("ik", True, 100, "lm", dict(method="chan", k=1.0)),
    "lm_chan_nojl_mask": ("ik", False, 40, "lm", dict(method="chan", k=0.1, joint_limits=False, mask=[1, 1, 1, 0.5, 0.5, 0.25])),
    "lm_chan_short": ("ik", False, 4, "lm", dict(method="chan", k=1.0, ilimit=5)),
    "nr_q0": ("ikn", True, 5, "nr", dict()),
    "gn_q0": ("ikn", True, 5, "gn", dict()),
    "lm_chan_ns": ("ikn", True, 3, "lm", dict(method="chan", k=1.0, kq=0.1, km=0.1, ps=0.0, pi=0.3)),
    "lm_sugihara_ns": ("ikn", True, 3, "lm", dict(method="sugihara", k=0.01, kq=0.5, km=0.0, ps=0.05, pi=0.4)),
    "lm_wampler_ns_km": ("ikn", True, 3, "lm", dict(method="wampler", k=0.01, kq=0.0, km=0.5)),
    "nr_ns": ("ikn", True, 3, "nr", dict(kq=0.01, km=1.0)),
    "gn_ns": ("ikn", True, 3, "gn", dict(kq=1.0, km=1.0)),
    "qp_class_default": ("ik", False, 100, "qp", dict(kj=0.01, ks=1.0)),
    "qp_ets_default_q0": ("ikn", True, 5, "qp", dict(kj=1.0, ks=1.0)),
    "qp_km": ("ikn", True, 3, "qp", dict(kj=0.1, ks=1.0, km=10.0)),
    "qp_mask_nojl": ("ik", True, 20, "qp", dict(kj=0.1, ks=2.0, joint_limits=False, mask=[1, 1, 1, 0.5, 0.5, 0.25])),
    "qp_kq": ("ikn", True, 3, "qp", dict(kj=0.01, ks=1.0, kq=1.0, ps=0.0, pi=0.3)),
    "qp_kq_km": ("ikn", True, 3, "qp", dict(kj=0.1, ks=1.0, kq=0.5, km=10.0, ps=0.05, pi=0.4)),
    "qp_kq_far": ("ik", False, 30, "qp", dict(kj=0.01, ks=1.0, kq=2.0, ps=0.0, pi=0.3)),
    "lm_chan_ns_pi_array": ("ikn", True, 3, "lm", dict(method="chan", k=1.0, kq=0.1, km=0.1, ps=0.0, pi=PI_ARRAY)),
    "qp_kq_pi_array": ("ikn", True, 3, "qp", dict(kj=0.01, ks=1.0, kq=1.0, ps=0.02, pi=PI_ARRAY)),
}


def angle_axis_tolerance(Te, Tep):
    """Per-pair absolute tolerance for angle-axis errors: a = atan2(|li|, tr - 1) li / |li| with li a difference of nearly
    equal entries of R, so near a half turn (|li| -> 0) a rounding difference of a few ulp in R moves the DIRECTION li / |li|
    by ~ulp / |li| (times an angle up to pi).  5e-15 away from that corner."""
    out = np.empty(len(Te))
    for i, (a, b) in enumerate(zip(Te, Tep)):
        R = b[:3, :3] @ a[:3, :3].T
        nrm = np.linalg.norm([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
        out[i] = 5e-15 + 2e-15 / max(nrm, 1e-6)
    return out


def py_ik_problem(PY, key):
    """(Tep (N,4,4), start table (N,slimit,n)) of one fixture case."""
    prob, first, slimit, step, kw = PY_IK_CASES[key]
    Tep = PY[prob + "_Tep"]
    tab = PY["ik_starts"][:, :slimit].copy()
    if first:
        tab[:, 0] = PY[prob + "_q0"]
    return Tep, tab


def mixed_spec():
    """The mixed chain of make_golden.py, as (oracle spec, builder for the product ETS)."""
    from oracle import chains
    se3 = chains.elementary("Rz", 0.3) @ chains.elementary("tx", 0.2) @ chains.elementary("Rx", 1.1)
    spec = [("Rx", None, True), ("tx", 0.3), ("ty", None), ("Ry", 0.4), ("Ry", None), se3,
            ("tz", None, True), ("Rz", None), ("tx", None), ("Rx", -0.7), ("Ry", None, True), ("tz", 0.25)]
    return spec


def product_ets(spec, qlim=None):
    """Build the product's ETS from the same (axis, eta, flip) spec the oracle Chain takes."""
    import rtbhip
    ets = rtbhip.ETS()
    for item in spec:
        if isinstance(item, np.ndarray):
            ets = ets * rtbhip.ET.SE3(item)
            continue
        axis = item[0]
        eta = item[1] if len(item) > 1 else None
        flip = bool(item[2]) if len(item) > 2 else False
        ctor = getattr(rtbhip.ET, axis)
        ets = ets * (ctor(eta) if eta is not None else ctor(flip=flip))
    if qlim is not None:
        ets.qlim = qlim
    return ets


TOOL = None
BASE = None


def tool_base():
    from oracle import chains
    tool = chains.elementary("tx", 0.1) @ chains.elementary("Ry", 0.3) @ chains.elementary("tz", -0.05)
    base = chains.elementary("Rz", 0.7) @ chains.elementary("tx", 0.2) @ chains.elementary("Rx", -0.4)
    return tool, base


def chain_from_ets(ets):
    """Oracle Chain (oracle/chains.py) with the same op-table as a product ETS whose joints are
    numbered 0..n-1 in order of appearance."""
    from oracle import chains
    spec = []
    for kind, flip, jindex, T in ets.optable():
        if kind == 6:
            spec.append(np.array(T, dtype=np.float64))
        else:
            spec.append((("Rx", "Ry", "Rz", "tx", "ty", "tz")[kind], None, bool(flip)))
    return chains.Chain(spec, qlim=ets._limits(False))


def urdf_fk_numpy(urdf_path, end, q):
    """Independent restatement of URDF forward kinematics straight from the XML (no product code):
    T = prod over the joints on the path of  Trans(xyz) * RPY(rpy) * Motion(axis, q_j), the joint
    motion being a Rodrigues rotation about / a slide along the axis AS WRITTEN in the file."""
    import math
    import xml.etree.ElementTree as XT
    root = XT.parse(urdf_path).getroot()
    parent_of = {}
    for j in root.findall("joint"):
        parent_of[j.find("child").get("link")] = j
    chain, link = [], end
    while link in parent_of:
        chain.append(parent_of[link])
        link = parent_of[link].find("parent").get("link")
    chain.reverse()
    T, k = np.eye(4), 0
    for j in chain:
        o = j.find("origin")
        xyz = [float(v) for v in (o.get("xyz", "0 0 0") if o is not None else "0 0 0").split()]
        r, p, y = [float(v) for v in (o.get("rpy", "0 0 0") if o is not None else "0 0 0").split()]
        Rx = np.array([[1, 0, 0], [0, math.cos(r), -math.sin(r)], [0, math.sin(r), math.cos(r)]])
        Ry = np.array([[math.cos(p), 0, math.sin(p)], [0, 1, 0], [-math.sin(p), 0, math.cos(p)]])
        Rz = np.array([[math.cos(y), -math.sin(y), 0], [math.sin(y), math.cos(y), 0], [0, 0, 1]])
        A = np.eye(4)
        A[:3, :3] = Rz @ Ry @ Rx
        A[:3, 3] = xyz
        T = T @ A
        typ = j.get("type")
        if typ in ("revolute", "continuous", "prismatic"):
            a = j.find("axis")
            ax = np.array([float(v) for v in (a.get("xyz") if a is not None else "1 0 0").split()])
            ax = ax / np.linalg.norm(ax)
            M = np.eye(4)
            if typ == "prismatic":
                M[:3, 3] = ax * q[k]
            else:
                K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
                M[:3, :3] = np.eye(3) + math.sin(q[k]) * K + (1 - math.cos(q[k])) * (K @ K)
            T = T @ M
            k += 1
    return T


def replaying():
    """True under the developer's CPU replay of the GPU suite (tests/conftest.py, RTBHIP_TEST_CPU_REPLAY=1)."""
    return os.environ.get("RTBHIP_TEST_CPU_REPLAY") == "1"


def DEV():
    """Where the GPU tests make their device tensors: "cuda" -- under the CPU replay host memory wears that label (tests/conftest.py)."""
    return "cpu" if replaying() else "cuda"


def full_size(N, replay_div=20):
    """A BASELINE-size batch: N rows on the GPU; the CPU replay is about the host layer's code paths, not about size, and takes N / replay_div."""
    return N // replay_div if replaying() else N


def large_sizes_served():
    """Joint counts beyond the built-in kernel sizes (IK / differential consumers / DH dynamics terms > 16 joints, tree dynamics > 20) are
    instantiated at run time by hipRTC (csrc/jit.cpp) -- on a real device with libhiprtc.so.  The CPU replay keeps the built-in set, and so does a
    box without hipRTC: there such a call is still refused loudly (RTBHIP_ELIMIT), never dropped."""
    if replaying():
        return False
    from rtbhip import jit
    return bool(jit.stats()["available"])
