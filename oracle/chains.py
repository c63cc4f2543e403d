"""oracle/chains.py -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Independent, dependency-free (NumPy only) description of the robots used by the parity tests,
written directly from the reference model files -- NOT through the product's ET/ETS/DHRobot
classes -- so that the product's chain compiler is itself checked against something.

A chain is a list of (axis, eta, flip) tuples; eta=None marks a variable joint, which is given
the next jindex in order of appearance (what ``Robot.ets()`` does for a serial arm,
reference robot/BaseRobot.py:1426-1467).
"""
import math
import numpy as np

AXES = {"Rx": 0, "Ry": 1, "Rz": 2, "tx": 3, "ty": 4, "tz": 5}  # reference robot/ET.py:244-266


def elementary(axis, eta):
    """4x4 of one elementary transform (row-major ndarray).
    Mirrors what spatialmath's trotx/troty/trotz/transl produce for a float argument
    (reference robot/ET.py:647,685,723,758-859): cos/sin evaluated by libm, no snapping to 0."""
    T = np.eye(4)
    k = AXES[axis]
    if k <= 2:
        c, s = math.cos(eta), math.sin(eta)
        b, d = (k + 1) % 3, (k + 2) % 3
        T[b, b] = c
        T[b, d] = -s
        T[d, b] = s
        T[d, d] = c
    else:
        T[k - 3, 3] = eta
    return T


class Chain:
    """Flat op-table in the oracle's array form (see rtb_oracle.h)."""

    def __init__(self, ets, qlim=None, name=""):
        self.name = name
        kind, flip, jindex, consts = [], [], [], []
        j = 0
        for item in ets:
            if isinstance(item, np.ndarray):  # arbitrary constant SE3 (ET.SE3)
                kind.append(6); flip.append(0); jindex.append(0); consts.append(item.astype(float))
                continue
            axis, eta = item[0], item[1] if len(item) > 1 else None
            fl = bool(item[2]) if len(item) > 2 else False
            if eta is None:
                kind.append(AXES[axis]); flip.append(int(fl)); jindex.append(j); consts.append(np.eye(4))
                j += 1
            else:
                kind.append(6); flip.append(0); jindex.append(0); consts.append(elementary(axis, eta))
        self.kind = np.array(kind, dtype=np.int32)
        self.flip = np.array(flip, dtype=np.int32)
        self.jindex = np.array(jindex, dtype=np.int32)
        self.consts = np.ascontiguousarray(np.array(consts, dtype=np.float64).reshape(-1, 16))
        self.m = len(kind)
        self.n = j
        if qlim is None:  # reference robot/ET.py:109-115 defaults
            lo = [(-math.pi if k <= 2 else 0.0) for k in self.kind if k != 6]
            hi = [(math.pi if k <= 2 else 1.0) for k in self.kind if k != 6]
            qlim = np.array([lo, hi])
        self.qlim = np.ascontiguousarray(np.asarray(qlim, dtype=np.float64).reshape(2, self.n))


deg = math.pi / 180.0

# reference models/ETS/Panda.py:32-54 (22 ETs, 7 joints, no joint limits on the ETS model)
PANDA_ETS = [
    ("tz", 0.333), ("Rz",),
    ("Rx", -90 * deg), ("Rz",),
    ("Rx", 90 * deg), ("tz", 0.316), ("Rz",),
    ("tx", 0.0825), ("Rx", 90 * deg), ("Rz",),
    ("tx", -0.0825), ("Rx", -90 * deg), ("tz", 0.384), ("Rz",),
    ("Rx", 90 * deg), ("Rz",),
    ("tx", 0.088), ("Rx", 90 * deg), ("tz", 0.107), ("Rz",),
    ("tz", 103 * 1e-3), ("Rz", -math.pi / 4),  # tool_offset = (103) * mm, models/ETS/Panda.py:30
]

# Franka joint limits, reference models/DH/Panda.py:49-145
PANDA_QLIM = np.array([
    [-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973],
    [2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973],
])


def panda_ets(with_limits=False):
    return Chain(PANDA_ETS, PANDA_QLIM if with_limits else None, "Panda-ETS")


# ---------------------------------------------------------------- DH robots
class DHTable:
    """Rows: alpha, a, theta, d, sigma, offset, flip ; plus dynamics per link."""

    def __init__(self, name, mdh, rows, dyn=None, qlim=None, tool=None, base=None,
                 gravity=(0.0, 0.0, -9.81)):
        self.name, self.mdh = name, int(mdh)
        self.dh = np.array(rows, dtype=np.float64).reshape(-1, 7)
        self.n = self.dh.shape[0]
        self.dyn = dyn
        self.qlim = None if qlim is None else np.asarray(qlim, dtype=np.float64)
        self.tool = tool
        self.base = base
        self.gravity = np.array(gravity, dtype=np.float64)

    def L24(self):
        """The 24-double/link block of reference robot/DHRobot.py:1342-1358."""
        L = np.zeros((self.n, 24))
        for i in range(self.n):
            alpha, a, theta, d, sigma, offset, _ = self.dh[i]
            L[i, 0:6] = [alpha, a, theta, d, sigma, offset]
            if self.dyn is not None:
                m, r, I6, Jm, G, B, Tc = self.dyn[i]
                I = np.array([[I6[0], I6[3], I6[5]], [I6[3], I6[1], I6[4]], [I6[5], I6[4], I6[2]]])
                L[i, 6] = m
                L[i, 7:10] = r
                L[i, 10:19] = I.flatten()  # reference robot/Link.py:733-742
                L[i, 19:24] = [Jm, G, B, Tc[0], Tc[1]]
        return np.ascontiguousarray(L)

    def ets(self):
        """DH -> ETS lowering, reference robot/DHLink.py:173-225 and robot/DHRobot.py:878-918."""
        out = []
        if self.base is not None:
            out.append(np.asarray(self.base, dtype=float))
        for alpha, a, theta, d, sigma, offset, flip in self.dh:
            rev = sigma == 0
            fl = bool(flip)
            if self.mdh:
                if a != 0: out.append(("tx", a))
                if alpha != 0: out.append(("Rx", alpha))
                if rev:
                    if offset != 0: out.append(("Rz", offset))
                    if d != 0: out.append(("tz", d))
                    out.append(("Rz", None, fl))
                else:
                    if theta != 0: out.append(("Rz", theta))
                    if offset != 0: out.append(("tz", offset))
                    out.append(("tz", None, fl))
            else:
                if rev:
                    if offset != 0: out.append(("Rz", offset))
                    out.append(("Rz", None, fl))
                    if d != 0: out.append(("tz", d))
                else:
                    if theta != 0: out.append(("Rz", theta))
                    if offset != 0: out.append(("tz", offset))
                    out.append(("tz", None, fl))
                if a != 0: out.append(("tx", a))
                if alpha != 0: out.append(("Rx", alpha))
        if self.tool is not None:
            out.append(np.asarray(self.tool, dtype=float))
        return Chain(out, self.qlim.T if self.qlim is not None else None, self.name + "-ets")


def puma560():
    """reference models/DH/Puma560.py:91-177 (standard DH, n=6, full dynamics)."""
    pi = math.pi
    inch = 0.0254
    rows = [  # alpha, a, theta, d, sigma, offset, flip
        [pi / 2, 0, 0, 26.45 * inch, 0, 0, 0],
        [0.0, 0.4318, 0, 0, 0, 0, 0],
        [-pi / 2, 0.0203, 0, 0.15005, 0, 0, 0],
        [pi / 2, 0, 0, 0.4318, 0, 0, 0],
        [-pi / 2, 0, 0, 0, 0, 0, 0],
        [0.0, 0, 0, 0, 0, 0, 0],
    ]
    dyn = [  # m, r, I6 = [Ixx Iyy Izz Ixy Iyz Ixz], Jm, G, B, Tc
        (0, [0, 0, 0], [0, 0.35, 0, 0, 0, 0], 200e-6, -62.6111, 1.48e-3, [0.395, -0.435]),
        (17.4, [-0.3638, 0.006, 0.2275], [0.13, 0.524, 0.539, 0, 0, 0], 200e-6, 107.815, 0.817e-3, [0.126, -0.071]),
        (4.8, [-0.0203, -0.0141, 0.070], [0.066, 0.086, 0.0125, 0, 0, 0], 200e-6, -53.7063, 1.38e-3, [0.132, -0.105]),
        (0.82, [0, 0.019, 0], [1.8e-3, 1.3e-3, 1.8e-3, 0, 0, 0], 33e-6, 76.0364, 71.2e-6, [11.2e-3, -16.9e-3]),
        (0.34, [0, 0, 0], [0.3e-3, 0.4e-3, 0.3e-3, 0, 0, 0], 33e-6, 71.923, 82.6e-6, [9.26e-3, -14.5e-3]),
        (0.09, [0, 0, 0.032], [0.15e-3, 0.15e-3, 0.04e-3, 0, 0, 0], 33e-6, 76.686, 36.7e-6, [3.96e-3, -10.5e-3]),
    ]
    qlim = np.array([[-160, 160], [-110, 110], [-135, 135], [-266, 266], [-100, 100], [-266, 266]]) * deg
    return DHTable("Puma560", 0, rows, dyn, qlim)


PUMA_QN = np.array([0, math.pi / 4, math.pi, 0, math.pi / 4, 0])


def panda_dh():
    """reference models/DH/Panda.py:44-157 (modified DH, n=7, masses + inertias, r=0, G=1)."""
    pi = math.pi
    rows = [
        [0.0, 0.0, 0, 0.333, 0, 0, 0],
        [-pi / 2, 0.0, 0, 0.0, 0, 0, 0],
        [pi / 2, 0.0, 0, 0.316, 0, 0, 0],
        [pi / 2, 0.0825, 0, 0.0, 0, 0, 0],
        [-pi / 2, -0.0825, 0, 0.384, 0, 0, 0],
        [pi / 2, 0.0, 0, 0.0, 0, 0, 0],
        [pi / 2, 0.088, 0, 107 * 1e-3, 0, 0, 0],  # flange = (107) * mm, models/DH/Panda.py:38
    ]
    masses = [4.970684, 0.646926, 3.228604, 3.587895, 1.225946, 1.666555, 7.35522e-01]
    I6 = [
        [7.03370e-01, 7.06610e-01, 9.11700e-03, -1.39000e-04, 1.91690e-02, 6.77200e-03],
        [7.96200e-03, 2.81100e-02, 2.59950e-02, -3.92500e-03, 7.04000e-04, 1.02540e-02],
        [3.72420e-02, 3.61550e-02, 1.08300e-02, -4.76100e-03, -1.28050e-02, -1.13960e-02],
        [2.58530e-02, 1.95520e-02, 2.83230e-02, 7.79600e-03, 8.64100e-03, -1.33200e-03],
        [3.55490e-02, 2.94740e-02, 8.62700e-03, -2.11700e-03, 2.29000e-04, -4.03700e-03],
        [1.96400e-03, 4.35400e-03, 5.43300e-03, 1.09000e-04, 3.41000e-04, -1.15800e-03],
        [1.25160e-02, 1.00270e-02, 4.81500e-03, -4.28000e-04, -7.41000e-04, -1.19600e-03],
    ]
    # link defaults: r=0, Jm=0, B=0, Tc=0 (reference robot/Link.py defaults), G=1 given
    dyn = [(masses[i], [0, 0, 0], I6[i], 0.0, 1.0, 0.0, [0.0, 0.0]) for i in range(7)]
    tool = elementary("tz", 103 * 1e-3) @ elementary("Rz", -pi / 4)  # models/DH/Panda.py:147
    return DHTable("Panda-DH", 1, rows, dyn, PANDA_QLIM.T, tool=tool)
