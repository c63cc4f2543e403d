"""Robot models on the hot path, with the parameters of the reference model files:
models/ETS/Panda.py:32-54, models/DH/Panda.py:44-157, models/DH/Puma560.py:91-177."""
import math

import numpy as np

from .et import ET, ETS
from .kinematics import RobotKinematics
from .dh import DHRobot, RevoluteDH, RevoluteMDH


class ERobot(RobotKinematics):
    """Minimal ETS-robot facade: the kinematic pass-throughs of reference
    robot/RobotKinematics.py:92-97 (fkine applies the robot base), :158 (jacob0), :219 (jacobe),
    :736 (ik_LM), :1209-1225 (ikine_LM)."""

    def __init__(self, ets, name="", manufacturer="", base=None, tool=None, qlim=None):
        self._ets = ets if isinstance(ets, ETS) else ETS(ets)
        if qlim is not None:
            self._ets.qlim = qlim
        self.name, self.manufacturer = name, manufacturer
        self.base = None if base is None else np.asarray(base, dtype=np.float64)
        self.tool = None if tool is None else np.asarray(tool, dtype=np.float64)
        self.q = np.zeros(self._ets.n)                         # the stored configuration (BaseRobot.q)

    @property
    def n(self): return self._ets.n
    @property
    def qlim(self): return self._ets.qlim
    @qlim.setter
    def qlim(self, v): self._ets.qlim = v

    def _links(self):
        """The link segments the reference's Robot(ETS) makes (robot/Robot.py:117-131): a link frame after every joint, named
        link0, link1, ...; trailing constants form a last, static link."""
        return self._ets.split()

    def ets(self, start=None, end=None):
        """The whole chain, or -- BaseRobot.ets(start, end), robot/BaseRobot.py:1555-1652 -- the part from link `start` to
        link `end` (names "link0".."linkK" or indices); as in the reference the start link's own transform is included."""
        if start is None and end is None:
            return self._ets
        segs = self._links()

        def index(x, default):
            if x is None:
                return default
            if isinstance(x, str):
                names = ["link%d" % k for k in range(len(segs))]
                if x not in names:
                    raise ValueError("no link named %s" % x)
                return names.index(x)
            k = int(x)
            if not 0 <= k < len(segs):
                raise ValueError("link not in robot links")
            return k
        a, b = index(start, 0), index(end, len(segs) - 1)
        if a > b:
            raise ValueError("Could not find the requested ETS in this robot")
        made = self.__dict__.setdefault("_ets_made", {})
        if (a, b) not in made:
            made[(a, b)] = ETS([e for seg in segs[a:b + 1] for e in seg])
        return made[(a, b)]

    def fkine_all(self, q):
        """Poses of the base and of every link frame: (K+1,4,4) or (N,K+1,4,4) (reference Robot.fkine_all robot/Robot.py:638-698), one
        chain walk per configuration."""
        from .et import _poses
        k = 0
        marks = [0]
        for seg in self._links():
            k += len(seg)
            marks.append(k)
        return _poses(self._ets.link_frames(q, marks, base=self.base))

    # fkine, jacob0, jacobe, hessian0/e, jacob0_dot, manipulability, jacobm, jacob0_analytical, partial_fkine0, ik_* and ikine_*
    # come from RobotKinematics: each is self.ets(start, end).<method>(...) with the robot's base / tool, as in the reference.


class Panda(ERobot):
    """Franka-Emika Panda as an ETS, 22 ETs / 7 joints (reference models/ETS/Panda.py:32-54)."""

    def __init__(self):
        deg = math.pi / 180
        mm = 1e-3
        tool_offset = 103 * mm
        ets = (ET.tz(0.333) * ET.Rz()
               * ET.Rx(-90 * deg) * ET.Rz()
               * ET.Rx(90 * deg) * ET.tz(0.316) * ET.Rz()
               * ET.tx(0.0825) * ET.Rx(90, "deg") * ET.Rz()
               * ET.tx(-0.0825) * ET.Rx(-90, "deg") * ET.tz(0.384) * ET.Rz()
               * ET.Rx(90, "deg") * ET.Rz()
               * ET.tx(0.088) * ET.Rx(90, "deg") * ET.tz(0.107) * ET.Rz()
               * ET.tz(tool_offset) * ET.Rz(-math.pi / 4))
        super().__init__(ets, name="Panda", manufacturer="Franka Emika")
        self.qr = np.array([0, -0.3, 0, -2.2, 0, 2.0, math.pi / 4])
        self.qz = np.zeros(7)


# Franka joint limits (reference models/DH/Panda.py:49-145)
PANDA_QLIM = np.array([
    [-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973],
    [2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973],
])


class DH:
    class Panda(DHRobot):
        """Modified-DH Panda with masses and inertias (reference models/DH/Panda.py:44-157)."""

        def __init__(self):
            pi = math.pi
            mm = 1e-3
            par = [  # a, d, alpha, qlim, m, I6
                (0.0, 0.333, 0.0, [-2.8973, 2.8973], 4.970684,
                 [7.03370e-01, 7.06610e-01, 9.11700e-03, -1.39000e-04, 1.91690e-02, 6.77200e-03]),
                (0.0, 0.0, -pi / 2, [-1.7628, 1.7628], 0.646926,
                 [7.96200e-03, 2.81100e-02, 2.59950e-02, -3.92500e-03, 7.04000e-04, 1.02540e-02]),
                (0.0, 0.316, pi / 2, [-2.8973, 2.8973], 3.228604,
                 [3.72420e-02, 3.61550e-02, 1.08300e-02, -4.76100e-03, -1.28050e-02, -1.13960e-02]),
                (0.0825, 0.0, pi / 2, [-3.0718, -0.0698], 3.587895,
                 [2.58530e-02, 1.95520e-02, 2.83230e-02, 7.79600e-03, 8.64100e-03, -1.33200e-03]),
                (-0.0825, 0.384, -pi / 2, [-2.8973, 2.8973], 1.225946,
                 [3.55490e-02, 2.94740e-02, 8.62700e-03, -2.11700e-03, 2.29000e-04, -4.03700e-03]),
                (0.0, 0.0, pi / 2, [-0.0175, 3.7525], 1.666555,
                 [1.96400e-03, 4.35400e-03, 5.43300e-03, 1.09000e-04, 3.41000e-04, -1.15800e-03]),
                (0.088, 107 * mm, pi / 2, [-2.8973, 2.8973], 7.35522e-01,
                 [1.25160e-02, 1.00270e-02, 4.81500e-03, -4.28000e-04, -7.41000e-04, -1.19600e-03]),
            ]
            L = [RevoluteMDH(a=a, d=d, alpha=al, qlim=np.array(ql), m=m, I=I, G=1) for a, d, al, ql, m, I in par]
            tool = np.eye(4)
            tool[2, 3] = 103 * mm
            c, s = math.cos(-pi / 4), math.sin(-pi / 4)
            rz = np.eye(4)
            rz[0, 0], rz[0, 1], rz[1, 0], rz[1, 1] = c, -s, s, c
            super().__init__(L, name="Panda", manufacturer="Franka Emika", tool=tool @ rz)
            self.qr = np.array([0, -0.3, 0, -2.2, 0, 2.0, pi / 4])
            self.qz = np.zeros(7)

    class Puma560(DHRobot):
        """Standard-DH Puma 560 with full dynamics (reference models/DH/Puma560.py:91-177)."""

        def __init__(self):
            pi = math.pi
            deg = pi / 180
            inch = 0.0254
            L = [
                RevoluteDH(d=26.45 * inch, a=0, alpha=pi / 2, I=[0, 0.35, 0, 0, 0, 0], r=[0, 0, 0], m=0,
                           Jm=200e-6, G=-62.6111, B=1.48e-3, Tc=[0.395, -0.435], qlim=[-160 * deg, 160 * deg]),
                RevoluteDH(d=0, a=0.4318, alpha=0.0, I=[0.13, 0.524, 0.539, 0, 0, 0], r=[-0.3638, 0.006, 0.2275],
                           m=17.4, Jm=200e-6, G=107.815, B=0.817e-3, Tc=[0.126, -0.071], qlim=[-110 * deg, 110 * deg]),
                RevoluteDH(d=0.15005, a=0.0203, alpha=-pi / 2, I=[0.066, 0.086, 0.0125, 0, 0, 0],
                           r=[-0.0203, -0.0141, 0.070], m=4.8, Jm=200e-6, G=-53.7063, B=1.38e-3,
                           Tc=[0.132, -0.105], qlim=[-135 * deg, 135 * deg]),
                RevoluteDH(d=0.4318, a=0, alpha=pi / 2, I=[1.8e-3, 1.3e-3, 1.8e-3, 0, 0, 0], r=[0, 0.019, 0],
                           m=0.82, Jm=33e-6, G=76.0364, B=71.2e-6, Tc=[11.2e-3, -16.9e-3],
                           qlim=[-266 * deg, 266 * deg]),
                RevoluteDH(d=0, a=0, alpha=-pi / 2, I=[0.3e-3, 0.4e-3, 0.3e-3, 0, 0, 0], r=[0, 0, 0], m=0.34,
                           Jm=33e-6, G=71.923, B=82.6e-6, Tc=[9.26e-3, -14.5e-3], qlim=[-100 * deg, 100 * deg]),
                RevoluteDH(d=0, a=0, alpha=0.0, I=[0.15e-3, 0.15e-3, 0.04e-3, 0, 0, 0], r=[0, 0, 0.032], m=0.09,
                           Jm=33e-6, G=76.686, B=36.7e-6, Tc=[3.96e-3, -10.5e-3], qlim=[-266 * deg, 266 * deg]),
            ]
            super().__init__(L, name="Puma 560", manufacturer="Unimation")
            self.qr = np.array([0, pi / 2, -pi / 2, 0, 0, 0])
            self.qz = np.zeros(6)
            self.qn = np.array([0, pi / 4, pi, 0, pi / 4, 0])


class Puma560ETS(ERobot):
    """Unimation Puma560 as an ETS, 13 ETs / 6 joints (reference models/ETS/Puma560.py:38-68: zero angles = the vertical pose)."""

    def __init__(self):
        l1, l2, l3, l4, l5, l6 = 0.672, -0.2337, 0.4318, 0.0203, 0.0837, 0.4318
        ets = (ET.tz(l1) * ET.Rz() * ET.ty(l2) * ET.Ry() * ET.tz(l3) * ET.tx(l4) * ET.ty(l5) * ET.Ry() * ET.tz(l6) * ET.Rz() * ET.Ry() * ET.Rz()
               * ET.tx(0.2))
        super().__init__(ets, name="Puma560", manufacturer="Unimation")
        self.qr = np.array([0, -math.pi / 2, math.pi / 2, 0, 0, 0])
        self.qz = np.zeros(6)


class ETSModels:
    """what the reference reaches as `rtb.models.ETS.<name>`"""
    Panda = Panda
    Puma560 = Puma560ETS
