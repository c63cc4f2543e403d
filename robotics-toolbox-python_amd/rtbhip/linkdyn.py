"""The dynamic parameters every link of the reference carries -- mass, centre of mass, inertia tensor, motor inertia, gear ratio, viscous and
Coulomb friction -- with the accepted input forms and the refusals of reference robot/Link.py (BaseLink: the property setters :700-900,
`friction` :1395-1448, `nofriction` :1350-1393, `islimit` :1300-1330, `dyn` :1229-1300, `_dyn2list` :1302-1355).  Shared by `DHLink`
(rtbhip/dh.py) and `Link` (rtbhip/erobot.py): these are the numbers `DHRobot.L24()` / `ERobot.group_table()` hand to the device tables.
Host-side bookkeeping only -- nothing here is on the batched path."""
import numpy as np


def inertia_matrix(I):
    """3x3 from (3,3) / 9 / 6 = [Ixx Iyy Izz Ixy Iyz Ixz] / 3 = the diagonal (reference robot/Link.py:719-751)."""
    if I is None:
        return np.zeros((3, 3))
    I = np.asarray(I, dtype=np.float64)
    if I.shape == (3, 3):
        M = I
    elif I.size == 9:
        M = I.reshape(3, 3)
    elif I.size == 6:
        I = I.reshape(6)
        M = np.array([[I[0], I[3], I[5]], [I[3], I[1], I[4]], [I[5], I[4], I[2]]])
    elif I.size == 3:
        M = np.diag(I.reshape(3))
    else:
        raise ValueError("invalid shape passed: must be (3,3), (6,), (3,)")
    if np.any(np.abs(M - M.T) > 1e-8):
        raise ValueError("3x3 matrix is not symmetric")
    return M.copy()


def _scalar(name, v):
    if v is None:
        return 0.0
    if np.ndim(v) != 0:
        raise TypeError("%s must be a scalar" % name)
    return float(v)


class LinkDynamics:
    """m, r, I, Jm, G, B, Tc as validated properties; friction / nofriction / islimit / dyn / _dyn2list."""

    def _set_dynamics(self, m=None, r=None, I=None, Jm=None, G=None, B=None, Tc=None):
        given = [x is not None for x in (m, r, I, Jm, G, B, Tc)]
        self.m, self.r, self.I, self.Jm, self.G, self.B, self.Tc = m, r, I, Jm, G, B, Tc
        self._hasdynamics = any(given)                                   # robot/Link.py:172-187

    @property
    def hasdynamics(self): return bool(getattr(self, "_hasdynamics", False))

    @property
    def m(self): return self._m
    @m.setter
    def m(self, v): self._m = _scalar("m", v)

    @property
    def r(self): return self._r
    @r.setter
    def r(self, v): self._r = np.zeros(3) if v is None else np.asarray(v, dtype=np.float64).reshape(3).copy()

    @property
    def I(self): return self._I
    @I.setter
    def I(self, v): self._I = inertia_matrix(v)

    @property
    def Jm(self): return self._Jm
    @Jm.setter
    def Jm(self, v): self._Jm = _scalar("Jm", v)

    @property
    def G(self): return self._G
    @G.setter
    def G(self, v): self._G = _scalar("G", v)

    @property
    def B(self): return self._B
    @B.setter
    def B(self, v): self._B = _scalar("B", v)

    @property
    def Tc(self): return self._Tc
    @Tc.setter
    def Tc(self, v):
        """A scalar (or one value) is symmetric friction [v, -v]; two values are [Tc+, Tc-] (robot/Link.py:830-880)."""
        if v is None:
            self._Tc = np.zeros(2)
            return
        a = np.asarray(v, dtype=np.float64).reshape(-1)
        if a.size == 1:
            self._Tc = np.array([a[0], -a[0]])
        elif a.size == 2:
            self._Tc = a.copy()
        else:
            raise ValueError("Coulomb friction vector must be length 2")

    def _copy(self):
        raise DeprecationWarning("Use copy method of Link class")         # robot/Link.py:409-410

    # ---------------------------------------------------------------- what the reference's link methods answer
    def islimit(self, q):
        """q outside [qlim0, qlim1]; no limits set -> False (robot/Link.py:1300-1330)."""
        ql = self.qlim
        return False if ql is None else bool(q < ql[0] or q > ql[1])

    def friction(self, qd, coulomb=True):
        """Joint friction torque at joint velocity qd, referred to the link side: -|G| (B |G| qd + Tc+/-) (robot/Link.py:1395-1448)."""
        tau = self.B * abs(self.G) * qd
        if coulomb:
            tau += self.Tc[0] if qd > 0 else (self.Tc[1] if qd < 0 else 0.0)
        return -abs(self.G) * tau

    def nofriction(self, coulomb=True, viscous=False):
        """A copy with the Coulomb (and, if asked, viscous) friction removed (robot/Link.py:1350-1393)."""
        l = self.copy()
        if viscous:
            l.B = 0.0
        if coulomb:
            l.Tc = [0.0, 0.0]
        return l

    def dyn(self, indent=0):
        """The inertial and motor parameters as the reference's ten-line text (robot/Link.py:1229-1300)."""
        ql = self.qlim
        lo, hi = (0.0, 0.0) if ql is None else (float(ql[0]), float(ql[1]))
        g = lambda x: "{:8.2g}".format(float(x))
        I = self.I
        rows = ["m     =  %s " % g(self.m),
                "r     =  %s %s %s " % tuple(g(x) for x in self.r),
                "        | %s %s %s | " % tuple(g(x) for x in I[0]),
                "I     = | %s %s %s | " % tuple(g(x) for x in I[1]),
                "        | %s %s %s | " % tuple(g(x) for x in I[2]),
                "Jm    =  %s " % g(self.Jm),
                "B     =  %s " % g(self.B),
                "Tc    =  %s(+) %s(-) " % (g(self.Tc[0]), g(self.Tc[1])),
                "G     =  %s " % g(self.G),
                "qlim  =  %s to %s" % (g(lo), g(hi))]
        pad = " " * int(indent)
        return "\n".join(pad + r for r in rows)

    def _dyn2list(self, fmt="{: .3g}"):
        """[m, r, I as (Ixx Iyy Izz Ixy Iyz Ixz), Jm, B, Tc, G] as strings: the rows of DHRobot.dynamics_list (robot/Link.py:1302-1355)."""
        I = self.I
        vec = lambda xs: ", ".join(fmt.format(float(x)) for x in xs)
        return [fmt.format(self.m), vec(self.r), vec([I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[1, 2], I[0, 2]]), fmt.format(self.Jm),
                fmt.format(self.B), vec(self.Tc), fmt.format(self.G)]
