"""oracle/ref_harness.py -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Drives the reference's OWN native extension modules (``fknm`` and ``frne``, built unmodified from
/root/reference by oracle/Makefile into oracle/_ref/) through their raw capsule API, without
``spatialmath`` (which is not installable here, so ``import roboticstoolbox`` is impossible).

The 7-tuple handed to ``ET_init`` is exactly what reference robot/ET.py:117-125 passes; ``T`` must
be a Fortran-order float64 4x4 kept alive by the caller because the C struct borrows the pointer
(reference core/fknm.cpp:1207).
"""
import glob
import importlib.util
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_MODS = {}


def available():
    return bool(glob.glob(os.path.join(_HERE, "_ref", "fknm*.so"))) and \
        bool(glob.glob(os.path.join(_HERE, "_ref", "frne*.so")))


def _load(name):
    if name not in _MODS:
        hits = glob.glob(os.path.join(_HERE, "_ref", name + "*.so"))
        if not hits:
            raise ImportError("oracle/_ref/%s*.so not built (run `make -f oracle/Makefile ref` "
                              "where /root/reference exists)" % name)
        spec = importlib.util.spec_from_file_location(name, hits[0])
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _MODS[name] = mod
    return _MODS[name]


class RefETS:
    """An ETS capsule of the reference extension built from an oracle.chains.Chain."""

    def __init__(self, ch):
        fknm = _load("fknm")
        self.ch = ch
        self.n, self.m = ch.n, ch.m
        self._keep = []
        caps = []
        jq = 0
        for i in range(ch.m):
            k = int(ch.kind[i])
            if k == 6:
                T = np.asfortranarray(ch.consts[i].reshape(4, 4).copy())
                ql = np.array([0.0, 0.0])
                cap = fknm.ET_init(0, 0, 0, 0, 0, T, ql)
            else:
                T = np.asfortranarray(np.eye(4))
                ql = np.array([ch.qlim[0, jq], ch.qlim[1, jq]])
                jq += 1
                cap = fknm.ET_init(0, 1, int(ch.flip[i]), int(ch.jindex[i]), k, T, ql)
            self._keep.append((T, ql))
            caps.append(cap)
        self._caps = caps
        self.cap = fknm.ETS_init(caps, ch.n, ch.m)
        self.fknm = fknm

    def fkine(self, q, base=None, tool=None, include_base=True):
        """ETS_fkine: one C call over the whole (N,n) array (fknm.cpp:923-1064)."""
        q = np.ascontiguousarray(q, dtype=np.float64)
        out = self.fknm.ETS_fkine(self.cap, q, base, tool, int(include_base))
        return np.array(out)  # (4,4) F-order for a single q, (N,4,4) C-order for a trajectory

    def jacob0(self, q, tool=None):
        return np.array(self.fknm.ETS_jacob0(self.cap, np.asarray(q, dtype=np.float64), tool))

    def jacobe(self, q, tool=None):
        return np.array(self.fknm.ETS_jacobe(self.cap, np.asarray(q, dtype=np.float64), tool))

    def jacob0_batch(self, q, tool=None):
        """What a reference user must do today: a Python loop (no C batch loop exists)."""
        q = np.ascontiguousarray(q, dtype=np.float64)
        f = self.fknm.ETS_jacob0
        cap = self.cap
        out = np.empty((q.shape[0], 6, self.n))
        for i in range(q.shape[0]):
            out[i] = f(cap, q[i], tool)
        return out

    def jacobe_batch(self, q, tool=None):
        q = np.ascontiguousarray(q, dtype=np.float64)
        out = np.empty((q.shape[0], 6, self.n))
        for i in range(q.shape[0]):
            out[i] = self.fknm.ETS_jacobe(self.cap, q[i], tool)
        return out

    def hessian0(self, q, tool=None):
        q = np.asarray(q, dtype=np.float64)
        J = self.fknm.ETS_jacob0(self.cap, q, tool)
        return np.array(self.fknm.ETS_hessian0(self.cap, q, J, tool))

    def ik_LM(self, Tep, q0=None, ilimit=30, slimit=100, tol=1e-6, joint_limits=True, mask=None,
              k=1.0, method="chan"):
        """IK_LM_c (fknm.cpp:394-525) with the argument order of ETS.ik_LM (ETS.py:2168-2170)."""
        Tep = np.ascontiguousarray(Tep, dtype=np.float64)
        return self.fknm.IK_LM_c(self.cap, Tep, q0, ilimit, slimit, tol, int(joint_limits), mask,
                                 float(k), method)


    def ik_GN(self, Tep, q0=None, ilimit=30, slimit=100, tol=1e-6, joint_limits=True, mask=None, pinv=True, pinv_damping=0.0):
        """IK_GN_c (fknm.cpp:279-392) with the argument order of ETS.ik_GN (ETS.py:2430-2442)."""
        Tep = np.ascontiguousarray(Tep, dtype=np.float64)
        return self.fknm.IK_GN_c(self.cap, Tep, q0, ilimit, slimit, tol, int(joint_limits), mask, int(pinv), float(pinv_damping))

    def ik_NR(self, Tep, q0=None, ilimit=30, slimit=100, tol=1e-6, joint_limits=True, mask=None, pinv=True, pinv_damping=0.0):
        """IK_NR_c (fknm.cpp:164-277) with the argument order of ETS.ik_NR (ETS.py:2287-2298)."""
        Tep = np.ascontiguousarray(Tep, dtype=np.float64)
        return self.fknm.IK_NR_c(self.cap, Tep, q0, ilimit, slimit, tol, int(joint_limits), mask, int(pinv), float(pinv_damping))


class RefRNE:
    """frne capsule from the 24-double/link block (DHRobot._init_rne, DHRobot.py:1340-1361)."""

    def __init__(self, L24, mdh, gravity=(0, 0, -9.81)):
        self.frne = _load("frne")
        L = np.ascontiguousarray(L24, dtype=np.float64).reshape(-1, 24)
        self.n = L.shape[0]
        self.gravity = np.array(gravity, dtype=np.float64)
        # "we negate gravity here, since the C code has the sign wrong" DHRobot.py:1360-1361
        self.cap = self.frne.init(self.n, int(mdh), L.flatten(), -self.gravity)

    def rne(self, q, qd, qdd, gravity=None, fext=None):
        """The per-row Python loop of DHRobot.rne (DHRobot.py:1442-1451)."""
        q = np.asarray(q, dtype=np.float64).reshape(-1, self.n)
        qd = np.asarray(qd, dtype=np.float64).reshape(-1, self.n)
        qdd = np.asarray(qdd, dtype=np.float64).reshape(-1, self.n)
        g = self.gravity if gravity is None else np.asarray(gravity, dtype=np.float64)
        f = np.zeros(6) if fext is None else np.asarray(fext, dtype=np.float64)
        tau = np.zeros((q.shape[0], self.n))
        frne = self.frne.frne
        cap = self.cap
        ng = -g
        for i in range(q.shape[0]):
            tau[i, :] = frne(cap, q[i, :], qd[i, :], qdd[i, :], ng, f)
        return tau

    def delete(self):
        self.frne.delete(self.cap)
        self.cap = None
