"""Mixed-fleet evaluation (BASELINE config 5): several different chains, each with its own batch,
walked by ONE kernel launch (variable-length chains, block -> chain map)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib, as_numeric, is_torch, MEM_HOST, MEM_DEVICE


def fleet_fkine_jacob_packed(chains, qs, frame=0, out=None):
    """As fleet_fkine_jacob, with ONE output array per chain: TJ_c is (N_c, 16 + 6 n_c), row = [T (16) | J (6 n_c)] (rtbhip_fleet_fkine_jacob_packed:
    a single write stream per chain).  Returns the list of TJ arrays; `out` = such a list of an earlier call."""
    if len(chains) != len(qs):
        raise ValueError("one q batch per chain")
    k = len(chains)
    if out is not None and len(out) != k:
        raise ValueError("out must hold one buffer per chain")
    tm = k > 0 and is_torch(qs[0]) and qs[0].is_cuda
    handles = (C.c_uint64 * max(1, k))()
    qp = (C.c_void_p * max(1, k))()
    Tp = (C.c_void_p * max(1, k))()
    Ns = (C.c_int64 * max(1, k))()
    keep, TJs = [], []
    for i, (ch, q) in enumerate(zip(chains, qs)):
        handles[i] = ch._handle()
        w = 16 + 6 * ch.n
        if tm:
            import torch
            q2 = q.reshape(-1, ch.q_width).contiguous()
            _lib.note_device(q2)
            TJ = torch.empty((q2.shape[0], w), dtype=torch.float64, device=q2.device) if out is None else out[i]
            if tuple(TJ.shape) != (q2.shape[0], w) or not TJ.is_contiguous() or TJ.dtype != torch.float64 or TJ.device != q2.device:
                raise ValueError("out buffer of chain %d does not match its batch" % i)
            qp[i], Tp[i] = q2.data_ptr(), TJ.data_ptr()
        else:
            q2 = np.ascontiguousarray(as_numeric(q).reshape(-1, ch.q_width))
            TJ = _lib.host_empty((q2.shape[0], w)) if out is None else out[i]
            if not isinstance(TJ, np.ndarray) or TJ.shape != (q2.shape[0], w) or not TJ.flags.c_contiguous or TJ.dtype != np.float64:
                raise ValueError("out buffer of chain %d does not match its batch" % i)
            qp[i], Tp[i] = q2.ctypes.data, TJ.ctypes.data
        Ns[i] = q2.shape[0]
        keep.append(q2)
        TJs.append(TJ)
    check(lib().rtbhip_fleet_fkine_jacob_packed(handles, k, qp, Ns, int(frame), Tp, MEM_DEVICE if tm else MEM_HOST,
                                                _lib.current_stream_ptr() if tm else None))
    return TJs


def fleet_fkine_jacob(chains, qs, frame=0, out=None):
    """chains: list of ETS; qs: list of (N_c, n_c) arrays (all NumPy or all CUDA float64 tensors).
    Returns (list of T (N_c,4,4), list of J (N_c,6,n_c)).  `out` = (Ts, Js) of an earlier call: the results are written into those
    buffers (a serving loop then allocates nothing per step)."""
    if len(chains) != len(qs):
        raise ValueError("one q batch per chain")
    k = len(chains)
    if out is not None and (len(out) != 2 or len(out[0]) != k or len(out[1]) != k):
        raise ValueError("out must be (Ts, Js) with one buffer per chain")
    tm = k > 0 and is_torch(qs[0]) and qs[0].is_cuda
    handles = (C.c_uint64 * max(1, k))()
    qp = (C.c_void_p * max(1, k))()
    Tp = (C.c_void_p * max(1, k))()
    Jp = (C.c_void_p * max(1, k))()
    Ns = (C.c_int64 * max(1, k))()
    keep, Ts, Js = [], [], []
    for i, (ch, q) in enumerate(zip(chains, qs)):
        handles[i] = ch._handle()
        if tm:
            import torch
            q2 = q.reshape(-1, ch.q_width).contiguous()
            _lib.note_device(q2)
            if out is None:
                T = torch.empty((q2.shape[0], 4, 4), dtype=torch.float64, device=q2.device)
                J = torch.empty((q2.shape[0], 6, ch.n), dtype=torch.float64, device=q2.device)
            else:
                T, J = out[0][i], out[1][i]
                if (tuple(T.shape) != (q2.shape[0], 4, 4) or tuple(J.shape) != (q2.shape[0], 6, ch.n) or not T.is_contiguous()
                        or not J.is_contiguous() or T.dtype != torch.float64 or J.dtype != torch.float64 or T.device != q2.device):
                    raise ValueError("out buffers of chain %d do not match its batch" % i)
            qp[i], Tp[i], Jp[i] = q2.data_ptr(), T.data_ptr(), J.data_ptr()
        else:
            q2 = np.ascontiguousarray(as_numeric(q).reshape(-1, ch.q_width))
            if out is None:
                T = _lib.host_empty((q2.shape[0], 4, 4))
                J = _lib.host_empty((q2.shape[0], 6, ch.n))
            else:
                T, J = out[0][i], out[1][i]
                if (not isinstance(T, np.ndarray) or T.shape != (q2.shape[0], 4, 4) or J.shape != (q2.shape[0], 6, ch.n)
                        or not T.flags.c_contiguous or not J.flags.c_contiguous or T.dtype != np.float64 or J.dtype != np.float64):
                    raise ValueError("out buffers of chain %d do not match its batch" % i)
            qp[i], Tp[i], Jp[i] = q2.ctypes.data, T.ctypes.data, J.ctypes.data
        Ns[i] = q2.shape[0]
        keep.append(q2)
        Ts.append(T)
        Js.append(J)
    check(lib().rtbhip_fleet_fkine_jacob(handles, k, qp, Ns, int(frame), Tp, Jp, MEM_DEVICE if tm else MEM_HOST,
                                         _lib.current_stream_ptr() if tm else None))
    return Ts, Js
