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
