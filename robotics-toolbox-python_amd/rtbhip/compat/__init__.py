"""rtbhip.compat -- plug-in modules with the reference extension modules' OWN module API.

The reference binds two CPython extension modules,
    roboticstoolbox.fknm   (core/fknm.cpp:23-93,  15 functions)
    roboticstoolbox.frne   (core/frne.c:42-62,     3 functions)
and imports names from them at module scope (robot/ET.py:21, robot/ETS.py:28-38, robot/DHRobot.py:35,
robot/BaseRobot.py:41, robot/Gripper.py:13, tools/p_servo.py:7).  `rtbhip.compat.fknm` and `rtbhip.compat.frne`
expose the same function names with the same argument tuples, return shapes, memory orders and error behaviour over
librtbhip.so, so the reference's unmodified Python runs on the GPU once the two names resolve here:

    import sys, rtbhip.compat
    rtbhip.compat.install()          # sys.modules["roboticstoolbox.fknm"/"roboticstoolbox.frne"] = the shims
    import roboticstoolbox           # robot/ETS.py etc. now bind the GPU backend

(`tools/params.py`-style opt-in: nothing is patched unless install() is called or RTB_BACKEND=rtbhip is set when
`rtbhip.compat.auto()` runs.)  Every function additionally accepts a leading batch axis where the reference takes one
configuration / target / triple; there is no CPU fallback -- without librtbhip.so or a GPU every call raises.
"""
import os
import sys

from . import fknm, frne  # noqa: F401


def install(package="roboticstoolbox"):
    """Make `from <package>.fknm import ...` / `from <package>.frne import ...` resolve to the GPU shims."""
    sys.modules[package + ".fknm"] = fknm
    sys.modules[package + ".frne"] = frne
    pkg = sys.modules.get(package)
    if pkg is not None:
        pkg.fknm, pkg.frne = fknm, frne
    return fknm, frne


def auto(package="roboticstoolbox"):
    """Opt-in switch in the style of the reference's tools/params.py: install() when RTB_BACKEND=rtbhip."""
    if os.environ.get("RTB_BACKEND", "").lower() == "rtbhip":
        install(package)
        return True
    return False
