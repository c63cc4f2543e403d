"""rtbhip -- MI355X-native batched kinematics/dynamics backend behind the Robotics Toolbox API.

Only the hot path named in BASELINE.json is here: ET/ETS (eval, fkine, jacob0, jacobe, hessian0,
ik_LM, ikine_LM), DHRobot (fkine, jacob0, jacobe, rne) and the Panda / Puma560 models.  All
arithmetic runs in hand-written HIP kernels (../csrc) behind the C ABI of include/rtbhip.h.
"""
from ._lib import RtbHipError, lib, device_count, tune, shard_range, last_launch  # noqa: F401
from .et import ET, ETS, IKSolution  # noqa: F401
from .dh import DHLink, DHRobot, RevoluteDH, PrismaticDH, RevoluteMDH, PrismaticMDH  # noqa: F401
from .erobot import Link, ERobot  # noqa: F401
from . import models  # noqa: F401
from . import urdf  # noqa: F401
from .fleet import fleet_fkine_jacob  # noqa: F401
from .shard import ShardedBatch  # noqa: F401

__all__ = ["ET", "ETS", "IKSolution", "DHLink", "DHRobot", "RevoluteDH", "PrismaticDH", "RevoluteMDH",
           "PrismaticMDH", "Link", "ERobot", "models", "urdf", "fleet_fkine_jacob", "ShardedBatch", "RtbHipError", "lib",
           "device_count", "tune", "shard_range", "last_launch"]
