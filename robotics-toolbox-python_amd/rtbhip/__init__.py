"""rtbhip -- MI355X-native batched kinematics/dynamics backend behind the Robotics Toolbox API.

Only the hot path named in BASELINE.json / SURVEY.md section 8 is here:
  ET / ETS            eval, fkine, jacob0, jacobe, hessian0/e, jacob0_dot, manipulability, jacobm,
                      ik_LM / ik_GN / ik_NR (C-solver semantics), ikine_LM / ikine_GN / ikine_NR (Python-solver semantics)
  DHLink / DHRobot    fkine, jacob0/e, rne, gravload, itorque, inertia, coriolis, accel (+ the ETS pass-throughs)
  Link / ERobot       ETS robots (link trees): rne, and the RobotKinematics surface over ets(start, end)
  PoERevolute / PoEPrismatic / PoERobot   product-of-exponentials robots: twists lowered to the same chain form
  urdf / xacro        URDF loader + the reference's URDF -> ETS lowering; xacro expander (stdlib only); 20 pre-expanded robot descriptions
  angle_axis / p_servo / hessian_from_jacobian    the exports of the extension module that take finished matrices
  compat.fknm / compat.frne   plug-in modules with the reference extension modules' own function tables
  fleet_fkine_jacob   many different chains in one call;  ShardedBatch / shard_range  one row block per GPU rank
All arithmetic runs in hand-written HIP kernels (../csrc) behind the C ABI of include/rtbhip.h; there is no
CPU fallback: every call raises RtbHipError when librtbhip.so or a GPU is missing.
"""
from ._lib import RtbHipError, lib, device_count, tune, shard_range, last_launch, ik_target_base, trim  # noqa: F401
from .et import ET, ETS, IKSolution, angle_axis, angle_axis_python, p_servo, hessian_from_jacobian, manipulability_from_jacobian, jacobm_from_jacobian  # noqa: F401
from .ik import IKSolver, IK_NR, IK_GN, IK_LM, IK_QP  # noqa: F401
from .dh import DHLink, DHRobot, RevoluteDH, PrismaticDH, RevoluteMDH, PrismaticMDH  # noqa: F401
SerialLink = DHRobot          # the reference keeps the old name as an alias (robot/DHRobot.py:2505-2520)
from .erobot import Link, ERobot  # noqa: F401
from .poe import PoELink, PoERevolute, PoEPrismatic, PoERobot  # noqa: F401
from .kinematics import RobotKinematics  # noqa: F401
from . import models  # noqa: F401
from . import urdf  # noqa: F401
from . import xacro  # noqa: F401
from . import jit  # noqa: F401
from .fleet import fleet_fkine_jacob, fleet_fkine_jacob_packed  # noqa: F401
from .shard import ShardedBatch, Communicator  # noqa: F401

__all__ = ["ET", "ETS", "IKSolution", "IKSolver", "IK_NR", "IK_GN", "IK_LM", "IK_QP", "angle_axis", "angle_axis_python", "p_servo", "hessian_from_jacobian", "manipulability_from_jacobian", "jacobm_from_jacobian", "DHLink", "DHRobot", "RevoluteDH", "PrismaticDH", "RevoluteMDH",
           "PrismaticMDH", "Link", "ERobot", "PoELink", "PoERevolute", "PoEPrismatic", "PoERobot", "RobotKinematics", "models", "urdf", "xacro", "jit", "fleet_fkine_jacob", "fleet_fkine_jacob_packed", "ShardedBatch", "Communicator", "RtbHipError", "lib",
           "device_count", "tune", "shard_range", "last_launch", "ik_target_base", "trim"]
