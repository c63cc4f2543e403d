/* rtbhip.h -- C ABI of librtbhip.so: the MI355X (gfx950) batched kinematics/dynamics backend.
 *
 * This is the drop-in boundary for the ONE hot path of petercorke/robotics-toolbox-python that
 * this project replaces: what sits under the reference's two CPython extension modules
 *     fknm  (src/roboticstoolbox/core/fknm.cpp:23-93, method table of 15 functions) and
 *     frne  (src/roboticstoolbox/core/frne.c:42-62,   method table of 3 functions).
 * Those are CPython-ABI modules (PyCapsule handles + NumPy arrays); the entry points below are the
 * plain-C equivalents a maintainer binds with ctypes (INTEGRATION.md shows the stub): opaque
 * 64-bit handles instead of capsules, caller-owned buffers instead of returned ndarrays, and -- new
 * -- a batch dimension N on every call, because the reference only loops over rows in C for
 * ETS_fkine (fknm.cpp:1038-1052) and in Python everywhere else.
 *
 * General rules
 *   - every function returns 0 on success, a negative RTBHIP_E* code otherwise; the text of the
 *     last error of the calling thread is rtbhip_last_error().  Nothing throws across the ABI.
 *   - all matrices are float64.  A 4x4 homogeneous transform is 16 doubles ROW-major (the
 *     layout of one [i,:,:] slice of the (N,4,4) C-order array ETS_fkine returns, fknm.cpp:1002-1005).
 *   - q is (N, q_width) C-contiguous; q_width = max jindex + 1 (== n for a serial arm).
 *   - mem: RTBHIP_MEM_HOST   pointers are host memory; the call streams the rows through the device
 *                            (chunked, double-buffered, see rtbhip_host_alloc) and returns when the
 *                            results are in the output arrays;
 *          RTBHIP_MEM_DEVICE pointers are device memory; the call only enqueues work on `stream` (a hipStream_t,
 *                            NULL = default stream).  The call runs on the GPU that OWNS the buffers: when that is
 *                            not the current device the library switches to it for the duration of the call and
 *                            switches back (`stream` must then be a stream of that GPU).
 *   - small per-call parameters (base, tool, gravity, fext, we, qlim) are always HOST pointers.
 *   - handles are bound to no device: the chain/dynamics tables (a few KB) are uploaded lazily
 *     to whichever device a call runs on and cached there.
 */
#ifndef RTBHIP_H
#define RTBHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTBHIP_OK 0
#define RTBHIP_EINVAL (-1)   /* bad argument (NULL, negative size, unknown handle, bad kind ...) */
#define RTBHIP_EHIP (-2)     /* a HIP runtime call failed (no device, OOM, launch failure)       */
#define RTBHIP_ELIMIT (-3)   /* chain too long / too many joints for this build                  */

#define RTBHIP_MEM_HOST 0
#define RTBHIP_MEM_DEVICE 1

/* elementary transform kinds: 0..5 are the reference's axis numbers (robot/ET.py:244-266) */
#define RTBHIP_ET_RX 0
#define RTBHIP_ET_RY 1
#define RTBHIP_ET_RZ 2
#define RTBHIP_ET_TX 3
#define RTBHIP_ET_TY 4
#define RTBHIP_ET_TZ 5
#define RTBHIP_ET_CONST 6

#define RTBHIP_MAX_JOINTS 32
#define RTBHIP_MAX_ETS 512

/* One elementary transform.  Replaces `struct ET` (core/structs.h:42-56) as built by ET_init
 * (core/fknm.cpp:1182-1239): isjoint <=> kind != RTBHIP_ET_CONST, axis == kind, isflip == flip,
 * jindex == jindex, T == T (but row-major and COPIED -- the reference borrows the NumPy buffer). */
typedef struct rtbhip_et {
    int32_t kind;
    int32_t flip;
    int32_t jindex;
    int32_t reserved;
    double T[16];
} rtbhip_et;

typedef uint64_t rtbhip_chain_t; /* replaces the "ETS" PyCapsule (fknm.cpp:1066-1114) */
typedef uint64_t rtbhip_dyn_t;   /* replaces the "Robot" PyCapsule of frne (frne.c:233-299) */

const char *rtbhip_last_error(void);
int rtbhip_version(void);
int rtbhip_device_count(int *count);
/* Optional: rtbhip_init(n) fails loudly unless at least max(n,1) HIP devices are visible (n = -1: any number);
 * rtbhip_shutdown() frees every cached device table (handles stay valid; tables are re-uploaded on next use).
 * Neither is required: tables are uploaded lazily to the device that is current when a call runs. */
int rtbhip_init(int32_t n_devices);
void rtbhip_shutdown(void);

/* ETS_init (fknm.cpp:1066-1114).  qlim: 2*n doubles, n lows then n highs, in chain joint order, or
 * NULL for the reference defaults [-pi,pi] / [0,1] (robot/ET.py:109-115). */
int rtbhip_chain_create(const rtbhip_et *ets, int32_t m, const double *qlim, rtbhip_chain_t *chain);
/* Product-of-exponentials robots (robot/PoERobot.py: PoERevolute / PoEPrismatic :124-154, PoERobot :157-324):
 * T(q) = exp([S_1] q_1) ... exp([S_n] q_n) T0.  twists is (n,6), row i = (v, w) of the unit joint twist S_i in the base frame
 * (spatialmath Twist3 order; PoERevolute(axis, point): w = axis/|axis|, v = -w x point; PoEPrismatic(axis): w = 0,
 * v = axis/|axis|); T0_16 the end-effector pose at q = 0 (row-major, NULL = identity).  The twists are lowered directly to the
 * chain form every other call of this library runs (csrc/chain.cpp compile_poe) -- the handle then serves
 * rtbhip_fkine / rtbhip_jacob (frame 0: PoERobot.jacob0 :230-250, frame 1: jacobe :252-270) and everything else that takes a
 * chain (hessian, ik_lm, ...), which the reference reaches only through its ETS re-expression (_update_ets :272-324).
 * Joint i reads column i of q.  Non-unit twists and twists with a pitch are refused (RTBHIP_EINVAL). */
int rtbhip_chain_create_poe(const double *twists, int32_t n, const double *T0_16, const double *qlim, rtbhip_chain_t *chain);
/* Make a handle's table resident on `device` (-1: the current device) now.  Tables are otherwise uploaded on a handle's first use on
 * a device (one allocation + one synchronous copy); after the explicit upload every device-pointer call with the handle on that
 * device only enqueues kernels on `stream`, so a sequence of calls can be captured into a hipGraph without a warm-up call.
 * rtbhip_chain_upload also sizes the per-device scheduler state of rtbhip_ik_lm. */
int rtbhip_chain_upload(rtbhip_chain_t chain, int32_t device);
int rtbhip_chain_destroy(rtbhip_chain_t chain);
int rtbhip_chain_info(rtbhip_chain_t chain, int32_t *n, int32_t *m, int32_t *q_width);
/* Row pitch of q (columns per configuration).  A chain created from a branch of a tree robot keeps the robot-wide joint numbers
 * (Robot.ets(start, end), robot/Robot.py:1974-1981: jacob0 / jacobe / fkine of a branch are evaluated on the ROBOT's q); by
 * default the pitch is max(jindex)+1, which is short of robot.n when the branch does not hold the robot's last joint.  Setting it
 * (max(jindex)+1 <= q_width <= 256) lets every branch of one robot read the same (N, robot.n) array. */
int rtbhip_chain_set_q_width(rtbhip_chain_t chain, int32_t q_width);

/* ETS_fkine (fknm.cpp:923-1064 -> _ETS_fkine methods.cpp:318-352): T[i] = base * chain(q[i]) * tool.
 * base16/tool16 may be NULL (identity). */
int rtbhip_fkine(rtbhip_chain_t chain, const double *q, int64_t N, const double *base16,
                 const double *tool16, double *T, int32_t mem, void *stream);

/* ETS_jacob0 / ETS_jacobe (fknm.cpp:785-921 -> methods.cpp:112-316), batched: J is (N,6,n)
 * C-order; frame 0 = jacob0 (expressed in the chain's start frame), 1 = jacobe. */
int rtbhip_jacob(rtbhip_chain_t chain, const double *q, int64_t N, const double *tool16,
                 int32_t frame, double *J, int32_t mem, void *stream);

/* The headline fused op (one chain walk per configuration): T as rtbhip_fkine (base applies to T
 * only -- Robot.jacob0 never sees the base, RobotKinematics.py:158), J as rtbhip_jacob. */
int rtbhip_fkine_jacob(rtbhip_chain_t chain, const double *q, int64_t N, const double *base16,
                       const double *tool16, int32_t frame, double *T, double *J, int32_t mem,
                       void *stream);

/* The same fused op with ONE output array: TJ is (N, 16 + 6n) C-order, row i = [T[i] (4x4 row-major, 16 doubles) | J[i] ((6,n) C-order)].
 * It is what ETS_fkine (fknm.cpp:923-1064) + ETS_jacob0 / ETS_jacobe (fknm.cpp:785-921) return for row i, laid side by side: the
 * 464-byte T||J message of the multi-GPU gather (rtbhip_shard_gather below), and a single write stream for the device
 * (the two-array form writes two, whose relative placement costs up to 14 %: profiles/r04_headline_stores.txt).  Values are bit for bit
 * those of rtbhip_fkine_jacob. */
int rtbhip_fkine_jacob_packed(rtbhip_chain_t chain, const double *q, int64_t N, const double *base16,
                              const double *tool16, int32_t frame, double *TJ, int32_t mem, void *stream);

/* ETS_hessian0 / ETS_hessiane (fknm.cpp:583-783 -> methods.cpp:16-32), batched: H is (N,n,6,n). */
int rtbhip_hessian(rtbhip_chain_t chain, const double *q, int64_t N, const double *tool16,
                   int32_t frame, double *H, int32_t mem, void *stream);

/* ETS_hessian0 / ETS_hessiane in the form the reference binds them, (ets, q, J, tool) WITH the Jacobian supplied
 * (fknm.cpp:583-783: when J is given only _ETS_hessian runs on it, methods.cpp:16-32 -- a pure function of J, so no
 * chain handle is needed): J (N,6,n) -> H (N,n,6,n); J0 gives hessian0, Je gives hessiane.  Device buffers must be
 * 16-byte aligned. */
int rtbhip_hessian_from_jacobian(const double *J, int64_t N, int32_t n, double *H, int32_t mem, void *stream);

/* Robot.manipulability(J=...) (robot/Robot.py:701-905: `if J is not None: w = [mfunc(self, J, q, axes_list)]` :896) and
 * Robot.jacobm(J=..., H=...) (robot/Robot.py:1101-1235: `verifymatrix(J, (6, n))` :1201, `H = self.hessian0(J0=J)` :1206), batched:
 * pure functions of the supplied arrays, so no chain handle.  J is (N,6,n), n = 1..16; m is (N); Jm is (N,n); H is (N,n,6,n) or NULL
 * (then the Hessian of J is formed on the fly, core/methods.cpp:16-32).  axes_mask and method as rtbhip_manipulability.  Device buffers
 * must be 16-byte aligned. */
int rtbhip_manipulability_from_jacobian(const double *J, int64_t N, int32_t n, int32_t axes_mask, int32_t method, double *m, int32_t mem,
                                        void *stream);
int rtbhip_jacobm_from_jacobian(const double *J, const double *H, int64_t N, int32_t n, int32_t axes_mask, double *Jm, int32_t mem,
                                void *stream);

/* fknm.Angle_Axis (fknm.cpp:112-162 -> _angle_axis ik.cpp:241-286; used by tools/p_servo.py:7-43 and IK.py:398),
 * batched: e (N,6) = [Tep.t - Te.t ; angle-axis vector of Tep.R Te.R^T], N = max(nTe, nTep); Te is (nTe,4,4) and Tep
 * (nTep,4,4) row-major, each count either N or 1 (that pose is then used for every pair).  Device buffers must be
 * 16-byte aligned. */
int rtbhip_angle_axis(const double *Te, int64_t nTe, const double *Tep, int64_t nTep, double *e, int32_t mem, void *stream);

/* The error vector of rtb.p_servo (tools/p_servo.py:46-117), batched, same buffers as rtbhip_angle_axis:
 *   method 0 "angle-axis": rtbhip_angle_axis (error in the base frame, p_servo.py:98-99);
 *   method 1 "rpy" -- the reference's DEFAULT (p_servo.py:88-97): eTep = inv(wTe) wTep, e = [eTep.t ; tr2rpy(eTep, order "zyx")]
 *     = (x, y, z, roll, pitch, yaw) seen from the end-effector frame.  tr2rpy belongs to spatialmath-python (>= 1.1.16,
 *     pyproject.toml:22, not vendored); its algorithm is restated in csrc/servo_device.h, singular branch included.
 * The gain and the `arrived` test (p_servo.py:101-108) are one multiply and one sum over e: the caller's. */
int rtbhip_p_servo_error(const double *Te, int64_t nTe, const double *Tep, int64_t nTep, int32_t method, double *e, int32_t mem,
                         void *stream);
/* tools/p_servo.py:46-117 in ONE launch: v (N,6) = diag(gain6) e with e as rtbhip_p_servo_error (method 0 angle-axis, 1 "rpy"), and
 * arrived (N bytes, 0 / 1) = sum|e| < threshold (:113; on e, before the gain).  gain6 is a HOST array of six gains (a scalar gain: six copies). */
int rtbhip_p_servo(const double *Te, int64_t nTe, const double *Tep, int64_t nTep, int32_t method, const double *gain6, double threshold, double *v,
                   uint8_t *arrived, int32_t mem, void *stream);

/* Differential-kinematics consumers computed from the Jacobian while it is still in registers (SURVEY 8f-4;
 * compile-time joint counts up to 16 -- 1..8 at two waves per SIMD, 9..16 at one, the longest with some scratch):
 *   rtbhip_jacob_dot       Robot.jacob0_dot (robot/Robot.py:964-1098, representation=None): Jd (N,6,n) = H(q) . qd,
 *                          frame 0 -> hessian0, 1 -> hessiane; qd is (N, q_width) like q
 *   rtbhip_manipulability  ETS.manipulability (robot/ETS.py:1687-1819): m (N); method 0 "yoshikawa",
 *                          1 "minsingular", 2 "invcondition"; axes_mask bit r = Cartesian row r used
 *                          (63 all, 7 trans, 56 rot)
 *   rtbhip_jacobm          ETS.jacobm / Robot.jacobm (robot/ETS.py:1628-1685, robot/Robot.py:1120-1235): Jm (N,n) */
/* ETS.jacob0_analytical (robot/ETS.py:1562-1626): Ja (N,6,n) = blkdiag(I, A^-1) jacob0, A = rotvelxform of the end-effector
 * rotation; representation 0 "rpy/xyz", 1 "rpy/zyx", 2 "eul", 3 "exp" (conventions of spatialmath-python, see diff_device.h). */
int rtbhip_jacob0_analytical(rtbhip_chain_t chain, const double *q, int64_t N, const double *tool16, int32_t representation,
                             double *Ja, int32_t mem, void *stream);
/* Robot.jacob0_dot with an orientation representation (robot/Robot.py:1065-1098): Jd (N,6,n) = tensordot(H, qd) with H the
 * reference's own numerical Hessian of jacob0_analytical -- spatialmath `numhess`, forward differences with dx = 1e-8
 * (the reference has no closed form for this quantity); representation codes as rtbhip_jacob0_analytical. */
int rtbhip_jacob0_dot_analytical(rtbhip_chain_t chain, const double *q, const double *qd, int64_t N, const double *tool16,
                                 int32_t representation, double *Jd, int32_t mem, void *stream);
int rtbhip_jacob_dot(rtbhip_chain_t chain, const double *q, const double *qd, int64_t N, const double *tool16,
                     int32_t frame, double *Jd, int32_t mem, void *stream);
int rtbhip_manipulability(rtbhip_chain_t chain, const double *q, int64_t N, const double *tool16, int32_t axes_mask,
                          int32_t method, double *m, int32_t mem, void *stream);
int rtbhip_jacobm(rtbhip_chain_t chain, const double *q, int64_t N, const double *tool16, int32_t axes_mask, double *Jm,
                  int32_t mem, void *stream);

/* DHRobot.fkine_all (robot/DHRobot.py:1012-1064) / Robot.fkine_all (robot/Robot.py:638-698), batched: the poses of
 * intermediate frames of the chain.  marks[m] = k (nondecreasing, 0 <= k <= number of transforms, at most 33 marks)
 * makes frame m the product  base * E_1(q) ... E_k(q)  of the first k elementary transforms; out is (N, nmarks, 4, 4). */
int rtbhip_link_frames(rtbhip_chain_t chain, const double *q, int64_t N, const double *base16, const int32_t *marks,
                       int32_t nmarks, double *out, int32_t mem, void *stream);

/* ETS.partial_fkine0 (robot/ETS.py:1821-2013; Robot.partial_fkine0, robot/RobotKinematics.py:456-500): the
 * order-th partial derivative of the forward kinematics, order 3..6 (order 1 is jacob0, order 2 hessian0).
 * out is (N, n^(order-1), 6, n), C order, i.e. (N,n,n,6,n) at order 3 -- the reference's tensor with a leading
 * batch axis.  The lower-order tensors are stream-ordered temporaries (hipMallocAsync on `stream`). */
int rtbhip_partial_fkine0(rtbhip_chain_t chain, const double *q, int64_t N, const double *tool16, int32_t order,
                          double *out, int32_t mem, void *stream);

/* IK_LM_c (fknm.cpp:394-525 -> ik.cpp:19-75,157-209), batched over N targets, LM loop resident
 * on the device.  Tep (N,4,4) row-major; q0 (N,n) or NULL; we6 host or NULL; method 0 chan /
 * 1 wampler / 2 sugihara, or -- the same search loop with the steps of IK_GN_c / IK_NR_c
 * (fknm.cpp:164-392 -> ik.cpp:79-156) -- 3 Gauss-Newton, 4 Newton-Raphson (`lambda` is then
 * pinv_damping; both take the minimum-norm step J^T (J J^T + d^2 I)^-1 e, see ik_device.h); seed keys the counter-based restart generator (the reference uses an
 * unseeded std::rand, ik.cpp:293).  flavour 0 reproduces the C loop (ETS.ik_LM), 1 the Python
 * solver's loop (ETS.ikine_LM, robot/IK.py:297-367: E tested after the step, %-wrap).
 * Outputs: q_out (N,n), success/iters/searches int32 (N), residual (N).  Chains of up to 16 joints (1..8 at two waves per
 * SIMD, 9..12 at one, 13..16 with their normal equations partly in scratch). */
int rtbhip_ik_lm(rtbhip_chain_t chain, const double *Tep, int64_t N, const double *q0,
                 int32_t ilimit, int32_t slimit, double tol, int32_t reject_jl, const double *we6,
                 double lambda, int32_t method, int32_t flavour, uint64_t seed, double *q_out,
                 int32_t *success, int32_t *iters, int32_t *searches, double *residual,
                 int32_t mem, void *stream);

/* The same with the null-space motion of the Python solvers (IK_LM / IK_GN / IK_NR keyword arguments kq, km, ps, pi;
 * robot/IK.py:507-576 `_null_Sigma`, `_calc_qnull`, added to the step at :758, :1015, :1215): joint-limit avoidance with
 * gain 1/kq inside the influence distance pi (minimum distance ps) and manipulability maximisation with gain 1/km,
 * projected into the null space of J.  pi: n influence distances, one per joint (a HOST array: the reference takes a scalar or an
 * array, IK.py:519-520; NULL = 0.3 for every joint).  flavour must be 1.  As in the reference the term is applied only when kq > 0;
 * chains of 6..12 joints; kq > 0 on any other chain returns RTBHIP_ELIMIT (nothing is dropped silently). */
int rtbhip_ik_lm_nullspace(rtbhip_chain_t chain, const double *Tep, int64_t N, const double *q0,
                           int32_t ilimit, int32_t slimit, double tol, int32_t reject_jl, const double *we6,
                           double lambda, int32_t method, int32_t flavour, uint64_t seed,
                           double kq, double km, double ps, const double *pi, double *q_out,
                           int32_t *success, int32_t *iters, int32_t *searches, double *residual,
                           int32_t mem, void *stream);

/* IK_QP (robot/IK.py:1222-1520, behind ETS.ikine_QP robot/ETS.py:2932-3110): the Python solver loop (flavour 1) whose step is the
 * quadratic programme  min 1/2 x^T Q x + c^T x  s.t. [J 1] x = e  over x = (dq, slack), Q = diag(kj 1_n, ks / sum|e| 1_6),
 * c = (-jacobm / km, 0) (IK.py:1437-1497).  Without inequality rows (kq = 0, the reference's default) the QP has the closed
 * form of ik_device.h (a minimum-norm step damped by kj sum|e| / ks); with kq > 0 every joint inside the influence distance pi
 * of a limit adds one row on its own velocity (IK.py:1453-1481) and a primal-dual active-set loop around the same 6x6 solve
 * finds the minimiser.  Both per lane, inside the search scheduler of rtbhip_ik_lm.  km > 0 or kq > 0 need a chain of 6..12
 * joints (RTBHIP_ELIMIT otherwise); pi as in rtbhip_ik_lm_nullspace (n values or NULL; IK.py:1441-1442). */
int rtbhip_ik_qp(rtbhip_chain_t chain, const double *Tep, int64_t N, const double *q0, int32_t ilimit, int32_t slimit, double tol,
                 int32_t reject_jl, const double *we6, uint64_t seed, double kj, double ks, double kq, double km, double ps, const double *pi,
                 double *q_out, int32_t *success, int32_t *iters, int32_t *searches, double *residual, int32_t mem, void *stream);

/* Sharded IK: the restart generator is keyed by (seed, target row, search, joint).  A rank that solves rows [begin, begin + count)
 * of a larger batch (rtbhip_shard_range) calls rtbhip_ik_target_base(begin) first: its targets then draw exactly the start vectors
 * they would draw in one call over the whole batch, so the gathered result does not depend on how the rows were split.  The base
 * is per calling thread and stays in force until set again (0 = the default).  (The reference draws from an unseeded std::rand,
 * ik.cpp:293: nothing to reproduce there.) */
int rtbhip_ik_target_base(int64_t base);

/* The restart vector the device generator yields for (seed, target index, search index, joint):
 * uniform in [qlim_lo, qlim_hi).  Exposed so tests can hand the CPU oracle the same sequence. */
int rtbhip_ik_restart(rtbhip_chain_t chain, uint64_t seed, int64_t target, int32_t search,
                      double *q_n);

/* frne.init (frne.c:233-299): L24 is the (n,24) block of DHRobot._init_rne (DHRobot.py:1342-1358). */
int rtbhip_dyn_create(const double *L24, int32_t n, int32_t mdh, rtbhip_dyn_t *dyn);
int rtbhip_dyn_destroy(rtbhip_dyn_t dyn); /* frne.delete (frne.c:80-103) */
int rtbhip_dyn_upload(rtbhip_dyn_t dyn, int32_t device); /* as rtbhip_chain_upload */

/* frne.frne (frne.c:106-230 -> newton_euler ne.c:62-493), batched: q,qd,qdd,tau are (N,n).
 * grav3 is what frne.frne is handed (already negated by DHRobot.rne, DHRobot.py:1449); fext6 may
 * be NULL (zero wrench).  qd and/or qdd may be NULL (all zeros: what Dynamics.gravload,
 * robot/Dynamics.py:863-922, and Dynamics.itorque, :1407-1465, feed the reference's rne). */
int rtbhip_rne(rtbhip_dyn_t dyn, const double *q, const double *qd, const double *qdd, int64_t N,
               const double *grav3, const double *fext6, double *tau, int32_t mem, void *stream);

/* DHRobot.rne(..., base_wrench=True) -> rne_python (robot/DHRobot.py:1409-1412, 1765-1770), batched: the torques as rtbhip_rne and
 * wbase (N,6) = [R_1 f_1, R_1 n_1], the force and moment the base exerts on link 1, rotated into frame 0 -- what the backward
 * recursion of rtbhip_rne holds when it ends; the moment refers to the origin of frame 0 (standard DH) / of frame 1 (modified DH).
 * (The reference allocates wbase as (N,n), so its own call only works for six-joint robots; here wbase is (N,6) for any n.  Its
 * rne_python contradicts its own frne for modified-DH chains, :1640 / :1711: there the definition is served, with frne's torques,
 * and checked against momentum balance, tests/test_base_wrench_balance.py.)  Served by the run-time-n kernel: correct for every
 * chain, not the fast path. */
int rtbhip_rne_base_wrench(rtbhip_dyn_t dyn, const double *q, const double *qd, const double *qdd, int64_t N,
                           const double *grav3, const double *fext6, double *tau, double *wbase, int32_t mem, void *stream);

/* The Dynamics-mixin terms the reference derives from repeated rne calls, one fused
 * kernel each, all passes of a configuration in one lane:
 *   rtbhip_inertia   Dynamics.inertia  (robot/Dynamics.py:704-763): M (N,n,n); row i = tau for qdd = e_i, qd = 0, g = 0
 *   rtbhip_coriolis  Dynamics.coriolis (:765-861): C (N,n,n), friction removed as nofriction(True, True) does
 *   rtbhip_accel     Dynamics.accel    (:424-509): qdd (N,n) = M^-1 (torque - rne(q, qd, 0)); grav3 in the
 *                    convention of rtbhip_rne (what frne.frne is handed)
 * Chains of up to 16 joints (beyond 10 the per-link state spills and one wave's (n,n) tile takes most of a CU's LDS: served,
 * not fast); longer ones return RTBHIP_ELIMIT. */
int rtbhip_inertia(rtbhip_dyn_t dyn, const double *q, int64_t N, double *M, int32_t mem, void *stream);
int rtbhip_coriolis(rtbhip_dyn_t dyn, const double *q, const double *qd, int64_t N, double *C, int32_t mem,
                    void *stream);
int rtbhip_accel(rtbhip_dyn_t dyn, const double *q, const double *qd, const double *torque, int64_t N,
                 const double *grav3, double *qdd, int32_t mem, void *stream);

/* Robot.rne for ETS robots (robot/Robot.py:1704-1903): inverse dynamics over a tree of LINK
 * GROUPS -- every joint link together with the static links that precede it in link order (:1777-1789).
 * One rtbhip_tree_group per group, in the reference's order (parents before children):
 *   parent  index of the group holding the first member link's parent link, -1 for the base
 *   T       the group's constant transform, row-major 4x4: product of the member links' constant parts
 *           (static links' A(), then the joint link's Ts, Link.A robot/Link.py:1642-1651)
 *   kind    joint axis 0..5 (Rx,Ry,Rz,tx,ty,tz); flip as ET.flip; jindex = column of q / qd / qdd
 *   m,h,I   the group's spatial inertia about the group-frame origin: mass, first moment sum(m r), rotational
 *           inertia (xx,yy,zz,xy,xz,yz).  The reference's own value is the plain sum of the member links'
 *           SpatialInertia(m, r) (:1793-1800), i.e. I = sum m (|r|^2 1 - r r^T) and no inertia tensor.
 * Up to 24 groups (17..24 with the per-group state partly in scratch).
 * rtbhip_tree_rne: q,qd,qdd (N,ng); gravity3 = the robot's gravity vector (e.g. 0,0,-9.81; the base is
 * accelerated by its negative, :1804-1807); tau (N,ng), column j = group j as the reference's Q[:, j]. */
typedef struct rtbhip_tree_group {
    int32_t parent, kind, flip, jindex;
    double T[16];
    double m, h[3], I[6];
} rtbhip_tree_group;
typedef uint64_t rtbhip_tree_t;
int rtbhip_tree_create(const rtbhip_tree_group *groups, int32_t ng, rtbhip_tree_t *tree);
int rtbhip_tree_destroy(rtbhip_tree_t tree);
int rtbhip_tree_upload(rtbhip_tree_t tree, int32_t device); /* as rtbhip_chain_upload */
int rtbhip_tree_rne(rtbhip_tree_t tree, const double *q, const double *qd, const double *qdd, int64_t N,
                    const double *gravity3, double *tau, int32_t mem, void *stream);

/* The Dynamics-mixin terms of an ETS robot -- Dynamics.inertia / coriolis / accel (robot/Dynamics.py:704-763, 765-861, 424-509), which the
 * reference builds from n, n + n(n-1)/2 and n + 1 calls of Robot.rne per configuration -- one fused kernel each, every pass of a
 * configuration in one lane:  M (N,n,n), row i = rne(q, 0, e_i, gravity 0);  C (N,n,n);  qdd (N,n) = M^-1 (torque - rne(q, qd, 0)) with
 * gravity3 as rtbhip_tree_rne takes it.  gravload / itorque are rtbhip_tree_rne with NULL qd / qdd.  Robots of up to 20 joints
 * (RTBHIP_ELIMIT beyond). */
int rtbhip_tree_inertia(rtbhip_tree_t tree, const double *q, int64_t N, double *M, int32_t mem, void *stream);
int rtbhip_tree_coriolis(rtbhip_tree_t tree, const double *q, const double *qd, int64_t N, double *C, int32_t mem, void *stream);
int rtbhip_tree_accel(rtbhip_tree_t tree, const double *q, const double *qd, const double *torque, int64_t N,
                      const double *gravity3, double *qdd, int32_t mem, void *stream);

/* Mixed fleet (BASELINE config 5): n_chains independent chains, each with its own batch; one
 * launch walks all of them (block -> chain map).  q[c] is (N[c], q_width_c), T[c] (N[c],4,4),
 * J[c] (N[c],6,n_c).  The pointer tables themselves are HOST arrays. */
int rtbhip_fleet_fkine_jacob(const rtbhip_chain_t *chains, int32_t n_chains,
                             const double *const *q, const int64_t *N, int32_t frame,
                             double *const *T, double *const *J, int32_t mem, void *stream);

/* ... with packed rows: TJ[c] is (N[c], 16 + 6 n_c), row = [T | J] as rtbhip_fkine_jacob_packed. */
int rtbhip_fleet_fkine_jacob_packed(const rtbhip_chain_t *chains, int32_t n_chains,
                                    const double *const *q, const int64_t *N, int32_t frame,
                                    double *const *TJ, int32_t mem, void *stream);

/* Pinned host memory for the RTBHIP_MEM_HOST boundary.  Host arrays are streamed through the device in row chunks that
 * alternate between two persistent slots (stream + device buffer + pinned staging each), so H2D / kernel of one chunk overlap
 * the D2H of the previous one and nothing is allocated per call.  Pageable arrays are copied through the pinned staging by a few
 * copy threads; arrays that are ALREADY pinned (these blocks, hipHostMalloc, hipHostRegister) are the DMA endpoints themselves --
 * results land directly in the caller's array.  Blocks are cached between uses (pinning is the expensive part; cap
 * RTBHIP_PINNED_CACHE_MB, default 1024); rtbhip_shutdown() returns them. */
int rtbhip_host_alloc(uint64_t bytes, void **ptr);
int rtbhip_host_free(void *ptr);
/* Idle cached memory back to the driver / the OS: device staging blocks of the host-pointer calls that are not row-pipelined (IK, the
 * dynamics terms, hessian_from_jacobian, the fleet; cached per device up to RTBHIP_DEVICE_CACHE_MB, default 512 -- a block released
 * above that goes back at once) down to keep_device_bytes per device, pinned host blocks (RTBHIP_PINNED_CACHE_MB, default 1024) down
 * to keep_pinned_bytes.  rtbhip_trim(0, 0) keeps nothing that is not in use. */
int rtbhip_trim(uint64_t keep_device_bytes, uint64_t keep_pinned_bytes);

/* Multi-GPU partition helper: contiguous row block [begin, begin+count) of rank `rank` out of
 * `world` (the first N % world ranks get one extra row).  Pure host arithmetic. */
int rtbhip_shard_range(int64_t N, int32_t rank, int32_t world, int64_t *begin, int64_t *count);

/* ---- The ONE exchange of the path: the gather of the ranks' output shards, on RCCL over xGMI. -------------------------------
 * The reference has no counterpart (it is single-threaded: the batch loop of ETS_fkine fknm.cpp:1038-1052, the Python row loops around
 * ETS_jacob0 / frne / IK_LM_c); this is the multi-GPU half of the replacement boundary.  The data path has NO collective -- rows are
 * independent, every rank evaluates rtbhip_shard_range(N, rank, world) and leaves its rows in HBM; a consumer that wants them in one place
 * calls rtbhip_shard_gather once.  librccl.so is dlopen'ed on first use (RTBHIP_RCCL_LIB overrides the name); nothing here needs PyTorch.
 *
 * Communicators.  One process per GPU (the usual form): rank 0 calls rtbhip_shard_comm_id, ships the 128 bytes to the other ranks by any
 * means, every rank then calls rtbhip_shard_comm_create on its current device.  One process driving several GPUs: rtbhip_shard_comm_create_all
 * (ndev communicators, devices[i] or 0..ndev-1), and the per-device rtbhip_shard_gather calls of one exchange go between rtbhip_shard_group(1)
 * and rtbhip_shard_group(0) (ncclGroupStart / ncclGroupEnd).  comm = NULL is allowed for world = 1 (no RCCL is loaded: a device copy). */
typedef void *rtbhip_comm_t;
int rtbhip_shard_comm_id(void *id128);
int rtbhip_shard_comm_create(const void *id128, int32_t world, int32_t rank, rtbhip_comm_t *comm);
int rtbhip_shard_comm_create_all(int32_t ndev, const int32_t *devices, rtbhip_comm_t *comms);
int rtbhip_shard_comm_destroy(rtbhip_comm_t comm);
/* world size and rank the communicator itself reports, and the RCCL version (any out pointer may be NULL; comm may be NULL for the version) */
int rtbhip_shard_comm_info(rtbhip_comm_t comm, int32_t *world, int32_t *rank, int32_t *rccl_version);
int rtbhip_shard_group(int32_t begin);
/* Gather: this rank holds `rows` = its rtbhip_shard_range count of the N global rows, each row_bytes long (56 for tau of the Panda, 464 for
 * a T||J row of rtbhip_fkine_jacob_packed, ...), in DEVICE memory at `local`.  root >= 0: rank `root` receives all N rows in `out` (global
 * row order; `out` may be NULL elsewhere) -- each shard crosses one xGMI link once.  root = -1: every rank receives them (all-gather: world
 * times the traffic and receive memory).  Equal shards are one ncclGather / ncclAllGather; ragged ones one group of ncclSend / ncclRecv
 * straight into place (no padding).  Enqueued on `stream` (a stream of the communicator's device); returns without waiting. */
int rtbhip_shard_gather(rtbhip_comm_t comm, const void *local, int64_t rows, int64_t row_bytes, int64_t N, int32_t world, int32_t rank,
                        int32_t root, void *out, void *stream);

/* Device memory, copies and streams for a host WITHOUT a HIP binding of its own (C, Go over cgo, Java over JNI ...): enough to keep inputs and
 * results resident (RTBHIP_MEM_DEVICE) and to drive one stream per GPU.  A consumer that already holds device pointers never needs them.
 * rtbhip_device_copy kind: 1 host -> device, 2 device -> host, 3 device -> device; stream = NULL waits for the copy, else it is enqueued. */
int rtbhip_device_alloc(int32_t device, uint64_t bytes, void **ptr);
int rtbhip_device_free(void *ptr);
int rtbhip_device_copy(void *dst, const void *src, uint64_t bytes, int32_t kind, void *stream);
int rtbhip_stream_create(int32_t device, void **stream);
int rtbhip_stream_destroy(void *stream);
int rtbhip_stream_sync(void *stream);

/* Which physical GPU is HIP device `device` of this process: its PCI bus id ("0000:05:00.0", a 32-byte buffer) and 16-byte UUID (either may be
 * NULL).  A multi-rank run prints these per rank, so that "did every rank get its own GPU?" is answered by the output itself. */
int rtbhip_device_identity(int32_t device, char *pci_bus_id32, unsigned char *uuid16);

/* Launch-geometry report for the last kernel a call on this thread enqueued (diagnostics). */
int rtbhip_last_launch(int32_t *grid, int32_t *block, int32_t *lds_bytes);

/* Tuning knobs for benchmarking (A/B of launch geometry / store paths); process-wide; unknown keys are ignored.  Results never
 * depend on them (every setting is covered by bit-equality tests); the defaults are the measured best.  IK scheduler:
 *   "ik_flat" 0 | 1 | 2       flat schedule (search ranges cut into chunks, (target, chunk) items drawn from one counter by whichever wave has
 *                             idle lanes): never / when the batch is resident at once (default) / always; "ik_flat_l0", "ik_flat_len": searches
 *                             in a target's first / in every later chunk (4, 8)
 *   "ik_share" 0 | 1 | 2      cross-wave sharing of search ranges: never (default) / when the batch is resident at once / always
 *   "ik_donate_after" k       ... ranges are cut only from targets with k failed searches (default 3)
 *   "ik_phased" 0 | 1 | 2     phased schedule (first searches, then compacted work lists): never (default) / automatic / always
 *   "ik_fresh_pct" p, "ik_pass_mask" m, "ik_waves_per_cu" w, "ik_spec_policy" 0 | 1     pacing of the per-wave scheduler
 *   "ik_sig" 1 | 0            a chain takes the kernel instantiated for its constants' structure (built in: Panda, UR; any other: compiled at run time, see
 *                             rtbhip_jit_* below) / always the general kernel -- the same bits either way
 *   "rne_sig", "tree_sig" 1 | 0   the same switch for the dynamics kernels: a DH link table (built in: Panda, Puma560) / a link tree (built in: UR3 / 5 / 10, the
 *                             Interbotix arms, Fetch, Mico, any serial arm of up to 8 revolute joints); every other robot: its run-time instantiation
 * Others: "coalesced", "reg", "tiles_per_wave", "hess_mode" (fkine / Jacobian / Hessian store paths), "rne_tiles_per_wave",
 *         "partial3" 1 | 0 (order-3 partial_fkine0 on workgroups that own whole configurations / on the general kernel),
 *         "partial3_fused" 1 | 0 (that kernel forms the Hessians from the Jacobians it stages / reads a Hessian tensor written by a launch of its own),
 *         "rne_persist", "rne_wpb", "ik_unit_we" (A/B forms that measured slower and are off),
 *         "host_chunk_kb" (host-pointer pipeline), "shard_p2p" 1 | 0 (rtbhip_shard_gather: the grouped send / receive form for equal shards too). */
/* Measurement aid, no reference counterpart: one launch of a plain streaming kernel that reads `read_doubles` doubles from `src` and writes
 * `write_doubles` doubles to `dst` (device pointers, 4 KiB-aligned; whole 4 KiB pages are moved, the tails are left alone) -- the memory rate this GPU delivers for a given read / write mix, which
 * bench.py reports next to the headline kernel's rate. */
int rtbhip_stream_probe(const double *src, int64_t read_doubles, double *dst, int64_t write_doubles, void *stream);

int rtbhip_tune(const char *key, int32_t value);

/* ---- run-time instantiation of the structure-signature kernels (csrc/jit.cpp).
 * The reference serves every robot through ONE general code path at one cost (core/methods.cpp:318-352 _ETS_fkine, core/ik.cpp:19-75 _IK_LM,
 * core/ne.c:62-493 newton_euler, robot/Robot.py:1704-1903 Robot.rne); here the straight-line forms of k_ik / k_rne / k_dyn / k_tree_rne /
 * k_tree_dyn are instantiated per robot structure.  A robot whose structure words match no instantiation built into the library gets its own,
 * compiled from the same sources by hipRTC on a worker thread (requested at rtbhip_chain_create / rtbhip_dyn_create / rtbhip_tree_create and
 * again at the first launch); until the code object is ready launches take the general kernel, which returns the same bits.  Code objects
 * are cached under $RTBHIP_JIT_CACHE (default $XDG_CACHE_HOME/rtbhip/jit or ~/.cache/rtbhip/jit; "-" disables the disk cache).
 * rtbhip_tune("jit", 0 | 1 | 2): off / asynchronous (default) / a launch waits for its instantiation; "jit_eager" 0: nothing is requested at create. */
typedef struct rtbhip_jit_info {
    int32_t available;              /* libhiprtc.so was found (otherwise the general kernels serve every robot without a built-in instantiation) */
    int32_t mode;                   /* rtbhip_tune("jit") */
    int64_t requested;              /* distinct instantiations asked for in this process */
    int64_t compiled;               /* ... compiled by hipRTC */
    int64_t disk_hits;              /* ... read from the disk cache instead */
    int64_t failed;                 /* ... that failed to compile or load (last_error) */
    int64_t pending;                /* ... queued or being compiled now */
    int64_t launches;               /* launches served by a run-time instantiation */
    int64_t general_while_pending;  /* launches that took the general kernel because their instantiation was not ready yet */
    double compile_seconds;         /* total / longest hipRTC time */
    double compile_seconds_max;
    int32_t sources;                /* embedded source files */
    char source_digest[20];         /* first 16 hex digits of their sha256 (part of the cache key) */
    char last_error[512];
} rtbhip_jit_info;
int rtbhip_jit_stats(rtbhip_jit_info *out);
/* Blocks until no instantiation is queued or being compiled (timeout_s < 0: no limit).  Returns RTBHIP_OK, or 1 when the time ran out. */
int rtbhip_jit_wait(double timeout_s);
/* Compile one instantiation NOW on the calling thread for a named architecture (no device needed: hipRTC cross-compiles) -- what the build
 * check and the tests use to prove that the embedded sources compile under hipRTC.  unit: "ik_kernels.hip", "rne_kernels.hip", "dyn_kernels.hip",
 * "tree_kernels.hip", "tree_dyn_kernels.hip"; expr: a C++ name expression, e.g. "rtbhip::k_rne<7, true, true, 0xe00047a99a2c7ea9ull>". */
int rtbhip_jit_compile(const char *unit, const char *expr, const char *arch, int64_t *code_bytes, double *seconds, int32_t *from_disk);
/* Ask for ALL of a handle's run-time instantiations now (worker thread; returns at once).  kind: 0 chain, 1 dyn, 2 tree.  The *_create calls ask
 * by themselves once the library has used a device in this process -- they never initialise the HIP runtime on their own (a process may build
 * its robots and fork its GPU workers afterwards); this call does, and so does the first launch. */
int rtbhip_jit_prepare(int32_t kind, uint64_t handle);
/* The name expressions a handle's run-time instantiations are requested under, '\n'-separated, into buf (truncated to cap - 1 characters);
 * kind: 0 chain (k_ik), 1 dyn (k_rne ...), 2 tree (k_tree_rne ...).  Empty when the structure has a built-in instantiation or none applies. */
int rtbhip_jit_names(int32_t kind, uint64_t handle, char *buf, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* RTBHIP_H */
