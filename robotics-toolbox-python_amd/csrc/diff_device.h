// diff_device.h -- per-lane differential-kinematics consumers of a finished Jacobian held in registers
// (slot r*NJ + j = row r of column j, as kin_reg.h leaves it):
//   jacob_dot      d/dt J = H(q) . qd   (reference Robot.jacob0_dot robot/Robot.py:964-1098:
//                  np.tensordot(hessian0(q), qd, (0, 0)); the Hessian blocks of core/methods.cpp:16-32 are
//                  formed on the fly and never stored)
//   manipulability Yoshikawa's measure sqrt|det(J_a J_a^T)|, or |det J_a| when J_a is square
//                  (reference ETS.manipulability robot/ETS.py:1687-1819, `yoshikawa` :1780-1787)
//   jacobm         manipulability Jacobian  Jm[i] = m vec(J H_i^T)^T vec((J J^T)^-1)
//                  (reference ETS.jacobm robot/ETS.py:1628-1685, Robot.jacobm robot/Robot.py:1120-1215)
// `axes` is a 6-bit row mask (bit r = Cartesian row r used): excluded rows are handled by replacing
// their row/column of J J^T by the identity (determinant and the remaining block of the inverse are
// unchanged), so every index stays static.
#pragma once
#include "kin_reg.h"
#include "ldl.h"

namespace rtbhip {

// H[j', :, i] as the reference fills it (methods.cpp:16-32): for j' <= i  (w_j' x v_i ; w_j' x w_i),
// for j' > i  (w_i x v_j' ; 0).
// (a b - c d) + (e f - g h) as one fixed sequence:  fma(e, f, fma(-g, h, fma(a, b, -round(c d))))
RTB_HD double cross_sum(double a, double b, double c, double d, double e, double f, double g, double h)
{
#pragma clang fp contract(off)
    return __builtin_fma(e, f, __builtin_fma(-g, h, __builtin_fma(a, b, -(c * d))));
}
// sum_k a(k) b(k), k = 0 .. N - 1, as round(a0 b0) followed by N - 1 fused multiply-adds in index order
template <int N, class A, class B>
RTB_HD double chain_dot(A a, B b)
{
#pragma clang fp contract(off)
    double s = a(0) * b(0);
#pragma unroll
    for (int k = 1; k < N; ++k) s = __builtin_fma(a(k), b(k), s);
    return s;
}
// Written with running sums instead of the pair loop the definition suggests (28 + 21 cross products for 7 joints):
//   Jd_v[i] = sum_{j<=i} qd_j (w_j x v_i) + sum_{j>i} qd_j (w_i x v_j) = W_i x v_i + w_i x V_i,   Jd_w[i] = sum_{j<=i} qd_j (w_j x w_i) = W_i x w_i
// with W_i = sum_{j<=i} qd_j w_j (a prefix sum) and V_i = sum_{j>i} qd_j v_j (a suffix sum): 6 n fused multiply-adds for the sums and three cross
// products per joint -- O(n) where the pair loop is O(n^2); the same quantity to rounding (round 6: 1 692 -> ~560 executed instructions per configuration).
template <int NJ>
RTB_HD void jacob_dot(const double (&jac)[6 * NJ], const double (&qd)[NJ], double (&jd)[6 * NJ])
{
    double Vx[NJ], Vy[NJ], Vz[NJ];          // V_i, i = NJ - 1 .. 0
    double sx = 0.0, sy = 0.0, sz = 0.0;
#pragma unroll
    for (int i = NJ - 1; i >= 0; --i) {
        Vx[i] = sx; Vy[i] = sy; Vz[i] = sz;
        sx = fmax_(qd[i], jac[i], sx); sy = fmax_(qd[i], jac[NJ + i], sy); sz = fmax_(qd[i], jac[2 * NJ + i], sz);
    }
    double Wx = 0.0, Wy = 0.0, Wz = 0.0;
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const double vix = jac[i], viy = jac[NJ + i], viz = jac[2 * NJ + i];
        const double wix = jac[3 * NJ + i], wiy = jac[4 * NJ + i], wiz = jac[5 * NJ + i];
        Wx = fmax_(qd[i], wix, Wx); Wy = fmax_(qd[i], wiy, Wy); Wz = fmax_(qd[i], wiz, Wz);
        // (every sum of products written out -- kin_device.h, mix_pp: a structure instantiation of the walk must not change what the compiler fuses here)
        jd[i] = cross_sum(Wy, viz, Wz, viy, wiy, Vz[i], wiz, Vy[i]);
        jd[NJ + i] = cross_sum(Wz, vix, Wx, viz, wiz, Vx[i], wix, Vz[i]);
        jd[2 * NJ + i] = cross_sum(Wx, viy, Wy, vix, wix, Vy[i], wiy, Vx[i]);
        jd[3 * NJ + i] = mix_pm(Wy, wiz, Wz, wiy);
        jd[4 * NJ + i] = mix_pm(Wz, wix, Wx, wiz);
        jd[5 * NJ + i] = mix_pm(Wx, wiy, Wy, wix);
    }
}

// B = J_a J_a^T embedded in 6x6 (excluded rows/columns -> identity)
template <int NJ>
RTB_HD void jjt_masked(const double (&jac)[6 * NJ], int axes, double (&B)[6][6])
{
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const bool ur = (axes >> r) & 1;
#pragma unroll
        for (int c = 0; c <= r; ++c) {
            const bool uc = (axes >> c) & 1;
            double s = chain_dot<NJ>([&](int k) { return jac[r * NJ + k]; }, [&](int k) { return jac[c * NJ + k]; });
            const double v = (ur && uc) ? s : (r == c ? 1.0 : 0.0);
            B[r][c] = v;
            B[c][r] = v;
        }
    }
}

template <int NJ>
RTB_HD double manipulability_yoshikawa(const double (&jac)[6 * NJ], int axes)
{
    if (NJ == 6 && (axes & 63) == 63) {          // square: |det J| (ETS.py:1782-1784)
        double a[6][6];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) a[r][c] = jac[r * NJ + (c < NJ ? c : 0)];
        return fabs(det_lu<6>(a));
    }
    double B[6][6];
    jjt_masked<NJ>(jac, axes, B);
    // ETS.py:1786-1787.  J_a J_a^T is symmetric positive semi-definite: its determinant is the product of the pivots of an LDL^T factorisation
    // without pivoting (ldl.h: det_psd) -- ~90 instructions where the row-exchanging LU that numpy.linalg.det runs costs ~350 as straight-line
    // selects.  The same number to rounding; an exactly singular matrix (the arm stretched out at q = 0) gives an exact 0 either way.
    return sqrt(fabs(det_psd<6>(B)));
}

// Smallest singular value of J_a (mode 1, ETS.py:1793-1796 `minsingular`) or 1/cond_2(J_a) = s_min / s_max
// (mode 2, :1789-1791 `condition`), from the eigenvalues of the smaller Gram matrix: J_a J_a^T when the row
// count does not exceed the joint count, else J_a^T J_a.
template <int NJ>
RTB_HD double manipulability_singular(const double (&jac)[6 * NJ], int axes, int mode)
{
    int rows = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r) rows += (axes >> r) & 1;
    double lo = 1e300, hi = 0.0;
    if (rows <= NJ) {
        double B[6][6];
        jjt_masked<NJ>(jac, axes, B);
        jacobi_eigenvalues<6>(B);
#pragma unroll
        for (int r = 0; r < 6; ++r)
            if ((axes >> r) & 1) { lo = B[r][r] < lo ? B[r][r] : lo; hi = B[r][r] > hi ? B[r][r] : hi; }
    } else {
        double G[NJ][NJ];
#pragma unroll
        for (int i = 0; i < NJ; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                const double s = chain_dot<6>([&](int r) { return ((axes >> r) & 1) ? jac[r * NJ + i] : 0.0; }, [&](int r) { return jac[r * NJ + j]; });
                G[i][j] = s; G[j][i] = s;
            }
        jacobi_eigenvalues<NJ>(G);
#pragma unroll
        for (int i = 0; i < NJ; ++i) { lo = G[i][i] < lo ? G[i][i] : lo; hi = G[i][i] > hi ? G[i][i] : hi; }
    }
    lo = lo < 0.0 ? 0.0 : lo;
    // (an all-zero selected block -- prismatic joints asked for their rotational part -- has cond = inf: 1 / cond = 0, ETS.py:1789-1791.  A block that
    //  is zero but for rounding counts as zero: a tool translation exactly along the last joint's axis leaves 1e-17 in  z x (p_e - p)  where the
    //  reference's column formula multiplies exact zeros -- largest singular value below 1e-14, profiles/r06_aj_fuzz_more.txt)
    return mode == 1 ? sqrt(lo) : (hi > 1e-28 ? sqrt(lo / hi) : 0.0);
}

// Analytical Jacobian (ETS.jacob0_analytical robot/ETS.py:1562-1626): Ja = blkdiag(I, A^-1) J0 with A the map from the
// rates of the orientation parameters Gamma to the angular velocity (spatialmath-python 1.1.x `rotvelxform(R, inverse=True,
// full=True)`, a third-party dependency absent from the reference tree; its conventions, restated):
//   rep 0 "rpy/xyz"  Gamma = (roll, pitch, yaw), R = Rx(yaw) Ry(pitch) Rz(roll)      (tr2rpy order "xyz")
//   rep 1 "rpy/zyx"  Gamma = (roll, pitch, yaw), R = Rz(yaw) Ry(pitch) Rx(roll)      (tr2rpy order "zyx")
//   rep 2 "eul"      Gamma = (phi, theta, psi),  R = Rz(phi) Ry(theta) Rz(psi)       (tr2eul)
//   rep 3 "exp"      Gamma = theta * axis,       R = exp([Gamma]x)                    (trlog)
// For a product of axis rotations R = R_a(x1) R_b(x2) R_c(x3):  omega = a x1' + R_a b x2' + R_a R_b c x3', so the
// columns of A are read off the pose itself (R_a R_b c is a column of R; R_a b needs one sine / cosine pair, a ratio
// of entries of R) and A^-1 is a 3x3 adjugate.  For the rotation vector A is the left Jacobian of SO(3) and
// A^-1 = I - [Gamma]x / 2 + (1/theta^2 - (1 + cos theta) / (2 theta sin theta)) [Gamma]x^2.
RTB_HD void rotvel_inverse(const Pose &P, int rep, double (&Ai)[3][3])
{
    // (no contraction: every operation below is rounded on its own, in whichever instantiation of whichever compiler -- a structure instantiation compiled
    //  at run time chose other multiply-add pairs of the rotation-vector branch to fuse than the built-in general kernel did: 1e-15 apart, 2.7 % of entries)
#pragma clang fp contract(off)
    if (rep == 3) {
        const double lx = P.r21 - P.r12, ly = P.r02 - P.r20, lz = P.r10 - P.r01;      // 2 sin(theta) axis
        const double nrm = sqrt(lx * lx + ly * ly + lz * lz), tr = P.r00 + P.r11 + P.r22;
        const double th = atan2(nrm, tr - 1.0);
        const double k = nrm > 1e-12 ? th / nrm : 0.5;
        const double gx = k * lx, gy = k * ly, gz = k * lz;
        double c;
        if (th < 1e-4) c = 1.0 / 12.0 + th * th / 720.0;
        else c = 1.0 / (th * th) - (1.0 + cos(th)) / (2.0 * th * sin(th));
        // [g]x^2 = g g^T - |g|^2 I
        const double g2 = gx * gx + gy * gy + gz * gz;
        Ai[0][0] = 1.0 + c * (gx * gx - g2); Ai[0][1] = 0.5 * gz + c * gx * gy;   Ai[0][2] = -0.5 * gy + c * gx * gz;
        Ai[1][0] = -0.5 * gz + c * gx * gy;  Ai[1][1] = 1.0 + c * (gy * gy - g2); Ai[1][2] = 0.5 * gx + c * gy * gz;
        Ai[2][0] = 0.5 * gy + c * gx * gz;   Ai[2][1] = -0.5 * gx + c * gy * gz;  Ai[2][2] = 1.0 + c * (gz * gz - g2);
        return;
    }
    double A[3][3];   // A[row][column], columns in Gamma order
    if (rep == 0) {
        const double cp = sqrt(P.r12 * P.r12 + P.r22 * P.r22);          // cos(pitch) >= 0
        const double cy = P.r22 / cp, sy = -P.r12 / cp;                  // yaw = -atan2(R12, R22)
        A[0][0] = P.r02; A[1][0] = P.r12; A[2][0] = P.r22;               // roll: Rx Ry z = third column of R
        A[0][1] = 0.0;   A[1][1] = cy;    A[2][1] = sy;                  // pitch: Rx(yaw) y
        A[0][2] = 1.0;   A[1][2] = 0.0;   A[2][2] = 0.0;                 // yaw: x
    } else if (rep == 1) {
        const double cp = sqrt(P.r00 * P.r00 + P.r10 * P.r10);
        const double cy = P.r00 / cp, sy = P.r10 / cp;                   // yaw = atan2(R10, R00)
        A[0][0] = P.r00; A[1][0] = P.r10; A[2][0] = P.r20;               // roll: Rz Ry x = first column of R
        A[0][1] = -sy;   A[1][1] = cy;    A[2][1] = 0.0;                 // pitch: Rz(yaw) y
        A[0][2] = 0.0;   A[1][2] = 0.0;   A[2][2] = 1.0;                 // yaw: z
    } else {
        const double st = sqrt(P.r02 * P.r02 + P.r12 * P.r12);          // sin(theta) >= 0
        const double cf = st > 0.0 ? P.r02 / st : 1.0, sf = st > 0.0 ? P.r12 / st : 0.0;   // phi = atan2(R12, R02), 0 when singular
        A[0][0] = 0.0;   A[1][0] = 0.0;   A[2][0] = 1.0;                 // phi: z
        A[0][1] = -sf;   A[1][1] = cf;    A[2][1] = 0.0;                 // theta: Rz(phi) y
        A[0][2] = P.r02; A[1][2] = P.r12; A[2][2] = P.r22;               // psi: Rz Ry z = third column of R
    }
    const double c00 = A[1][1] * A[2][2] - A[1][2] * A[2][1], c01 = A[1][2] * A[2][0] - A[1][0] * A[2][2], c02 = A[1][0] * A[2][1] - A[1][1] * A[2][0];
    const double det = A[0][0] * c00 + A[0][1] * c01 + A[0][2] * c02;
    const double id = 1.0 / det;
    Ai[0][0] = c00 * id; Ai[1][0] = c01 * id; Ai[2][0] = c02 * id;
    Ai[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) * id; Ai[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) * id; Ai[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) * id;
    Ai[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) * id; Ai[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) * id; Ai[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) * id;
}

template <int NJ>
RTB_HD void jacob_analytical(const Pose &P, const double (&jac)[6 * NJ], int rep, double (&ja)[6 * NJ])
{
    double Ai[3][3];
    rotvel_inverse(P, rep, Ai);
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        ja[i] = jac[i]; ja[NJ + i] = jac[NJ + i]; ja[2 * NJ + i] = jac[2 * NJ + i];
        const double wx = jac[3 * NJ + i], wy = jac[4 * NJ + i], wz = jac[5 * NJ + i];
        ja[3 * NJ + i] = dot3x(Ai[0][0], wx, Ai[0][1], wy, Ai[0][2], wz);
        ja[4 * NJ + i] = dot3x(Ai[1][0], wx, Ai[1][1], wy, Ai[1][2], wz);
        ja[5 * NJ + i] = dot3x(Ai[2][0], wx, Ai[2][1], wy, Ai[2][2], wz);
    }
}

// Rate of the analytical Jacobian as the reference computes it (kin_kernels.hip, mode kDiffAnalyticalDot): forward differences
// of jacob0_analytical, dx = 1e-8, contracted with qd.  n + 1 chain walks; the loop over joints is a real loop (one copy of
// the walk), so the chain pointers are re-derived per pass on the device -- hoisted, its scalar loads overflow the SGPR file.
template <int NJ, class CV, class TL>
RTB_HD void jacob_analytical_dot(const CV &cv, TL tail, const double (&qv)[NJ], const double (&v)[NJ], int rep, double (&jd)[6 * NJ])
{
    constexpr double dx = 1e-8;                               // spatialmath.base.numhess default
    Pose P;
    double jac[6 * NJ], ja0[6 * NJ];
    reg_core<NJ, true>(cv, tail, 0, qv, P, jac);
    jacob_analytical<NJ>(P, jac, rep, ja0);
#pragma unroll
    for (int k = 0; k < 6 * NJ; ++k) jd[k] = 0.0;
#pragma nounroll
    for (int i = 0; i < NJ; ++i) {
        double q2[NJ], vi = 0.0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            q2[j] = qv[j] + (j == i ? dx : 0.0);              // x + I[:, i] * dx
            vi = j == i ? v[j] : vi;
        }
        double ja[6 * NJ];
        CV cvi = cv;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+s"(cvi.seg), "+s"(cvi.jmeta));
#endif
        reg_core<NJ, true>(cvi, tail, 0, q2, P, jac);
        jacob_analytical<NJ>(P, jac, rep, ja);
#pragma unroll
        for (int k = 0; k < 6 * NJ; ++k) jd[k] += ((ja[k] - ja0[k]) / dx) * vi;   // Hi = (Ji - J0) / dx ; Jd += Hi * qd[i]
    }
}

// jacobm given the LDL^T factorisation (B, dinv; pivots dval) of the masked J J^T.
//   m = sqrt|det(J_a J_a^T)| = sqrt|prod dval|  (Robot.py:1216-1222: the factorisation is there already -- rounds 1-5 formed J J^T a second time and ran
//   the row-exchanging LU on it);
//   Jm[i] = m sum_{b,k} H[i,b,k] G[b,k] with G = (J_a J_a^T)^-1 J_a and the Hessian blocks of methods.cpp:16-32.  With Ga_k / Gb_k the translational /
//   rotational halves of column k of G and the triple product  (a x b) . c = a . (b x c):
//       sum_{k>=i} (w_i x v_k) . Ga_k + (w_i x w_k) . Gb_k  =  w_i . S_i,   S_i = sum_{k>=i} (v_k x Ga_k + w_k x Gb_k)      (a suffix sum)
//       sum_{k<i}  (w_k x v_i) . Ga_k                       =  v_i . P_i,   P_i = sum_{k<i} (Ga_k x w_k)                    (a prefix sum)
//   -- three cross products per joint instead of one or two per joint PAIR (49 blocks for 7 joints): 2 732 -> ~1 500 executed instructions.
template <int NJ>
RTB_HD void jacobm_factored(const double (&jac)[6 * NJ], int axes, const double (&B)[6][6], const double (&dval)[6], const double (&dinv)[6], double (&jm)[NJ])
{
    double det = dval[0];
#pragma unroll
    for (int r = 1; r < 6; ++r) det *= dval[r];
    // (a square J_a: |det J| itself, ETS.py:1782-1784 -- near a singularity it keeps the digits the square root of det(J J^T) loses)
    const double m = (NJ == 6 && (axes & 63) == 63) ? manipulability_yoshikawa<NJ>(jac, axes) : sqrt(fabs(det));
    double cx[NJ], cy[NJ], cz[NJ];          // v_k x Ga_k + w_k x Gb_k
    double px = 0.0, py = 0.0, pz = 0.0;    // P_k, consumed as it grows (no array of prefixes is kept)
#pragma unroll
    for (int k = 0; k < NJ; ++k) {
        // column k of G = (J_a J_a^T)^-1 J_a (rows outside `axes` come out zero)
        double g[6], x[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) g[r] = ((axes >> r) & 1) ? jac[r * NJ + k] : 0.0;
        ldl_backsolve<6>(B, dinv, g, x);
        const double vkx = jac[k], vky = jac[NJ + k], vkz = jac[2 * NJ + k];
        const double wkx = jac[3 * NJ + k], wky = jac[4 * NJ + k], wkz = jac[5 * NJ + k];
        cx[k] = cross_sum(vky, x[2], vkz, x[1], wky, x[5], wkz, x[4]);
        cy[k] = cross_sum(vkz, x[0], vkx, x[2], wkz, x[3], wkx, x[5]);
        cz[k] = cross_sum(vkx, x[1], vky, x[0], wkx, x[4], wky, x[3]);
        jm[k] = dot3x(vkx, px, vky, py, vkz, pz);                                                           // v_k . P_k
        px = fmax_(x[1], wkz, fmax_(-x[2], wky, px)); py = fmax_(x[2], wkx, fmax_(-x[0], wkz, py)); pz = fmax_(x[0], wky, fmax_(-x[1], wkx, pz));      // + Ga_k x w_k
    }
    double sx = 0.0, sy = 0.0, sz = 0.0;
#pragma unroll
    for (int i = NJ - 1; i >= 0; --i) {
        sx += cx[i]; sy += cy[i]; sz += cz[i];
        jm[i] = m * (dot3x(jac[3 * NJ + i], sx, jac[4 * NJ + i], sy, jac[5 * NJ + i], sz) + jm[i]);
    }
}

template <int NJ>
RTB_HD void jacobm(const double (&jac)[6 * NJ], int axes, double (&jm)[NJ])
{
    double B[6][6], dval[6], dinv[6];
    jjt_masked<NJ>(jac, axes, B);
    ldl_factor<6>(B, dval, dinv);
    jacobm_factored<NJ>(jac, axes, B, dval, dinv, jm);
}

}  // namespace rtbhip
