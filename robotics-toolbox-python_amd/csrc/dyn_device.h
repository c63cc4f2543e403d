// dyn_device.h -- per-lane body of the dynamics-term kernels (dyn_kernels.hip): all the Newton-Euler
// passes the reference's Dynamics.inertia / coriolis / accel make for ONE configuration
// (robot/Dynamics.py:704-861, 424-509), on the rne_lane recursion of rne_device.h.
// __host__ __device__ so tests/emu can run the same code lane by lane on the CPU.
#pragma once
#include "rne_device.h"
#include "ldl.h"

namespace rtbhip {
#pragma clang fp contract(off)      // every operation written out, as in rne_device.h (why: there)


enum { kDynInertia = 0, kDynCoriolis = 1, kDynAccel = 2 };
// accel keeps M as a packed lower triangle -- except for modified-DH chains with prismatic joints, whose reference matrix can be unsymmetric
// (a prismatic first joint: dyn_lane below): those keep the full n x n tile
template <bool MDH, bool ALLREV> constexpr bool kDynFullTile = MDH && !ALLREV;

// mine : this lane's inputs  [q (n) | qd (n) | torque (n)]   (what the mode needs)
// mA   : n x n work/output tile (row-major): M for inertia / accel (accel leaves qdd in mA[0..n-1]), C for coriolis
// With sin/cos fixed across the passes, most of the forward recursion becomes loop-invariant in the
// compiler's eyes and LICM hoists it out of the pass loop -- into ~100 VGPRs the kernel does not have
// (first build: 428 B/lane of scratch, inertia 0.54 -> 0.90 ms).  Making the trig values opaque once per
// pass keeps every pass self-contained.
template <int NJ>
RTB_HD void dyn_opaque(double (&st)[NJ], double (&ct)[NJ])
{
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(st[j]), "+v"(ct[j]));
#endif
}

// a[i] for a wave-uniform run-time i without indexing the register array (that would send it to scratch)
template <int NJ>
RTB_HD double dyn_pick(const double (&a)[NJ], int i)
{
    double r = 0.0;
#pragma unroll
    for (int k = 0; k < NJ; ++k) r += (i == k) ? a[k] : 0.0;      // (a select chain gets turned back into an indexed load)
    return r;
}

#ifndef RTB_DYN_BILINEAR
#define RTB_DYN_BILINEAR 1      // 0: Dynamics.coriolis by the polar form over full passes (two per column) -- the first implementation, A/B
#endif
// ---- one column of C(q, qd), evaluated directly.  A Newton-Euler pass with gravity, friction and acceleration removed (what
// Dynamics.coriolis runs, robot/Dynamics.py:811-861) is a quadratic form tau(v) = B(v, v) of the joint velocities, and the matrix the reference
// assembles from its n + n (n - 1) / 2 unit-velocity passes is  C[:, k] = B(qd, e_k) = sum_j qd_j B(e_j, e_k).  B(u, w) comes out of ONE
// recursion that carries both velocity fields -- the angular velocities under u = qd and under w = e_k -- and in which every product of two
// velocities of ne.c:133-348, x(v) y(v), is replaced by x(u) y(w) + x(w) y(u) (that is 2 B; the caller halves): the terms are
//      wd :  (R^T w) x z qd                       a :  w x (w x p*)   [+ 2 w x z qd for a prismatic joint]
//      F  :  m ( .. + w x (w x r) )               N :  w x (I w)
// Against the polar form  (tau(qd + s e_k) - tau(qd - s e_k)) / 4s  this is one pass of ~1.4x the arithmetic instead of two, it is exact for
// any spread of velocities (no probe scale to choose, no cancellation: the rows whose velocities span many orders of magnitude need no other
// scheme), and the links before `first` -- at rest under w when first <= k, so their wd, a, F, N vanish -- only advance w(u).
// acc + wd x p* + wu x (ww x p*) + ww x (wu x p*), p*'s zero components dropped (rne_device.h: offset_accel; pm = 7: the general chain)
RTB_HD V3 offset_accel2(int pm, V3 wu, V3 ww, V3 wd, V3 ps, V3 acc)
{
    if (pm == 7) return cross_add(wd, ps, cross_add(wu, cross(ww, ps), cross_add(ww, cross(wu, ps), acc)));
    int cu, cw;
    const V3 xu = cross_bm(pm, wu, ps, cu), xw = cross_bm(pm, ww, ps, cw);
    return cross_add_bm(pm, wd, ps, cross_add_bm(cw, wu, xw, cross_add_bm(cu, ww, xu, acc)));
}
// SIG: the chain's structure signature (rne_device.h: RneSig) -- flags, alpha classes and p* masks at compile time
template <int NJ, bool MDH, bool ALLREV, RneSig SIG = 0, class LinksP, class InQ, class InQd, class Out>
RTB_HD void rne_bilinear_core(LinksP links, double (&st)[NJ], double (&ct)[NJ], InQ qin, InQd qdin, int k, Out tau, int first)
{
    static_assert(SIG == 0 || (ALLREV && NJ <= kRneSigMaxLinks), "a structure signature describes an all-revolute chain");
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(first), "+s"(k));       // (rne_core ACC: keeps the caller's column loop one loop)
#else
    asm volatile("" : "+r"(first), "+r"(k));
#endif
    const V3 o = v3(0, 0, 0);
    V3 F[NJ], Nn[NJ];
    int flg[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) flg[j] = SIG ? rsig_flags(SIG, j) : links[j].flags;
    V3 wu = o, ww = o, wd = o, a = o;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const auto &li = links[j];
        const LinkFwd l = link_fwd<ALLREV>(li);
        const bool pris = ALLREV ? false : (l.sigma != 0);
        const double qdu = qdin(j), qdw = j == k ? 1.0 : 0.0;
        const double d = pris ? qin(j) + l.offset : l.d;
        const int acls = SIG ? rsig_alpha(SIG, j) : 0, pm = SIG ? rsig_pmask(SIG, j) : 7;
        const bool ps0 = SIG != 0 && (flg[j] & kLinkPsZero) != 0;          // compile-time only: a run-time branch on it costs more than it saves here
        const Rot R = {st[j], ct[j], l.sa, l.ca, acls};
        const V3 ps = link_offset<MDH>(l, d);
        if (j < first) {               // wave-uniform: nothing moves under w yet
            if (MDH) wu = (j == 0) ? v3(0, 0, qdu) : (pris ? rot_inv<MDH>(R, wu) : addz(rot_inv<MDH>(R, wu), qdu));
            else wu = pris ? ((j == 0) ? o : rot_inv<MDH>(R, wu)) : rot_inv<MDH>(R, (j == 0) ? v3(0, 0, qdu) : addz(wu, qdu));
            F[j] = o; Nn[j] = o;
            sched_fence();
            continue;
        }
        V3 nu, nw, wdn, an;
        if (MDH) {
            if (j == 0) {
                nu = v3(0, 0, qdu); nw = v3(0, 0, qdw); wdn = o; an = o;
            } else {
                const V3 tu = rot_inv<MDH>(R, wu), tw = rot_inv<MDH>(R, ww);
                const V3 lin = rot_inv<MDH>(R, ps0 ? a : offset_accel2(pm, wu, ww, wd, ps, a));
                if (!pris) {
                    nu = addz(tu, qdu); nw = addz(tw, qdw);
                    wdn = rot_inv_add<MDH>(R, wd, crossz(tu, qdw) + crossz(tw, qdu));
                    an = lin;
                } else {
                    nu = tu; nw = tw;
                    wdn = rot_inv<MDH>(R, wd);
                    an = lin + 2.0 * (crossz(tu, qdw) + crossz(tw, qdu));
                }
            }
        } else {
            if (!pris) {
                nu = rot_inv<MDH>(R, (j == 0) ? v3(0, 0, qdu) : addz(wu, qdu));
                nw = rot_inv<MDH>(R, (j == 0) ? v3(0, 0, qdw) : addz(ww, qdw));
                wdn = (j == 0) ? o : rot_inv<MDH>(R, wd + (crossz(wu, qdw) + crossz(ww, qdu)));
                {
                    const V3 ra = (j == 0) ? o : rot_inv<MDH>(R, a);
                    an = ps0 ? ra : offset_accel2(pm, nu, nw, wdn, ps, ra);
                }
            } else {
                nu = (j == 0) ? o : rot_inv<MDH>(R, wu);
                nw = (j == 0) ? o : rot_inv<MDH>(R, ww);
                wdn = (j == 0) ? o : rot_inv<MDH>(R, wd);
                an = (j == 0) ? o : rot_inv<MDH>(R, a);
                an = an + cross(wdn, ps);
                an = an + 2.0 * (cross(nu, rot_inv<MDH>(R, v3(0, 0, qdw))) + cross(nw, rot_inv<MDH>(R, v3(0, 0, qdu))));
                an = cross_add(nu, cross(nw, ps), cross_add(nw, cross(nu, ps), an));
            }
        }
        wu = nu; ww = nw; wd = wdn; a = an;
        V3 ac = a;
        if (!(flg[j] & kLinkRZero)) {
            const V3 rc = v3(li.rx, li.ry, li.rz);
            ac = cross_add(wd, rc, cross_add(wu, cross(ww, rc), cross_add(ww, cross(wu, rc), a)));
        }
        F[j] = l.m * ac;
        if (flg[j] & kLinkIDiag) {
            const V3 iu = v3(li.I[0] * wu.x, li.I[4] * wu.y, li.I[8] * wu.z), iw = v3(li.I[0] * ww.x, li.I[4] * ww.y, li.I[8] * ww.z);
            Nn[j] = cross_add(wu, iw, cross_add(ww, iu, v3(li.I[0] * wd.x, li.I[4] * wd.y, li.I[8] * wd.z)));
        } else {
            Nn[j] = cross_add(wu, inertia_times(li, ww), cross_add(ww, inertia_times(li, wu), inertia_times(li, wd)));
        }
        sched_fence();
    }
    // ---- backward recursion + joint projection: rne_core's (ne.c:354-492) without friction, motor inertia (qdd = 0) and tip wrench
    V3 f = o, nn = o;
    Rot Rn = {0, 1, 0, 1, 0};
    V3 psn = o;
    int pmn = 7;
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int j = NJ - 1 - jj;
        const auto &li = links[j];
        const LinkBwd l = link_bwd<ALLREV, false>(li);
        const bool last = (jj == 0);
        const bool pris = ALLREV ? false : (l.sigma != 0);
        const bool rzero = (flg[j] & kLinkRZero) != 0;
        const V3 rc = rzero ? o : v3(li.rx, li.ry, li.rz);
        const double d = pris ? qin(j) + l.offset : l.d;
        const int acls = SIG ? rsig_alpha(SIG, j) : 0, pm = SIG ? rsig_pmask(SIG, j) : 7;
        const Rot R = {st[j], ct[j], l.sa, l.ca, acls};
        const V3 ps = link_offset<MDH>(l, d);
        V3 fj, nj;
        if (MDH) {
            const V3 fn = last ? f : rot_fwd<MDH>(Rn, f);
            fj = fn + F[j];
            const V3 base = rzero ? Nn[j] : cross_add(rc, F[j], Nn[j]);
            if (last) nj = nn + base;
            else nj = cross_add_am(pmn, psn, fn, rot_fwd_add<MDH>(Rn, nn, base));
        } else {
            fj = last ? F[j] + f : rot_fwd_add<MDH>(Rn, f, F[j]);
            const V3 base = cross_add(ps + rc, F[j], Nn[j]);
            if (!last) nj = rot_fwd_add<MDH>(Rn, cross_add(rot_inv<MDH>(Rn, ps), f, nn), base);
            else nj = cross_add(ps, f, nn + base);
        }
        const V3 prj = pris ? fj : nj;
        tau(j, MDH ? prj.z : fmad(l.sa, prj.y, l.ca * prj.z));
        f = fj; nn = nj; Rn = R; psn = ps; pmn = pm;
        sched_fence();
    }
}

template <int NJ, bool MDH, int MODE, bool ALLREV, RneSig SIG = 0, class LinksP>
RTB_HD void dyn_lane(LinksP links, const double *mine, double *mA, V3 grav, const double *qrow = nullptr)
{
    const V3 zero = v3(0, 0, 0);
    // qrow: where this lane's q lives when it is not mine[0..n) -- all-revolute chains read q only for the trig below,
    // so the kernel may keep it in the output tile (overwritten by the first pass) instead of a row of its own
    const double *qsrc = qrow ? qrow : mine;
    auto qin = [&](int j) { return qsrc[j]; };
    double st[NJ], ct[NJ];
    rne_trig<NJ, ALLREV>(links, qin, st, ct);      // every pass below is at the same q: sin/cos once
    double b[NJ];
    if (MODE == kDynAccel) {
        // tau_0 = rne(q, qd, 0) with gravity and friction (Dynamics.py:500); b = torque - tau_0 stays in registers across the
        // passes for M.  This pass comes FIRST so that the kernel may keep q, qd and torque in the M tile itself until here (the
        // passes below overwrite them): 22.5 -> 14.8 KB of LDS per wave for n = 7, 7 -> 10 waves per CU
        dyn_opaque<NJ>(st, ct);
        rne_core<NJ, MDH, true, ALLREV, true, false, false, SIG>(links, NJ, st, ct, grav, zero, zero, qin, [&](int j) { return mine[NJ + j]; },
                                [&](int) { return 0.0; }, [&](int j, double v) { b[j] = mine[2 * NJ + j] - v; });
#if defined(__HIP_DEVICE_COMPILE__)
        // the pass must be over -- its link state dead -- before the passes below start (scheduled together they spill)
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(b[j]));
        sched_fence();
#endif
    }
    if (MODE == kDynInertia || MODE == kDynAccel) {
        // row i of the result = tau for qdd = e_i, qd = 0, no gravity (Dynamics.py:752-758, :492-496)
#pragma unroll 1
        for (int i = 0; i < NJ; ++i) {
            dyn_opaque<NJ>(st, ct);
            if constexpr (ALLREV) {
                // column i of M from the acceleration-only pass (rne_device.h): torques of joints j >= i = entries (j, i) of the
                // packed lower triangle -- what accel's LDL^T solve reads, and what the inertia kernel's flush mirrors into (n, n)
                rne_core<NJ, MDH, false, true, true, true, false, SIG>(links, NJ, st, ct, zero, zero, zero, qin, [&](int) { return 0.0; },
                                        [&](int j) { return j == i ? 1.0 : 0.0; },
                                        [&](int j, double v) { mA[j * (j + 1) / 2 + i] = v; }, i);
            } else {
                // inertia: the full row as computed (the reference returns the unsymmetrised matrix); accel: only the lower
                // triangle the LDL^T solve reads, packed -- 28 instead of 49 doubles of LDS per lane for n = 7
                rne_core<NJ, MDH, true, ALLREV, true, false, false>(links, NJ, st, ct, zero, zero, zero, qin, [&](int) { return 0.0; },
                                        [&](int j) { return j == i ? 1.0 : 0.0; },
                                        [&](int j, double v) {
                                            if (MODE == kDynAccel && !kDynFullTile<MDH, ALLREV>) { if (j <= i) mA[i * (i + 1) / 2 + j] = v; }
                                            else mA[i * NJ + j] = v;
                                        });
            }
        }
    }
    if (MODE == kDynAccel) {
        // M qdd = torque - tau_0
        double x[NJ], M[NJ][NJ];
        bool general = false;
        if constexpr (kDynFullTile<MDH, ALLREV>) general = links[0].sigma != 0;        // wave-uniform
        if (general) {
            // the one chain shape whose reference matrix is not symmetric: modified DH with a prismatic FIRST joint (core/ne.c:187-196 gives link 1
            // the joint rate and acceleration as ANGULAR velocity and acceleration).  The reference solves with the matrix as its passes return
            // it (numpy.linalg.solve, robot/Dynamics.py:505): so does this, in memory, with the right-hand side in the torque row it came from
            double *rhs = const_cast<double *>(mine) + 2 * NJ;
#pragma unroll
            for (int j = 0; j < NJ; ++j) rhs[j] = b[j];
            lu_solve_mem<NJ>(mA, rhs);
#pragma unroll
            for (int j = 0; j < NJ; ++j) x[j] = rhs[j];
        } else {
#pragma unroll
            for (int r = 0; r < NJ; ++r)
#pragma unroll
                for (int c = 0; c <= r; ++c) M[r][c] = kDynFullTile<MDH, ALLREV> ? mA[r * NJ + c] : mA[r * (r + 1) / 2 + c];
            ldl_solve<NJ>(M, b, x);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) mA[j] = x[j];        // the tile's first n slots become the output row
    }
    if (MODE == kDynCoriolis) {
        // What ships is rne_bilinear_core above (RTB_DYN_BILINEAR = 1): one two-field pass per column.  The rest of this comment describes the
        // first implementation, kept under RTB_DYN_BILINEAR = 0 as the A/B baseline (scripts/build_variant.sh dyn_kernels dyn_polar ...):
        // Dynamics.coriolis (robot/Dynamics.py:811-861) builds C from n passes at QD = e_i (Csq) and n (n - 1) / 2 passes at
        // QD = e_i + e_j (gravity, friction and acceleration removed).  Such a pass is a homogeneous quadratic form of qd,
        //      tau_r(v) = sum_ab h_rab v_a v_b   (h symmetric in a, b),
        // the reference's combination (T_jk - Csq_j - Csq_k) / 2 is h_rjk, and its result is  C[r, k] = sum_j h_rjk qd_j  -- the
        // polar form of tau against e_k.  The same numbers come from TWO passes per column instead of the reference's 28 in all:
        //      C[:, k] = ( tau(qd + s e_k) - tau(qd - s e_k) ) / (4 s)          (exact for a quadratic form)
        // with s the power of two next above max|qd_j| (1 for qd = 0), so that the probe is neither lost in qd nor qd in the probe,
        // whatever the overall scale of the velocities, and the scaling is exact: 14 passes for a 7-joint arm; agreement with
        // the reference's order of operations ~1e-15 of max|C| (tests: oracle.coriolis_dh statement for statement, emu and
        // GPU, velocities from 1e-9 to 1e9).  Its rounding error scales with (max|qd|)^2 max|h| / s, the reference's with the
        // largest single term |h_rjk qd_j|: for a row whose nonzero velocities span more than 2^16 the two can differ by that
        // ratio, so such rows take the reference's own scheme below instead -- each row on its own (a wave that holds such a row, ~1 % of
        // the waves with normally distributed velocities, runs both bodies under the execution mask), so that a row's result never depends
        // on which other rows share its tile.
        // qd in registers: the kernel may keep the input row in the C tile itself, which the first pass starts to overwrite
        double qdv[NJ], vmax = 0.0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) { qdv[j] = mine[NJ + j]; vmax = fmax(vmax, fabs(qdv[j])); }
#if RTB_DYN_BILINEAR
        // (what ships: the two-field pass above, one per column; the polar form below is kept as the A/B baseline)
        (void)vmax;
#pragma unroll 1
        for (int k = 0; k < NJ; ++k) {
            dyn_opaque<NJ>(st, ct);
            rne_bilinear_core<NJ, MDH, ALLREV, SIG>(links, st, ct, qin, [&](int j) { return qdv[j]; }, k,
                                               [&](int r, double v) { mA[r * NJ + k] = 0.5 * v; }, k);
        }
#else
        // s = 2^ceil(log2(vmax)) through the exponent field (exact); 1 for qd = 0 and for non-finite rows (which come out NaN as
        // they should); the exponent is kept where s^2 neither overflows nor underflows
        int ex = 0;
        const double mant = frexp(vmax, &ex);                      // vmax = mant 2^ex, mant in [0.5, 1)
        if (mant == 0.5) ex -= 1;
        if (!(vmax > 0.0) || !(vmax < 1.7e308)) ex = 0;
        ex = ex > 400 ? 400 : (ex < -400 ? -400 : ex);
        const double sc = ldexp(1.0, ex);
        // (a row at rest is exactly zero, as in the reference -- not the rounding difference of the two probes)
        const double inv4s = vmax > 0.0 || vmax != vmax ? 0.25 / sc : 0.0;
        bool wide = false;
#pragma unroll
        for (int j = 0; j < NJ; ++j) wide = wide || (qdv[j] != 0.0 && fabs(qdv[j]) * 65536.0 < vmax);
        auto polar = [&]() {
#pragma unroll 1
        for (int k = 0; k < NJ; ++k) {
            dyn_opaque<NJ>(st, ct);
            rne_core<NJ, MDH, false, ALLREV, true, false, false>(links, NJ, st, ct, zero, zero, zero, qin, [&](int j) { return j == k ? qdv[j] + sc : qdv[j]; },
                                     [&](int) { return 0.0; }, [&](int r, double v) { mA[r * NJ + k] = v; });
            dyn_opaque<NJ>(st, ct);
            rne_core<NJ, MDH, false, ALLREV, true, false, false>(links, NJ, st, ct, zero, zero, zero, qin, [&](int j) { return j == k ? qdv[j] - sc : qdv[j]; },
                                     [&](int) { return 0.0; }, [&](int r, double v) { mA[r * NJ + k] = (mA[r * NJ + k] - v) * inv4s; });
        }
        };
        auto reference_scheme = [&]() {
        // Dynamics.py:828-856 regrouped so that ONE n x n tile per lane suffices (the reference keeps Csq and C):
        //   C[:,k] = sum_{j != k} (T_jk - Csq_k - Csq_j) qd_j / 2 + Csq_k qd_k
        //          = 1/2 sum_{j != k} T_jk qd_j + Csq_k (2 qd_k - S / 2) - U / 2,   S = sum_j qd_j,  U = sum_j Csq_j qd_j
        // with T_jk the pass at QD = e_j + e_k and Csq_j the pass at QD = e_j (friction removed, :820).  Same terms,
        // different association: agreement with the reference order is ~1e-15 relative.
        double S = 0.0, U[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { S += qdv[j]; U[j] = 0.0; }
#pragma unroll 1
        for (int i = 0; i < NJ; ++i) {
            const double qdi = dyn_pick<NJ>(qdv, i), wi = 2.0 * qdi - 0.5 * S;
            dyn_opaque<NJ>(st, ct);
            rne_core<NJ, MDH, false, ALLREV, true, false, false>(links, NJ, st, ct, zero, zero, zero, qin, [&](int j) { return j == i ? 1.0 : 0.0; },
                                     [&](int) { return 0.0; }, [&](int r, double v) { mA[r * NJ + i] = v * wi; U[r] += v * qdi; });
        }
#pragma unroll
        for (int r = 0; r < NJ; ++r)
#pragma unroll
            for (int c = 0; c < NJ; ++c) mA[r * NJ + c] -= 0.5 * U[r];
#pragma unroll 1
        for (int i = 0; i < NJ; ++i) {
#pragma unroll 1
            for (int j = i + 1; j < NJ; ++j) {
                const double hi = 0.5 * dyn_pick<NJ>(qdv, i), hj = 0.5 * dyn_pick<NJ>(qdv, j);
                dyn_opaque<NJ>(st, ct);
                rne_core<NJ, MDH, false, ALLREV, true, false, false>(links, NJ, st, ct, zero, zero, zero, qin,
                                         [&](int k) { return (k == i || k == j) ? 1.0 : 0.0; }, [&](int) { return 0.0; },
                                         [&](int r, double tau) {
                                             mA[r * NJ + j] += tau * hi;
                                             mA[r * NJ + i] += tau * hj;
                                         });
            }
        }
        };
        // the common case (no such row in the wave) is one straight wave-uniform body; only a wave that holds a wide row runs the per-lane choice.
        // Price of the per-row choice against round 2's per-wave one, same box (visit z): 0.56 -> 0.60 ms per 1e6 (the form that runs the polar
        // body for every lane and lets wide rows redo their tile measured the same)
        if (!wave_any(wide)) polar();
        else if (!wide) polar();
        else reference_scheme();
#endif
    }
}

#pragma clang fp contract(fast)
}  // namespace rtbhip
