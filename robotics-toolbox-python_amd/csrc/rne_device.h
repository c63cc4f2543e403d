// rne_device.h -- per-lane recursive Newton-Euler for DH / modified-DH serial chains.
//
// Replaces newton_euler (core/ne.c:62-493) + rot_mat (core/frne.c:310-351) + the 3-vector
// helpers of core/vmath.c.  The reference keeps all intermediate state in the shared Robot->links[]
// struct (core/frne.h:70-77, not re-entrant) and runs the forward pass, the backward pass and the
// joint projection as three loops over it.  Here one lane owns one (q, qd, qdd) sample and the two
// recursions are fused around per-lane registers: the forward pass leaves, per link, only what the
// backward pass needs -- F_j = m a_c, N_j = I wd + w x (I w), sin/cos(theta_j) and d_j -- and the
// link rotation is rebuilt from (st, ct) x (sa, ca) instead of being stored (9 doubles saved).
// NJ is a compile-time joint count so every array below is statically indexed (registers, no
// scratch); NJ == 0 selects the run-time-n fallback, which the compiler places in private memory.
#pragma once
#include "rtbhip_internal.h"
#include <cmath>
#include "trig.h"

#ifndef RTB_HD
#define RTB_HD __host__ __device__ __forceinline__
#endif

namespace rtbhip {

struct V3 { double x, y, z; };
RTB_HD V3 v3(double x, double y, double z) { V3 r = {x, y, z}; return r; }
RTB_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RTB_HD V3 operator*(double s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
RTB_HD V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RTB_HD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

struct R3 { double m[9]; };  // row-major
RTB_HD V3 rmul(const R3 &r, V3 v)   // R v
{
    return v3(r.m[0] * v.x + r.m[1] * v.y + r.m[2] * v.z, r.m[3] * v.x + r.m[4] * v.y + r.m[5] * v.z,
              r.m[6] * v.x + r.m[7] * v.y + r.m[8] * v.z);
}
RTB_HD V3 rtmul(const R3 &r, V3 v)  // R^T v
{
    return v3(r.m[0] * v.x + r.m[3] * v.y + r.m[6] * v.z, r.m[1] * v.x + r.m[4] * v.y + r.m[7] * v.z,
              r.m[2] * v.x + r.m[5] * v.y + r.m[8] * v.z);
}

// link rotation and offset vector from the joint state: frne.c:329-347
template <bool MDH, class LinkT>
RTB_HD void link_frame(const LinkT &l, double st, double ct, double d, R3 &R, V3 &ps)
{
    const double sa = l.sa, ca = l.ca;
    if (!MDH) {
        R.m[0] = ct; R.m[1] = -ca * st; R.m[2] = sa * st;
        R.m[3] = st; R.m[4] = ca * ct;  R.m[5] = -sa * ct;
        R.m[6] = 0;  R.m[7] = sa;       R.m[8] = ca;
        ps = v3(l.a, d * sa, d * ca);
    } else {
        R.m[0] = ct;      R.m[1] = -st;     R.m[2] = 0;
        R.m[3] = st * ca; R.m[4] = ca * ct; R.m[5] = -sa;
        R.m[6] = st * sa; R.m[7] = ct * sa; R.m[8] = ca;
        ps = v3(l.a, -d * sa, d * ca);
    }
}

template <class LinkT>
RTB_HD V3 inertia_times(const LinkT &l, V3 v)  // vmath.c mat_vect_mult: m[r + 3c]
{
    return v3(l.I[0] * v.x + l.I[3] * v.y + l.I[6] * v.z, l.I[1] * v.x + l.I[4] * v.y + l.I[7] * v.z,
              l.I[2] * v.x + l.I[5] * v.y + l.I[8] * v.z);
}

// One sample.  links: wave-uniform link table (scalar loads on the GPU).
// qv/qdv/qddv/tau: per-lane accessors  in(j) -> double, out(j, v).
template <int NJ, bool MDH, class LinksP, class InQ, class InQd, class InQdd, class Out>
RTB_HD void rne_lane(LinksP links, int n_rt, V3 grav, V3 ftip, V3 ntip, InQ qin, InQd qdin, InQdd qddin, Out tau)
{
    constexpr int CAP = NJ > 0 ? NJ : RTBHIP_MAX_JOINTS;
    const int n = NJ > 0 ? NJ : n_rt;
    double st[CAP], ct[CAP], dj[CAP];
    V3 F[CAP], Nn[CAP];

    // ---- forward recursion (ne.c:133-348)
    V3 w = v3(0, 0, 0), wd = v3(0, 0, 0), a = v3(0, 0, 0);
    double qddx = 0.0, qddy = 0.0;  // ne.c:311 lets gravity leak into qddv.x/.y for later links
#pragma unroll
    for (int j = 0; j < n; ++j) {
        const auto &l = links[j];
        const bool pris = l.sigma != 0;
        const double qj = qin(j), qdj = qdin(j), qddj = qddin(j);
        const double th = pris ? l.theta : qj + l.offset;   // frne.c:196-202
        const double d = pris ? qj + l.offset : l.d;
        double s, c;
        rtb_sincos(th, &s, &c);
        st[j] = s; ct[j] = c; dj[j] = d;
        R3 R; V3 ps;
        link_frame<MDH>(l, s, c, d, R, ps);
        const V3 qdv = v3(0, 0, qdj);
        V3 qddv = v3(qddx, qddy, qddj);
        V3 wn, wdn, an;
        if (MDH) {
            if (!pris) {
                if (j == 0) {
                    wn = qdv; wdn = qddv; an = rtmul(R, grav);
                } else {
                    const V3 t1 = rtmul(R, w);
                    wn = t1 + qdv;
                    wdn = (cross(t1, qdv) + rtmul(R, wd)) + qddv;
                    an = rtmul(R, (cross(wd, ps) + cross(w, cross(w, ps))) + a);
                }
            } else {
                if (j == 0) {
                    wn = qdv; wdn = qddv; an = grav;
                } else {
                    wn = rtmul(R, w);
                    wdn = rtmul(R, wd);
                    an = rtmul(R, (cross(wd, ps) + cross(w, cross(w, ps))) + a);
                    an = (an + 2.0 * cross(wn, qdv)) + qddv;
                }
            }
        } else {
            if (!pris) {
                const V3 t1 = (j == 0) ? qdv : w + qdv;
                wn = rtmul(R, t1);
                const V3 t3 = (j == 0) ? qddv : (wd + qddv) + cross(w, qdv);
                wdn = rtmul(R, t3);
                an = (cross(wdn, ps) + cross(wn, cross(wn, ps))) + rtmul(R, (j == 0) ? grav : a);
            } else {
                wn = (j == 0) ? v3(0, 0, 0) : rtmul(R, w);
                wdn = (j == 0) ? v3(0, 0, 0) : rtmul(R, wd);
                if (j == 0) {
                    qddv = qddv + grav;
                    qddx = qddv.x; qddy = qddv.y;
                    an = rtmul(R, qddv);
                } else {
                    an = rtmul(R, qddv + a);
                }
                an = an + cross(wdn, ps);
                an = an + 2.0 * cross(wn, rtmul(R, qdv));
                an = an + cross(wn, cross(wn, ps));
            }
        }
        w = wn; wd = wdn; a = an;
        const V3 rc = v3(l.rx, l.ry, l.rz);
        const V3 ac = (cross(wd, rc) + cross(w, cross(w, rc))) + a;   // ne.c:228-232
        F[j] = l.m * ac;
        Nn[j] = inertia_times(l, wd) + cross(w, inertia_times(l, w));
    }

    // ---- backward recursion + joint projection (ne.c:354-492), fused
    V3 f = ftip, nn = ntip;   // f_{j+1}, n_{j+1} expressed as the reference's "tip" values for the last link
    R3 Rn; V3 psn = v3(0, 0, 0);  // frame of link j+1
#pragma unroll
    for (int jj = 0; jj < n; ++jj) {
        const int j = n - 1 - jj;
        const auto &l = links[j];
        const bool last = (jj == 0);
        const V3 rc = v3(l.rx, l.ry, l.rz);
        R3 R; V3 ps;
        link_frame<MDH>(l, st[j], ct[j], dj[j], R, ps);
        V3 fj, nj;
        if (MDH) {
            const V3 fn = last ? f : rmul(Rn, f);
            fj = fn + F[j];
            const V3 t1 = last ? nn : rmul(Rn, nn) + cross(psn, fn);
            nj = (t1 + cross(rc, F[j])) + Nn[j];
        } else {
            fj = F[j] + (last ? f : rmul(Rn, f));
            V3 t1 = cross(ps + rc, F[j]);
            if (!last) t1 = t1 + rmul(Rn, cross(rtmul(Rn, ps), f) + nn);
            else t1 = (t1 + cross(ps, f)) + nn;
            nj = t1 + Nn[j];
        }
        const V3 ax = MDH ? v3(0, 0, 1) : rtmul(R, v3(0, 0, 1));
        const double qdj = qdin(j), qddj = qddin(j);
        double t = (l.sigma != 0) ? dot(fj, ax) : dot(nj, ax);
        t += l.G * l.G * l.Jm * qddj;
        t += l.G * l.G * l.B * qdj;
        t += fabs(l.G) * ((qdj > 0 ? l.Tc0 : 0.0) + (qdj < 0 ? l.Tc1 : 0.0));
        tau(j, t);
        f = fj; nn = nj; Rn = R; psn = ps;
    }
}

}  // namespace rtbhip
