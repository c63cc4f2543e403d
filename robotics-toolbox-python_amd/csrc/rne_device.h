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
#ifndef __HIPCC_RTC__
#include <cmath>
#endif
#include "trig.h"
#include "exactform.h"

#ifndef RTB_HD
#define RTB_HD __host__ __device__ __forceinline__
#endif

namespace rtbhip {

// ---- EVERY floating-point operation of the dynamics recursions (this header, dyn_device.h, tree_device.h) is written out: `fp contract(off)` for
// the whole header, fused multiply-adds only where the source says fma.  Why: a kernel instantiated for a robot's structure (RneSig, a tree's
// knowledge type -- built in or compiled at run time, jit.cpp) must return the general kernel's bits, and with the compiler free to contract it
// does not: knowing a joint kind or a flag at compile time turns wave-uniform branches into straight-line code, the DAG combiner then sees a
// product and an addition in ONE block that the general kernel keeps in two, fuses them, and the torques differ in the last bits (round 6, first
// device run: 80 % of the entries of a 13-joint chain, 1e-13).  Written out, the operation sequence is the same whatever the optimiser knows.
#pragma clang fp contract(off)
struct V3 { double x, y, z; };
RTB_HD V3 v3(double x, double y, double z) { V3 r = {x, y, z}; return r; }
RTB_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RTB_HD V3 operator*(double s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
RTB_HD V3 cross(V3 a, V3 b)      // each component  fma(u1, v1, -round(u2 v2))
{
    return v3(__builtin_fma(a.y, b.z, -(a.z * b.y)), __builtin_fma(a.z, b.x, -(a.x * b.z)), __builtin_fma(a.x, b.y, -(a.y * b.x)));
}
RTB_HD double dot(V3 a, V3 b) { return __builtin_fma(a.z, b.z, __builtin_fma(a.y, b.y, a.x * b.x)); }

// Link frame (frne.c:329-347).  The rotation is never formed as a matrix: standard DH is
// R = Rz(theta) Rx(alpha), modified DH is R = Rx(alpha) Rz(theta), so R v and R^T v are two planar
// rotations applied in turn straight from (sin theta, cos theta) [per lane] and (sin alpha, cos
// alpha) [wave-uniform] -- 12 flops instead of 15 and no 9-register matrix to keep live.
// acls: the STRUCTURE CLASS of alpha when the kernel knows it at compile time (RneSig below; 0 = nothing known): 1 alpha = 0 (sa = 0, ca = 1
// exactly: the x rotation is the identity), 2 / 3 sa = +1 / -1 exactly (alpha = +-pi/2; ca keeps its libm value, 6.1e-17).  The class forms are
// the general ones with the products by an exact 0 dropped and the products by an exact +-1 taken as the operand.
struct Rot { double s, c, sa, ca; int acls; };
// The x rotation by alpha acts on a pair (y, z):  fwd (ca y - sa z, sa y + ca z);  inv (ca y + sa z, ca z - sa y).  ONE form for every class of
// alpha: each output is dotk_rt (exactform.h) -- fma(ca, ., round(sa .)) with sin / cos alpha's exact 0 / +-1 rewritten away -- so a kernel
// instantiated for a table's alpha classes returns the general kernel's bits by construction.  (acls is a constant of the unrolled link step.)
RTB_HD int rot_ksa(int acls) { return acls == 1 ? kC0 : (acls == 2 ? kCP : (acls == 3 ? kCN : kCA)); }
RTB_HD int rot_kca(int acls) { return acls == 1 ? kCP : kCA; }
RTB_HD double alpha_mix(const Rot &r, double with_sa, double with_ca)      // sa * with_sa + ca * with_ca
{
    return dotk_rt(rot_ksa(r.acls), rot_kca(r.acls), kC0, r.sa, with_sa, r.ca, with_ca, 0.0, 0.0);
}
RTB_HD void alpha_fwd(const Rot &r, double y, double z, double &oy, double &oz) { oy = alpha_mix(r, -z, y); oz = alpha_mix(r, y, z); }
RTB_HD void alpha_inv(const Rot &r, double y, double z, double &oy, double &oz) { oy = alpha_mix(r, z, y); oz = alpha_mix(r, -y, z); }
// the z rotation by theta (per lane) on a pair (x, y):  fwd (c x - s y, s x + c y);  inv (c x + s y, c y - s x) -- explicit operations, no contraction
RTB_HD void theta_fwd(const Rot &r, double x, double y, double &ox, double &oy)
{
#pragma clang fp contract(off)
    ox = __builtin_fma(r.c, x, -(r.s * y)); oy = __builtin_fma(r.s, x, r.c * y);
}
RTB_HD void theta_inv(const Rot &r, double x, double y, double &ox, double &oy)
{
#pragma clang fp contract(off)
    ox = __builtin_fma(r.c, x, r.s * y); oy = __builtin_fma(r.c, y, -(r.s * x));
}
template <bool MDH>
RTB_HD V3 rot_fwd(const Rot &r, V3 v)   // R v:  standard DH  Rz(theta) Rx(alpha) v;  modified DH  Rx(alpha) Rz(theta) v
{
    double a, b, oy, oz;
    if (!MDH) { alpha_fwd(r, v.y, v.z, oy, oz); theta_fwd(r, v.x, oy, a, b); return v3(a, b, oz); }
    theta_fwd(r, v.x, v.y, a, b);
    alpha_fwd(r, b, v.z, oy, oz);
    return v3(a, oy, oz);
}
template <bool MDH>
RTB_HD V3 rot_inv(const Rot &r, V3 v)   // R^T v
{
    double a, b, oy, oz;
    if (!MDH) { theta_inv(r, v.x, v.y, a, b); alpha_inv(r, b, v.z, oy, oz); return v3(a, oy, oz); }
    alpha_inv(r, v.y, v.z, oy, oz);
    theta_inv(r, v.x, oy, a, b);
    return v3(a, b, oz);
}
template <bool MDH, class LinkT>
RTB_HD V3 link_offset(const LinkT &l, double d)   // p* (frne.c:337,347)
{
    return MDH ? v3(l.a, -d * l.sa, d * l.ca) : v3(l.a, d * l.sa, d * l.ca);
}

template <class LinkT>
RTB_HD V3 inertia_times(const LinkT &l, V3 v)  // vmath.c mat_vect_mult: m[r + 3c]
{
    return v3(__builtin_fma(l.I[6], v.z, __builtin_fma(l.I[3], v.y, l.I[0] * v.x)), __builtin_fma(l.I[7], v.z, __builtin_fma(l.I[4], v.y, l.I[1] * v.x)),
              __builtin_fma(l.I[8], v.z, __builtin_fma(l.I[5], v.y, l.I[2] * v.x)));
}

// One sample.  links: wave-uniform link table (scalar loads on the GPU).
// qin/qdin/qddin/tau: per-lane accessors  in(j) -> double, out(j, v).  q(j) must stay readable until
// tau(j) has been written (the kernel lets tau overwrite the q slots).
// a + (0,0,z)  and  a x (0,0,s): the joint rate / acceleration vectors of a revolute joint have only a z component;
// the compiler may not drop the zero terms of the general formulas itself (IEEE: 0*x is not 0 for inf / NaN).
// Finite inputs give bit-identical results (up to the sign of a zero).
RTB_HD V3 addz(V3 a, double z) { return v3(a.x, a.y, a.z + z); }
RTB_HD V3 crossz(V3 a, double s) { return v3(a.y * s, -(a.x * s), 0.0); }

// FRICTION = false drops the viscous and Coulomb terms (Dynamics.nofriction(True, True), used by coriolis()).
// ALLREV = true promises that every link is revolute (sigma == 0): the prismatic branches, the per-lane
// selects between joint kinds and the gravity leak of ne.c:311 disappear at compile time.
// sin/cos of the joint angles of one sample.  Compile-time NJ: all NJ evaluations in one basic block
// (branch-free reduction, trig.h) so the scheduler interleaves the independent chains; one wave-wide
// fallback.  Split from the recursions so that the multi-pass kernels (inertia, coriolis, accel: 7..28
// Newton-Euler passes over the SAME q) evaluate it once.
template <int NJ, bool ALLREV, class LinksP, class InQ>
RTB_HD void rne_trig(LinksP links, InQ qin, double (&st)[NJ], double (&ct)[NJ])
{
    bool big = false;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const auto &l = links[j];
        const double th = (!ALLREV && l.sigma != 0) ? l.theta : qin(j) + l.offset;   // frne.c:196-202
        st[j] = th;
        big = big || !(fabs(th) < kTrigFastLimit);
    }
    if (wave_any(big)) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) { double s, c; sincos(st[j], &s, &c); st[j] = s; ct[j] = c; }
    } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) { double s, c; sincos_reduced(st[j], s, c); st[j] = s; ct[j] = c; }
    }
    sched_fence();
}

// The fields one link's forward / backward step reads, copied out of the (scalar-memory) link table in ONE batch at the
// top of the step.  Left to itself the compiler loads each field in the basic block that uses it and waits for it there:
// with the zero-skipping branches that is 6-7 load -> s_waitcnt -> use sequences per link (SQ counters: 77 scalar loads
// per wave, a third of the wave's lifetime in s_waitcnt).  A batch is one wait.
struct LinkFwd { double sa, ca, a, d, theta, offset, m; int sigma; };
struct LinkBwd { double sa, ca, a, d, offset, gjm, gb, ag, Tc0, Tc1; int sigma; };
template <bool ALLREV, class LinkT>
RTB_HD LinkFwd link_fwd(const LinkT &l)
{
    LinkFwd o;
    o.sa = l.sa; o.ca = l.ca; o.a = l.a; o.d = l.d; o.m = l.m;
    o.theta = ALLREV ? 0.0 : l.theta; o.offset = ALLREV ? 0.0 : l.offset; o.sigma = ALLREV ? 0 : l.sigma;
    return o;
}
template <bool ALLREV, bool FRICTION, class LinkT>
RTB_HD LinkBwd link_bwd(const LinkT &l)
{
    LinkBwd o;
    o.sa = l.sa; o.ca = l.ca; o.a = l.a; o.d = l.d; o.offset = ALLREV ? 0.0 : l.offset; o.sigma = ALLREV ? 0 : l.sigma;
    o.gjm = l.gjm;
    o.gb = FRICTION ? l.gb : 0.0; o.ag = FRICTION ? l.ag : 0.0; o.Tc0 = FRICTION ? l.Tc0 : 0.0; o.Tc1 = FRICTION ? l.Tc1 : 0.0;
    return o;
}

#ifndef RTB_RNE_NOFENCE
#define RTB_RNE_NOFENCE 0
#endif
#ifndef RTB_RNE_PREFETCH
#define RTB_RNE_PREFETCH 0
#endif
// ---- fused forms of the vector algebra of the recursions.  A sum of cross products and rotated vectors is accumulated as a
// chain of fused multiply-adds (two per component and product) instead of "cross, cross, add, add": the same terms, one rounding
// per product less, ~10 % fewer fp64 instructions per link.
RTB_HD double fmad(double a, double b, double c) { return __builtin_fma(a, b, c); }
RTB_HD V3 cross_add(V3 a, V3 b, V3 acc)          // acc + a x b
{
    return v3(fmad(a.y, b.z, fmad(-a.z, b.y, acc.x)), fmad(a.z, b.x, fmad(-a.x, b.z, acc.y)), fmad(a.x, b.y, fmad(-a.y, b.x, acc.z)));
}
// ---- STRUCTURE SIGNATURE of a DH / modified-DH chain of revolute links (up to 8): what the kernel may know about the link table at compile
// time.  7 bits per link: the shortcut flags the host sets from EXACT zeros (kLinkRZero | kLinkIDiag | kLinkPsZero), the class of alpha (Rot),
// a == 0, d == 0; bit 63 present, bit 62 no link has friction (B = 0 and Tc = 0: the viscous and Coulomb terms are exact zeros), bit 61 no motor inertia.  rne_core
// instantiated for a signature reads no flag word and takes no wave-uniform branch on one, rotates about x in the form of alpha's class and
// drops the cross-product terms of the zero components of p* = (a, -+d sa, d ca).  A kernel instantiated for a signature serves exactly the
// robots whose table has it (rne_kernels.hip compares); every other robot takes the general kernels.
typedef unsigned long long RneSig;
constexpr RneSig kRneSigPresent = 1ull << 63, kRneSigNoFriction = 1ull << 62, kRneSigNoMotor = 1ull << 61;      // NoMotor: every G^2 Jm is an exact zero
constexpr int kRneSigMaxLinks = 8;
RTB_HD constexpr int rsig_flags(RneSig s, int j) { return (int)((s >> (7 * j)) & 7u); }
RTB_HD constexpr int rsig_alpha(RneSig s, int j) { return (int)((s >> (7 * j + 3)) & 3u); }
RTB_HD constexpr int rsig_pmask(RneSig s, int j)      // which components of p* may be non-zero: x = a; y = -+d sa; z = d ca
{
    const bool a0 = (s >> (7 * j + 5)) & 1u, d0 = (s >> (7 * j + 6)) & 1u;
    return (a0 ? 0 : 1) | ((d0 || rsig_alpha(s, j) == 1) ? 0 : 2) | (d0 ? 0 : 4);
}
constexpr RneSig rsig_of(int j, int flags, int acls, bool a0, bool d0)
{
    return (RneSig)((flags & 7) | ((acls & 3) << 3) | ((a0 ? 1 : 0) << 5) | ((d0 ? 1 : 0) << 6)) << (7 * j);
}
template <class LinkT>
inline RneSig rne_signature(const LinkT *links, int n)       // host: from the table api.cpp compiled (flags from exact zeros, sa / ca from libm)
{
    if (n < 1 || n > kRneSigMaxLinks) return 0;
    RneSig s = kRneSigPresent | kRneSigNoFriction | kRneSigNoMotor;
    for (int j = 0; j < n; ++j) {
        const LinkT &l = links[j];
        if (l.sigma != 0) return 0;
        const int acls = (l.sa == 0.0 && l.ca == 1.0) ? 1 : (l.sa == 1.0 ? 2 : (l.sa == -1.0 ? 3 : 0));
        s |= rsig_of(j, l.flags, acls, l.a == 0.0, l.d == 0.0);
        if (l.gb != 0.0 || l.Tc0 != 0.0 || l.Tc1 != 0.0) s &= ~kRneSigNoFriction;
        if (l.gjm != 0.0) s &= ~kRneSigNoMotor;
    }
    return s;
}
// Signatures with instantiations in this build (rne_kernels.hip, dyn_kernels.hip): the two DH models the reference ships with dynamics.
//   Panda (modified DH, models/DH/Panda.py:44-157): centres of mass at the link origins, alpha = 0 / -+pi/2, a or d zero on most links, links 2 and
//   6 at their predecessor's origin, no friction, no motor inertia
constexpr RneSig kRneSigPanda = kRneSigPresent | kRneSigNoFriction | kRneSigNoMotor | rsig_of(0, 1, 1, true, false) | rsig_of(1, 5, 3, true, true) |
                                rsig_of(2, 1, 2, true, false) | rsig_of(3, 1, 2, false, true) | rsig_of(4, 1, 3, false, false) | rsig_of(5, 5, 2, true, true) |
                                rsig_of(6, 1, 2, false, false);
//   Puma560 (standard DH, models/DH/Puma560.py:91-177): diagonal inertias, alpha = 0 / +-pi/2, friction and motor inertia on every joint
constexpr RneSig kRneSigPuma560 = kRneSigPresent | rsig_of(0, 3, 2, true, false) | rsig_of(1, 2, 1, false, true) | rsig_of(2, 2, 3, false, false) |
                                  rsig_of(3, 2, 2, true, false) | rsig_of(4, 7, 3, true, true) | rsig_of(5, 6, 1, true, true);
static_assert(kRneSigPanda == 0xe00047a99a2c7ea9ull && kRneSigPuma560 == 0x80000377f646a533ull, "signatures as rne_signature computes them for the shipped models");
// acc + u1 v1 - u2 v2 with the products of an absent (exactly zero) component dropped
RTB_HD double fm2(bool has1, bool has2, double u1, double v1, double u2, double v2, double acc)
{
    if (has1 && has2) return fmad(u1, v1, fmad(-u2, v2, acc));
    if (has1) return fmad(u1, v1, acc);
    if (has2) return fmad(-u2, v2, acc);
    return acc;
}
RTB_HD V3 cross_add_bm(int bm, V3 a, V3 b, V3 acc)     // acc + a x b, b's components outside bm exact zeros (7: cross_add)
{
    const bool X = bm & 1, Y = bm & 2, Z = bm & 4;
    return v3(fm2(Z, Y, a.y, b.z, a.z, b.y, acc.x), fm2(X, Z, a.z, b.x, a.x, b.z, acc.y), fm2(Y, X, a.x, b.y, a.y, b.x, acc.z));
}
RTB_HD V3 cross_add_am(int am, V3 a, V3 b, V3 acc)     // acc + a x b, a's components outside am exact zeros
{
    const bool X = am & 1, Y = am & 2, Z = am & 4;
    return v3(fm2(Y, Z, a.y, b.z, a.z, b.y, acc.x), fm2(Z, X, a.z, b.x, a.x, b.z, acc.y), fm2(X, Y, a.x, b.y, a.y, b.x, acc.z));
}
RTB_HD double df2(bool has1, bool has2, double u1, double v1, double u2, double v2)      // u1 v1 - u2 v2
{
    if (has1 && has2) return __builtin_fma(u1, v1, -(u2 * v2));      // cross()'s component; one factor an exact zero: what is left of it
    if (has1) return u1 * v1;
    if (has2) return -(u2 * v2);
    return 0.0;
}
RTB_HD V3 cross_bm(int bm, V3 a, V3 b, int &cm)        // a x b as cross_add_bm; cm: which components of the product may be non-zero
{
    const bool X = bm & 1, Y = bm & 2, Z = bm & 4;
    cm = ((Z || Y) ? 1 : 0) | ((X || Z) ? 2 : 0) | ((Y || X) ? 4 : 0);
    return v3(df2(Z, Y, a.y, b.z, a.z, b.y), df2(X, Z, a.z, b.x, a.x, b.z), df2(Y, X, a.x, b.y, a.y, b.x));
}
// acc + w x (w x p*) + wd x p*: the centripetal and tangential terms of a link offset, p*'s zero components dropped (pm = 7: the general chain)
RTB_HD V3 offset_accel(int pm, V3 w, V3 wd, V3 ps, V3 acc)
{
    if (pm == 7) return cross_add(wd, ps, cross_add(w, cross(w, ps), acc));
    int cm;
    const V3 c = cross_bm(pm, w, ps, cm);
    return cross_add_bm(pm, wd, ps, cross_add_bm(cm, w, c, acc));
}

// acc + R^T v and acc + R v as chains of fused multiply-adds seeded with acc; alpha's part through kacc (exactform.h): the class forms are the
// general chain with  fma(0, x, a) -> a  and  fma(+-1, x, a) -> a +- x  -- the same bits by construction
template <bool MDH>
RTB_HD V3 rot_inv_add(const Rot &r, V3 v, V3 acc)   // acc + R^T v
{
#pragma clang fp contract(off)
    const int ks = rot_ksa(r.acls), kc = rot_kca(r.acls);
    if (!MDH) {
        const double uy = fmad(r.c, v.y, -(r.s * v.x));
        return v3(fmad(r.c, v.x, fmad(r.s, v.y, acc.x)), kacc(kc, r.ca, uy, kacc(ks, r.sa, v.z, acc.y)), kacc(kc, r.ca, v.z, kacc(ks, r.sa, -uy, acc.z)));
    }
    const double uy = alpha_mix(r, v.z, v.y);
    return v3(fmad(r.c, v.x, fmad(r.s, uy, acc.x)), fmad(r.c, uy, fmad(-r.s, v.x, acc.y)), kacc(kc, r.ca, v.z, kacc(ks, r.sa, -v.y, acc.z)));
}
template <bool MDH>
RTB_HD V3 rot_fwd_add(const Rot &r, V3 v, V3 acc)   // acc + R v
{
#pragma clang fp contract(off)
    const int ks = rot_ksa(r.acls), kc = rot_kca(r.acls);
    if (!MDH) {
        const double uy = alpha_mix(r, -v.z, v.y);
        return v3(fmad(r.c, v.x, fmad(-r.s, uy, acc.x)), fmad(r.s, v.x, fmad(r.c, uy, acc.y)), kacc(ks, r.sa, v.y, kacc(kc, r.ca, v.z, acc.z)));
    }
    const double uy = fmad(r.s, v.x, r.c * v.y);
    return v3(fmad(r.c, v.x, fmad(-r.s, v.y, acc.x)), kacc(kc, r.ca, uy, kacc(ks, r.sa, -v.z, acc.y)), kacc(ks, r.sa, uy, kacc(kc, r.ca, v.z, acc.z)));
}

// Everything one forward step reads from the link table and from the q / qd / qdd tile, fetched as ONE batch.
// RTB_RNE_PREFETCH=1 (A/B knob, off): the batch of step j+1 is issued at the top of step j, so that the scalar-memory and LDS
// round trips (about 300 cycles each; SQ counters: a third of the wave's lifetime in s_waitcnt) overlap a whole step of
// arithmetic.  Measured on MI355X (1.25e6 Panda triples, kernel min of 30): 0.0648 ms without, 0.0798 ms with -- the second
// operand set costs 26 VGPRs (182: two waves per SIMD instead of three), and forced back to three waves it spills (0.083 ms).
// (s_waitcnt lgkmcnt counts scalar loads and LDS reads together and scalar loads return out of order: any wait is a wait for
// everything outstanding, so a prefetch has to carry ALL of the next step's operands.)
struct FwdOps { LinkFwd f; double qj, qdj, qddj; };
template <bool ALLREV, class LinkT, class InQ, class InQd, class InQdd>
RTB_HD FwdOps fwd_ops(const LinkT &lt, int j, InQ qin, InQd qdin, InQdd qddin)
{
    FwdOps o;
    o.f = link_fwd<ALLREV>(lt);
    o.qj = (!ALLREV && o.f.sigma != 0) ? qin(j) : 0.0;       // only a prismatic link needs q again (its d)
    o.qdj = qdin(j); o.qddj = qddin(j);
    return o;
}
struct BwdOps { LinkBwd b; double rx, ry, rz; double qj, qdj, qddj; };
template <bool ALLREV, bool FRICTION, class LinkT, class InQ, class InQd, class InQdd>
RTB_HD BwdOps bwd_ops(const LinkT &lt, int flags, int j, InQ qin, InQd qdin, InQdd qddin)
{
    BwdOps o;
    o.b = link_bwd<ALLREV, FRICTION>(lt);
    if (flags & kLinkRZero) { o.rx = 0.0; o.ry = 0.0; o.rz = 0.0; }
    else { o.rx = lt.rx; o.ry = lt.ry; o.rz = lt.rz; }
    o.qj = (!ALLREV && o.b.sigma != 0) ? qin(j) : 0.0;
    o.qdj = qdin(j); o.qddj = qddin(j);
    return o;
}

// ACC (all-revolute chains, compile-time n): the pass is known to run at qd = 0 and zero gravity with the links before `first`
// at rest -- what Dynamics.inertia / accel ask for column `first` of M(q): qdd = e_first (robot/Dynamics.py:752-758, 492-496).
// Then w = 0 throughout, every link before `first` has F = N = 0, and -- M being symmetric -- only the torques of joints
// >= first are needed: the forward recursion starts at `first` without its velocity terms, the backward one stops there, and
// the caller mirrors the column.  The terms left out are exact zeros of the general formulas (0 x + y = y), so the values are
// those of the full pass (up to the sign of a zero); the mirrored half differs from the reference's separately rounded entries
// by a few ulp.  ~2.3x fewer fp64 operations per column of a 7-joint arm.
// Base wrench (DHRobot.rne(base_wrench=True) -> rne_python, robot/DHRobot.py:1765-1770): the force and moment the first link
// exerts on the base, `R_1 f_1` and `R_1 n_1` -- what the backward recursion holds when it ends, rotated into frame 0.  The
// receiver is a functor `wout(k, v)`, k = 0..5 = (f, n); NoWrench compiles the epilogue away.
struct NoWrench {
    static constexpr bool on = false;
    RTB_HD bool wanted() const { return false; }
    RTB_HD void operator()(int, double) const {}
};

// PSZ: honour the p* = 0 link flag (kLinkPsZero).  The single-pass kernels do (k_rne: -1.4 %); the multi-pass dynamics kernels do not -- there the
// extra wave-uniform branches in every inlined pass cost 6-16 % (round 3, visit z: the round-2 tree beside this one on one box).
template <int NJ, bool MDH, bool FRICTION_ARG, bool ALLREV, bool HAVE_TRIG, bool ACC = false, bool PSZ_ARG = true, RneSig SIG = 0, class LinksP, class InQ, class InQd, class InQdd,
          class Out, class WOut = NoWrench>
RTB_HD void rne_core(LinksP links, int n_rt, double (&st)[NJ > 0 ? NJ : RTBHIP_MAX_JOINTS], double (&ct)[NJ > 0 ? NJ : RTBHIP_MAX_JOINTS],
                     V3 grav, V3 ftip, V3 ntip, InQ qin, InQd qdin, InQdd qddin, Out tau, int first = 0, WOut wout = WOut())
{
    static_assert(!ACC || (ALLREV && NJ > 0 && HAVE_TRIG), "the acceleration-only pass is built for all-revolute chains with compile-time n");
    static_assert(SIG == 0 || (ALLREV && NJ > 0 && NJ <= kRneSigMaxLinks && HAVE_TRIG), "a structure signature describes an all-revolute chain with compile-time n");
    constexpr bool FRICTION = FRICTION_ARG && !(SIG & kRneSigNoFriction);
    constexpr bool PSZ = PSZ_ARG || SIG != 0;            // compile-time flags cost nothing
    if constexpr (ACC) {
        // `first` is the caller's pass counter: keep the optimiser from splitting the caller's loop into one specialised copy of this
        // whole recursion per value of it (the host build of tests/emu did not finish in 45 minutes); on the device it stays scalar
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+s"(first));
#else
        asm volatile("" : "+r"(first));
#endif
    }
    constexpr int CAP = NJ > 0 ? NJ : RTBHIP_MAX_JOINTS;
    const int n = NJ > 0 ? NJ : n_rt;
    V3 F[CAP], Nn[CAP];
    int flg[CAP];                  // all the links' shortcut flags in one batch of scalar loads, ahead of the branches on them
#pragma unroll
    for (int j = 0; j < n; ++j) flg[j] = SIG ? rsig_flags(SIG, j) : links[j].flags;

    // ---- forward recursion (ne.c:133-348)
    // (ACC: the gravity field enters as the base's linear acceleration -- the general formulas inject it at link 0 in exactly that
    // role -- so a pass at qd = 0 WITH gravity, `first` = 0, is Dynamics.gravload / the qd = NULL calls of rtbhip_rne)
    V3 w = v3(0, 0, 0), wd = v3(0, 0, 0), a = ACC ? grav : v3(0, 0, 0);
    double qddx = 0.0, qddy = 0.0;  // ne.c:311 lets gravity leak into qddv.x/.y for later links
    constexpr bool PF = RTB_RNE_PREFETCH != 0 && NJ > 0;
    FwdOps cur = fwd_ops<ALLREV>(links[0], 0, qin, qdin, qddin);
#pragma unroll
    for (int j = 0; j < n; ++j) {
        if (!PF && j > 0) cur = fwd_ops<ALLREV>(links[j], j, qin, qdin, qddin);
        FwdOps nxt = cur;
        if (PF) sched_fence();     // cur's wait sits above this line, the next batch below it
        if (PF && j + 1 < n) nxt = fwd_ops<ALLREV>(links[j + 1], j + 1, qin, qdin, qddin);
        const auto &li = links[j];                    // r and I: loaded inside the branches that need them
        const LinkFwd &l = cur.f;
        const bool pris = ALLREV ? false : (l.sigma != 0);
        const double qdj = cur.qdj, qddj = cur.qddj;
        if (NJ == 0) {
            const double th = pris ? l.theta : qin(j) + l.offset;
            rtb_sincos(th, &st[j], &ct[j]);
        }
        const double d = pris ? (NJ == 0 ? qin(j) : cur.qj) + l.offset : l.d;
        const int acls = SIG ? rsig_alpha(SIG, j) : 0, pm = SIG ? rsig_pmask(SIG, j) : 7;      // constants of the unrolled copy
        const Rot R = {st[j], ct[j], l.sa, l.ca, acls};
        const V3 ps = link_offset<MDH>(l, d);
        if constexpr (ACC) {
            if (j >= first) {          // wave-uniform
                V3 wdn, an;
                const bool ps0 = PSZ && (flg[j] & kLinkPsZero) != 0;      // wave-uniform: p* = 0, the cross products with it are exact zeros
                if (MDH) {             // w = 0:  wd' = R^T wd + z qdd,  a' = R^T (a + wd x p*)
                    wdn = addz(rot_inv<MDH>(R, wd), qddj);
                    an = rot_inv<MDH>(R, ps0 ? a : cross_add_bm(pm, wd, ps, a));
                } else {               //         wd' = R^T (wd + z qdd),  a' = wd' x p* + R^T a
                    wdn = rot_inv<MDH>(R, v3(wd.x, wd.y, wd.z + qddj));
                    an = ps0 ? rot_inv<MDH>(R, a) : cross_add_bm(pm, wdn, ps, rot_inv<MDH>(R, a));
                }
                wd = wdn; a = an;
                V3 ac = a;
                if (!(flg[j] & kLinkRZero)) ac = cross_add(wd, v3(li.rx, li.ry, li.rz), a);
                F[j] = l.m * ac;
                Nn[j] = (flg[j] & kLinkIDiag) ? v3(li.I[0] * wd.x, li.I[4] * wd.y, li.I[8] * wd.z) : inertia_times(li, wd);
            }
            cur = nxt;
            if (!PF && !RTB_RNE_NOFENCE) sched_fence();
            continue;
        }
        const V3 qdv = v3(0, 0, qdj);
        V3 qddv = v3(qddx, qddy, qddj);
        V3 wn, wdn, an;
        if (MDH) {
            if (!pris) {
                if (j == 0) {
                    wn = qdv; wdn = qddv; an = rot_inv<MDH>(R, grav);
                } else {
                    const V3 t1 = rot_inv<MDH>(R, w);
                    wn = addz(t1, qdj);
                    // crossz(t1, qdj) + R^T wd [+ qddv]
                    const V3 u = rot_inv_add<MDH>(R, wd, v3(t1.y * qdj, -(t1.x * qdj), 0.0));
                    wdn = ALLREV ? addz(u, qddj) : u + qddv;
                    // (p* = 0 -- a link whose origin coincides with its predecessor's: wave-uniform -- leaves a' = R^T a)
                    an = rot_inv<MDH>(R, (PSZ && (flg[j] & kLinkPsZero)) ? a : offset_accel(pm, w, wd, ps, a));
                }
            } else {
                if (j == 0) {
                    wn = qdv; wdn = qddv; an = grav;
                } else {
                    wn = rot_inv<MDH>(R, w);
                    wdn = rot_inv<MDH>(R, wd);
                    an = rot_inv<MDH>(R, cross_add(wd, ps, cross_add(w, cross(w, ps), a)));
                    an = (an + 2.0 * cross(wn, qdv)) + qddv;
                }
            }
        } else {
            if (!pris) {
                const V3 t1 = (j == 0) ? qdv : addz(w, qdj);
                wn = rot_inv<MDH>(R, t1);
                // (wd + qddv) + crossz(w, qdj)
                const V3 t3 = (j == 0) ? qddv : (ALLREV ? v3(fmad(w.y, qdj, wd.x), fmad(-w.x, qdj, wd.y), wd.z + qddj)
                                                         : (wd + qddv) + crossz(w, qdj));
                wdn = rot_inv<MDH>(R, t3);
                {
                    const V3 ra = rot_inv<MDH>(R, (j == 0) ? grav : a);
                    an = (PSZ && (flg[j] & kLinkPsZero)) ? ra : offset_accel(pm, wn, wdn, ps, ra);
                }
            } else {
                wn = (j == 0) ? v3(0, 0, 0) : rot_inv<MDH>(R, w);
                wdn = (j == 0) ? v3(0, 0, 0) : rot_inv<MDH>(R, wd);
                if (j == 0) {
                    qddv = qddv + grav;
                    qddx = qddv.x; qddy = qddv.y;
                    an = rot_inv<MDH>(R, qddv);
                } else {
                    an = rot_inv<MDH>(R, qddv + a);
                }
                an = an + cross(wdn, ps);
                an = an + 2.0 * cross(wn, rot_inv<MDH>(R, qdv));
                an = an + cross(wn, cross(wn, ps));
            }
        }
        w = wn; wd = wdn; a = an;
        // Exact zeros in the link table are skipped through wave-uniform branches (scalar loads + s_cbranch): a centre of
        // mass at the link origin (the DH Panda) removes 33 fp64 operations per link, a diagonal inertia tensor (Puma560)
        // 12; the values are those of the general formulas (up to the sign of a zero)
        V3 ac = a;
        if (!(flg[j] & kLinkRZero)) {
            const V3 rc = v3(li.rx, li.ry, li.rz);
            ac = cross_add(wd, rc, cross_add(w, cross(w, rc), a));     // ne.c:228-232
        }
        F[j] = l.m * ac;
        if (flg[j] & kLinkIDiag) {
            const V3 iw = v3(li.I[0] * w.x, li.I[4] * w.y, li.I[8] * w.z);
            Nn[j] = cross_add(w, iw, v3(li.I[0] * wd.x, li.I[4] * wd.y, li.I[8] * wd.z));
        } else {
            Nn[j] = cross_add(w, inertia_times(li, w), inertia_times(li, wd));
        }
        cur = nxt;
        if (!PF && NJ > 0 && !RTB_RNE_NOFENCE) sched_fence();
    }

    // ---- backward recursion + joint projection (ne.c:354-492), fused
    V3 f = ftip, nn = ntip;   // f_{j+1}, n_{j+1}; the reference's "tip" values for the last link
    Rot Rn = {0, 1, 0, 1, 0};    // frame of link j+1
    V3 psn = v3(0, 0, 0);
    int pmn = 7;
    BwdOps bc = bwd_ops<ALLREV, FRICTION>(links[n - 1], flg[n - 1], n - 1, qin, qdin, qddin);
#pragma unroll
    for (int jj = 0; jj < n; ++jj) {
        const int j = n - 1 - jj;
        if (ACC && j < first) break;   // wave-uniform: the torques of the joints before `first` come from the mirror
        if (!PF && jj > 0) bc = bwd_ops<ALLREV, FRICTION>(links[j], flg[j], j, qin, qdin, qddin);
        BwdOps bn = bc;
        if (PF) sched_fence();
        if (PF && j > 0) bn = bwd_ops<ALLREV, FRICTION>(links[j - 1], flg[j - 1], j - 1, qin, qdin, qddin);
        const LinkBwd &l = bc.b;
        const bool last = (jj == 0);
        const bool pris = ALLREV ? false : (l.sigma != 0);
        const bool rzero = (flg[j] & kLinkRZero) != 0;
        const V3 rc = v3(bc.rx, bc.ry, bc.rz);
        const double d = pris ? (NJ == 0 ? qin(j) : bc.qj) + l.offset : l.d;
        const int acls = SIG ? rsig_alpha(SIG, j) : 0, pm = SIG ? rsig_pmask(SIG, j) : 7;
        const Rot R = {st[j], ct[j], l.sa, l.ca, acls};
        const V3 ps = link_offset<MDH>(l, d);
        V3 fj, nj;
        if (MDH) {
            const V3 fn = last ? f : rot_fwd<MDH>(Rn, f);
            fj = fn + F[j];
            const V3 base = rzero ? Nn[j] : cross_add(rc, F[j], Nn[j]);
            // (psn = p* of link j + 1: zero for a kLinkPsZero link, wave-uniform)
            if (last) nj = nn + base;
            else if (PSZ && (flg[j + 1 < n ? j + 1 : j] & kLinkPsZero)) nj = rot_fwd_add<MDH>(Rn, nn, base);
            else nj = cross_add_am(pmn, psn, fn, rot_fwd_add<MDH>(Rn, nn, base));
        } else {
            fj = last ? F[j] + f : rot_fwd_add<MDH>(Rn, f, F[j]);
            const V3 base = cross_add(ps + rc, F[j], Nn[j]);
            if (!last) nj = rot_fwd_add<MDH>(Rn, cross_add(rot_inv<MDH>(Rn, ps), f, nn), base);
            else nj = cross_add(ps, f, nn + base);
        }
        // joint axis in the link frame: z for MDH, R^T z = (0, sin alpha, cos alpha) for DH
        const V3 prj = pris ? fj : nj;
        const double qdj = bc.qdj, qddj = bc.qddj;
        double t = MDH ? prj.z : fmad(l.sa, prj.y, l.ca * prj.z);
        if (!(SIG & kRneSigNoMotor)) t = fmad(l.gjm, qddj, t);
        if (FRICTION) {
            t = fmad(l.gb, qdj, t);
            t = fmad(l.ag, (qdj > 0 ? l.Tc0 : 0.0) + (qdj < 0 ? l.Tc1 : 0.0), t);
        }
        tau(j, t);
        f = fj; nn = nj; Rn = R; psn = ps; pmn = pm;
        bc = bn;
        if (!PF && NJ > 0 && !RTB_RNE_NOFENCE) sched_fence();
    }
    if constexpr (WOut::on && !ACC) {
        if (wout.wanted()) {           // wave-uniform
            const V3 fb = rot_fwd<MDH>(Rn, f), nb = rot_fwd<MDH>(Rn, nn);
            wout(0, fb.x); wout(1, fb.y); wout(2, fb.z);
            wout(3, nb.x); wout(4, nb.y); wout(5, nb.z);
        }
    }
}

// One sample, trig included (the single-pass kernels and the run-time-n path).
// ATREST: the caller knows qd = 0 for every joint (rtbhip_rne with qd = NULL: Dynamics.gravload, Dynamics.itorque): the
// acceleration-only recursion from link 0, whole backward pass (all-revolute chains with compile-time n)
template <int NJ, bool MDH, bool FRICTION = true, bool ALLREV = false, bool ATREST = false, RneSig SIG = 0, class LinksP, class InQ, class InQd, class InQdd, class Out,
          class WOut = NoWrench>
RTB_HD void rne_lane(LinksP links, int n_rt, V3 grav, V3 ftip, V3 ntip, InQ qin, InQd qdin, InQdd qddin, Out tau, WOut wout = WOut())
{
    constexpr int CAP = NJ > 0 ? NJ : RTBHIP_MAX_JOINTS;
    double st[CAP], ct[CAP];
    if constexpr (NJ > 0) rne_trig<NJ, ALLREV>(links, qin, st, ct);
    if constexpr (ATREST) rne_core<NJ, MDH, false, true, true, true, false, SIG>(links, n_rt, st, ct, grav, ftip, ntip, qin, qdin, qddin, tau, 0);   // (PSZ off: +7 % on gravload with it, visit z)
    else rne_core<NJ, MDH, FRICTION, ALLREV, (NJ > 0), false, true, SIG>(links, n_rt, st, ct, grav, ftip, ntip, qin, qdin, qddin, tau, 0, wout);
}

#pragma clang fp contract(fast)      // (what follows this header is compiled as before)
}  // namespace rtbhip
