// tree_device.h -- per-lane recursive Newton-Euler for ETS robots (link trees with point... spatial
// inertias), the device restatement of Robot.rne (reference robot/Robot.py:1704-1903).
//
// The reference walks "link groups" (every joint link together with the static links that precede it
// in link order, :1777-1789) with spatialmath's 6-D spatial vectors: v, a per group from the parent
// group's (:1853-1870), f = I a + v x* I v (:1872), then tau_j = s_j . f_j and f_parent += X^T f_j
// backwards (:1875-1893).  Here one lane owns one (q, qd, qdd) sample and the 6-vectors are pairs of
// 3-vectors in registers ([linear; angular], spatialmath's order); the group transform is the
// canonical segment  C_j Z_j(q)  of chain.cpp (constant affine, then a rotation about / slide along
// the local z axis), so  X_up  is "constant 3x3 transpose times vector, then a planar rotation".
//
// Tree support without per-group state arrays in LDS: groups are processed in the reference's
// (topological) order; a group whose parent is not the immediately preceding group reads the parent's
// (v, a) from a small LDS slot that the parent wrote, and in the backward pass adds its transformed
// force into that slot instead of handing it on in registers.  A serial chain uses no slot at all.
//
// Reference quirks reproduced on purpose (the oracle restates them too): the group inertia is the plain
// sum of the member links' SpatialInertia(m, r) -- no inertia tensor, no re-expression of a static
// link's centre of mass in the joint frame (:1793-1800); a flipped joint negates the angle in the
// transform but not the motion subspace (robot/ET.py:592-608 ignores `flip`); no friction, no motor
// inertia; gravity enters as the base acceleration -g (:1804-1807).
#pragma once
#include "dyn_device.h"

namespace rtbhip {
#pragma clang fp contract(off)      // every operation written out, as in rne_device.h (why: there)

struct alignas(16) DevGroup {   // wave-uniform, read through the scalar cache
    DevSeg C;                   // parent-group frame -> this group's joint frame before the joint motion
    double M, h[3];             // mass and first moment (sum m r) in the group frame
    double I[6];                // rotational inertia about the group-frame origin: xx, yy, zz, xy, xz, yz
    int32_t parent;             // parent group (-1: attached to the base)
    int32_t jmeta;              // bit 0 prismatic, bits 8..15 q column, bit 16 flip
    int32_t save_slot;          // >= 0: this group's (v, a) and force accumulator live in that LDS slot
    int32_t parent_slot;        // >= 0: parent != previous group -> parent state is read from that slot
    int32_t out_col;            // column of tau (the reference's group order)
    int32_t pad[3];
};
static_assert(sizeof(DevGroup) == 208, "DevGroup layout");

constexpr int kTreeSlotDoubles = 18;   // v (6), a (6), force accumulator (6)
constexpr int kTreeBilinearSlotDoubles = 24;   // tree_bilinear_core: v of both velocity fields (12), a (6), force accumulator (6)

RTB_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }

// R_C^T v and R_C v for a group constant of structure class cls (0: general).  ONE form for every class: each component is dotk_rt (exactform.h) -- the
// general product's fixed operation sequence  fma(c2, z, fma(c0, x, round(c1 y)))  with the entries the class knows to be exact 0 / +-1 rewritten
// away -- so a kernel instantiated for a robot's classes returns the general kernel's bits by construction.  cls is a constant of the unrolled
// group step in every caller: the kind lookups fold, entries whose kind is not kCA are never loaded.
template <class G> RTB_HD V3 seg_rt_c(int cls, const G &g, V3 v)   // R_C^T v: component k = r[k] x + r[3 + k] y + r[6 + k] z
{
    const auto &r = g.C.r;
    return v3(dotk_rt(seg_kind(cls, 3), seg_kind(cls, 0), seg_kind(cls, 6), r[3], v.y, r[0], v.x, r[6], v.z),
              dotk_rt(seg_kind(cls, 4), seg_kind(cls, 1), seg_kind(cls, 7), r[4], v.y, r[1], v.x, r[7], v.z),
              dotk_rt(seg_kind(cls, 5), seg_kind(cls, 2), seg_kind(cls, 8), r[5], v.y, r[2], v.x, r[8], v.z));
}
template <class G> RTB_HD V3 seg_r_c(int cls, const G &g, V3 v)    // R_C v: component k = r[3k] x + r[3k + 1] y + r[3k + 2] z
{
    const auto &r = g.C.r;
    return v3(dotk_rt(seg_kind(cls, 1), seg_kind(cls, 0), seg_kind(cls, 2), r[1], v.y, r[0], v.x, r[2], v.z),
              dotk_rt(seg_kind(cls, 4), seg_kind(cls, 3), seg_kind(cls, 5), r[4], v.y, r[3], v.x, r[5], v.z),
              dotk_rt(seg_kind(cls, 7), seg_kind(cls, 6), seg_kind(cls, 8), r[7], v.y, r[6], v.x, r[8], v.z));
}
template <class G> RTB_HD V3 seg_rt(const G &g, V3 v) { return seg_rt_c(kSegGeneral, g, v); }
template <class G> RTB_HD V3 seg_r(const G &g, V3 v) { return seg_r_c(kSegGeneral, g, v); }
RTB_HD V3 rz_t(double s, double c, V3 v) { return v3(fmad(c, v.x, s * v.y), fmad(c, v.y, -(s * v.x)), v.z); }   // Rz(theta)^T v
RTB_HD V3 rz(double s, double c, V3 v) { return v3(fmad(c, v.x, -(s * v.y)), fmad(s, v.x, c * v.y), v.z); }     // Rz(theta) v
RTB_HD V3 crossz_add(V3 v, double t, V3 acc) { return v3(fmad(v.y, t, acc.x), fmad(-v.x, t, acc.y), acc.z); }   // acc + v x (0, 0, t)

// ---- STRUCTURE SIGNATURES (rtbhip_internal.h: SegSig, kSeg*).  73 % of the group constants of the URDF robots are not general rotations
// (identity 21 %, the cyclic permutations of the axis conjugation 23 %, quarter turns 18 %, one-axis rotations 11 %), and most of their
// translations have one or two non-zero components.  A core instantiated for a signature (SIG != 0) multiplies by every group constant in the
// form of its class and drops the cross-product terms of the zero translation components; the class is read from SIG at the unrolled
// group index -- a compile-time constant in every copy of the loop body, so the `if` chains below fold away (a run-time switch costs more
// than it saves: profiles/r05_ik_structured_constants.txt).  The forms are the general products' operation sequences with the exact zeros dropped
// and the exact +-1 taken as the operand (exactform.h): the general kernel's bits, by construction.
// kTreeSigPlain: additionally a serial chain of revolute joints (parent of group j is group j - 1, no branch slots, no prismatic joint) --
// the parent selection, slot traffic and prismatic branches are not compiled in: one straight-line basic block per group.
constexpr SegSig kTreeSigPlain = 1ull << 56;
// kTreeSigAnyConstants (with kTreeSigPlain): nothing is assumed about the group constants -- every one is multiplied as a general rotation, every
// translation component may be non-zero.  The instantiation for "a serial chain of NG revolute joints", whatever the robot: most of what a
// signature buys is the straight-line code (UR5: classes + masks + plain 1.65-2.3x; the branched Interbotix arms, classes alone, 1.08-1.14x).
constexpr SegSig kTreeSigAnyConstants = 1ull << 57;
constexpr SegSig kTreeSigPlainChain = kSegSigPresent | kTreeSigPlain | kTreeSigAnyConstants;
constexpr int kTreePlainChainMax = 8;          // sizes with a plain-chain instantiation (tree_kernels.hip)
// A BRANCHED tree's bookkeeping at compile time: TreeTopo, 12 bits per group -- parent + 1 (4 bits; 0 = the base), prismatic (1), parent slot + 1 (3),
// save slot + 1 (3) -- bit 127 = present; for trees of up to 10 groups numbered in group order (group j moves q column j and owns torque column j).
// With it the parent selection, the branch-slot addresses and the joint kind are constants of each unrolled group step, as kTreeSigPlain makes
// them for a serial chain, and the translation masks of the signature apply to every non-prismatic group.
constexpr TreeTopo kTreeTopoPresent = (TreeTopo)1 << 127;
constexpr int kTreeTopoMaxGroups = 10, kTreeSig2MaxGroups = 16;
RTB_HD constexpr int topo_parent(TreeTopo t, int j) { return (int)((unsigned)(t >> (12 * j)) & 15u) - 1; }
RTB_HD constexpr bool topo_pris(TreeTopo t, int j) { return ((unsigned)(t >> (12 * j + 4)) & 1u) != 0; }
RTB_HD constexpr int topo_parent_slot(TreeTopo t, int j) { return (int)((unsigned)(t >> (12 * j + 5)) & 7u) - 1; }
RTB_HD constexpr int topo_save_slot(TreeTopo t, int j) { return (int)((unsigned)(t >> (12 * j + 8)) & 7u) - 1; }
constexpr TreeTopo topo_of(int j, int parent, bool pris, int parent_slot, int save_slot)
{
    return (TreeTopo)((unsigned)(parent + 1) | ((pris ? 1u : 0u) << 4) | ((unsigned)(parent_slot + 1) << 5) | ((unsigned)(save_slot + 1) << 8)) << (12 * j);
}
constexpr int kTreeSigMaxGroups = 8;
// The signatures with instantiations in this build (tree_kernels.hip; tree.cpp: tree_signature computes a robot's).  UR3 / UR5 / UR10 read
// from their URDF: base translation; the shoulder's rpy = (0, pi/2, 0) with the file's 12-digit pi (general); a pure translation; a quarter
// turn about z; the two cyclic permutations of the axis conjugation -- five of six constants structured, one or two translation components.
constexpr SegSig kTreeSigUR = kSegSigPresent | kTreeSigPlain | seg_sig_of(0, kSegIdentity, 4) | seg_sig_of(1, kSegGeneral, 2) | seg_sig_of(2, kSegIdentity, 5) |
                              seg_sig_of(3, kSegRzP, 1) | seg_sig_of(4, kSegPermA, 4) | seg_sig_of(5, kSegPermB, 4);
// The Interbotix arms with eight link groups (px150, rx150, rx200, vx300, wx200, wx250: one signature): a branched tree -- two prismatic
// gripper fingers hang off the wrist -- so not plain; seven of the eight constants are multiplied in the form of their class.
constexpr SegSig kTreeSigIbx8 = kSegSigPresent | seg_sig_of(0, kSegIdentity, 4) | seg_sig_of(1, kSegPermB, 4) | seg_sig_of(2, kSegRy, 3) | seg_sig_of(3, kSegIdentity, 2) |
                                seg_sig_of(4, kSegGeneral, 2) | seg_sig_of(5, kSegIdentity, 4) | seg_sig_of(6, kSegPermA, 4) | seg_sig_of(7, kSegIdentity, 0);
static_assert(kTreeSigIbx8 == 0x80032e0a042de641ull && kTreeSigUR == 0x81000264b3145041ull, "signatures as tree.cpp computes them for the URDF files");
// ... and their bookkeeping (tree.cpp: tree_topology): the group table the reference's grouping gives them is a serial chain -- six revolute arm joints, then
// the two prismatic fingers one after the other -- so with the joint kinds known the whole recursion is straight-line code
constexpr TreeTopo kTreeTopoIbx8 = kTreeTopoPresent | topo_of(0, -1, false, -1, -1) | topo_of(1, 0, false, -1, -1) | topo_of(2, 1, false, -1, -1) | topo_of(3, 2, false, -1, -1) |
                                   topo_of(4, 3, false, -1, -1) | topo_of(5, 4, false, -1, -1) | topo_of(6, 5, true, -1, -1) | topo_of(7, 6, true, -1, -1);
static_assert((unsigned long long)(kTreeTopoIbx8 >> 64) == 0x8000000001701600ull && (unsigned long long)kTreeTopoIbx8 == 0x5004003002001000ull, "as tree_topology computes it");
// Longer trees (9 .. 10 groups: a second class word; scripts/print_signatures.py prints all three words with their fields):
//   vx300s / wx250s   nine groups in series, seven revolute + the two prismatic fingers           classes I pB Ry pB pA G I pA | I
//   Fetch             ten groups in series, the prismatic torso lift and its second slide first   classes I pA G I pB pB pA pB | pA pB
//   Mico              ten groups, three two-joint fingers off the wrist (one branch slot)         classes Ry G Ry G G I G I | G I
constexpr TreeTopo topo_words(unsigned long long hi, unsigned long long lo) { return ((TreeTopo)hi << 64) | lo; }
constexpr SegSig kTreeSigIbx9 = 0x80970504b58de641ull, kTreeSig2Ibx9 = 0x8000000000000001ull;
constexpr TreeTopo kTreeTopoIbx9 = topo_words(0x8000001801700600ull, 0x5004003002001000ull);
constexpr SegSig kTreeSigFetch = 0x80592d65ca380581ull, kTreeSig2Fetch = 0x800000000000164bull;
constexpr TreeTopo kTreeTopoFetch = topo_words(0x8000900800700600ull, 0x5004003012011000ull);
constexpr SegSig kTreeSigMico = 0x8063c18f0c09f047ull, kTreeSig2Mico = 0x80000000000018f0ull;
constexpr TreeTopo kTreeTopoMico = topo_words(0x8000902400702400ull, 0x5004103002001000ull);
// px100, the 7-group member of that family (five revolute joints and the two fingers): scripts/print_signatures.py prints these words for any robot
constexpr SegSig kTreeSigPx100 = kSegSigPresent | seg_sig_of(0, kSegIdentity, 4) | seg_sig_of(1, kSegPermB, 4) | seg_sig_of(2, kSegRy, 3) | seg_sig_of(3, kSegIdentity, 2) |
                                 seg_sig_of(4, kSegGeneral, 2) | seg_sig_of(5, kSegPermA, 4) | seg_sig_of(6, kSegIdentity, 0);
constexpr TreeTopo kTreeTopoPx100 = kTreeTopoPresent | topo_of(0, -1, false, -1, -1) | topo_of(1, 0, false, -1, -1) | topo_of(2, 1, false, -1, -1) | topo_of(3, 2, false, -1, -1) |
                                    topo_of(4, 3, false, -1, -1) | topo_of(5, 4, true, -1, -1) | topo_of(6, 5, true, -1, -1);
static_assert(kTreeSigPx100 == 0x8000065a042de641ull && (unsigned long long)(kTreeTopoPx100 >> 64) == 0x8000000000001601ull &&
              (unsigned long long)kTreeTopoPx100 == 0x5004003002001000ull, "as tree.cpp computes them for the URDF file");

// b + (u1 v1 - u2 v2) as  fma(u1, v1, fma(-u2, v2, b))  with the products of an absent (exactly zero) translation component dropped (exactform.h: fm2k)
RTB_HD double tree_acc2(bool has1, bool has2, double b, double u1, double v1, double u2, double v2) { return fm2k(has1, has2, u1, v1, u2, v2, b); }
RTB_HD V3 add_cross_ap(int tm, V3 b, V3 a, V3 p)      // b + a x p;  tm: which components of p are not exact zeros (7: no knowledge)
{
    const bool X = tm & 1, Y = tm & 2, Z = tm & 4;
    return v3(tree_acc2(Z, Y, b.x, a.y, p.z, a.z, p.y), tree_acc2(X, Z, b.y, a.z, p.x, a.x, p.z), tree_acc2(Y, X, b.z, a.x, p.y, a.y, p.x));
}
RTB_HD V3 add_cross_pa(int tm, V3 b, V3 p, V3 a)      // b + p x a
{
    const bool X = tm & 1, Y = tm & 2, Z = tm & 4;
    return v3(tree_acc2(Y, Z, b.x, p.y, a.z, p.z, a.y), tree_acc2(Z, X, b.y, p.z, a.x, p.x, a.z), tree_acc2(X, Y, b.z, p.x, a.y, p.y, a.x));
}
template <class G> RTB_HD V3 tree_origin(bool revolute, const G &g, double d)      // p = t_C (+ R_C z d for a prismatic joint)
{
    if (revolute) return v3(g.C.t[0], g.C.t[1], g.C.t[2]);
    return v3(fmad(g.C.r[2], d, g.C.t[0]), fmad(g.C.r[5], d, g.C.t[1]), fmad(g.C.r[8], d, g.C.t[2]));
}
// what a core knows about group j at compile time (SIG = 0: nothing)
// (groups 8 .. 15 of a longer tree have their fields in a second word, SIG2, at positions 0 .. 7)
template <SegSig SIG, SegSig SIG2 = 0> RTB_HD constexpr int tree_cls(int j)
{
    return (SIG && !(SIG & kTreeSigAnyConstants)) ? (j < kTreeSigMaxGroups ? seg_sig_cls(SIG, j) : seg_sig_cls(SIG2, j - kTreeSigMaxGroups)) : kSegGeneral;
}
template <SegSig SIG, SegSig SIG2 = 0> RTB_HD constexpr int tree_tm(int j, bool revolute)      // a prismatic joint adds R z d to p: masks only where the joint is known to be revolute
{
    return (SIG && !(SIG & kTreeSigAnyConstants) && revolute) ? (j < kTreeSigMaxGroups ? seg_sig_tm(SIG, j) : seg_sig_tm(SIG2, j - kTreeSigMaxGroups)) : 7;
}
// What a core knows about the robot at compile time -- a KNOWLEDGE type KN with static constexpr members:
//   known      the classes / translation masks of the group constants (cls, tm)          plain   serial chain of revolute joints in group order
//   topo       every group's parent, joint kind and branch slots (parent, pris, ...)     any     plain || topo: group j moves q column j
// TreeKnown<SIG, TOPO, SIG2> reads them from the packed words of the built-in instantiations (up to 16 groups of classes, 10 of bookkeeping);
// a run-time instantiation (jit.cpp, tree_kernels.hip: tree_jit_knowledge) supplies a generated type with the same members for ANY tree -- YuMi's
// 18 groups, the 13 of a Kinova Gen3 -- so the straight-line recursion is not limited to what fits two 64-bit words.
template <SegSig SIG, TreeTopo TOPO, SegSig SIG2 = 0> struct TreeKnown {
    static constexpr bool known = SIG != 0 && !(SIG & kTreeSigAnyConstants);
    static constexpr bool plain = (SIG & kTreeSigPlain) != 0, topo = TOPO != 0, any = plain || topo;
    RTB_HD static constexpr bool revolute(int j) { return plain || (TOPO != 0 && !topo_pris(TOPO, j)); }
    RTB_HD static constexpr int cls(int j) { return tree_cls<SIG, SIG2>(j); }
    RTB_HD static constexpr int tm(int j, bool rev) { return tree_tm<SIG, SIG2>(j, rev); }
    RTB_HD static constexpr int parent(int j) { return topo_parent(TOPO, j); }
    RTB_HD static constexpr bool pris(int j) { return topo_pris(TOPO, j); }
    RTB_HD static constexpr int parent_slot(int j) { return topo_parent_slot(TOPO, j); }
    RTB_HD static constexpr int save_slot(int j) { return topo_save_slot(TOPO, j); }
};
typedef TreeKnown<0, 0, 0> TreeNothing;      // the general kernels
template <class G> RTB_HD V3 inertia_rot(const G &g, V3 w)   // I_bar w
{
    return v3(fmad(g.I[4], w.z, fmad(g.I[3], w.y, g.I[0] * w.x)), fmad(g.I[5], w.z, fmad(g.I[1], w.y, g.I[3] * w.x)),
              fmad(g.I[2], w.z, fmad(g.I[5], w.y, g.I[4] * w.x)));
}

// One sample.  groups: wave-uniform table; qin/qdin/qddin(column) -> double; tau(column, value);
// slot(index) -> double& into this lane's kTreeSlotDoubles * nslots scratch (LDS on the GPU).
// joint angles -> sin/cos up front (branch-free reduction; one wave-wide library fallback)
template <int NG, class KN = TreeNothing, class GroupsP, class InQ>
RTB_HD void tree_trig(GroupsP groups, InQ qin, double (&sn)[NG], double (&cs)[NG])
{
    {
        bool big = false;
        typedef KN K;                             // known: group j on q column j, the joint kinds
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const auto &g = groups[j];
            const double th = K::any ? (K::revolute(j) ? qin(j) * (jm_flip(g.jmeta) ? -1.0 : 1.0) : 0.0)
                                     : (jm_prismatic(g.jmeta) ? 0.0 : qin(jm_jq(g.jmeta)) * (jm_flip(g.jmeta) ? -1.0 : 1.0));
            sn[j] = th;
            big = big || !(fabs(th) < kTrigFastLimit);
        }
        if (wave_any(big)) {
#pragma unroll
            for (int j = 0; j < NG; ++j) { double s, c; sincos(sn[j], &s, &c); sn[j] = s; cs[j] = c; }
        } else {
#pragma unroll
            for (int j = 0; j < NG; ++j) { double s, c; sincos_reduced(sn[j], s, c); sn[j] = s; cs[j] = c; }
        }
        sched_fence();
    }
}

// The two recursions with the sines and cosines supplied (the dynamics terms run several passes at one configuration).
#ifndef RTB_TREE_ACC_ONLY
#define RTB_TREE_ACC_ONLY 1      // 0: the unit-acceleration passes run the full recursion (A/B switch, scripts/build_variant.sh)
#endif
// VEL = false (compile-time): every joint velocity is zero -- the unit-acceleration passes that build the columns of M(q) (Dynamics.inertia,
// the inertia half of Dynamics.accel).  All velocities stay exactly zero through the recursion, so everything that multiplies them is left
// out: the two motion transforms of (v, w), the v x vJ terms, I v and the v x* (I v) forces -- about 90 of the ~210 operations a group's forward
// step costs.  The terms left out are exact zeros in the full recursion (x * 0 + y == y), so the torques are the same numbers.
// first (VEL = false and gravity = 0 only; wave-uniform): the groups before position `first` are at rest AND their torques are not wanted --
// the column pass for joint i of a robot whose groups are numbered in joint order starts at group i: every group before it has zero
// acceleration (it is no descendant of i), so its forward step is skipped (its saved state is written as zeros for the branches that hang
// off it) and so is its backward step (M's entries above the diagonal come from the mirror).  dyn_device.h does the same for DH chains.
#ifndef RTB_TREE_BILINEAR
#define RTB_TREE_BILINEAR 1      // 0: Dynamics.coriolis by the polar form over full passes (two per column), the first implementation
#endif
#ifndef RTB_TREE_SKIP_PREFIX
#define RTB_TREE_SKIP_PREFIX 1
#endif
template <int NG, bool VEL = true, class KN = TreeNothing, class GroupsP, class InQ, class InQd, class InQdd, class Out, class Slot>
RTB_HD void tree_rne_core(GroupsP groups, int nslots, const double (&sn)[NG], const double (&cs)[NG], V3 gravity, InQ qin, InQd qdin, InQdd qddin,
                          Out tau, Slot slot, int first = 0)
{
    V3 Fl[NG], Fa[NG];
    typedef KN K;
    constexpr bool kPlain = K::plain;
    if (!kPlain)
        for (int k = 0; k < nslots; ++k)
            for (int e = 12; e < 18; ++e) slot(k * kTreeSlotDoubles + e) = 0.0;

    // ---- forward recursion (Robot.py:1822-1872)
    V3 vl = v3(0, 0, 0), va = v3(0, 0, 0), al = v3(0, 0, 0), aa = v3(0, 0, 0);   // state of the previous group
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const auto &g = groups[j];
        const int cls = K::cls(j), tm = K::tm(j, K::revolute(j));          // constants of the unrolled copy (SIG = 0: general, 7)
        // (read where they are used: loading the three words up front costs the general kernels some 70 registers)
        auto parent_of = [&]() { return kPlain ? j - 1 : (K::topo ? K::parent(j) : g.parent); };
        auto parent_slot_of = [&]() { return kPlain ? -1 : (K::topo ? K::parent_slot(j) : g.parent_slot); };
        auto save_slot_of = [&]() { return kPlain ? -1 : (K::topo ? K::save_slot(j) : g.save_slot); };
        if (!VEL && j < first) {
            al = v3(0, 0, 0); aa = v3(0, 0, 0);
            if (save_slot_of() >= 0) {
                const int b = save_slot_of() * kTreeSlotDoubles;
                for (int e = 6; e < 12; ++e) slot(b + e) = 0.0;
            }
            Fl[j] = v3(0, 0, 0); Fa[j] = v3(0, 0, 0);
            continue;
        }
        const bool pris = K::topo ? K::pris(j) : (!kPlain && jm_prismatic(g.jmeta) != 0);
        const int col = K::any ? j : jm_jq(g.jmeta);
        const double qdj = VEL ? qdin(col) : 0.0, qddj = qddin(col);
        const double d = pris ? qin(col) * (jm_flip(g.jmeta) ? -1.0 : 1.0) : 0.0;
        V3 pvl, pva, pal, paa;     // parent state
        if (parent_of() < 0) {
            pvl = v3(0, 0, 0); pva = v3(0, 0, 0);
            pal = v3(-gravity.x, -gravity.y, -gravity.z); paa = v3(0, 0, 0);     // a_grav = -SpatialAcceleration(gravity)
        } else if (parent_slot_of() >= 0) {
            const int b = parent_slot_of() * kTreeSlotDoubles;
            if (VEL) { pvl = v3(slot(b + 0), slot(b + 1), slot(b + 2)); pva = v3(slot(b + 3), slot(b + 4), slot(b + 5)); }
            else { pvl = v3(0, 0, 0); pva = v3(0, 0, 0); }
            pal = v3(slot(b + 6), slot(b + 7), slot(b + 8)); paa = v3(slot(b + 9), slot(b + 10), slot(b + 11));
        } else {
            pvl = vl; pva = va; pal = al; paa = aa;
        }
        // frame j in the parent frame: R = R_C Rz(theta), p = t_C (+ R_C z d for a prismatic joint)
        const V3 p = tree_origin(K::revolute(j), g, d);
        const double s = sn[j], c = cs[j];
        // X_up on motion vectors: w' = R^T w ; v' = R^T (v + w x p)
        if (VEL) {
            va = rz_t(s, c, seg_rt_c(cls, g, pva));
            vl = rz_t(s, c, seg_rt_c(cls, g, add_cross_ap(tm, pvl, pva, p)));
        }
        aa = rz_t(s, c, seg_rt_c(cls, g, paa));
        al = rz_t(s, c, seg_rt_c(cls, g, add_cross_ap(tm, pal, paa, p)));
        // joint velocity vJ = s_j qd (the motion subspace ignores `flip`, ET.py:592-608), then
        // a += s_j qdd + v x vJ with the spatial motion cross product (Robot.py:1866-1870)
        if (pris) {
            if (VEL) {
                vl.z += qdj;
                al = crossz_add(va, qdj, al);
            }
            al.z += qddj;
        } else {
            if (VEL) {
                va.z += qdj;
                al = crossz_add(vl, qdj, al);
                aa = crossz_add(va, qdj, aa);
            }
            aa.z += qddj;
        }
        if (save_slot_of() >= 0) {
            const int b = save_slot_of() * kTreeSlotDoubles;
            if (VEL) { slot(b + 0) = vl.x; slot(b + 1) = vl.y; slot(b + 2) = vl.z; slot(b + 3) = va.x; slot(b + 4) = va.y; slot(b + 5) = va.z; }
            slot(b + 6) = al.x; slot(b + 7) = al.y; slot(b + 8) = al.z; slot(b + 9) = aa.x; slot(b + 10) = aa.y; slot(b + 11) = aa.z;
        }
        // f = I a + v x* (I v) with I = [[M 1, -h x], [h x, I_bar]]  (Robot.py:1872)
        const V3 h = v3(g.h[0], g.h[1], g.h[2]);
        const V3 Ial = cross_add(aa, h, g.M * al), Iaa = cross_add(h, al, inertia_rot(g, aa));
        if (VEL) {
            const V3 Ivl = cross_add(va, h, g.M * vl), Iva = cross_add(h, vl, inertia_rot(g, va));
            Fl[j] = cross_add(va, Ivl, Ial);
            Fa[j] = cross_add(vl, Ivl, cross_add(va, Iva, Iaa));
        } else {
            Fl[j] = Ial;
            Fa[j] = Iaa;
        }
        sched_fence();
    }

    // ---- backward recursion (Robot.py:1875-1893)
    V3 cl = v3(0, 0, 0), ca = v3(0, 0, 0);   // force handed down by group j+1 when its parent is group j
#pragma unroll
    for (int jj = 0; jj < NG; ++jj) {
        const int j = NG - 1 - jj;
        if (!VEL && j < first) continue;
        const auto &g = groups[j];
        const int cls = K::cls(j), tm = K::tm(j, K::revolute(j));
        // (read where they are used: loading the three words up front costs the general kernels some 70 registers)
        auto parent_of = [&]() { return kPlain ? j - 1 : (K::topo ? K::parent(j) : g.parent); };
        auto parent_slot_of = [&]() { return kPlain ? -1 : (K::topo ? K::parent_slot(j) : g.parent_slot); };
        auto save_slot_of = [&]() { return kPlain ? -1 : (K::topo ? K::save_slot(j) : g.save_slot); };
        const bool pris = K::topo ? K::pris(j) : (!kPlain && jm_prismatic(g.jmeta) != 0);
        V3 fl = Fl[j] + cl, fa = Fa[j] + ca;
        if (save_slot_of() >= 0) {
            const int b = save_slot_of() * kTreeSlotDoubles + 12;
            fl = fl + v3(slot(b + 0), slot(b + 1), slot(b + 2));
            fa = fa + v3(slot(b + 3), slot(b + 4), slot(b + 5));
        }
        tau(K::any ? j : g.out_col, pris ? fl.z : fa.z);                    // Q[k, j] = sum(f[j] * s[j])
        cl = v3(0, 0, 0); ca = v3(0, 0, 0);
        if (parent_of() >= 0) {
            // f_parent += X_up^T f: lin' = R lin ; ang' = R ang + p x (R lin)
            const double d = pris ? qin(K::any ? j : jm_jq(g.jmeta)) * (jm_flip(g.jmeta) ? -1.0 : 1.0) : 0.0;
            const V3 p = tree_origin(K::revolute(j), g, d);
            const V3 tl = seg_r_c(cls, g, rz(sn[j], cs[j], fl));
            const V3 ta = add_cross_pa(tm, seg_r_c(cls, g, rz(sn[j], cs[j], fa)), p, tl);
            if (parent_slot_of() >= 0) {
                const int b = parent_slot_of() * kTreeSlotDoubles + 12;
                slot(b + 0) += tl.x; slot(b + 1) += tl.y; slot(b + 2) += tl.z;
                slot(b + 3) += ta.x; slot(b + 4) += ta.y; slot(b + 5) += ta.z;
            } else {
                cl = tl; ca = ta;
            }
        }
        sched_fence();
    }
}

// ---- one column of C(q, qd), evaluated directly.  The velocity torque of Robot.rne (gravity 0, qdd = 0) is a quadratic form tau(v) = B(v, v)
// of the joint velocities, and what Dynamics.coriolis assembles from its n + n (n - 1) / 2 unit-velocity passes (robot/Dynamics.py:811-861)
// is column k = B(qd, e_k): sum_i qd_i B(e_i, e_k).  B is evaluated here by carrying BOTH velocity fields through one recursion -- u = qd
// and w = e_k -- and replacing every product of two velocities x(v) y(v) of Robot.py:1866-1872 by x(u) y(w) + x(w) y(u) (= 2 B; the caller
// halves).  Against the two full passes per column of the polar form tau(qd + s e_k) - tau(qd - s e_k) this is one pass of ~1.1x the
// arithmetic, exact for any spread of velocities (no scale s to choose, no cancellation), and the groups before `first` (w = 0 there: their
// accelerations and forces vanish) only advance u.  `first` as in tree_rne_core; slots of kTreeBilinearSlotDoubles.
template <int NG, class KN = TreeNothing, class GroupsP, class InQ, class InQd, class Out, class Slot>
RTB_HD void tree_bilinear_core(GroupsP groups, int nslots, const double (&sn)[NG], const double (&cs)[NG], InQ qin, InQd qdin, int k, Out tau,
                               Slot slot, int first)
{
    constexpr int SD = kTreeBilinearSlotDoubles;
    V3 Fl[NG], Fa[NG];
    typedef KN K;
    constexpr bool kPlain = K::plain;
    if (!kPlain)
        for (int i = 0; i < nslots; ++i)
            for (int e = 18; e < 24; ++e) slot(i * SD + e) = 0.0;
    const V3 o = v3(0, 0, 0);
    V3 ul = o, ua = o, wl = o, wa = o, al = o, aa = o;       // previous group: velocity under u, under w, the bilinear acceleration
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const auto &g = groups[j];
        const int cls = K::cls(j), tm = K::tm(j, K::revolute(j));
        // (read where they are used: loading the three words up front costs the general kernels some 70 registers)
        auto parent_of = [&]() { return kPlain ? j - 1 : (K::topo ? K::parent(j) : g.parent); };
        auto parent_slot_of = [&]() { return kPlain ? -1 : (K::topo ? K::parent_slot(j) : g.parent_slot); };
        auto save_slot_of = [&]() { return kPlain ? -1 : (K::topo ? K::save_slot(j) : g.save_slot); };
        const bool pris = K::topo ? K::pris(j) : (!kPlain && jm_prismatic(g.jmeta) != 0);
        const int col = K::any ? j : jm_jq(g.jmeta);
        const double qdu = qdin(col), qdw = col == k ? 1.0 : 0.0;
        const double d = pris ? qin(col) * (jm_flip(g.jmeta) ? -1.0 : 1.0) : 0.0;
        const bool rest = j < first;                           // wave-uniform: w, a and f are zero up to here
        V3 pul, pua, pwl, pwa, pal, paa;
        if (parent_of() < 0) {
            pul = o; pua = o; pwl = o; pwa = o; pal = o; paa = o;
        } else if (parent_slot_of() >= 0) {
            const int b = parent_slot_of() * SD;
            pul = v3(slot(b + 0), slot(b + 1), slot(b + 2)); pua = v3(slot(b + 3), slot(b + 4), slot(b + 5));
            if (rest) { pwl = o; pwa = o; pal = o; paa = o; }
            else {
                pwl = v3(slot(b + 6), slot(b + 7), slot(b + 8)); pwa = v3(slot(b + 9), slot(b + 10), slot(b + 11));
                pal = v3(slot(b + 12), slot(b + 13), slot(b + 14)); paa = v3(slot(b + 15), slot(b + 16), slot(b + 17));
            }
        } else {
            pul = ul; pua = ua; pwl = wl; pwa = wa; pal = al; paa = aa;
        }
        const V3 p = tree_origin(K::revolute(j), g, d);
        const double s = sn[j], c = cs[j];
        ua = rz_t(s, c, seg_rt_c(cls, g, pua));
        ul = rz_t(s, c, seg_rt_c(cls, g, add_cross_ap(tm, pul, pua, p)));
        if (rest) {
            if (pris) ul.z += qdu; else ua.z += qdu;
            wl = o; wa = o; al = o; aa = o;
            if (save_slot_of() >= 0) {
                const int b = save_slot_of() * SD;
                slot(b + 0) = ul.x; slot(b + 1) = ul.y; slot(b + 2) = ul.z; slot(b + 3) = ua.x; slot(b + 4) = ua.y; slot(b + 5) = ua.z;
                for (int e = 6; e < 18; ++e) slot(b + e) = 0.0;
            }
            Fl[j] = o; Fa[j] = o;
            sched_fence();
            continue;
        }
        wa = rz_t(s, c, seg_rt_c(cls, g, pwa));
        wl = rz_t(s, c, seg_rt_c(cls, g, add_cross_ap(tm, pwl, pwa, p)));
        aa = rz_t(s, c, seg_rt_c(cls, g, paa));
        al = rz_t(s, c, seg_rt_c(cls, g, add_cross_ap(tm, pal, paa, p)));
        // a += v x vJ (Robot.py:1866-1870), both ways round; cross(v, (0, 0, t)) = (v.y t, -v.x t, 0)
        if (pris) {
            ul.z += qdu; wl.z += qdw;
            al = crossz_add(wa, qdu, crossz_add(ua, qdw, al));
        } else {
            ua.z += qdu; wa.z += qdw;
            al = crossz_add(wl, qdu, crossz_add(ul, qdw, al));
            aa = crossz_add(wa, qdu, crossz_add(ua, qdw, aa));
        }
        if (save_slot_of() >= 0) {
            const int b = save_slot_of() * SD;
            slot(b + 0) = ul.x; slot(b + 1) = ul.y; slot(b + 2) = ul.z; slot(b + 3) = ua.x; slot(b + 4) = ua.y; slot(b + 5) = ua.z;
            slot(b + 6) = wl.x; slot(b + 7) = wl.y; slot(b + 8) = wl.z; slot(b + 9) = wa.x; slot(b + 10) = wa.y; slot(b + 11) = wa.z;
            slot(b + 12) = al.x; slot(b + 13) = al.y; slot(b + 14) = al.z; slot(b + 15) = aa.x; slot(b + 16) = aa.y; slot(b + 17) = aa.z;
        }
        // f = I a + v x* (I v) (Robot.py:1872) -> I a + u x* (I w) + w x* (I u);  v x* (fl, fa) = (va x fl, va x fa + vl x fl)
        const V3 h = v3(g.h[0], g.h[1], g.h[2]);
        const V3 Iul = cross_add(ua, h, g.M * ul), Iua = cross_add(h, ul, inertia_rot(g, ua));
        const V3 Iwl = cross_add(wa, h, g.M * wl), Iwa = cross_add(h, wl, inertia_rot(g, wa));
        const V3 Ial = cross_add(aa, h, g.M * al), Iaa = cross_add(h, al, inertia_rot(g, aa));
        Fl[j] = cross_add(wa, Iul, cross_add(ua, Iwl, Ial));
        Fa[j] = cross_add(wl, Iul, cross_add(ul, Iwl, cross_add(wa, Iua, cross_add(ua, Iwa, Iaa))));
        sched_fence();
    }
    // ---- backward recursion: tree_rne_core's, on the wider slots
    V3 cl = o, ca = o;
#pragma unroll
    for (int jj = 0; jj < NG; ++jj) {
        const int j = NG - 1 - jj;
        const auto &g = groups[j];
        const int cls = K::cls(j), tm = K::tm(j, K::revolute(j));
        // (read where they are used: loading the three words up front costs the general kernels some 70 registers)
        auto parent_of = [&]() { return kPlain ? j - 1 : (K::topo ? K::parent(j) : g.parent); };
        auto parent_slot_of = [&]() { return kPlain ? -1 : (K::topo ? K::parent_slot(j) : g.parent_slot); };
        auto save_slot_of = [&]() { return kPlain ? -1 : (K::topo ? K::save_slot(j) : g.save_slot); };
        const bool pris = K::topo ? K::pris(j) : (!kPlain && jm_prismatic(g.jmeta) != 0);
        V3 fl = Fl[j] + cl, fa = Fa[j] + ca;
        if (save_slot_of() >= 0) {
            const int b = save_slot_of() * SD + 18;
            fl = fl + v3(slot(b + 0), slot(b + 1), slot(b + 2));
            fa = fa + v3(slot(b + 3), slot(b + 4), slot(b + 5));
        }
        tau(K::any ? j : g.out_col, pris ? fl.z : fa.z);
        cl = o; ca = o;
        if (parent_of() >= 0) {
            const double d = pris ? qin(K::any ? j : jm_jq(g.jmeta)) * (jm_flip(g.jmeta) ? -1.0 : 1.0) : 0.0;
            const V3 p = tree_origin(K::revolute(j), g, d);
            const V3 tl = seg_r_c(cls, g, rz(sn[j], cs[j], fl));
            const V3 ta = add_cross_pa(tm, seg_r_c(cls, g, rz(sn[j], cs[j], fa)), p, tl);
            if (parent_slot_of() >= 0) {
                const int b = parent_slot_of() * SD + 18;
                slot(b + 0) += tl.x; slot(b + 1) += tl.y; slot(b + 2) += tl.z;
                slot(b + 3) += ta.x; slot(b + 4) += ta.y; slot(b + 5) += ta.z;
            } else {
                cl = tl; ca = ta;
            }
        }
        sched_fence();
    }
}

// ATREST: the caller has no joint velocities (rtbhip_tree_rne with qd = NULL: Dynamics.gravload, Dynamics.itorque) -- the recursion
// without its velocity half (tree_rne_core VEL = false), gravity still the base's acceleration.
template <int NG, bool ATREST = false, class KN = TreeNothing, class GroupsP, class InQ, class InQd, class InQdd, class Out, class Slot>
RTB_HD void tree_rne_lane(GroupsP groups, int nslots, V3 gravity, InQ qin, InQd qdin, InQdd qddin, Out tau, Slot slot)
{
    double sn[NG], cs[NG];
    tree_trig<NG, KN>(groups, qin, sn, cs);
    tree_rne_core<NG, !ATREST, KN>(groups, nslots, sn, cs, gravity, qin, qdin, qddin, tau, slot);
}

// ---- the Dynamics-mixin terms of an ETS robot (robot/Dynamics.py:704-861, 424-509 on Robot.rne): every Newton-Euler pass the
// reference makes for ONE configuration, in one lane, as dyn_device.h does for DH chains.
//   mine : this lane's inputs [q (n) | qd (n) | torque (n)] (what the mode needs)      mA : n x n tile, row-major
//   inertia   pass i = rne(q, 0, e_c, gravity 0) for the column c that group i moves (:752-758); kept: its entries j >= i, in the packed lower triangle
//             mA[j (j + 1) / 2 + i] -- the kernel's flush mirrors them (M is symmetric; the mirrored half differs from the reference's
//             separately rounded entries by rounding only), 21 instead of 36 doubles of LDS per lane for n = 6
//   coriolis  mA = C(q, qd): column k = B(qd, e_k) from one two-field pass (tree_bilinear_core); RTB_TREE_BILINEAR = 0: the polar form /
//             the reference's own scheme of rounds 1-3
//   accel     mA[0..n) = qdd = M^-1 (torque - rne(q, qd, 0))          (:492-505; M's lower triangle, LDL^T)
template <int NG>
RTB_HD void tree_opaque(double (&sn)[NG], double (&cs)[NG])
{
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int j = 0; j < NG; ++j) asm volatile("" : "+v"(sn[j]), "+v"(cs[j]));     // keeps each pass self-contained (dyn_device.h: dyn_opaque)
#endif
}

// a[j] = v / a[j] += v for a wave-uniform run-time j (the tau column of a group) without indexing the register array
template <int NG>
RTB_HD void tree_put(double (&a)[NG], int j, double v)
{
#pragma unroll
    for (int k = 0; k < NG; ++k) a[k] = (j == k) ? v : a[k];
}
template <int NG>
RTB_HD void tree_add(double (&a)[NG], int j, double v)
{
#pragma unroll
    for (int k = 0; k < NG; ++k) a[k] += (j == k) ? v : 0.0;
}

// group j moves q column j for every j: the robot is numbered in group order (wave-uniform)
template <int NG, class GroupsP>
RTB_HD bool tree_in_group_order(GroupsP groups)
{
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NG; ++j) ok = ok && jm_jq(groups[j].jmeta) == j;
    return ok;
}
// the position of the group that moves q column r (the row of Mp that is row r of the reference's inertia matrix)
template <int NG, class GroupsP>
RTB_HD int tree_row_position(GroupsP groups, int r)
{
    int a = 0;
#pragma unroll
    for (int j = 0; j < NG; ++j) a = jm_jq(groups[j].jmeta) == r ? j : a;
    return a;
}

template <int NG, int MODE, class KN = TreeNothing, class GroupsP, class Slot>
RTB_HD void tree_dyn_lane(GroupsP groups, int nslots, const double *mine, double *mA, V3 grav, Slot slot)
{
    const V3 zero = v3(0, 0, 0);
    auto qin = [&](int j) { return mine[j]; };
    auto none = [&](int) { return 0.0; };
    double sn[NG], cs[NG];
    tree_trig<NG, KN>(groups, qin, sn, cs);
    constexpr bool kPlain = KN::any;          // group j moves q column j: nothing to permute
    // The unit-acceleration passes run in GROUP order: pass i accelerates the joint of the group at position i (q column jq_i), so that
    // Mp[j][i] = torque of group j is the symmetric joint-space inertia in group order -- only the entries j >= i are computed (the groups before i are
    // no descendants of i: at rest), packed lower triangle.  The reference's matrix is M[c, :] = rne(q, 0, e_c) with c a q COLUMN and the torques
    // in group order (Robot.py:1875-1893): row jq_i of the reference's M is row i of Mp.  For a robot numbered in group order (every URDF robot,
    // every robot whose jindex was assigned automatically) the two coincide; otherwise the inertia kernel's flush permutes the rows and accel
    // permutes its right-hand side (below) -- tree_row_position().
    constexpr bool skip = RTB_TREE_SKIP_PREFIX && RTB_TREE_ACC_ONLY;
    if (MODE == kDynInertia) {
#pragma unroll 1
        for (int i = 0; i < NG; ++i) {
            tree_opaque<NG>(sn, cs);
            const int ci = kPlain ? i : jm_jq(groups[i].jmeta);
            tree_rne_core<NG, !RTB_TREE_ACC_ONLY, KN>(groups, nslots, sn, cs, zero, qin, none, [&](int c) { return c == ci ? 1.0 : 0.0; },
                                     [&](int j, double v) { if (j >= i) mA[j * (j + 1) / 2 + i] = v; }, slot, skip ? i : 0);
        }
    }
    if (MODE == kDynAccel) {
        double b[NG];
        tree_opaque<NG>(sn, cs);
        tree_rne_core<NG, true, KN>(groups, nslots, sn, cs, grav, qin, [&](int j) { return mine[NG + j]; }, none,
                          [&](int j, double v) { tree_put<NG>(b, j, mine[2 * NG + j] - v); }, slot);
        // the reference solves M qdd = torque - tau_0 with ITS M (rows by q column): row i of Mp stands in row jq_i, so the right-hand side of
        // the group-ordered system is entry jq_i of (torque - tau_0)
        if (!kPlain && !tree_in_group_order<NG>(groups)) {         // wave-uniform
            double b2[NG];
#pragma unroll
            for (int a = 0; a < NG; ++a) b2[a] = dyn_pick<NG>(b, jm_jq(groups[a].jmeta));
#pragma unroll
            for (int a = 0; a < NG; ++a) b[a] = b2[a];
        }
#pragma unroll 1
        for (int i = 0; i < NG; ++i) {
            tree_opaque<NG>(sn, cs);
            const int ci = kPlain ? i : jm_jq(groups[i].jmeta);
            tree_rne_core<NG, !RTB_TREE_ACC_ONLY, KN>(groups, nslots, sn, cs, zero, qin, none, [&](int c) { return c == ci ? 1.0 : 0.0; },
                                     [&](int j, double v) { if (j >= i) mA[j * (j + 1) / 2 + i] = v; }, slot, skip ? i : 0);
        }
        double x[NG], M[NG][NG];
#pragma unroll
        for (int r = 0; r < NG; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) M[r][c] = mA[r * (r + 1) / 2 + c];
        ldl_solve<NG>(M, b, x);
#pragma unroll
        for (int j = 0; j < NG; ++j) mA[j] = x[j];
    }
    if (MODE == kDynCoriolis) {
#if RTB_TREE_BILINEAR
        // column k = B(qd, e_k), one two-field pass each (tree_bilinear_core)
        bool ordered = true;
#pragma unroll
        for (int j = 0; j < NG; ++j) ordered = ordered && (kPlain || jm_jq(groups[j].jmeta) == j);
#pragma unroll 1
        for (int k = 0; k < NG; ++k) {
            tree_opaque<NG>(sn, cs);
            tree_bilinear_core<NG, KN>(groups, nslots, sn, cs, qin, [&](int j) { return mine[NG + j]; }, k,
                                   [&](int r, double v) { mA[r * NG + k] = 0.5 * v; }, slot, ordered ? k : 0);
        }
#else
        // dyn_device.h, the same two schemes and the same per-row choice between them
        double qdv[NG], vmax = 0.0;
#pragma unroll
        for (int j = 0; j < NG; ++j) { qdv[j] = mine[NG + j]; vmax = fmax(vmax, fabs(qdv[j])); }
        int ex = 0;
        const double mant = frexp(vmax, &ex);
        if (mant == 0.5) ex -= 1;
        if (!(vmax > 0.0) || !(vmax < 1.7e308)) ex = 0;
        ex = ex > 400 ? 400 : (ex < -400 ? -400 : ex);
        const double sc = ldexp(1.0, ex);
        const double inv4s = vmax > 0.0 || vmax != vmax ? 0.25 / sc : 0.0;
        bool wide = false;
#pragma unroll
        for (int j = 0; j < NG; ++j) wide = wide || (qdv[j] != 0.0 && fabs(qdv[j]) * 65536.0 < vmax);
        auto polar = [&]() {
#pragma unroll 1
            for (int k = 0; k < NG; ++k) {
                tree_opaque<NG>(sn, cs);
                tree_rne_core<NG>(groups, nslots, sn, cs, zero, qin, [&](int j) { return j == k ? qdv[j] + sc : qdv[j]; }, none,
                                  [&](int r, double v) { mA[r * NG + k] = v; }, slot);
                tree_opaque<NG>(sn, cs);
                tree_rne_core<NG>(groups, nslots, sn, cs, zero, qin, [&](int j) { return j == k ? qdv[j] - sc : qdv[j]; }, none,
                                  [&](int r, double v) { mA[r * NG + k] = (mA[r * NG + k] - v) * inv4s; }, slot);
            }
        };
        auto reference_scheme = [&]() {
            double S = 0.0, U[NG];
#pragma unroll
            for (int j = 0; j < NG; ++j) { S += qdv[j]; U[j] = 0.0; }
#pragma unroll 1
            for (int i = 0; i < NG; ++i) {
                const double qdi = dyn_pick<NG>(qdv, i), wi = 2.0 * qdi - 0.5 * S;
                tree_opaque<NG>(sn, cs);
                tree_rne_core<NG>(groups, nslots, sn, cs, zero, qin, [&](int j) { return j == i ? 1.0 : 0.0; }, none,
                                  [&](int r, double v) { mA[r * NG + i] = v * wi; tree_add<NG>(U, r, v * qdi); }, slot);
            }
#pragma unroll
            for (int r = 0; r < NG; ++r)
#pragma unroll
                for (int c = 0; c < NG; ++c) mA[r * NG + c] -= 0.5 * U[r];
#pragma unroll 1
            for (int i = 0; i < NG; ++i) {
#pragma unroll 1
                for (int j = i + 1; j < NG; ++j) {
                    const double hi = 0.5 * dyn_pick<NG>(qdv, i), hj = 0.5 * dyn_pick<NG>(qdv, j);
                    tree_opaque<NG>(sn, cs);
                    tree_rne_core<NG>(groups, nslots, sn, cs, zero, qin, [&](int k) { return (k == i || k == j) ? 1.0 : 0.0; }, none,
                                      [&](int r, double tq) {
                                          mA[r * NG + j] += tq * hi;
                                          mA[r * NG + i] += tq * hj;
                                      }, slot);
                }
            }
        };
        polar();                                   // every lane; wide rows redo their tile below (dyn_device.h)
        if (wave_any(wide)) {
            if (wide) reference_scheme();
        }
#endif
    }
}

#pragma clang fp contract(fast)
}  // namespace rtbhip
