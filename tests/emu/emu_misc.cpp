// tests/emu/emu_misc.cpp -- TEST INFRASTRUCTURE: differential-kinematics consumers, link frames, partial_fkine0, tree RNE replayed on the CPU.
#include "emu_common.h"

template <int NJ>
static void diff_run(const KinParams &kp, const DevChain &cv, int mode, int axes, const double *q, const double *qd, int64_t N, double *out)
{
    for (int64_t s = 0; s < N; ++s) {
        Pose P;
        double jac[6 * NJ];
        reg_compute<NJ, true>(kp, cv, q, s, P, jac);
        if (mode == 0) {
            double v[NJ], jd[6 * NJ];
            for (int j = 0; j < NJ; ++j) v[j] = qd[s * kp.qw + jm_jq(cv.jmeta[j])];
            jacob_dot<NJ>(jac, v, jd);
            for (int k = 0; k < 6 * NJ; ++k) out[s * 6 * NJ + k] = jd[k];
        } else if (mode == 3) {
            double ja[6 * NJ];
            jacob_analytical<NJ>(P, jac, axes, ja);
            for (int k = 0; k < 6 * NJ; ++k) out[s * 6 * NJ + k] = ja[k];
        } else if (mode == 4) {
            double qv[NJ], v[NJ], jd[6 * NJ];
            for (int j = 0; j < NJ; ++j) { qv[j] = q[s * kp.qw + jm_jq(cv.jmeta[j])]; v[j] = qd[s * kp.qw + jm_jq(cv.jmeta[j])]; }
            jacob_analytical_dot<NJ>(cv, kp.tail, qv, v, axes, jd);
            for (int k = 0; k < 6 * NJ; ++k) out[s * 6 * NJ + k] = jd[k];
        } else if (mode == 1) {
            const int method = (axes >> 8) & 3;
            out[s] = method == 0 ? manipulability_yoshikawa<NJ>(jac, axes & 63) : manipulability_singular<NJ>(jac, axes & 63, method);
        } else {
            double jm[NJ];
            jacobm<NJ>(jac, axes, jm);
            for (int k = 0; k < NJ; ++k) out[s * NJ + k] = jm[k];
        }
    }
}
extern "C" int emu_diff(rtbhip_chain_t h, int mode, int axes, const double *q, const double *qd, int64_t N,
                        const double *tool16, int frame, double *out)
{
    const std::shared_ptr<Chain> c_owner = chain_from_handle(h);
    Chain *c = c_owner.get();
    if (!c || c->n < 1 || c->n > 16) return -1;
    KinParams kp;
    kp.n = c->n; kp.qw = c->q_width; kp.stride = kin_stride(c->n); kp.frame = frame; kp.N = N; kp.pad = 0; kp.has_base = 0;
    Affine t = aff16(tool16);
    chain_tail(c, t, kp.tail);
    const DevChain cv = chain_host_view(c);
    switch (c->n) {
    case 1: diff_run<1>(kp, cv, mode, axes, q, qd, N, out); break;
    case 2: diff_run<2>(kp, cv, mode, axes, q, qd, N, out); break;
    case 3: diff_run<3>(kp, cv, mode, axes, q, qd, N, out); break;
    case 4: diff_run<4>(kp, cv, mode, axes, q, qd, N, out); break;
    case 5: diff_run<5>(kp, cv, mode, axes, q, qd, N, out); break;
    case 6: diff_run<6>(kp, cv, mode, axes, q, qd, N, out); break;
    case 7: diff_run<7>(kp, cv, mode, axes, q, qd, N, out); break;
    case 8: diff_run<8>(kp, cv, mode, axes, q, qd, N, out); break;
    case 9: diff_run<9>(kp, cv, mode, axes, q, qd, N, out); break;
    case 10: diff_run<10>(kp, cv, mode, axes, q, qd, N, out); break;
    case 11: diff_run<11>(kp, cv, mode, axes, q, qd, N, out); break;
    case 12: diff_run<12>(kp, cv, mode, axes, q, qd, N, out); break;
    case 13: diff_run<13>(kp, cv, mode, axes, q, qd, N, out); break;
    case 14: diff_run<14>(kp, cv, mode, axes, q, qd, N, out); break;
    case 15: diff_run<15>(kp, cv, mode, axes, q, qd, N, out); break;
    default: diff_run<16>(kp, cv, mode, axes, q, qd, N, out); break;
    }
    return 0;
}

// fkine_all: compile_frames (chain.cpp) + frames_walk (frames_device.h) on the CPU
extern "C" int emu_link_frames(rtbhip_chain_t h, const double *q, int64_t N, const double *base16, const int32_t *marks, int nmarks, double *out)
{
    const std::shared_ptr<Chain> c_owner = chain_from_handle(h);
    Chain *c = c_owner.get();
    if (!c) return -1;
    FrameTable ft;
    if (compile_frames(c, marks, nmarks, &ft) != RTBHIP_OK) return -2;
    Affine b = aff16(base16);
    ft.has_base = b.used;
    for (int i = 0; i < 12; i++) ft.base[i] = b.v[i];
    const DevChain cv = chain_host_view(c);
    for (int64_t s = 0; s < N; ++s) {
        const double *row = q + s * c->q_width;
        double *dst = out + s * (int64_t)nmarks * 16;
        frames_walk(cv, c->n, ft, [&](int k) { return row[k]; }, [&](int m, const Pose &P) {
            pose_store16(P, [&](int k, double v) { dst[m * 16 + k] = v; });
        });
    }
    return 0;
}

// ETS.partial_fkine0: the host plan + the per-column device function of partial_device.h, every column of
// every order replayed on the CPU from the emulated Jacobian and Hessian
extern "C" int emu_partial(rtbhip_chain_t h, const double *q, int64_t N, const double *tool16, int order, double *out)
{
    const std::shared_ptr<Chain> c_owner = chain_from_handle(h);
    Chain *c = c_owner.get();
    if (!c || order < 3 || order > kPartialMaxOrder) return -1;
    const int n = c->n;
    std::vector<std::vector<double>> lower(order - 1);
    for (int a = 1; a < order; ++a) lower[a - 1].assign((size_t)N * partial_size(n, a), 0.0);
    if (emu_kin(h, q, N, nullptr, tool16, 0, nullptr, lower[0].data(), lower[1].data(), 1) != 0) return -1;
    auto src = [&](int o, int64_t cfg, int off) -> double { return lower[o - 1][(size_t)cfg * partial_size(n, o) + off]; };
    for (int a = 3; a <= order; ++a) {
        PartialPlan plan;
        partial_plan(n, a, &plan);
        plan.N = N;
        double *dst = a == order ? out : lower[a - 1].data();
        for (int64_t cfg = 0; cfg < N; ++cfg)
            for (uint32_t col = 0; col < (uint32_t)plan.cols; ++col) {
                double *o = dst + cfg * partial_size(n, a) + (col / n) * 6 * n + col % n;      // (.., 6, n) block col / n, column col % n
                auto put = [&](int r, double v) { o[r * n] = v; };
                if (a == 3 && !(getenv("EMU_PARTIAL3") && atoi(getenv("EMU_PARTIAL3")) == 0)) {
                    // order 3 as k_partial3 runs it: the written-out column on this configuration's Jacobian and Hessian
                    const double *Jc = lower[0].data() + (size_t)cfg * 6 * n, *Hc = lower[1].data() + (size_t)cfg * 6 * n * n;
                    const uint32_t d0 = col % n, d1 = (col / n) % n, d2 = col / (n * n);
                    partial3_column(n, [&](uint32_t off) -> double { return Jc[off]; }, [&](uint32_t off) -> double { return Hc[off]; }, d0, d1, d2, put);
                    continue;
                }
                switch (a) {
                case 3: partial_column<3>(plan, src, cfg, col, put); break;
                case 4: partial_column<4>(plan, src, cfg, col, put); break;
                case 5: partial_column<5>(plan, src, cfg, col, put); break;
                default: partial_column<6>(plan, src, cfg, col, put); break;
                }
            }
    }
    return 0;
}

// ETS-robot inverse dynamics: tree.cpp's compiled table + tree_device.h's per-lane recursion on the CPU
namespace rtbhip { int tree_sig_enabled(); }      // tree_kernels.hip: rtbhip_tune("tree_sig")
template <int NG, SegSig SIG = 0, TreeTopo TOPO = 0, SegSig SIG2 = 0>
static void tree_run(const Tree *t, const double *q, const double *qd, const double *qdd, int64_t N, V3 g, double *tau)
{
    std::vector<double> slots((size_t)kTreeSlotDoubles * std::max(1, t->nslots));
    for (int64_t s = 0; s < N; ++s) {
        const double *a = q + s * NG, *b = qd ? qd + s * NG : nullptr, *c = qdd + s * NG;
        double *o = tau + s * NG;
        // qd == NULL on a robot of up to 12 groups: the at-rest instantiation, as launch_tree_rne dispatches (tree_kernels.hip kTreeAtRestMax)
        if (!b && NG <= 12)
            tree_rne_lane<NG, true, TreeKnown<SIG, TOPO, SIG2>>(t->groups.data(), t->nslots, g, [&](int k) { return a[k]; }, [&](int) { return 0.0; },
                                    [&](int k) { return c[k]; }, [&](int k, double v) { o[k] = v; },
                                    [&](int i) -> double & { return slots[i]; });
        else
            tree_rne_lane<NG, false, TreeKnown<SIG, TOPO, SIG2>>(t->groups.data(), t->nslots, g, [&](int k) { return a[k]; }, [&](int k) { return b ? b[k] : 0.0; },
                              [&](int k) { return c[k]; }, [&](int k, double v) { o[k] = v; },
                              [&](int i) -> double & { return slots[i]; });
    }
}
#include "emu_tree.h"
// the structure signature tree.cpp computes for a robot (tree_device.h), for the test that pins the UR family's
extern "C" unsigned long long emu_tree_signature(const rtbhip_tree_group *groups, int ng)
{
    Tree t;
    if (compile_tree(groups, ng, &t) != RTBHIP_OK) return 0ull;
    return t.sig;
}
// the tree's bookkeeping word (tree_device.h: TreeTopo) as two 64-bit halves
extern "C" int emu_tree_topology(const rtbhip_tree_group *groups, int ng, unsigned long long *hi, unsigned long long *lo)
{
    Tree t;
    if (compile_tree(groups, ng, &t) != RTBHIP_OK) return -1;
    *hi = (unsigned long long)(t.topo >> 64); *lo = (unsigned long long)t.topo;
    return 0;
}
extern "C" void emu_tree_topology_ibx8(unsigned long long *hi, unsigned long long *lo) { *hi = (unsigned long long)(kTreeTopoIbx8 >> 64); *lo = (unsigned long long)kTreeTopoIbx8; }
extern "C" unsigned long long emu_tree_signature2(const rtbhip_tree_group *groups, int ng)      // the second class word (groups 8 .. 15)
{
    Tree t;
    if (compile_tree(groups, ng, &t) != RTBHIP_OK) return 0ull;
    return t.sig2;
}
extern "C" unsigned long long emu_tree_signature_ur() { return kTreeSigUR; }
extern "C" unsigned long long emu_tree_signature_ibx8() { return kTreeSigIbx8; }
extern "C" int emu_tree_dyn(const rtbhip_tree_group *groups, int ng, int mode, const double *q, const double *qd, const double *tq,
                            int64_t N, const double *grav3, double *out)
{
    Tree t;
    if (compile_tree(groups, ng, &t) != RTBHIP_OK) return -1;
    V3 g = grav3 ? v3(grav3[0], grav3[1], grav3[2]) : v3(0, 0, 0);
    if (rtbhip::tree_sig_enabled() && t.sig == kTreeSigUR) { tree_dyn_mode<6, kTreeSigUR>(mode, &t, q, qd, tq, N, g, out); return 0; }      // as launch_tree_dyn dispatches
    if (rtbhip::tree_sig_enabled() && t.sig == kTreeSigIbx8 && t.topo == kTreeTopoIbx8) { tree_dyn_mode<8, kTreeSigIbx8, kTreeTopoIbx8>(mode, &t, q, qd, tq, N, g, out); return 0; }
    if (rtbhip::tree_sig_enabled() && t.sig == kTreeSigPx100 && t.topo == kTreeTopoPx100) { tree_dyn_mode<7, kTreeSigPx100, kTreeTopoPx100>(mode, &t, q, qd, tq, N, g, out); return 0; }
    if (rtbhip::tree_sig_enabled() && t.sig == kTreeSigIbx9 && t.sig2 == kTreeSig2Ibx9 && t.topo == kTreeTopoIbx9) { tree_dyn_mode<9, kTreeSigIbx9, kTreeTopoIbx9, kTreeSig2Ibx9>(mode, &t, q, qd, tq, N, g, out); return 0; }
    if (rtbhip::tree_sig_enabled() && t.sig == kTreeSigFetch && t.sig2 == kTreeSig2Fetch && t.topo == kTreeTopoFetch) { tree_dyn_mode<10, kTreeSigFetch, kTreeTopoFetch, kTreeSig2Fetch>(mode, &t, q, qd, tq, N, g, out); return 0; }
    if (rtbhip::tree_sig_enabled() && t.sig == kTreeSigMico && t.sig2 == kTreeSig2Mico && t.topo == kTreeTopoMico) { tree_dyn_mode<10, kTreeSigMico, kTreeTopoMico, kTreeSig2Mico>(mode, &t, q, qd, tq, N, g, out); return 0; }
    if (rtbhip::tree_sig_enabled() && (t.sig & kTreeSigPlain)) {          // any other serial chain of up to 8 revolute joints: the plain-chain instantiation
        switch (t.n) {
        case 1: tree_dyn_mode<1, kTreeSigPlainChain>(mode, &t, q, qd, tq, N, g, out); return 0;
        case 2: tree_dyn_mode<2, kTreeSigPlainChain>(mode, &t, q, qd, tq, N, g, out); return 0;
        case 3: tree_dyn_mode<3, kTreeSigPlainChain>(mode, &t, q, qd, tq, N, g, out); return 0;
        case 4: tree_dyn_mode<4, kTreeSigPlainChain>(mode, &t, q, qd, tq, N, g, out); return 0;
        case 5: tree_dyn_mode<5, kTreeSigPlainChain>(mode, &t, q, qd, tq, N, g, out); return 0;
        case 6: tree_dyn_mode<6, kTreeSigPlainChain>(mode, &t, q, qd, tq, N, g, out); return 0;
        case 7: tree_dyn_mode<7, kTreeSigPlainChain>(mode, &t, q, qd, tq, N, g, out); return 0;
        case 8: tree_dyn_mode<8, kTreeSigPlainChain>(mode, &t, q, qd, tq, N, g, out); return 0;
        default: break;
        }
    }
    switch (t.n) {
    case 1: tree_dyn_mode<1>(mode, &t, q, qd, tq, N, g, out); break;
    case 2: tree_dyn_mode<2>(mode, &t, q, qd, tq, N, g, out); break;
    case 3: tree_dyn_mode<3>(mode, &t, q, qd, tq, N, g, out); break;
    case 4: tree_dyn_mode<4>(mode, &t, q, qd, tq, N, g, out); break;
    case 5: tree_dyn_mode<5>(mode, &t, q, qd, tq, N, g, out); break;
    case 6: tree_dyn_mode<6>(mode, &t, q, qd, tq, N, g, out); break;
    case 7: tree_dyn_mode<7>(mode, &t, q, qd, tq, N, g, out); break;
    case 8: tree_dyn_mode<8>(mode, &t, q, qd, tq, N, g, out); break;
    case 9: tree_dyn_mode<9>(mode, &t, q, qd, tq, N, g, out); break;
    case 10: tree_dyn_mode<10>(mode, &t, q, qd, tq, N, g, out); break;
    case 11: tree_dyn_mode<11>(mode, &t, q, qd, tq, N, g, out); break;
    case 12: tree_dyn_mode<12>(mode, &t, q, qd, tq, N, g, out); break;
    default: return emu_tree_dyn_big(&t, mode, q, qd, tq, N, g, out);
    }
    return 0;
}

extern "C" int emu_tree_rne(const rtbhip_tree_group *groups, int ng, const double *q, const double *qd, const double *qdd,
                            int64_t N, const double *grav3, double *tau)
{
    Tree t;
    if (compile_tree(groups, ng, &t) != RTBHIP_OK) return -1;
    V3 g = v3(grav3[0], grav3[1], grav3[2]);
    if (rtbhip::tree_sig_enabled() && t.sig == kTreeSigUR) { tree_run<6, kTreeSigUR>(&t, q, qd, qdd, N, g, tau); return 0; }      // as launch_tree_rne dispatches
    if (rtbhip::tree_sig_enabled() && t.sig == kTreeSigIbx8 && t.topo == kTreeTopoIbx8) { tree_run<8, kTreeSigIbx8, kTreeTopoIbx8>(&t, q, qd, qdd, N, g, tau); return 0; }
    if (rtbhip::tree_sig_enabled() && t.sig == kTreeSigPx100 && t.topo == kTreeTopoPx100) { tree_run<7, kTreeSigPx100, kTreeTopoPx100>(&t, q, qd, qdd, N, g, tau); return 0; }
    if (rtbhip::tree_sig_enabled() && t.sig == kTreeSigIbx9 && t.sig2 == kTreeSig2Ibx9 && t.topo == kTreeTopoIbx9) { tree_run<9, kTreeSigIbx9, kTreeTopoIbx9, kTreeSig2Ibx9>(&t, q, qd, qdd, N, g, tau); return 0; }
    if (rtbhip::tree_sig_enabled() && t.sig == kTreeSigFetch && t.sig2 == kTreeSig2Fetch && t.topo == kTreeTopoFetch) { tree_run<10, kTreeSigFetch, kTreeTopoFetch, kTreeSig2Fetch>(&t, q, qd, qdd, N, g, tau); return 0; }
    if (rtbhip::tree_sig_enabled() && t.sig == kTreeSigMico && t.sig2 == kTreeSig2Mico && t.topo == kTreeTopoMico) { tree_run<10, kTreeSigMico, kTreeTopoMico, kTreeSig2Mico>(&t, q, qd, qdd, N, g, tau); return 0; }
    if (rtbhip::tree_sig_enabled() && (t.sig & kTreeSigPlain)) {
        switch (t.n) {
        case 1: tree_run<1, kTreeSigPlainChain>(&t, q, qd, qdd, N, g, tau); return 0;
        case 2: tree_run<2, kTreeSigPlainChain>(&t, q, qd, qdd, N, g, tau); return 0;
        case 3: tree_run<3, kTreeSigPlainChain>(&t, q, qd, qdd, N, g, tau); return 0;
        case 4: tree_run<4, kTreeSigPlainChain>(&t, q, qd, qdd, N, g, tau); return 0;
        case 5: tree_run<5, kTreeSigPlainChain>(&t, q, qd, qdd, N, g, tau); return 0;
        case 6: tree_run<6, kTreeSigPlainChain>(&t, q, qd, qdd, N, g, tau); return 0;
        case 7: tree_run<7, kTreeSigPlainChain>(&t, q, qd, qdd, N, g, tau); return 0;
        case 8: tree_run<8, kTreeSigPlainChain>(&t, q, qd, qdd, N, g, tau); return 0;
        default: break;
        }
    }
    switch (ng) {
    case 1: tree_run<1>(&t, q, qd, qdd, N, g, tau); break;
    case 2: tree_run<2>(&t, q, qd, qdd, N, g, tau); break;
    case 3: tree_run<3>(&t, q, qd, qdd, N, g, tau); break;
    case 4: tree_run<4>(&t, q, qd, qdd, N, g, tau); break;
    case 5: tree_run<5>(&t, q, qd, qdd, N, g, tau); break;
    case 6: tree_run<6>(&t, q, qd, qdd, N, g, tau); break;
    case 7: tree_run<7>(&t, q, qd, qdd, N, g, tau); break;
    case 8: tree_run<8>(&t, q, qd, qdd, N, g, tau); break;
    case 9: tree_run<9>(&t, q, qd, qdd, N, g, tau); break;
    case 10: tree_run<10>(&t, q, qd, qdd, N, g, tau); break;
    case 11: tree_run<11>(&t, q, qd, qdd, N, g, tau); break;
    case 12: tree_run<12>(&t, q, qd, qdd, N, g, tau); break;
    case 13: tree_run<13>(&t, q, qd, qdd, N, g, tau); break;
    case 14: tree_run<14>(&t, q, qd, qdd, N, g, tau); break;
    case 15: tree_run<15>(&t, q, qd, qdd, N, g, tau); break;
    case 16: tree_run<16>(&t, q, qd, qdd, N, g, tau); break;
    case 17: tree_run<17>(&t, q, qd, qdd, N, g, tau); break;
    case 18: tree_run<18>(&t, q, qd, qdd, N, g, tau); break;
    case 19: tree_run<19>(&t, q, qd, qdd, N, g, tau); break;
    case 20: tree_run<20>(&t, q, qd, qdd, N, g, tau); break;
    case 21: tree_run<21>(&t, q, qd, qdd, N, g, tau); break;
    case 22: tree_run<22>(&t, q, qd, qdd, N, g, tau); break;
    case 23: tree_run<23>(&t, q, qd, qdd, N, g, tau); break;
    case 24: tree_run<24>(&t, q, qd, qdd, N, g, tau); break;
    default: return -2;
    }
    return 0;
}

// pose_mul_seg_cls (kin_device.h) against the general product on segment j of a chain: P_in (12: R row-major, t) -> out_cls, out_gen; returns the
// descriptor's class bits (class | tmask << 4), -1 for a bad handle / index
extern "C" int emu_pose_mul_seg(rtbhip_chain_t h, int j, const double *Pin, double *out_cls, double *out_gen)
{
    const std::shared_ptr<Chain> c_owner = chain_from_handle(h);
    Chain *c = c_owner.get();
    if (!c || j < 0 || j > c->n) return -1;
    const EmuChainIk cv = emu_chain_ik(c);
    auto load = [&](Pose &P) { P.r00 = Pin[0]; P.r01 = Pin[1]; P.r02 = Pin[2]; P.r10 = Pin[3]; P.r11 = Pin[4]; P.r12 = Pin[5]; P.r20 = Pin[6]; P.r21 = Pin[7]; P.r22 = Pin[8];
                               P.tx = Pin[9]; P.ty = Pin[10]; P.tz = Pin[11]; };
    auto store = [&](const Pose &P, double *o) { o[0] = P.r00; o[1] = P.r01; o[2] = P.r02; o[3] = P.r10; o[4] = P.r11; o[5] = P.r12; o[6] = P.r20; o[7] = P.r21; o[8] = P.r22;
                                                 o[9] = P.tx; o[10] = P.ty; o[11] = P.tz; };
    Pose A, B;
    load(A); load(B);
    const int jm = cv.jmeta[j];
    pose_mul_seg_cls(A, cv, j, jm);
    pose_mul_seg<true>(B, cv, j);
    store(A, out_cls); store(B, out_gen);
    return jm_cls(jm) | (jm_tmask(jm) << 4);
}

// the COMPILE-TIME form (pose_mul_seg_sig<class, mask>, what k_ik's signature instantiations execute) on the same segment: every (class, mask) pair
// is instantiated here and picked by the descriptor's bits
template <int CLS, int TM>
static void emu_sig_one(Pose &P, const EmuChainIk &cv, int j) { pose_mul_seg_sig<CLS, TM>(P, cv, j); }
template <int CLS>
static void emu_sig_tm(Pose &P, const EmuChainIk &cv, int j, int tm)
{
    switch (tm) {
    case 0: emu_sig_one<CLS, 0>(P, cv, j); break; case 1: emu_sig_one<CLS, 1>(P, cv, j); break; case 2: emu_sig_one<CLS, 2>(P, cv, j); break;
    case 3: emu_sig_one<CLS, 3>(P, cv, j); break; case 4: emu_sig_one<CLS, 4>(P, cv, j); break; case 5: emu_sig_one<CLS, 5>(P, cv, j); break;
    case 6: emu_sig_one<CLS, 6>(P, cv, j); break; default: emu_sig_one<CLS, 7>(P, cv, j); break;
    }
}
extern "C" int emu_pose_mul_seg_sig(rtbhip_chain_t h, int j, const double *Pin, double *out_sig)
{
    const std::shared_ptr<Chain> c_owner = chain_from_handle(h);
    Chain *c = c_owner.get();
    if (!c || j < 0 || j > c->n) return -1;
    const EmuChainIk cv = emu_chain_ik(c);
    Pose P;
    P.r00 = Pin[0]; P.r01 = Pin[1]; P.r02 = Pin[2]; P.r10 = Pin[3]; P.r11 = Pin[4]; P.r12 = Pin[5]; P.r20 = Pin[6]; P.r21 = Pin[7]; P.r22 = Pin[8];
    P.tx = Pin[9]; P.ty = Pin[10]; P.tz = Pin[11];
    const int jm = cv.jmeta[j], tm = jm_tmask(jm);
    switch (jm_cls(jm)) {
    case 0: emu_sig_tm<0>(P, cv, j, tm); break; case 1: emu_sig_tm<1>(P, cv, j, tm); break; case 2: emu_sig_tm<2>(P, cv, j, tm); break;
    case 3: emu_sig_tm<3>(P, cv, j, tm); break; case 4: emu_sig_tm<4>(P, cv, j, tm); break; case 5: emu_sig_tm<5>(P, cv, j, tm); break;
    case 6: emu_sig_tm<6>(P, cv, j, tm); break; case 7: emu_sig_tm<7>(P, cv, j, tm); break; case 8: emu_sig_tm<8>(P, cv, j, tm); break;
    case 9: emu_sig_tm<9>(P, cv, j, tm); break; case 10: emu_sig_tm<10>(P, cv, j, tm); break; case 11: emu_sig_tm<11>(P, cv, j, tm); break;
    default: emu_sig_tm<12>(P, cv, j, tm); break;
    }
    out_sig[0] = P.r00; out_sig[1] = P.r01; out_sig[2] = P.r02; out_sig[3] = P.r10; out_sig[4] = P.r11; out_sig[5] = P.r12; out_sig[6] = P.r20; out_sig[7] = P.r21; out_sig[8] = P.r22;
    out_sig[9] = P.tx; out_sig[10] = P.ty; out_sig[11] = P.tz;
    return 0;
}
// the structure signature of a chain (kin_reg.h: chain_signature), for the test that pins the Panda's
extern "C" unsigned long long emu_chain_signature(rtbhip_chain_t h)
{
    const std::shared_ptr<Chain> c_owner = chain_from_handle(h);
    Chain *c = c_owner.get();
    return c ? chain_signature(c->jmeta.data(), c->n) : 0ull;
}

// the folded constant segments of a chain (n + 1 rows of 12 doubles: R row-major, t) and their descriptor words -- for probes and tests
extern "C" int emu_chain_segments(rtbhip_chain_t h, double *seg12, int32_t *jmeta)
{
    const std::shared_ptr<Chain> c_owner = chain_from_handle(h);
    Chain *c = c_owner.get();
    if (!c) return -1;
    for (int j = 0; j <= c->n; ++j) {
        for (int k = 0; k < 9; ++k) seg12[12 * j + k] = c->seg[j].r[k];
        for (int k = 0; k < 3; ++k) seg12[12 * j + 9 + k] = c->seg[j].t[k];
        jmeta[j] = c->jmeta[j];
    }
    return c->n;
}
