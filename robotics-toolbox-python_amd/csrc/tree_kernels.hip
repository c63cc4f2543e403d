// tree_kernels.hip -- gfx950 kernel for batched inverse dynamics of ETS robots (link trees): replaces the
// Python triple loop of Robot.rne (reference robot/Robot.py:1809-1893: samples x groups, spatialmath
// objects per step).  One lane = one (q, qd, qdd) sample, the per-lane recursion of tree_device.h with
// the group count a template parameter (per-group force / sin / cos in named registers), one tile of 64
// samples per single-wave workgroup, inputs and torques through the wave's LDS transposer so global
// accesses are contiguous, branch-point state in per-lane LDS slots.
// Bytes: 3*8n in + 8n out per sample; ~0.25 kflop per group: fp64-issue bound like k_rne.
#include "tree_kernels.h"

namespace rtbhip {

constexpr int kTreeMaxGroups = 24;   // 17..24 link groups: the per-group state spills to scratch; served, not fast

// ATREST (qd == NULL, robots of up to kTreeAtRestMax groups): the velocity half of the recursion is not compiled in and the qd row is neither
// read nor staged.
constexpr int kTreeAtRestMax = 12;
// kTreeSigUR, kTreeSigIbx8, kTreeSigPx100 (tree_device.h): the signatures this build has instantiations for; every other robot takes the general kernels
#if RTB_HOST_SIDE
static int g_tree_sig = 1;          // rtbhip_tune("tree_sig", 0): every robot does (A/B, tests)
void tree_tune(const char *key, int value) { if (std::string(key) == "tree_sig") g_tree_sig = value != 0; }
int tree_sig_enabled() { return g_tree_sig; }
#endif

template <int NG, bool ATREST, class KN = TreeNothing>
__global__ __launch_bounds__(kWave, (NG <= 8 ? 2 : 1)) void k_tree_rne(TreeParams tp, const DevGroup *groups_g, const double *__restrict__ q,
                                                      const double *__restrict__ qd, const double *__restrict__ qdd,
                                                      double *__restrict__ tau)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    ConstGroups groups = (ConstGroups)groups_g;
    const int lane = threadIdx.x;
    constexpr int stride = (4 * NG) | 1;            // per lane: q | qd | qdd | tau
    double *slots = lds + kWave * stride;                       // [slot * 18 + k][lane]
    const int64_t cfg0 = (int64_t)blockIdx.x * kWave;
    const int64_t left = tp.N - cfg0;
    const int ncfg = left < kWave ? (int)left : kWave;
    const int count = ncfg * NG;
    {
        double r0[NG], r1[NG], r2[NG];
        const double *g0 = q + cfg0 * NG, *g1 = qd + cfg0 * NG, *g2 = qdd + cfg0 * NG;
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int f = lane + kWave * k;
            const bool in = f < count;
            r0[k] = in ? g0[f] : 0.0;
            r1[k] = (!ATREST && in && qd) ? g1[f] : 0.0;
            r2[k] = (in && qdd) ? g2[f] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int f = lane + kWave * k;
            const int r = f / NG, c = f - r * NG;
            double *dst = lds + r * stride + c;
            dst[0] = r0[k]; dst[2 * NG] = r2[k];
            if (!ATREST) dst[NG] = r1[k];
        }
    }
    __syncthreads();
    double *mine = lds + lane * stride;
    if (lane < ncfg)
        tree_rne_lane<NG, ATREST, KN>(groups, tp.nslots, v3(tp.grav[0], tp.grav[1], tp.grav[2]), [&](int c) { return mine[c]; },
                          [&](int c) { return ATREST ? 0.0 : mine[NG + c]; }, [&](int c) { return mine[2 * NG + c]; },
                          [&](int c, double v) { mine[3 * NG + c] = v; },
                          [&](int i) -> double & { return slots[i * kWave + lane]; });
    __syncthreads();
    flush_run(lds + 3 * NG, stride, NG, ncfg, tau + cfg0 * NG, lane);
}

#if RTB_HOST_SIDE      // the launchers (the kernel above is also what jit.cpp hands to hipRTC, one instantiation at a time)
// ---- run-time instantiation: the KNOWLEDGE TYPE of a tree (tree_device.h: TreeKnown's members) as generated source.  Any tree whose groups are
// numbered in group order (group j moves q column j and owns torque column j: every URDF robot) qualifies -- no limit on groups or branch slots.
static bool tree_builtin(const Tree *t)
{
    const SegSig sig = t->sig, sig2 = t->sig2;
    const TreeTopo topo = t->topo;
    if (!jit_builtin_enabled()) return false;       // (the dispatch below still finds the built-in kernel while the run-time one is being compiled: same bits)
    return (sig == kTreeSigUR && t->n == 6) || (sig == kTreeSigIbx8 && topo == kTreeTopoIbx8 && t->n == 8) || (sig == kTreeSigPx100 && topo == kTreeTopoPx100 && t->n == 7) ||
           (sig == kTreeSigIbx9 && sig2 == kTreeSig2Ibx9 && topo == kTreeTopoIbx9 && t->n == 9) || (sig == kTreeSigFetch && sig2 == kTreeSig2Fetch && topo == kTreeTopoFetch && t->n == 10) ||
           (sig == kTreeSigMico && sig2 == kTreeSig2Mico && topo == kTreeTopoMico && t->n == 10);
}
bool tree_jit_applies(const Tree *t)
{
    if (t->n < 1 || tree_builtin(t)) return false;
    for (int j = 0; j < t->n; ++j) {
        const DevGroup &g = t->groups[j];
        if (jm_jq(g.jmeta) != j || g.out_col != j || g.parent < -1 || g.parent >= j) return false;
    }
    return true;
}
// the type's name carries a hash of its body: the same structure gives the same name (and the same cache key) in every process
std::string tree_jit_knowledge(const Tree *t, std::string *type_name)
{
    const int n = t->n;
    bool plain = true;
    for (int j = 0; j < n; ++j) {
        const DevGroup &g = t->groups[j];
        plain = plain && g.parent == j - 1 && g.save_slot < 0 && g.parent_slot < 0 && !jm_prismatic(g.jmeta);
    }
    auto table = [&](const char *ret, const char *name, const char *args, const std::function<int(int)> &val, const char *dflt) {
        std::string o = std::string("    RTB_HD static constexpr ") + ret + " " + name + "(" + args + ") { return ";
        for (int j = 0; j < n; ++j) o += "j == " + std::to_string(j) + " ? " + std::to_string(val(j)) + " : ";
        return o + dflt + "; }\n";
    };
    std::string body;
    body += std::string("    static constexpr bool known = true, plain = ") + (plain ? "true" : "false") + ", topo = true, any = true;\n";
    body += table("int", "cls", "int j", [&](int j) { return jm_cls(seg_class_bits(t->groups[j].C)); }, "0");
    body += table("int", "tm_", "int j", [&](int j) { return jm_tmask(seg_class_bits(t->groups[j].C)); }, "7");
    body += "    RTB_HD static constexpr int tm(int j, bool rev) { return rev ? tm_(j) : 7; }\n";
    body += table("bool", "pris", "int j", [&](int j) { return jm_prismatic(t->groups[j].jmeta) ? 1 : 0; }, "false");
    body += "    RTB_HD static constexpr bool revolute(int j) { return !pris(j); }\n";
    body += table("int", "parent", "int j", [&](int j) { return (int)t->groups[j].parent; }, "-1");
    body += table("int", "parent_slot", "int j", [&](int j) { return (int)t->groups[j].parent_slot; }, "-1");
    body += table("int", "save_slot", "int j", [&](int j) { return (int)t->groups[j].save_slot; }, "-1");
    unsigned long long h = 1469598103934665603ull;
    for (unsigned char c : body) { h ^= c; h *= 1099511628211ull; }
    char nm[48];
    std::snprintf(nm, sizeof nm, "JitTree%d_%016llx", n, h);
    *type_name = std::string("rtbhip::") + nm;
    return std::string("namespace rtbhip {\nstruct ") + nm + " {\n" + body + "};\n}\n";
}
// variant 0 k_tree_rne, 1 k_tree_rne at rest (tree_kernels.hip); 2 + mode k_tree_dyn (tree_dyn_kernels.hip)
std::string tree_jit_expr(const Tree *t, const std::string &type_name, int variant)
{
    const std::string ng = std::to_string(t->n);
    if (variant == 0) return "rtbhip::k_tree_rne<" + ng + ", false, " + type_name + ">";
    if (variant == 1) return "rtbhip::k_tree_rne<" + ng + ", true, " + type_name + ">";
    return "rtbhip::k_tree_dyn<" + ng + ", " + std::to_string(variant - 2) + ", " + type_name + ">";
}
std::vector<std::string> tree_jit_names(const Tree *t)
{
    std::vector<std::string> out;
    if (!tree_jit_applies(t) || t->n > RTBHIP_MAX_JOINTS) return out;
    std::string type_name;
    (void)tree_jit_knowledge(t, &type_name);
    out.push_back(tree_jit_expr(t, type_name, 0));
    if (t->n <= 12) out.push_back(tree_jit_expr(t, type_name, 1));
    for (int m = 0; m < 3; ++m) out.push_back(tree_jit_expr(t, type_name, 2 + m));
    return out;
}
// the function of variant `variant` on the current device, or NULL (not compiled yet / no hipRTC / jit or signatures off / not applicable)
hipFunction_t tree_jit_function(const Tree *t, int variant)
{
    if (!g_tree_sig || !jit_enabled() || !tree_jit_applies(t)) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    {
        std::lock_guard<std::mutex> lk(t->jit.mu);
        auto it = t->jit.fn.find(((uint64_t)dev << 8) | (uint64_t)variant);
        if (it != t->jit.fn.end()) return it->second;
    }
    std::string type_name;
    const std::string pre = tree_jit_knowledge(t, &type_name);
    return t->jit.get(variant < 2 ? "tree_kernels.hip" : "tree_dyn_kernels.hip", variant, [&] { return tree_jit_expr(t, type_name, variant); }, pre);
}

template <int NG, SegSig SIG = 0, TreeTopo TOPO = 0, SegSig SIG2 = 0>
static void launch_ng(dim3 grid, size_t lds, hipStream_t s, const TreeParams &tp, const DevGroup *g, const double *q,
                      const double *qd, const double *qdd, double *tau, bool plain = false)
{
    if constexpr (SIG == 0 && NG <= kTreePlainChainMax) {
        if (plain) { launch_ng<NG, kTreeSigPlainChain>(grid, lds, s, tp, g, q, qd, qdd, tau); return; }      // a serial chain of revolute joints: straight-line code
    }
    if constexpr (NG <= kTreeAtRestMax) {
        if (!qd) {
            auto k = k_tree_rne<NG, true, TreeKnown<SIG, TOPO, SIG2>>;
            if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k, grid, dim3(kWave), lds, s, tp, g, q, qd, qdd, tau);
            return;
        }
    }
    auto k = k_tree_rne<NG, false, TreeKnown<SIG, TOPO, SIG2>>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, dim3(kWave), lds, s, tp, g, q, qd, qdd, tau);
}

int launch_tree_rne(const Tree *t, const DevGroup *groups, const double *q, const double *qd, const double *qdd, int64_t N,
                    const double *grav3, double *tau, hipStream_t s)
{
    if (N == 0) return RTBHIP_OK;
    if (t->n > RTBHIP_MAX_JOINTS) { set_error("tree_rne: more than RTBHIP_MAX_JOINTS joints (link groups)"); return RTBHIP_ELIMIT; }
    const int64_t tiles = (N + kWave - 1) / kWave;
    if (tiles > 0x7fffffff) { set_error("tree_rne: batch too large for one launch"); return RTBHIP_ELIMIT; }
    TreeParams tp;
    tp.n = t->n; tp.nslots = t->nslots; tp.N = N; tp.tile = kWave; tp.pad_ = 0;
    for (int i = 0; i < 3; i++) tp.grav[i] = grav3[i];
    const size_t lds = (size_t)kWave * (((4 * t->n) | 1) + kTreeSlotDoubles * t->nslots) * sizeof(double);
    if (lds > 160 * 1024) { set_error("tree_rne: tree needs more LDS than a CU has"); return RTBHIP_ELIMIT; }
    dim3 grid((unsigned)tiles);
    const SegSig sig = g_tree_sig ? t->sig : 0;
    const bool plain = (sig & kTreeSigPlain) != 0;
    const TreeTopo topo = g_tree_sig ? t->topo : 0;
    const SegSig sig2 = g_tree_sig ? t->sig2 : 0;
    // a robot without a built-in instantiation: its own, compiled at run time (jit.cpp); the general kernels below serve until it is there
    hipFunction_t f = tree_jit_function(t, (!qd && t->n <= kTreeAtRestMax) ? 1 : 0);
    if (!f && t->n > kTreeMaxGroups) {
        // more groups than the built-in sizes (1 .. 24) and no knowledge-type instantiation to hand: the general kernel of this size, instantiated
        // at run time; the caller waits (the reference's loops take any n: robot/Robot.py:1704-1903)
        f = t->jit.get_wait("tree_kernels.hip", 8, [&] { return "rtbhip::k_tree_rne<" + std::to_string(t->n) + ", false, rtbhip::TreeNothing>"; });
        if (!f) return RTBHIP_ELIMIT;
    }
    if (f) {
        TreeParams tpv = tp;
        void *args[] = {&tpv, &groups, &q, &qd, &qdd, &tau};
        const int rc = jit_launch(f, grid, dim3(kWave), lds, s, args);
        if (rc != RTBHIP_OK) return rc;
        note_launch((int)grid.x, kWave, (int)lds);
        return RTBHIP_OK;
    }
    if (sig == kTreeSigUR && t->n == 6) {       // (a signature does not encode the group count: a trailing General / t = 0 group reads as "absent")
        launch_ng<6, kTreeSigUR>(grid, lds, s, tp, groups, q, qd, qdd, tau);
    } else if (sig == kTreeSigIbx8 && topo == kTreeTopoIbx8 && t->n == 8) {
        launch_ng<8, kTreeSigIbx8, kTreeTopoIbx8>(grid, lds, s, tp, groups, q, qd, qdd, tau);
    } else if (sig == kTreeSigPx100 && topo == kTreeTopoPx100 && t->n == 7) {
        launch_ng<7, kTreeSigPx100, kTreeTopoPx100>(grid, lds, s, tp, groups, q, qd, qdd, tau);
    } else if (sig == kTreeSigIbx9 && sig2 == kTreeSig2Ibx9 && topo == kTreeTopoIbx9 && t->n == 9) {
        launch_ng<9, kTreeSigIbx9, kTreeTopoIbx9, kTreeSig2Ibx9>(grid, lds, s, tp, groups, q, qd, qdd, tau);
    } else if (sig == kTreeSigFetch && sig2 == kTreeSig2Fetch && topo == kTreeTopoFetch && t->n == 10) {
        launch_ng<10, kTreeSigFetch, kTreeTopoFetch, kTreeSig2Fetch>(grid, lds, s, tp, groups, q, qd, qdd, tau);
    } else if (sig == kTreeSigMico && sig2 == kTreeSig2Mico && topo == kTreeTopoMico && t->n == 10) {
        launch_ng<10, kTreeSigMico, kTreeTopoMico, kTreeSig2Mico>(grid, lds, s, tp, groups, q, qd, qdd, tau);
    } else
#ifdef RTB_TREE_DEV_NG      // development builds (seconds instead of minutes): only this size is instantiated
    launch_ng<RTB_TREE_DEV_NG>(grid, lds, s, tp, groups, q, qd, qdd, tau, plain);
#else
    switch (t->n) {
    case 1: launch_ng<1>(grid, lds, s, tp, groups, q, qd, qdd, tau, plain); break;
    case 2: launch_ng<2>(grid, lds, s, tp, groups, q, qd, qdd, tau, plain); break;
    case 3: launch_ng<3>(grid, lds, s, tp, groups, q, qd, qdd, tau, plain); break;
    case 4: launch_ng<4>(grid, lds, s, tp, groups, q, qd, qdd, tau, plain); break;
    case 5: launch_ng<5>(grid, lds, s, tp, groups, q, qd, qdd, tau, plain); break;
    case 6: launch_ng<6>(grid, lds, s, tp, groups, q, qd, qdd, tau, plain); break;
    case 7: launch_ng<7>(grid, lds, s, tp, groups, q, qd, qdd, tau, plain); break;
    case 8: launch_ng<8>(grid, lds, s, tp, groups, q, qd, qdd, tau, plain); break;
    case 9: launch_ng<9>(grid, lds, s, tp, groups, q, qd, qdd, tau, plain); break;
    case 10: launch_ng<10>(grid, lds, s, tp, groups, q, qd, qdd, tau, plain); break;
    case 11: launch_ng<11>(grid, lds, s, tp, groups, q, qd, qdd, tau, plain); break;
    case 12: launch_ng<12>(grid, lds, s, tp, groups, q, qd, qdd, tau, plain); break;
    case 13: launch_ng<13>(grid, lds, s, tp, groups, q, qd, qdd, tau, plain); break;
    case 14: launch_ng<14>(grid, lds, s, tp, groups, q, qd, qdd, tau, plain); break;
    case 15: launch_ng<15>(grid, lds, s, tp, groups, q, qd, qdd, tau, plain); break;
    case 16: launch_ng<16>(grid, lds, s, tp, groups, q, qd, qdd, tau, plain); break;
    case 17: launch_ng<17>(grid, lds, s, tp, groups, q, qd, qdd, tau, plain); break;
    case 18: launch_ng<18>(grid, lds, s, tp, groups, q, qd, qdd, tau, plain); break;
    case 19: launch_ng<19>(grid, lds, s, tp, groups, q, qd, qdd, tau, plain); break;
    case 20: launch_ng<20>(grid, lds, s, tp, groups, q, qd, qdd, tau, plain); break;
    case 21: launch_ng<21>(grid, lds, s, tp, groups, q, qd, qdd, tau, plain); break;
    case 22: launch_ng<22>(grid, lds, s, tp, groups, q, qd, qdd, tau, plain); break;
    case 23: launch_ng<23>(grid, lds, s, tp, groups, q, qd, qdd, tau, plain); break;
    default: launch_ng<24>(grid, lds, s, tp, groups, q, qd, qdd, tau, plain); break;
    }
#endif
    note_launch((int)grid.x, kWave, (int)lds);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "k_tree_rne launch");
    return RTBHIP_OK;
}

#endif  // RTB_HOST_SIDE

}  // namespace rtbhip
