// rne_kernels.hip -- gfx950 kernel for batched inverse dynamics of DH / MDH chains.
// Replaces the Python per-row loop of DHRobot.rne (robot/DHRobot.py:1442-1451) around frne.frne
// (core/frne.c:106-230) -> newton_euler (core/ne.c:62-493).
//
// One lane = one (q, qd, qdd) sample; both Newton-Euler recursions fused in registers (rne_device.h).
// I/O per sample: 3*8n bytes in, 8n bytes out (224 B for the 7-DOF Panda) against ~2.2 kflop + n
// sincos of fp64 VALU: the HBM and fp64-issue roofs are within 2x of each other for this kernel.
// The wave's (64 x n) input tiles are contiguous in memory; they are fetched with fully coalesced
// 8-byte-per-lane loads into a lane-major LDS tile (odd row stride => conflict-free) and the torques
// leave the same way.
#include "rne_device.h"
#if RTB_HOST_SIDE
#include <cstdlib>
#endif

namespace rtbhip {

typedef const __attribute__((address_space(4))) DevLink *ConstLinks;
constexpr int kW = 64;

struct RneParams {
    int32_t n, has_fext;
    int64_t N;
    double grav[3];
    double fext[6];
    double *wbase;       // (N, 6) base wrench, or NULL (served by the run-time-n kernel only)
};

// receiver of the base wrench in the run-time-n kernel: row `cfg` of rp.wbase
struct WrenchRow {
    static constexpr bool on = true;
    double *row;
    RTB_HD bool wanted() const { return row != nullptr; }
    RTB_HD void operator()(int k, double v) const { row[k] = v; }
};

__device__ __forceinline__ int rne_stride(int n) { int s = 3 * n; return (s & 1) ? s : s + 1; }

// coalesced copy of `count` doubles between a contiguous global run and lane-major LDS rows
// (row = element index / n, col = element % n, placed at rows[row*stride + col_off + col]).
template <bool TO_LDS>
__device__ __forceinline__ void tile_copy(double *rows, int stride, int col_off, int n, int count,
                                          const double *__restrict__ gsrc, double *__restrict__ gdst, int lane)
{
    int f = lane;
    int r = f / n, c = f - r * n;
    const int da = kW / n, db = kW - da * n;
#ifndef RTB_RNE_RT_BATCH
#define RTB_RNE_RT_BATCH 1
#endif
    if (RTB_RNE_RT_BATCH && TO_LDS && gsrc) {
        // four loads in flight per trip (the trip count is a run-time value: a plain trip is load -> wait -> LDS store, one HBM round trip per 8 bytes)
        for (; f + 3 * kW < count; f += 4 * kW) {
            const double v0 = gsrc[f], v1 = gsrc[f + kW], v2 = gsrc[f + 2 * kW], v3 = gsrc[f + 3 * kW];
            const double v[4] = {v0, v1, v2, v3};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                rows[r * stride + col_off + c] = v[k];
                c += db; r += da;
                if (c >= n) { c -= n; r += 1; }
            }
        }
    }
    for (; f < count; f += kW) {
        if (TO_LDS) rows[r * stride + col_off + c] = gsrc ? gsrc[f] : 0.0;
        else gdst[f] = rows[r * stride + col_off + c];
        c += db; r += da;
        if (c >= n) { c -= n; r += 1; }
    }
}

#ifndef RTB_RNE_WAVES
#define RTB_RNE_WAVES 2   // waves per SIMD the register allocator must leave room for (<= 256 VGPRs)
#endif

// One tile: stage the wave's (64 x n) q / qd / qdd blocks in LDS (coalesced), run the per-lane
// recursion with the LDS row as the accessor target (q is read once, up front; qd/qdd are re-read
// from LDS where they are used, so they occupy no registers across the recursions; tau overwrites the
// q slots, which are dead by then), write the torques back coalesced.
template <int NJ, bool MDH, bool ALLREV, bool ATREST = false, RneSig SIG = 0>
__device__ __forceinline__ void rne_tile(const RneParams &rp, ConstLinks links, int n, int stride, int64_t tile,
                                         const double *__restrict__ q, const double *__restrict__ qd,
                                         const double *__restrict__ qdd, double *__restrict__ tau, double *lds, int lane)
{
    const V3 grav = v3(rp.grav[0], rp.grav[1], rp.grav[2]);
    const V3 ftip = rp.has_fext ? v3(rp.fext[0], rp.fext[1], rp.fext[2]) : v3(0, 0, 0);
    const V3 ntip = rp.has_fext ? v3(rp.fext[3], rp.fext[4], rp.fext[5]) : v3(0, 0, 0);
    double *mine = lds + lane * stride;
    const int64_t cfg0 = tile * kW;
    const int64_t left = rp.N - cfg0;
    const int ncfg = left < kW ? (int)left : kW;
    const int count = ncfg * n;
    if (NJ > 0) {
        // all 3*NJ coalesced 8-byte loads of the tile in flight at once (a run-time-trip-count copy
        // loop serialises load -> wait -> LDS write: 21 dependent HBM round trips per tile measured as
        // 65 % of the wave lifetime in s_waitcnt), then the LDS transposition.
        constexpr int C = NJ > 0 ? NJ : 1;
        double r0[C], r1[C], r2[C];
        const double *g0 = q + cfg0 * NJ, *g1 = qd + cfg0 * NJ, *g2 = qdd + cfg0 * NJ;
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const int f = lane + kW * k;
            const bool in = f < count;
#if defined(RTB_RNE_PROBE_NOLOAD)   // timing probe only (wrong results): what the tile loads cost the wave's lifetime
            r0[k] = in ? 1e-3 * f : 0.0; r1[k] = in ? 2e-3 * f : 0.0; r2[k] = in ? 3e-3 * f : 0.0;
            (void)g0; (void)g1; (void)g2;
#else
            r0[k] = in ? g0[f] : 0.0;
            r1[k] = (in && qd) ? g1[f] : 0.0;      // NULL qd / qdd = zeros (gravload, itorque)
            r2[k] = (in && qdd) ? g2[f] : 0.0;
#endif
        }
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const int f = lane + kW * k;
            const int r = f / C, c = f - r * C;
            double *dst = lds + r * stride + c;
            dst[0] = r0[k];
            dst[C] = r1[k];
            dst[2 * C] = r2[k];
        }
    } else {
        tile_copy<true>(lds, stride, 0, n, count, q + cfg0 * n, nullptr, lane);
        tile_copy<true>(lds, stride, n, n, count, qd ? qd + cfg0 * n : nullptr, nullptr, lane);
        tile_copy<true>(lds, stride, 2 * n, n, count, qdd ? qdd + cfg0 * n : nullptr, nullptr, lane);
    }
    __syncthreads();
    if (lane < ncfg) {
        if constexpr (ATREST)
            rne_lane<NJ, MDH, false, true, true, SIG>(links, n, grav, ftip, ntip, [&](int j) { return mine[j]; }, [&](int) { return 0.0; },
                              [&](int j) { return mine[2 * n + j]; }, [&](int j, double v) { mine[j] = v; });
        else if constexpr (NJ == 0)
            rne_lane<NJ, MDH, true, ALLREV>(links, n, grav, ftip, ntip, [&](int j) { return mine[j]; }, [&](int j) { return mine[n + j]; },
                              [&](int j) { return mine[2 * n + j]; }, [&](int j, double v) { mine[j] = v; },
                              WrenchRow{rp.wbase ? rp.wbase + (cfg0 + lane) * 6 : nullptr});
        else
            rne_lane<NJ, MDH, true, ALLREV, false, SIG>(links, n, grav, ftip, ntip, [&](int j) { return mine[j]; }, [&](int j) { return mine[n + j]; },
                              [&](int j) { return mine[2 * n + j]; }, [&](int j, double v) { mine[j] = v; });
    }
    __syncthreads();
    if (NJ > 0) {
        constexpr int C = NJ > 0 ? NJ : 1;
        double *g = tau + cfg0 * NJ;
        double r0[C];
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const int f = lane + kW * k;
            const int r = f / C, c = f - r * C;
            r0[k] = lds[r * stride + c];
        }
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const int f = lane + kW * k;
            if (f < count) __builtin_nontemporal_store(r0[k], g + f);
        }
    } else {
        tile_copy<false>(lds, stride, 0, n, count, nullptr, tau + cfg0 * n, lane);
    }
}

// compile-time joint count: ONE tile per single-wave workgroup (no grid-stride loop -- with a loop
// LICM hoists every link's scalar table loads into the preheader, where they overflow the SGPR file;
// first MI355X measurement of the looped version: 256 VGPRs + 90 AGPRs + 112 spilled SGPRs, one
// wave per SIMD, 0.31 ms per 1.25e6 Panda triples).
template <int NJ, bool MDH, bool ALLREV, RneSig SIG = 0>
__global__ __launch_bounds__(kW, (SIG ? 3 : RTB_RNE_WAVES)) void k_rne(RneParams rp, const DevLink *links_g, const double *__restrict__ q,
                                           const double *__restrict__ qd, const double *__restrict__ qdd,
                                           double *__restrict__ tau)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    rne_tile<NJ, MDH, ALLREV, false, SIG>(rp, (ConstLinks)links_g, NJ, rne_stride(NJ), blockIdx.x, q, qd, qdd, tau, lds, threadIdx.x);
}

// The same tile function, WPB waves per workgroup, every wave on a tile of its own (rtbhip_tune("rne_wpb", 2 | 4)): a quarter of the
// workgroup launches for the same waves.  The two barriers of rne_tile then couple the workgroup's waves (they load, compute and store in
// step); tiles past the end are empty (count <= 0: nothing loaded, nothing stored).
template <int NJ, bool MDH, bool ALLREV, int WPB>
__global__ __launch_bounds__(kW * WPB, RTB_RNE_WAVES) void k_rne_wpb(RneParams rp, const DevLink *links_g, const double *__restrict__ q,
                                                                     const double *__restrict__ qd, const double *__restrict__ qdd,
                                                                     double *__restrict__ tau)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int wave = threadIdx.x / kW, lane = threadIdx.x % kW;
    const int stride = rne_stride(NJ);
    rne_tile<NJ, MDH, ALLREV>(rp, (ConstLinks)links_g, NJ, stride, (int64_t)blockIdx.x * WPB + wave, q, qd, qdd, tau, lds + wave * kW * stride, lane);
}

// Persistent form of k_rne (rtbhip_tune("rne_persist", 1)): the grid is sized to the chip (three waves per SIMD) and every wave walks
// tiles blockIdx.x, + gridDim.x, ... .  The loop the comment above warns about, made safe the way k_ik does it: the link-table pointer is
// laundered once per trip, so the scalar loads stay inside the trip (the scalar cache serves them) instead of being hoisted into SGPRs that
// do not exist.  What it buys: no workgroup dispatch between tiles (a slot that finishes a tile starts the next one at once).
template <int NJ, bool MDH, bool ALLREV>
__global__ __launch_bounds__(kW, RTB_RNE_WAVES) void k_rne_persist(RneParams rp, const DevLink *links_g, const double *__restrict__ q,
                                                   const double *__restrict__ qd, const double *__restrict__ qdd,
                                                   double *__restrict__ tau)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int64_t tiles = (rp.N + kW - 1) / kW;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        ConstLinks l = (ConstLinks)links_g;
        asm volatile("" : "+s"(l));
        rne_tile<NJ, MDH, ALLREV>(rp, l, NJ, rne_stride(NJ), tile, q, qd, qdd, tau, lds, threadIdx.x);
        __syncthreads();
    }
}

// qd = NULL on an all-revolute chain (Dynamics.gravload: qd = qdd = 0; Dynamics.itorque: qd = 0, no gravity): every link's
// angular velocity is zero, so the forward recursion is the acceleration-only one of rne_device.h (ACC) from link 0 -- about
// half its fp64 operations -- with gravity entering as the base's linear acceleration; the backward recursion is the usual one.
template <int NJ, bool MDH, RneSig SIG = 0>
__global__ __launch_bounds__(kW, RTB_RNE_WAVES) void k_rne_atrest(RneParams rp, const DevLink *links_g, const double *__restrict__ q,
                                                  const double *__restrict__ qdd, double *__restrict__ tau)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    rne_tile<NJ, MDH, true, true, SIG>(rp, (ConstLinks)links_g, NJ, rne_stride(NJ), blockIdx.x, q, nullptr, qdd, tau, lds, threadIdx.x);
}

// run-time joint count (n > 8): grid-stride over tiles, per-link state in private memory
template <bool MDH>
__global__ __launch_bounds__(kW) void k_rne_rt(RneParams rp, const DevLink *links_g, const double *__restrict__ q,
                                              const double *__restrict__ qd, const double *__restrict__ qdd,
                                              double *__restrict__ tau)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int n = rp.n;
    const int stride = rne_stride(n);
    const int64_t tiles = (rp.N + kW - 1) / kW;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        rne_tile<0, MDH, false>(rp, (ConstLinks)links_g, n, stride, tile, q, qd, qdd, tau, lds, threadIdx.x);
        __syncthreads();
    }
}

#if RTB_HOST_SIDE      // the launchers (the kernels above are also what jit.cpp hands to hipRTC, one instantiation at a time)
namespace { int g_rne_tiles_per_wave = 1; int g_rne_persist = 0; int g_rne_wpb = 1; int g_rne_sig = 1; }
int rne_sig_enabled() { return g_rne_sig; }
void rne_tune(const char *key, int value)
{
    if (std::string(key) == "rne_sig") g_rne_sig = value != 0;          // 0: never take a structure signature's instantiation (A/B, tests)
    if (std::string(key) == "rne_persist") g_rne_persist = value < 0 ? 0 : (value > 4 ? 4 : value);      // waves per SIMD of the persistent grid, 0 = one tile per workgroup
    if (std::string(key) == "rne_wpb") g_rne_wpb = (value == 2 || value == 4) ? value : 1;
    if (std::string(key) == "rne_tiles_per_wave") g_rne_tiles_per_wave = value < 1 ? 1 : value;
}

// a robot whose link table has a structure signature this build is instantiated for (rne_device.h: kRneSig*): straight-line kernels
template <int NJ, bool MDH, RneSig SIG>
static void launch_sig(dim3 grid, size_t lds, hipStream_t s, const RneParams &rp, const DevLink *links, const double *q, const double *qd, const double *qdd,
                       double *tau)
{
    if (!qd) hipLaunchKernelGGL((k_rne_atrest<NJ, MDH, SIG>), grid, dim3(kW), lds, s, rp, links, q, qdd, tau);
    else hipLaunchKernelGGL((k_rne<NJ, MDH, true, SIG>), grid, dim3(kW), lds, s, rp, links, q, qd, qdd, tau);
}

// The signatures with an instantiation built into the library; every other signature gets one at run time (jit.cpp).
static bool rne_sig_builtin(int n, bool mdh, RneSig sig) { return jit_builtin_enabled() && ((n == 7 && mdh && sig == kRneSigPanda) || (n == 6 && !mdh && sig == kRneSigPuma560)); }
// name expressions of the run-time instantiations of a DH table with signature `sig` (all links revolute, n <= 8): variant 0 k_rne, 1 k_rne_atrest
// (rne_kernels.hip), 2 + mode k_dyn (dyn_kernels.hip)
std::string rne_jit_expr(int n, bool mdh, RneSig sig, int variant)
{
    const std::string nj = std::to_string(n), m = mdh ? "true" : "false", sg = jit_hex(sig);
    if (variant == 0) return "rtbhip::k_rne<" + nj + ", " + m + ", true, " + sg + ">";
    if (variant == 1) return "rtbhip::k_rne_atrest<" + nj + ", " + m + ", " + sg + ">";
    return "rtbhip::k_dyn<" + nj + ", " + m + ", " + std::to_string(variant - 2) + ", true, " + sg + ">";
}
std::vector<std::string> rne_jit_names(const Dyn *d)
{
    std::vector<std::string> out;
    const RneSig sig = rne_signature(d->links.data(), d->n);
    if (!sig || rne_sig_builtin(d->n, d->mdh != 0, sig)) return out;
    for (int v = 0; v < 5; ++v) out.push_back(rne_jit_expr(d->n, d->mdh != 0, sig, v));
    return out;
}

template <int NJ>
static void launch_nj(const Dyn *d, bool mdh, bool allrev, dim3 grid, size_t lds, hipStream_t s, const RneParams &rp, const DevLink *links,
                      const double *q, const double *qd, const double *qdd, double *tau, RneSig sig = 0)
{
    if constexpr (NJ == 7) {
        if (sig == kRneSigPanda && mdh && !g_rne_persist && g_rne_wpb == 1 && jit_builtin_enabled()) { launch_sig<7, true, kRneSigPanda>(grid, lds, s, rp, links, q, qd, qdd, tau); return; }
    }
    if constexpr (NJ == 6) {
        if (sig == kRneSigPuma560 && !mdh && !g_rne_persist && jit_builtin_enabled()) { launch_sig<6, false, kRneSigPuma560>(grid, lds, s, rp, links, q, qd, qdd, tau); return; }
    }
    // any other robot with a signature (all links revolute, n <= 8): its own instantiation of the same kernels, compiled at run time; until the
    // code object is there (or when hipRTC is not) the general kernels below serve -- the same numbers
    if (sig && !rne_sig_builtin(NJ, mdh, sig) && jit_enabled() && !g_rne_persist && g_rne_wpb == 1) {
        const int variant = qd ? 0 : 1;
        if (hipFunction_t f = d->jit.get("rne_kernels.hip", variant, [&] { return rne_jit_expr(NJ, mdh, sig, variant); })) {
            RneParams rpv = rp;
            if (qd) { void *args[] = {&rpv, &links, &q, &qd, &qdd, &tau}; (void)jit_launch(f, grid, dim3(kW), lds, s, args); }
            else { void *args[] = {&rpv, &links, &q, &qdd, &tau}; (void)jit_launch(f, grid, dim3(kW), lds, s, args); }
            return;
        }
    }
    if (allrev && !qd) {
        if (mdh) hipLaunchKernelGGL((k_rne_atrest<NJ, true>), grid, dim3(kW), lds, s, rp, links, q, qdd, tau);
        else hipLaunchKernelGGL((k_rne_atrest<NJ, false>), grid, dim3(kW), lds, s, rp, links, q, qdd, tau);
        return;
    }
    if (g_rne_persist && allrev) {
        int cus = 0;
        if (device_cu_count(&cus) == RTBHIP_OK && (int64_t)grid.x > (int64_t)cus * 4 * g_rne_persist) {
            const dim3 pg((unsigned)(cus * 4 * g_rne_persist));          // g_rne_persist waves per SIMD
            if (mdh) hipLaunchKernelGGL((k_rne_persist<NJ, true, true>), pg, dim3(kW), lds, s, rp, links, q, qd, qdd, tau);
            else hipLaunchKernelGGL((k_rne_persist<NJ, false, true>), pg, dim3(kW), lds, s, rp, links, q, qd, qdd, tau);
            return;
        }
    }
    if constexpr (NJ == 7) {                              // A/B knob, the benchmark's instantiation only
        if (g_rne_wpb > 1 && mdh && allrev) {
            const dim3 g2((grid.x + g_rne_wpb - 1) / g_rne_wpb);
            if (g_rne_wpb == 2) hipLaunchKernelGGL((k_rne_wpb<NJ, true, true, 2>), g2, dim3(kW * 2), lds * 2, s, rp, links, q, qd, qdd, tau);
            else hipLaunchKernelGGL((k_rne_wpb<NJ, true, true, 4>), g2, dim3(kW * 4), lds * 4, s, rp, links, q, qd, qdd, tau);
            return;
        }
    }
    if (mdh && allrev) hipLaunchKernelGGL((k_rne<NJ, true, true>), grid, dim3(kW), lds, s, rp, links, q, qd, qdd, tau);
    else if (mdh) hipLaunchKernelGGL((k_rne<NJ, true, false>), grid, dim3(kW), lds, s, rp, links, q, qd, qdd, tau);
    else if (allrev) hipLaunchKernelGGL((k_rne<NJ, false, true>), grid, dim3(kW), lds, s, rp, links, q, qd, qdd, tau);
    else hipLaunchKernelGGL((k_rne<NJ, false, false>), grid, dim3(kW), lds, s, rp, links, q, qd, qdd, tau);
}
static void launch_rt(bool mdh, dim3 grid, size_t lds, hipStream_t s, const RneParams &rp, const DevLink *links,
                      const double *q, const double *qd, const double *qdd, double *tau)
{
    if (mdh) hipLaunchKernelGGL((k_rne_rt<true>), grid, dim3(kW), lds, s, rp, links, q, qd, qdd, tau);
    else hipLaunchKernelGGL((k_rne_rt<false>), grid, dim3(kW), lds, s, rp, links, q, qd, qdd, tau);
}

int launch_rne(const Dyn *d, const DevLink *links, const double *q, const double *qd, const double *qdd,
               int64_t N, const double *grav3, const double *fext6, double *tau, hipStream_t s, double *wbase)
{
    if (N == 0) return RTBHIP_OK;
    RneParams rp;
    rp.n = d->n;
    rp.has_fext = fext6 != nullptr;
    rp.N = N;
    rp.wbase = wbase;
    for (int i = 0; i < 3; i++) rp.grav[i] = grav3[i];
    for (int i = 0; i < 6; i++) rp.fext[i] = fext6 ? fext6[i] : 0.0;
    int stride = 3 * d->n;
    if (!(stride & 1)) stride += 1;
    const size_t lds = (size_t)kW * stride * sizeof(double);
    const int64_t tiles = (N + kW - 1) / kW;
    const bool mdh = d->mdh != 0;
    bool allrev = true;
    for (const DevLink &l : d->links) allrev = allrev && l.sigma == 0;
    const bool rt = d->n > 8 || tiles > 0x7fffffff || wbase != nullptr;
    int64_t g = rt ? (tiles + g_rne_tiles_per_wave - 1) / g_rne_tiles_per_wave : tiles;
    if (g > 0x7fffffff) g = 0x7fffffff;
    dim3 grid((unsigned)g);
    RneSig sig = (g_rne_sig && !rt) ? rne_signature(d->links.data(), d->n) : 0;
    // diagnostic (scripts/jit_diag.py): RTBHIP_RNE_SIG_AND=<hex> clears fields of the signature before the dispatch -- a kernel that KNOWS less about the
    // table is still a correct kernel for it (a cleared flag is a shortcut not taken, a cleared alpha class the general rotation ...)
    static const RneSig sig_and = std::getenv("RTBHIP_RNE_SIG_AND") ? std::strtoull(std::getenv("RTBHIP_RNE_SIG_AND"), nullptr, 16) : ~0ull;
    if (sig) sig = (sig & sig_and) | kRneSigPresent;
    switch (rt ? 0 : d->n) {
    case 1: launch_nj<1>(d, mdh, allrev, grid, lds, s, rp, links, q, qd, qdd, tau, sig); break;
    case 2: launch_nj<2>(d, mdh, allrev, grid, lds, s, rp, links, q, qd, qdd, tau, sig); break;
    case 3: launch_nj<3>(d, mdh, allrev, grid, lds, s, rp, links, q, qd, qdd, tau, sig); break;
    case 4: launch_nj<4>(d, mdh, allrev, grid, lds, s, rp, links, q, qd, qdd, tau, sig); break;
    case 5: launch_nj<5>(d, mdh, allrev, grid, lds, s, rp, links, q, qd, qdd, tau, sig); break;
    case 6: launch_nj<6>(d, mdh, allrev, grid, lds, s, rp, links, q, qd, qdd, tau, sig); break;
    case 7: launch_nj<7>(d, mdh, allrev, grid, lds, s, rp, links, q, qd, qdd, tau, sig); break;
    case 8: launch_nj<8>(d, mdh, allrev, grid, lds, s, rp, links, q, qd, qdd, tau, sig); break;
    default: launch_rt(mdh, grid, lds, s, rp, links, q, qd, qdd, tau); break;
    }
    note_launch((int)grid.x, kW, (int)lds);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "k_rne launch");
    return RTBHIP_OK;
}

#endif  // RTB_HOST_SIDE

}  // namespace rtbhip
