// tests/emu/emu_rne.cpp -- TEST INFRASTRUCTURE: Newton-Euler kernel body (and the entry point of the dynamics terms) replayed on the CPU.
#include "emu_dyn.h"

namespace rtbhip { int rne_sig_enabled(); }      // rne_kernels.hip: rtbhip_tune("rne_sig")
// SIG != 0: the instantiation launch_rne picks for a robot with that structure signature (rne_device.h: kRneSig*)
template <int NJ, bool MDH, RneSig SIG>
static void rne_run_sig(const Dyn *d, const double *q, const double *qd, const double *qdd, int64_t N, V3 g, V3 f, V3 nt, double *tau)
{
    const DevLink *links = d->links.data();
    for (int64_t s = 0; s < N; ++s) {
        const double *a = q + s * NJ, *b = qd ? qd + s * NJ : nullptr, *c = qdd ? qdd + s * NJ : nullptr;
        double *o = tau + s * NJ;
        auto qi = [&](int j) { return a[j]; };
        auto qdi = [&](int j) { return b ? b[j] : 0.0; };
        auto qddi = [&](int j) { return c ? c[j] : 0.0; };
        auto out = [&](int j, double v) { o[j] = v; };
        if (!qd) rne_lane<NJ, MDH, false, true, true, SIG>(links, NJ, g, f, nt, qi, qdi, qddi, out);
        else rne_lane<NJ, MDH, true, true, false, SIG>(links, NJ, g, f, nt, qi, qdi, qddi, out);
    }
}

template <int NJ>
static void rne_run(const Dyn *d, const double *q, const double *qd, const double *qdd, int64_t N, V3 g, V3 f, V3 nt,
                    double *tau)
{
    const DevLink *links = d->links.data();
    const int n = d->n;
    for (int64_t s = 0; s < N; ++s) {
        const double *a = q + s * n, *b = qd ? qd + s * n : nullptr, *c = qdd ? qdd + s * n : nullptr;
        double *o = tau + s * n;
        auto qi = [&](int j) { return a[j]; };
        auto qdi = [&](int j) { return b ? b[j] : 0.0; };
        auto qddi = [&](int j) { return c ? c[j] : 0.0; };
        auto out = [&](int j, double v) { o[j] = v; };
        bool allrev = true;
        for (const DevLink &l : d->links) allrev = allrev && l.sigma == 0;
        if constexpr (NJ > 0) {
            if (!qd && allrev) {             // the kernel launcher's choice for qd = NULL on an all-revolute chain: k_rne_atrest
                if (d->mdh) rne_lane<NJ, true, false, true, true>(links, n, g, f, nt, qi, qdi, qddi, out);
                else rne_lane<NJ, false, false, true, true>(links, n, g, f, nt, qi, qdi, qddi, out);
                continue;
            }
        }
        if (d->mdh) rne_lane<NJ, true>(links, n, g, f, nt, qi, qdi, qddi, out);
        else rne_lane<NJ, false>(links, n, g, f, nt, qi, qdi, qddi, out);
    }
}

// the structure signature rne_kernels.hip computes for a robot (rne_device.h: rne_signature), for the tests that pin the shipped models'
extern "C" unsigned long long emu_rne_signature(rtbhip_dyn_t h)
{
    const std::shared_ptr<Dyn> d_owner = dyn_from_handle(h);
    return d_owner ? rne_signature(d_owner->links.data(), d_owner->n) : 0ull;
}

extern "C" int emu_rne(rtbhip_dyn_t h, const double *q, const double *qd, const double *qdd, int64_t N,
                       const double *grav3, const double *fext6, double *tau, int force_generic)
{
    const std::shared_ptr<Dyn> d_owner = dyn_from_handle(h);
    Dyn *d = d_owner.get();
    if (!d) return -1;
    V3 g = v3(grav3[0], grav3[1], grav3[2]);
    V3 f = fext6 ? v3(fext6[0], fext6[1], fext6[2]) : v3(0, 0, 0);
    V3 nt = fext6 ? v3(fext6[3], fext6[4], fext6[5]) : v3(0, 0, 0);
    if (force_generic) { rne_run<0>(d, q, qd, qdd, N, g, f, nt, tau); return 0; }
    const RneSig sig = rtbhip::rne_sig_enabled() ? rne_signature(d->links.data(), d->n) : 0;
    if (sig == kRneSigPanda && d->mdh) { rne_run_sig<7, true, kRneSigPanda>(d, q, qd, qdd, N, g, f, nt, tau); return 0; }
    if (sig == kRneSigPuma560 && !d->mdh) { rne_run_sig<6, false, kRneSigPuma560>(d, q, qd, qdd, N, g, f, nt, tau); return 0; }
    switch (d->n) {
    case 1: rne_run<1>(d, q, qd, qdd, N, g, f, nt, tau); break;
    case 2: rne_run<2>(d, q, qd, qdd, N, g, f, nt, tau); break;
    case 3: rne_run<3>(d, q, qd, qdd, N, g, f, nt, tau); break;
    case 4: rne_run<4>(d, q, qd, qdd, N, g, f, nt, tau); break;
    case 5: rne_run<5>(d, q, qd, qdd, N, g, f, nt, tau); break;
    case 6: rne_run<6>(d, q, qd, qdd, N, g, f, nt, tau); break;
    case 7: rne_run<7>(d, q, qd, qdd, N, g, f, nt, tau); break;
    case 8: rne_run<8>(d, q, qd, qdd, N, g, f, nt, tau); break;
    default: rne_run<0>(d, q, qd, qdd, N, g, f, nt, tau); break;
    }
    return 0;
}

// the base wrench of rtbhip_rne_base_wrench: the run-time-n lane function with the wrench receiver, as k_rne_rt runs it
struct EmuWrench {
    static constexpr bool on = true;
    double *row;
    bool wanted() const { return row != nullptr; }
    void operator()(int k, double v) const { row[k] = v; }
};

extern "C" int emu_rne_base_wrench(rtbhip_dyn_t h, const double *q, const double *qd, const double *qdd, int64_t N,
                                   const double *grav3, const double *fext6, double *tau, double *wbase)
{
    const std::shared_ptr<Dyn> d_owner = dyn_from_handle(h);
    Dyn *d = d_owner.get();
    if (!d) return -1;
    V3 g = v3(grav3[0], grav3[1], grav3[2]);
    V3 f = fext6 ? v3(fext6[0], fext6[1], fext6[2]) : v3(0, 0, 0);
    V3 nt = fext6 ? v3(fext6[3], fext6[4], fext6[5]) : v3(0, 0, 0);
    const DevLink *links = d->links.data();
    const int n = d->n;
    for (int64_t s = 0; s < N; ++s) {
        const double *a = q + s * n, *b = qd ? qd + s * n : nullptr, *c = qdd ? qdd + s * n : nullptr;
        double *o = tau + s * n;
        auto qi = [&](int j) { return a[j]; };
        auto qdi = [&](int j) { return b ? b[j] : 0.0; };
        auto qddi = [&](int j) { return c ? c[j] : 0.0; };
        auto out = [&](int j, double v) { o[j] = v; };
        if (d->mdh) rne_lane<0, true, true, false>(links, n, g, f, nt, qi, qdi, qddi, out, EmuWrench{wbase + s * 6});
        else rne_lane<0, false, true, false>(links, n, g, f, nt, qi, qdi, qddi, out, EmuWrench{wbase + s * 6});
    }
    return 0;
}

extern "C" int emu_dyn(rtbhip_dyn_t h, int mode, const double *q, const double *qd, const double *tq, int64_t N,
                       const double *grav3, double *out)
{
    const std::shared_ptr<Dyn> d_owner = dyn_from_handle(h);
    Dyn *d = d_owner.get();
    if (!d || d->n > 16) return -1;
    V3 g = grav3 ? v3(grav3[0], grav3[1], grav3[2]) : v3(0, 0, 0);
    const int n = d->n;
    if (n <= 7) return emu_dyn_r1(d, mode, q, qd, tq, N, g, out);
    if (n <= 10) return emu_dyn_r2(d, mode, q, qd, tq, N, g, out);
    if (n <= 12) return emu_dyn_r3(d, mode, q, qd, tq, N, g, out);
    if (n <= 14) return emu_dyn_r4(d, mode, q, qd, tq, N, g, out);
    if (n <= 15) return emu_dyn_r5(d, mode, q, qd, tq, N, g, out);
    return emu_dyn_r6(d, mode, q, qd, tq, N, g, out);
}
