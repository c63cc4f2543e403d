// tests/emu/emu.cpp -- TEST INFRASTRUCTURE: executes the kernels' own per-lane phase functions
// (robotics-toolbox-python_amd/csrc/kin_tile.h, rne_device.h, ik_device.h -- all __host__ __device__)
// lane by lane on the CPU, in the same phase order and with the same LDS layout as the gfx950
// kernels.  The build container has no GPU; this lets `pytest -m "not gpu"` catch logic errors in the
// kernel bodies (indexing, staging, flush arithmetic, recursion order) before GPU minutes are spent.
// It is NOT a product path: librtbhip.so contains none of this and fails loudly without a GPU.
#include "../../robotics-toolbox-python_amd/csrc/ik_device.h"
#include "../../robotics-toolbox-python_amd/csrc/rne_device.h"
#include "../../robotics-toolbox-python_amd/csrc/dyn_device.h"
#include "../../robotics-toolbox-python_amd/csrc/diff_device.h"
#include "../../robotics-toolbox-python_amd/csrc/tree_device.h"
#include "../../robotics-toolbox-python_amd/csrc/partial_device.h"
#include "../../robotics-toolbox-python_amd/csrc/frames_device.h"
#include "../../robotics-toolbox-python_amd/csrc/servo_device.h"
#include <vector>
#include <cstdio>
#include <cstdlib>

using namespace rtbhip;

static Affine aff16(const double *m)
{
    Affine a;
    a.used = m != nullptr;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) a.v[4 * r + c] = m ? m[4 * r + c] : (r == c ? 1.0 : 0.0);
    return a;
}

extern "C" int emu_kin(rtbhip_chain_t h, const double *q, int64_t N, const double *base16, const double *tool16,
                       int frame, double *T, double *J, double *H, int coalesced)
{
    const std::shared_ptr<Chain> c_owner = chain_from_handle(h);
    Chain *c = c_owner.get();
    if (!c) return -1;
    KinParams kp;
    kp.n = c->n; kp.qw = c->q_width; kp.stride = kin_stride(c->n);
    kp.frame = frame; kp.N = N; kp.pad = 0;
    Affine b = aff16(base16), t = aff16(tool16);
    kp.has_base = b.used;
    for (int i = 0; i < 12; i++) kp.base[i] = b.v[i];
    chain_tail(c, t, kp.tail);
    std::vector<double> lds(kin_lds_bytes(kp.n, kp.qw) / sizeof(double), -777.0);
    double *rows = lds.data(), *qs = lds.data() + kWave * kp.stride;
    const DevChain ops = chain_host_view(c);
    const int W = 6 * kp.n;
    const int64_t tiles = (N + kWave - 1) / kWave;
    for (int64_t tile = 0; tile < tiles; ++tile) {
        const int64_t cfg0 = tile * kWave;
        const int ncfg = (int)std::min<int64_t>(kWave, N - cfg0);
        Pose P[kWave];
        for (int l = 0; l < kWave; ++l) kin_load_q(kp, q, cfg0 + l, l, qs);
        for (int l = 0; l < kWave; ++l) {
            if (J || H) kin_walk<true>(kp, ops, l, qs, rows, P[l]);
            else kin_walk<false>(kp, ops, l, qs, rows, P[l]);
        }
        if (H) for (int l = 0; l < kWave; ++l) kin_hessian(kp, l, rows, l < ncfg, H + (cfg0 + l) * (int64_t)(kp.n * W));
        if (J) for (int l = 0; l < kWave; ++l) {
            if (coalesced) kin_flush(rows, kp.stride, W, ncfg, J + cfg0 * W, l);
            else kin_store_own(rows, kp.stride, W, l < ncfg, J + (cfg0 + l) * W, l);
        }
        if (T) {
            for (int l = 0; l < kWave; ++l) kin_stage_T(kp, l, rows, P[l]);
            for (int l = 0; l < kWave; ++l) {
                if (coalesced) kin_flush(rows, kp.stride, 16, ncfg, T + cfg0 * 16, l);
                else kin_store_own(rows, kp.stride, 16, l < ncfg, T + (cfg0 + l) * 16, l);
            }
        }
    }
    return 0;
}

template <int NJ>
static void emu_reg_run(const KinParams &kp, const DevChain &cv, const double *q, int64_t N, double *T, double *J)
{
    std::vector<double> buf(reg_lds_doubles(NJ), -777.0);
    constexpr int W = 6 * NJ;
    const int64_t tiles = (N + kWave - 1) / kWave;
    for (int64_t tile = 0; tile < tiles; ++tile) {
        const int64_t cfg0 = tile * kWave;
        const int ncfg = (int)std::min<int64_t>(kWave, N - cfg0);
        Pose P[kWave];
        double jac[kWave][6 * NJ];
        for (int l = 0; l < kWave; ++l) {
            if (J) reg_compute<NJ, true>(kp, cv, q, cfg0 + l, P[l], jac[l]);
            else reg_compute<NJ, false>(kp, cv, q, cfg0 + l, P[l], jac[l]);
        }
        if (J)
            for (int r = 0; r < kWave / kJRound; ++r) {
                for (int l = 0; l < kWave; ++l)
                    if (l / kJRound == r) reg_stage_J<NJ>(jac[l], buf.data(), l % kJRound);
                int rows = std::max(0, std::min(kJRound, ncfg - r * kJRound));
                for (int l = 0; l < kWave; ++l) kin_flush(buf.data(), W + 1, W, rows, J + (cfg0 + r * kJRound) * W, l);
            }
        if (T) {
            for (int l = 0; l < kWave; ++l) reg_stage_T(kp, P[l], buf.data(), l);
            for (int l = 0; l < kWave; ++l) kin_flush(buf.data(), 17, 16, ncfg, T + cfg0 * 16, l);
        }
    }
}

extern "C" int emu_kin_reg(rtbhip_chain_t h, const double *q, int64_t N, const double *base16, const double *tool16,
                           int frame, double *T, double *J)
{
    const std::shared_ptr<Chain> c_owner = chain_from_handle(h);
    Chain *c = c_owner.get();
    if (!c || c->n < 1 || c->n > kKinRegMax) return -1;
    KinParams kp;
    kp.n = c->n; kp.qw = c->q_width; kp.stride = kin_stride(c->n);
    kp.frame = frame; kp.N = N; kp.pad = 0;
    Affine b = aff16(base16), t = aff16(tool16);
    kp.has_base = b.used;
    for (int i = 0; i < 12; i++) kp.base[i] = b.v[i];
    chain_tail(c, t, kp.tail);
    const DevChain cv = chain_host_view(c);
    switch (c->n) {
    case 1: emu_reg_run<1>(kp, cv, q, N, T, J); break;
    case 2: emu_reg_run<2>(kp, cv, q, N, T, J); break;
    case 3: emu_reg_run<3>(kp, cv, q, N, T, J); break;
    case 4: emu_reg_run<4>(kp, cv, q, N, T, J); break;
    case 5: emu_reg_run<5>(kp, cv, q, N, T, J); break;
    case 6: emu_reg_run<6>(kp, cv, q, N, T, J); break;
    case 7: emu_reg_run<7>(kp, cv, q, N, T, J); break;
    case 8: emu_reg_run<8>(kp, cv, q, N, T, J); break;
    case 9: emu_reg_run<9>(kp, cv, q, N, T, J); break;
    default: emu_reg_run<10>(kp, cv, q, N, T, J); break;
    }
    return 0;
}

template <int NJ>
static void emu_hess_run(const KinParams &kp, const DevChain &cv, const double *q, int64_t N, double *H)
{
    constexpr int W = 6 * NJ;
    std::vector<double> buf(kWave * (W + 1), -777.0);
    const int64_t tiles = (N + kWave - 1) / kWave;
    for (int64_t tile = 0; tile < tiles; ++tile) {
        const int64_t cfg0 = tile * kWave;
        const int ncfg = (int)std::min<int64_t>(kWave, N - cfg0);
        for (int l = 0; l < kWave; ++l) {
            Pose P;
            double jac[6 * NJ];
            reg_compute<NJ, true>(kp, cv, q, cfg0 + l, P, jac);
            for (int k = 0; k < W; ++k) buf[l * (W + 1) + k] = jac[k];
        }
        double *dst = H + cfg0 * (int64_t)(NJ * W);
        for (int l = 0; l < kWave; ++l)
            hessian_run<NJ>(buf.data(), W + 1, ncfg, l, [&](int f, double a, double b, bool both) { dst[f] = a; if (both) dst[f + 1] = b; });
    }
}

// k_kin_hess_tile<NJ, R>: rounds of 64/R lanes expanding their Hessians into the tile, whole-wave flush_rows
template <int NJ, int R>
static void emu_hess_tile_run(const KinParams &kp, const DevChain &cv, const double *q, int64_t N, double *H)
{
    constexpr int HW = NJ * 6 * NJ, S = HW | 1, G = kWave / R;
    std::vector<double> buf((size_t)G * S, -777.0);
    std::vector<double> jacs((size_t)kWave * 6 * NJ);
    const int64_t tiles = (N + kWave - 1) / kWave;
    for (int64_t tile = 0; tile < tiles; ++tile) {
        const int64_t cfg0 = tile * kWave;
        const int ncfg = (int)std::min<int64_t>(kWave, N - cfg0);
        for (int l = 0; l < kWave; ++l) {
            Pose P;
            double jac[6 * NJ];
            reg_compute<NJ, true>(kp, cv, q, cfg0 + l, P, jac);
            for (int k = 0; k < 6 * NJ; ++k) jacs[(size_t)l * 6 * NJ + k] = jac[k];
        }
        for (int r = 0; r < R; ++r) {
            const int cnt = std::min(G, ncfg - r * G);
            if (cnt <= 0) break;
            for (int l = r * G; l < (r + 1) * G; ++l) {
                const double *jac = &jacs[(size_t)l * 6 * NJ];
                double *mine = buf.data() + (size_t)(l - r * G) * S;
                hessian_from_jacobian(NJ, [&](int k) { return jac[k]; }, [&](int idx, double v) { mine[idx] = v; });
            }
            double *dst = H + (cfg0 + r * G) * (int64_t)HW;
            for (int l = 0; l < kWave; ++l)
                flush_rows<HW>(buf.data(), S, cnt, l, [&](int f, double a, double b) { dst[f] = a; dst[f + 1] = b; });
        }
    }
}

extern "C" int emu_kin_hess_tile(rtbhip_chain_t h, const double *q, int64_t N, const double *tool16, int frame, int rounds, double *H)
{
    const std::shared_ptr<Chain> c_owner = chain_from_handle(h);
    Chain *c = c_owner.get();
    if (!c || c->n < 1 || c->n > kKinRegMax) return -1;
    KinParams kp;
    kp.n = c->n; kp.qw = c->q_width; kp.stride = kin_stride(c->n); kp.frame = frame; kp.N = N; kp.pad = 0; kp.has_base = 0;
    Affine t = aff16(tool16);
    chain_tail(c, t, kp.tail);
    const DevChain cv = chain_host_view(c);
#define RTB_HT(NJ) case NJ: if (rounds == 4) emu_hess_tile_run<NJ, 4>(kp, cv, q, N, H); else if (rounds == 16) emu_hess_tile_run<NJ, 16>(kp, cv, q, N, H); else emu_hess_tile_run<NJ, 8>(kp, cv, q, N, H); break;
    switch (c->n) {
    RTB_HT(1) RTB_HT(2) RTB_HT(3) RTB_HT(4) RTB_HT(5) RTB_HT(6) RTB_HT(7) RTB_HT(8) RTB_HT(9)
    default: if (rounds == 4) emu_hess_tile_run<10, 4>(kp, cv, q, N, H); else if (rounds == 16) emu_hess_tile_run<10, 16>(kp, cv, q, N, H); else emu_hess_tile_run<10, 8>(kp, cv, q, N, H); break;
    }
#undef RTB_HT
    return 0;
}

extern "C" int emu_kin_hess(rtbhip_chain_t h, const double *q, int64_t N, const double *tool16, int frame, double *H)
{
    const std::shared_ptr<Chain> c_owner = chain_from_handle(h);
    Chain *c = c_owner.get();
    if (!c || c->n < 1 || c->n > kKinRegMax) return -1;
    KinParams kp;
    kp.n = c->n; kp.qw = c->q_width; kp.stride = kin_stride(c->n); kp.frame = frame; kp.N = N; kp.pad = 0; kp.has_base = 0;
    Affine t = aff16(tool16);
    chain_tail(c, t, kp.tail);
    const DevChain cv = chain_host_view(c);
    switch (c->n) {
    case 1: emu_hess_run<1>(kp, cv, q, N, H); break;
    case 2: emu_hess_run<2>(kp, cv, q, N, H); break;
    case 3: emu_hess_run<3>(kp, cv, q, N, H); break;
    case 4: emu_hess_run<4>(kp, cv, q, N, H); break;
    case 5: emu_hess_run<5>(kp, cv, q, N, H); break;
    case 6: emu_hess_run<6>(kp, cv, q, N, H); break;
    case 7: emu_hess_run<7>(kp, cv, q, N, H); break;
    case 8: emu_hess_run<8>(kp, cv, q, N, H); break;
    case 9: emu_hess_run<9>(kp, cv, q, N, H); break;
    default: emu_hess_run<10>(kp, cv, q, N, H); break;
    }
    return 0;
}

// k_hess_from_jac<NJ, 4>: the tile's Jacobians through the LDS staging (hj_load_tile), each lane's into "registers", then the
// tile emission of k_kin_hess_tile
template <int NJ>
static void emu_hess_from_jac_run(const double *J, int64_t N, double *H)
{
    constexpr int R = 4, W = 6 * NJ, HW = NJ * W, S = HW | 1, G = kWave / R;
    std::vector<double> buf(std::max<size_t>((size_t)G * S, (size_t)kWave * (W + 1)), -777.0);
    std::vector<double> jacs((size_t)kWave * W);
    const int64_t tiles = (N + kWave - 1) / kWave;
    for (int64_t tile = 0; tile < tiles; ++tile) {
        const int64_t cfg0 = tile * kWave;
        const int ncfg = (int)std::min<int64_t>(kWave, N - cfg0);
        for (int l = 0; l < kWave; ++l) hj_load_tile(J + cfg0 * W, W, ncfg, buf.data(), l);
        for (int l = 0; l < kWave; ++l)
            for (int k = 0; k < W; ++k) jacs[(size_t)l * W + k] = l < ncfg ? buf[(size_t)l * (W + 1) + k] : 0.0;
        for (int r = 0; r < R; ++r) {
            const int cnt = std::min(G, ncfg - r * G);
            if (cnt <= 0) break;
            for (int l = r * G; l < (r + 1) * G; ++l) {
                const double *jac = &jacs[(size_t)l * W];
                double *mine = buf.data() + (size_t)(l - r * G) * S;
                hessian_from_jacobian(NJ, [&](int k) { return jac[k]; }, [&](int idx, double v) { mine[idx] = v; });
            }
            double *dst = H + (cfg0 + r * G) * (int64_t)HW;
            for (int l = 0; l < kWave; ++l)
                flush_rows<HW>(buf.data(), S, cnt, l, [&](int f, double a, double b) { dst[f] = a; dst[f + 1] = b; });
        }
    }
}

extern "C" int emu_hess_from_jac(const double *J, int64_t N, int n, double *H)
{
    switch (n) {
    case 1: emu_hess_from_jac_run<1>(J, N, H); break;
    case 2: emu_hess_from_jac_run<2>(J, N, H); break;
    case 3: emu_hess_from_jac_run<3>(J, N, H); break;
    case 4: emu_hess_from_jac_run<4>(J, N, H); break;
    case 5: emu_hess_from_jac_run<5>(J, N, H); break;
    case 6: emu_hess_from_jac_run<6>(J, N, H); break;
    case 7: emu_hess_from_jac_run<7>(J, N, H); break;
    case 8: emu_hess_from_jac_run<8>(J, N, H); break;
    case 9: emu_hess_from_jac_run<9>(J, N, H); break;
    case 10: emu_hess_from_jac_run<10>(J, N, H); break;
    default:                                              // k_hess_from_jac_any: one lane per Jacobian
        for (int64_t i = 0; i < N; ++i) {
            const double *Jr = J + i * (int64_t)(6 * n);
            double *Hr = H + i * (int64_t)(6 * n * n);
            hessian_from_jacobian(n, [&](int k) { return Jr[k]; }, [&](int idx, double v) { Hr[idx] = v; });
        }
    }
    return 0;
}

// k_angle_axis: both operand tiles through LDS (aa_load_tile), per-lane aa_lane, staged e rows flushed as one run
extern "C" int emu_angle_axis(const double *Te, int64_t nTe, const double *Tep, int64_t nTep, double *e)
{
    const int64_t N = std::max(nTe, nTep);
    std::vector<double> a(kWave * kAaStride, -777.0), b(kWave * kAaStride, -777.0);
    const int64_t tiles = (N + kWave - 1) / kWave;
    for (int64_t tile = 0; tile < tiles; ++tile) {
        const int64_t cfg0 = tile * kWave;
        const int ncfg = (int)std::min<int64_t>(kWave, N - cfg0);
        const bool ea = nTe == N, eb = nTep == N;
        for (int l = 0; l < kWave; ++l) {
            if (ea) aa_load_tile(Te + cfg0 * 16, ncfg, a.data(), l); else aa_load_tile(Te, 1, a.data(), l);
            if (eb) aa_load_tile(Tep + cfg0 * 16, ncfg, b.data(), l); else aa_load_tile(Tep, 1, b.data(), l);
        }
        double t1[kWave][12], t2[kWave][12];
        for (int l = 0; l < kWave; ++l) {
            const int la = ea ? (l < ncfg ? l : 0) : 0, lb = eb ? (l < ncfg ? l : 0) : 0;
            for (int k = 0; k < 12; ++k) { t1[l][k] = a[la * kAaStride + k]; t2[l][k] = b[lb * kAaStride + k]; }
        }
        for (int l = 0; l < kWave; ++l) aa_lane(t1[l], t2[l], a.data() + l * 7);
        for (int l = 0; l < kWave; ++l) kin_flush(a.data(), 7, 6, ncfg, e + cfg0 * 6, l);
    }
    return 0;
}

extern "C" unsigned emu_xcd_tile(unsigned g, unsigned b) { return xcd_tile_of(g, b); }

extern "C" void emu_sincos(const double *x, int64_t n, double *s, double *c, int reduced_only)
{
    for (int64_t i = 0; i < n; ++i) {
        if (reduced_only) sincos_reduced(x[i], s[i], c[i]);
        else rtb_sincos(x[i], &s[i], &c[i]);
    }
}

template <int NJ>
static void emu_ik_run(const Chain *c, const IkDev &p, const double *Tep, const double *q0, double *q_out, int32_t *success,
                       int32_t *iters, int32_t *searches, double *residual)
{
    const DevChain cv = chain_host_view(c);
    const double *qlim = c->qlim.data();
    for (int64_t t = 0; t < p.N; ++t)   // the specification: searches one after another
        ik_solve_sequential<NJ>(p, cv, qlim, t, Tep, q0, q_out, success, iters, searches, residual);
}

// Replays k_ik's wave-level driver (ik_kernels.hip) on the CPU: `waves` single-wave workgroups advanced
// round-robin, one scheduling pass + one LM iteration each per turn, sharing the fresh-target counter.
template <int NJ>
struct EmuWave {
    IkWaveSharedT<kIkMaxJoints> sh;
    IkLane<NJ> st[kWave];
    unsigned long long busy = 0;
    bool exhausted = false, first = true, done = false, drained = false;
    unsigned long long pool_next = 0, pool_end = 0;
    unsigned tick = 0;
    long long passes = 0, iters = 0, lane_iters_useful = 0;
    long long quiet = 0;       // the kernel's watchdog counter, replayed: a false fire fails the run (-4)
    unsigned long long pend_item = kIkNoItem;   // sharing: a range handed to this wave (row N + pend_tick), started at its next pass
    unsigned pend_tick = 0;
    bool waiting = false;      // sharing: holds ticket pend_tick and polls its word
};

template <int NJ>
static int emu_ik_wave_run(const Chain *c, const IkDev &p, int waves, const double *Tep, const double *q0, double *q_out,
                           int32_t *success, int32_t *iters, int32_t *searches, double *residual, double *stats,
                           const IkWork *work = nullptr, const IkShareCtl *share = nullptr)
{
    const DevChain cv = chain_host_view(c);
    const double *qlim = c->qlim.data();
    const int s_last = ik_s_last(p);
    unsigned long long counter = 0;
    std::vector<EmuWave<NJ>> W(waves);
    for (auto &w : W)
        for (int l = 0; l < kWave; ++l) {
            IkLane<NJ> &st = w.st[l];
            st.status = kIkIdle; st.E = 0; st.iter = 0; st.s = 0; st.slot = 0; st.fin = 0; st.ok = 0;
            for (int j = 0; j < NJ; ++j) w.sh.q[j][l] = 0.0;
            for (int k = 0; k < 12; ++k) w.sh.Td[k][l] = (k == 0 || k == 4 || k == 8) ? 1.0 : 0.0;
        }
    auto ballot = [&](EmuWave<NJ> &w, auto pred) { unsigned long long m = 0; for (int l = 0; l < kWave; ++l) if (pred(l)) m |= 1ull << l; return m; };
    int live = waves;
    long long guard = 0;
    while (live > 0) {
        if (++guard > 400000) {
            if (getenv("EMU_IK_DEBUG"))
                for (size_t wi = 0; wi < W.size(); ++wi) {
                    auto &w = W[wi];
                    if (w.done) continue;
                    fprintf(stderr, "wave %zu busy=%llx exhausted=%d counter=%llu\n", wi, w.busy, (int)w.exhausted, counter);
                    for (int l = 0; l < kWave; ++l)
                        if ((w.busy >> l) & 1ull)
                            fprintf(stderr, "  slot %d item=%lld b=%d next=%d best=%d it=%d res=%d\n", l, (long long)w.sh.vix[l], w.sh.b[l], w.sh.next[l], w.sh.best[l], w.sh.it[l], w.sh.res[l]);
                    for (int l = 0; l < kWave; ++l)
                        fprintf(stderr, "  lane %d status=%d slot=%d s=%d iter=%d fin=%d\n", l, w.st[l].status, w.st[l].slot, w.st[l].s, w.st[l].iter, w.st[l].fin);
                }
            return -2;
        }
        for (size_t wi = 0; wi < W.size(); ++wi) {
            auto &w = W[wi];
            if (w.done) continue;
            bool anyfin = false;
            for (int l = 0; l < kWave; ++l) anyfin = anyfin || w.st[l].fin != 0;
            if (w.first || ((w.tick++ & p.pass_mask) == 0 && anyfin)) {
                w.first = false;
                w.passes++;
                for (int l = 0; l < kWave; ++l) ik_report<NJ>(w.st[l], w.sh, residual, p, qlim, ik_lds_q(w.sh, l));
                for (int l = 0; l < kWave; ++l) if ((w.busy >> l) & 1ull) ik_account(l, w.sh);
                for (int l = 0; l < kWave; ++l) ik_finalize<NJ>(w.st[l], w.sh, l, p, qlim, q_out, success, iters, searches, residual);
                const unsigned long long freed = ballot(w, [&](int l) { return ((w.busy >> l) & 1ull) && w.sh.res[l] != 0; });
                if (freed) w.quiet = 0;
                w.busy &= ~freed;
                unsigned long long idle = ballot(w, [&](int l) { return w.st[l].status == kIkIdle; });
                const unsigned long long starved = ballot(w, [&](int l) { return ((w.busy >> l) & 1ull) && ik_starved(l, w.sh); });
                if (starved) {
                    for (int l = 0; l < kWave; ++l) if ((starved >> l) & 1ull) w.sh.list[ik_rank(starved, l)] = l;
                    const int ns = __builtin_popcountll(starved);
                    if (__builtin_popcountll(idle) < ns) return -3;      // cannot happen: every newly starved slot just released a lane
                    for (int l = 0; l < kWave; ++l) {
                        const int r = ik_rank(idle, l);
                        if (((idle >> l) & 1ull) && r < ns) {
                            const int slot = w.sh.list[r];
                            ik_start_spec<NJ>(w.st[l], w.sh, l, p, qlim, slot, w.sh.next[slot], Tep, q0);
                        }
                    }
                    idle = ballot(w, [&](int l) { return w.st[l].status == kIkIdle; });
                }
                if ((!w.exhausted || w.pend_item != kIkNoItem) && idle) {
                    const unsigned long long freeslots = ~w.busy;
                    int nf = __builtin_popcountll(idle);
                    nf = nf > p.fresh_cap ? p.fresh_cap : nf;
                    { static const int mb = getenv("EMU_IK_MAX_BUSY") ? atoi(getenv("EMU_IK_MAX_BUSY")) : 64;
                      const int room = mb - __builtin_popcountll(w.busy); nf = nf > room ? (room > 0 ? room : 0) : nf; }
                    unsigned long long base = 0;
                    long long nvalid = 0;
                    if (!w.exhausted) {
                    if (w.pool_next == w.pool_end) {
                        const unsigned long long chunk = p.pool_chunk > 0 ? (unsigned long long)p.pool_chunk : (unsigned long long)nf;
                        const unsigned long long got = counter;
                        counter += chunk;
                        const unsigned long long NN = (unsigned long long)p.N;
                        w.pool_next = got < NN ? got : NN;
                        w.pool_end = got + chunk < NN ? got + chunk : NN;
                        if (w.pool_end == NN) w.drained = true;
                    }
                    base = w.pool_next;
                    nvalid = (long long)(w.pool_end - w.pool_next);
                    nvalid = nvalid > nf ? nf : nvalid;
                    w.pool_next += (unsigned long long)nvalid;
                    if (w.drained && w.pool_next == w.pool_end) w.exhausted = true;
                    }
                    const IkWork pend = ik_unpack(w.pend_item);
                    if (w.exhausted && w.pend_item != kIkNoItem) {
                        base = (unsigned long long)ik_item_row(*share, p.N, (int)(wi % kIkQueues), w.pend_tick); nvalid = 1;
                        w.pend_item = kIkNoItem;
                    }
                    for (int l = 0; l < kWave; ++l) if ((freeslots >> l) & 1ull) w.sh.list[ik_rank(freeslots, l)] = l;
                    for (int l = 0; l < kWave; ++l) {
                        const int r = ik_rank(idle, l);
                        if (((idle >> l) & 1ull) && r < nvalid) {
                            const int64_t v = (int64_t)base + r;
                            IkWork it;
                            if (share && v >= p.N) it = pend;
                            else if (work) it = work[v];
                            else { it.tgt = (int32_t)v; it.s0 = (int16_t)ik_s_first(p); it.s1 = (int16_t)ik_s_last(p); }
                            ik_start_target<NJ>(w.st[l], w.sh, l, p, qlim, w.sh.list[r], v, it, Tep, q0);
                        }
                    }
                    w.busy |= ballot(w, [&](int l) { return ((freeslots >> l) & 1ull) && ik_rank(freeslots, l) < nvalid; });
                    idle = ballot(w, [&](int l) { return w.st[l].status == kIkIdle; });
                }
                if (idle && w.busy) {
                    for (int l = 0; l < kWave; ++l) if ((w.busy >> l) & 1ull) w.sh.list[ik_rank(w.busy, l)] = l;
                    const int nb = __builtin_popcountll(w.busy);
                    int slot[kWave], ss[kWave];
                    bool mine[kWave];
                    for (int l = 0; l < kWave; ++l)
                        mine[l] = ((idle >> l) & 1ull) && ik_pick(w.sh, ik_rank(idle, l), nb, __builtin_popcountll(idle), p.spec_policy, ik_s_first(p), slot[l], ss[l]);
                    for (int l = 0; l < kWave; ++l) if (mine[l]) ik_start_spec<NJ>(w.st[l], w.sh, l, p, qlim, slot[l], ss[l], Tep, q0);
                }
                if (share && w.exhausted && w.busy) {                            // phase D3: give work to waiting waves
                    const unsigned long long cand = ballot(w, [&](int l) { return ((w.busy >> l) & 1ull) && ik_donatable(w.sh, l, ik_s_first(p), (int)share->after); });
                    if (cand) {
                        unsigned long long x[kIkQueues], open = 0;
                        for (int g = 0; g < kIkQueues; ++g) {
                            x[g] = ik_aload(ik_queue_word(*share, g));
                            if (ik_word_waiting(x[g]) > 0 && ik_word_count(x[g]) < share->qlimit) open |= 1ull << g;
                        }
                        if (open) {
                            const int start = (int)((wi + w.tick) % kIkQueues);
                            const unsigned long long rot = ((open >> start) | (open << (kIkQueues - start))) & ((1ull << kIkQueues) - 1ull);
                            const int g = (start + __builtin_ctzll(rot)) % kIkQueues;
                            unsigned give = ik_word_waiting(x[g]);
                            const unsigned nc = (unsigned)__builtin_popcountll(cand);
                            give = give > nc ? nc : give;
                            give = give > (unsigned)kIkGiveMax ? (unsigned)kIkGiveMax : give;
                            const unsigned k0 = ik_word_count(ik_aadd(ik_queue_word(*share, g), (unsigned long long)give));
                            unsigned i = 0;
                            for (unsigned long long m = cand; m && i < give; m &= m - 1ull, ++i) ik_donate(*share, p.N, w.sh, __builtin_ctzll(m), g, k0 + i);
                        }
                    }
                }
            }
            if (share && w.busy == 0 && w.exhausted && w.pend_item == kIkNoItem) {
                // the kernel's wait loop, one look per turn: ticket first, then this ticket's own word
                const int g = (int)(wi % kIkQueues);
                if (!w.waiting) {
                    w.pend_tick = ik_ticket(*share, g);
                    unsigned long long x1[kIkQueues];
                    unsigned sum = 0;
                    for (int q = 0; q < kIkQueues; ++q) { x1[q] = ik_aload(ik_queue_word(*share, q)); sum += ik_word_waiting(x1[q]); }
                    if (sum == share->waves) {        // (single-threaded replay: the second read cannot differ)
                        for (int q = 0; q < kIkQueues; ++q)
                            for (int l = 0; l < kWave; ++l) ik_release_queue(*share, q, x1[q], l);
                        w.done = true; --live; continue;
                    }
                    w.waiting = true;
                }
                const unsigned long long x = share->wdyn[(size_t)g * share->qcap + w.pend_tick];
                if (x == kIkNoItem) continue;                                  // keep waiting
                w.waiting = false;
                if (x == kIkExitItem) { w.done = true; --live; continue; }
                w.pend_item = x; w.first = true;                               // a pass at the next turn starts it
                continue;
            }
            if (w.busy == 0 && w.exhausted) { w.done = true; --live; continue; }
            if (++w.quiet > ik_patience(p, s_last)) return -4;   // the kernel would overwrite valid results with its NaN markers here
            w.iters++;
            for (int l = 0; l < kWave; ++l) {
                if (w.st[l].status == kIkRun) w.lane_iters_useful++;
                static const bool emu_ik_status = getenv("EMU_IK_STATUS") != nullptr; if (emu_ik_status) { static long long cnt[4] = {0, 0, 0, 0}; static long long total = 0; cnt[w.st[l].status]++;
                    if ((++total % 4000000) == 0) fprintf(stderr, "status idle %lld run %lld parkedok %lld parkedlast %lld\n", cnt[0], cnt[1], cnt[2], cnt[3]); }
                ik_iter_any<NJ>(w.st[l], p, cv, qlim, [&](int k) { return w.sh.Td[k][w.st[l].slot]; }, ik_lds_q(w.sh, l));
            }
        }
    }
    if (stats) {
        long long mx = 0, tot = 0, useful = 0, passes = 0;
        for (auto &w : W) { mx = std::max(mx, w.iters); tot += w.iters; useful += w.lane_iters_useful; passes += w.passes; }
        stats[0] = (double)mx; stats[1] = (double)tot; stats[2] = (double)useful; stats[3] = (double)passes;
    }
    return 0;
}

// launch_ik's phased schedule (ik_kernels.hip) replayed with the same planning / item / merge functions of ik_device.h: phase A in
// plain mode into the final arrays, then the work lists of phases B and C through the wave scheduler, merged in search order.
template <int NJ>
static int emu_ik_phased_run(const Chain *c, const IkDev &p, int waves, const double *Tep, const double *q0, double *q_out,
                             int32_t *success, int32_t *iters, int32_t *searches, double *residual, double *stats)
{
    const int n = NJ;
    const IkPhases ph = ik_phases(p);
    if (ph.b_last <= ph.a_last) return emu_ik_wave_run<NJ>(c, p, waves, Tep, q0, q_out, success, iters, searches, residual, stats);
    double st3[3][4] = {{0}};
    IkDev pa = p;
    pa.slimit = p.flavour == 0 ? ph.a_last : ph.a_last + 1;
    int rc = emu_ik_wave_run<NJ>(c, pa, waves, Tep, q0, q_out, success, iters, searches, residual, st3[0]);
    if (rc) return rc;
    std::vector<IkWork> wB;
    for (int64_t t = p.N - 1; t >= 0; --t)                 // any order will do (the device's is whatever the atomics give): reversed here
        if (!success[t]) wB.push_back(ik_item_b(ph, t));
    auto run_items = [&](const std::vector<IkWork> &w, std::vector<double> &vq, std::vector<int32_t> &vok, std::vector<int32_t> &vit,
                         std::vector<int32_t> &vse, std::vector<double> &vE, double *stats_out) {
        const size_t m = w.size();
        vq.assign(m * n + 1, 0.0); vok.assign(m + 1, 0); vit.assign(m + 1, 0); vse.assign(m + 1, 0); vE.assign(m + 1, 0.0);
        if (!m) return 0;
        IkDev pi = p;
        pi.N = (int64_t)m;
        const int64_t cap = ((int64_t)m + waves - 1) / waves;
        pi.fresh_cap = cap > 64 ? 64 : (cap < 1 ? 1 : (int)cap);
        pi.pool_chunk = 0;
        return emu_ik_wave_run<NJ>(c, pi, waves, Tep, q0, vq.data(), vok.data(), vit.data(), vse.data(), vE.data(), stats_out, w.data());
    };
    std::vector<double> vq, vE; std::vector<int32_t> vok, vit, vse;
    rc = run_items(wB, vq, vok, vit, vse, vE, st3[1]);
    if (rc) return rc;
    std::vector<IkWork> wC; std::vector<int32_t> own;
    for (size_t v = 0; v < wB.size(); ++v) {
        const int64_t tgt = wB[v].tgt;
        if (ik_merge_item<0>(n, ph.c_chunks == 0, tgt, (int64_t)v, vq.data(), vok.data(), vit.data(), vse.data(), vE.data(), q_out, success, iters, searches, residual)) continue;
        own.push_back((int32_t)tgt);
        for (int k = 0; k < ph.c_chunks; ++k) wC.push_back(ik_item_c(ph, tgt, k));
    }
    if (ph.c_chunks > 0) {
        rc = run_items(wC, vq, vok, vit, vse, vE, st3[2]);
        if (rc) return rc;
        for (size_t r = 0; r < own.size(); ++r)
            for (int k = 0; k < ph.c_chunks; ++k) {
                const int64_t v = (int64_t)r * ph.c_chunks + k;
                if (ik_merge_item<0>(n, wC[v].s1 == ph.s_last, own[r], v, vq.data(), vok.data(), vit.data(), vse.data(), vE.data(), q_out, success, iters, searches, residual)) break;
            }
    }
    if (stats) for (int k = 0; k < 4; ++k) stats[k] = st3[0][k] + st3[1][k] + st3[2][k];   // [0]: the launches follow one another
    if (getenv("EMU_IK_DEBUG"))
        fprintf(stderr, "phases: A max %g tot %g useful %g | B items %zu max %g tot %g useful %g | C items %zu max %g tot %g useful %g\n", st3[0][0], st3[0][1], st3[0][2],
                wB.size(), st3[1][0], st3[1][1], st3[1][2], wC.size(), st3[2][0], st3[2][1], st3[2][2]);
    return 0;
}

// launch_ik's sharing mode (ik_kernels.hip): rows N .. N+M for donated ranges, the chain merge at the end
template <int NJ>
static int emu_ik_shared_run(const Chain *c, const IkDev &p, int waves, const double *Tep, const double *q0, double *q_out,
                             int32_t *success, int32_t *iters, int32_t *searches, double *residual, double *stats)
{
    const int n = NJ;
    const size_t qlimit = (size_t)std::min<int64_t>(p.N, 65536) / 8 + 256, qcap = qlimit + (size_t)(kIkGiveMax + 1) * waves + 64;
    const size_t M = (size_t)kIkQueues * qcap, rows = (size_t)p.N + M;
    std::vector<unsigned long long> wdyn(M, kIkNoItem), tc((size_t)kIkQueues * kIkQueueStride, 0ull);
    std::vector<int32_t> link(rows, -1), vok(rows, 0), vit(rows, 0), vse(rows, 0);
    std::vector<double> vq(rows * n, 0.0), vE(rows, 0.0);
    IkShareCtl sc;
    sc.tc = tc.data(); sc.wdyn = wdyn.data(); sc.link = link.data();
    sc.qlimit = (uint32_t)qlimit; sc.qcap = (uint32_t)qcap; sc.waves = (uint32_t)waves;
    sc.after = getenv("EMU_IK_DONATE_AFTER") ? (uint32_t)atoi(getenv("EMU_IK_DONATE_AFTER")) : 3u;
    const int rc = emu_ik_wave_run<NJ>(c, p, waves, Tep, q0, vq.data(), vok.data(), vit.data(), vse.data(), vE.data(), stats, nullptr, &sc);
    if (rc) return rc;
    // every wave ends holding one unserved ticket, so every appended item has been handed to a ticket before it
    unsigned long long unserved = 0, donated = 0;
    for (int g = 0; g < kIkQueues; ++g) {
        const unsigned long long x = tc[(size_t)g * kIkQueueStride];
        if (ik_word_tickets(x) < ik_word_count(x) || ik_word_count(x) > qcap) return -5;
        unserved += ik_word_waiting(x); donated += ik_word_count(x);
    }
    if (unserved != (unsigned long long)waves) return -5;
    for (int64_t t = 0; t < p.N; ++t)
        ik_merge_chain(n, t, link.data(), vq.data(), vok.data(), vit.data(), vse.data(), vE.data(), q_out, success, iters, searches, residual);
    if (getenv("EMU_IK_DEBUG")) fprintf(stderr, "sharing: %llu ranges donated\n", donated);
    return 0;
}

// null-space terms for the next emu_ik / emu_ik_wave calls (kq <= 0: none)
static double g_emu_ns[4] = {0.0, 0.0, 0.1, 0.3};
extern "C" void emu_ik_nullspace(double kq, double km, double ps, double pi) { g_emu_ns[0] = kq; g_emu_ns[1] = km; g_emu_ns[2] = ps; g_emu_ns[3] = pi; }
// IK_QP (method 5): slack gain for the next emu_ik / emu_ik_wave calls (kj is passed as lambda)
static double g_emu_ks = 1.0;
extern "C" void emu_ik_qp_ks(double ks) { g_emu_ks = ks; }
// restart-generator key of row 0 for the next emu_ik / emu_ik_wave calls (rtbhip_ik_target_base)
static int64_t g_emu_target0 = 0;
extern "C" void emu_ik_target_base(int64_t b) { g_emu_target0 = b; }

extern "C" int emu_ik(rtbhip_chain_t h, const double *Tep, int64_t N, const double *q0, int ilimit, int slimit, double tol,
                      int reject_jl, const double *we6, double lambda, int method, int flavour, uint64_t seed,
                      double *q_out, int32_t *success, int32_t *iters, int32_t *searches, double *residual)
{
    const std::shared_ptr<Chain> c_owner = chain_from_handle(h);
    Chain *c = c_owner.get();
    if (!c || c->n < 1 || c->n > kIkMaxJoints) return -1;
    IkDev p;
    p.ilimit = ilimit; p.slimit = slimit; p.reject_jl = reject_jl; p.method = method; p.flavour = flavour;
    p.has_q0 = q0 != nullptr; p.tol = tol; p.lambda = lambda; p.seed = seed; p.N = N; p.fresh_cap = 64; p.pool_chunk = 64; p.pass_mask = 0; p.spec_policy = getenv("EMU_IK_SPEC_POLICY") ? atoi(getenv("EMU_IK_SPEC_POLICY")) : 0;
    p.kq = g_emu_ns[0]; p.km = g_emu_ns[1]; p.ps = g_emu_ns[2]; p.pi = g_emu_ns[3]; p.ks = g_emu_ks; p.target0 = g_emu_target0;
    for (int k = 0; k < 6; ++k) p.we[k] = we6 ? we6[k] : 1.0;
    switch (c->n) {
    case 1: emu_ik_run<1>(c, p, Tep, q0, q_out, success, iters, searches, residual); break;
    case 2: emu_ik_run<2>(c, p, Tep, q0, q_out, success, iters, searches, residual); break;
    case 3: emu_ik_run<3>(c, p, Tep, q0, q_out, success, iters, searches, residual); break;
    case 4: emu_ik_run<4>(c, p, Tep, q0, q_out, success, iters, searches, residual); break;
    case 5: emu_ik_run<5>(c, p, Tep, q0, q_out, success, iters, searches, residual); break;
    case 6: emu_ik_run<6>(c, p, Tep, q0, q_out, success, iters, searches, residual); break;
    case 7: emu_ik_run<7>(c, p, Tep, q0, q_out, success, iters, searches, residual); break;
    case 8: emu_ik_run<8>(c, p, Tep, q0, q_out, success, iters, searches, residual); break;
    case 9: emu_ik_run<9>(c, p, Tep, q0, q_out, success, iters, searches, residual); break;
    case 10: emu_ik_run<10>(c, p, Tep, q0, q_out, success, iters, searches, residual); break;
    case 11: emu_ik_run<11>(c, p, Tep, q0, q_out, success, iters, searches, residual); break;
    case 12: emu_ik_run<12>(c, p, Tep, q0, q_out, success, iters, searches, residual); break;
    case 13: emu_ik_run<13>(c, p, Tep, q0, q_out, success, iters, searches, residual); break;
    case 14: emu_ik_run<14>(c, p, Tep, q0, q_out, success, iters, searches, residual); break;
    case 15: emu_ik_run<15>(c, p, Tep, q0, q_out, success, iters, searches, residual); break;
    default: emu_ik_run<16>(c, p, Tep, q0, q_out, success, iters, searches, residual); break;
    }
    return 0;
}

extern "C" int emu_ik_wave(rtbhip_chain_t h, int waves, double *stats, const double *Tep, int64_t N, const double *q0, int ilimit, int slimit, double tol,
                      int reject_jl, const double *we6, double lambda, int method, int flavour, uint64_t seed,
                      double *q_out, int32_t *success, int32_t *iters, int32_t *searches, double *residual)
{
    const std::shared_ptr<Chain> c_owner = chain_from_handle(h);
    Chain *c = c_owner.get();
    if (!c || c->n < 1 || c->n > kIkMaxJoints) return -1;
    IkDev p;
    p.ilimit = ilimit; p.slimit = slimit; p.reject_jl = reject_jl; p.method = method; p.flavour = flavour;
    p.has_q0 = q0 != nullptr; p.tol = tol; p.lambda = lambda; p.seed = seed; p.N = N; p.fresh_cap = 64; p.pool_chunk = 64; p.pass_mask = 0; p.spec_policy = getenv("EMU_IK_SPEC_POLICY") ? atoi(getenv("EMU_IK_SPEC_POLICY")) : 0;
    p.kq = g_emu_ns[0]; p.km = g_emu_ns[1]; p.ps = g_emu_ns[2]; p.pi = g_emu_ns[3]; p.ks = g_emu_ks; p.target0 = g_emu_target0;
    for (int k = 0; k < 6; ++k) p.we[k] = we6 ? we6[k] : 1.0;
    if (const char *pm = getenv("EMU_IK_PASS_MASK")) p.pass_mask = atoi(pm);
    { const int64_t g = waves; const int64_t cap = (N + g - 1) / g; p.fresh_cap = cap > 64 ? 64 : (int)cap;
      const int64_t lanes = g * kWave; p.pool_chunk = N >= 8 * lanes ? 64 : (N >= 3 * lanes ? 16 : 0); }
    if (const char *fc = getenv("EMU_IK_FRESH_CAP")) p.fresh_cap = atoi(fc);
    int rc = 0;
    const bool phased = getenv("EMU_IK_PHASED") != nullptr && atoi(getenv("EMU_IK_PHASED")) != 0;
    const bool shared = getenv("EMU_IK_SHARE") != nullptr && atoi(getenv("EMU_IK_SHARE")) != 0;
#define RTB_EMU_IK(NJ) case NJ: rc = shared ? emu_ik_shared_run<NJ>(c, p, waves, Tep, q0, q_out, success, iters, searches, residual, stats) \
                                    : phased ? emu_ik_phased_run<NJ>(c, p, waves, Tep, q0, q_out, success, iters, searches, residual, stats) \
                                             : emu_ik_wave_run<NJ>(c, p, waves, Tep, q0, q_out, success, iters, searches, residual, stats); break;
    switch (c->n) {
    RTB_EMU_IK(1) RTB_EMU_IK(2) RTB_EMU_IK(3) RTB_EMU_IK(4) RTB_EMU_IK(5) RTB_EMU_IK(6) RTB_EMU_IK(7) RTB_EMU_IK(8) RTB_EMU_IK(9) RTB_EMU_IK(10) RTB_EMU_IK(11) RTB_EMU_IK(12) RTB_EMU_IK(13) RTB_EMU_IK(14) RTB_EMU_IK(15)
    default: rc = shared ? emu_ik_shared_run<16>(c, p, waves, Tep, q0, q_out, success, iters, searches, residual, stats)
                : phased ? emu_ik_phased_run<16>(c, p, waves, Tep, q0, q_out, success, iters, searches, residual, stats)
                         : emu_ik_wave_run<16>(c, p, waves, Tep, q0, q_out, success, iters, searches, residual, stats); break;
    }
#undef RTB_EMU_IK
    return rc;
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

// dynamics terms: the kernel's per-lane body (dyn_device.h) on host arrays laid out like the LDS tile rows
template <int NJ, bool MDH, int MODE>
static void dyn_run(const Dyn *d, const double *q, const double *qd, const double *tq, int64_t N, V3 g, double *out)
{
    const DevLink *links = d->links.data();
    std::vector<double> in(3 * NJ), A(NJ * NJ);
    for (int64_t s = 0; s < N; ++s) {
        for (int j = 0; j < NJ; ++j) {
            in[j] = q[s * NJ + j];
            in[NJ + j] = qd ? qd[s * NJ + j] : 0.0;
            in[2 * NJ + j] = tq ? tq[s * NJ + j] : 0.0;
        }
        bool allrev = true;
        for (const DevLink &l : d->links) allrev = allrev && l.sigma == 0;
        // accel of an all-revolute chain with n >= 5: the kernel keeps q | qd | torque in the M tile itself (dyn_kernels.hip: alias_all)
        const bool alias_all = (MODE == kDynAccel && allrev && 3 * NJ <= NJ * (NJ + 1) / 2) || (MODE == kDynCoriolis && allrev && NJ >= 2);
        if (alias_all) for (int k = 0; k < (MODE == kDynAccel ? 3 : 2) * NJ; ++k) A[k] = in[k];
        const double *mine = alias_all ? A.data() : in.data();
        if (allrev) dyn_lane<NJ, MDH, MODE, true>(links, mine, A.data(), g);
        else dyn_lane<NJ, MDH, MODE, false>(links, mine, A.data(), g);
        const int W = MODE == kDynAccel ? NJ : NJ * NJ;
        if (MODE == kDynInertia && allrev) {           // packed lower triangle -> (n, n), as the kernel's flush_symmetric
            for (int r = 0; r < NJ; ++r)
                for (int c = 0; c < NJ; ++c) {
                    const int hi = r > c ? r : c, lo = r > c ? c : r;
                    out[s * W + r * NJ + c] = A[hi * (hi + 1) / 2 + lo];
                }
        } else {
            for (int k = 0; k < W; ++k) out[s * W + k] = A[k];
        }
    }
}
template <int NJ>
static void dyn_nj(const Dyn *d, int mode, const double *q, const double *qd, const double *tq, int64_t N, V3 g, double *out)
{
    if (d->mdh) {
        if (mode == 0) dyn_run<NJ, true, kDynInertia>(d, q, qd, tq, N, g, out);
        else if (mode == 1) dyn_run<NJ, true, kDynCoriolis>(d, q, qd, tq, N, g, out);
        else dyn_run<NJ, true, kDynAccel>(d, q, qd, tq, N, g, out);
    } else {
        if (mode == 0) dyn_run<NJ, false, kDynInertia>(d, q, qd, tq, N, g, out);
        else if (mode == 1) dyn_run<NJ, false, kDynCoriolis>(d, q, qd, tq, N, g, out);
        else dyn_run<NJ, false, kDynAccel>(d, q, qd, tq, N, g, out);
    }
}
extern "C" int emu_dyn(rtbhip_dyn_t h, int mode, const double *q, const double *qd, const double *tq, int64_t N,
                       const double *grav3, double *out)
{
    const std::shared_ptr<Dyn> d_owner = dyn_from_handle(h);
    Dyn *d = d_owner.get();
    if (!d || d->n > 16) return -1;
    V3 g = grav3 ? v3(grav3[0], grav3[1], grav3[2]) : v3(0, 0, 0);
    switch (d->n) {
    case 1: dyn_nj<1>(d, mode, q, qd, tq, N, g, out); break;
    case 2: dyn_nj<2>(d, mode, q, qd, tq, N, g, out); break;
    case 3: dyn_nj<3>(d, mode, q, qd, tq, N, g, out); break;
    case 4: dyn_nj<4>(d, mode, q, qd, tq, N, g, out); break;
    case 5: dyn_nj<5>(d, mode, q, qd, tq, N, g, out); break;
    case 6: dyn_nj<6>(d, mode, q, qd, tq, N, g, out); break;
    case 7: dyn_nj<7>(d, mode, q, qd, tq, N, g, out); break;
    case 8: dyn_nj<8>(d, mode, q, qd, tq, N, g, out); break;
    case 9: dyn_nj<9>(d, mode, q, qd, tq, N, g, out); break;
    case 10: dyn_nj<10>(d, mode, q, qd, tq, N, g, out); break;
    case 11: dyn_nj<11>(d, mode, q, qd, tq, N, g, out); break;
    case 12: dyn_nj<12>(d, mode, q, qd, tq, N, g, out); break;
    case 13: dyn_nj<13>(d, mode, q, qd, tq, N, g, out); break;
    case 14: dyn_nj<14>(d, mode, q, qd, tq, N, g, out); break;
    case 15: dyn_nj<15>(d, mode, q, qd, tq, N, g, out); break;
    default: dyn_nj<16>(d, mode, q, qd, tq, N, g, out); break;
    }
    return 0;
}

// differential-kinematics consumers: diff_device.h on the register-resident Jacobian of kin_reg.h
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
template <int NG>
static void tree_run(const Tree *t, const double *q, const double *qd, const double *qdd, int64_t N, V3 g, double *tau)
{
    std::vector<double> slots((size_t)kTreeSlotDoubles * std::max(1, t->nslots));
    for (int64_t s = 0; s < N; ++s) {
        const double *a = q + s * NG, *b = qd + s * NG, *c = qdd + s * NG;
        double *o = tau + s * NG;
        tree_rne_lane<NG>(t->groups.data(), t->nslots, g, [&](int k) { return a[k]; }, [&](int k) { return b[k]; },
                          [&](int k) { return c[k]; }, [&](int k, double v) { o[k] = v; },
                          [&](int i) -> double & { return slots[i]; });
    }
}
extern "C" int emu_tree_rne(const rtbhip_tree_group *groups, int ng, const double *q, const double *qd, const double *qdd,
                            int64_t N, const double *grav3, double *tau)
{
    Tree t;
    if (compile_tree(groups, ng, &t) != RTBHIP_OK) return -1;
    V3 g = v3(grav3[0], grav3[1], grav3[2]);
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
