// tests/emu/emu_kin.cpp -- TEST INFRASTRUCTURE: fkine / Jacobian / Hessian kernel bodies replayed on the CPU (see emu_common.h).
#include "emu_common.h"

extern "C" int emu_kin(rtbhip_chain_t h, const double *q, int64_t N, const double *base16, const double *tool16,
                       int frame, double *T, double *J, double *H, int coalesced)
{
    const std::shared_ptr<Chain> c_owner = chain_from_handle(h);
    Chain *c = c_owner.get();
    if (!c) return -1;
    if (c->n == 0) { J = nullptr; H = nullptr; if (!T) return 0; }      // as kin_entry (api.cpp): empty Jacobian / Hessian of a chain of constants
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

// packed rows (rtbhip_fkine_jacob_packed): the register tile's rounds of kPRound lanes / the run-time-n tile's single run, as k_kin_reg<.., true>
// and k_kin's packed branch execute them
template <int NJ>
static void emu_reg_packed_run(const KinParams &kp, const DevChain &cv, const double *q, int64_t N, double *TJ)
{
    std::vector<double> buf(reg_lds_doubles_packed(NJ), -777.0);
    constexpr int W = 6 * NJ;
    double *bufT = buf.data(), *bufJ = buf.data() + kPRound * 17;
    const int64_t tiles = (N + kWave - 1) / kWave;
    for (int64_t tile = 0; tile < tiles; ++tile) {
        const int64_t cfg0 = tile * kWave;
        const int ncfg = (int)std::min<int64_t>(kWave, N - cfg0);
        Pose P[kWave];
        double jac[kWave][6 * NJ];
        for (int l = 0; l < kWave; ++l) reg_compute<NJ, true>(kp, cv, q, cfg0 + l, P[l], jac[l]);
        for (int r = 0; r < kWave / kPRound; ++r) {
            for (int l = 0; l < kWave; ++l)
                if (l / kPRound == r) reg_stage_packed<NJ>(kp, P[l], jac[l], bufT, bufJ, l % kPRound);
            int rows = std::max(0, std::min(kPRound, ncfg - r * kPRound));
            for (int l = 0; l < kWave; ++l) kin_flush_packed(bufT, bufJ, W + 1, W, rows, TJ + (cfg0 + r * kPRound) * (16 + W), l);
        }
    }
}

extern "C" int emu_kin_packed(rtbhip_chain_t h, const double *q, int64_t N, const double *base16, const double *tool16,
                              int frame, double *TJ, int reg)
{
    const std::shared_ptr<Chain> c_owner = chain_from_handle(h);
    Chain *c = c_owner.get();
    if (!c) return -1;
    KinParams kp;
    kp.n = c->n; kp.qw = c->q_width; kp.stride = kin_stride(c->n);
    kp.frame = frame; kp.N = N; kp.pad = 2;
    Affine b = aff16(base16), t = aff16(tool16);
    kp.has_base = b.used;
    for (int i = 0; i < 12; i++) kp.base[i] = b.v[i];
    chain_tail(c, t, kp.tail);
    const DevChain cv = chain_host_view(c);
    if (c->n == 0) return emu_kin(h, q, N, base16, tool16, frame, TJ, nullptr, nullptr, 1);      // as kin_packed_entry (api.cpp)
    if (reg && c->n <= kKinRegMax) {
        switch (c->n) {
        case 1: emu_reg_packed_run<1>(kp, cv, q, N, TJ); break;
        case 2: emu_reg_packed_run<2>(kp, cv, q, N, TJ); break;
        case 3: emu_reg_packed_run<3>(kp, cv, q, N, TJ); break;
        case 4: emu_reg_packed_run<4>(kp, cv, q, N, TJ); break;
        case 5: emu_reg_packed_run<5>(kp, cv, q, N, TJ); break;
        case 6: emu_reg_packed_run<6>(kp, cv, q, N, TJ); break;
        case 7: emu_reg_packed_run<7>(kp, cv, q, N, TJ); break;
        case 8: emu_reg_packed_run<8>(kp, cv, q, N, TJ); break;
        case 9: emu_reg_packed_run<9>(kp, cv, q, N, TJ); break;
        default: emu_reg_packed_run<10>(kp, cv, q, N, TJ); break;
        }
        return 0;
    }
    std::vector<double> lds(kin_lds_bytes(kp.n, kp.qw) / sizeof(double) + kWave * 17, -777.0);
    double *rows = lds.data(), *qs = lds.data() + kWave * kp.stride, *rowsT = qs + kWave * kp.qw;
    const int W = 6 * kp.n;
    const int64_t tiles = (N + kWave - 1) / kWave;
    for (int64_t tile = 0; tile < tiles; ++tile) {
        const int64_t cfg0 = tile * kWave;
        const int ncfg = (int)std::min<int64_t>(kWave, N - cfg0);
        for (int l = 0; l < kWave; ++l) kin_load_q(kp, q, cfg0 + l, l, qs);
        for (int l = 0; l < kWave; ++l) {
            Pose P;
            kin_walk<true>(kp, cv, l, qs, rows, P);
            if (kp.has_base) pose_premul(P, kp.base);
            pose_store16(P, [&](int k, double v) { rowsT[l * 17 + k] = v; });
        }
        for (int l = 0; l < kWave; ++l) kin_flush_packed(rowsT, rows, kp.stride, W, ncfg, TJ + cfg0 * (16 + W), l);
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
static int emu_pose_error(const double *Te, int64_t nTe, const double *Tep, int64_t nTep, int method, double *e);
extern "C" int emu_angle_axis(const double *Te, int64_t nTe, const double *Tep, int64_t nTep, double *e) { return emu_pose_error(Te, nTe, Tep, nTep, 0, e); }
// k_angle_axis<true>: the same staging around servo_rpy_lane (method 1, p_servo's "rpy")
extern "C" int emu_p_servo_error(const double *Te, int64_t nTe, const double *Tep, int64_t nTep, int method, double *e) { return emu_pose_error(Te, nTe, Tep, nTep, method, e); }
static int emu_pose_error(const double *Te, int64_t nTe, const double *Tep, int64_t nTep, int method, double *e)
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
        for (int l = 0; l < kWave; ++l) {
            if (method == 1) servo_rpy_lane(t1[l], t2[l], a.data() + l * 7);
            else aa_lane(t1[l], t2[l], a.data() + l * 7);
        }
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
