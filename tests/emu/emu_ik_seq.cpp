// tests/emu/emu_ik_seq.cpp -- TEST INFRASTRUCTURE: the IK specification (searches one after another) on the kernel's own lane functions.
#include "emu_common.h"

template <int NJ>
static void emu_ik_run(const Chain *c, const IkDev &p, const double *Tep, const double *q0, double *q_out, int32_t *success,
                       int32_t *iters, int32_t *searches, double *residual)
{
    const EmuChainIk cv = emu_chain_ik(c);
    const double *qlim = c->qlim.data();
    for (int64_t t = 0; t < p.N; ++t)   // the specification: searches one after another
        ik_solve_sequential<NJ>(p, cv, qlim, t, Tep, q0, q_out, success, iters, searches, residual);
}

// null-space terms for the next emu_ik / emu_ik_wave calls (kq <= 0: none)
double g_emu_ns[4] = {0.0, 0.0, 0.1, 0.3};
double g_emu_pi[16] = {0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3, 0.3};
extern "C" void emu_ik_nullspace(double kq, double km, double ps, double pi)
{
    g_emu_ns[0] = kq; g_emu_ns[1] = km; g_emu_ns[2] = ps; g_emu_ns[3] = pi;
    for (int j = 0; j < 16; ++j) g_emu_pi[j] = pi;
}
extern "C" void emu_ik_nullspace_pi(const double *pi, int n) { for (int j = 0; j < 16; ++j) g_emu_pi[j] = pi[j < n ? j : n - 1]; }
// IK_QP (method 5): slack gain for the next emu_ik / emu_ik_wave calls (kj is passed as lambda)
double g_emu_ks = 1.0;
extern "C" void emu_ik_qp_ks(double ks) { g_emu_ks = ks; }
// restart-generator key of row 0 for the next emu_ik / emu_ik_wave calls (rtbhip_ik_target_base)
int64_t g_emu_target0 = 0;
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
    p.kq = g_emu_ns[0]; p.km = g_emu_ns[1]; p.ps = g_emu_ns[2]; for (int j = 0; j < 16; ++j) p.pi[j] = g_emu_pi[j]; p.ks = g_emu_ks; p.target0 = g_emu_target0;
    p.flat_chunks = 0; p.flat_l0 = 0; p.flat_len = 0; p.flat_n = 0; p.flat_done = nullptr; p.stats = nullptr;
    for (int k = 0; k < 6; ++k) p.we[k] = we6 ? we6[k] : 1.0;
    p.unit_we = 1; p.pad_we = 0;
    for (int k = 0; k < 6; ++k) p.unit_we = p.unit_we && p.we[k] == 1.0;
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
