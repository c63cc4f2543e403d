#include "emu_ik_wave.h"
RTB_EMU_IK_DISPATCH(emu_ik_wave_hi, RTB_EMU_IK(13) RTB_EMU_IK(14) RTB_EMU_IK(15) RTB_EMU_IK(16))

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
    p.kq = g_emu_ns[0]; p.km = g_emu_ns[1]; p.ps = g_emu_ns[2]; for (int j = 0; j < 16; ++j) p.pi[j] = g_emu_pi[j]; p.ks = g_emu_ks; p.target0 = g_emu_target0;
    p.flat_chunks = 0; p.flat_l0 = 0; p.flat_len = 0; p.flat_n = 0; p.flat_done = nullptr; p.stats = nullptr;
    for (int k = 0; k < 6; ++k) p.we[k] = we6 ? we6[k] : 1.0;
    p.unit_we = 1; p.pad_we = 0;
    for (int k = 0; k < 6; ++k) p.unit_we = p.unit_we && p.we[k] == 1.0;
    if (const char *pm = getenv("EMU_IK_PASS_MASK")) p.pass_mask = atoi(pm);
    { const int64_t g = waves; const int64_t cap = (N + g - 1) / g; p.fresh_cap = cap > 64 ? 64 : (int)cap;
      const int64_t lanes = g * kWave; p.pool_chunk = N >= 8 * lanes ? 64 : (N >= 3 * lanes ? 16 : 0); }
    if (const char *fc = getenv("EMU_IK_FRESH_CAP")) p.fresh_cap = atoi(fc);
    const bool phased = getenv("EMU_IK_PHASED") != nullptr && atoi(getenv("EMU_IK_PHASED")) != 0;
    const bool shared = getenv("EMU_IK_SHARE") != nullptr && atoi(getenv("EMU_IK_SHARE")) != 0;
    const int n = c->n;
    if (n <= 7) return emu_ik_wave_lo(n, shared, phased, c, p, waves, Tep, q0, q_out, success, iters, searches, residual, stats);
    if (n <= 12) return emu_ik_wave_mid(n, shared, phased, c, p, waves, Tep, q0, q_out, success, iters, searches, residual, stats);
    return emu_ik_wave_hi(n, shared, phased, c, p, waves, Tep, q0, q_out, success, iters, searches, residual, stats);
}
