// ik_kernels.hip -- gfx950 kernel for batched Levenberg-Marquardt IK with the whole restart loop
// resident on the device.  Replaces the one-target-per-call IK_LM_c (core/fknm.cpp:394-525 ->
// core/ik.cpp:19-75,157-209) and the Python loop over targets of IKSolver.solve (robot/IK.py:263-290).
//
// Persistent lanes: the grid is sized to the chip, not to N.  Every lane runs the per-target state
// machine of ik_device.h; a lane whose target finishes (after 5 or after 3000 iterations -- the
// spread is that wide) takes the next unsolved target from a device-wide counter.  The fetch is
// aggregated per wave (one atomicAdd for all idle lanes of the wave) so the counter sees at most
// one atomic per wave per iteration.  This is compute/latency-bound work (~1.5 kflop of dependent
// fp64 per iteration, 204 B of I/O per target): MFMA does not apply (7x7 normal equations per lane).
#include "ik_device.h"
#include <atomic>

namespace rtbhip {

#define RTB_CONST __attribute__((address_space(4)))
struct ConstChainIk {
    const RTB_CONST DevSeg *seg;
    const RTB_CONST int32_t *jmeta;
};

template <int NJ>
__global__ __launch_bounds__(kWave) void k_ik(IkDev p, DevChain dc, const double *qlim_g, const double *__restrict__ Tep,
                                             const double *__restrict__ q0, unsigned long long *counter,
                                             double *__restrict__ q_out, int32_t *__restrict__ success,
                                             int32_t *__restrict__ iters, int32_t *__restrict__ searches,
                                             double *__restrict__ residual)
{
    ConstChainIk cv;
    cv.seg = (const RTB_CONST DevSeg *)dc.seg;
    cv.jmeta = (const RTB_CONST int32_t *)dc.jmeta;
    const RTB_CONST double *qlim = (const RTB_CONST double *)qlim_g;
    const int lane = threadIdx.x;
    IkState<NJ> st;
    st.tgt = -1; st.E = 0.0; st.iter = 0; st.search = 0; st.it = 0; st.draws = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) st.q[j] = 0.0;
#pragma unroll
    for (int k = 0; k < 12; ++k) st.Td[k] = (k == 0 || k == 4 || k == 8) ? 1.0 : 0.0;
    bool exhausted = false;
    for (;;) {
        const bool need = st.tgt < 0 && !exhausted;
        const unsigned long long m = __ballot(need);
        if (m) {   // wave-uniform
            const int leader = __ffsll((long long)m) - 1;
            unsigned long long base = 0;
            if (lane == leader) base = atomicAdd(counter, (unsigned long long)__popcll(m));
            const unsigned lo = __shfl((unsigned)(base & 0xffffffffu), leader);
            const unsigned hi = __shfl((unsigned)(base >> 32), leader);
            base = ((unsigned long long)hi << 32) | lo;
            if (need) {
                const int64_t t = (int64_t)base + __popcll(m & ((1ull << lane) - 1ull));
                if (t < p.N) ik_begin<NJ>(st, p, qlim, t, Tep + 16 * t, p.has_q0 ? q0 + (int64_t)NJ * t : nullptr);
                else exhausted = true;
            }
        }
        const bool active = st.tgt >= 0;
        if (!__any(active)) break;
        int ok = 0;
        const bool fin = ik_advance<NJ>(st, p, cv, qlim, ok);
        if (active && fin) {
            const int64_t t = st.tgt;
#pragma unroll
            for (int j = 0; j < NJ; ++j) q_out[t * NJ + j] = st.q[j];
            success[t] = ok;
            iters[t] = st.it;
            searches[t] = st.search;
            residual[t] = st.E;
            st.tgt = -1;
        }
        if (st.tgt < 0) {   // parked lanes keep executing the iteration: keep their state finite
#pragma unroll
            for (int j = 0; j < NJ; ++j) st.q[j] = 0.0;
        }
    }
}

namespace {
int g_ik_waves_per_cu = 8;
std::mutex g_ctr_mu;
std::map<int, unsigned long long *> g_ctr;     // per-device ring of work counters
std::atomic<unsigned> g_ctr_next{0};
constexpr int kCtrRing = 256;
}  // namespace

void ik_tune(const char *key, int value)
{
    if (std::string(key) == "ik_waves_per_cu") g_ik_waves_per_cu = value < 1 ? 1 : value;
}

void ik_restart_host(const Chain *c, uint64_t seed, int64_t target, int draw, double *q_n)
{
    const int n = c->n;
    for (int j = 0; j < n; ++j) {
        const double lo = c->qlim[j], hi = c->qlim[n + j];
        q_n[j] = lo + ik_uniform(seed, target, draw, j) * (hi - lo);
    }
}

template <int NJ>
static void launch_nj(dim3 grid, hipStream_t s, const IkDev &p, const DevChain &dc, const double *qlim, const double *Tep,
                      const double *q0, unsigned long long *ctr, double *q_out, int32_t *success, int32_t *iters,
                      int32_t *searches, double *residual)
{
    hipLaunchKernelGGL((k_ik<NJ>), grid, dim3(kWave), 0, s, p, dc, qlim, Tep, q0, ctr, q_out, success, iters, searches, residual);
}

int launch_ik(const Chain *c, const DevChain &dc, const double *qlim, const double *Tep, int64_t N, const double *q0,
              const IkParams &ip, double *q_out, int32_t *success, int32_t *iters, int32_t *searches, double *residual,
              hipStream_t s)
{
    if (N == 0) return RTBHIP_OK;
    if (c->n > kRegMaxJoints) { set_error("ik_lm: this build solves chains of up to 8 joints on the device"); return RTBHIP_ELIMIT; }
    for (int j = 0; j < c->n; ++j)
        if (jm_jq(c->jmeta[j]) != j) { set_error("ik_lm: jindex must equal the joint order (the reference's ik.cpp:57 adds dq in that order)"); return RTBHIP_EINVAL; }
    IkDev p;
    p.ilimit = ip.ilimit; p.slimit = ip.slimit; p.reject_jl = ip.reject_jl; p.method = ip.method;
    p.flavour = ip.flavour; p.has_q0 = q0 != nullptr; p.tol = ip.tol; p.lambda = ip.lambda;
    for (int k = 0; k < 6; ++k) p.we[k] = ip.we[k];
    Affine none; none.used = 0;
    for (int k = 0; k < 12; ++k) none.v[k] = 0.0;
    chain_tail(c, none, p.tail);
    p.seed = ip.seed;
    p.N = N;
    int dev = 0, cus = 0;
    RTB_HIP(hipGetDevice(&dev));
    if (device_cu_count(&cus) != RTBHIP_OK) return RTBHIP_EHIP;
    unsigned long long *ring = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_ctr_mu);
        auto it = g_ctr.find(dev);
        if (it == g_ctr.end()) {
            RTB_HIP(hipMalloc((void **)&ring, kCtrRing * sizeof(unsigned long long)));
            g_ctr[dev] = ring;
        } else {
            ring = it->second;
        }
    }
    unsigned long long *ctr = ring + (g_ctr_next.fetch_add(1) % kCtrRing);
    RTB_HIP(hipMemsetAsync(ctr, 0, sizeof(unsigned long long), s));
    const int64_t tiles = (N + kWave - 1) / kWave;
    int64_t g = (int64_t)cus * g_ik_waves_per_cu;
    if (g > tiles) g = tiles;
    dim3 grid((unsigned)g);
    switch (c->n) {
    case 1: launch_nj<1>(grid, s, p, dc, qlim, Tep, q0, ctr, q_out, success, iters, searches, residual); break;
    case 2: launch_nj<2>(grid, s, p, dc, qlim, Tep, q0, ctr, q_out, success, iters, searches, residual); break;
    case 3: launch_nj<3>(grid, s, p, dc, qlim, Tep, q0, ctr, q_out, success, iters, searches, residual); break;
    case 4: launch_nj<4>(grid, s, p, dc, qlim, Tep, q0, ctr, q_out, success, iters, searches, residual); break;
    case 5: launch_nj<5>(grid, s, p, dc, qlim, Tep, q0, ctr, q_out, success, iters, searches, residual); break;
    case 6: launch_nj<6>(grid, s, p, dc, qlim, Tep, q0, ctr, q_out, success, iters, searches, residual); break;
    case 7: launch_nj<7>(grid, s, p, dc, qlim, Tep, q0, ctr, q_out, success, iters, searches, residual); break;
    default: launch_nj<8>(grid, s, p, dc, qlim, Tep, q0, ctr, q_out, success, iters, searches, residual); break;
    }
    note_launch((int)grid.x, kWave, 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "k_ik launch");
    return RTBHIP_OK;
}

}  // namespace rtbhip
