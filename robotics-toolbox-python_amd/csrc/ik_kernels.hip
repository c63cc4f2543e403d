// ik_kernels.hip -- gfx950 kernel for batched Levenberg-Marquardt IK with the whole restart loop
// resident on the device.  Replaces the one-target-per-call IK_LM_c (core/fknm.cpp:394-525 ->
// core/ik.cpp:19-75,157-209) and the Python loop over targets of IKSolver.solve (robot/IK.py:263-290).
//
// Persistent waves: the grid is sized to the chip, not to N.  Every lane runs one SEARCH of one target
// at a time (ik_device.h); idle lanes take fresh targets from a device-wide counter (one atomicAdd
// per wave per scheduling pass) and, once the supply is exhausted, later search indices of their own
// wave's unresolved targets -- the 1 % of targets that need dozens of restarts (up to ~3000
// sequential iterations in the reference, against a mean of 75) no longer set the kernel's duration.
// First MI355X measurement of the previous one-lane-per-target version: 13.4 ms per 1e5 targets
// with 97 % of the lane-iterations idle.  This is compute/latency-bound work (~1.5 kflop of dependent
// fp64 per iteration, 204 B of I/O per target): MFMA does not apply (7x7 normal equations per lane).
#include "ik_device.h"
#if RTB_HOST_SIDE
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <vector>
#endif

namespace rtbhip {

#define RTB_CONST __attribute__((address_space(4)))
struct ConstChainIk {
    const RTB_CONST DevSeg *seg;
    const RTB_CONST int32_t *jmeta;
    const RTB_CONST double *trig;       // kIkSincosTable, laundered with the other two once per iteration (trig.h: sincos_reduced_tab)
};
__constant__ double kIkSincosTable[kSincosTableLen] = RTB_SINCOS_TABLE_INIT;

// The kernel's argument block, member for member.  Inside the persistent loop everything is read through a pointer to
// the kernarg segment that is laundered once per iteration: as plain by-value arguments the eleven pointers and the
// solver parameters stayed in SGPRs across the whole loop (the scheduling pass needs them, the LM iteration does not)
// and the register allocator paid for that with v_readlane / v_writelane traffic inside the iteration.
struct IkKernArgs {
    IkDev p;
    DevChain dc;
    const double *qlim, *Tep, *q0;
    unsigned long long *counter;
    double *q_out;
    int32_t *success, *iters, *searches;
    double *residual;
    const IkWork *work;              // NULL: item v is target v with the whole search range
    const unsigned *count;           // NULL: p.N items; else the item count is read from the device (a compacted list)
    IkShareCtl share;                // share.counter != NULL: cross-wave sharing of search ranges (ik_device.h)
};

// the sharing control block out of the (constant address space) argument block, member by member
__device__ __forceinline__ IkShareCtl share_of(const RTB_CONST IkKernArgs *ka)
{
    IkShareCtl c;
    c.tc = ka->share.tc; c.wdyn = ka->share.wdyn; c.qlimit = ka->share.qlimit; c.qcap = ka->share.qcap;
    c.link = ka->share.link; c.waves = ka->share.waves; c.after = ka->share.after;
    return c;
}

// Wave-level driver of the scheduler phases of ik_device.h (the same sequence tests/emu replays on the CPU).
constexpr int kIkNullMax = 12;   // null-space step variants: 6..8 joints in registers, 9..12 with scratch
#ifndef RTB_IK_WAVES
#define RTB_IK_WAVES 2
#endif
#ifndef RTB_IK_PASS_LOW
#define RTB_IK_PASS_LOW -1      // >= 0: a wave with at most this many running lanes passes at every iteration in which a search has ended (A/B, round 5)
#endif
#ifndef RTB_IK_SHARE
#define RTB_IK_SHARE 1          // 0: build without the cross-wave sharing code (A/B of what its presence costs the plain schedule)
#endif
// AUX: bit 0 = the flat schedule, bit 1 = the per-wave diagnostic counters (RTBHIP_IK_STATS).  Compile-time, because carrying either through the
// persistent loop as run-time switches cost the plain schedule 6-9 % (20 VGPRs; round 3, visit x: the round-2 build against this one on one box).
constexpr int kIkStatWords = 6;  // per wave: loop iterations, scheduling passes, lane-iterations on a running search, items started, shader cycles, 100 MHz ticks
constexpr int kIkAuxFlat = 1, kIkAuxStats = 2, kIkAuxUnitW = 4;      // bit 2: every mask weight is 1 (the default), LM steps: ik_iter<..., UNITW>
#ifndef RTB_IK_MASK_IDLE
#define RTB_IK_MASK_IDLE 0
#endif
constexpr int kIkAuxPlain = 8;                                         // bit 3: all-revolute chain, no flipped joint: ik_iter<..., PLAIN>
// Structure signatures this build has straight-line instantiations for (kin_reg.h: SegSig; the chain compiler's classes, rtbhip_internal.h).  A
// signature is a property of the robot's constants; the launcher compares a chain's own with this list and falls back to the general kernels.
//   Franka Panda as the reference models it (models/ETS/Panda.py:32-54), BASELINE config 3's arm: C_0 = tz, C_1 .. C_6 quarter turns about x with
//   translations on some axes, the flange Rz(-pi/4) tz(0.103) as the tail.
constexpr SegSig kIkSigPandaETS = kSegSigPresent | seg_sig_of(0, kSegIdentity, 4) | seg_sig_of(1, kSegRxN, 0) | seg_sig_of(2, kSegRxP, 6) | seg_sig_of(3, kSegRxP, 1) |
                                  seg_sig_of(4, kSegRxN, 7) | seg_sig_of(5, kSegRxP, 0) | seg_sig_of(6, kSegRxP, 7) | seg_sig_of(7, kSegRz, 4);
//   The same arm read from its URDF (rtb-data franka_description, to the default end effector): the constants' tiny cos(pi/2) terms fall on other entries.
constexpr SegSig kIkSigPandaURDF = kSegSigPresent | seg_sig_of(0, kSegIdentity, 4) | seg_sig_of(1, kSegRxN, 0) | seg_sig_of(2, kSegRxP, 2) | seg_sig_of(3, kSegRxP, 1) |
                                   seg_sig_of(4, kSegRxN, 3) | seg_sig_of(5, kSegRxP, 0) | seg_sig_of(6, kSegRxP, 1) | seg_sig_of(7, kSegRz, 4);
//   Universal Robots UR3 / UR5 / UR10 from their URDFs (ur_description, to tool0): six joints, one signature for the three sizes.
constexpr SegSig kIkSigUR = kSegSigPresent | seg_sig_of(0, kSegIdentity, 4) | seg_sig_of(1, kSegGeneral, 2) | seg_sig_of(2, kSegIdentity, 5) | seg_sig_of(3, kSegRzP, 1) |
                            seg_sig_of(4, kSegPermA, 4) | seg_sig_of(5, kSegPermB, 4) | seg_sig_of(6, kSegGeneral, 4);
static int g_ik_sig = 1;          // rtbhip_tune("ik_sig", 0): never take a signature's instantiation (A/B, tests)
template <int NJ, int STEP, int AUX = 0, SegSig SIG = 0>
__global__ __launch_bounds__(kWave, (NJ <= kRegMaxJoints && !(STEP & kIkStepNull) ? RTB_IK_WAVES : 1)) void k_ik(IkDev p, DevChain dc, const double *qlim_g, const double *__restrict__ Tep,
                                                const double *__restrict__ q0, unsigned long long *counter,
                                                double *__restrict__ q_out, int32_t *__restrict__ success,
                                                int32_t *__restrict__ iters, int32_t *__restrict__ searches,
                                                double *__restrict__ residual, const IkWork *work_g, const unsigned *count_g, IkShareCtl share_g)
{
    __shared__ IkWaveSharedFor<NJ> sh;
    const RTB_CONST IkKernArgs *ka = (const RTB_CONST IkKernArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane = threadIdx.x;
    const int s_last = ik_s_last(p);
    IkLane<NJ> st;
    st.status = kIkIdle; st.E = 0.0; st.iter = 0; st.s = 0; st.slot = 0; st.fin = 0; st.ok = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) sh.q[j][lane] = 0.0;
#pragma unroll
    for (int k = 0; k < 12; ++k) sh.Td[k][lane] = (k == 0 || k == 4 || k == 8) ? 1.0 : 0.0;   // slot `lane`
    // The scheduler's wave-uniform state -- which slots are busy, the wave's pool of reserved work, what it knows about the counter -- changes
    // only inside a scheduling pass (every 4th iteration) but, as loop-carried scalars, was alive across the LM iteration as well, whose FK walk
    // needs the whole scalar register file for the chain constants: the compiler parked ~70 SGPRs in VGPR lanes before the walk and fetched
    // them back after it, every iteration (212 v_writelane / v_readlane of the 1486 VALU instructions of an iteration; round 4).  It lives in
    // LDS now: a pass loads it, works on locals, decides whether the wave leaves, and stores it back; between passes nothing of it is live.
    __shared__ struct {
        unsigned long long busy;         // slots holding an unresolved target
        unsigned long long pool_next, pool_end;   // targets reserved by this wave and not started yet
        unsigned long long pool_live;    // flat schedule: bit k = item pool_next + k was drawn, is still worth starting and has not been started
        unsigned long long pend_item;    // sharing: a range handed to this wave (ticket pend_tick of its queue), started at the next pass
        unsigned long long NN;           // number of work items of this launch
        unsigned pend_tick;
        unsigned flags;                  // kWsExhausted | kWsDrained | kWsEvidence | kWsC0Out
    } ws;
    constexpr unsigned kWsExhausted = 1u;   // the global supply of fresh targets has run out
    constexpr unsigned kWsDrained = 2u;     // the device-wide counter has passed N
    constexpr unsigned kWsEvidence = 4u;    // flat schedule: one of this wave's own chunk-0 items has FAILED -- first chunks do fail in this batch
    constexpr unsigned kWsC0Out = 8u;       // flat schedule: the device-wide counter has passed the chunk-0 numbers
    auto ws_get64 = [](const unsigned long long &w) -> unsigned long long {      // a uniform LDS word into scalar registers
        const unsigned long long v = w;
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    };
    auto ws_get32 = [](const unsigned &w) -> unsigned { return (unsigned)__builtin_amdgcn_readfirstlane((int)w); };
    constexpr bool flat = (AUX & kIkAuxFlat) != 0;    // the launcher picks this instantiation exactly when p.flat_chunks > 0
    constexpr bool kStats = (AUX & kIkAuxStats) != 0;
    ws.busy = 0; ws.pool_next = 0; ws.pool_end = 0; ws.pool_live = 0; ws.pend_item = kIkNoItem; ws.pend_tick = 0; ws.flags = 0;
    ws.NN = count_g ? (unsigned long long)*count_g : (unsigned long long)p.N;
    __syncthreads();
    unsigned long long st_iters = 0, st_passes = 0, st_lane = 0, st_items = 0;   // diagnostics (p.stats), wave-uniform
    unsigned long long st_t0 = 0, st_r0 = 0;                                     // ... the wave's life in shader cycles (s_memtime) and in 100 MHz ticks (s_memrealtime)
    if constexpr (kStats) { st_t0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
    bool first = true;
    unsigned tick = 0;
    int leave = 0;                    // set by a pass: 1 the wave is done, 2 out of work with sharing on (take a ticket and wait)
    const long long patience = ik_patience(p, s_last);    // watchdog budget (ik_device.h), the pass latency included
    long long quiet = 0;
    const bool sharing = RTB_IK_SHARE && share_g.tc != nullptr;   // wave-uniform
    for (;;) {
        asm volatile("" : "+s"(ka));
        // the scheduling pass runs when some search has ended -- at most every (pass_mask + 1)-th iteration: a
        // finished lane then idles for up to pass_mask iterations (of ~31 per search) and the pass, several
        // hundred mostly scalar / LDS instructions, is amortised over more useful iterations
#if RTB_IK_PASS_LOW >= 0
        // ... and at ANY iteration once at most RTB_IK_PASS_LOW lanes of the wave still run a search: the pass then costs less than the lanes it refills
        // (0: only when nothing runs at all -- the iterations a wave would otherwise burn waiting for its pass slot)
        const bool pass_slot = (tick++ & ka->p.pass_mask) == 0 || __popcll(__ballot(st.status == kIkRun && !st.fin)) <= RTB_IK_PASS_LOW;
        if (first || (pass_slot && __any(st.fin != 0))) {
#else
        if (first || ((tick++ & ka->p.pass_mask) == 0 && __any(st.fin != 0))) {
#endif
            first = false;
            if constexpr (kStats) ++st_passes;
            unsigned long long busy = ws_get64(ws.busy), pool_next = ws_get64(ws.pool_next), pool_end = ws_get64(ws.pool_end), pool_live = ws_get64(ws.pool_live);
            unsigned long long pend_item = ws_get64(ws.pend_item);
            const unsigned long long NN = ws_get64(ws.NN);
            const unsigned pend_tick = ws_get32(ws.pend_tick);
            const unsigned wsf = ws_get32(ws.flags);
            bool exhausted = (wsf & kWsExhausted) != 0, drained = (wsf & kWsDrained) != 0, evidence = (wsf & kWsEvidence) != 0, c0_out = (wsf & kWsC0Out) != 0;
            const RTB_CONST IkDev &p = ka->p;      // shadows the by-value arguments for the whole pass
            const RTB_CONST double *qlim = (const RTB_CONST double *)ka->qlim;
            const double *Tep = ka->Tep, *q0 = ka->q0;
            unsigned long long *counter = ka->counter;
            double *q_out = ka->q_out, *residual = ka->residual;
            int32_t *success = ka->success, *iters = ka->iters, *searches = ka->searches;
            const IkWork *work = ka->work;
            ik_report<NJ>(st, sh, residual, p, qlim, ik_lds_q(sh, lane));              // phase A
            __syncthreads();
            if ((busy >> lane) & 1ull) ik_account(lane, sh);                            // phase B
            if (flat) {
                // a later-chunk item whose target has meanwhile succeeded in an EARLIER chunk (on whichever wave) is dropped
                const bool chk = ((busy >> lane) & 1ull) && sh.chunk[lane] > 0;
                if (__any(chk)) {
                    const int32_t d = chk ? ik_aload(p.flat_done + sh.tgt[lane]) : kIkFlatNone;
                    if (chk && sh.res[lane] == 0 && d < (int32_t)sh.chunk[lane]) sh.res[lane] = 3;
                }
            }
            __syncthreads();
            ik_finalize<NJ>(st, sh, lane, p, qlim, q_out, success, iters, searches, residual);   // phase C
            if (flat && ((busy >> lane) & 1ull) && sh.res[lane] == 1) ik_flat_publish(p.flat_done, sh.tgt[lane], sh.chunk[lane]);
            if (flat && !evidence) evidence = __any(((busy >> lane) & 1ull) && sh.res[lane] == 2 && sh.chunk[lane] == 0);
            const unsigned long long freed = __ballot(((busy >> lane) & 1ull) && sh.res[lane] != 0);
            if (freed) quiet = 0;
            busy &= ~freed;
            __syncthreads();
            unsigned long long idle = __ballot(st.status == kIkIdle);
            const unsigned long long starved = __ballot(((busy >> lane) & 1ull) && ik_starved(lane, sh));
            if (starved) {                                                              // phase D0: continuations
                if ((starved >> lane) & 1ull) sh.list[ik_rank(starved, lane)] = lane;
                __syncthreads();
                const int r = ik_rank(idle, lane);
                if (((idle >> lane) & 1ull) && r < __popcll(starved)) {
                    const int slot = sh.list[r];
                    ik_start_spec<NJ>(st, sh, lane, p, qlim, slot, sh.next[slot], Tep, q0);
                }
                __syncthreads();
                idle = __ballot(st.status == kIkIdle);
            }
            // flat schedule, once the chunk-0 items are all out: the item numbers left are later chunks -- speculation for other waves' targets --
            // so the wave's own next searches (D2) come first and numbers are drawn (D1) only for the lanes still idle after that
            const bool late = flat && c0_out;
            for (int step = 0; step < 2; ++step) {
            if ((step == 0) != late) {
                // (flat schedule: later-chunk numbers are drawn only by waves that have seen a first chunk fail.  Where first chunks do not fail --
                // the ik_benchmark-notebook setting: every first search succeeds -- nobody draws them, nothing is started on speculation
                // that the whole batch contradicts, and the launch behaves like the plain schedule: 0.53 -> 0.71 ms per 1e5 targets without this)
                if ((!exhausted || pend_item != kIkNoItem) && idle) {                       // phase D1: fresh targets
                    const unsigned long long freeslots = ~busy;
                    // free slots >= idle lanes (every busy slot keeps a lane); the per-pass cap spreads a batch
                    // smaller than the grid's lane count evenly over the waves
                    int nf = __popcll(idle);
                    int cap = p.fresh_cap;
                    if (ka->count) {                    // compacted list: its size is only known here
                        const unsigned per_wave = ((unsigned)NN + gridDim.x - 1u) / gridDim.x;
                        cap = per_wave < 1u ? 1 : (per_wave > 64u ? 64 : (int)per_wave);
                    }
                    nf = nf > cap ? cap : nf;
                    unsigned long long base = 0;
                    long long nvalid = 0;
                    int flat_off = 0;
                    if (flat && !exhausted) {
                        // flat schedule: while chunk-0 numbers are being drawn, exactly as many as this pass starts (nothing is hoarded: every
                        // one of them is alive); afterwards 64 per draw, each lane looking at one (a later-chunk item whose target has already
                        // succeeded is dead on arrival), and up to four draws per pass while draws come back with nothing alive
                        if (c0_out) nf = __popcll(idle);
                        // (the gate is on DRAWING later-chunk numbers; numbers already drawn and alive are started whenever lanes are idle)
                        for (int round = 0; (round < 4 || busy == 0) && pool_live == 0 && !drained; ++round) {   // (a wave with nothing else to do keeps drawing)
                            if (!c0_out) {           // a look at the counter before drawing: a wave without evidence must not draw later-chunk numbers
                                unsigned long long peek = 0;
                                if (lane == 0) peek = ik_aload(counter);
                                const unsigned plo = __shfl((unsigned)(peek & 0xffffffffu), 0), phi = __shfl((unsigned)(peek >> 32), 0);
                                c0_out = (((unsigned long long)phi << 32) | plo) >= (unsigned long long)p.flat_n;
                            }
                            if (c0_out && !evidence) break;
                            const unsigned long long want = !c0_out ? (unsigned long long)nf : 64ull;
                            unsigned long long got = 0;
                            if (lane == 0) got = atomicAdd(counter, want);
                            const unsigned lo = __shfl((unsigned)(got & 0xffffffffu), 0);
                            const unsigned hi = __shfl((unsigned)(got >> 32), 0);
                            got = ((unsigned long long)hi << 32) | lo;
                            pool_next = got < NN ? got : NN;
                            if (got + want >= NN) drained = true;
                            if (got + want >= (unsigned long long)p.flat_n) c0_out = true;
                            const unsigned long long id = got + (unsigned long long)lane;
                            pool_live = __ballot((unsigned long long)lane < want && id < NN && ik_flat_live(p, (uint32_t)id));
                        }
                        nvalid = __popcll(pool_live);
                        nvalid = nvalid > nf ? nf : nvalid;
                        base = pool_next;
                        // the r-th idle lane takes the r-th live number: offsets through the scratch list
                        if ((pool_live >> lane) & 1ull) { const int k = ik_rank(pool_live, lane); if (k < nvalid) sh.list[k] = (uint8_t)lane; }
                        __syncthreads();
                        { const int r0 = ik_rank(idle, lane); flat_off = (((idle >> lane) & 1ull) && r0 < nvalid) ? (int)sh.list[r0] : 0; }
                        __syncthreads();
                        for (long long k = 0; k < nvalid; ++k) pool_live &= pool_live - 1ull;      // the nvalid lowest live numbers are taken
                        if (drained && pool_live == 0) exhausted = true;
                    } else if (!exhausted) {
                    // targets are reserved from the device-wide counter in chunks and handed out from the wave's
                    // own pool: the atomic's round trip (and its s_waitcnt) is paid once per chunk, not per pass
                    if (pool_next == pool_end) {
                        unsigned long long got = 0;
                        const unsigned long long chunk = p.pool_chunk > 0 ? (unsigned long long)p.pool_chunk : (unsigned long long)nf;
                        if (lane == 0) got = atomicAdd(counter, chunk);
                        const unsigned lo = __shfl((unsigned)(got & 0xffffffffu), 0);
                        const unsigned hi = __shfl((unsigned)(got >> 32), 0);
                        got = ((unsigned long long)hi << 32) | lo;
                        pool_next = got < NN ? got : NN;
                        pool_end = got + chunk < NN ? got + chunk : NN;
                        if (pool_end == NN) drained = true;      // the counter has passed N: this is the wave's last refill
                    }
                    base = pool_next;
                    nvalid = (long long)(pool_end - pool_next);
                    nvalid = nvalid > nf ? nf : nvalid;
                    pool_next += (unsigned long long)nvalid;
                    if (drained && pool_next == pool_end) exhausted = true;
                    }
                    IkWork pend = ik_unpack(pend_item);
                    if (exhausted && pend_item != kIkNoItem) {
                        // the range another wave cut off for this one (the row of its ticket): started like a fresh target
                        base = (unsigned long long)ik_item_row(share_of(ka), p.N, (int)(blockIdx.x % kIkQueues), pend_tick);
                        nvalid = 1;
                        pend_item = kIkNoItem;
                    }
                    if ((freeslots >> lane) & 1ull) sh.list[ik_rank(freeslots, lane)] = lane;
                    __syncthreads();
                    const int r = ik_rank(idle, lane);
                    int myslot = -1;
                    if (((idle >> lane) & 1ull) && r < nvalid) {
                        myslot = sh.list[r];
                        const int64_t v = flat ? (int64_t)base + flat_off : (int64_t)base + r;
                        IkWork w;
                        int chunk = 0;
                        if (flat) w = ik_flat_item(p, (uint32_t)v, &chunk);
                        else if (sharing && v >= p.N) w = pend;
                        else if (work) w = work[v];
                        else { w.tgt = (int32_t)v; w.s0 = (int16_t)ik_s_first(p); w.s1 = (int16_t)ik_s_last(p); }
                        ik_start_target<NJ>(st, sh, lane, p, qlim, myslot, v, w, Tep, q0);
                        sh.chunk[myslot] = (uint8_t)chunk;
                    }
                    if constexpr (kStats) st_items += (unsigned long long)nvalid;
                    busy |= __ballot(((freeslots >> lane) & 1ull) && ik_rank(freeslots, lane) < nvalid);
                    __syncthreads();
                    idle = __ballot(st.status == kIkIdle);
                }
            } else {
                if (idle && busy) {                                                         // phase D2: speculative searches
                    if ((busy >> lane) & 1ull) sh.list[ik_rank(busy, lane)] = lane;
                    __syncthreads();
                    const int nb = __popcll(busy);
                    int slot = 0, s = 0;
                    const bool mine = ((idle >> lane) & 1ull) && ik_pick(sh, ik_rank(idle, lane), nb, __popcll(idle), p.spec_policy, ik_s_first(p), slot, s);
                    __syncthreads();
                    if (mine) ik_start_spec<NJ>(st, sh, lane, p, qlim, slot, s, Tep, q0);
                    __syncthreads();
                    idle = __ballot(st.status == kIkIdle);
                }
            }
            }
            if (sharing && exhausted && busy) {                                         // phase D3: give work to waiting waves
                // only a wave that holds a range worth cutting looks at the control words at all
                const unsigned long long cand = __ballot(((busy >> lane) & 1ull) && ik_donatable(sh, lane, ik_s_first(p), (int)ka->share.after));
                if (cand) {
                    const IkShareCtl shc = share_of(ka);
                    unsigned long long x = 0;         // one vector load: lane g reads queue g's word
                    if (lane < kIkQueues) x = ik_aload(ik_queue_word(shc, lane));
                    const unsigned long long open = __ballot(lane < kIkQueues && ik_word_waiting(x) > 0 && ik_word_count(x) < shc.qlimit);
                    if (open) {
                        // the first queue with waiters at or after a start that differs from wave to wave and from pass to pass
                        const int start = (int)((blockIdx.x + tick) % kIkQueues);
                        const unsigned long long rot = ((open >> start) | (open << (kIkQueues - start))) & ((1ull << kIkQueues) - 1ull);
                        const int g = (start + __builtin_ctzll(rot)) % kIkQueues;
                        const unsigned long long xg = ((unsigned long long)__shfl((unsigned)(x >> 32), g) << 32) | __shfl((unsigned)(x & 0xffffffffull), g);
                        unsigned give = ik_word_waiting(xg);
                        const unsigned nc = (unsigned)__popcll(cand);
                        give = give > nc ? nc : give;
                        give = give > (unsigned)kIkGiveMax ? (unsigned)kIkGiveMax : give;
                        if (lane == 0) {              // the pass is wave-uniform here: lane 0 does the bookkeeping
                            const unsigned k0 = ik_word_count(ik_aadd(ik_queue_word(shc, g), (unsigned long long)give));
                            unsigned i = 0;
                            for (unsigned long long m = cand; m && i < give; m &= m - 1ull, ++i) ik_donate(shc, p.N, sh, __builtin_ctzll(m), g, k0 + i);
                        }
                        __syncthreads();
                    }
                }
            }
        
            // flat schedule: a wave without evidence leaves once the chunk-0 numbers are out.  Every later chunk that is NEEDED belongs to a target whose
            // first chunk failed, and the wave that saw that failure has evidence and stays until the counter is drained -- it draws what is left
            leave = 0;
            if (flat && busy == 0 && !evidence && pool_live == 0 && c0_out) leave = 1;
            else if (busy == 0 && exhausted && pend_item == kIkNoItem) leave = sharing ? 2 : 1;
            __syncthreads();                  // every lane has read the state this pass started from
            ws.busy = busy; ws.pool_next = pool_next; ws.pool_end = pool_end; ws.pool_live = pool_live; ws.pend_item = pend_item;
            ws.flags = (exhausted ? kWsExhausted : 0u) | (drained ? kWsDrained : 0u) | (evidence ? kWsEvidence : 0u) | (c0_out ? kWsC0Out : 0u);
            __syncthreads();
        }
        if (leave == 1) break;
        if (leave == 2) {
            leave = 0;
            // sharing: out of work -- take a ticket, then wait on this ticket's own word for a range or for the EXIT mark
            const IkShareCtl shc = share_of(ka);
            const int g = (int)(blockIdx.x % kIkQueues);
            unsigned t = 0, lo = 0, hi = 0;
            if (lane == 0) t = ik_ticket(shc, g);
            t = __shfl(t, 0);
            // the last ticket of the grid?  all queue words, twice (ik_device.h)
            unsigned long long x1 = 0, x2 = 0;
            if (lane < kIkQueues) x1 = ik_aload(ik_queue_word(shc, lane));
            unsigned w = lane < kIkQueues ? ik_word_waiting(x1) : 0u;
            for (int o = 32; o; o >>= 1) w += __shfl_xor(w, o);
            bool fin = false;
            if (w == shc.waves) {
                if (lane < kIkQueues) x2 = ik_aload(ik_queue_word(shc, lane));
                fin = !__any(x1 != x2);
            }
            if (fin) {                                // every wave is dry: let the others go
                for (int q = 0; q < kIkQueues; ++q) {
                    const unsigned long long xq = ((unsigned long long)__shfl((unsigned)(x1 >> 32), q) << 32) | __shfl((unsigned)(x1 & 0xffffffffull), q);
                    ik_release_queue(shc, q, xq, lane);
                }
                break;
            }
            if (lane == 0) {
                // ~1 us between looks at first, ~8 us later; bounded (about a second) so that a protocol error ends as missing
                // results, which the merge reports, and not as a hung GPU
                const unsigned long long *word = shc.wdyn + (size_t)g * shc.qcap + t;
                unsigned long long x = kIkNoItem;
                int nap = 1;
                for (int spin = 0; spin < (1 << 17); ++spin) {
                    x = ik_aload(word);
                    if (x != kIkNoItem) break;
                    for (int k = 0; k < nap; ++k) __builtin_amdgcn_s_sleep(32);
                    nap = nap < 8 ? nap + 1 : 8;
                }
                lo = (unsigned)(x & 0xffffffffull); hi = (unsigned)(x >> 32);
            }
            lo = __shfl(lo, 0); hi = __shfl(hi, 0);
            const unsigned long long x = ((unsigned long long)hi << 32) | lo;
            if (x == kIkNoItem || x == kIkExitItem) break;
            ws.pend_item = x; ws.pend_tick = t;
            __syncthreads();
            first = true;                             // run a scheduling pass now: it starts the range
            continue;
        }
        if (++quiet > patience) {
            // Watchdog (never expected to fire): no target of this wave was resolved for longer than any single
            // target can take.  Leave loudly recognisable outputs instead of whatever the buffers held:
            // success 0, searches / iterations -1, residual and q NaN for the wave's unresolved and unstarted targets.
            const double nan = __longlong_as_double(0x7ff8000000000000ll);
            double *q_out = ka->q_out, *residual = ka->residual;
            int32_t *success = ka->success, *iters = ka->iters, *searches = ka->searches;
            const unsigned long long busy = ws.busy, pool_next = ws.pool_next, pool_end = ws.pool_end;
            if ((busy >> lane) & 1ull) {
                const int64_t t = sh.vix[lane];
                for (int j = 0; j < NJ; ++j) q_out[t * NJ + j] = nan;
                success[t] = 0; iters[t] = -1; searches[t] = -1; residual[t] = nan;
            }
            for (unsigned long long t = pool_next + lane; t < pool_end; t += kWave) {
                for (int j = 0; j < NJ; ++j) q_out[t * NJ + j] = nan;
                success[t] = 0; iters[t] = -1; searches[t] = -1; residual[t] = nan;
            }
            break;
        }
        {
            // The chain / limit tables are loop-invariant, and LICM would hoist all ~100 scalar loads out of
            // this persistent loop into SGPRs that do not exist (184 spilled SGPRs, 670 v_readlane in the
            // first build).  Laundering the table pointers once per iteration keeps the loads inside it,
            // where the scalar cache serves them and the VALU never sees them.
            ConstChainIk cvi;
            cvi.seg = (const RTB_CONST DevSeg *)ka->dc.seg;
            cvi.jmeta = (const RTB_CONST int32_t *)ka->dc.jmeta;
            const RTB_CONST double *ql = (const RTB_CONST double *)ka->qlim;
            cvi.trig = (const RTB_CONST double *)kIkSincosTable;
            asm volatile("" : "+s"(cvi.seg), "+s"(cvi.jmeta), "+s"(ql), "+s"(cvi.trig));
            const int myslot = st.slot;
            if constexpr (kStats) { ++st_iters; st_lane += (unsigned long long)__popcll(__ballot(st.status == kIkRun && !st.fin)); }
#if RTB_IK_MASK_IDLE
            // lanes without a running search sit the iteration out with their EXEC bit cleared: the instruction stream is the same, but a
            // third of the lane slots (idle and parked lanes, r04_g_ik_occupancy.txt) no longer toggle operands -- on a power-limited part
            // that is clock (A/B: round 4 visit q)
            if (st.status == kIkRun && !st.fin)
#endif
            ik_iter<NJ, STEP, (AUX & kIkAuxUnitW) != 0 && STEP == 0, (AUX & kIkAuxPlain) != 0, SIG>(st, ka->p, cvi, ql, [&](int k) { return sh.Td[k][myslot]; }, ik_lds_q(sh, lane));
        }
    }
    if (kStats && ka->p.stats && lane == 0) {
        unsigned long long *o = ka->p.stats + (unsigned long long)kIkStatWords * blockIdx.x;
        o[0] = st_iters; o[1] = st_passes; o[2] = st_lane; o[3] = st_items;
        o[4] = __builtin_amdgcn_s_memtime() - st_t0; o[5] = __builtin_amdgcn_s_memrealtime() - st_r0;
    }
}

// ---- phased schedule: compaction and in-order merge of the work items (ik_device.h: ik_phases / ik_item_* / ik_merge_item)
__global__ __launch_bounds__(256) void k_ik_list_b(IkPhases ph, int64_t N, const int32_t *__restrict__ success, IkWork *__restrict__ wB,
                                                   unsigned *cntB)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= N || success[t]) return;
    wB[atomicAdd(cntB, 1u)] = ik_item_b(ph, t);
}

__global__ __launch_bounds__(256) void k_ik_merge_b(IkPhases ph, int n, const IkWork *__restrict__ wB, const unsigned *cntB, const double *vq,
                                                    const int32_t *vok, const int32_t *vit, const int32_t *vse, const double *vE, double *q_out,
                                                    int32_t *success, int32_t *iters, int32_t *searches, double *residual, IkWork *__restrict__ wC,
                                                    int32_t *__restrict__ ownC, unsigned *cntOwn, unsigned *cntC)
{
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= (int64_t)*cntB) return;
    const int64_t tgt = wB[v].tgt;
    if (ik_merge_item<0>(n, ph.c_chunks == 0, tgt, v, vq, vok, vit, vse, vE, q_out, success, iters, searches, residual)) return;
    const unsigned r = atomicAdd(cntOwn, 1u);
    ownC[r] = (int32_t)tgt;
    for (int c = 0; c < ph.c_chunks; ++c) wC[(size_t)r * ph.c_chunks + c] = ik_item_c(ph, tgt, c);
    atomicAdd(cntC, (unsigned)ph.c_chunks);
}

__global__ __launch_bounds__(256) void k_ik_merge_c(IkPhases ph, int n, const IkWork *__restrict__ wC, const int32_t *__restrict__ ownC,
                                                    const unsigned *cntOwn, const double *vq, const int32_t *vok, const int32_t *vit, const int32_t *vse,
                                                    const double *vE, double *q_out, int32_t *success, int32_t *iters, int32_t *searches,
                                                    double *residual)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= (int64_t)*cntOwn) return;
    const int64_t tgt = ownC[r];
    for (int c = 0; c < ph.c_chunks; ++c) {          // in search order: the first success wins, later items are discarded speculation
        const int64_t v = r * ph.c_chunks + c;
        if (ik_merge_item<0>(n, wC[v].s1 == ph.s_last, tgt, v, vq, vok, vit, vse, vE, q_out, success, iters, searches, residual)) break;
    }
}

__global__ __launch_bounds__(256) void k_ik_merge_chain(int64_t N, int n, const int32_t *__restrict__ link, const double *vq, const int32_t *vok,
                                                        const int32_t *vit, const int32_t *vse, const double *vE, double *q_out, int32_t *success,
                                                        int32_t *iters, int32_t *searches, double *residual)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < N) ik_merge_chain(n, t, link, vq, vok, vit, vse, vE, q_out, success, iters, searches, residual);
}

// what a flat-schedule call needs cleared before k_ik starts, in ONE launch: every target's "lowest chunk that has succeeded" word and the launch's
// work counter (two hipMemsetAsync before: 5.0 + 4.5 us of a 0.83 ms call, profiles/r05_w_ik_call_timeline.txt)
__global__ __launch_bounds__(256) void k_ik_flat_prep(int32_t *__restrict__ done, int64_t N, unsigned long long *ctr)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < N) done[i] = kIkFlatNone;
    if (i == 0) *ctr = 0ull;
}
__global__ __launch_bounds__(256) void k_ik_merge_flat(int64_t N, int n, int chunks, const double *vq, const int32_t *vok, const int32_t *vit, const int32_t *vse,
                                                       const double *vE, double *q_out, int32_t *success, int32_t *iters, int32_t *searches, double *residual)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < N) ik_merge_flat(n, chunks, N, t, vq, vok, vit, vse, vE, q_out, success, iters, searches, residual);
}

#if RTB_HOST_SIDE      // the launchers (the kernels above are also what jit.cpp hands to hipRTC, one instantiation at a time)
namespace {
int g_ik_flat = 1;        // flat schedule (ik_device.h): 0 never, 1 automatic (the batch is resident at once), 2 always (tests)
int g_ik_flat_l0 = 0;     // searches in a target's first chunk: 0 = automatic -- 8 while the batch is at most 1.5 items per lane of the grid, else 4 ...
int g_ik_flat_len = 0;    // ... and in every later one: 0 = automatic -- 12 from 0.75 items per lane up, else 8 (re-measured on the final iteration,
                          // sustained, three rounds, outputs bit-equal: 1e5 targets 0.9517 -> 0.9396 ms, 3e5 2.971 -> 2.889, 1e6 and the notebook
                          // setting unchanged; 2e4 targets 0.535 with 8 against 0.558 with 12: profiles/r04_ik_ab.txt).  The earlier sweep:  Round 4, SUSTAINED timing (scripts/ik_ab.py, profiles/r04_ik_ab.txt; round 3 had tuned these on
                          // 3-launch bursts, i.e. on the boost clock): 1e5 Panda targets, fresh share 100 %: 8/8 1.103 ms, 8/12 1.105, 10/10 1.105, 6/8 1.114,
                          // 4/8 1.138, 5/6 1.145; plain 1.36.  From ~2.5 items per lane up a short first chunk wins again (3e5 targets: 4/8 3.18, 8/8 3.29 ms)
int g_ik_donate_after = 3;   // sharing: failed searches of a target before its range may be cut (rtbhip_tune "ik_donate_after")
int g_ik_share = 0;       // cross-wave sharing of search ranges: 0 never, 1 automatic (batch resident at once), 2 always (tests)
int g_ik_phased = 0;      // 0 never (default: the CPU replay and the GPU both say it loses), 1 automatic, 2 always (tests)
int g_ik_spec_policy = 0;
int g_ik_fresh_pct = 100; // share of a wave's even part of the batch it may start per scheduling pass, in percent: what is left is drawn as lanes
                          // fall idle, so quick waves take more.  Sustained, flat 4/8, 1e5 Panda targets: 25 % 1.35 ms, 50 % 1.194 (the round-3
                          // default, chosen on burst timings of the plain schedule), 75 % 1.166, 90-100 % 1.138, 110 % 1.17 (with 8/8: 1.103 at
                          // 90-100 %, 1.138 at 110 %); 2e4 targets 0.795 -> 0.63, notebook setting 0.43 -> 0.38; no effect from ~2.6e5 targets up
                          // (the per-pass cap is 64 either way)
int g_ik_unit_we = 1;          // rtbhip_tune("ik_unit_we", 0): the weighted LM step even for a mask of ones (A/B)
int g_ik_plain = 1;            // rtbhip_tune("ik_plain", 0): the general FK + Jacobian walk even for an all-revolute chain without flips (A/B)
int g_ik_waves_per_cu = 8;
int g_ik_pass_mask = 3;   // measured on MI355X, 1e6 Panda targets: 8.20 (0) / 7.75 (1) / 7.66 (3) / 8.08 ms (7)
std::mutex g_ctr_mu;
std::map<int, unsigned long long *> g_ctr;     // per-device ring of work counters
std::atomic<unsigned> g_ctr_next{0};
constexpr int kCtrRing = 256;
}  // namespace

void ik_tune(const char *key, int value)
{
    if (std::string(key) == "ik_unit_we") g_ik_unit_we = value != 0;
    if (std::string(key) == "ik_plain") g_ik_plain = value != 0;
    if (std::string(key) == "ik_waves_per_cu") g_ik_waves_per_cu = value < 1 ? 1 : value;
    if (std::string(key) == "ik_flat") g_ik_flat = value < 0 ? 0 : (value > 2 ? 2 : value);
    if (std::string(key) == "ik_flat_l0") g_ik_flat_l0 = value < 0 ? 0 : value;      // 0 = automatic
    if (std::string(key) == "ik_flat_len") g_ik_flat_len = value < 0 ? 0 : value;      // 0 = automatic
    if (std::string(key) == "ik_pass_mask") g_ik_pass_mask = value < 0 ? 0 : value;
    if (std::string(key) == "ik_share") g_ik_share = value < 0 ? 0 : (value > 2 ? 2 : value);
    if (std::string(key) == "ik_phased") g_ik_phased = value < 0 ? 0 : (value > 2 ? 2 : value);
    if (std::string(key) == "ik_fresh_pct") g_ik_fresh_pct = value < 1 ? 1 : value;
    if (std::string(key) == "ik_sig") g_ik_sig = value != 0;
    if (std::string(key) == "ik_spec_policy") g_ik_spec_policy = value != 0;
    if (std::string(key) == "ik_donate_after") g_ik_donate_after = value < 0 ? 0 : value;
}

void ik_release_device_state()
{
    std::lock_guard<std::mutex> lk(g_ctr_mu);
    for (auto &kv : g_ctr) (void)hipFree(kv.second);
    g_ctr.clear();
}

// the per-device ring of fresh-target counters a launch draws from: allocated here or on the first rtbhip_ik_lm of a device
static int ik_counter_ring(unsigned long long **ring_out)
{
    int dev = 0;
    RTB_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_ctr_mu);
    auto it = g_ctr.find(dev);
    if (it == g_ctr.end()) {
        unsigned long long *ring = nullptr;
        RTB_HIP(hipMalloc((void **)&ring, kCtrRing * sizeof(unsigned long long)));
        g_ctr[dev] = ring;
        *ring_out = ring;
    } else {
        *ring_out = it->second;
    }
    return RTBHIP_OK;
}

int ik_prepare_device()
{
    unsigned long long *ring = nullptr;
    return ik_counter_ring(&ring);
}

void ik_restart_host(const Chain *c, uint64_t seed, int64_t target, int draw, double *q_n)
{
    const int n = c->n;
    for (int j = 0; j < n; ++j) {
        const double lo = c->qlim[j], hi = c->qlim[n + j];
        q_n[j] = lo + ik_uniform(seed, target, draw, j) * (hi - lo);
    }
}

// Signatures with an instantiation built into the library (kIkSig*); every other plain chain of up to 8 joints gets its own at run time (jit.cpp).
static bool ik_sig_builtin(int n, SegSig sig) { return jit_builtin_enabled() && ((n == 7 && (sig == kIkSigPandaETS || sig == kIkSigPandaURDF)) || (n == 6 && sig == kIkSigUR)); }
static std::string ik_jit_expr(int n, bool flat, SegSig sig)
{
    return "rtbhip::k_ik<" + std::to_string(n) + ", 0, " + std::to_string((flat ? kIkAuxFlat : 0) | kIkAuxUnitW | kIkAuxPlain) + ", " + jit_hex(sig) + ">";
}
static bool chain_is_plain(const Chain *c)       // every joint revolute, none flipped: the straight-line walk (and with it a signature) applies
{
    for (int j = 0; j < c->n; ++j) if (jm_prismatic(c->jmeta[j]) || jm_flip(c->jmeta[j])) return false;
    return c->n >= 1;
}
std::vector<std::string> ik_jit_names(const Chain *c)
{
    std::vector<std::string> out;
    const SegSig sig = chain_signature(c->jmeta.data(), c->n);
    if (!sig || c->n > kRegMaxJoints || !chain_is_plain(c) || ik_sig_builtin(c->n, sig)) return out;
    out.push_back(ik_jit_expr(c->n, true, sig));
    out.push_back(ik_jit_expr(c->n, false, sig));
    return out;
}

template <int NJ>
static void launch_nj(const Chain *c, dim3 grid, hipStream_t s, const IkDev &p, const DevChain &dc, const double *qlim, const double *Tep,
                      const double *q0, unsigned long long *ctr, double *q_out, int32_t *success, int32_t *iters,
                      int32_t *searches, double *residual, const IkWork *work, const unsigned *count, const IkShareCtl &share, SegSig chain_sig)
{
    const int v = ik_step_variant(p, NJ);
    if constexpr (NJ >= 6 && NJ <= kIkNullMax) {        // the null-space variants exist for 6..12 joints (launch_ik checks)
        if (v == 2) { hipLaunchKernelGGL((k_ik<NJ, 2>), grid, dim3(kWave), 0, s, p, dc, qlim, Tep, q0, ctr, q_out, success, iters, searches, residual, work, count, share); return; }
        if (v == 3) { hipLaunchKernelGGL((k_ik<NJ, 3>), grid, dim3(kWave), 0, s, p, dc, qlim, Tep, q0, ctr, q_out, success, iters, searches, residual, work, count, share); return; }
    }
    if (v & kIkStepPinv) { hipLaunchKernelGGL((k_ik<NJ, 1>), grid, dim3(kWave), 0, s, p, dc, qlim, Tep, q0, ctr, q_out, success, iters, searches, residual, work, count, share); return; }
    const bool flat = p.flat_chunks > 0, stats = p.stats != nullptr;       // launch_ik offers these only where they are instantiated (ik_aux_served)
    if constexpr (NJ <= kRegMaxJoints) {
        // the default mask (all ones) has its own instantiations for the arms the register-resident kernel serves: no products with the weights
        if (p.unit_we && !stats) {
            if (p.pad_we /* plain chain */ && g_ik_plain) {
                // a known robot: the walk specialised to its constants' structure
#define RTB_IK_SIG_LAUNCH(SIG)                                                                                                                          \
    if (g_ik_sig && chain_sig == SIG && ik_sig_builtin(NJ, SIG)) {                                                                                      \
        if (flat) hipLaunchKernelGGL((k_ik<NJ, 0, kIkAuxFlat | kIkAuxUnitW | kIkAuxPlain, SIG>), grid, dim3(kWave), 0, s, p, dc, qlim, Tep, q0, ctr,   \
                                     q_out, success, iters, searches, residual, work, count, share);                                                   \
        else hipLaunchKernelGGL((k_ik<NJ, 0, kIkAuxUnitW | kIkAuxPlain, SIG>), grid, dim3(kWave), 0, s, p, dc, qlim, Tep, q0, ctr, q_out, success,     \
                                iters, searches, residual, work, count, share);                                                                        \
        return;                                                                                                                                         \
    }
                if constexpr (NJ == 7) {
                    RTB_IK_SIG_LAUNCH(kIkSigPandaETS)
                    RTB_IK_SIG_LAUNCH(kIkSigPandaURDF)
                }
                if constexpr (NJ == 6) {
                    RTB_IK_SIG_LAUNCH(kIkSigUR)
                }
#undef RTB_IK_SIG_LAUNCH
                // any other robot: its own instantiation of the same kernel, compiled at run time; the general walk below serves until it is there
                if (g_ik_sig && chain_sig && !ik_sig_builtin(NJ, chain_sig) && jit_enabled()) {
                    if (hipFunction_t f = c->jit.get("ik_kernels.hip", flat ? 1 : 0, [&] { return ik_jit_expr(NJ, flat, chain_sig); })) {
                        void *args[] = {(void *)&p, (void *)&dc, &qlim, &Tep, &q0, &ctr, &q_out, &success, &iters, &searches, &residual, &work, &count, (void *)&share};
                        (void)jit_launch(f, grid, dim3(kWave), 0, s, args);
                        return;
                    }
                }
                if (flat) hipLaunchKernelGGL((k_ik<NJ, 0, kIkAuxFlat | kIkAuxUnitW | kIkAuxPlain>), grid, dim3(kWave), 0, s, p, dc, qlim, Tep, q0, ctr, q_out, success, iters, searches, residual, work, count, share);
                else hipLaunchKernelGGL((k_ik<NJ, 0, kIkAuxUnitW | kIkAuxPlain>), grid, dim3(kWave), 0, s, p, dc, qlim, Tep, q0, ctr, q_out, success, iters, searches, residual, work, count, share);
                return;
            }
            if (flat) hipLaunchKernelGGL((k_ik<NJ, 0, kIkAuxFlat | kIkAuxUnitW>), grid, dim3(kWave), 0, s, p, dc, qlim, Tep, q0, ctr, q_out, success, iters, searches, residual, work, count, share);
            else hipLaunchKernelGGL((k_ik<NJ, 0, kIkAuxUnitW>), grid, dim3(kWave), 0, s, p, dc, qlim, Tep, q0, ctr, q_out, success, iters, searches, residual, work, count, share);
            return;
        }
        if (flat && !stats) { hipLaunchKernelGGL((k_ik<NJ, 0, kIkAuxFlat>), grid, dim3(kWave), 0, s, p, dc, qlim, Tep, q0, ctr, q_out, success, iters, searches, residual, work, count, share); return; }
    }
    if constexpr (NJ == 7) {                              // the counters: the benchmark's arm only
        // ... on the very instantiation that serves config 3 (the Panda's signature, unit mask): lane utilisation and the effective clock on the bench line
        // (benchsecondary.py: ik_loss_factors) are then those of the kernel that is timed, not of the general one
        if (stats && p.unit_we && p.pad_we && g_ik_plain && g_ik_sig && chain_sig == kIkSigPandaETS && ik_sig_builtin(NJ, chain_sig)) {
            if (flat) hipLaunchKernelGGL((k_ik<NJ, 0, kIkAuxFlat | kIkAuxStats | kIkAuxUnitW | kIkAuxPlain, kIkSigPandaETS>), grid, dim3(kWave), 0, s, p, dc, qlim, Tep, q0, ctr, q_out, success, iters, searches, residual, work, count, share);
            else hipLaunchKernelGGL((k_ik<NJ, 0, kIkAuxStats | kIkAuxUnitW | kIkAuxPlain, kIkSigPandaETS>), grid, dim3(kWave), 0, s, p, dc, qlim, Tep, q0, ctr, q_out, success, iters, searches, residual, work, count, share);
            return;
        }
        if (stats && flat) { hipLaunchKernelGGL((k_ik<NJ, 0, kIkAuxFlat | kIkAuxStats>), grid, dim3(kWave), 0, s, p, dc, qlim, Tep, q0, ctr, q_out, success, iters, searches, residual, work, count, share); return; }
        if (stats) { hipLaunchKernelGGL((k_ik<NJ, 0, kIkAuxStats>), grid, dim3(kWave), 0, s, p, dc, qlim, Tep, q0, ctr, q_out, success, iters, searches, residual, work, count, share); return; }
    }
    hipLaunchKernelGGL((k_ik<NJ, 0>), grid, dim3(kWave), 0, s, p, dc, qlim, Tep, q0, ctr, q_out, success, iters, searches, residual, work, count, share);
}

// where the flat schedule / the counters exist as instantiations: plain LM-family steps (STEP 0), chains of up to 8 joints / the 7-joint arm
static bool ik_aux_served(const IkDev &p, int n, bool stats) { return ik_step_variant(p, n) == 0 && (stats ? n == 7 : n <= kRegMaxJoints); }

// What this build serves: limits of the device scheduler and of the step variants that are instantiated.  Arguments only -- api.cpp asks
// BEFORE it touches the device (a refusal must not depend on a GPU being there), launch_ik asks again for callers that come straight to it.
int ik_check_limits(const Chain *c, const IkParams &ip, int64_t N)
{
    if (ip.slimit > kIkMaxSlimit) { set_error("ik_lm: slimit above 32000 is not supported by the device scheduler"); return RTBHIP_ELIMIT; }
    if (N >= (1ll << 32)) { set_error("ik_lm: at most 2^32 - 1 targets per call"); return RTBHIP_ELIMIT; }
    if (ip.ilimit > kIkMaxIlimit) { set_error("ik_lm: ilimit above 16000 is not supported by the device scheduler"); return RTBHIP_ELIMIT; }
    // (chains of up to kIkMaxJoints joints have built-in kernels; longer ones -- up to RTBHIP_MAX_JOINTS -- get theirs at run time: launch_ik)
    if (c->n > RTBHIP_MAX_JOINTS) { set_error("ik_lm: more than RTBHIP_MAX_JOINTS joints"); return RTBHIP_ELIMIT; }
    for (int j = 0; j < c->n; ++j)
        if (jm_jq(c->jmeta[j]) != j) { set_error("ik_lm: jindex must equal the joint order (the reference's ik.cpp:57 adds dq in that order)"); return RTBHIP_EINVAL; }
    if (ip.method == 5 && (ip.km > 0.0 || ip.kq > 0.0) && c->n < 6) {
        // IK_QP (ik_device.h): the manipulability term needs J J^T invertible, and both live in the one-wave-per-SIMD step variants
        set_error("ik_qp: the manipulability term (km > 0) and the joint-limit rows (kq > 0) are built for chains of 6..12 joints");
        return RTBHIP_ELIMIT;
    }
    if (ip.method != 5 && ip.kq > 0.0 && c->n < 6) {
        // below 6 joints I - pinv(J) J vanishes only away from singularities; the reference still applies it there, so the
        // parameters are refused rather than silently dropped
        set_error("ik_lm: null-space terms (kq > 0) are built for chains of 6..12 joints");
        return RTBHIP_ELIMIT;
    }
    return RTBHIP_OK;
}

#define RTB_TRY_IK(expr) do { int _rc = (expr); if (_rc != RTBHIP_OK) return _rc; } while (0)
int launch_ik(const Chain *c, const DevChain &dc, const double *qlim, const double *Tep, int64_t N, const double *q0,
              const IkParams &ip, double *q_out, int32_t *success, int32_t *iters, int32_t *searches, double *residual,
              hipStream_t s)
{
    if (N == 0) return RTBHIP_OK;
    RTB_TRY_IK(ik_check_limits(c, ip, N));
    IkDev p;
    p.ilimit = ip.ilimit; p.slimit = ip.slimit; p.reject_jl = ip.reject_jl; p.method = ip.method;
    p.flavour = ip.flavour; p.has_q0 = q0 != nullptr; p.tol = ip.tol; p.lambda = ip.lambda;
    for (int k = 0; k < 6; ++k) p.we[k] = ip.we[k];
    p.unit_we = g_ik_unit_we; p.pad_we = 1;
    for (int k = 0; k < 6; ++k) p.unit_we = p.unit_we && p.we[k] == 1.0;
    for (int j = 0; j < c->n; ++j) p.pad_we = p.pad_we && !jm_prismatic(c->jmeta[j]) && !jm_flip(c->jmeta[j]);   // (launcher only) an all-revolute chain, no flips
    p.seed = ip.seed; p.target0 = ip.target0;
    p.N = N;
    p.kq = ip.kq; p.km = ip.km; p.ps = ip.ps; p.ks = ip.ks;
    for (int j = 0; j < RTBHIP_MAX_JOINTS; ++j) p.pi[j] = ip.pi[j];
    p.flat_chunks = 0; p.flat_l0 = 0; p.flat_len = 0; p.flat_n = 0; p.flat_done = nullptr; p.stats = nullptr;
    int dev = 0, cus = 0;
    RTB_HIP(hipGetDevice(&dev));
    if (device_cu_count(&cus) != RTBHIP_OK) return RTBHIP_EHIP;
    unsigned long long *ring = nullptr;
    RTB_TRY_IK(ik_counter_ring(&ring));
    // a batch smaller than the grid's lane count is spread over ALL the waves (fresh_cap targets per
    // wave and pass) instead of filling ceil(N/64) of them: every SIMD then holds its share of the tail
    const bool one_wave = c->n > kRegMaxJoints || (ik_step_variant(p, c->n) & kIkStepNull);   // 9..12 joints, null-space: one wave per SIMD
    const int64_t gmax = (int64_t)cus * (one_wave ? 4 : g_ik_waves_per_cu);
    const int n = c->n;
    const SegSig chain_sig = chain_signature(c->jmeta.data(), n);
    // one launch of the scheduler kernel over `items` work items (the targets themselves when work == NULL)
    unsigned long long *ctr_ready = nullptr;      // a counter the caller of run() has already taken from the ring AND cleared on the stream (flat schedule)
    auto take_counter = [&]() { return ring + (g_ctr_next.fetch_add(1) % kCtrRing); };
    auto run = [&](const IkDev &pp, int64_t items, const IkWork *work, const unsigned *count, double *qo, int32_t *ok, int32_t *it,
                   int32_t *se, double *E, IkShareCtl share = IkShareCtl(), int64_t first_items = -1) -> int {
        IkDev p2 = pp;
        int64_t g = gmax;
        if (!count && g > items) g = items;
        if (g < 1) g = 1;
        share.waves = (uint32_t)g;
        // (flat schedule: the per-pass cap spreads the chunk-0 items -- one per target -- not the whole item count)
        const int64_t cap_items = first_items > 0 ? first_items : items;
        const int64_t cap = ((cap_items + g - 1) / g * g_ik_fresh_pct + 99) / 100;
        p2.fresh_cap = cap > 64 ? 64 : (cap < 1 ? 1 : (int32_t)cap);
        p2.pass_mask = g_ik_pass_mask;
        p2.spec_policy = g_ik_spec_policy;
        // big batches reserve in chunks (one atomic round trip per 64 / 16 targets); batches of the order of the
        // grid's lane count reserve exactly what a pass starts, so no wave sits on targets another could run
        const int64_t lanes = g * kWave;
        p2.pool_chunk = count ? 0 : (items >= 8 * lanes ? 64 : (items >= 3 * lanes ? 16 : 0));
        p2.N = items;
        unsigned long long *ctr = ctr_ready ? ctr_ready : take_counter();
        if (!ctr_ready) RTB_HIP(hipMemsetAsync(ctr, 0, sizeof(unsigned long long), s));
        ctr_ready = nullptr;
        // diagnostics: RTBHIP_IK_STATS=<file> appends one JSON line per scheduler launch with the per-wave counters (loop iterations,
        // scheduling passes, lane-iterations spent on a running search, items started).  Synchronises the stream: not for timed runs.
        const char *stats_path = std::getenv("RTBHIP_IK_STATS");           // (read at every launch: a caller may switch it on for one call)
        unsigned long long *dstats = nullptr;
        if (stats_path && *stats_path && ik_aux_served(p2, n, true)) {
            RTB_HIP(hipMallocAsync((void **)&dstats, (size_t)g * kIkStatWords * sizeof(unsigned long long), s));
            RTB_HIP(hipMemsetAsync(dstats, 0, (size_t)g * kIkStatWords * sizeof(unsigned long long), s));
            p2.stats = dstats;
        }
        dim3 grid((unsigned)g);
        // a SIZE without a built-in instantiation -- a chain of 17 .. RTBHIP_MAX_JOINTS joints, the null-space / IK_QP step variants beyond 12 --
        // gets its kernel at run time: the same k_ik template, its joint count a template argument as ever, compiled by hipRTC on first use
        // (seconds to a minute, then a file read: jit.cpp).  The reference loops over any n (robot/IK.py:542-576, core/ik.cpp:19-75).
        const int variant = ik_step_variant(p2, n);
        if (n > kIkMaxJoints || ((variant == 2 || variant == 3) && n > kIkNullMax)) {
            const int V = (variant == 2 || variant == 3) ? variant : ((variant & kIkStepPinv) ? 1 : 0);
            hipFunction_t f = c->jit.get_wait("ik_kernels.hip", 16 + V, [&] { return "rtbhip::k_ik<" + std::to_string(n) + ", " + std::to_string(V) + ", 0, 0>"; });
            if (!f) return RTBHIP_ELIMIT;                       // (the reason is in rtbhip_last_error: no hipRTC on this box, or the compiler's message)
            void *args[] = {(void *)&p2, (void *)&dc, (void *)&qlim, (void *)&Tep, (void *)&q0, (void *)&ctr, (void *)&qo, (void *)&ok, (void *)&it, (void *)&se,
                            (void *)&E, (void *)&work, (void *)&count, (void *)&share};
            RTB_TRY_IK(jit_launch(f, grid, dim3(kWave), 0, s, args));
            note_launch((int)grid.x, kWave, 0);
            return RTBHIP_OK;
        }
        switch (n) {
        case 1: launch_nj<1>(c, grid, s, p2, dc, qlim, Tep, q0, ctr, qo, ok, it, se, E, work, count, share, chain_sig); break;
        case 2: launch_nj<2>(c, grid, s, p2, dc, qlim, Tep, q0, ctr, qo, ok, it, se, E, work, count, share, chain_sig); break;
        case 3: launch_nj<3>(c, grid, s, p2, dc, qlim, Tep, q0, ctr, qo, ok, it, se, E, work, count, share, chain_sig); break;
        case 4: launch_nj<4>(c, grid, s, p2, dc, qlim, Tep, q0, ctr, qo, ok, it, se, E, work, count, share, chain_sig); break;
        case 5: launch_nj<5>(c, grid, s, p2, dc, qlim, Tep, q0, ctr, qo, ok, it, se, E, work, count, share, chain_sig); break;
        case 6: launch_nj<6>(c, grid, s, p2, dc, qlim, Tep, q0, ctr, qo, ok, it, se, E, work, count, share, chain_sig); break;
        case 7: launch_nj<7>(c, grid, s, p2, dc, qlim, Tep, q0, ctr, qo, ok, it, se, E, work, count, share, chain_sig); break;
        case 8: launch_nj<8>(c, grid, s, p2, dc, qlim, Tep, q0, ctr, qo, ok, it, se, E, work, count, share, chain_sig); break;
        case 9: launch_nj<9>(c, grid, s, p2, dc, qlim, Tep, q0, ctr, qo, ok, it, se, E, work, count, share, chain_sig); break;
        case 10: launch_nj<10>(c, grid, s, p2, dc, qlim, Tep, q0, ctr, qo, ok, it, se, E, work, count, share, chain_sig); break;
        case 11: launch_nj<11>(c, grid, s, p2, dc, qlim, Tep, q0, ctr, qo, ok, it, se, E, work, count, share, chain_sig); break;
        case 12: launch_nj<12>(c, grid, s, p2, dc, qlim, Tep, q0, ctr, qo, ok, it, se, E, work, count, share, chain_sig); break;
        case 13: launch_nj<13>(c, grid, s, p2, dc, qlim, Tep, q0, ctr, qo, ok, it, se, E, work, count, share, chain_sig); break;
        case 14: launch_nj<14>(c, grid, s, p2, dc, qlim, Tep, q0, ctr, qo, ok, it, se, E, work, count, share, chain_sig); break;
        case 15: launch_nj<15>(c, grid, s, p2, dc, qlim, Tep, q0, ctr, qo, ok, it, se, E, work, count, share, chain_sig); break;
        default: launch_nj<16>(c, grid, s, p2, dc, qlim, Tep, q0, ctr, qo, ok, it, se, E, work, count, share, chain_sig); break;
        }
        note_launch((int)grid.x, kWave, 0);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "k_ik launch");
        if (dstats) {
            std::vector<unsigned long long> hs((size_t)g * kIkStatWords);
            RTB_HIP(hipMemcpyAsync(hs.data(), dstats, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            RTB_HIP(hipStreamSynchronize(s));
            (void)hipFreeAsync(dstats, s);
            if (FILE *f = std::fopen(stats_path, "a")) {
                std::fprintf(f, "{\"grid\": %lld, \"items\": %lld, \"flat_chunks\": %d, \"waves_per_cu\": %d, \"pass_mask\": %d, \"n\": %d, \"per_wave\": [",
                             (long long)g, (long long)items, (int)p2.flat_chunks, g_ik_waves_per_cu, g_ik_pass_mask, n);
                for (int64_t w = 0; w < g; ++w)
                    std::fprintf(f, "%s[%llu,%llu,%llu,%llu,%llu,%llu]", w ? "," : "", hs[6 * w], hs[6 * w + 1], hs[6 * w + 2], hs[6 * w + 3], hs[6 * w + 4], hs[6 * w + 5]);
                std::fprintf(f, "]}\n");
                std::fclose(f);
            }
        }
        return RTBHIP_OK;
    };

    // Flat schedule (ik_device.h): one launch over (target, chunk) items drawn chunk-major from the device-wide counter; rows go to
    // temporaries, a merge kernel folds each target's rows in chunk order.  rtbhip_tune("ik_flat", 0 / 1 / 2) = never / automatic (the
    // batch is resident at once: the regime in which a wave is stuck with the targets it drew) / always (tests).
    {
        const int l0_auto = 2 * N <= 3 * gmax * kWave ? 8 : 4;
        const int len_auto = 4 * (long long)N >= 3ll * gmax * kWave ? 12 : 8;
        const IkFlatPlan fp = ik_flat_plan(p, g_ik_flat_l0 > 0 ? g_ik_flat_l0 : l0_auto, g_ik_flat_len > 0 ? g_ik_flat_len : len_auto);
        const bool flat_fits = fp.chunks > 1 && fp.chunks < 250 && (long long)N * fp.chunks < (1ll << 32) - 4096 && ik_aux_served(p, n, false);
        // automatic: the batch is resident at once (a wave cannot trade targets) AND large enough that waves hold several targets each -- below
        // that a wave's 64 lanes already serve its one or two targets' searches in parallel and the temporaries would only add latency
        // ... AND converged searches can still be rejected (joint limits): that is what makes first chunks fail (Panda defaults: a search succeeds
        // with p = 0.27, 28 % of the targets fail their first four).  Without the rejection practically every first search succeeds, no later
        // chunk is ever needed, and the few waves that do see a failed first chunk would have to drain the whole dead item range by themselves
        // (ik_benchmark-notebook setting, 1e5 targets: 0.60 ms plain, 0.98 flat) -- the plain schedule is the right one there
        const bool flat_on = flat_fits && (g_ik_flat == 2 || (g_ik_flat == 1 && N <= 3 * gmax * kWave && N >= 4 * gmax && p.reject_jl != 0));
        if (flat_on) {
            const size_t rows = (size_t)N * fp.chunks;
            auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
            const size_t o_done = 0, o_vq = up((size_t)N * sizeof(int32_t)), o_vE = o_vq + up(rows * n * sizeof(double)), o_vok = o_vE + up(rows * sizeof(double));
            const size_t o_vit = o_vok + up(rows * sizeof(int32_t)), o_vse = o_vit + up(rows * sizeof(int32_t)), total = o_vse + up(rows * sizeof(int32_t));
            char *blk = nullptr;
            {
                if (pool_keep_cached() != RTBHIP_OK) (void)hipGetLastError();       // the rows' block stays in the pool between calls
                hipError_t e = hipMallocAsync((void **)&blk, total, s);
                if (e != hipSuccess) return hip_fail(e, "hipMallocAsync (ik flat schedule)");
            }
            int rc = RTBHIP_OK;
            {
                unsigned long long *c0 = take_counter();
                hipLaunchKernelGGL(k_ik_flat_prep, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, (int32_t *)(blk + o_done), (int64_t)N, c0);
                hipError_t e = hipGetLastError();
                if (e != hipSuccess) rc = hip_fail(e, "k_ik_flat_prep launch");
                else ctr_ready = c0;
            }
            int32_t *vok = (int32_t *)(blk + o_vok), *vit = (int32_t *)(blk + o_vit), *vse = (int32_t *)(blk + o_vse);
            double *vq = (double *)(blk + o_vq), *vE = (double *)(blk + o_vE);
            if (rc == RTBHIP_OK) {
                IkDev pf = p;
                pf.flat_chunks = fp.chunks; pf.flat_l0 = fp.l0; pf.flat_len = fp.len; pf.flat_n = (uint32_t)N; pf.flat_done = (int32_t *)(blk + o_done);
                rc = run(pf, (int64_t)rows, nullptr, nullptr, vq, vok, vit, vse, vE, IkShareCtl(), N);
            }
            if (rc == RTBHIP_OK) {
                hipLaunchKernelGGL(k_ik_merge_flat, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, N, n, fp.chunks, vq, vok, vit, vse, vE, q_out, success,
                                   iters, searches, residual);
                hipError_t e = hipGetLastError();
                if (e != hipSuccess) rc = hip_fail(e, "k_ik_merge_flat launch");
            }
            (void)hipFreeAsync(blk, s);
            return rc;
        }
    }

    // Cross-wave sharing of search ranges (ik_device.h) when the whole batch is resident at once: rtbhip_tune("ik_share",
    // 0 / 1 / 2) = never / automatic / always (tests).  Rows go to temporaries; a merge kernel walks each target's chain.
    // (the protocol's termination needs every wave of the grid resident at once: the grid is at most 8 single-wave workgroups per CU)
    const bool share_fits = g_ik_waves_per_cu <= 8 && N <= (1 << 24);
    const bool share_on = share_fits && (g_ik_share == 2 || (g_ik_share == 1 && N <= 4 * gmax * kWave &&
                                                              ik_s_last(p) - ik_s_first(p) + 1 >= 2 * kIkDonateMin));
    if (share_on) {
        const int64_t g = gmax > N ? N : gmax;
        // every queue takes up to qlimit items; its table also has a word for every ticket beyond them (one per wave of the queue)
        // and for what donors racing past the limit may add (kIkGiveMax each)
        const size_t qlimit = (size_t)(N < 65536 ? N : 65536) / 8 + 256, qcap = qlimit + (size_t)(kIkGiveMax + 1) * (size_t)g + 64;
        const size_t M = (size_t)kIkQueues * qcap, rows = (size_t)N + M, ctl_bytes = (size_t)kIkQueues * kIkQueueStride * sizeof(unsigned long long);
        // one stream-ordered allocation (the device pool keeps it cached between calls), carved up; two fills: the control words
        // to zero, item table + links (adjacent) to all-ones (= "no item yet" / "end of chain")
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t o_ctl = 0, o_wdyn = o_ctl + up(ctl_bytes), o_link = o_wdyn + M * sizeof(unsigned long long);
        const size_t o_vq = up(o_link + rows * sizeof(int32_t)), o_vE = o_vq + up(rows * n * sizeof(double)), o_vok = o_vE + up(rows * sizeof(double));
        const size_t o_vit = o_vok + up(rows * sizeof(int32_t)), o_vse = o_vit + up(rows * sizeof(int32_t)), total = o_vse + up(rows * sizeof(int32_t));
        char *blk = nullptr;
        {
            hipError_t e = hipMallocAsync((void **)&blk, total, s);
            if (e != hipSuccess) return hip_fail(e, "hipMallocAsync (ik sharing)");
        }
        int rc = RTBHIP_OK;
        {
            hipError_t e = hipMemsetAsync(blk + o_ctl, 0, ctl_bytes, s);
            if (e == hipSuccess) e = hipMemsetAsync(blk + o_wdyn, 0xFF, o_link + rows * sizeof(int32_t) - o_wdyn, s);
            if (e != hipSuccess) rc = hip_fail(e, "hipMemsetAsync (ik sharing)");
        }
        int32_t *link = (int32_t *)(blk + o_link), *vok = (int32_t *)(blk + o_vok), *vit = (int32_t *)(blk + o_vit), *vse = (int32_t *)(blk + o_vse);
        double *vq = (double *)(blk + o_vq), *vE = (double *)(blk + o_vE);
        if (rc == RTBHIP_OK) {
            IkShareCtl sc;
            sc.tc = (unsigned long long *)(blk + o_ctl); sc.wdyn = (unsigned long long *)(blk + o_wdyn); sc.link = link;
            sc.qlimit = (uint32_t)qlimit; sc.qcap = (uint32_t)qcap; sc.after = (uint32_t)g_ik_donate_after; sc.waves = 0;      // waves: set by run()
            rc = run(p, N, nullptr, nullptr, vq, vok, vit, vse, vE, sc);
        }
        if (rc == RTBHIP_OK) {
            hipLaunchKernelGGL(k_ik_merge_chain, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, N, n, link, vq, vok, vit, vse, vE, q_out, success, iters,
                               searches, residual);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) rc = hip_fail(e, "k_ik_merge_chain launch");
        }
        (void)hipFreeAsync(blk, s);
        return rc;
    }

    // Phased schedule (ik_device.h) when the whole batch is resident at once and the search range is long enough to split;
    // rtbhip_tune("ik_phased", 0 / 1 / 2) = never / automatic / always (tests).
    const IkPhases ph = ik_phases(p);
    const bool phased = g_ik_phased == 2 ? ph.b_last > ph.a_last
                                         : (g_ik_phased == 1 && ph.b_last > ph.a_last && N <= 4 * gmax * kWave && N <= (1 << 24));
    if (!phased) return run(p, N, nullptr, nullptr, q_out, success, iters, searches, residual);

    // ---- phase A: the first searches of every target, results straight into the caller's arrays
    IkDev pa = p;
    pa.slimit = p.flavour == 0 ? ph.a_last : ph.a_last + 1;
    {
        const int rca = run(pa, N, nullptr, nullptr, q_out, success, iters, searches, residual);
        if (rca != RTBHIP_OK) return rca;
    }
    // stream-ordered temporaries (kept cached by the device pool): work lists, their counters, item result rows
    const int kc = ph.c_chunks > 0 ? ph.c_chunks : 1;
    const size_t nB = (size_t)N, nC = (size_t)N * kc;       // worst case: nothing resolves
    // (lists and rows are sized for that worst case -- nothing can overflow; the automatic mode only phases batches of at most
    // 4 x the grid's lanes, so this is a few hundred MB at the very most and normally a few MB are touched)
    unsigned *cnt = nullptr; IkWork *wB = nullptr, *wC = nullptr; int32_t *ownC = nullptr;
    double *vq = nullptr, *vE = nullptr; int32_t *vok = nullptr, *vit = nullptr, *vse = nullptr;
    const size_t rows = nB > nC ? nB : nC;
    auto alloc = [&](void **ptr, size_t bytes) -> int {
        hipError_t e = hipMallocAsync(ptr, bytes, s);
        return e == hipSuccess ? RTBHIP_OK : hip_fail(e, "hipMallocAsync (ik phases)");
    };
    int rc = alloc((void **)&cnt, 4 * sizeof(unsigned));
    if (rc == RTBHIP_OK) rc = alloc((void **)&wB, nB * sizeof(IkWork));
    if (rc == RTBHIP_OK) rc = alloc((void **)&wC, nC * sizeof(IkWork));
    if (rc == RTBHIP_OK) rc = alloc((void **)&ownC, nB * sizeof(int32_t));
    if (rc == RTBHIP_OK) rc = alloc((void **)&vq, rows * n * sizeof(double));
    if (rc == RTBHIP_OK) rc = alloc((void **)&vE, rows * sizeof(double));
    if (rc == RTBHIP_OK) rc = alloc((void **)&vok, rows * sizeof(int32_t));
    if (rc == RTBHIP_OK) rc = alloc((void **)&vit, rows * sizeof(int32_t));
    if (rc == RTBHIP_OK) rc = alloc((void **)&vse, rows * sizeof(int32_t));
    const unsigned blocks = (unsigned)((N + 255) / 256);
    if (rc == RTBHIP_OK) {
        hipError_t e = hipMemsetAsync(cnt, 0, 4 * sizeof(unsigned), s);
        if (e != hipSuccess) rc = hip_fail(e, "hipMemsetAsync (ik phases)");
    }
    if (rc == RTBHIP_OK) {
        // ---- phase B: unresolved targets -> one item each for the next searches
        hipLaunchKernelGGL(k_ik_list_b, dim3(blocks), dim3(256), 0, s, ph, N, success, wB, cnt + 0);
        rc = run(p, N, wB, cnt + 0, vq, vok, vit, vse, vE);
    }
    if (rc == RTBHIP_OK) {
        // merge B; the targets still unresolved get their phase-C items (c_chunks consecutive rows each)
        hipLaunchKernelGGL(k_ik_merge_b, dim3(blocks), dim3(256), 0, s, ph, n, wB, cnt + 0, vq, vok, vit, vse, vE, q_out, success, iters, searches,
                           residual, wC, ownC, cnt + 1, cnt + 2);
        if (ph.c_chunks > 0) {
            rc = run(p, (int64_t)nC, wC, cnt + 2, vq, vok, vit, vse, vE);
            if (rc == RTBHIP_OK)
                hipLaunchKernelGGL(k_ik_merge_c, dim3(blocks), dim3(256), 0, s, ph, n, wC, ownC, cnt + 1, vq, vok, vit, vse, vE, q_out, success, iters,
                                   searches, residual);
        }
        hipError_t e = hipGetLastError();
        if (rc == RTBHIP_OK && e != hipSuccess) rc = hip_fail(e, "ik phase kernels");
    }
    for (void *ptr : {(void *)cnt, (void *)wB, (void *)wC, (void *)ownC, (void *)vq, (void *)vE, (void *)vok, (void *)vit, (void *)vse})
        if (ptr) (void)hipFreeAsync(ptr, s);
    return rc;
}

#endif  // RTB_HOST_SIDE

}  // namespace rtbhip
