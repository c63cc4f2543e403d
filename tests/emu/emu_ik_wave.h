// tests/emu/emu_ik_wave.h -- TEST INFRASTRUCTURE: replay of k_ik's wave-level scheduler (templates; instantiated per joint-count range).
#pragma once
#include "emu_common.h"

// Replays k_ik's wave-level driver (ik_kernels.hip) on the CPU: `waves` single-wave workgroups advanced
// round-robin, one scheduling pass + one LM iteration each per turn, sharing the fresh-target counter.
template <int NJ>
struct EmuWave {
    IkWaveSharedT<kIkMaxJoints> sh;
    IkLane<NJ> st[kWave];
    unsigned long long busy = 0;
    bool exhausted = false, first = true, done = false, drained = false;
    unsigned long long pool_next = 0, pool_end = 0;
    unsigned long long pool_live = 0;      // flat schedule: live, unstarted numbers of the last draw (bit k = number pool_next + k)
    bool evidence = false;                 // flat schedule: one of this wave's own chunk-0 items has failed
    bool c0_out = false;                   // flat schedule: the counter has passed the chunk-0 numbers
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
    const EmuChainIk cv = emu_chain_ik(c);
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
    static const bool trace = getenv("EMU_IK_TRACE") != nullptr;
    long long turn = 0;
    while (live > 0) {
        if (trace && (turn++ % 10) == 0) {          // every tenth turn: live waves, running lanes, busy slots by chunk class
            long long run = 0, slots0 = 0, slots1 = 0;
            for (auto &w : W) {
                if (w.done) continue;
                for (int l = 0; l < kWave; ++l) {
                    run += w.st[l].status == kIkRun && !w.st[l].fin;
                    if ((w.busy >> l) & 1ull) { if (w.sh.chunk[l] == 0) ++slots0; else ++slots1; }
                }
            }
            fprintf(stderr, "turn %lld live %d running_lanes %lld (%.2f) slots chunk0 %lld later %lld counter %llu\n", turn - 1, live, run, (double)run / (64.0 * waves), slots0, slots1, counter);
        }
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
                const bool flat = p.flat_chunks > 0;
                if (flat)                  // drop later-chunk items whose target has meanwhile succeeded in an earlier chunk
                    for (int l = 0; l < kWave; ++l)
                        if (((w.busy >> l) & 1ull) && w.sh.chunk[l] > 0 && w.sh.res[l] == 0 && ik_aload(p.flat_done + w.sh.tgt[l]) < (int32_t)w.sh.chunk[l]) w.sh.res[l] = 3;
                for (int l = 0; l < kWave; ++l) ik_finalize<NJ>(w.st[l], w.sh, l, p, qlim, q_out, success, iters, searches, residual);
                if (flat)
                    for (int l = 0; l < kWave; ++l)
                        if ((w.busy >> l) & 1ull) {
                            if (w.sh.res[l] == 1) ik_flat_publish(p.flat_done, w.sh.tgt[l], w.sh.chunk[l]);
                            if (w.sh.res[l] == 2 && w.sh.chunk[l] == 0) w.evidence = true;
                        }
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
                const bool late = flat && w.c0_out;      // the kernel's order of D1 / D2 (ik_kernels.hip)
                for (int step = 0; step < 2; ++step) {
                if ((step == 0) != late) {
                    if ((!w.exhausted || w.pend_item != kIkNoItem) && idle) {
                        const unsigned long long freeslots = ~w.busy;
                        int nf = __builtin_popcountll(idle);
                        nf = nf > p.fresh_cap ? p.fresh_cap : nf;
                        { static const int mb = getenv("EMU_IK_MAX_BUSY") ? atoi(getenv("EMU_IK_MAX_BUSY")) : 64;
                          const int room = mb - __builtin_popcountll(w.busy); nf = nf > room ? (room > 0 ? room : 0) : nf; }
                        unsigned long long base = 0;
                        long long nvalid = 0;
                        int flat_off[kWave] = {0};
                        if (flat && !w.exhausted) {
                            const unsigned long long NN = (unsigned long long)p.N;
                            if (w.c0_out) nf = __builtin_popcountll(idle);
                            for (int round = 0; (round < 4 || w.busy == 0) && w.pool_live == 0 && !w.drained; ++round) {
                                if (!w.c0_out) w.c0_out = counter >= (unsigned long long)p.flat_n;
                                if (w.c0_out && !w.evidence) break;
                                const unsigned long long want = !w.c0_out ? (unsigned long long)nf : 64ull;
                                const unsigned long long got = counter;
                                counter += want;
                                w.pool_next = got < NN ? got : NN;
                                if (got + want >= NN) w.drained = true;
                                if (got + want >= (unsigned long long)p.flat_n) w.c0_out = true;
                                w.pool_live = 0;
                                for (int l = 0; l < kWave; ++l) {
                                    const unsigned long long id = got + (unsigned long long)l;
                                    if ((unsigned long long)l < want && id < NN && ik_flat_live(p, (uint32_t)id)) w.pool_live |= 1ull << l;
                                }
                            }
                            nvalid = __builtin_popcountll(w.pool_live);
                            nvalid = nvalid > nf ? nf : nvalid;
                            base = w.pool_next;
                            uint8_t offs[kWave];
                            for (int l = 0; l < kWave; ++l) if ((w.pool_live >> l) & 1ull) { const int k = ik_rank(w.pool_live, l); if (k < nvalid) offs[k] = (uint8_t)l; }
                            for (int l = 0; l < kWave; ++l) { const int r0 = ik_rank(idle, l); flat_off[l] = (((idle >> l) & 1ull) && r0 < nvalid) ? (int)offs[r0] : 0; }
                            for (long long k = 0; k < nvalid; ++k) w.pool_live &= w.pool_live - 1ull;
                            if (w.drained && w.pool_live == 0) w.exhausted = true;
                        } else if (!w.exhausted) {
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
                                const int64_t v = flat ? (int64_t)base + flat_off[l] : (int64_t)base + r;
                                IkWork it;
                                int chunk = 0;
                                if (flat) it = ik_flat_item(p, (uint32_t)v, &chunk);
                                else if (share && v >= p.N) it = pend;
                                else if (work) it = work[v];
                                else { it.tgt = (int32_t)v; it.s0 = (int16_t)ik_s_first(p); it.s1 = (int16_t)ik_s_last(p); }
                                ik_start_target<NJ>(w.st[l], w.sh, l, p, qlim, w.sh.list[r], v, it, Tep, q0);
                                w.sh.chunk[w.sh.list[r]] = (uint8_t)chunk;
                            }
                        }
                        w.busy |= ballot(w, [&](int l) { return ((freeslots >> l) & 1ull) && ik_rank(freeslots, l) < nvalid; });
                        idle = ballot(w, [&](int l) { return w.st[l].status == kIkIdle; });
                    }
                } else {
                    if (idle && w.busy) {
                        for (int l = 0; l < kWave; ++l) if ((w.busy >> l) & 1ull) w.sh.list[ik_rank(w.busy, l)] = l;
                        const int nb = __builtin_popcountll(w.busy);
                        int slot[kWave], ss[kWave];
                        bool mine[kWave];
                        for (int l = 0; l < kWave; ++l)
                            mine[l] = ((idle >> l) & 1ull) && ik_pick(w.sh, ik_rank(idle, l), nb, __builtin_popcountll(idle), p.spec_policy, ik_s_first(p), slot[l], ss[l]);
                        for (int l = 0; l < kWave; ++l) if (mine[l]) ik_start_spec<NJ>(w.st[l], w.sh, l, p, qlim, slot[l], ss[l], Tep, q0);
                        idle = ballot(w, [&](int l) { return w.st[l].status == kIkIdle; });
                    }
                }
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
            if (p.flat_chunks > 0 && w.busy == 0 && !w.evidence && w.pool_live == 0 && w.c0_out) { w.done = true; --live; continue; }
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

// launch_ik's flat schedule (ik_kernels.hip) replayed: one run of the wave scheduler over the (target, chunk) items with the planning /
// item / liveness / merge functions of ik_device.h; rows in temporaries, merged chunk by chunk.
template <int NJ>
static int emu_ik_flat_run(const Chain *c, const IkDev &p, int waves, int l0, int len, const double *Tep, const double *q0, double *q_out,
                           int32_t *success, int32_t *iters, int32_t *searches, double *residual, double *stats)
{
    const IkFlatPlan fp = ik_flat_plan(p, l0, len);
    if (fp.chunks <= 1) return emu_ik_wave_run<NJ>(c, p, waves, Tep, q0, q_out, success, iters, searches, residual, stats);
    const int64_t N = p.N;
    const size_t rows = (size_t)N * fp.chunks;
    std::vector<double> vq(rows * NJ, 0.0), vE(rows, 0.0);
    std::vector<int32_t> vok(rows, -7), vit(rows, -7), vse(rows, -7), done((size_t)N, kIkFlatNone);
    IkDev pf = p;
    pf.flat_chunks = fp.chunks; pf.flat_l0 = fp.l0; pf.flat_len = fp.len; pf.flat_n = (uint32_t)N; pf.flat_done = done.data();
    pf.N = (int64_t)rows;
    const int rc = emu_ik_wave_run<NJ>(c, pf, waves, Tep, q0, vq.data(), vok.data(), vit.data(), vse.data(), vE.data(), stats);
    if (rc != 0) return rc;
    for (int64_t t = 0; t < N; ++t) ik_merge_flat(NJ, fp.chunks, N, t, vq.data(), vok.data(), vit.data(), vse.data(), vE.data(), q_out, success, iters, searches, residual);
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

// one dispatcher per joint-count range (emu_ik_wave_a/b/c.cpp)
int emu_ik_wave_lo(int n, bool shared, bool phased, const Chain *c, const IkDev &p, int waves, const double *Tep, const double *q0, double *q_out,
                   int32_t *success, int32_t *iters, int32_t *searches, double *residual, double *stats);
int emu_ik_wave_mid(int n, bool shared, bool phased, const Chain *c, const IkDev &p, int waves, const double *Tep, const double *q0, double *q_out,
                    int32_t *success, int32_t *iters, int32_t *searches, double *residual, double *stats);
int emu_ik_wave_hi(int n, bool shared, bool phased, const Chain *c, const IkDev &p, int waves, const double *Tep, const double *q0, double *q_out,
                   int32_t *success, int32_t *iters, int32_t *searches, double *residual, double *stats);
#define RTB_EMU_IK(NJ) case NJ: if (getenv("EMU_IK_FLAT") && atoi(getenv("EMU_IK_FLAT")) != 0) \
            return emu_ik_flat_run<NJ>(c, p, waves, getenv("EMU_IK_FLAT_L0") ? atoi(getenv("EMU_IK_FLAT_L0")) : 4, getenv("EMU_IK_FLAT_LEN") ? atoi(getenv("EMU_IK_FLAT_LEN")) : 8, \
                                       Tep, q0, q_out, success, iters, searches, residual, stats); \
        return shared ? emu_ik_shared_run<NJ>(c, p, waves, Tep, q0, q_out, success, iters, searches, residual, stats) \
                                     : phased ? emu_ik_phased_run<NJ>(c, p, waves, Tep, q0, q_out, success, iters, searches, residual, stats) \
                                              : emu_ik_wave_run<NJ>(c, p, waves, Tep, q0, q_out, success, iters, searches, residual, stats);
#define RTB_EMU_IK_DISPATCH(NAME, CASES) \
    int NAME(int n, bool shared, bool phased, const Chain *c, const IkDev &p, int waves, const double *Tep, const double *q0, double *q_out, \
             int32_t *success, int32_t *iters, int32_t *searches, double *residual, double *stats) \
    { switch (n) { CASES default: return -1; } }
