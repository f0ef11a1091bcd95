// sdrm_tune.hip -- the two tuners of a batch's schedule: the self-calibration at creation and the online refinement on the
// caller's own calls.  Neither touches a result: they choose between settings (clock-stage workgroup shape, front-end hold,
// companion grid) that every call is bit-identical under.  (The rest of the C-ABI: sdrm_batch.hip, sdrm_call.hip, sdrm_handle.hip.)
#include "sdrm_batch_impl.h"

// ---- self-calibration of the schedule ------------------------------------------------------------------------------
// Which clock-stage shape, whether the front-end holds back for the clock stage's placement and whether the clock stage gets
// a companion grid used to be decided by constants fitted on one box at one power state (channel-count thresholds, 18.4e12
// multiply-adds per second, 97 ns per symbol).  Those constants now only give the STARTING point: a batch of at least 32
// channels times its own pipeline at creation -- full-length calls on a synthetic row that every channel reads (input
// stride 0: no buffer of the batch's size is needed), a few calls per candidate setting (at least ~4 ms of them: short calls are launch-bound and noisy), one dimension after the other --
// keeps what was fastest by more than the noise, and then puts every stream back to its initial state.  Costs a few dozen
// calls (tens of milliseconds for 256 channels, a few hundred for 4096) once per batch.
// SDRM_AUTOTUNE=0 switches it off; SDRM_K3_LANES / SDRM_FRONT_HOLD / SDRM_K3_COMPANY pin their dimension as before.
// Round 6: a dimension is measured only where its rule is in doubt.  On the bench's five batches the calibration had confirmed the
// rules on four (88 - 376 ms each, `ms_per_call_before == ms_per_call_after`, BENCH_r05.json); the measurements of rounds 3-5
// say where the rules are safe: the clock stage's shape away from the two channel counts where it changes (1280, 2560: asked
// within a quarter of them), the front-end's hold outside 640 .. 1280 channels (it costs 2-3 % below, gains 2-5 % from 896 to
// 1024, profiles/r03_front_hold_bounds.txt), the companion grid where the front-end's estimate is under half the rule's
// threshold and the clock stage runs long (on), or beyond 1024 channels / a front-end well above the threshold / a clock stage
// too short to pay for the grid (off).  A batch with no open question is not timed at all.  SDRM_AUTOTUNE=2 asks every
// question regardless (measurements, and the regression test that needs a calibrated small batch).
int sdrm_calibrate(sdrm_batch_t *b, const sdrm_fsk_config *cfgs) {
    const size_t C = b->plan.design.size();
    const char *env = getenv("SDRM_AUTOTUNE");
    if (b->serial || C < 32 || (env != nullptr && atoi(env) == 0) || b->n_gen > 0 ||
        (b->flags & SDRM_FLAG_NO_CALIBRATION) != 0) {
        return 0;
    }
    const bool every = env != nullptr && atoi(env) >= 2;
    auto near = [&](size_t at) { return 4 * C >= 3 * at && 4 * C <= 5 * at; };
    const bool ask_shape = !sdrm::k3_shape_is_forced() && b->plan.clock_carried_max <= 128 && C >= 512 && (every || near(1280) || near(2560));
    const bool ask_hold = !sdrm::front_hold_is_forced() && C >= 256 && (every || (C >= 640 && C <= 1280));
    const bool company_on_for_sure = C <= 576 && b->est_front_ms < 0.35f * b->est_clock_ms && b->est_clock_ms >= 0.6f;
    const bool company_off_for_sure = C > 1024 || b->est_front_ms > 1.4f * b->est_clock_ms || b->est_clock_ms < 0.15f;
    const bool ask_company = getenv("SDRM_K3_COMPANY") == nullptr && C <= 2048 && (every || !(company_on_for_sure || company_off_for_sure));
    if (!ask_shape && !ask_hold && !ask_company) {
        return 0;  // the rules decide: nothing to time
    }
    uint32_t longest = 0;
    std::vector<size_t> lens(C);
    for (size_t c = 0; c < C; c++) {
        lens[c] = cfgs[c].max_input_buffer_length;
        longest = std::max(longest, cfgs[c].max_input_buffer_length);
    }
    if (longest < 1024) {
        return 0;  // calls this short are launch-bound whatever the schedule
    }
    // one row of plausible IQ: unit-amplitude FM of a slow square wave plus a little deterministic noise (finite, no zeros:
    // the discriminator stays on its short form, the clock loop on its finite one, as with real signals)
    std::vector<sdrm_f2> row(longest);
    uint32_t lcg = 12345u;
    double ph = 0.0;
    for (uint32_t i = 0; i < longest; i++) {
        ph += ((i / 5) % 7 < 3 ? 0.16 : -0.16);
        lcg = lcg * 1664525u + 1013904223u;
        const float n1 = (float) ((lcg >> 8) & 0xffff) / 65536.0f - 0.5f;
        lcg = lcg * 1664525u + 1013904223u;
        const float n2 = (float) ((lcg >> 8) & 0xffff) / 65536.0f - 0.5f;
        row[i].x = (float) cos(ph) + 0.1f * n1;
        row[i].y = (float) sin(ph) + 0.1f * n2;
    }
    sdrm_f2 *d_row = nullptr;
    if (hipMalloc((void **) &d_row, sizeof(sdrm_f2) * longest) != hipSuccess) {
        return 0;  // no room for the row: keep the starting point
    }
    int code = 0;
    auto t_start = std::chrono::steady_clock::now();
    if (hipMemcpy(d_row, row.data(), sizeof(sdrm_f2) * longest, hipMemcpyHostToDevice) != hipSuccess) {
        code = -EIO;
    }
    // ms per call of the batch as it is set up now: `warm` calls to fill the pipeline, then `timed` calls between two waits
    int timed = 5;  // raised below so that a measurement lasts >= ~4 ms: short calls are launch-bound and noisy
    auto measure = [&](double *ms) -> int {
        const int warm = 3;
        for (int k = 0; k < warm; k++) {
            int c2 = sdrm_impl::enqueue_call(b, d_row, 0, lens.data(), b->stream, nullptr, 0);
            if (c2 != 0) return c2;
        }
        int c2 = sdrm_impl::wait_for_all_calls(b);
        if (c2 != 0) return c2;
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < timed; k++) {
            c2 = sdrm_impl::enqueue_call(b, d_row, 0, lens.data(), b->stream, nullptr, 0);
            if (c2 != 0) return c2;
        }
        c2 = sdrm_impl::wait_for_all_calls(b);
        *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / timed;
        return c2;
    };
    double best = 0.0;
    code = code ? code : measure(&best);  // first pass: also pays the kernels' first-launch costs
    if (code == 0 && best > 0.0 && best * timed < 4.0) {
        timed = std::min(64, (int) ceil(4.0 / best));
    }
    code = code ? code : measure(&best);
    const double before = best;
    const double margin = 0.97;  // a candidate replaces the incumbent only when it is more than 3 % faster
    // (1) the clock stage's workgroup shape
    if (code == 0 && ask_shape) {
        const int shapes[3][3] = {{16, 1024, 0}, {32, 512, 0}, {64, 256, 1}};
        const sdrm_k3_shape cur = sdrm_k3_shape_for((int) C, 0, 0, 0, (int) b->plan.clock_carried_max);
        int keep[3] = {cur.lanes, cur.ring, cur.plain};
        for (const auto &sh : shapes) {
            if (sh[0] == cur.lanes && sh[1] == cur.ring && sh[2] == cur.plain) {
                continue;
            }
            b->dev.k3_lanes = sh[0];
            b->dev.k3_ring = sh[1];
            b->dev.k3_plain = sh[2];
            double ms = 0.0;
            code = measure(&ms);
            if (code != 0) {
                break;
            }
            if (ms < best * margin) {
                best = ms;
                keep[0] = sh[0];
                keep[1] = sh[1];
                keep[2] = sh[2];
            }
        }
        b->dev.k3_lanes = keep[0];
        b->dev.k3_ring = keep[1];
        b->dev.k3_plain = keep[2];
    }
    // (2) the front-end's hold for the clock stage's placement
    if (code == 0 && ask_hold) {
        const bool was = b->hold_front;
        b->hold_front = !was;
        double ms = 0.0;
        code = measure(&ms);
        if (code == 0 && ms < best * margin) {
            best = ms;
        } else {
            b->hold_front = was;
        }
    }
    // (3) the companion grid beside the clock stage
    if (code == 0 && ask_company) {
        const int was = b->company_blocks;
        b->company_blocks = was > 0 ? 0 : b->company_grid;
        double ms = 0.0;
        code = measure(&ms);
        if (code == 0 && ms < best * margin) {
            best = ms;
        } else {
            b->company_blocks = was;
        }
    }
    if (code == 0) {
        code = sdrm_impl::reset_all_streams(b);
    }
    (void) hipFree(d_row);
    b->calibrated = code == 0;
    b->calib_ms[0] = (float) before;
    b->calib_ms[1] = (float) best;
    b->calib_ms[2] = (float) std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    if (getenv("SDRM_AUTOTUNE_LOG") != nullptr) {
        const sdrm_k3_shape sh = sdrm_k3_shape_for((int) C, b->dev.k3_lanes, b->dev.k3_ring, b->dev.k3_plain, (int) b->plan.clock_carried_max);
        fprintf(stderr, "sdrmodem_hip: calibrated %zu channels in %.0f ms: %.3f -> %.3f ms per call; clock stage %dx%d%s, front hold %s, "
                        "companion grid %d\n", C, b->calib_ms[2], before, best, sh.lanes, sh.ring, sh.plain ? "p" : "",
                b->hold_front ? "on" : "off", b->company_blocks);
    }
    return code;
}

// ---- online refinement of the schedule for calls the calibration did not cover ---------------------------------------
// Why: the NCO stages are a fourth pipeline stage (a dependent chain as long as the clock stage's in BASELINE configs[4]'s mix);
// what the creation-time calibration found best without them (there: front hold on, companion grid on, -14 %) cost that
// workload 9 % (profiles/r04_config5_schedule.txt).  Calls of less than half the calibrated length are the second class it does
// not cover (the companion grid cost 4096-sample calls 14 % at 256 channels in round 3).  The streams hold the caller's state by
// then, so nothing can be replayed -- but front hold and companion grid may change between any two calls without touching a
// result.  So, on the caller's own calls, from the 17th call of such a class on:
//   1. the starting point's steady state: 40 calls, the median of the last 32 completion-to-completion intervals of the clock stage;
//   2. four settings for eight calls each (the median of the last five intervals), then the starting point and the winner again:
//      a winner must win both rounds by more than 3 %;
//   3. the winner's probation: 40 calls like (1); it stays only if its steady state beats (1) by more than 3 % -- a block of
//      eight calls can flatter a setting whose cost builds up over tens of calls (seen: 1.05 ms per call in its blocks, 2.9 in
//      the steady state, profiles/r04_online_refinement.txt);
//   4. a standing guard: every 64th call of the class starts a five-interval sample; two bad samples in a row (5 % behind (1)) give
//      the starting point back for good.
// Medians, because the host may stall between two calls (the HIP runtime grows its pools 6 ms at a time during a process's first
// dozens of calls) and the device then idles for reasons no setting is to blame for.  The winner serves calls of its class (same
// NCO flag, total length within a factor of two), other calls keep the calibrated setting.  Results do not depend on any of it.
#define SDRM_TUNE_SKIP 3   // calls of a block before its first timed completion (the pipeline holds three calls)
#define SDRM_TUNE_TIMED 5  // completion-to-completion intervals per block (ev[][TIMED + 1])
#define SDRM_WATCH_SKIP 8
#define SDRM_WATCH_TIMED 32
static void online_tune_apply(sdrm_batch_t *b, int cand) {
    b->hold_front = (cand & 1) ? !b->tune.base_hold : b->tune.base_hold;
    b->company_blocks = (cand & 2) ? (b->tune.base_company > 0 ? 0 : b->company_grid) : b->tune.base_company;
}
static void online_tune_settle(sdrm_batch_t *b, int cand) {
    online_tune_apply(b, cand);
    b->tune.chosen = cand;
    b->tune.state = 2;
    if (getenv("SDRM_AUTOTUNE_LOG") != nullptr) {
        const float *ms = b->tune.ms;
        fprintf(stderr, "sdrmodem_hip: refined online for calls %s NCO batches, %llu samples per call: steady %.3f ms per call; %.3f / %.3f / "
                        "%.3f / %.3f (as is, hold toggled, companion grid toggled, both), again %.3f as is / %.3f the winner, the winner's "
                        "steady state %.3f; front hold %s, companion grid %d\n",
                b->tune.nco ? "with" : "without", (unsigned long long) b->tune.sig, ms[6], ms[0], ms[1], ms[2], ms[3], ms[4], ms[5], ms[7],
                b->hold_front ? "on" : "off", b->company_blocks);
    }
}
// what the creation-time calibration measured: full-length calls without NCO batches
static uint64_t full_length_samples(const sdrm_batch_t *b) {
    uint64_t n = 0;
    for (const sdrm_chan_params &p : b->plan.params) {
        n += p.max_len;
    }
    return n;
}
// median of the intervals between n + 1 consecutive completion events
static bool median_interval(const hipEvent_t *ev, int n, float *out) {
    float iv[SDRM_WATCH_TIMED];
    for (int j = 0; j < n; j++) {
        if (hipEventElapsedTime(&iv[j], ev[j], ev[j + 1]) != hipSuccess) {
            return false;
        }
    }
    std::sort(iv, iv + n);
    *out = iv[n / 2];
    return true;
}
void sdrm_online_tune_before(sdrm_batch_t *b, bool with_nco, uint64_t sig) {
    sdrm_batch_t::OnlineTune &t = b->tune;
    if (t.state == 2) {
        if (t.chosen < 0) {
            return;  // never measured (switched off, forced, small batch): the batch's settings stand
        }
        // settled: the refined setting serves the class of calls it was measured on, the calibrated one everything else
        const bool alike = t.chosen > 0 && with_nco == t.nco && sig * 2 >= t.sig && sig <= t.sig * 2;
        if (t.guard_pending && hipEventQuery(t.ev[0][SDRM_TUNE_TIMED]) == hipSuccess) {
            t.guard_pending = false;
            float ms = 0.0f;
            if (t.chosen > 0 && median_interval(t.ev[0], SDRM_TUNE_TIMED, &ms) && ms > t.ms[6] * 1.05f) {
                if (++t.guard_bad >= 2) {
                    if (getenv("SDRM_AUTOTUNE_LOG") != nullptr) {
                        fprintf(stderr, "sdrmodem_hip: the refined setting fell behind (%.3f ms per call, the starting point's steady state was "
                                        "%.3f): the starting point is back\n", ms, t.ms[6]);
                    }
                    t.chosen = 0;
                }
            } else {
                t.guard_bad = 0;
            }
        }
        t.guard_alike = alike && sig == t.sig;
        online_tune_apply(b, alike ? t.chosen : 0);
        return;
    }
    if (t.state == 0) {
        // calls the calibration did not cover: Doppler correction (a fourth stage), or less than half its length -- or any call
        // of a batch that was created without calibration
        if (!with_nco && sig * 2 > full_length_samples(b) && (b->flags & SDRM_FLAG_NO_CALIBRATION) == 0) {
            return;
        }
        const char *env = getenv("SDRM_AUTOTUNE");  // read per batch, like the calibration does
        if (b->serial || b->plan.design.size() < 32 || (env != nullptr && atoi(env) == 0) || b->n_gen > 0 ||
            sdrm::front_hold_is_forced() || getenv("SDRM_K3_COMPANY") != nullptr) {
            t.state = 2;
            return;
        }
        if (b->calls < 16 || sig == 0) {
            return;
        }
        bool ok = true;
        for (auto &row : t.ev) {
            for (hipEvent_t &e : row) {
                ok = ok && (e != nullptr || hipEventCreate(&e) == hipSuccess);  // (a restarted measurement has them)
            }
        }
        for (auto &row : t.watch) {
            for (hipEvent_t &e : row) {
                ok = ok && (e != nullptr || hipEventCreate(&e) == hipSuccess);
            }
        }
        if (!ok) {
            t.state = 2;
            return;
        }
        t.base_hold = b->hold_front;
        t.base_company = b->company_blocks;
        t.sig = sig;
        t.nco = with_nco;
        t.phase = 1;
        t.cand = 0;
        t.n = 0;
        t.best = -1;
        t.idle = false;
        t.state = 1;
    }
    // a batcher's rounds differ by a client's buffer or two: calls within an eighth of the first one's length are one class
    const uint64_t apart = sig > t.sig ? sig - t.sig : t.sig - sig;
    if (with_nco != t.nco || apart * 8 > t.sig) {
        // the calls stopped looking alike: nothing to compare.  The starting point is back, and the measurement starts over on
        // the class that follows -- three times; a caller whose calls never settle keeps the starting point for good
        if (++t.restarts > 3) {
            online_tune_settle(b, 0);
        } else {
            online_tune_apply(b, 0);
            t.state = 0;
        }
        return;
    }
    t.idle = false;
    if (t.phase == 1) {
        online_tune_apply(b, 0);
        return;
    }
    if (t.phase == 3) {
        online_tune_apply(b, t.best);
        if (t.n < SDRM_WATCH_SKIP + SDRM_WATCH_TIMED) {
            return;
        }
        t.idle = true;  // every call of the probation is enqueued: the winner stays on until their completions are in
        if (hipEventQuery(t.watch[1][SDRM_WATCH_TIMED]) != hipSuccess) {
            return;
        }
        const bool ok = median_interval(t.watch[1], SDRM_WATCH_TIMED, &t.ms[7]);
        online_tune_settle(b, ok && t.ms[7] < t.ms[6] * 0.97f ? t.best : 0);
        return;
    }
    // phase 2: the blocks
    if (t.cand == 5 && t.best < 0) {
        // the first round is enqueued (blocks 0-3, then the starting point again as block 4): its winner runs again as block 5
        if (hipEventQuery(t.ev[3][SDRM_TUNE_TIMED]) != hipSuccess) {
            online_tune_apply(b, 0);
            t.idle = true;
            return;
        }
        int best = 0;
        for (int k = 0; k < 4; k++) {
            if (!median_interval(t.ev[k], SDRM_TUNE_TIMED, &t.ms[k])) {
                online_tune_settle(b, 0);
                return;
            }
            best = t.ms[k] < t.ms[best] ? k : best;
        }
        if (!median_interval(t.watch[0], SDRM_WATCH_TIMED, &t.ms[6]) || best == 0 || t.ms[best] >= t.ms[0] * 0.97f) {
            online_tune_settle(b, 0);
            return;
        }
        t.best = best;
    }
    if (t.cand < 4) {
        online_tune_apply(b, t.cand);
    } else if (t.cand == 4) {
        online_tune_apply(b, 0);
    } else if (t.cand == 5) {
        online_tune_apply(b, t.best);
    } else {
        // both rounds are enqueued: the starting point until the second round's completions are in
        online_tune_apply(b, 0);
        t.idle = true;
        if (hipEventQuery(t.ev[5][SDRM_TUNE_TIMED]) != hipSuccess) {
            return;
        }
        const bool ok = median_interval(t.ev[4], SDRM_TUNE_TIMED, &t.ms[4]) && median_interval(t.ev[5], SDRM_TUNE_TIMED, &t.ms[5]);
        if (!ok || t.ms[5] >= t.ms[4] * 0.97f) {
            online_tune_settle(b, 0);
            return;
        }
        t.phase = 3;  // the winner's probation starts with this call
        t.n = 0;
        t.idle = false;
        online_tune_apply(b, t.best);
    }
}
void sdrm_online_tune_after(sdrm_batch_t *b, hipStream_t s_clock) {
    sdrm_batch_t::OnlineTune &t = b->tune;
    if (t.state == 2 && t.chosen > 0) {
        // the guard's samples (see OnlineTune): calls 64 + SKIP .. 64 + SKIP + TIMED of a run of like calls
        if (!t.guard_alike) {
            t.guard_n = 0;
            return;
        }
        t.guard_n++;
        const int k = t.guard_n - 64;
        if (k >= SDRM_TUNE_SKIP && k <= SDRM_TUNE_SKIP + SDRM_TUNE_TIMED && !t.guard_pending) {
            (void) hipEventRecord(t.ev[0][k - SDRM_TUNE_SKIP], s_clock);
        }
        if (k == SDRM_TUNE_SKIP + SDRM_TUNE_TIMED) {
            t.guard_pending = true;
            t.guard_n = 0;
        }
        return;
    }
    if (t.state != 1 || t.idle) {
        return;
    }
    if (t.phase == 1 || t.phase == 3) {
        hipEvent_t *w = t.watch[t.phase == 1 ? 0 : 1];
        if (t.n >= SDRM_WATCH_SKIP + SDRM_WATCH_TIMED) {
            return;
        }
        t.n++;
        if (t.n >= SDRM_WATCH_SKIP) {
            (void) hipEventRecord(w[t.n - SDRM_WATCH_SKIP], s_clock);
        }
        if (t.n == SDRM_WATCH_SKIP + SDRM_WATCH_TIMED && t.phase == 1) {
            t.phase = 2;
            t.cand = 0;
            t.n = 0;
        }
        return;
    }
    if (t.cand >= 6) {
        return;
    }
    t.n++;
    if (t.n >= SDRM_TUNE_SKIP) {
        (void) hipEventRecord(t.ev[t.cand][t.n - SDRM_TUNE_SKIP], s_clock);
    }
    if (t.n == SDRM_TUNE_SKIP + SDRM_TUNE_TIMED) {
        t.cand++;
        t.n = 0;
    }
}
