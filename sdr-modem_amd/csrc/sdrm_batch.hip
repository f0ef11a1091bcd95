// sdrm_batch.hip -- the batch object's life: creation (plan, device memory, streams, tables), geometry growth, channel reset,
// inspection, the per-kernel timing lanes, the oscillators' buffers.  The calls themselves: sdrm_call.hip; the schedule and its
// tuners: sdrm_tune.hip; the reference operator (fsk_demod_*), probes and diagnostics: sdrm_handle.hip.
// There is no CPU fallback here: every compute entry point needs a HIP device and says so when it has none.
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <initializer_list>
#include <vector>

#include "../../include/sdrmodem_hip.h"
#include "sdrm_design.h"
#include "sdrm_plan.h"
#include "sdrm_launch.h"
#include "sdrm_tables.h"

#include "sdrm_batch_impl.h"

using namespace sdrm_impl;

extern "C" int sdrm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        return 0;
    }
    return n;
}

extern "C" const char *sdrm_version(void) { return "sdrmodem_hip 0.1 (gfx950, exact mode)"; }

void sdrm_impl::batch_free(sdrm_batch_t *b) {
    if (b == nullptr) {
        return;
    }
    (void) hipSetDevice(b->device);
    // this batch's own streams only: a client that leaves does not make a server's other clients drain the device
    if (quiesce(b) != 0) {
        (void) hipDeviceSynchronize();
    }
    hand_release(b);  // its entry in the device's ledger of waiting workgroups names an event that is about to go
    for (auto &pair : b->tune.ev) {
        for (hipEvent_t e : pair) {
            if (e != nullptr) {
                (void) hipEventDestroy(e);
            }
        }
    }
    for (auto &row : b->tune.watch) {
        for (hipEvent_t e : row) {
            if (e != nullptr) {
                (void) hipEventDestroy(e);
            }
        }
    }
    for (float *g : b->gen_ptr) {
        if (g != nullptr) {
            (void) hipFree(g);
        }
    }
    if (b->d_gen_state) (void) hipFree(b->d_gen_state);
    if (b->d_gen_list) (void) hipFree(b->d_gen_list);
    for (auto &lane : b->lanes) {
        for (auto &pr : lane.pending) {
            (void) hipEventDestroy(pr.first);
            (void) hipEventDestroy(pr.second);
        }
        for (auto &pr : lane.free_list) {
            (void) hipEventDestroy(pr.first);
            (void) hipEventDestroy(pr.second);
        }
        if (lane.has_prev) {
            (void) hipEventDestroy(lane.prev.first);
            (void) hipEventDestroy(lane.prev.second);
        }
    }
    for (int i = 0; i < SDRM_CTL_SLOTS; i++) {
        hipEvent_t evs[5] = {b->slot_done[i], b->ev_in[i], b->ev_front[i], b->ev_dc[i], b->ev_phase[i]};
        for (hipEvent_t e : evs) {
            if (e) {
                (void) hipEventDestroy(e);
            }
        }
    }
    hipStream_t streams[5] = {b->s_front, b->serial ? nullptr : b->s_dc, b->serial ? nullptr : b->s_clock,
                              b->s_nco != b->s_front ? b->s_nco : nullptr, b->s_company};
    if (b->ev_company) {
        (void) hipEventDestroy(b->ev_company);
    }
    if (b->ev_hand_done) {
        (void) hipEventDestroy(b->ev_hand_done);  // (its ledger entry went with hand_release above)
    }
    (void) hipFree(b->d_k3_done);
    (void) hipFree(b->d_placed);
    (void) hipFree(b->d_counters);
    if (b->s_hand_dc) {
        (void) hipStreamDestroy(b->s_hand_dc);
    }
    if (b->s_hand_clock) {
        (void) hipStreamDestroy(b->s_hand_clock);
    }
    (void) hipFree(b->d_hand_tiles);
    (void) hipFree(b->d_hand_prog);
    for (hipEvent_t e2 : b->ev_ctl) {
        if (e2) {
            (void) hipEventDestroy(e2);
        }
    }
    for (hipStream_t st : streams) {
        if (st) {
            (void) hipStreamDestroy(st);
        }
    }
    for (int i = 0; i < SDRM_RES_SETS; i++) {
        if (b->ev_res[i]) {
            (void) hipEventDestroy(b->ev_res[i]);
        }
        if (b->h_res8[i]) {
            (void) hipHostFree(b->h_res8[i]);
        }
        if (b->h_reslen[i]) {
            (void) hipHostFree(b->h_reslen[i]);
        }
    }
    for (int i = 0; i < 2; i++) {
        if (b->ev_out_free[i]) {
            (void) hipEventDestroy(b->ev_out_free[i]);
        }
        if (b->d_in_ring[i]) {
            (void) hipFree(b->d_in_ring[i]);
        }
    }
    if (b->s_h2d) {
        (void) hipStreamDestroy(b->s_h2d);
    }
    if (b->s_d2h) {
        (void) hipStreamDestroy(b->s_d2h);
    }
    if (b->h_arena) {
        (void) hipHostFree(b->h_arena);
    }
    if (b->d_out8_b) {
        (void) hipFree(b->d_out8_b);
    }
    if (b->d_outlen_b) {
        (void) hipFree(b->d_outlen_b);
    }
    if (b->dev.k3_stamps) {
        (void) hipFree(b->dev.k3_stamps);
    }
    if (b->d_timeline) {
        (void) hipFree(b->d_timeline);
    }
    void *dev_ptrs[] = {b->d_params, b->d_ctl, b->d_taps, b->d_atan, b->d_bank, b->d_hist, b->d_z, b->d_dcout,
                        b->d_dcstate, b->d_clock, b->d_out8, b->d_outf, b->d_outlen, b->d_in, b->d_flags, b->d_z2, b->d_dcout2,
                        b->d_nco_segs, b->d_nco_state, b->d_nco_phase, b->d_nco_phase2, b->d_nco_out};
    for (void *p : dev_ptrs) {
        if (p) {
            (void) hipFree(p);
        }
    }
    if (b->h_ctl) {
        (void) hipHostFree(b->h_ctl);
    }
    if (b->h_outlen) {
        (void) hipHostFree(b->h_outlen);
    }
    if (b->h_out8) {
        (void) hipHostFree(b->h_out8);
    }
    (void) hipFree(b->d_pre_state);
    (void) hipFree(b->d_pre_phase);
    (void) hipFree(b->d_pre_segs);
    (void) hipFree(b->d_ctl_pre);
    if (b->h_pre_segs) {
        (void) hipHostFree(b->h_pre_segs);
    }
    if (b->h_ctl_pre) {
        (void) hipHostFree(b->h_ctl_pre);
    }
    if (b->h_nco_segs) {
        (void) hipHostFree(b->h_nco_segs);
    }
    if (b->sg_exec) {
        (void) hipGraphExecDestroy(b->sg_exec);
    }
    if (b->h_in_stage) {
        (void) hipHostFree(b->h_in_stage);
    }
    if (b->stream) {
        (void) hipStreamDestroy(b->stream);
    }
    delete b;
}


// One-stream batches (plain handles): from this many clock-stage input samples on, a blocking call is served faster by the
// in-call hand-off on the handle's stream plus two side streams (enqueue_call) than by the graph replay of the three stages
// one after the other -- measured, 48 kHz / 4800 baud / decimation 2: 16384 outputs 735 -> 650 us, 32768: 1340 -> 1178;
// below, the replay's saved launches win (2048 outputs: 157 against 185 us).  profiles/r05_latency.txt

// the most samples a channel's clock stage carries from one call into the next (what bounds a call's symbol count from above)
uint32_t sdrm_impl::carried_cap(const sdrm_chan_params &p) {
    return p.generic ? sdrm_gen_layout_for(p.dc_len, p.omega_mid, p.max_len, p.decim).hcap : (uint32_t) SDRM_CLOCK_HCAP;
}

// The most symbols a call that brings `nz` samples to the channel's clock stage can produce: what sizes the int8 conversion's
// grid and the copy back to the host.  In lock a symbol advances by at least floor(omega_mid - omega_lim) samples; a channel
// whose timing loop can leave its tame range (sdrm_kernels.h "wild channels") may stand still and fill its output buffer
// (clock_recovery_mm.c:103 `oo < output_len`).  A tame channel far out of lock can exceed the first bound too (it advances
// by at least ONE sample per symbol): the conversion kernel's last workgroup then walks on to the real count, the blocking
// calls fetch the tail with a second copy, the pipelined path drops it with a message (sdrm_batch_collect).
uint32_t sdrm_impl::symbols_bound(const sdrm_chan_params &p, uint32_t nz) {
    const float adv = floorf(p.omega_mid - p.omega_lim);
    if (p.can_wild || !(adv >= 1.0f)) {
        return p.max_len;
    }
    return std::min<uint32_t>(p.max_len, (uint32_t) ((nz + carried_cap(p)) / (uint32_t) adv) + 8u);
}

// ---- generic channels: one device allocation per such channel (its DC rings and clock working buffer), zeroed -------------
static int sync_generic(sdrm_batch_t *b, long only_channel) {
    const sdrm::BatchPlan &pl = b->plan;
    const size_t C = pl.design.size();
    if (b->gen_ptr.size() != C) {
        b->gen_ptr.assign(C, nullptr);
    }
    bool any = false;
    for (size_t c = 0; c < C; c++) {
        const sdrm_chan_params &p = pl.params[c];
        const bool touch = only_channel < 0 || (size_t) only_channel == c;
        if (touch && b->gen_ptr[c] != nullptr) {
            (void) hipFree(b->gen_ptr[c]);
            b->gen_ptr[c] = nullptr;
        }
        if (touch && p.generic) {
            const sdrm_gen_layout g = sdrm_gen_layout_for(p.dc_len, p.omega_mid, p.max_len, p.decim);
            if (hipMalloc((void **) &b->gen_ptr[c], sizeof(float) * g.total) != hipSuccess) {
                b->gen_ptr[c] = nullptr;
                return -ENOMEM;
            }
            HIP_TRY(hipMemset(b->gen_ptr[c], 0, sizeof(float) * g.total));
        }
        any = any || p.generic;
    }
    if (!any && b->d_gen_state == nullptr) {
        b->n_gen = 0;
        b->dev.n_gen = 0;
        return 0;
    }
    if (b->d_gen_state == nullptr) {
        HIP_TRY(hipMalloc((void **) &b->d_gen_state, sizeof(float *) * C));
        HIP_TRY(hipMalloc((void **) &b->d_gen_list, sizeof(int) * C));
    }
    std::vector<int> list;
    for (size_t c = 0; c < C; c++) {
        if (pl.params[c].generic) {
            list.push_back((int) c);
        }
    }
    HIP_TRY(hipMemcpy(b->d_gen_state, b->gen_ptr.data(), sizeof(float *) * C, hipMemcpyHostToDevice));
    if (!list.empty()) {
        HIP_TRY(hipMemcpy(b->d_gen_list, list.data(), sizeof(int) * list.size(), hipMemcpyHostToDevice));
    }
    b->n_gen = (int) list.size();
    b->dev.n_gen = b->n_gen;
    b->dev.gen_list = b->d_gen_list;
    b->dev.gen_state = b->d_gen_state;
    return 0;
}



// Waits for everything this batch has put on the device -- its own streams only: another batch on the same device (a node with
// several batchers per GPU, a server with a handle per client) is not made to drain because this one resets a channel.
int sdrm_impl::quiesce(sdrm_batch_t *b) {
    hipStream_t all[10] = {b->stream, b->s_h2d, b->s_d2h, b->s_front, b->s_dc, b->s_clock, b->s_nco, b->s_company,
                           b->s_hand_dc, b->s_hand_clock};
    for (int i = 0; i < 10; i++) {
        bool seen = all[i] == nullptr;
        for (int j = 0; j < i && !seen; j++) {
            seen = all[j] == all[i];
        }
        if (!seen) {
            HIP_TRY(hipStreamSynchronize(all[i]));
        }
    }
    return 0;
}

int sdrm_impl::reset_all_streams(sdrm_batch_t *b) {
    const sdrm::BatchPlan &pl = b->plan;
    const size_t C = pl.design.size();
    if (b->n_gen > 0 || b->d_nco_state != nullptr || b->d_pre_state != nullptr) {
        return -1;  // generic channels and oscillators keep state this does not clear: the calibration never runs with them
    }
    if (int code = quiesce(b)) {
        return code;
    }
    HIP_TRY(hipMemset(b->d_hist, 0, sizeof(sdrm_f2) * C * 2 * (size_t) pl.hist_stride));
    if (b->d_dcstate != nullptr) {
        HIP_TRY(hipMemset(b->d_dcstate, 0, sizeof(float) * pl.dc_state_floats));
    }
    std::vector<sdrm_clock_state> cs(C);
    for (size_t c = 0; c < C; c++) {
        memset(&cs[c], 0, sizeof(cs[c]));
        cs[c].mu = 0.5f;
        cs[c].omega = pl.design[c].sps;
    }
    HIP_TRY(hipMemcpy(b->d_clock, cs.data(), sizeof(sdrm_clock_state) * C, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(b->d_flags, 0, sizeof(uint32_t) * C * SDRM_CTL_SLOTS));
    HIP_TRY(hipMemset(b->d_outlen, 0, sizeof(uint32_t) * C));
    HIP_TRY(hipMemset(b->d_counters, 0, 64));  // sdrm_batch_wild_calls counts the caller's calls, not the calibration's
    if (b->d_hand_tiles != nullptr) {
        HIP_TRY(hipMemset(b->d_hand_tiles, 0, sizeof(uint32_t) * C * (size_t) b->hand_tiles_cap));
    }
    if (b->d_hand_prog != nullptr) {
        HIP_TRY(hipMemset(b->d_hand_prog, 0, sizeof(unsigned long long) * C));
    }
    HIP_TRY(hipStreamSynchronize(nullptr));
    std::fill(b->plan.phase.begin(), b->plan.phase.end(), 0u);
    std::fill(b->plan.parity.begin(), b->plan.parity.end(), 0u);
    std::fill(b->plan.zbase.begin(), b->plan.zbase.end(), 0u);
    std::fill(b->last_lens.begin(), b->last_lens.end(), 0u);
    for (int i = 0; i < SDRM_CTL_SLOTS; i++) {
        b->slot_used[i] = false;
    }
    b->calls = 0;
    b->hand_calls = 0;
    b->last_slot = -1;
    return 0;
}

extern "C" int sdrm_batch_create(const sdrm_fsk_config *cfgs, size_t n_channels, int device, uint32_t flags,
                                 sdrm_batch **out) {
    if (cfgs == nullptr || n_channels == 0 || out == nullptr) {
        return -1;
    }
    if (flags & SDRM_FLAG_FAST_FMA) {
        fprintf(stderr, "<3>sdrmodem_hip: SDRM_FLAG_FAST_FMA was removed (it did not hold the reference's own +-2 LSB tolerance)\n");
        return -ENOTSUP;
    }
    // design + planning first: parameter errors are reported exactly like the reference, GPU or not
    sdrm_batch_t *b = new sdrm_batch_t();
    int code = sdrm::plan_batch(cfgs, n_channels, b->plan);
    if (code != 0) {
        delete b;
        return code;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "<3>sdrmodem_hip: no HIP device available; this library has no CPU fallback\n");
        delete b;
        return -ENODEV;
    }
    if (device < 0 && hipGetDevice(&device) != hipSuccess) {
        delete b;
        return -ENODEV;
    }
    if (device >= ndev) {
        fprintf(stderr, "<3>sdrmodem_hip: device %d out of range (%d devices)\n", device, ndev);
        delete b;
        return -ENODEV;
    }
    if (hipSetDevice(device) != hipSuccess) {
        delete b;
        return -ENODEV;
    }
    b->device = device;
    b->flags = flags;
    const size_t C = n_channels;
    b->last_lens.assign(C, 0);
    const sdrm::BatchPlan &pl = b->plan;
    const std::vector<float> &pool = pl.tap_pool;
    const int any_dc = pl.any_dc;
    const size_t dc_floats = pl.dc_state_floats;

    const uint32_t hist_stride = pl.hist_stride, z_stride = pl.z_stride, out_stride = pl.out_stride;
    code = code ? code : dev_alloc_zero(&b->d_params, C);
    code = code ? code : dev_alloc_zero(&b->d_ctl, C * SDRM_CTL_SLOTS);
    code = code ? code : dev_alloc_zero(&b->d_taps, pl.private_taps_base + C * pl.private_taps_slot + 16);
    code = code ? code : dev_alloc_zero(&b->d_atan, 260);
    code = code ? code : dev_alloc_zero(&b->d_bank, 129 * 8);
    code = code ? code : dev_alloc_zero(&b->d_hist, C * 2 * (size_t) hist_stride);
    code = code ? code : dev_alloc_zero(&b->d_z, C * (size_t) z_stride);
    code = code ? code : dev_alloc_zero(&b->d_z2, C * (size_t) z_stride);
    if (any_dc) {
        code = code ? code : dev_alloc_zero(&b->d_dcout, C * (size_t) z_stride);
        code = code ? code : dev_alloc_zero(&b->d_dcout2, C * (size_t) z_stride);
        code = code ? code : dev_alloc_zero(&b->d_dcstate, dc_floats);
    }
    code = code ? code : dev_alloc_zero(&b->d_clock, C);
    code = code ? code : dev_alloc_zero(&b->d_out8, C * (size_t) out_stride);
    // the clock stage writes float soft bits, a pointwise kernel behind it the int8 ones: the float buffer always exists
    // (SDRM_FLAG_KEEP_SOFT_F32 only promises the caller that it may read it)
    code = code ? code : dev_alloc_zero(&b->d_outf, C * (size_t) out_stride);
    code = code ? code : dev_alloc_zero(&b->d_outlen, C);
    code = code ? code : dev_alloc_zero(&b->d_flags, C * SDRM_CTL_SLOTS);
    if (code != 0) {
        batch_free(b);
        return code;
    }
    if (hipHostMalloc((void **) &b->h_ctl, sizeof(sdrm_chunk_ctl) * C * SDRM_CTL_SLOTS) != hipSuccess ||
        hipHostMalloc((void **) &b->h_outlen, sizeof(uint32_t) * C) != hipSuccess) {
        batch_free(b);
        return -ENOMEM;
    }
    // initial clock state: mu = 0.5, omega = sps (fsk_demod.c:63, clock_recovery_mm.c:37-45)
    std::vector<sdrm_clock_state> cs(C);
    for (size_t c = 0; c < C; c++) {
        memset(&cs[c], 0, sizeof(cs[c]));
        cs[c].mu = 0.5f;
        cs[c].omega = pl.design[c].sps;
        cs[c].last = 0.0f;
        cs[c].kept = 0;
    }
    hipError_t e = hipSuccess;
    e = e ? e : hipMemcpy(b->d_params, pl.params.data(), sizeof(sdrm_chan_params) * C, hipMemcpyHostToDevice);
    e = e ? e : hipMemcpy(b->d_taps, pool.data(), sizeof(float) * pool.size(), hipMemcpyHostToDevice);
    e = e ? e : hipMemcpy(b->d_atan, sdrm_atan_tab, sizeof(float) * 257, hipMemcpyHostToDevice);
    e = e ? e : hipMemcpy(b->d_bank, sdrm_mmse_bank, sizeof(float) * 129 * 8, hipMemcpyHostToDevice);
    e = e ? e : hipMemcpy(b->d_clock, cs.data(), sizeof(sdrm_clock_state) * C, hipMemcpyHostToDevice);
    e = e ? e : hipMalloc((void **) &b->d_counters, 64);
    e = e ? e : hipMemset(b->d_counters, 0, 64);
    e = e ? e : hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    // HIP multiplexes streams onto a few hardware queues, and two streams on one queue run back to back.  Streams of
    // different priority never share a queue, so give each stage its own level: the clock stage (the longest
    // dependent chain, a handful of waves) highest, the wide front-end lowest.
    int prio_low = 0, prio_high = 0;
    e = e ? e : hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    const int prio_mid = (prio_low + prio_high) / 2;
    e = e ? e : hipStreamCreateWithPriority(&b->s_front, hipStreamNonBlocking, prio_low);
    if (getenv("SDRM_SERIAL_STAGES") != nullptr || n_channels == 1) {
        // A batch of one channel (a plain fsk_demod handle) gains nothing from overlapping its stages across calls --
        // the caller waits for every call -- and a server with one handle per client would otherwise hold four streams
        // per client on a handful of hardware queues.
        // escape hatch: all stages on one stream (no overlap between consecutive calls); same kernels, same results
        b->s_dc = b->s_front;
        b->s_clock = b->s_front;
        b->serial = true;
    } else {
        e = e ? e : hipStreamCreateWithPriority(&b->s_dc, hipStreamNonBlocking, prio_mid);
        e = e ? e : hipStreamCreateWithPriority(&b->s_clock, hipStreamNonBlocking, prio_high);
        e = e ? e : hipMalloc((void **) &b->d_placed, 64);
        e = e ? e : hipMemset(b->d_placed, 0, 64);
        // company for the clock stage while the batch is too small to keep the chip busy by itself: the front-end of a
        // full-length call must be expected to take well under the clock stage's time (at 1024 channels of the bench
        // workload it does not, nor with the 397-tap filters of 240 kHz channels: BASELINE configs[4] in one GPU's share
        // ran 2.53 ms per call with the companion grid and 1.94 without).  Estimates: the front-end's multiply-adds at the
        // rate it reaches beside the other stages (18.4 T/s: 0.53 ms for 256 x 131072 samples of 291), the clock stage's
        // longest symbol sequence at 97 ns per symbol (220 cycles at 2.27 GHz).
        // SDRM_K3_COMPANY="blocks,first,last" overrides grid and channel range (blocks 0: none)
        int blocks = 4096, lo = 32, hi = 768;
        if (const char *env = getenv("SDRM_K3_COMPANY")) {
            sscanf(env, "%d,%d,%d,%d", &blocks, &lo, &hi, &b->company_nops);
        } else {
            double macs = 0.0, symbols = 0.0;
            for (size_t c = 0; c < C; c++) {
                const double n = (double) cfgs[c].max_input_buffer_length;
                const double d = cfgs[c].decimation ? (double) cfgs[c].decimation : 1.0;
                macs += n * (2.0 * pl.params[c].T1 + pl.params[c].T2 / d);
                const double sym = cfgs[c].sampling_freq ? n * (double) cfgs[c].baud_rate / (double) cfgs[c].sampling_freq : 0.0;
                symbols = sym > symbols ? sym : symbols;
            }
            b->est_front_ms = (float) (macs / 18.4e12 * 1e3);   // (the calibration asks only where these leave the rule in doubt)
            b->est_clock_ms = (float) (symbols * 97e-9 * 1e3);
            if (macs / 18.4e12 > 0.7 * symbols * 97e-9) {
                blocks = 0;
            }
            // and the clock stage of a full call must run long enough to pay for the grid's launch and wind-down: with
            // 4096-sample calls (0.08 ms of clock stage) the grid cost 14 % at 256 channels, with 32768-sample calls
            // (0.64 ms) it gains 8 % (profiles/r03_heuristics_before.txt / _after.txt)
            if (symbols * 97e-9 < 0.3e-3) {
                blocks = 0;
            }
            // the grid lives as long as a full-length call's clock stage may take (half as long again), at least ~1 ms
            const double rounds = symbols * 97e-9 * 1.5 / 50e-6 + 20.0;
            b->company_rounds = rounds > 4000.0 ? 4000 : (int) rounds;
        }
        if (blocks > 0 && (int) n_channels >= lo && (int) n_channels <= hi) {
            b->company_blocks = blocks;
            b->company_grid = blocks;
        }
        // the side stream and its event exist whether the grid starts switched on or not: the calibration may switch it
        e = e ? e : hipStreamCreateWithFlags(&b->s_company, hipStreamNonBlocking);
        e = e ? e : hipEventCreateWithFlags(&b->ev_company, hipEventDisableTiming);
        b->hold_front = sdrm::front_waits_for_clock_start((int) n_channels);
        e = e ? e : hipMalloc((void **) &b->d_k3_done, 64);
        e = e ? e : hipMemset(b->d_k3_done, 0, 64);
    }
    for (int i = 0; i < SDRM_CTL_SLOTS && e == hipSuccess; i++) {
        e = hipEventCreateWithFlags(&b->slot_done[i], hipEventDisableTiming);
        e = e ? e : hipEventCreateWithFlags(&b->ev_in[i], hipEventDisableTiming);
        e = e ? e : hipEventCreateWithFlags(&b->ev_front[i], hipEventDisableTiming);
        e = e ? e : hipEventCreateWithFlags(&b->ev_dc[i], hipEventDisableTiming);
        e = e ? e : hipEventCreateWithFlags(&b->ev_ctl[i], hipEventDisableTiming);
    }
    if (const char *env = getenv("SDRM_HANDOFF")) {
        b->hand_allowed = atoi(env) != 0;
    }
    if (const char *env = getenv("SDRM_HAND_EPOCH0")) {  // tests: start the hand-off's call count near the end of its 32-bit stamp values
        b->hand_epoch = strtoull(env, nullptr, 0);
    }
    for (size_t c = 0; c < C; c++) {
        b->any_nodc = b->any_nodc || pl.params[c].dc_len == 0;
    }
    if (e != hipSuccess) {
        fprintf(stderr, "<3>sdrmodem_hip: device initialisation failed: %s\n", hipGetErrorString(e));
        batch_free(b);
        return -EIO;
    }
    // K1 uses more than the default 64 KiB of dynamic LDS only for very long filters; raise the cap once
    sdrm::DeviceBatch &d = b->dev;
    d.n_channels = (int) C;
    d.params = b->d_params;
    d.tap_pool = b->d_taps;
    d.atan_tab = b->d_atan;
    d.mmse_bank = b->d_bank;
    d.counters = b->d_counters;
    d.raw_hist = b->d_hist;
    d.hist_stride = hist_stride;
    d.z = b->d_z;
    d.dcout = b->d_dcout;
    d.z_stride = z_stride;
    d.dc_state = b->d_dcstate;
    d.clock_state = b->d_clock;
    d.out_i8 = b->d_out8;
    d.out_f32 = b->d_outf;
    d.out_len = b->d_outlen;
    d.out_stride = out_stride;
    d.t1_max = pl.t1_max;
    d.t2_max = pl.t2_max;
    d.dc_hx_cap = pl.dc_hx_cap;
    d.dc_l_cap = pl.dc_l_cap;
    d.dc_group = pl.dc_group;
    d.dc_rpitch = sdrm_k2_ring_pitch((pl.dc_l_cap + SDRM_K2_BLK - 1) / SDRM_K2_BLK * SDRM_K2_BLK + SDRM_K2_BLK);
    d.dc_lds = (uint32_t) pl.dc_lds_bytes();
    d.any_dc = any_dc;
    d.k3_carried_max = (int) pl.clock_carried_max;
    b->in_stride = pl.in_stride;
    code = sync_generic(b, -1);
    if (code != 0) {
        batch_free(b);
        return code;
    }
    code = sdrm_calibrate(b, cfgs);
    if (code != 0) {
        batch_free(b);
        return code;
    }
    *out = b;
    return 0;
}

extern "C" void sdrm_batch_destroy(sdrm_batch *b) { batch_free(b); }

extern "C" int sdrm_batch_schedule(const sdrm_batch *b, sdrm_batch_schedule_info *info) {
    if (b == nullptr || info == nullptr) {
        return -1;
    }
    const sdrm_k3_shape sh = sdrm::describe_shape(b->dev);
    info->k3_lanes = sh.lanes;
    info->k3_ring = sh.ring;
    info->k3_plain = sh.plain;
    info->front_hold = (!b->serial && b->hold_front) ? 1 : 0;
    info->company_blocks = b->company_blocks;
    info->calibrated = b->calibrated ? 1 : 0;
    info->ms_before = b->calib_ms[0];
    info->ms_after = b->calib_ms[1];
    info->ms_spent = b->calib_ms[2];
    info->online_state = b->tune.state;
    info->online_choice = b->tune.chosen;
    for (int k = 0; k < 8; k++) {
        info->online_ms[k] = b->tune.ms[k];
    }
    return 0;
}

extern "C" size_t sdrm_batch_channels(const sdrm_batch *b) { return b ? b->plan.design.size() : 0; }

extern "C" int sdrm_batch_info(const sdrm_batch *b, size_t c, sdrm_fsk_info *info) {
    if (b == nullptr || c >= b->plan.design.size() || info == nullptr) {
        return -1;
    }
    const sdrm::ChannelDesign &d = b->plan.design[c];
    info->taps1_len = (uint32_t) d.taps1.size();
    info->taps2_len = (uint32_t) d.taps2.size();
    info->dc_length = d.dc_length;
    info->quad_gain = d.quad_gain;
    info->sps = d.sps;
    info->gain_omega = d.gain_omega;
    info->gain_mu = d.gain_mu;
    info->omega_lim = d.omega_lim;
    return 0;
}

// The batch grows: longer filters, a longer raw history or a longer (or the first) DC boxcar than any channel had when the
// batch was created.  Nothing is in flight (the caller has synchronised).  The other channels keep their streams: their raw
// histories move to the new stride, their DC states to the new region layout (carried samples, three tails, four sums), the
// taps of channels in private slots to the new slot size.  Rare (a client with a lower baud rate than anybody before), so
// done the simple way, through the host.
static int grow_geometry(sdrm_batch_t *b, const sdrm::GeometryGrowth &g) {
    const sdrm::BatchPlan &old = b->plan;
    const size_t C = old.design.size();
    const uint32_t old_hist = old.hist_stride, old_hx = old.dc_hx_cap, old_l = old.dc_l_cap;
    const size_t old_region = old.dc_region_floats;
    const bool had_dc = old.any_dc != 0;
    std::vector<sdrm_f2> hist(C * 2 * (size_t) old_hist);
    HIP_TRY(hipMemcpy(hist.data(), b->d_hist, sizeof(sdrm_f2) * hist.size(), hipMemcpyDeviceToHost));
    std::vector<float> dc(had_dc ? C * old_region : 0);
    if (had_dc) {
        HIP_TRY(hipMemcpy(dc.data(), b->d_dcstate, sizeof(float) * dc.size(), hipMemcpyDeviceToHost));
    }
    // the new plan is worked out on a copy and the new buffers are filled before anything of the batch changes: a failed
    // allocation leaves the batch exactly as it was
    sdrm::BatchPlan pl = old;
    std::vector<size_t> moved;
    sdrm::apply_growth(pl, g, moved);
    // raw histories: [C][2][stride], a row holds the channel's hist_len samples from its start: copied as they are
    std::vector<sdrm_f2> hist2(C * 2 * (size_t) pl.hist_stride);
    memset(hist2.data(), 0, sizeof(sdrm_f2) * hist2.size());
    for (size_t r = 0; r < C * 2; r++) {
        memcpy(hist2.data() + r * pl.hist_stride, hist.data() + r * old_hist, sizeof(sdrm_f2) * old_hist);
    }
    std::vector<float> dc2(pl.any_dc ? C * pl.dc_region_floats : 0, 0.0f);
    if (had_dc) {
        for (size_t ch = 0; ch < C; ch++) {
            const float *src = dc.data() + ch * old_region;
            float *dst = dc2.data() + ch * pl.dc_region_floats;
            memcpy(dst, src, sizeof(float) * old_hx);  // carried samples of x: at the front of the array
            for (int ring = 0; ring < 3; ring++) {
                memcpy(dst + pl.dc_hx_cap + (size_t) ring * pl.dc_l_cap, src + old_hx + (size_t) ring * old_l, sizeof(float) * old_l);
            }
            memcpy(dst + pl.dc_hx_cap + 3 * (size_t) pl.dc_l_cap, src + old_hx + 3 * (size_t) old_l, sizeof(float) * 8);
        }
    }
    sdrm_f2 *n_hist = nullptr;
    float *n_dc = nullptr, *n_taps = nullptr, *n_dcout = nullptr, *n_dcout2 = nullptr;
    sdrm_chan_params *n_params = nullptr;
    int code = dev_alloc_zero(&n_hist, hist2.size());
    code = code ? code : dev_alloc_zero(&n_taps, pl.private_taps_base + C * pl.private_taps_slot + 16);
    code = code ? code : dev_alloc_zero(&n_params, C);
    if (pl.any_dc) {
        code = code ? code : dev_alloc_zero(&n_dc, dc2.size());
        if (!had_dc) {
            code = code ? code : dev_alloc_zero(&n_dcout, C * (size_t) pl.z_stride);
            code = code ? code : dev_alloc_zero(&n_dcout2, C * (size_t) pl.z_stride);
        }
    }
    hipError_t e = hipSuccess;
    if (code == 0) {
        e = e ? e : hipMemcpy(n_hist, hist2.data(), sizeof(sdrm_f2) * hist2.size(), hipMemcpyHostToDevice);
        if (pl.any_dc) {
            e = e ? e : hipMemcpy(n_dc, dc2.data(), sizeof(float) * dc2.size(), hipMemcpyHostToDevice);
        }
        e = e ? e : hipMemcpy(n_taps, pl.tap_pool.data(), sizeof(float) * pl.tap_pool.size(), hipMemcpyHostToDevice);
        for (size_t ch : moved) {  // private slots: taps from the channel's design, reversed, at the new offsets
            const sdrm::ChannelDesign &d = pl.design[ch];
            std::vector<float> t1(d.taps1.rbegin(), d.taps1.rend()), t2(d.taps2.rbegin(), d.taps2.rend());
            e = e ? e : hipMemcpy(n_taps + pl.params[ch].taps1_off, t1.data(), sizeof(float) * t1.size(), hipMemcpyHostToDevice);
            e = e ? e : hipMemcpy(n_taps + pl.params[ch].taps2_off, t2.data(), sizeof(float) * t2.size(), hipMemcpyHostToDevice);
        }
        e = e ? e : hipMemcpy(n_params, pl.params.data(), sizeof(sdrm_chan_params) * C, hipMemcpyHostToDevice);
    }
    if (code != 0 || e != hipSuccess) {
        void *fresh[] = {n_hist, n_taps, n_params, n_dc, n_dcout, n_dcout2};
        for (void *q : fresh) {
            if (q) {
                (void) hipFree(q);
            }
        }
        if (code == 0) {
            fprintf(stderr, "<3>sdrmodem_hip: growing the batch failed: %s\n", hipGetErrorString(e));
        }
        return code ? code : -EIO;
    }
    // commit
    (void) hipFree(b->d_hist);
    (void) hipFree(b->d_taps);
    (void) hipFree(b->d_params);
    if (b->d_dcstate != nullptr) {
        (void) hipFree(b->d_dcstate);
    }
    b->plan = std::move(pl);
    b->d_hist = n_hist;
    b->d_taps = n_taps;
    b->d_params = n_params;
    b->d_dcstate = n_dc;
    if (!had_dc && b->plan.any_dc) {
        b->d_dcout = n_dcout;
        b->d_dcout2 = n_dcout2;
    }
    sdrm::DeviceBatch &d = b->dev;
    const sdrm::BatchPlan &np = b->plan;
    d.params = b->d_params;
    d.tap_pool = b->d_taps;
    d.raw_hist = b->d_hist;
    d.hist_stride = np.hist_stride;
    d.dc_state = b->d_dcstate;
    d.dcout = b->d_dcout;
    d.t1_max = np.t1_max;
    d.t2_max = np.t2_max;
    d.dc_hx_cap = np.dc_hx_cap;
    d.dc_l_cap = np.dc_l_cap;
    d.dc_group = np.dc_group;
    d.dc_rpitch = sdrm_k2_ring_pitch((np.dc_l_cap + SDRM_K2_BLK - 1) / SDRM_K2_BLK * SDRM_K2_BLK + SDRM_K2_BLK);
    d.dc_lds = (uint32_t) np.dc_lds_bytes();
    d.any_dc = np.any_dc;
    return 0;
}

// Hand channel `c` to a new stream: zero its streaming state (filter histories, DC blocker, timing loop, NCO phase)
// and, with `cfg` != NULL, give it a new configuration.  Filters, raw history and DC boxcar longer than anything the
// batch has held so far make the batch grow (round 3; the other channels keep their streams); only the buffer length
// is fixed for the batch's life (-ENOTSUP beyond it).  Waits for enqueued calls.
extern "C" int sdrm_batch_reset_channel(sdrm_batch *b, size_t c, const sdrm_fsk_config *cfg) {
    if (b == nullptr || c >= b->plan.design.size()) {
        return -1;
    }
    HIP_TRY(hipSetDevice(b->device));
    if (int code = quiesce(b)) {
        return code;
    }
    sdrm::BatchPlan &pl = b->plan;
    const sdrm_fsk_config use = cfg ? *cfg : pl.design[c].cfg;
    sdrm::GeometryGrowth growth;
    int code = sdrm::plan_growth(pl, use, growth);
    if (code != 0) {
        return code;
    }
    if (growth.needed) {
        code = grow_geometry(b, growth);
        if (code != 0) {
            return code;
        }
    }
    std::vector<float> slot;
    code = sdrm::replan_channel(pl, c, use, slot);
    if (code != 0) {
        return code;
    }
    const sdrm_chan_params &p = pl.params[c];
    HIP_TRY(hipMemcpy(b->d_taps + p.taps1_off, slot.data(), sizeof(float) * slot.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b->d_params + c, &p, sizeof(p), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(b->d_hist + c * 2 * (size_t) pl.hist_stride, 0, sizeof(sdrm_f2) * 2 * (size_t) pl.hist_stride));
    if (b->d_dcstate != nullptr && pl.dc_region_floats) {
        HIP_TRY(hipMemset(b->d_dcstate + c * pl.dc_region_floats, 0, sizeof(float) * pl.dc_region_floats));
    }
    sdrm_clock_state cs;
    memset(&cs, 0, sizeof(cs));
    cs.mu = 0.5f;
    cs.omega = pl.design[c].sps;
    HIP_TRY(hipMemcpy(b->d_clock + c, &cs, sizeof(cs), hipMemcpyHostToDevice));
    for (int s = 0; s < SDRM_CTL_SLOTS; s++) {
        HIP_TRY(hipMemset(b->d_flags + (size_t) s * pl.design.size() + c, 0, sizeof(uint32_t)));
    }
    if (b->d_pre_state != nullptr) {  // the new stream has no oscillator in front until it asks for one
        HIP_TRY(hipMemset(b->d_pre_state + c, 0, sizeof(float)));
    }
    if (c < b->pre_offset.size() && b->pre_offset[c] != 0) {
        b->pre_offset[c] = 0;
        b->any_pre = false;
        for (int64_t f : b->pre_offset) {
            b->any_pre = b->any_pre || f != 0;
        }
    }
    if (b->d_nco_state != nullptr) {
        HIP_TRY(hipMemset(b->d_nco_state + c, 0, sizeof(float)));
    }
    HIP_TRY(hipStreamSynchronize(nullptr));  // the memsets above have landed before a (non-blocking) pipeline stream runs
    if (b->sg_exec != nullptr) {  // a graph built for the channel's previous parameters (grids, widths) is stale
        (void) hipGraphExecDestroy(b->sg_exec);
        b->sg_exec = nullptr;
        b->sg_len = 0;
    }
    b->any_nodc = false;
    for (const sdrm_chan_params &q : pl.params) {
        b->any_nodc = b->any_nodc || q.dc_len == 0;
    }
    b->last_lens[c] = 0;
    b->dev.k3_carried_max = (int) pl.clock_carried_max;
    return sync_generic(b, (long) c);  // the channel's generic state goes, comes or starts afresh with its configuration
}

extern "C" size_t sdrm_batch_taps(const sdrm_batch *b, size_t c, int stage, float *dst, size_t cap) {
    if (b == nullptr || c >= b->plan.design.size()) {
        return 0;
    }
    const std::vector<float> &t = (stage == 2) ? b->plan.design[c].taps2 : b->plan.design[c].taps1;
    if (dst != nullptr) {
        memcpy(dst, t.data(), sizeof(float) * std::min(cap, t.size()));
    }
    return t.size();
}

// --- timing helpers ------------------------------------------------------------------------------

void sdrm_impl::timing_begin(sdrm_batch_t *b, int which, hipStream_t s, std::pair<hipEvent_t, hipEvent_t> *pr) {
    TimingLane &lane = b->lanes[which];
    if (lane.free_list.empty()) {
        hipEvent_t a, z;
        (void) hipEventCreate(&a);
        (void) hipEventCreate(&z);
        *pr = {a, z};
    } else {
        *pr = lane.free_list.back();
        lane.free_list.pop_back();
    }
    (void) hipEventRecord(pr->first, s);
}

void sdrm_impl::timing_end(sdrm_batch_t *b, int which, hipStream_t s, const std::pair<hipEvent_t, hipEvent_t> &pr) {
    (void) hipEventRecord(pr.second, s);
    b->lanes[which].pending.push_back(pr);
}

void sdrm_impl::timing_collect(sdrm_batch_t *b) {
    for (auto &lane : b->lanes) {
        for (auto &pr : lane.pending) {
            (void) hipEventSynchronize(pr.second);
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                float waited = 0.0f;  // from this launch's start to the end of the one before it
                if (lane.overlapped && lane.has_prev && hipEventElapsedTime(&waited, pr.first, lane.prev.second) == hipSuccess &&
                    waited > 0.0f) {
                    ms -= waited < ms ? waited : ms;
                }
                lane.total_ms += ms;
                lane.launches++;
            }
            if (lane.overlapped) {
                if (lane.has_prev) {
                    lane.free_list.push_back(lane.prev);
                }
                lane.prev = pr;
                lane.has_prev = true;
            } else {
                lane.free_list.push_back(pr);
            }
        }
        lane.pending.clear();
    }
}

extern "C" int sdrm_batch_timing_enable(sdrm_batch *b, int enable) {
    if (b == nullptr) {
        return -1;
    }
    timing_collect(b);
    for (auto &lane : b->lanes) {
        lane.total_ms = 0.0;
        lane.launches = 0;
    }
    b->timing = enable != 0;
    return 0;
}

extern "C" int sdrm_batch_timing_read(sdrm_batch *b, int which, double *total_ms, uint64_t *launches) {
    if (b == nullptr || which < 0 || which > 2) {
        return -1;
    }
    timing_collect(b);
    if (total_ms) {
        *total_ms = b->lanes[which].total_ms;
    }
    if (launches) {
        *launches = b->lanes[which].launches;
    }
    return 0;
}

// --- the call ------------------------------------------------------------------------------------

// Enqueue one call.  `caller` is the stream on which the caller's input becomes ready; the stages run on the
// batch's own streams so that the front-end of call i+1, the DC blocker of call i and the clock recovery of call i-1
// can be resident together (the sequential stages only occupy a few waves).  Nothing waits on the host.
int sdrm_impl::ensure_nco(sdrm_batch_t *b) {
    if (b->d_nco_out != nullptr) {
        return 0;
    }
    const size_t C = b->plan.design.size();
    b->nco_seg_cap = 8 * C + 64;
    int code = 0;
    code = code ? code : dev_alloc_zero(&b->d_nco_segs, b->nco_seg_cap * SDRM_CTL_SLOTS);
    code = code ? code : dev_alloc_zero(&b->d_nco_state, C);
    code = code ? code : dev_alloc_zero(&b->d_nco_phase, C * (size_t) SDRM_PHASE_STRIDE(b->in_stride));
    if (b->serial) {
        b->d_nco_phase2 = nullptr;
        b->s_nco = b->s_front;
    } else {
        // The phase recursion depends on nothing but the batch table and its own state: it gets a stream of its own and
        // a second phase buffer, and runs while the previous call is still in its later stages.
        code = code ? code : dev_alloc_zero(&b->d_nco_phase2, C * (size_t) SDRM_PHASE_STRIDE(b->in_stride));
        int prio_low = 0, prio_high = 0;
        if (code == 0 && (hipDeviceGetStreamPriorityRange(&prio_low, &prio_high) != hipSuccess ||
                          hipStreamCreateWithPriority(&b->s_nco, hipStreamNonBlocking, prio_high) != hipSuccess)) {
            code = -ENOMEM;
        }
    }
    for (int i = 0; i < SDRM_CTL_SLOTS && code == 0; i++) {
        if (hipEventCreateWithFlags(&b->ev_phase[i], hipEventDisableTiming) != hipSuccess) {
            code = -ENOMEM;
        }
    }
    code = code ? code : dev_alloc_zero(&b->d_nco_out, C * (size_t) b->in_stride);
    if (code == 0 && hipHostMalloc((void **) &b->h_nco_segs, sizeof(sdrm_nco_seg) * b->nco_seg_cap * SDRM_CTL_SLOTS) != hipSuccess) {
        code = -ENOMEM;
    }
    return code;
}

// buffers of the constant-frequency oscillator in front of the path (allocated when the first channel asks for one)
static int ensure_pre(sdrm_batch_t *b) {
    int code = ensure_nco(b);
    if (code != 0 || b->d_pre_state != nullptr) {
        return code;
    }
    const size_t C = b->plan.design.size();
    code = code ? code : dev_alloc_zero(&b->d_pre_state, C);
    code = code ? code : dev_alloc_zero(&b->d_pre_phase, C * (size_t) SDRM_PHASE_STRIDE(b->in_stride));
    code = code ? code : dev_alloc_zero(&b->d_pre_segs, C * SDRM_CTL_SLOTS);
    code = code ? code : dev_alloc_zero(&b->d_ctl_pre, C * SDRM_CTL_SLOTS);
    if (code == 0 && (hipHostMalloc((void **) &b->h_pre_segs, sizeof(sdrm_nco_seg) * C * SDRM_CTL_SLOTS) != hipSuccess ||
                      hipHostMalloc((void **) &b->h_ctl_pre, sizeof(sdrm_chunk_ctl) * C * SDRM_CTL_SLOTS) != hipSuccess)) {
        code = -ENOMEM;
    }
    return code;
}

// From the next call on, the channel's input is mixed with ONE oscillator at the integer frequency freq_hz (fp32 phase carried
// across calls, started at 0 now) in front of everything else -- what the reference's file source does with RxRequest.rx_offset
// (src/sdr/file_source.c:120-128, sig_source_multiply) before the samples reach dsp_worker_put.  NCO batches of the same call
// (the Doppler correction, src/dsp_worker.c:65-71) then run BEHIND it: two oscillators in series, every sample rounded to fp32
// in between, as in the reference.  freq_hz == 0 switches it off.  Waits for enqueued calls.
extern "C" int sdrm_batch_set_pre_offset(sdrm_batch *b, size_t channel, int64_t freq_hz) {
    if (b == nullptr || channel >= b->plan.design.size()) {
        return -1;
    }
    HIP_TRY(hipSetDevice(b->device));
    int code = wait_for_all_calls(b);
    if (code != 0) {
        return code;
    }
    if (freq_hz != 0) {
        code = ensure_pre(b);
        if (code != 0) {
            return code;
        }
    }
    b->pre_offset.resize(b->plan.design.size(), 0);
    b->pre_offset[channel] = freq_hz;
    b->any_pre = false;
    for (int64_t f : b->pre_offset) {
        b->any_pre = b->any_pre || f != 0;
    }
    if (b->d_pre_state != nullptr) {
        HIP_TRY(hipMemset(b->d_pre_state + channel, 0, sizeof(float)));
        HIP_TRY(hipStreamSynchronize(nullptr));
    }
    if (b->sg_exec != nullptr) {  // the one-channel graph has no such pass: calls take the plain path from here on
        (void) hipGraphExecDestroy(b->sg_exec);
        b->sg_exec = nullptr;
    }
    return 0;
}

