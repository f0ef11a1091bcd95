// sdrm_api.hip -- the C-ABI of libsdrmodem_hip.so (include/sdrmodem_hip.h): batched demodulator object,
// the reference-compatible fsk_demod_* operator on top of it, and small test probes.
// There is no CPU fallback here: every compute entry point needs a HIP device and says so when it has none.
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
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

extern "C" int sdrm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        return 0;
    }
    return n;
}

extern "C" const char *sdrm_version(void) { return "sdrmodem_hip 0.1 (gfx950, exact mode)"; }

static void batch_free(sdrm_batch_t *b) {
    if (b == nullptr) {
        return;
    }
    (void) hipSetDevice(b->device);
    if (b->stream) {
        (void) hipStreamSynchronize(b->stream);
    }
    (void) hipDeviceSynchronize();
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

template <typename T>
static int dev_alloc_zero(T **ptr, size_t count) {
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    hipError_t e = hipMalloc((void **) ptr, bytes);
    if (e != hipSuccess) {
        fprintf(stderr, "<3>sdrmodem_hip: hipMalloc(%zu) failed: %s\n", bytes, hipGetErrorString(e));
        return -ENOMEM;
    }
    // hipMemset returns before the device has written the zeros, and the pipeline's streams are non-blocking streams:
    // they do not wait for the null stream.  A buffer allocated lazily (the staged input of the first host call, the NCO
    // buffers) would otherwise be zeroed while the first copy or kernel is already using it.
    e = hipMemset(*ptr, 0, bytes);
    e = e ? e : hipStreamSynchronize(nullptr);
    return e == hipSuccess ? 0 : -EIO;
}

// One-stream batches (plain handles): from this many clock-stage input samples on, a blocking call is served faster by the
// in-call hand-off on the handle's stream plus two side streams (enqueue_call) than by the graph replay of the three stages
// one after the other -- measured, 48 kHz / 4800 baud / decimation 2: 16384 outputs 735 -> 650 us, 32768: 1340 -> 1178;
// below, the replay's saved launches win (2048 outputs: 157 against 185 us).  profiles/r05_latency.txt
#define SDRM_HAND_SERIAL_MIN_NZ 12288u

// the most samples a channel's clock stage carries from one call into the next (what bounds a call's symbol count from above)
static uint32_t carried_cap(const sdrm_chan_params &p) {
    return p.generic ? sdrm_gen_layout_for(p.dc_len, p.omega_mid, p.max_len, p.decim).hcap : (uint32_t) SDRM_CLOCK_HCAP;
}

// The most symbols a call that brings `nz` samples to the channel's clock stage can produce: what sizes the int8 conversion's
// grid and the copy back to the host.  In lock a symbol advances by at least floor(omega_mid - omega_lim) samples; a channel
// whose timing loop can leave its tame range (sdrm_kernels.h "wild channels") may stand still and fill its output buffer
// (clock_recovery_mm.c:103 `oo < output_len`).  A tame channel far out of lock can exceed the first bound too (it advances
// by at least ONE sample per symbol): the conversion kernel's last workgroup then walks on to the real count, the blocking
// calls fetch the tail with a second copy, the pipelined path drops it with a message (sdrm_batch_collect).
static uint32_t symbols_bound(const sdrm_chan_params &p, uint32_t nz) {
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

static int enqueue_call(sdrm_batch_t *b, const sdrm_f2 *d_in, size_t in_stride, const size_t *lens, hipStream_t caller,
                        const sdrm_nco_segment *segs, size_t n_segs);
static int wait_for_all_calls(sdrm_batch_t *b);

static int reset_all_streams(sdrm_batch_t *b);
int sdrm_enqueue_call(sdrm_batch_t *b, const sdrm_f2 *d_in, size_t in_stride, const size_t *lens, hipStream_t caller,
                      const sdrm_nco_segment *segs, size_t n_segs) {
    return enqueue_call(b, d_in, in_stride, lens, caller, segs, n_segs);
}
int sdrm_wait_for_all_calls(sdrm_batch_t *b) { return wait_for_all_calls(b); }
int sdrm_reset_all_streams(sdrm_batch_t *b) { return reset_all_streams(b); }

// Waits for everything this batch has put on the device -- its own streams only: another batch on the same device (a node with
// several batchers per GPU, a server with a handle per client) is not made to drain because this one resets a channel.
static int quiesce(sdrm_batch_t *b) {
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

static int reset_all_streams(sdrm_batch_t *b) {
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
    if (const char *env = getenv("SDRM_HAND_FOLLOW")) {  // measurements: 0 = no placement hold for the two calls behind a hand-off call
        b->hand_follow = atoi(env) != 0;
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
    {
        const char *q = getenv("SDRM_K1_QUAD");
        d.quad_flat = (q != nullptr && strcmp(q, "flat") == 0) ? 1 : 0;
    }
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

static void timing_begin(sdrm_batch_t *b, int which, hipStream_t s, std::pair<hipEvent_t, hipEvent_t> *pr) {
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

static void timing_end(sdrm_batch_t *b, int which, hipStream_t s, const std::pair<hipEvent_t, hipEvent_t> &pr) {
    (void) hipEventRecord(pr.second, s);
    b->lanes[which].pending.push_back(pr);
}

static void timing_collect(sdrm_batch_t *b) {
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
static int ensure_nco(sdrm_batch_t *b) {
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

static int enqueue_call(sdrm_batch_t *b, const sdrm_f2 *d_in, size_t in_stride, const size_t *lens, hipStream_t caller,
                        const sdrm_nco_segment *segs, size_t n_segs) {
    const size_t C = b->plan.design.size();
    const uint64_t i = b->calls;
    const int slot = (int) (i % SDRM_CTL_SLOTS);
    if (b->slot_used[slot]) {
        HIP_TRY(hipEventSynchronize(b->slot_done[slot]));  // that call's kernels have consumed the slot
    }
    sdrm_chunk_ctl *h = b->h_ctl + (size_t) slot * C;
    const uint32_t max_tiles = sdrm::plan_call(b->plan, lens, h);
    sdrm_chunk_ctl *d_ctl = b->d_ctl + (size_t) slot * C;
    sdrm::DeviceBatch d = b->dev;
    if (b->stamp_only_call != 0 && b->stamp_only_call != i + 1) {
        d.k3_stamps = nullptr;
    }
    d.timeline = (b->d_timeline != nullptr && i - b->timeline_first_call < 64) ? b->d_timeline : nullptr;
    d.tl_row = (uint32_t) (i - b->timeline_first_call);
    bool with_nco = false;
    uint32_t nco_max_len = 0;
    if (segs != nullptr && n_segs > 0) {
        int code = ensure_nco(b);
        if (code != 0) {
            return code;
        }
        if (sdrm::plan_nco(b->plan, segs, n_segs, h, b->nco_table) != 0 || b->nco_table.size() > b->nco_seg_cap) {
            // the bookkeeping of plan_call already advanced: an invalid table is a caller bug, keep the stream alive
            // by running the call without correction
            fprintf(stderr, "<3>sdrmodem_hip: invalid NCO segment table, call runs uncorrected\n");
            for (size_t c = 0; c < C; c++) {
                h[c].nco_cnt = 0;
            }
        } else if (!b->nco_table.empty()) {
            with_nco = true;
            memcpy(b->h_nco_segs + (size_t) slot * b->nco_seg_cap, b->nco_table.data(), sizeof(sdrm_nco_seg) * b->nco_table.size());
            for (size_t c = 0; c < C; c++) {
                if (h[c].nco_cnt) {
                    nco_max_len = std::max(nco_max_len, h[c].n_in);
                }
            }
        }
    }
    // the oscillator in front (sdrm_batch_set_pre_offset): one batch per channel and call, a control block of its own
    bool with_pre = false;
    uint32_t pre_max_len = 0;
    if (b->any_pre) {
        sdrm_chunk_ctl *hp = b->h_ctl_pre + (size_t) slot * C;
        sdrm_nco_seg *sp = b->h_pre_segs + (size_t) slot * C;
        for (size_t c = 0; c < C; c++) {
            hp[c] = h[c];
            hp[c].nco_off = (uint32_t) c;
            hp[c].nco_cnt = 0;
            sp[c].len = 0;
            sp[c].step = 0.0f;
            if (b->pre_offset[c] != 0 && h[c].n_in > 0 && h[c].absent == 0) {
                hp[c].nco_cnt = 1;
                sp[c].len = h[c].n_in;
                const float two_pi = (float) (2 * 3.14159265358979323846);  // as plan_nco: sig_source.c:44 in fp32
                sp[c].step = two_pi * (float) b->pre_offset[c] / b->plan.design[c].cfg.sampling_freq;
                h[c].pre = 1;
                with_pre = true;
                pre_max_len = std::max(pre_max_len, h[c].n_in);
            }
        }
    }
    {
        uint64_t sig = 0;
        for (size_t c = 0; c < C; c++) {
            sig += h[c].n_in;
        }
        sdrm_online_tune_before(b, with_nco || with_pre, sig);
    }
    d.nco_segs = with_nco ? b->d_nco_segs + (size_t) slot * b->nco_seg_cap : nullptr;
    d.nco_phase_state = b->d_nco_state;
    d.nco_phase = (b->d_nco_phase2 != nullptr && (i & 1)) ? b->d_nco_phase2 : b->d_nco_phase;
    d.nco_out = b->d_nco_out;
    d.nco_stride = b->in_stride;
    d.nco_phase_stride = SDRM_PHASE_STRIDE(b->in_stride);
    d.ctl = d_ctl;
    d.nonfinite = b->d_flags + (size_t) slot * C;
    d.max_tiles = max_tiles;
    {
        // no channel can produce more symbols than this in the call (grid of the int8 conversion): every symbol advances
        // by at least floor(omega_mid - omega_lim) samples of what the call brings plus the carried ones (< SDRM_CLOCK_HCAP)
        uint32_t most = 0;
        for (size_t c = 0; c < C; c++) {
            most = std::max(most, symbols_bound(b->plan.params[c], h[c].nz));
        }
        d.max_symbols = most;
        b->last_max_symbols = most;
    }
    d.z = (i & 1) ? b->d_z2 : b->d_z;
    d.dcout = (i & 1) ? b->d_dcout2 : b->d_dcout;
    d.out_i8 = out8_of(b, i);
    d.out_len = outlen_of(b, i);
    const int prev2 = (int) ((i + SDRM_CTL_SLOTS - 2) % SDRM_CTL_SLOTS);  // the call that last used these buffers
    const bool have_prev2 = i >= 2;

    // ---- in-call hand-off?  A call that meets an idle batch -- every blocking call, the first of a pipelined run -- cannot
    // hide its front-end and DC blocker behind an earlier call's clock stage: its three stages are made resident together
    // instead, each starting on the first finished pieces of the one in front (tile stamps / output counts, sdrm_launch.h).
    // Waiting workgroups hold their CUs, so this is bounded: at most 192 of them (every batch of the 16 x 1024 clock-stage shape: 160 at
    // 1280 channels, one per CU; batches of the 32 x 512 shape up to 2048 channels -- the front-end keeps the other CUs and the room beside the DC workgroups; measured 1024 channels
    // 5.71 -> 3.07 ms per blocking call, the limit had been 64) when a DC workgroup leaves room for a front-end workgroup beside it
    // (then the front-end can always be placed, whatever else waits on the chip), 16 when it does not; the
    // clock stage is launched only when the DC stage's workgroups are resident, the front-end only when both are -- then the
    // front-end, which waits for nobody, always finds a CU, the DC stage waits only for the front-end and the clock stage
    // only for the DC stage.  Every wait in the kernels is bounded besides (a void call, loudly, never a hung device).
    bool hand = false;
    if (b->hand_allowed && b->n_gen == 0 && (b->serial || b->d_placed != nullptr) && max_tiles > 0) {
        const bool idle = b->last_slot < 0 || hipEventQuery(b->slot_done[b->last_slot]) == hipSuccess;
        const unsigned waiting = sdrm::clock_workgroups(d) + (d.any_dc ? sdrm::dc_workgroups(d) : 0u);
        const bool room = !d.any_dc || (size_t) d.dc_lds + sdrm::k1_lds_bytes(d.t1_max, d.t2_max) <= 160 * 1024;
        // (a DC workgroup that fills its CU -- long boxcars -- leaves the front-end no room beside it: then only a handful may wait)
        unsigned most = room ? 192u : 16u;
        if (const char *env = getenv("SDRM_HAND_MAX_WAITING")) {  // measurements (profiles/r05_incall_handoff.txt)
            most = (unsigned) atoi(env);
        }
        hand = idle && waiting <= most && sdrm::clock_shape_hands_off(d);
        if (hand && b->serial) {
            // a plain handle keeps its stages on one stream (a server holds one per client); a call long enough for the overlap
            // to pay (SDRM_HAND_SERIAL_MIN_NZ) gets two side streams, created when the first such call comes
            uint32_t longest = 0;
            for (size_t c = 0; c < C; c++) {
                longest = std::max(longest, h[c].nz);
            }
            hand = longest >= SDRM_HAND_SERIAL_MIN_NZ;
            if (hand && b->s_hand_clock == nullptr) {
                int prio_low = 0, prio_high = 0;
                (void) hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
                if (hipStreamCreateWithPriority(&b->s_hand_dc, hipStreamNonBlocking, (prio_low + prio_high) / 2) != hipSuccess ||
                    hipStreamCreateWithPriority(&b->s_hand_clock, hipStreamNonBlocking, prio_high) != hipSuccess) {
                    (void) hipGetLastError();
                    hand = false;
                }
            }
        }
    }
    // the streams of this call's DC and clock stages
    const bool hand_side = hand && b->serial;
    hipStream_t s_dc = hand_side ? b->s_hand_dc : b->s_dc;
    if (b->hand_side_last && b->last_slot >= 0) {
        HIP_TRY(hipStreamWaitEvent(b->s_front, b->slot_done[b->last_slot], 0));  // the previous call ended on a side stream
    }
    b->hand_side_last = hand_side;
    if (hand && (b->d_hand_prog == nullptr || b->hand_tiles_cap < max_tiles)) {
        // stamps and counts: allocated on first use, grown when a call has more tiles (the batch is idle here)
        (void) hipFree(b->d_hand_tiles);
        b->d_hand_tiles = nullptr;
        b->hand_tiles_cap = 0;
        const uint32_t cap = std::max<uint32_t>(max_tiles, 64u);
        if (dev_alloc_zero(&b->d_hand_tiles, C * (size_t) cap) != 0 || (b->d_hand_prog == nullptr && dev_alloc_zero(&b->d_hand_prog, C) != 0)) {
            hand = false;
        } else {
            b->hand_tiles_cap = cap;
        }
    }
    if (hand) {
        d.handoff = 1;
        // (a count of its own, never reset: the stamps of earlier calls -- the calibration's, before reset_all_streams put the call
        // count back to 0 -- must never look like this call's; found by the batcher soak, profiles/r05_soak.txt)
        if (b->hand_epoch != 0 && b->hand_epoch % 0xfffffff0ull == 0) {
            // the 32-bit stamp values start over (weeks of calls): no stamp of the last time round may survive (the batch is idle)
            HIP_TRY(hipMemset(b->d_hand_tiles, 0, sizeof(uint32_t) * C * (size_t) b->hand_tiles_cap));
            HIP_TRY(hipMemset(b->d_hand_prog, 0, sizeof(unsigned long long) * C));
            HIP_TRY(hipStreamSynchronize(nullptr));
        }
        d.epoch = (uint32_t) (b->hand_epoch++ % 0xfffffff0ull) + 1u;
        d.hand_tiles = b->d_hand_tiles;
        d.hand_tiles_cap = b->hand_tiles_cap;
        d.hand_prog = b->d_hand_prog;
        b->hand_used = true;
        b->hand_calls++;
        b->last_hand_call = i;
    }

    // ---- NCO phases: need neither the input nor an earlier stage, only the phase buffer released by the mix of call i-2
    const bool nco_aside = with_nco && b->s_nco != b->s_front;
    if (nco_aside) {
        if (have_prev2) {
            HIP_TRY(hipStreamWaitEvent(b->s_nco, b->ev_front[prev2], 0));
        }
        HIP_TRY(hipMemcpyAsync(d_ctl, h, sizeof(sdrm_chunk_ctl) * C, hipMemcpyHostToDevice, b->s_nco));
        HIP_TRY(hipMemcpyAsync(b->d_nco_segs + (size_t) slot * b->nco_seg_cap, b->h_nco_segs + (size_t) slot * b->nco_seg_cap,
                               sizeof(sdrm_nco_seg) * b->nco_table.size(), hipMemcpyHostToDevice, b->s_nco));
        sdrm::launch_nco_phase(d, b->s_nco);
        HIP_TRY(hipEventRecord(b->ev_phase[slot], b->s_nco));
    }

    // ---- front-end: needs the input, and z[i&1] released by its readers of call i-2
    HIP_TRY(hipEventRecord(b->ev_in[slot], caller));
    HIP_TRY(hipStreamWaitEvent(b->s_front, b->ev_in[slot], 0));
    if (have_prev2) {
        HIP_TRY(hipStreamWaitEvent(b->s_front, d.any_dc ? b->ev_dc[prev2] : b->slot_done[prev2], 0));
        if (d.any_dc && b->any_nodc) {
            HIP_TRY(hipStreamWaitEvent(b->s_front, b->slot_done[prev2], 0));  // channels without DC: K3 reads z
        }
    }
    d.placed = b->d_placed;
    if (!b->serial && i >= 3 && b->hold_front) {
        // let the clock stage of call i-2 (released by the end of call i-3's) take its CUs before this grid floods the chip
        HIP_TRY(hipStreamWaitEvent(b->s_front, b->slot_done[(i + SDRM_CTL_SLOTS - 3) % SDRM_CTL_SLOTS], 0));
        sdrm::launch_hold_until(b->d_placed + 1, b->k3_placed_after[(i + SDRM_CTL_SLOTS - 2) % SDRM_CTL_SLOTS], 100, b->s_front);
    }
    if (!b->serial && !hand && d.any_dc && b->d_placed != nullptr && b->hand_calls > 0 && i - b->last_hand_call <= 2 &&
        b->hand_follow) {
        // The two calls behind a hand-off call: this front-end and the DC stage of the call before it are released by the same event
        // (the hand-off call's DC stage ending), and the hand-off call's companion grid, started on an empty chip, sits on every
        // CU until its clock stage ends.  If this grid covers the chip first, a DC workgroup (117 KB of LDS, 11 waves) may find
        // no CU until the companion grid leaves -- seen in 20 of 240 20-call runs, in bursts (0 to 14 of a process's 40): the second
        // call's DC stage 2.9 ms instead of 1.1, its clock stage 1.2 ms late, the run 2.3 % slower.  With the DC workgroups placed
        // first: 3 of 240 (profiles/r05_incall_handoff.txt).  Bounded.
        sdrm::launch_hold_until(b->d_placed + 0, b->k2_placed_target, 150, b->s_front);
    }
    if (nco_aside) {
        HIP_TRY(hipStreamWaitEvent(b->s_front, b->ev_phase[slot], 0));
    } else {
        HIP_TRY(hipMemcpyAsync(d_ctl, h, sizeof(sdrm_chunk_ctl) * C, hipMemcpyHostToDevice, b->s_front));
    }
    if (hand) {
        HIP_TRY(hipEventRecord(b->ev_ctl[slot], b->s_front));  // what the other stages need before they can start: the control block
        // The two stages behind must have their workgroups placed BEFORE the front-end's grid covers the chip (a DC workgroup
        // needs 117 KB of a CU's LDS, a clock-stage workgroup 141 KB: neither finds that between front-end workgroups, and
        // the dispatcher reserves nothing -- measured: the DC stage started at 0.56 ms of a 0.62 ms front-end).  Bounded.
        // (a one-stream batch has one channel: a few dozen front-end workgroups, nothing to hold back)
        if (d.any_dc && b->d_placed != nullptr) {
            sdrm::launch_hold_until(b->d_placed + 0, b->k2_placed_target + sdrm::dc_workgroups(d), 400, b->s_front);
        }
        if (b->d_placed != nullptr) {
            sdrm::launch_hold_until(b->d_placed + 1, b->k3_placed_target + sdrm::clock_workgroups(d), 400, b->s_front);
        }
    }
    if (with_pre) {
        sdrm::DeviceBatch dp = d;
        dp.ctl = b->d_ctl_pre + (size_t) slot * C;
        dp.nco_segs = b->d_pre_segs + (size_t) slot * C;
        dp.nco_phase_state = b->d_pre_state;
        dp.nco_phase = b->d_pre_phase;
        HIP_TRY(hipMemcpyAsync(b->d_ctl_pre + (size_t) slot * C, b->h_ctl_pre + (size_t) slot * C, sizeof(sdrm_chunk_ctl) * C,
                               hipMemcpyHostToDevice, b->s_front));
        HIP_TRY(hipMemcpyAsync(b->d_pre_segs + (size_t) slot * C, b->h_pre_segs + (size_t) slot * C, sizeof(sdrm_nco_seg) * C,
                               hipMemcpyHostToDevice, b->s_front));
        sdrm::launch_nco_phase(dp, b->s_front);
        sdrm::launch_nco_mix(dp, d_in, in_stride, pre_max_len, b->s_front);
    }
    if (with_nco) {
        if (!nco_aside) {
            HIP_TRY(hipMemcpyAsync(b->d_nco_segs + (size_t) slot * b->nco_seg_cap, b->h_nco_segs + (size_t) slot * b->nco_seg_cap,
                                   sizeof(sdrm_nco_seg) * b->nco_table.size(), hipMemcpyHostToDevice, b->s_front));
            sdrm::launch_nco_phase(d, b->s_front);
        }
        sdrm::launch_nco_mix(d, d_in, in_stride, nco_max_len, b->s_front);
    }
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (b->timing) {
        timing_begin(b, 0, b->s_front, &ev);
    }
    sdrm::launch_front(d, d_in, in_stride, b->s_front);
    if (b->timing) {
        timing_end(b, 0, b->s_front, ev);
    }
    HIP_TRY(hipEventRecord(b->ev_front[slot], b->s_front));

    // ---- DC blocker: needs z of this call, and dcout[i&1] released by the clock stage of call i-2
    if (d.any_dc) {
        HIP_TRY(hipStreamWaitEvent(s_dc, hand ? b->ev_ctl[slot] : b->ev_front[slot], 0));
        if (have_prev2) {
            HIP_TRY(hipStreamWaitEvent(s_dc, b->slot_done[prev2], 0));
        }
        b->k2_placed_target += sdrm::dc_workgroups(d);
        if (b->timing) {
            timing_begin(b, 1, s_dc, &ev);
        }
        sdrm::launch_dc(d, s_dc);
        sdrm::launch_dc_generic(d, s_dc);
        if (b->timing) {
            timing_end(b, 1, s_dc, ev);
        }
        HIP_TRY(hipEventRecord(b->ev_dc[slot], s_dc));
    }

    // ---- clock recovery + int8
    hipStream_t s_clock = hand_side ? b->s_hand_clock : b->s_clock;
    if (hand) {
        HIP_TRY(hipStreamWaitEvent(s_clock, b->ev_ctl[slot], 0));
        if (d.any_dc && b->d_placed != nullptr) {
            // ... and the DC stage's workgroups resident (they count themselves in, k2_dc): bounded, ~2 ms
            sdrm::launch_hold_until(b->d_placed + 0, b->k2_placed_target, 2000, s_clock);
        }
    } else {
        HIP_TRY(hipStreamWaitEvent(s_clock, d.any_dc ? b->ev_dc[slot] : b->ev_front[slot], 0));
        if (d.any_dc && b->any_nodc) {
            HIP_TRY(hipStreamWaitEvent(s_clock, b->ev_front[slot], 0));
        }
    }
    if (b->out_busy[i & 1]) {
        HIP_TRY(hipStreamWaitEvent(s_clock, b->ev_out_free[i & 1], 0));  // that output set is still being copied back
    }
    b->k3_placed_target += sdrm::clock_workgroups(d);
    b->k3_placed_after[slot] = b->k3_placed_target;
    if (b->timing) {
        timing_begin(b, 2, s_clock, &ev);
    }
    if (b->d_k3_done != nullptr) {
        d.k3_done = b->d_k3_done;
        b->k3_done_target += sdrm::clock_workgroups(d);
    }
    if (b->company_blocks > 0) {
        // starts when the clock stage may start, leaves when the clock stage's last workgroup has
        HIP_TRY(hipEventRecord(b->ev_company, s_clock));
        HIP_TRY(hipStreamWaitEvent(b->s_company, b->ev_company, 0));
        sdrm::launch_clock_company(d, b->k3_done_target, b->company_blocks, b->company_rounds, b->company_nops, b->s_company);
    }
    sdrm::launch_clock(d, s_clock);
    sdrm::launch_clock_generic(d, s_clock);
    if (b->timing) {
        timing_end(b, 2, s_clock, ev);
    }
    HIP_TRY(hipGetLastError());
    if (hand) {
        // the clock stage can be through before the DC kernel has written its last state back: the call is done when both are
        HIP_TRY(hipStreamWaitEvent(s_clock, d.any_dc ? b->ev_dc[slot] : b->ev_front[slot], 0));
        if (d.any_dc && b->any_nodc) {
            HIP_TRY(hipStreamWaitEvent(s_clock, b->ev_front[slot], 0));
        }
    }
    HIP_TRY(hipEventRecord(b->slot_done[slot], s_clock));
    sdrm_online_tune_after(b, s_clock);
    b->slot_used[slot] = true;
    b->last_slot = slot;
    b->calls++;
    if (b->timing && b->lanes[0].pending.size() > 4096) {
        timing_collect(b);
    }
    return 0;
}

// make `stream` wait for the results of the most recent call (device-side dependency, no host wait)
extern "C" int sdrm_batch_wait(sdrm_batch *b, void *stream) {
    if (b == nullptr) {
        return -1;
    }
    if (b->last_slot >= 0) {
        HIP_TRY(hipStreamWaitEvent((hipStream_t) stream, b->slot_done[b->last_slot], 0));
    }
    return 0;
}

// make `stream` wait until the most recent call has READ its input (front-end and history roll done): the caller may
// then refill or free the input buffer on that stream without waiting for the rest of the call
extern "C" int sdrm_batch_wait_input(sdrm_batch *b, void *stream) {
    if (b == nullptr) {
        return -1;
    }
    if (b->last_slot >= 0) {
        HIP_TRY(hipStreamWaitEvent((hipStream_t) stream, b->ev_front[b->last_slot], 0));
    }
    return 0;
}

// A stage of a hand-off call that gave up waiting for the stage in front of it (bounded looks, ~2 s) raises the word at
// d_counters[1]: that call's results cannot be trusted (its clock stage answers with the count SDRM_OUT_LEN_FAILED), and the
// batch is in error for good.  The word is final only once the LAST hand-off call has finished: a look that comes earlier
// (sdrm_batch_collect of an older call while a newer hand-off call is still running) keeps `hand_used` set, so that the look
// behind that call is not skipped.
static int check_device_error(sdrm_batch_t *b) {
    if (b->device_error == 0 && b->hand_used) {
        const int slot = (int) (b->last_hand_call % SDRM_CTL_SLOTS);
        const bool over = b->calls > b->last_hand_call + SDRM_CTL_SLOTS - 1 /* its slot has been waited for and reused */ ||
                          hipEventQuery(b->slot_done[slot]) == hipSuccess;
        uint32_t word = 0;
        HIP_TRY(hipMemcpy(&word, b->d_counters + 1, sizeof(word), hipMemcpyDeviceToHost));
        if (word != 0) {
            b->device_error = -ETIMEDOUT;
            fprintf(stderr, "<3>sdrmodem_hip: a stage timed out waiting for the stage in front of it inside a call; the batch is unusable\n");
        }
        if (over) {
            b->hand_used = false;
        }
    }
    return b->device_error;
}

// the host waits until every enqueued call has finished
static int wait_for_all_calls(sdrm_batch_t *b) {
    if (b->device_error != 0) {
        return b->device_error;
    }
    if (b->last_slot >= 0) {
        HIP_TRY(hipEventSynchronize(b->slot_done[b->last_slot]));
        return check_device_error(b);
    }
    return 0;
}

// block the host until every enqueued call has finished
extern "C" int sdrm_batch_sync(sdrm_batch *b) {
    if (b == nullptr) {
        return -1;
    }
    return wait_for_all_calls(b);
}

// calls enqueued with the in-call hand-off since the batch was created
extern "C" int sdrm_batch_handoff_calls(sdrm_batch *b, uint64_t *count) {
    if (b == nullptr || count == nullptr) {
        return -1;
    }
    *count = b->hand_calls;
    return 0;
}

// channel-calls the clock stage ran from global memory (sdrm_kernels.h "wild channels"), since the batch was created
extern "C" int sdrm_batch_wild_calls(sdrm_batch *b, uint64_t *count) {
    if (b == nullptr || count == nullptr) {
        return -1;
    }
    HIP_TRY(hipSetDevice(b->device));
    const int code = wait_for_all_calls(b);
    if (code != 0) {
        return code;
    }
    uint32_t word = 0;
    HIP_TRY(hipMemcpy(&word, b->d_counters, sizeof(word), hipMemcpyDeviceToHost));
    *count = word;
    return 0;
}

extern "C" int sdrm_batch_process_device(sdrm_batch *b, const void *d_input, size_t in_stride, const size_t *input_lens,
                                         void *stream) {
    if (b == nullptr || input_lens == nullptr) {
        return -1;
    }
    HIP_TRY(hipSetDevice(b->device));
    // the K1 LDS request can exceed the 64 KiB default for long filters
    return enqueue_call(b, (const sdrm_f2 *) d_input, in_stride, input_lens, (hipStream_t) stream, nullptr, 0);
}

extern "C" int sdrm_batch_process_device_nco(sdrm_batch *b, const void *d_input, size_t in_stride, const size_t *input_lens,
                                             const sdrm_nco_segment *segments, size_t n_segments, void *stream) {
    if (b == nullptr || input_lens == nullptr) {
        return -1;
    }
    HIP_TRY(hipSetDevice(b->device));
    return enqueue_call(b, (const sdrm_f2 *) d_input, in_stride, input_lens, (hipStream_t) stream, segments, n_segments);
}

extern "C" int sdrm_batch_last_mixed(sdrm_batch *b, size_t c, float *dst, size_t cap, size_t *len) {
    if (b == nullptr || c >= b->plan.design.size() || b->d_nco_out == nullptr || b->last_slot < 0) {
        return -1;
    }
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipDeviceSynchronize());
    const sdrm_chunk_ctl &k = b->h_ctl[(size_t) b->last_slot * b->plan.design.size() + c];
    const size_t n = k.nco_cnt ? k.n_in : 0;
    if (len) {
        *len = n;
    }
    if (dst != nullptr && n > 0) {
        HIP_TRY(hipMemcpy(dst, b->d_nco_out + c * (size_t) b->in_stride, sizeof(sdrm_f2) * std::min(n, cap), hipMemcpyDeviceToHost));
    }
    return 0;
}

struct sdrm_doppler_t {
    sdrm::DopplerPlanner planner;
};

extern "C" int sdrm_doppler_create(uint64_t sampling_freq, sdrm_doppler_shift_fn fn, void *user, sdrm_doppler **out) {
    if (sampling_freq == 0 || fn == nullptr || out == nullptr) {
        return -1;
    }
    sdrm_doppler_t *d = new sdrm_doppler_t();
    d->planner.interval = sampling_freq;     // one update per second (doppler.c:84)
    d->planner.in_interval = sampling_freq;  // "expired": the first batch evaluates the shift (doppler.c:85)
    d->planner.fn = fn;
    d->planner.user = user;
    *out = d;
    return 0;
}

extern "C" size_t sdrm_doppler_plan(sdrm_doppler *d, uint32_t channel, size_t input_len, sdrm_nco_segment *segments, size_t cap) {
    if (d == nullptr || segments == nullptr) {
        return 0;
    }
    return d->planner.plan(channel, input_len, segments, cap);
}

extern "C" void sdrm_doppler_destroy(sdrm_doppler *d) { delete d; }

extern "C" int sdrm_batch_device_outputs(sdrm_batch *b, void **d_out_i8, size_t *out_stride, void **d_out_len,
                                         void **d_out_f32) {
    if (b == nullptr) {
        return -1;
    }
    const uint64_t last = b->calls ? b->calls - 1 : 0;
    if (d_out_i8) {
        *d_out_i8 = out8_of(b, last);
    }
    if (out_stride) {
        *out_stride = b->dev.out_stride;
    }
    if (d_out_len) {
        *d_out_len = outlen_of(b, last);
    }
    if (d_out_f32) {
        *d_out_f32 = b->d_outf;
    }
    return 0;
}

static int ensure_host_staging(sdrm_batch_t *b) {
    const size_t C = b->plan.design.size();
    if (b->d_in == nullptr) {
        int code = dev_alloc_zero(&b->d_in, C * (size_t) b->in_stride);
        if (code != 0) {
            return code;
        }
    }
    if (b->h_out8 == nullptr) {
        if (hipHostMalloc((void **) &b->h_out8, C * (size_t) b->dev.out_stride) != hipSuccess) {
            return -ENOMEM;
        }
    }
    return 0;
}

static int process_host(sdrm_batch *b, const sdrm_cf32 *const *inputs, const size_t *input_lens, int8_t **outputs,
                        size_t *output_lens, const sdrm_nco_segment *segs, size_t n_segs);

extern "C" int sdrm_batch_process(sdrm_batch *b, const sdrm_cf32 *const *inputs, const size_t *input_lens,
                                  int8_t **outputs, size_t *output_lens) {
    return process_host(b, inputs, input_lens, outputs, output_lens, nullptr, 0);
}

extern "C" int sdrm_batch_process_nco(sdrm_batch *b, const sdrm_cf32 *const *inputs, const size_t *input_lens,
                                      const sdrm_nco_segment *segments, size_t n_segments, int8_t **outputs,
                                      size_t *output_lens) {
    return process_host(b, inputs, input_lens, outputs, output_lens, segments, n_segments);
}

// ---- one-channel blocking call through a replayed graph ----------------------------------------------------------------
// The reference's own usage (one handle per DSP thread, perf_fsk_modem.c: 100 calls of 4096 samples) is bound by launch
// and synchronisation overhead here, not by the kernels: eight enqueue calls and the gaps between five small kernels.
// Everything that changes from call to call lives in memory the graph reads through fixed addresses -- the staged
// input, the control record written by plan_call, the results -- so a graph built once per input length is replayed.
// Grids and copy widths are those of the most outputs a call of that length can have (the decimation phase moves
// nz by one between calls); workgroups beyond the call's own tile count leave at once.
#define SDRM_GRAPH_MAX_SAMPLES 65536u
static const int SG_SLOT = SDRM_CTL_SLOTS - 1;

static bool serial_call_hands_off(const sdrm_batch_t *b, size_t n) {
    const sdrm_chan_params &p = b->plan.params[0];
    return b->hand_allowed && b->n_gen == 0 && n / p.decim >= SDRM_HAND_SERIAL_MIN_NZ;
}

static bool serial_graph_usable(const sdrm_batch_t *b, size_t n, const sdrm_nco_segment *segs) {
    return getenv("SDRM_NO_GRAPH") == nullptr &&  // escape hatch for measurements
           !serial_call_hands_off(b, n) && !b->any_pre &&
           !b->sg_broken && b->serial && b->plan.design.size() == 1 && b->n_gen == 0 && segs == nullptr && !b->timing &&
           b->d_timeline == nullptr && b->dev.k3_stamps == nullptr && b->d_out8_b == nullptr && b->calls > 0 && n > 0 &&
           n <= SDRM_GRAPH_MAX_SAMPLES && n <= b->plan.params[0].max_len;
}

static int serial_graph_build(sdrm_batch_t *b, size_t n, const sdrm_chunk_ctl *h) {
    if (b->sg_exec != nullptr) {
        (void) hipGraphExecDestroy(b->sg_exec);
        b->sg_exec = nullptr;
    }
    const sdrm_chan_params &p = b->plan.params[0];
    sdrm::DeviceBatch d = b->dev;
    d.k3_stamps = nullptr;
    d.timeline = nullptr;
    d.placed = nullptr;
    d.nco_segs = nullptr;
    d.ctl = b->d_ctl + (size_t) SG_SLOT;
    d.nonfinite = b->d_flags + (size_t) SG_SLOT;
    const uint32_t nz_cap = (uint32_t) ((n + p.decim - 1) / p.decim) + 1u;
    d.max_tiles = (nz_cap + p.tile_m - 1) / p.tile_m;
    const uint32_t most = symbols_bound(p, nz_cap);
    d.max_symbols = most;
    d.z = b->d_z;
    d.dcout = b->d_dcout;
    d.out_i8 = b->d_out8;
    d.out_len = b->d_outlen;
    b->sg_width = (uint32_t) std::min<size_t>(most, b->dev.out_stride);
    // The graph is BUILT, node by node, not captured from a stream: while any stream of the process is being captured
    // ROCm fails legacy-stream calls of every other thread (another client's handle being created or reset), whatever
    // the capture mode -- and fails the capture with them.
    hipGraph_t graph = nullptr;
    if (hipGraphCreate(&graph, 0) != hipSuccess) {
        (void) hipGetLastError();
        return -1;
    }
    bool ok = true;
    typedef std::vector<hipGraphNode_t> deps_t;
    auto add_copy = [&](const deps_t &deps, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) -> hipGraphNode_t {
        hipGraphNode_t node = nullptr;
        ok = ok && bytes > 0 && hipGraphAddMemcpyNode1D(&node, graph, deps.data(), deps.size(), dst, src, bytes, kind) == hipSuccess;
        return node;
    };
    const sdrm_f2 *d_in = b->d_in;
    size_t in_stride = b->in_stride;
    // a kernel takes as many of the three arguments as it declares; nullptr: nothing to launch (the dependences pass through)
    auto add_kernel = [&](const deps_t &deps, const sdrm::KernelLaunch &k) -> hipGraphNode_t {
        if (k.func == nullptr || !ok) {
            return nullptr;
        }
        void *args[3] = {(void *) &d, (void *) &d_in, (void *) &in_stride};
        hipKernelNodeParams kp = {};
        kp.func = const_cast<void *>(k.func);
        kp.gridDim = k.grid;
        kp.blockDim = k.block;
        kp.sharedMemBytes = (unsigned) k.lds;
        kp.kernelParams = args;
        kp.extra = nullptr;
        hipGraphNode_t node = nullptr;
        ok = hipGraphAddKernelNode(&node, graph, deps.data(), deps.size(), &kp) == hipSuccess;
        return node;
    };
    hipGraphNode_t n_in = add_copy({}, b->d_in, b->h_in_stage, n * sizeof(sdrm_f2), hipMemcpyHostToDevice);
    hipGraphNode_t n_ctl = add_copy({n_in}, b->d_ctl + (size_t) SG_SLOT, h, sizeof(sdrm_chunk_ctl), hipMemcpyHostToDevice);
    hipGraphNode_t last = n_ctl;
    // (The stages stay a chain here.  The in-call hand-off as three parallel branches of a graph was measured and lost --
    // 4096 samples 157 -> 376 us, 65536: 1340 -> 1985 -- while on the handle's stream plus two side streams it wins from
    // ~24000 samples on: calls that long leave the graph to it, process_host.)
    if (ok) {
        for (const sdrm::KernelLaunch &k : {sdrm::describe_front(d), sdrm::describe_dc(d), sdrm::describe_clock(d), sdrm::describe_quantize(d)}) {
            hipGraphNode_t node = add_kernel({last}, k);
            last = node ? node : last;
        }
        last = add_copy({last}, b->h_outlen, d.out_len, sizeof(uint32_t), hipMemcpyDeviceToHost);
    }
    if (ok && b->sg_width > 0) {
        last = add_copy({last}, b->h_out8, d.out_i8, b->sg_width, hipMemcpyDeviceToHost);
    }
    if (!ok) {
        (void) hipGraphDestroy(graph);
        (void) hipGetLastError();
        return -1;
    }
    const hipError_t inst = hipGraphInstantiate(&b->sg_exec, graph, nullptr, nullptr, 0);
    (void) hipGraphDestroy(graph);
    if (inst != hipSuccess) {
        b->sg_exec = nullptr;
        (void) hipGetLastError();
        return -1;
    }
    b->sg_len = n;
    return 0;
}

// returns 1 when the call was served, 0 when the caller should take the plain path, < 0 on a device error
static int serial_graph_call(sdrm_batch_t *b, const sdrm_cf32 *input, size_t n, int8_t **output, size_t *output_len) {
    if (b->h_in_stage == nullptr &&
        hipHostMalloc((void **) &b->h_in_stage, (size_t) SDRM_GRAPH_MAX_SAMPLES * sizeof(sdrm_f2)) != hipSuccess) {
        (void) hipGetLastError();
        b->sg_broken = true;
        return 0;
    }
    {
        const int code = wait_for_all_calls(b);  // an asynchronous device-resident call may still run
        if (code != 0) {
            return code;
        }
    }
    sdrm_chunk_ctl *h = b->h_ctl + (size_t) SG_SLOT;
    if (b->sg_exec == nullptr || b->sg_len != n) {
        // build BEFORE the call's bookkeeping advances: a failure leaves the plain path an untouched stream
        sdrm_chunk_ctl probe = {};
        *h = probe;
        if (serial_graph_build(b, n, h) != 0) {
            b->sg_broken = true;
            return 0;
        }
    }
    memcpy(b->h_in_stage, input, n * sizeof(sdrm_f2));
    const size_t lens[1] = {n};
    (void) sdrm::plan_call(b->plan, lens, h);
    HIP_TRY(hipGraphLaunch(b->sg_exec, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    const uint32_t got = b->h_outlen[0];
    if (got > b->sg_width) {  // more symbols than the bound the graph's copy was sized for: the rest, now
        HIP_TRY(hipMemcpy(b->h_out8 + b->sg_width, b->d_out8 + b->sg_width, got - b->sg_width, hipMemcpyDeviceToHost));
    }
    b->last_lens[0] = got;
    b->last_max_symbols = b->sg_width;
    b->last_slot = -1;  // nothing of this call is left in flight
    b->calls++;
    *output = b->h_out8;
    *output_len = got;
    return 1;
}

static int process_host(sdrm_batch *b, const sdrm_cf32 *const *inputs, const size_t *input_lens, int8_t **outputs,
                        size_t *output_lens, const sdrm_nco_segment *segs, size_t n_segs) {
    if (b == nullptr || input_lens == nullptr || outputs == nullptr || output_lens == nullptr) {
        return -1;
    }
    HIP_TRY(hipSetDevice(b->device));
    int code = ensure_host_staging(b);
    if (code != 0) {
        return code;
    }
    const size_t C = b->plan.design.size();
    const bool repeats = C == 1 && input_lens[0] == b->sg_prev_len;  // ragged streams are not worth a capture per call
    if (C == 1) {
        b->sg_prev_len = input_lens[0];
    }
    if (repeats && inputs != nullptr && inputs[0] != nullptr && serial_graph_usable(b, input_lens[0], segs)) {
        const int served = serial_graph_call(b, inputs[0], input_lens[0], &outputs[0], &output_lens[0]);
        if (served != 0) {
            return served < 0 ? served : 0;
        }
    }
    for (size_t c = 0; c < C; c++) {
        size_t n = input_lens[c];
        if (n == 0 || n > b->plan.params[c].max_len || inputs == nullptr || inputs[c] == nullptr) {
            continue;
        }
        HIP_TRY(hipMemcpyAsync(b->d_in + c * (size_t) b->in_stride, inputs[c], n * sizeof(sdrm_f2), hipMemcpyHostToDevice,
                               b->stream));
    }
    code = enqueue_call(b, b->d_in, b->in_stride, input_lens, b->stream, segs, n_segs);
    if (code != 0) {
        return code;
    }
    HIP_TRY(hipStreamWaitEvent(b->stream, b->slot_done[b->last_slot], 0));
    const uint64_t me = b->calls - 1;
    // counts and soft bits come back behind ONE synchronisation: every channel's copy is as long as the most symbols any
    // channel can have produced in this call (known before the call runs), the counts say how much of it is valid
    HIP_TRY(hipMemcpyAsync(b->h_outlen, outlen_of(b, me), sizeof(uint32_t) * C, hipMemcpyDeviceToHost, b->stream));
    const size_t width = std::min<size_t>(b->last_max_symbols, b->dev.out_stride);
    if (width > 0) {
        if (C == 1) {
            HIP_TRY(hipMemcpyAsync(b->h_out8, out8_of(b, me), width, hipMemcpyDeviceToHost, b->stream));
        } else {
            HIP_TRY(hipMemcpy2DAsync(b->h_out8, b->dev.out_stride, out8_of(b, me), b->dev.out_stride, width, C,
                                     hipMemcpyDeviceToHost, b->stream));
        }
    }
    HIP_TRY(hipStreamSynchronize(b->stream));
    if (b->hand_used) {
        for (size_t c = 0; c < C; c++) {
            if (b->h_outlen[c] == SDRM_OUT_LEN_FAILED) {  // a stage gave up waiting inside the call: says so, once, and fails the batch
                const int failed = check_device_error(b);
                return failed != 0 ? failed : -EIO;
            }
        }
        b->hand_used = false;  // every count is a real one: nobody gave up
    }
    for (size_t c = 0; c < C; c++) {
        const uint32_t n = b->h_outlen[c];
        if (n > width) {  // a loop far out of lock produced more symbols than the bound (symbols_bound): the rest, now
            HIP_TRY(hipMemcpy(b->h_out8 + c * (size_t) b->dev.out_stride + width, out8_of(b, me) + c * (size_t) b->dev.out_stride + width,
                              n - width, hipMemcpyDeviceToHost));
        }
        b->last_lens[c] = n;
        outputs[c] = b->h_out8 + c * (size_t) b->dev.out_stride;
        output_lens[c] = n;
    }
    return 0;
}


// --- pipelined host-buffer path -------------------------------------------------------------------------------------
// The reference hands fsk_demod_process a host buffer (src/dsp/fsk_demod.h:13), so the drop-in rate is bounded by the
// host link.  Here the producers write IQ straight into a pinned arena ([slots][C][in_stride]), one slot per call; a
// call is one large copy (56 GB/s measured, vs 30 GB/s for one copy per channel) on its own stream, so the copy of
// call k+1 overlaps the kernels of call k, and the int8 results come back through two pinned result sets.

// Copy the results of call k back to its pinned result set.  The copy rides on the DC stage's stream: HIP multiplexes
// streams onto a few hardware queues, and a separate copy stream that lands on the queue of the copy-in stream holds the
// next call's input behind this call's results (seen: the whole pipeline serialised).  It is enqueued only after the
// DC stage of call k+1 (or at collect time), i.e. in front of K2 of call k+2, which waits for the clock stage of call k
// anyway because it reuses its input buffer -- the 0.2 ms copy then delays nothing.
static int issue_copy_back(sdrm_batch_t *b, uint64_t k) {
    const size_t C = b->plan.design.size();
    const int set = (int) (k % SDRM_RES_SETS), par = (int) (k & 1);
    hipStream_t back = b->s_dc;
    HIP_TRY(hipStreamWaitEvent(back, b->slot_done[b->back_slot[set]], 0));
    HIP_TRY(hipMemcpyAsync(b->h_reslen[set], outlen_of(b, k), sizeof(uint32_t) * C, hipMemcpyDeviceToHost, back));
    if (b->back_rows[set] > 0 && b->back_width[set] > 0) {
        const size_t at = (size_t) b->back_first[set] * b->dev.out_stride;
        HIP_TRY(hipMemcpy2DAsync(b->h_res8[set] + at, b->dev.out_stride, out8_of(b, k) + at, b->dev.out_stride, b->back_width[set],
                                 b->back_rows[set], hipMemcpyDeviceToHost, back));
    }
    HIP_TRY(hipEventRecord(b->ev_res[set], back));
    HIP_TRY(hipEventRecord(b->ev_out_free[par], back));
    b->out_busy[par] = true;
    b->res_width[set] = b->back_width[set];
    b->back_pending = false;
    return 0;
}

extern "C" int sdrm_batch_arena(sdrm_batch *b, size_t slots, sdrm_cf32 **base, size_t *chan_stride, size_t *slot_stride) {
    if (b == nullptr || slots < 2 || base == nullptr) {
        return -1;
    }
    HIP_TRY(hipSetDevice(b->device));
    const size_t C = b->plan.design.size();
    const size_t slot_samples = C * (size_t) b->in_stride;
    if (b->h_arena == nullptr) {
        HIP_TRY(hipDeviceSynchronize());  // the output-set switch below must not race a running call
        // No companion grid on the pipelined host path (unless SDRM_K3_COMPANY asks for one): with a copy stream and a copy-back
        // among the batch's streams the hardware queues are shared, and a grid that lives as long as the clock stage holds back
        // whatever lands on its queue -- a sparse 512-slot batcher ran 4.3 instead of 3.0 ms per round with it (3 live clients;
        // 5.5 instead of 3.5 with BASELINE configs[4]'s mix), a full one is bound by the host link either way
        // (profiles/r05_node_schedule.txt).  The calibration at creation timed device-resident calls: it cannot see this.
        if (getenv("SDRM_K3_COMPANY") == nullptr) {
            b->company_blocks = 0;
            b->company_grid = 0;
        }
        int code = 0;
        for (int i = 0; i < 2 && code == 0; i++) {
            code = dev_alloc_zero(&b->d_in_ring[i], slot_samples);
        }
        code = code ? code : dev_alloc_zero(&b->d_out8_b, C * (size_t) b->dev.out_stride);
        code = code ? code : dev_alloc_zero(&b->d_outlen_b, C);
        if (code != 0) {
            return code;
        }
        for (int i = 0; i < SDRM_RES_SETS; i++) {
            if (hipHostMalloc((void **) &b->h_res8[i], C * (size_t) b->dev.out_stride) != hipSuccess ||
                hipHostMalloc((void **) &b->h_reslen[i], sizeof(uint32_t) * C) != hipSuccess) {
                return -ENOMEM;
            }
            HIP_TRY(hipEventCreateWithFlags(&b->ev_res[i], hipEventDisableTiming));
        }
        for (int i = 0; i < 2; i++) {
            HIP_TRY(hipEventCreateWithFlags(&b->ev_out_free[i], hipEventDisableTiming));
        }
        if (hipHostMalloc((void **) &b->h_arena, slots * slot_samples * sizeof(sdrm_f2)) != hipSuccess) {
            fprintf(stderr, "<3>sdrmodem_hip: cannot pin %zu bytes of host memory for the input arena\n",
                    slots * slot_samples * sizeof(sdrm_f2));
            return -ENOMEM;
        }
        HIP_TRY(hipStreamCreateWithFlags(&b->s_h2d, hipStreamNonBlocking));
        b->arena_slots = slots;
    } else if (slots != b->arena_slots) {
        return -1;
    }
    *base = reinterpret_cast<sdrm_cf32 *>(b->h_arena);
    if (chan_stride) {
        *chan_stride = b->in_stride;
    }
    if (slot_stride) {
        *slot_stride = slot_samples;
    }
    return 0;
}

extern "C" int sdrm_batch_submit(sdrm_batch *b, size_t slot, const size_t *input_lens, const sdrm_nco_segment *segments,
                                 size_t n_segments) {
    if (b == nullptr || b->h_arena == nullptr || slot >= b->arena_slots || input_lens == nullptr) {
        return -1;
    }
    if (b->submitted - b->collected >= SDRM_MAX_FLIGHT) {
        return -EAGAIN;  // collect the oldest call first
    }
    HIP_TRY(hipSetDevice(b->device));
    const size_t C = b->plan.design.size();
    const uint64_t k = b->calls;
    const int par = (int) (k & 1);
    // the device input buffer of this parity was last read by the front-end of call k-2
    if (k >= 2 && b->slot_used[(k - 2) % SDRM_CTL_SLOTS]) {
        HIP_TRY(hipStreamWaitEvent(b->s_h2d, b->ev_front[(k - 2) % SDRM_CTL_SLOTS], 0));
    }
    // Only the rows of channels that take part travel: a server's batcher is sized for its busiest hour, and a call of 3 live
    // clients on 512 slots used to copy all 512 rows (537 MB, 10 ms per round -- profiles/r05_node_schedule.txt).  Runs of rows
    // (gaps of up to 7 absent channels stay inside a run: one copy costs ~5 us to issue), at most 16 copies per call.
    const sdrm_f2 *src = b->h_arena + slot * C * (size_t) b->in_stride;
    auto takes_part = [&](size_t c) { return input_lens[c] != SDRM_LEN_ABSENT && input_lens[c] > 0 && input_lens[c] <= b->plan.params[c].max_len; };
    size_t first_present = C, last_present = 0, runs = 0;
    for (size_t c = 0; c < C;) {
        if (!takes_part(c)) {
            c++;
            continue;
        }
        size_t end = c + 1, last = c, longest = input_lens[c];
        while (end < C && (end - last <= 8 || runs >= 15)) {
            if (takes_part(end)) {
                last = end;
                longest = std::max(longest, input_lens[end]);
            }
            end++;
        }
        const size_t rows = last - c + 1;
        const size_t at = c * (size_t) b->in_stride;
        if (longest == b->in_stride) {
            HIP_TRY(hipMemcpyAsync(b->d_in_ring[par] + at, src + at, rows * (size_t) b->in_stride * sizeof(sdrm_f2), hipMemcpyHostToDevice, b->s_h2d));
        } else {
            HIP_TRY(hipMemcpy2DAsync(b->d_in_ring[par] + at, (size_t) b->in_stride * sizeof(sdrm_f2), src + at,
                                     (size_t) b->in_stride * sizeof(sdrm_f2), longest * sizeof(sdrm_f2), rows, hipMemcpyHostToDevice, b->s_h2d));
        }
        first_present = std::min(first_present, c);
        last_present = std::max(last_present, last);
        runs++;
        c = last + 1;
    }
    int code = enqueue_call(b, b->d_in_ring[par], b->in_stride, input_lens, b->s_h2d, segments, n_segments);
    if (code != 0) {
        return code;
    }
    // copy-back: counts, then the soft bits up to the most symbols a channel can have produced
    // (every symbol consumes at least floor(omega_min) - 1 samples once in lock; the hard cap is max_len)
    uint32_t width = 0;
    for (size_t c = 0; c < C; c++) {
        const sdrm_chan_params &p = b->plan.params[c];
        const double step = std::max(1.0, (double) p.omega_mid - (double) p.omega_lim - 1.0);
        const double n_in = input_lens[c] == SDRM_LEN_ABSENT ? 0.0 : (double) input_lens[c];
        const double bound = p.can_wild ? (double) p.max_len : (n_in / (double) p.decim + carried_cap(p)) / step + 16.0;
        width = std::max<uint32_t>(width, (uint32_t) std::min<double>(bound, (double) p.max_len));
    }
    width = std::min<uint32_t>((width + 63u) & ~63u, b->dev.out_stride);
    b->back_width[k % SDRM_RES_SETS] = width;
    b->back_slot[k % SDRM_RES_SETS] = b->last_slot;
    // soft bits come back for the span of channels that may have produced some: those that took part, and -- an empty call is
    // answered from the carried samples (at >= 8 samples per symbol) -- those with a length of 0
    size_t lo = first_present, hi = last_present;
    for (size_t c = 0; c < C; c++) {
        if (input_lens[c] == 0) {
            lo = std::min(lo, c);
            hi = std::max(hi, c);
        }
    }
    b->back_first[k % SDRM_RES_SETS] = (uint32_t) (lo < C ? lo : 0);
    b->back_rows[k % SDRM_RES_SETS] = (uint32_t) (lo < C ? hi - lo + 1 : 0);
    // the previous call's copy-back goes in now, BEHIND this call's DC stage (see issue_copy_back)
    if (b->back_pending) {
        code = issue_copy_back(b, k - 1);
        if (code != 0) {
            return code;
        }
    }
    b->back_pending = true;
    if (b->submitted == b->collected) {
        b->first_pipelined_call = k;
    }
    b->submitted++;
    return 0;
}

extern "C" int sdrm_batch_collect(sdrm_batch *b, int8_t **outputs, size_t *output_lens) {
    if (b == nullptr || outputs == nullptr || output_lens == nullptr || b->submitted == b->collected) {
        return -1;
    }
    HIP_TRY(hipSetDevice(b->device));
    const size_t C = b->plan.design.size();
    // calls made through this path are consecutive while any is uncollected: the oldest one is
    const uint64_t k = b->calls - (b->submitted - b->collected);
    const int set = (int) (k % SDRM_RES_SETS);
    if (b->back_pending && k + 1 == b->calls) {
        int code = issue_copy_back(b, k);  // nothing was submitted after it
        if (code != 0) {
            return code;
        }
    }
    HIP_TRY(hipEventSynchronize(b->ev_res[set]));
    if (int code = check_device_error(b)) {
        return code;
    }
    for (size_t c = 0; c < C; c++) {
        if (b->h_reslen[set][c] == SDRM_OUT_LEN_FAILED) {
            // this call's clock stage says that a stage gave up waiting inside the call (the in-call hand-off's bounded looks):
            // no count of the call is a count.  Sticky, like every device failure.
            const int failed = check_device_error(b);
            if (failed == 0) {
                b->device_error = -ETIMEDOUT;
                fprintf(stderr, "<3>sdrmodem_hip: a call came back void (a stage gave up waiting inside it); the batch is unusable\n");
            }
            return b->device_error;
        }
    }
    for (size_t c = 0; c < C; c++) {
        const uint32_t n = b->h_reslen[set][c];
        int8_t *dst = b->h_res8[set] + c * (size_t) b->dev.out_stride;
        if (n > b->res_width[set]) {
            // More symbols than the copy-back bound (a loop far out of lock).  The device set may already belong to
            // call k+2 by now, so the tail is dropped with a message rather than read from the wrong call.
            fprintf(stderr, "<3>sdrmodem_hip: channel %zu produced %u symbols, %u copied back\n", c, n, b->res_width[set]);
            b->h_reslen[set][c] = b->res_width[set];
        }
        const uint32_t n_ok = b->h_reslen[set][c];
        outputs[c] = dst;
        output_lens[c] = n_ok;
        b->last_lens[c] = n_ok;
    }
    b->collected++;
    return 0;
}

extern "C" int sdrm_batch_fetch(sdrm_batch *b, int8_t *data, size_t stride, size_t *lens) {
    if (b == nullptr || lens == nullptr) {
        return -1;
    }
    HIP_TRY(hipSetDevice(b->device));
    const size_t C = b->plan.design.size();
    if (int code = quiesce(b)) {
        return code;
    }
    if (int code = wait_for_all_calls(b)) {
        return code;  // sticky device error (a bounded in-kernel wait expired)
    }
    const uint64_t last = b->calls ? b->calls - 1 : 0;
    HIP_TRY(hipMemcpy(b->h_outlen, outlen_of(b, last), sizeof(uint32_t) * C, hipMemcpyDeviceToHost));
    for (size_t c = 0; c < C; c++) {
        uint32_t n = b->h_outlen[c];
        b->last_lens[c] = n;
        lens[c] = n;
        if (data != nullptr && n > 0) {
            HIP_TRY(hipMemcpy(data + c * stride, out8_of(b, last) + c * (size_t) b->dev.out_stride, std::min<size_t>(n, stride),
                              hipMemcpyDeviceToHost));
        }
    }
    return 0;
}

extern "C" int sdrm_batch_last_soft(sdrm_batch *b, size_t c, float *dst, size_t cap, size_t *len) {
    if (b == nullptr || c >= b->plan.design.size() || b->d_outf == nullptr) {
        return -1;
    }
    HIP_TRY(hipSetDevice(b->device));
    if (int code = quiesce(b)) {
        return code;
    }
    if (int code = wait_for_all_calls(b)) {
        return code;  // sticky device error (a bounded in-kernel wait expired)
    }
    uint32_t n = 0;
    HIP_TRY(hipMemcpy(&n, outlen_of(b, b->calls ? b->calls - 1 : 0) + c, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (len) {
        *len = n;
    }
    if (dst != nullptr && n > 0) {
        HIP_TRY(hipMemcpy(dst, b->d_outf + c * (size_t) b->dev.out_stride, sizeof(float) * std::min<size_t>(n, cap),
                          hipMemcpyDeviceToHost));
    }
    return 0;
}

// ================================================================================================
// Reference operator API (src/dsp/fsk_demod.h:11-15): a batch of one channel.

// ---- the reference operator.  By default a handle is a private batch of one channel.  With SDRM_SHARED_SLOTS=n in the
// environment the handles of the process share ONE batcher of n slots instead (created by the first handle, whose
// configuration fixes the batch's geometry): fsk_demod_process then puts its buffer on the handle's slot and blocks
// until the round it went into has come back, so the per-client DSP threads of an unmodified sdr-modem
// (src/dsp_worker.c:44-106, one fsk_demod_process per buffer each) are served by one device call per round.
// SDRM_SHARED_WAIT_US (default 1000) is how long a round waits for more handles to join.
struct fsk_demod_t {
    sdrm_batch_t *batch;
    sdrm_batcher *shared;
    size_t slot;
    uint32_t max_len;
    int8_t *out;  // the handle's own copy of its last result (valid until its next call, as in the reference)
    int error;    // sticky: the device path failed under this handle (every later call produces nothing)
};

static thread_local int g_last_error = 0;
extern "C" int sdrm_last_error(void) { return g_last_error; }
extern "C" int sdrm_fsk_demod_error(const fsk_demod *demod) { return demod ? demod->error : -1; }

namespace {
struct SharedPool {
    std::mutex m;
    sdrm_batcher *bt = nullptr;
    std::vector<uint8_t> used;
    bool failed = false;
};
SharedPool g_pool;

bool shared_attach(fsk_demod_t *d, const sdrm_fsk_config &cfg) {
    const char *env = getenv("SDRM_SHARED_SLOTS");
    const long n = env ? atol(env) : 0;
    if (n <= 0) {
        return false;
    }
    std::lock_guard<std::mutex> g(g_pool.m);
    if (g_pool.bt == nullptr && !g_pool.failed) {
        std::vector<sdrm_fsk_config> cfgs((size_t) n, cfg);
        const char *w = getenv("SDRM_SHARED_WAIT_US");
        sdrm_batcher_config bc = {4, (uint32_t) (w ? atol(w) : 1000), true};
        if (sdrm_batcher_create(cfgs.data(), cfgs.size(), -1, &bc, &g_pool.bt) != 0) {
            g_pool.bt = nullptr;
            g_pool.failed = true;
        } else {
            g_pool.used.assign((size_t) n, 0);
            // a round waits for every OPEN channel: slots without a handle stay closed (the reset that attaches a handle
            // reopens its slot), so that a round is launched as soon as the live handles have delivered
            for (size_t s = 0; s < (size_t) n; s++) {
                sdrm_batcher_abandon(g_pool.bt, s);
            }
        }
    }
    if (g_pool.bt == nullptr) {
        return false;
    }
    for (size_t s = 0; s < g_pool.used.size(); s++) {
        if (!g_pool.used[s]) {
            // the slot gets this handle's parameters and a clean state; what does not fit the shared batch's geometry
            // (longer filters, larger buffers than the first handle's) gets a private batch instead
            if (sdrm_batcher_reset_channel(g_pool.bt, s, &cfg) != 0) {
                return false;
            }
            g_pool.used[s] = 1;
            d->shared = g_pool.bt;
            d->slot = s;
            return true;
        }
    }
    return false;
}
}  // namespace

extern "C" int fsk_demod_create(uint64_t sampling_freq, uint32_t baud_rate, int64_t deviation, uint8_t decimation,
                                uint32_t transition_width, bool use_dc_block, uint32_t max_input_buffer_length,
                                fsk_demod **demod) {
    fsk_demod_t *d = (fsk_demod_t *) calloc(1, sizeof(fsk_demod_t));
    if (d == nullptr) {
        return -ENOMEM;
    }
    sdrm_fsk_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.sampling_freq = sampling_freq;
    cfg.baud_rate = baud_rate;
    cfg.deviation = deviation;
    cfg.decimation = decimation;
    cfg.transition_width = transition_width;
    cfg.use_dc_block = use_dc_block;
    cfg.max_input_buffer_length = max_input_buffer_length;
    d->max_len = max_input_buffer_length;
    if (shared_attach(d, cfg)) {
        d->out = (int8_t *) malloc(max_input_buffer_length ? max_input_buffer_length : 1);
        if (d->out == nullptr) {
            fsk_demod_destroy(d);
            return -ENOMEM;
        }
        *demod = d;
        return 0;
    }
    int code = sdrm_batch_create(&cfg, 1, -1, 0, &d->batch);
    if (code != 0) {
        free(d);
        return code;
    }
    *demod = d;
    return 0;
}

// The reference's process() returns void and cannot fail.  A device failure here must neither look like "no symbols
// this time" for ever after nor take the whole server down with every other client attached: the handle goes into a
// sticky error state ("<3>" message once, *output_len = 0 from then on, sdrm_fsk_demod_error() / sdrm_last_error() tell),
// and the worker above ends THAT client, as the reference does on a socket or disk error (src/dsp_worker.c:56-64, 83-101).
static void demod_failed(fsk_demod *demod, int code, const char *what) {
    if (demod->error == 0) {
        fprintf(stderr, "<3>sdrmodem_hip: fsk_demod_process: %s (code %d); this handle produces nothing from now on\n", what, code);
    }
    demod->error = code ? code : -EIO;
    g_last_error = demod->error;
}

extern "C" void fsk_demod_process(const sdrm_cf32 *input, size_t input_len, int8_t **output, size_t *output_len,
                                  fsk_demod *demod) {
    if (demod->error != 0) {
        *output = demod->out;
        *output_len = 0;
        g_last_error = demod->error;
        return;
    }
    if (demod->shared != nullptr) {
        *output = demod->out;
        *output_len = 0;
        if (input_len > demod->max_len) {
            fprintf(stderr, "<3>requested buffer %zu is more than max: %zu\n", input_len, (size_t) demod->max_len);
            return;
        }
        int8_t *soft = nullptr;
        size_t n = 0;
        sdrm_batcher_put(demod->shared, demod->slot, input, input_len);
        sdrm_batcher_take(demod->shared, demod->slot, &soft, &n);
        if (soft == nullptr) {
            const int dev = sdrm_batcher_error(demod->shared);  // a failed device call ends every handle on the batcher
            demod_failed(demod, dev != 0 ? dev : -EPIPE, dev != 0 ? "the shared batcher's device call failed" : "the shared batcher went away");
            return;
        }
        memcpy(demod->out, soft, n);
        sdrm_batcher_complete(demod->shared, demod->slot);
        *output_len = n;
        return;
    }
    const sdrm_cf32 *ins[1] = {input};
    size_t lens[1] = {input_len};
    int8_t *outs[1] = {nullptr};
    size_t olens[1] = {0};
    int code = sdrm_batch_process(demod->batch, ins, lens, outs, olens);
    if (code != 0) {
        demod_failed(demod, code, "the device call failed");
        *output = nullptr;
        *output_len = 0;
        return;
    }
    *output = outs[0];
    *output_len = olens[0];
}

extern "C" void fsk_demod_destroy(fsk_demod *demod) {
    if (demod == nullptr) {
        return;
    }
    if (demod->shared != nullptr) {
        std::lock_guard<std::mutex> g(g_pool.m);
        g_pool.used[demod->slot] = 0;  // the next handle that takes the slot resets it
        sdrm_batcher_abandon(demod->shared, demod->slot);  // rounds stop waiting for this slot; nothing of it is kept
    }
    if (demod->batch != nullptr) {
        batch_free(demod->batch);
    }
    free(demod->out);
    free(demod);
}

// ================================================================================================ probes

// every probe: one exit path (all device buffers freed whatever failed), and a failed launch is an error, not a vector of
// uninitialised results
static int probe_finish(hipError_t e, const char *what, std::initializer_list<void *> buffers) {
    for (void *p : buffers) {
        (void) hipFree(p);
    }
    if (e != hipSuccess) {
        fprintf(stderr, "<3>sdrmodem_hip: %s failed: %s\n", what, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? -ENOMEM : -EIO;
    }
    return 0;
}

extern "C" int sdrm_probe_atan2(const float *y, const float *x, float *out, size_t n) {
    if (sdrm_device_count() <= 0) {
        fprintf(stderr, "<3>sdrmodem_hip: no HIP device available\n");
        return -ENODEV;
    }
    float *dy = nullptr, *dx = nullptr, *dt = nullptr, *dout = nullptr;
    hipError_t e = hipSuccess;
    e = e ? e : hipMalloc((void **) &dy, n * 4 + 4);
    e = e ? e : hipMalloc((void **) &dx, n * 4 + 4);
    e = e ? e : hipMalloc((void **) &dout, n * 4 + 4);
    e = e ? e : hipMalloc((void **) &dt, 260 * 4);
    e = e ? e : hipMemcpy(dy, y, n * 4, hipMemcpyHostToDevice);
    e = e ? e : hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice);
    e = e ? e : hipMemcpy(dt, sdrm_atan_tab, 257 * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        sdrm::launch_probe_atan2(dy, dx, dt, dout, n, nullptr);
        e = hipGetLastError();
    }
    e = e ? e : hipDeviceSynchronize();
    e = e ? e : hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost);
    return probe_finish(e, "sdrm_probe_atan2", {dy, dx, dt, dout});
}

// the front-end's discriminator phase as the kernel runs it (short form with its per-wave fall-back): out[i] = gain *
// fast_atan2f(y[i] conj(y[i-1])) for a stream of n complex samples y (y[-1] = 0); fast_waves (may be NULL) receives one
// word per 960 samples: did that wave take the short form
extern "C" int sdrm_probe_quad(const float *iq, size_t n, float gain, float *out, uint32_t *fast_waves) {
    if (sdrm_device_count() <= 0) {
        fprintf(stderr, "<3>sdrmodem_hip: no HIP device available\n");
        return -ENODEV;
    }
    const size_t waves = (n + 64 * SDRM_K1_R - 1) / (64 * SDRM_K1_R) + 4;
    sdrm_f2 *dy = nullptr;
    float *dt = nullptr, *dout = nullptr;
    uint32_t *df = nullptr;
    hipError_t e = hipSuccess;
    e = e ? e : hipMalloc((void **) &dy, n * 8 + 8);
    e = e ? e : hipMalloc((void **) &dout, n * 4 + 4);
    e = e ? e : hipMalloc((void **) &dt, 260 * 4);
    e = e ? e : hipMalloc((void **) &df, waves * 4);
    e = e ? e : hipMemset(df, 0, waves * 4);
    e = e ? e : hipMemcpy(dy, iq, n * 8, hipMemcpyHostToDevice);
    e = e ? e : hipMemcpy(dt, sdrm_atan_tab, 257 * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        sdrm::launch_probe_quad(dy, n, gain, dt, dout, df, nullptr);
        e = hipGetLastError();
    }
    e = e ? e : hipDeviceSynchronize();
    e = e ? e : hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost);
    if (fast_waves != nullptr) {
        e = e ? e : hipMemcpy(fast_waves, df, ((n + 64 * SDRM_K1_R - 1) / (64 * SDRM_K1_R)) * 4, hipMemcpyDeviceToHost);
    }
    return probe_finish(e, "sdrm_probe_quad", {dy, dt, dout, df});
}

// quotients of the DC blocker's boxcars: the three-instruction form with its fall-back, as the DC kernel runs it
extern "C" int sdrm_probe_boxcar_div(const float *sums, uint32_t length, float *out, size_t n) {
    if (sdrm_device_count() <= 0) {
        fprintf(stderr, "<3>sdrmodem_hip: no HIP device available\n");
        return -ENODEV;
    }
    float *dt = nullptr, *dout = nullptr;
    hipError_t e = hipSuccess;
    e = e ? e : hipMalloc((void **) &dt, n * 4 + 4);
    e = e ? e : hipMalloc((void **) &dout, n * 4 + 4);
    e = e ? e : hipMemcpy(dt, sums, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        sdrm::launch_probe_boxcar_div(dt, length, dout, n, nullptr);
        e = hipGetLastError();
    }
    e = e ? e : hipDeviceSynchronize();
    e = e ? e : hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost);
    return probe_finish(e, "sdrm_probe_boxcar_div", {dt, dout});
}

// diagnostics: allocate (once) and return the device buffer K3 writes its per-wave cycle stamps into; enable != 0
// turns stamping on for subsequent calls.  Copies the stamps of the last call to `out` (4 x uint64 per wave).
extern "C" int sdrm_batch_k3_stamps(sdrm_batch *b, int enable, unsigned long long *out, size_t max_waves) {
    if (b == nullptr) {
        return -1;
    }
    HIP_TRY(hipSetDevice(b->device));
    const size_t waves = SDRM_STAMP_K3_WAVES(b->plan.params.size());
    // enable == 1: every call from now on; enable = k > 1: only the k-th call from now (a call in the middle of a
    // pipelined run can then be looked at)
    if (enable > 1) {
        b->stamp_only_call = b->calls + (uint64_t) enable;
    } else if (enable == 1) {
        b->stamp_only_call = 0;
    }
    if (b->dev.k3_stamps == nullptr && enable) {
        HIP_TRY(hipMalloc((void **) &b->dev.k3_stamps, (waves * 4 + 24) * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(b->dev.k3_stamps, 0, (waves * 4 + 24) * sizeof(unsigned long long)));
    }
    if (out != nullptr && b->dev.k3_stamps != nullptr) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(out, b->dev.k3_stamps, (std::min(waves, max_waves) * 4 + 24) * sizeof(unsigned long long),
                          hipMemcpyDeviceToHost));
    }
    return (int) waves;
}

// diagnostics: when and for how long each kernel of the next (up to 64) calls really runs on the device, whatever
// the streams and the dispatcher make of the dependencies.  enable != 0 attaches a fresh table; `out` (may be NULL)
// receives rows of {front start, front end, dc start, dc end, clock start, clock end} in 10 ns ticks of the device's
// reference clock for the calls made since the table was attached.  Returns the number of rows written.
extern "C" int sdrm_batch_timeline(sdrm_batch *b, int enable, unsigned long long *out, size_t max_rows) {
    if (b == nullptr) {
        return -1;
    }
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipDeviceSynchronize());
    int rows = 0;
    if (out != nullptr && b->d_timeline != nullptr) {
        rows = (int) std::min<uint64_t>(std::min<uint64_t>(b->calls - b->timeline_first_call, 64), max_rows);
        HIP_TRY(hipMemcpy(out, b->d_timeline, (size_t) rows * 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
    if (enable) {
        if (b->d_timeline == nullptr) {
            HIP_TRY(hipMalloc((void **) &b->d_timeline, 64 * 6 * sizeof(unsigned long long)));
        }
        std::vector<unsigned long long> init(64 * 6);
        for (size_t k = 0; k < init.size(); k++) {
            init[k] = (k & 1) ? 0ull : ~0ull;  // starts take the minimum, ends the maximum
        }
        HIP_TRY(hipMemcpy(b->d_timeline, init.data(), init.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
        b->timeline_first_call = b->calls;
    }
    return rows;
}

