// sdrm_handle.hip -- the reference operator fsk_demod_create / _process / _destroy (src/dsp/fsk_demod.h:11-15) on a batch of one
// channel (or a slot of a shared batcher), the stage probes the parity tests use, and the diagnostics (cycle stamps, device timeline).
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
// what sdrm_fsk_demod_share asked for (-1: nothing, the environment decides)
std::atomic<long> g_share_slots{-1};
std::atomic<long> g_share_wait_us{-1};

bool shared_attach(fsk_demod_t *d, const sdrm_fsk_config &cfg) {
    long n = g_share_slots.load();
    if (n < 0) {
        const char *env = getenv("SDRM_SHARED_SLOTS");
        n = env ? atol(env) : 0;
    }
    if (n <= 0) {
        return false;
    }
    std::lock_guard<std::mutex> g(g_pool.m);
    if (g_pool.bt == nullptr && !g_pool.failed) {
        std::vector<sdrm_fsk_config> cfgs((size_t) n, cfg);
        long wait_us = g_share_wait_us.load();
        if (wait_us < 0) {
            const char *w = getenv("SDRM_SHARED_WAIT_US");
            wait_us = w ? atol(w) : 1000;
        }
        sdrm_batcher_config bc = {4, (uint32_t) wait_us, true};
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

// the programmatic form of SDRM_SHARED_SLOTS / SDRM_SHARED_WAIT_US: handles created from now on share one batcher of `slots`
// slots (0: private batches for handles created from now on).  The pool itself is made by the first handle that joins it.
extern "C" int sdrm_fsk_demod_share(size_t slots, uint32_t max_wait_us) {
    if (slots > 65536) {
        return -1;
    }
    std::lock_guard<std::mutex> g(g_pool.m);
    if (g_pool.bt != nullptr && slots != 0 && slots != g_pool.used.size()) {
        return -EBUSY;  // the pool exists with another size: handles are attached to it
    }
    g_share_wait_us.store((long) max_wait_us);
    g_share_slots.store((long) slots);
    return 0;
}

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

