// sdrm_call.hip -- the process path: one call of the batch enqueued on its streams (enqueue_call: NCO, front-end, DC blocker,
// clock recovery, the in-call hand-off and its device-wide admission), the waits, and the entry points built on it --
// device-resident calls, blocking host-buffer calls (a plain handle's replayed graph among them), the pipelined host path
// (arena / submit / collect), fetch.  Batch life cycle: sdrm_batch.hip; schedule and tuners: sdrm_tune.hip.
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
#include "../host/ledger.h"

using namespace sdrm_impl;

#define SDRM_HAND_SERIAL_MIN_NZ 12288u  // LPF2 outputs from which a plain handle's blocking call takes the in-call hand-off

// ---- Device-wide admission of in-call hand-offs: the ledger itself is host logic of its own (../host/ledger.h, also built with the
// sanitizers); here its per-device instances and what ties an entry to a batch's HIP event.
namespace {
sdrm::WaitLedger g_hand_ledger[16];
sdrm::WaitLedger &hand_ledger(int device) { return g_hand_ledger[device & 15]; }
bool hand_event_fired(void *event) {
    const bool fired = hipEventQuery((hipEvent_t) event) == hipSuccess;
    (void) hipGetLastError();  // hipErrorNotReady is not an error
    return fired;
}
}  // namespace

void sdrm_impl::hand_release(sdrm_batch_t *b) {
    if (b->hand_listed) {
        hand_ledger(b->device).release(b);
        b->hand_listed = false;
    }
}

// true: `waiting` workgroups of b's next call may wait on the device; false: the device's budget is taken
static bool hand_admit(sdrm_batch_t *b, unsigned waiting, unsigned limit, hipEvent_t done) {
    const bool ok = hand_ledger(b->device).admit(b, (void *) done, waiting, limit, b->serial, hand_event_fired);
    b->hand_listed = ok;
    return ok;
}

static void hand_arm(sdrm_batch_t *b) { hand_ledger(b->device).arm(b); }

// a call that would qualify for the hand-off by its length, on a device with too many plain handles' calls in flight: refused at once
// (what hundreds of client threads calling together meet on every call: no event query, no kernel geometry, one atomic load)
static bool hand_crowded(sdrm_batch_t *b, const sdrm_chunk_ctl *h) {
    if (b->serial) {
        uint32_t longest = 0;
        for (size_t c = 0; c < b->plan.design.size(); c++) {
            longest = std::max(longest, h[c].nz);
        }
        if (longest < SDRM_HAND_SERIAL_MIN_NZ) {
            return false;  // does not qualify anyway (the block below says so too): not a refusal
        }
    }
    if (!hand_ledger(b->device).crowded(b->serial)) {
        return false;
    }
    b->hand_refused++;
    return true;
}

// a plain handle's blocking call, from its first enqueue to its results (process_host)
struct PlainCallInFlight {
    sdrm::WaitLedger *ledger = nullptr;
    explicit PlainCallInFlight(const sdrm_batch_t *b) {
        if (b->serial) {
            ledger = &hand_ledger(b->device);
            ledger->plain_begin();
        }
    }
    ~PlainCallInFlight() {
        if (ledger != nullptr) {
            ledger->plain_end();
        }
    }
    PlainCallInFlight(const PlainCallInFlight &) = delete;
    PlainCallInFlight &operator=(const PlainCallInFlight &) = delete;
};

extern "C" int sdrm_handoff_stats(int device, uint64_t *taken, uint64_t *refused, uint32_t *peak_waiting) {
    if (device < 0 && hipGetDevice(&device) != hipSuccess) {
        return -ENODEV;
    }
    hand_ledger(device).stats(taken, refused, peak_waiting);
    return 0;
}

int sdrm_impl::enqueue_call(sdrm_batch_t *b, const sdrm_f2 *d_in, size_t in_stride, const size_t *lens, hipStream_t caller,
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
    // Waiting workgroups hold their CUs, so this is bounded twice.  Per call: at most 192 of them (every batch of the 16 x 1024
    // clock-stage shape -- 160 at 1280 channels, one per CU -- and those of the 32 x 512 shape up to 2048 channels: the front-end
    // keeps the other CUs and the room beside the DC workgroups) when a DC workgroup leaves room for a front-end workgroup
    // beside it, 16 when it does not.  Per device: the ledger (../host/ledger.h) adds up what every batch and handle of the
    // process has waiting, and counts plain handles' calls in flight.  Order: the clock stage is launched only when the DC
    // stage's workgroups are resident, the front-end only when both are -- then the front-end, which waits for nobody, always
    // finds a CU, the DC stage waits only for the front-end and the clock stage only for the DC stage.  Every wait in the
    // kernels is bounded besides (a void call, loudly, never a hung device).
    bool hand = false;
    if (b->hand_allowed && b->n_gen == 0 && (b->serial || b->d_placed != nullptr) && max_tiles > 0 && !hand_crowded(b, h)) {
        const bool idle = b->last_slot < 0 || hipEventQuery(b->slot_done[b->last_slot]) == hipSuccess;
        const unsigned waiting = sdrm::clock_workgroups(d) + (d.any_dc ? sdrm::dc_workgroups(d) : 0u);
        const bool room = !d.any_dc || (size_t) d.dc_lds + sdrm::k1_lds_bytes(d.t1_max, d.t2_max) <= 160 * 1024;
        // (a DC workgroup that fills its CU -- long boxcars -- leaves the front-end no room beside it: then only a handful may wait)
        const unsigned most = room ? 192u : 16u;  // (192 against 64: 1024 channels 5.71 -> 3.07 ms per blocking call, profiles/r05_incall_handoff.txt)
        // (32-bit byte offsets inside a DC workgroup's rows of z / dcout: sdrm_kernels.hip, hand_rsrc)
        const bool offsets_fit = (uint64_t) (d.dc_group ? d.dc_group : 1u) * d.z_stride * sizeof(float) < (1ull << 32);
        hand = idle && waiting <= most && offsets_fit && sdrm::clock_shape_hands_off(d);
        if (idle) {
            hand_release(b);  // the previous hand-off call of this batch, if any, is over
        }
        if (hand && b->serial) {
            // a plain handle keeps its stages on one stream (a server holds one per client); a call long enough for the overlap
            // to pay (SDRM_HAND_SERIAL_MIN_NZ) gets two side streams, created when the first such call comes
            uint32_t longest = 0;
            for (size_t c = 0; c < C; c++) {
                longest = std::max(longest, h[c].nz);
            }
            hand = longest >= SDRM_HAND_SERIAL_MIN_NZ;
        }
        // (the ledger's entry names an event of its own, recorded behind this call only: the slot's event is recorded again eight
        // calls later, and a pipelined run that began with a hand-off call would look unfinished for as long as it lasts)
        if (hand && b->ev_hand_done == nullptr && hipEventCreateWithFlags(&b->ev_hand_done, hipEventDisableTiming) != hipSuccess) {
            (void) hipGetLastError();
            b->ev_hand_done = nullptr;
            hand = false;
        }
        const bool admitted = hand && hand_admit(b, waiting, most, b->ev_hand_done);
        if (hand && !admitted) {
            hand = false;  // the DEVICE's budget is taken by other batches and handles: stages in stream order, always safe
            b->hand_refused++;
        }
        if (hand && b->serial && b->s_hand_clock == nullptr) {
            // Both side streams at the MIDDLE priority level, the one the handle's copy stream lives on anyway.  HIP keeps a pool of
            // hardware queues per level in use: with the clock stage's stream at the highest level (round 5) the first hand-off call
            // of a process that otherwise runs plain handles only brought a third pool to life, and every later call of every
            // handle paid for it -- 256 handles x 131072 samples: 8.5-9.5 s per 20 rounds after a single admitted call, 7.4-7.9
            // without (profiles/r06_handles.txt); a lone handle has nothing to take priority over.
            int prio_low = 0, prio_high = 0;
            (void) hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
            const int prio_mid = (prio_low + prio_high) / 2;
            if (hipStreamCreateWithPriority(&b->s_hand_dc, hipStreamNonBlocking, prio_mid) != hipSuccess ||
                hipStreamCreateWithPriority(&b->s_hand_clock, hipStreamNonBlocking, prio_mid) != hipSuccess) {
                (void) hipGetLastError();
                hand = false;
            }
        }
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
        if (admitted && !hand) {
            hand_release(b);  // (admitted, then no streams or no memory: the place in the ledger goes back)
        }
    }
    // the streams of this call's DC and clock stages
    const bool hand_side = hand && b->serial;
    hipStream_t s_dc = hand_side ? b->s_hand_dc : b->s_dc;
    if (b->hand_side_last && b->last_slot >= 0) {
        HIP_TRY(hipStreamWaitEvent(b->s_front, b->slot_done[b->last_slot], 0));  // the previous call ended on a side stream
    }
    b->hand_side_last = hand_side;
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
    if (!b->serial && !hand && d.any_dc && b->d_placed != nullptr && b->hand_calls > 0 && i - b->last_hand_call <= 2) {
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
    if (hand) {
        HIP_TRY(hipEventRecord(b->ev_hand_done, s_clock));
        hand_arm(b);  // from here on anybody's admission may find the call over and take its place
    }
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
            hand_release(b);  // its waiting workgroups are gone: the place in the device's ledger is free
        }
    }
    return b->device_error;
}

// the host waits until every enqueued call has finished
int sdrm_impl::wait_for_all_calls(sdrm_batch_t *b) {
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
    const PlainCallInFlight in_flight(b);
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
        hand_release(b);
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

