// sdrm_batch_impl.h -- what the translation units of the library share: the batch's state and the internal functions that cross
// file boundaries.  Internal: not installed, not part of the C-ABI.
#ifndef SDRM_BATCH_IMPL_H
#define SDRM_BATCH_IMPL_H

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <vector>

#include "../../include/sdrmodem_hip.h"
#include "sdrm_design.h"
#include "sdrm_plan.h"
#include "sdrm_launch.h"

#define SDRM_CTL_SLOTS 8
#define SDRM_RES_SETS 4    // pinned result sets of the pipelined host path
// row pitch of the NCO phase buffers: every channel's generator writes the same column at the same time, and a pitch that
// is a power of two would put all of those writes on one memory channel; 4 KiB + 256 B more per row spreads them
#define SDRM_PHASE_STRIDE(in_stride) ((in_stride) + 1088u)
#define SDRM_MAX_FLIGHT 3  // uncollected calls it allows (copy-in, kernels and copy-back of different calls overlap)

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            fprintf(stderr, "<3>sdrmodem_hip: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_),   \
                    __FILE__, __LINE__);                                                                \
            return -EIO;                                                                                \
        }                                                                                               \
    } while (0)

struct TimingLane {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> free_list;
    double total_ms = 0.0;
    uint64_t launches = 0;
    // a stage whose consecutive launches overlap (the clock stage resident early): a launch's time counts from the end of
    // the launch before it, if that came later than its own start
    bool overlapped = false;
    bool has_prev = false;
    std::pair<hipEvent_t, hipEvent_t> prev;
};

struct sdrm_batch_t {
    int device = 0;
    uint32_t flags = 0;
    sdrm::BatchPlan plan;  // designs, packed parameters, host-side streaming bookkeeping
    // device memory
    sdrm_chan_params *d_params = nullptr;
    sdrm_chunk_ctl *d_ctl = nullptr;  // [SLOTS][C]
    float *d_taps = nullptr, *d_atan = nullptr, *d_bank = nullptr;
    sdrm_f2 *d_hist = nullptr;
    float *d_z = nullptr, *d_dcout = nullptr, *d_dcstate = nullptr;
    sdrm_clock_state *d_clock = nullptr;
    int8_t *d_out8 = nullptr;
    float *d_outf = nullptr;
    uint32_t *d_outlen = nullptr;
    uint32_t *d_flags = nullptr;  // [SLOTS][C] non-finite flags, one set per control slot
    // generic channels (sdrm_kernels.h): per-channel state in global memory, the list of such channels and the pointer table
    std::vector<float *> gen_ptr;  // [C] device allocations (null for the others)
    float **d_gen_state = nullptr;
    int *d_gen_list = nullptr;
    int n_gen = 0;
    // NCO pre-mix (allocated on first use)
    sdrm_nco_seg *d_nco_segs = nullptr, *h_nco_segs = nullptr;  // [SLOTS][nco_seg_cap]
    // constant-frequency oscillator in front of everything else, per channel (sdrm_batch_set_pre_offset: the file source's rx_offset)
    std::vector<int64_t> pre_offset;   // [C] Hz, 0 = none
    bool any_pre = false;
    float *d_pre_state = nullptr;      // [C] its fp32 phase, carried across calls
    float *d_pre_phase = nullptr;      // [C][phase stride] phase of every sample of the call
    sdrm_nco_seg *d_pre_segs = nullptr, *h_pre_segs = nullptr;    // [SLOTS][C] one batch per channel and call
    sdrm_chunk_ctl *d_ctl_pre = nullptr, *h_ctl_pre = nullptr;    // [SLOTS][C] the control block as that pass sees it
    size_t nco_seg_cap = 0;
    float *d_nco_state = nullptr, *d_nco_phase = nullptr, *d_nco_phase2 = nullptr;  // phases: one buffer per call parity
    hipStream_t s_nco = nullptr;                 // phase accumulator of the next call runs beside this call's stages
    hipEvent_t ev_phase[SDRM_CTL_SLOTS] = {};    // phases (and control block) of the call are on the device
    sdrm_f2 *d_nco_out = nullptr;
    std::vector<sdrm_nco_seg> nco_table;
    sdrm_f2 *d_in = nullptr;  // staging for the host-buffer API (lazy)
    // host (pinned) mirrors
    sdrm_chunk_ctl *h_ctl = nullptr;  // [SLOTS][C]
    uint32_t *h_outlen = nullptr;
    int8_t *h_out8 = nullptr;  // lazy
    hipEvent_t slot_done[SDRM_CTL_SLOTS] = {};   // clock stage of the call that used the slot has finished
    hipEvent_t ev_in[SDRM_CTL_SLOTS] = {};       // caller's stream position when the call was made (input ready)
    hipEvent_t ev_front[SDRM_CTL_SLOTS] = {};    // front-end (K1 + history roll) finished
    hipEvent_t ev_dc[SDRM_CTL_SLOTS] = {};       // DC blocker finished
    bool slot_used[SDRM_CTL_SLOTS] = {};
    // The three stages of consecutive calls overlap: each stage has its own stream, the stage-to-stage buffers
    // (z, dcout) are double buffered, and events order producer -> consumer and buffer reuse.
    hipStream_t s_front = nullptr, s_dc = nullptr, s_clock = nullptr;
    // Small batches: a grid of idle-spinning waves beside every clock-stage launch (sdrm_kernels.hip, k3_company)
    hipStream_t s_company = nullptr;
    hipEvent_t ev_company = nullptr;
    uint32_t *d_k3_done = nullptr;   // clock-stage workgroups finished, all launches
    uint32_t *d_counters = nullptr;  // [16] batch-lifetime device counters (DeviceBatch::counters)
    // in-call hand-off (DESIGN.md "stages of one call overlap"): tile stamps [C][hand_tiles_cap], DC-blocker counts [C]
    uint32_t *d_hand_tiles = nullptr;
    uint32_t hand_tiles_cap = 0;
    unsigned long long *d_hand_prog = nullptr;
    hipEvent_t ev_ctl[SDRM_CTL_SLOTS] = {};  // the call's control block is on the device
    bool hand_allowed = true;        // SDRM_HANDOFF=0 switches it off
    bool hand_used = false;          // a hand-off call has been enqueued since the device error word was last looked at
    hipStream_t s_hand_dc = nullptr, s_hand_clock = nullptr;  // a one-stream (serial) batch: side streams for a hand-off call's DC and clock stages, created on first use
    bool hand_side_last = false;     // the call enqueued last ran on them: the next call's first stream waits for its end
    uint64_t hand_calls = 0;         // diagnostics: calls enqueued with the hand-off
    uint64_t hand_refused = 0;       // ... calls that qualified but found the DEVICE's budget of waiting workgroups taken (../host/ledger.h)
    hipEvent_t ev_hand_done = nullptr;  // recorded behind a hand-off call (and nothing else): what its entry in the device's ledger names
    bool hand_listed = false;        // this batch may have an entry in the device's ledger (only the owner's thread touches this)
    uint64_t last_hand_call = 0;     // index of the last call that took the hand-off
    uint64_t hand_epoch = 0;  // hand-off calls since the batch was created -- never reset: a call's stamp value must be new
    uint32_t *d_placed = nullptr;    // [2] DC / clock-stage workgroups started, all launches (what the stream holds wait for)
    uint32_t k2_placed_target = 0, k3_placed_target = 0;  // the counters' values once every enqueued launch has started
    uint32_t k3_placed_after[SDRM_CTL_SLOTS] = {};        // ... once the clock stage of the call in that slot has
    uint32_t k3_done_target = 0;     // what the counter reads when the launch enqueued last has finished
    int device_error = 0;            // sticky: a kernel reported (d_counters[1]) that it gave up a bounded wait
    int company_blocks = 0;
    int company_rounds = 120;        // bound on the companion grid's life, in ~50 us looks at the counter
    int company_nops = 1;            // s_nop 7 between two vector instructions of a companion wave (1 / 4 / 16 / 64)
    int company_grid = 4096;         // the grid the companion stage takes when it is on
    bool hold_front = false;         // the front-end waits for the clock stage of call i-2 to have its workgroups placed
    // what the batch's self-calibration decided (sdrm_batch_create -> calibrate), for inspection: sdrm_batch_schedule
    bool calibrated = false;
    float est_front_ms = 0.0f, est_clock_ms = 0.0f;  // the creation-time estimates behind the companion grid's rule (full-length call)
    float calib_ms[3] = {0.0f, 0.0f, 0.0f};  // ms per full-length call: before, after, and what the calibration itself took
    // Online refinement (online_tune_*): the calibration at creation times calls WITHOUT Doppler correction; the first
    // stretch of calls that carry NCO batches re-decides the two settings that may change between any two calls (front
    // hold, companion grid) on the caller's own workload, from the device-side spacing of the clock stages' completions.
    struct OnlineTune {
        int state = 0;            // 0 not started, 1 measuring, 2 settled
        int phase = 0;            // while measuring: 1 the starting point's steady state, 2 the blocks, 3 the winner's probation
        int cand = 0, n = 0;      // block being run (0-3: bit 0 hold toggled, bit 1 companion grid toggled; 4 as is again; 5 the winner again), calls of the phase / block so far
        hipEvent_t ev[6][6] = {};     // per block: the clock stage's completion of the calls SKIP .. SKIP + TIMED of the block
        hipEvent_t watch[2][33] = {};  // the same over 32 intervals: [0] the starting point before the blocks, [1] the winner after them
        // ms per call: the four settings, as is / the winner again, the starting point's and the winner's steady state
        float ms[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        int best = -1;            // the first round's winner
        bool idle = false;        // this call is not part of a block (waiting for completions)
        bool base_hold = false;
        int base_company = 0;
        uint64_t sig = 0;         // the class of calls being refined: total samples (calls within a factor of two count as alike
        bool nco = false;         // once settled) and whether they carry NCO batches
        int chosen = -1;
        int restarts = 0;         // measurements given up because the calls stopped looking alike
        // after a winner has been kept: every 64th call of its class starts a sample of five intervals; two bad samples in a row
        // (more than 5 % behind the starting point's steady state) give the starting point back for good
        int guard_n = 0, guard_bad = 0;
        bool guard_pending = false, guard_alike = false;
    } tune;
    float *d_z2 = nullptr, *d_dcout2 = nullptr;
    bool any_nodc = false;
    bool serial = false;
    uint64_t calls = 0;
    uint32_t last_max_symbols = 0;  // upper bound of any channel's symbol count in the call enqueued last
    // Blocking calls of a one-channel batch (a plain fsk_demod handle) replay a graph: staged input -> control
    // record -> kernels -> counts and soft bits back, one launch and one wait per call.  One graph per input length.
    hipGraphExec_t sg_exec = nullptr;
    size_t sg_len = 0;            // input length the graph was built for
    uint32_t sg_width = 0;        // soft-bit bytes it copies back
    sdrm_f2 *h_in_stage = nullptr;  // pinned staging for the caller's (pageable) buffer
    bool sg_broken = false;       // building or instantiating the graph failed once: stay on the plain path
    size_t sg_prev_len = 0;       // length of the previous blocking call: a graph is built when a length repeats
    int last_slot = -1;
    hipStream_t stream = nullptr;  // private stream of the host-buffer API
    sdrm::DeviceBatch dev = {};
    uint32_t in_stride = 0;  // staging stride (samples)
    bool timing = false;
    TimingLane lanes[3];
    std::vector<uint32_t> last_lens;
    // pipelined host-buffer path (sdrm_batch_arena / _submit / _collect): the caller fills pinned arena slots, the
    // copy of call k+1 runs while call k computes, results come back through two pinned result sets
    sdrm_f2 *h_arena = nullptr;
    size_t arena_slots = 0;
    sdrm_f2 *d_in_ring[2] = {nullptr, nullptr};
    int8_t *d_out8_b = nullptr;      // second output set: calls alternate between the two once the arena exists
    uint32_t *d_outlen_b = nullptr;
    int8_t *h_res8[SDRM_RES_SETS] = {};
    uint32_t *h_reslen[SDRM_RES_SETS] = {};
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    hipEvent_t ev_res[SDRM_RES_SETS] = {};  // results of call k are in h_res8 / h_reslen [k % SDRM_RES_SETS]
    hipEvent_t ev_out_free[2] = {};         // the device output set of that parity has been copied back
    bool out_busy[2] = {false, false};
    uint32_t res_width[SDRM_RES_SETS] = {};  // bytes per channel copied back for that call
    uint32_t back_width[SDRM_RES_SETS] = {};
    uint32_t back_first[SDRM_RES_SETS] = {}, back_rows[SDRM_RES_SETS] = {};  // the span of channels that took part in that call
    int back_slot[SDRM_RES_SETS] = {};
    bool back_pending = false;               // the newest submitted call's copy-back is not enqueued yet
    uint64_t submitted = 0, collected = 0;
    uint64_t first_pipelined_call = 0;
    uint64_t stamp_only_call = 0;  // diagnostics: 0 = every call writes the cycle stamps, else only that call
    unsigned long long *d_timeline = nullptr;  // diagnostics: see sdrm_batch_timeline
    uint64_t timeline_first_call = 0;
};

static inline int8_t *out8_of(const sdrm_batch_t *b, uint64_t call) { return (b->d_out8_b && (call & 1)) ? b->d_out8_b : b->d_out8; }
static inline uint32_t *outlen_of(const sdrm_batch_t *b, uint64_t call) {
    return (b->d_outlen_b && (call & 1)) ? b->d_outlen_b : b->d_outlen;
}

// What the translation units of the library share beyond the C-ABI (internal linkage would do if they were one file).
//   sdrm_batch.hip   the batch object's life: create / destroy / grow / reset, timing lanes, oscillator buffers
//   sdrm_call.hip    the process path: enqueue_call and every entry point built on it, hand-off admission, waits
//   sdrm_tune.hip    the schedule: creation-time calibration, online refinement
//   sdrm_handle.hip  fsk_demod_* on a batch of one, probes, diagnostics
namespace sdrm_impl {
template <typename T>
int dev_alloc_zero(T **ptr, size_t count) {
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
// sdrm_batch.hip
void batch_free(sdrm_batch_t *b);
int quiesce(sdrm_batch_t *b);              // waits for everything this batch has put on the device -- its own streams only
int reset_all_streams(sdrm_batch_t *b);    // every stream of the batch back to its initial state (after a calibration)
uint32_t carried_cap(const sdrm_chan_params &p);                 // most samples the clock stage carries between calls
uint32_t symbols_bound(const sdrm_chan_params &p, uint32_t nz);  // most symbols a channel can produce from nz samples
int ensure_nco(sdrm_batch_t *b);           // the Doppler oscillator's buffers, allocated on first use
void timing_begin(sdrm_batch_t *b, int which, hipStream_t s, std::pair<hipEvent_t, hipEvent_t> *pr);
void timing_end(sdrm_batch_t *b, int which, hipStream_t s, const std::pair<hipEvent_t, hipEvent_t> &pr);
void timing_collect(sdrm_batch_t *b);
// sdrm_call.hip
int enqueue_call(sdrm_batch_t *b, const sdrm_f2 *d_in, size_t in_stride, const size_t *lens, hipStream_t caller,
                 const sdrm_nco_segment *segs, size_t n_segs);
int wait_for_all_calls(sdrm_batch_t *b);
void hand_release(sdrm_batch_t *b);        // this batch's entry in the device's ledger of waiting workgroups, if it has one
}  // namespace sdrm_impl
// the schedule tuning (sdrm_tune.hip)
int sdrm_calibrate(sdrm_batch_t *b, const sdrm_fsk_config *cfgs);
void sdrm_online_tune_before(sdrm_batch_t *b, bool with_nco, uint64_t sig);
void sdrm_online_tune_after(sdrm_batch_t *b, hipStream_t s_clock);

#endif  // SDRM_BATCH_IMPL_H
