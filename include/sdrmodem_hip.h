/*
 * sdrmodem_hip.h -- C-ABI of libsdrmodem_hip.so: the MI355X (gfx950) GMSK/FSK demodulation path.
 *
 * Drop-in boundary for dernasherbrezon/sdr-modem's fsk_demod operator and the dsp_worker push/pull
 * surface around it.  Plain pointers and sizes only; no C++/torch types.
 *
 * Layers, bottom-up:
 *   1. sdrm_batch_*      many independent RX channels demodulated per launch on one GPU (the new part): blocking
 *                        host-buffer call, device-resident call, and the pipelined host path (arena/submit/collect);
 *   2. fsk_demod_*       the reference operator, same names/signature/semantics (a batch of one);
 *   3. sdrm_batcher_*    the queue + worker surface of many clients in front of ONE batch (one device call per round);
 *   4. create_queue/...  and dsp_worker_*: the reference's queue + dsp_worker surface, feeding (2) or (3);
 *   5. sdrm_node_*       one process, many GPUs: a batcher per device and cost-based placement of each new client;
 *      sdrm_doppler_*    the reference's Doppler batching (per-second shifts from the caller's orbit model) for the
 *                        device-side NCO in front of the demodulator.
 *
 * Every entry point fails loudly (non-zero return and a message on stderr with the "<3>" systemd prefix the
 * reference uses; the void fsk_demod_process puts its handle into a sticky error state, see below) when no HIP
 * device is usable or the device path fails: there is NO CPU fallback in this library, and nothing in it aborts.
 */
#ifndef SDRMODEM_HIP_H
#define SDRMODEM_HIP_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
typedef struct sdrm_cf32_s { float re, im; } sdrm_cf32; /* layout of C99 `float complex` */
#else
#include <complex.h>
typedef float complex sdrm_cf32;
#endif

/* ------------------------------------------------------------------------------------------------
 * (2) Reference operator API -- replaces src/dsp/fsk_demod.h:11-15 (implementation src/dsp/fsk_demod.c:28-135).
 * Same argument meaning, same return codes (0, -ENOMEM, -1 for a bad cutoff/transition width,
 * lpf_taps.c:14-31), same ownership: *output is a buffer owned by the handle, valid until the next
 * process/destroy on it; oversize input prints "<3>requested buffer N is more than max: M" and yields
 * *output_len = 0 (fir_filter.c:147-152).  A handle is used by one thread at a time.
 * ------------------------------------------------------------------------------------------------ */
typedef struct fsk_demod_t fsk_demod;

int fsk_demod_create(uint64_t sampling_freq, uint32_t baud_rate, int64_t deviation, uint8_t decimation,
                     uint32_t transition_width, bool use_dc_block, uint32_t max_input_buffer_length,
                     fsk_demod **demod);
void fsk_demod_process(const sdrm_cf32 *input, size_t input_len, int8_t **output, size_t *output_len,
                       fsk_demod *demod);
void fsk_demod_destroy(fsk_demod *demod);
/* The reference's process() is void.  If the device path fails under a handle, the handle enters a sticky error state: a
 * "<3>" message on stderr once, *output_len = 0 on this and every later call.  sdrm_fsk_demod_error returns the handle's
 * sticky code (0 = healthy, negative errno otherwise); sdrm_last_error the code of the calling thread's last failed
 * fsk_demod_process.  The worker mirror ends that client, as the reference does on I/O errors (src/dsp_worker.c:56-64). */
int sdrm_fsk_demod_error(const fsk_demod *demod);
int sdrm_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * (1) Batched extension (not in the reference): C channels, each with its own fsk_demod_create()
 * parameters and its own streaming state, advanced together by one call.
 * ------------------------------------------------------------------------------------------------ */
typedef struct sdrm_batch_t sdrm_batch;

/* exactly the arguments of fsk_demod_create(), src/dsp/fsk_demod.h:11 */
typedef struct {
    uint64_t sampling_freq;
    uint32_t baud_rate;
    int64_t deviation;
    uint8_t decimation;
    uint32_t transition_width;
    bool use_dc_block;
    uint32_t max_input_buffer_length;
} sdrm_fsk_config;

/* parameters derived by create (fsk_demod.c:36-63), for inspection and parity tests */
typedef struct {
    uint32_t taps1_len, taps2_len, dc_length;
    float quad_gain, sps, gain_omega, gain_mu, omega_lim;
} sdrm_fsk_info;

#define SDRM_FLAG_KEEP_SOFT_F32 1u /* the caller reads the float soft bits (clock-recovery output) of a call; the stage
                                    * produces them in any case, the int8 output is converted from them */

#define SDRM_FLAG_FAST_FMA 2u      /* REMOVED in round 6: sdrm_batch_create answers -ENOTSUP.  (Rounds 2-5: both low-pass filters
                                    * with fused multiply-adds.  Not the reference's arithmetic: it failed the reference's own
                                    * +-2 LSB tolerance, test/test_fsk_demod.c:47, on lucky7 without DC blocker -- 19 LSB,
                                    * two hard-bit flips -- so nobody could ship it.  The value stays reserved.) */

#define SDRM_FLAG_NO_CALIBRATION 4u /* skip the creation-time timing of the batch's own pipeline (0.1 - 0.4 s for large batches): the
                                     * schedule starts from its rules and is refined online on the caller's own calls, whatever
                                     * their class.  What a batcher passes for its batch: its slots are placeholders when it is
                                     * created, its real clients arrive later.  Results never depend on the schedule. */

/* device < 0: current HIP device.  Returns 0, -ENOMEM, -1 (bad parameters), -ENODEV (no usable GPU), -ENOTSUP (a filter too long
 * for a tile's LDS -- about ten thousand taps --, more than 16384 samples per symbol, decimation beyond the filter length).
 * Any samples per symbol the reference accepts (src/dsp/fsk_demod.c:53-63) is accepted, fewer than one included (round 5: such a
 * channel's calls run from global memory, sdrm_batch_wild_calls counts them): up to ~228 a channel runs the fast LDS-resident DC and
 * clock stages, beyond that (or with a DC boxcar longer than 7712 samples) their generic forms with the state in global memory
 * ("generic channels") -- same bits, ~2 ms per 131072-sample call. */
int sdrm_batch_create(const sdrm_fsk_config *cfgs, size_t n_channels, int device, uint32_t flags, sdrm_batch **batch);
void sdrm_batch_destroy(sdrm_batch *batch);
size_t sdrm_batch_channels(const sdrm_batch *batch);
int sdrm_batch_info(const sdrm_batch *batch, size_t channel, sdrm_fsk_info *info);
/* copy out the designed low-pass taps (design order) of stage 1 or 2; returns the tap count */
size_t sdrm_batch_taps(const sdrm_batch *batch, size_t channel, int stage, float *dst, size_t dst_cap);

/* input_lens[c] == SDRM_LEN_ABSENT: channel c takes no part in this call -- no output, and its stream state (filter
 * histories, DC blocker, timing loop) stays exactly as it is.  That is NOT what a length of 0 means: an empty call is
 * answered the way the reference answers it (its clock stage may emit a symbol from the samples it carries when
 * samples/symbol >= 8, src/dsp/clock_recovery_mm.c:94-135).  The batcher marks the clients that have no buffer in a round
 * this way.  Accepted by every call that takes input_lens. */
#define SDRM_LEN_ABSENT ((size_t) -1)

/* Host-buffer call: inputs[c] points at input_lens[c] complex samples (may be NULL when the length is 0).
 * Blocks until done.  outputs[c] / output_lens[c] receive borrowed pointers into handle-owned host memory,
 * valid until the next process/destroy.  Channels whose input exceeds their max get output_len 0. */
int sdrm_batch_process(sdrm_batch *batch, const sdrm_cf32 *const *inputs, const size_t *input_lens,
                       int8_t **outputs, size_t *output_lens);

/* Device-resident call (what bench.py times): d_input is a device pointer to [C][in_stride] complex samples,
 * channel-major; input_lens is a HOST array.  Enqueues on `stream` (a hipStream_t, NULL = default stream) and
 * returns without synchronising.  Results stay on the device: see sdrm_batch_device_outputs(). */
int sdrm_batch_process_device(sdrm_batch *batch, const void *d_input, size_t in_stride, const size_t *input_lens,
                              void *stream);
/* Stages of consecutive device-resident calls overlap on the batch's own streams; `stream` only marks when the input
 * is ready.  sdrm_batch_wait makes a stream wait (on the device) for the latest call's results; sdrm_batch_sync blocks
 * the host until they are there. */
int sdrm_batch_wait(sdrm_batch *batch, void *stream);
int sdrm_batch_sync(sdrm_batch *batch);
/* The call reads d_input on the batch's own front-end stream, AFTER sdrm_batch_process_device has returned: refilling or
 * freeing the buffer on `stream` right away (a caching allocator reusing the block) would race it.  This makes `stream`
 * wait (on the device) until the latest call has consumed its input -- the front-end and the history roll, not the rest. */
int sdrm_batch_wait_input(sdrm_batch *batch, void *stream);
/* device pointers: int8 soft bits [C][out_stride], per-channel counts uint32[C], float soft bits or NULL.
 * A count of 0xffffffff (SDRM_COUNT_VOID) is not a count: the call took the in-call hand-off and one of its stages gave up a
 * bounded wait for the stage in front of it (a failed launch, a device in trouble) -- every result of that call is void and
 * the batch is in error for good.  sdrm_batch_sync / _collect / _fetch / _process report it as -ETIMEDOUT; a consumer that
 * reads the counts on its own stream (sdrm_batch_wait) must test for the value itself. */
#define SDRM_COUNT_VOID 0xffffffffu
int sdrm_batch_device_outputs(sdrm_batch *batch, void **d_out_i8, size_t *out_stride, void **d_out_len,
                              void **d_out_f32);
/* after a synchronised call: copy channel c's float soft bits of the last call to host */
int sdrm_batch_last_soft(sdrm_batch *batch, size_t channel, float *dst, size_t dst_cap, size_t *len);
/* copy the last call's int8 outputs of every channel to host: lens[C], data[C][stride] */
int sdrm_batch_fetch(sdrm_batch *batch, int8_t *data, size_t stride, size_t *lens);

/* ---- Doppler pre-correction / NCO (next scope row: reference src/dsp/doppler.c:116-190, src/dsp/sig_source.c:43-75).
 * A segment mixes `len` consecutive input samples of `channel` with an oscillator at the integer frequency freq_hz
 * (fp32 phase accumulator carried across segments and calls, cos/sin in double, exactly as sig_source does).
 * Segments of one channel must be consecutive in the array and cover that channel's whole input of the call;
 * channels without segments are demodulated uncorrected.  The _nco calls are the plain calls plus the mix in front. */
typedef struct {
    uint32_t channel;
    uint32_t len;
    int64_t freq_hz;
} sdrm_nco_segment;
int sdrm_batch_process_nco(sdrm_batch *batch, const sdrm_cf32 *const *inputs, const size_t *input_lens,
                           const sdrm_nco_segment *segments, size_t n_segments, int8_t **outputs, size_t *output_lens);
int sdrm_batch_process_device_nco(sdrm_batch *batch, const void *d_input, size_t in_stride, const size_t *input_lens,
                                  const sdrm_nco_segment *segments, size_t n_segments, void *stream);
/* A channel of a batch can be handed to a new stream (a client disconnects, another connects): its streaming state is
 * cleared and, with config != NULL, its configuration replaced -- what fsk_demod_destroy + fsk_demod_create do for a
 * single handle.  Longer filters, a longer DC boxcar or the batch's first DC blocker make the batch grow (the other
 * channels keep their streams); only the buffer length is fixed for the batch's life: max_input_buffer_length must not
 * exceed the largest the batch was created with (-ENOTSUP).  Waits for all enqueued calls first. */
int sdrm_batch_reset_channel(sdrm_batch *batch, size_t channel, const sdrm_fsk_config *config);
/* From the next call on the channel's input is mixed with ONE oscillator at the integer frequency freq_hz (fp32 phase carried
 * across calls, started at 0 now) in front of everything else: what the reference's file source does with RxRequest.rx_offset
 * (src/sdr/file_source.c:120-128) before the samples reach dsp_worker_put.  NCO batches of the same call (the Doppler
 * correction) run BEHIND it -- two oscillators in series, every sample rounded to fp32 in between.  0 switches it off;
 * sdrm_batch_reset_channel switches it off for the channel's next client.  Waits for enqueued calls. */
int sdrm_batch_set_pre_offset(sdrm_batch *batch, size_t channel, int64_t freq_hz);

/* Pipelined host-buffer path.  The reference's boundary hands over HOST buffers (src/dsp/fsk_demod.h:13, filled by
 * queue_put's memcpy, src/queue.c:99-154), so at batch scale the host link decides the rate.  sdrm_batch_arena pins
 * `slots` (>= 2) input slots of [channels][*chan_stride] complex samples each (slot s starts at base + s * *slot_stride)
 * that producers fill directly -- the memcpy of queue_put lands here instead of in a queue node.  sdrm_batch_submit
 * enqueues one call on the data of a slot: ONE host-to-device copy on a copy stream (it overlaps the kernels of the
 * previous call), the kernels, and the copy-back of the soft bits; it returns at once (-EAGAIN when three calls are
 * already uncollected).  The slot may be refilled once the call has been collected.  sdrm_batch_collect blocks until
 * the oldest submitted call is done; outputs[c] point into pinned result buffers owned by the batch, valid until the
 * next collect returns.  `segments` as in sdrm_batch_process_nco (NULL, 0 = no Doppler correction).  Do not mix with the
 * other process calls while a submitted call is uncollected. */
int sdrm_batch_arena(sdrm_batch *batch, size_t slots, sdrm_cf32 **base, size_t *chan_stride, size_t *slot_stride);
int sdrm_batch_submit(sdrm_batch *batch, size_t slot, const size_t *input_lens, const sdrm_nco_segment *segments,
                      size_t n_segments);
int sdrm_batch_collect(sdrm_batch *batch, int8_t **outputs, size_t *output_lens);

/* diagnostics: copy channel c's mixed IQ of the last call (interleaved re,im) to host */
int sdrm_batch_last_mixed(sdrm_batch *batch, size_t channel, float *dst, size_t dst_cap_samples, size_t *len);

/* The reference's batching of the correction (doppler.c:128-180): batches end at one-second boundaries, the shift is
 * evaluated once per second by `fn` (the orbit model -- SGP4 in the reference, doppler.c:31-42 -- stays with the
 * caller), interpolated linearly inside the second and truncated to integer Hz.  One planner per channel. */
typedef double (*sdrm_doppler_shift_fn)(void *user, uint64_t second);
typedef struct sdrm_doppler_t sdrm_doppler;
int sdrm_doppler_create(uint64_t sampling_freq, sdrm_doppler_shift_fn fn, void *user, sdrm_doppler **out);
size_t sdrm_doppler_plan(sdrm_doppler *d, uint32_t channel, size_t input_len, sdrm_nco_segment *segments, size_t cap);
void sdrm_doppler_destroy(sdrm_doppler *d);

/* Per-kernel device time, measured with HIP events on the launch stream when enabled.
 * which: 0 = front-end (LPF1+quadrature demod+LPF2), 1 = DC blocker, 2 = clock recovery + int8. */
int sdrm_batch_timing_enable(sdrm_batch *batch, int enable);
int sdrm_batch_timing_read(sdrm_batch *batch, int which, double *total_ms, uint64_t *launches);

/* Channel-calls whose timing loop left the range in which it provably advances (a sample of the clock stage's input beyond
 * the channel's safe amplitude -- discriminator gains in the thousands, i.e. a deviation of a few Hz, on noise; or fewer
 * than ~1.01 samples per symbol): the reference's loop may then stand still or walk BACKWARDS through its buffer
 * (src/dsp/clock_recovery_mm.c:121-122), which the LDS-resident stage cannot follow, so such a call is run from global
 * memory, statement by statement -- same bits, slower.  Counted since the batch was created; waits for enqueued calls.
 * One bound applies to such a channel as to every other: at most 255 samples are carried from one call to the next (the
 * newest ones).  The reference carries working_len - last_index samples (clock_recovery_mm.c:127-135), which after a backward
 * walk that ends at a negative position can be tens of thousands -- as long as they and the next call still fit its
 * output_len + 8 working buffer; a full-length next call writes past it.  From such a call on the two streams may differ;
 * the oracle applies the same bound, so "equals the oracle" holds with it.  A loop in lock carries fewer than 24 samples. */
int sdrm_batch_wild_calls(sdrm_batch *batch, uint64_t *count);

/* In-call hand-off.  A call that meets an idle batch -- every blocking call (the reference's caller waits for
 * fsk_demod_process, src/dsp_worker.c:75), the first call of a pipelined run -- cannot hide its front-end and DC blocker behind
 * an earlier call's clock recovery.  Its three stages are then made resident together: the DC blocker starts on a channel's
 * first finished front-end tiles, the clock recovery on the first DC blocks (per-tile stamps and per-channel output counts in
 * device memory, device-scope accesses; DESIGN.md "stages of one call overlap").  Same bits; 256 channels x 131072 samples
 * blocking: 4.28 -> 2.72 ms.  Used for batches small enough that waiting workgroups cannot starve the stage they wait for;
 * SDRM_HANDOFF=0 (environment, read at batch creation) switches it off.  This counts the calls that took it. */
int sdrm_batch_handoff_calls(sdrm_batch *batch, uint64_t *count);
/* Many clients, one handle each (the reference's layout, src/dsp_worker.c:188): every private handle's call is a chain of
 * one-workgroup kernels and the device runs about three such chains at once -- 70-95 Msamples/s in total whatever the number of
 * handles.  sdrm_fsk_demod_share(n, wait_us) makes the fsk_demod handles created FROM NOW ON share one batcher of n slots
 * (geometry = the first such handle's configuration; a handle that does not fit it gets a private batch): fsk_demod_process
 * stays the blocking call it is, and the clients' buffers go to the device as one call per round (a round waits up to wait_us
 * for the clients that have not delivered yet) -- 1.7 Gsamples/s with 64 clients, 3.0 with 256 (profiles/r06_handles.txt).
 * The same as SDRM_SHARED_SLOTS / SDRM_SHARED_WAIT_US in the environment, for a server that would rather say it in main().
 * slots = 0: handles created from now on are private again.  Returns 0, -1 (more than 65536 slots), -EBUSY (the pool exists
 * with another size). */
int sdrm_fsk_demod_share(size_t slots, uint32_t max_wait_us);
/* The workgroups of a hand-off call that wait for the stage in front of them hold compute units of the DEVICE, and a server
 * runs one handle per client (src/dsp_worker.c:188, src/tcp_server.c:659): admission is therefore counted per device across
 * every batch and handle of the process -- a call takes the hand-off only while the waiting workgroups on the device, its own
 * included, stay within the limit (192; 16 where a DC workgroup leaves a front-end workgroup no room beside it) and no OTHER plain
 * handle's blocking call is in flight (such calls are chains of one-workgroup kernels that share a handful of hardware queues:
 * the hand-off pays for a client that calls alone and costs the others when several call together), and runs its stages in
 * stream order otherwise.  Process-wide totals for a device (< 0: the current one): calls admitted, calls that
 * qualified but found the budget taken, the largest number of workgroups that were waiting at once.  Any pointer may be NULL. */
int sdrm_handoff_stats(int device, uint64_t *taken, uint64_t *refused, uint32_t *peak_waiting);

/* Stage probes for tests: run ONE stage of the device pipeline on a host vector (state-free where the
 * stage is). Return 0 on success. */
int sdrm_probe_atan2(const float *y, const float *x, float *out, size_t n);
/* the front-end's discriminator phase as the kernel runs it -- out[i] = gain * fast_atan2f(y[i] conj(y[i-1])) for a stream
 * of n complex samples (interleaved re, im; y[-1] = 0), reference src/dsp/quadrature_demod.c:57-73 -- including the
 * kernel's choice, per wave of 960 samples, between its short form and the general one (fast_waves[w], may be NULL) */
int sdrm_probe_quad(const float *iq, size_t n, float gain, float *out, uint32_t *fast_waves);
/* sums[i] / length as the DC blocker computes it (reference src/dsp/dc_blocker.c:63): three instructions with a fall-back
 * to the division proper for denormal / non-finite quotients; must equal the IEEE quotient bit for bit */
int sdrm_probe_boxcar_div(const float *sums, uint32_t length, float *out, size_t n);

/* Diagnostics (tools/k3_probe.py, tools/sweep_point.py; never needed for results).
 * sdrm_batch_k3_stamps: enable = 1 makes every call (enable = k > 1: only the k-th call from now) record cycle counts
 * inside the kernels; `out` receives 4 x uint64 per clock-stage wave {cycles waiting for staged samples, cycles in the
 * symbol loops, steps | 100 MHz ticks << 32, loop iterations} followed by the front-end's per-phase cycle sums and one
 * DC-blocker channel's cycles.  Returns the number of wave records (one per 16 channels; a batch run with 64
 * channels per clock-stage workgroup fills the first quarter).
 * sdrm_batch_timeline: enable != 0 attaches a table for the next 64 calls; `out` receives one row per call made since,
 * {front start, front end, dc start, dc end, clock start, clock end} in 10 ns ticks of the device's reference clock (first
 * workgroup start / last workgroup end).  Returns the number of rows. */
int sdrm_batch_k3_stamps(sdrm_batch *batch, int enable, unsigned long long *out, size_t max_waves);
int sdrm_batch_timeline(sdrm_batch *batch, int enable, unsigned long long *out, size_t max_rows);

/* The schedule a batch runs with.  A batch of 32 channels or more calibrates itself when it is created: it times full-length
 * calls of its own pipeline on a synthetic row (a few calls, at least ~4 ms, per candidate: clock-stage workgroup shape, the front-end's hold
 * for the clock stage's placement, the companion grid beside the clock stage), keeps what is fastest by more than 3 %, and
 * puts every stream back to its initial state.  SDRM_AUTOTUNE=0 keeps the built-in starting point (channel-count rules). */
typedef struct {
    int k3_lanes, k3_ring, k3_plain; /* clock stage: channels per workgroup, ring length in samples, plain ring */
    int front_hold;                  /* 1: the front-end waits for the clock stage two calls back to have its workgroups placed */
    int company_blocks;              /* workgroups of the companion grid beside each clock stage (0: none) */
    int calibrated;                  /* 1: measured at creation; 0: the starting point (small batch, switched off, forced) */
    float ms_before, ms_after;       /* ms per full-length call with the starting point / with what was kept */
    float ms_spent;                  /* what the calibration took */
    /* The calibration times calls without Doppler correction.  The first calls that carry NCO batches re-decide the two settings
     * that may change between any two calls -- front hold, companion grid -- on the caller's own workload: four settings, eight
     * calls each, timed by the clock stages' completions on the device (median interval), then the starting point and the winner
     * once more; a winner of both rounds (by more than 3 %) then has to beat the starting point's steady state (40 calls before
     * the blocks) with its own (40 calls after them).  The same for calls of less than half the calibrated length.  Other calls
     * keep the calibrated setting; results never depend on the schedule. */
    int online_state;                /* 0: not started (no such calls yet), 1: measuring, 2: settled */
    int online_choice;               /* -1, or what was kept: 0 as it was, bit 0: hold toggled, bit 1: companion grid toggled */
    float online_ms[8];              /* ms per call: the four settings, as it was / the winner again, the starting point's and the
                                      * winner's steady state (0: not measured) */
} sdrm_batch_schedule_info;
int sdrm_batch_schedule(const sdrm_batch *batch, sdrm_batch_schedule_info *info);

const char *sdrm_version(void);
int sdrm_device_count(void);

/* ---------------------------------------------------------------------------------------------------
 * Per-GPU batcher: the queue + worker surface of many clients in front of ONE batch.
 * Channel c of the batcher plays the part of one client's queue and demodulator together: its producer thread calls
 * sdrm_batcher_put (the memcpy of queue_put, src/queue.c:99-154, lands in the pinned input arena), its consumer thread
 * calls sdrm_batcher_take / sdrm_batcher_complete (take_buffer_for_processing + fsk_demod_process +
 * complete_buffer_processing of src/dsp_worker.c:58-97 in one step: what comes back are the soft bits).  A batcher
 * thread launches one batched call per round -- when every open channel has delivered its buffer or max_wait_us after
 * the round's first buffer -- keeps up to three rounds on the device and hands the results back in order per channel.
 * Queue behaviour is the reference's: `blocking` producers (file source) wait while all `slots` rounds are taken, live
 * producers overwrite their newest buffer that is not on the device yet and log "<3>queue is full";
 * sdrm_batcher_interrupt is the poison pill (buffers put before it are still delivered, then take returns NULL).
 * --------------------------------------------------------------------------------------------------- */
typedef struct sdrm_batcher_t sdrm_batcher;
typedef struct {
    uint32_t slots;       /* rounds that can be filling / in flight / waiting for their consumers (>= 4) */
    uint32_t max_wait_us; /* launch a partly filled round this long after its first buffer */
    bool blocking;        /* true: file-source behaviour; false: live-source behaviour */
} sdrm_batcher_config;
int sdrm_batcher_create(const sdrm_fsk_config *configs, size_t n_channels, int device, const sdrm_batcher_config *config,
                        sdrm_batcher **batcher); /* config NULL = {4, 2000, true}; -ENODEV without a HIP device */
void sdrm_batcher_put(sdrm_batcher *batcher, size_t channel, const sdrm_cf32 *buffer, size_t len);
void sdrm_batcher_take(sdrm_batcher *batcher, size_t channel, int8_t **output, size_t *output_len);
void sdrm_batcher_complete(sdrm_batcher *batcher, size_t channel);
void sdrm_batcher_interrupt(sdrm_batcher *batcher, size_t channel);
/* the channel's consumer is gone for good (write error, client torn down): closes the channel like the poison pill and
 * discards everything of it that has not been taken, finished or not, so that the shared rounds retire and the other
 * clients keep moving -- in the reference a dead consumer only fills its own queue (src/queue.c).  The slot can be
 * handed to a new client with sdrm_batcher_reset_channel afterwards. */
void sdrm_batcher_abandon(sdrm_batcher *batcher, size_t channel);
/* A channel changes hands (one client leaves, another arrives): waits until everything put on the channel has been
 * consumed, lets the rounds in flight finish, then resets the channel as sdrm_batch_reset_channel does (config NULL =
 * same configuration) and reopens it if it had been interrupted.  Other channels keep their streams. */
int sdrm_batcher_reset_channel(sdrm_batcher *batcher, size_t channel, const sdrm_fsk_config *config);
/* the same, and the new client's input passes the constant-frequency oscillator of sdrm_batch_set_pre_offset first (the file
 * source's RxRequest.rx_offset; 0 = none) */
int sdrm_batcher_reset_channel_offset(sdrm_batcher *batcher, size_t channel, const sdrm_fsk_config *config, int64_t rx_offset_hz);
/* Doppler pre-correction for one channel: `planner` (borrowed, see sdrm_doppler_create) is asked for the segments of
 * every buffer of that channel when its round is launched; NULL switches it off */
int sdrm_batcher_set_doppler(sdrm_batcher *batcher, size_t channel, sdrm_doppler *planner);
/* 0 while the device answers; the code of the first failed device call afterwards (sticky).  A failed device ends every
 * client of the batcher: sdrm_batcher_take returns NULL from then on (as after a poison pill -- this query tells the two
 * apart), sdrm_batcher_put drops.  The streams' state lived on the device: the batcher has to be destroyed. */
int sdrm_batcher_error(const sdrm_batcher *batcher);
size_t sdrm_batcher_channels(const sdrm_batcher *batcher);
uint64_t sdrm_batcher_rounds(const sdrm_batcher *batcher); /* batched calls launched so far */
void sdrm_batcher_destroy(sdrm_batcher *batcher);

/* ---------------------------------------------------------------------------------------------------
 * Node front door: ONE process, every GPU of the node.  sdr-modem is one process with one DSP thread per RX client
 * (src/tcp_server.c:659 creates the client's dsp_worker, src/sdr_worker.c:25-55 feeds the workers of a source from that
 * source's thread, src/dsp_worker.c:188 starts the thread); nothing in it knows about devices.  A node owns one batcher per
 * device (or several) and PLACES every new client:
 *   - a client costs fs x (4 T1 + 2 T2 / decimation) -- the front-end's multiply-adds per second of signal, the stage that
 *     bounds a full GPU; sdrm_channel_cost returns it (the function sdr-modem_amd/shard.py cuts a known table with);
 *   - the least-loaded healthy device with a free slot takes the client; a client whose source (source_id != 0: the SDR /
 *     centre frequency it listens to) already has clients on a device joins them while that device is within 8 % of the
 *     least-loaded one (a source's channels together);
 *   - sdrm_node_detach frees the slot for the next client, whichever device that turns out to serve;
 *   - a device whose batcher has failed (sdrm_batcher_error, sticky) takes no new clients; its own clients end through the
 *     batcher's error path, the other devices' clients are not touched.
 * sdrm_node_attach only RESERVES the slot (closed, so that it holds up nobody's rounds): the client's worker -- created with
 * sdrm_worker_config.batcher / .batcher_channel from the slot, or simply with sdrm_worker_config.node, which does the
 * attach and the detach by itself -- gives the slot the client's parameters and a clean state and opens it
 * (sdrm_batcher_reset_channel).  No collective is involved: one process, so each device gets its tables by its own
 * host-to-device copies.  Returns: 0; -ENODEV without a HIP device; -EBUSY when every slot of every healthy device is
 * taken; the devices' error code when all of them have failed.
 * --------------------------------------------------------------------------------------------------- */
typedef struct sdrm_node_t sdrm_node;
typedef struct {
    const int *devices;         /* HIP device of each batcher; NULL = device i % visible devices for batcher i */
    size_t n_batchers;          /* 0 = one per visible device */
    size_t slots_per_batcher;   /* client slots of each batcher */
    sdrm_fsk_config geometry;   /* what every slot is created with; a client's own parameters replace it when its worker
                                 * takes the slot (longer filters make that device's batch grow).  max_input_buffer_length
                                 * = the server's buffer_size, fixed for the node's life */
    sdrm_batcher_config batcher; /* slots 0 = {4, 2000, true} */
} sdrm_node_config;
typedef struct {
    sdrm_batcher *batcher;      /* borrowed: owned by the node */
    size_t channel;
    int device;
    size_t batcher_index;
} sdrm_node_slot;
typedef struct {
    int device;
    sdrm_batcher *batcher;
    size_t slots, clients;
    double load;                /* sum of sdrm_channel_cost over the device's clients */
    uint64_t attached;          /* clients placed on this batcher since the node was created */
    int error;                  /* sdrm_batcher_error of the device's batcher */
} sdrm_node_stat;
int sdrm_node_create(const sdrm_node_config *config, sdrm_node **node);
int sdrm_node_attach(sdrm_node *node, const sdrm_fsk_config *client, uint64_t source_id, sdrm_node_slot *slot);
int sdrm_node_detach(sdrm_node *node, const sdrm_node_slot *slot);
size_t sdrm_node_batchers(const sdrm_node *node);
int sdrm_node_stat_read(const sdrm_node *node, size_t index, sdrm_node_stat *stat);
double sdrm_channel_cost(const sdrm_fsk_config *config);
void sdrm_node_destroy(sdrm_node *node); /* after the clients' workers have been destroyed */

/* ------------------------------------------------------------------------------------------------
 * (3) Queue + worker surface (host side, C, pthreads).
 * Queue: same names/semantics as src/queue.h:10-18 (blocking put for file sources, overwrite-newest for
 * live sources, poison pill, detached node while processing).
 * Worker: mirror of src/dsp_worker.h:14-22 taking a plain C config instead of the protobuf RxRequest /
 * libconfig server_config (those headers are outside the path; field-by-field mapping in INTEGRATION.md).
 * ------------------------------------------------------------------------------------------------ */
typedef struct queue_t queue;
int create_queue(uint32_t buffer_size, uint16_t queue_size, bool blocking, queue **queue);
int queue_put(const sdrm_cf32 *buffer, size_t len, queue *queue);
void take_buffer_for_processing(sdrm_cf32 **buffer, size_t *len, queue *queue);
void complete_buffer_processing(queue *queue);
void interrupt_waiting_the_data(queue *queue);
void destroy_queue(queue *queue);

typedef void (*sdrm_release_fn)(void *user);
typedef struct {
    /* from RxRequest (src/api.pb-c.h:104-121, read at src/dsp_worker.c:120-163) */
    uint64_t rx_sampling_freq;
    uint32_t demod_baud_rate;
    int64_t demod_fsk_deviation;
    uint32_t demod_decimation;
    uint32_t demod_fsk_transition_width;
    bool demod_fsk_use_dc_block;
    bool rx_dump_file;      /* write rx.sdr2demod.<id>.cf32 */
    int demod_destination;  /* 0 FILE, 1 SOCKET, 2 BOTH (api.proto DemodDestination) */
    /* from server_config (src/server_config.h:16-40) */
    uint32_t buffer_size;
    uint16_t queue_size;
    bool rx_file_source;    /* rx_sdr_type == RX_SDR_TYPE_FILE => blocking queue */
    const char *base_path;
    /* optional Doppler pre-correction (RxRequest.doppler in the reference, src/dsp_worker.c:120-136): the per-second
     * shift in Hz; NULL = none.  The orbit model behind it (SGP4 from the TLE) stays with the caller. */
    sdrm_doppler_shift_fn doppler_shift;
    void *doppler_user;
    /* optional: demodulate through channel `batcher_channel` of a shared per-GPU batcher instead of a private
     * fsk_demod handle and queue (NULL = the reference's one-demodulator-per-worker layout).  The channel's
     * configuration in the batcher must equal this worker's.  queue_size / rx_file_source are then properties of the
     * batcher (its slots / blocking). */
    sdrm_batcher *batcher;
    size_t batcher_channel;
    /* optional (batcher == NULL): let a node place this client -- the worker attaches itself (sdrm_node_attach with
     * source_id), serves the slot it is given and detaches when it is destroyed.  A device that fails between the placement
     * and the slot's reset is skipped and the placement repeated. */
    sdrm_node *node;
    uint64_t source_id;
    /* optional: the file source's frequency offset (RxRequest.rx_offset applied by src/sdr/file_source.c:120-128 with a
     * sig_source of its own, upstream of dsp_worker_put): every buffer is mixed with ONE oscillator at this integer
     * frequency on the device, phase carried across buffers, in front of everything else (sdrm_batch_set_pre_offset).
     * Together with doppler_shift the two oscillators run in series with separate phases, every sample rounded to fp32 in
     * between, as in the reference (file_source.c:122, then src/dsp_worker.c:65-71). */
    int64_t rx_offset_hz;
    /* optional: called with doppler_user when the worker is destroyed (a factory's state, integration/doppler_factory_ref.c) */
    sdrm_release_fn doppler_release;
} sdrm_worker_config;

typedef struct dsp_worker_t dsp_worker;
#ifndef SDRM_REFERENCE_DSP_WORKER_CREATE /* defined by a translation unit that declares the reference's signature itself */
int dsp_worker_create(uint32_t id, int client_socket, const sdrm_worker_config *config, dsp_worker **result);
#endif
/* the same constructor under a name of its own: integration/dsp_worker_ref.c defines dsp_worker_create WITH THE REFERENCE'S
 * SIGNATURE (uint32_t, int, struct server_config *, struct RxRequest *, dsp_worker **; src/dsp_worker.h:22) for a build
 * inside sdr-modem -- that definition is then the program's dsp_worker_create, and it reaches the library through this name */
int sdrm_dsp_worker_create(uint32_t id, int client_socket, const sdrm_worker_config *config, dsp_worker **result);
void dsp_worker_put(sdrm_cf32 *output, size_t output_len, dsp_worker *worker);
void dsp_worker_shutdown(void *arg, void *data);
bool dsp_worker_find_by_id(void *id, void *data);
void dsp_worker_destroy(void *data);

/* ---- wire framing of the RX path (reference src/api.h:8-27, api.proto:35-49,69-72, src/api_utils.c:82-108).
 * Header = {u8 protocol version 0, u8 type, u32 body length in network order}.  The server answers an RxRequest with a
 * Response and then streams the worker's soft bits raw on the same socket (src/tcp_server.c:677, src/dsp_worker.c:93-95):
 * sdrm_wire_write_response + a dsp_worker created with that socket and demod_destination SOCKET reproduce those bytes.
 * sdrm_wire_decode_rx_request fills the request half of a worker configuration from an RxRequest body (0 / -1). */
#define SDRM_WIRE_PROTOCOL_VERSION 0
#define SDRM_WIRE_TYPE_RX_REQUEST 0
#define SDRM_WIRE_TYPE_SHUTDOWN 1
#define SDRM_WIRE_TYPE_RESPONSE 2
#define SDRM_WIRE_STATUS_SUCCESS 0
#define SDRM_WIRE_STATUS_FAILURE 1
#define SDRM_WIRE_MAX_MESSAGE 65536u /* largest body sdrm_wire_read_header accepts (an RxRequest with two TLE lines is < 300 bytes) */
int sdrm_wire_write_response(int socket, uint32_t status, uint32_t details);
/* 0; -1 peer closed / read error; -2 unknown protocol version; -3 body longer than SDRM_WIRE_MAX_MESSAGE */
int sdrm_wire_read_header(int socket, uint8_t *type, uint32_t *message_length);
int sdrm_wire_decode_rx_request(const uint8_t *body, size_t len, sdrm_worker_config *config, int *has_doppler);

#ifdef __cplusplus
}
#endif
#endif /* SDRMODEM_HIP_H */
