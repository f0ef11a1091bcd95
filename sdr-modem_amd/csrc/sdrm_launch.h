// sdrm_launch.h -- host-callable launchers of the gfx950 kernels (defined in sdrm_kernels.hip).
#ifndef SDRM_LAUNCH_H
#define SDRM_LAUNCH_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "sdrm_kernels.h"

namespace sdrm {

// device-resident, batch-wide pointers and strides
struct DeviceBatch {
    int n_channels;
    const sdrm_chan_params *params;  // [C]
    const sdrm_chunk_ctl *ctl;       // [C] for the call being launched
    const float *tap_pool;           // reversed taps of all channels
    const float *atan_tab;           // [257]
    const float *mmse_bank;          // [129*8]
    sdrm_f2 *raw_hist;               // [C][2][hist_stride]
    uint32_t hist_stride;
    float *z;                        // [C][z_stride]   LPF2 output
    float *dcout;                    // [C][z_stride]   DC-blocker output
    uint32_t z_stride;
    float *dc_state;                 // pool, per-channel offsets in params
    sdrm_clock_state *clock_state;   // [C]
    int8_t *out_i8;                  // [C][out_stride]
    float *out_f32;                  // [C][out_stride] or nullptr
    uint32_t *out_len;               // [C]
    uint32_t *nonfinite;             // [C] of the call's control slot: set by K1/K2 when NaN/Inf reaches the clock stage
    uint32_t out_stride;
    unsigned long long *timeline;    // diagnostics (normally null): [64 calls][front, dc, clock][first start, last end], 100 MHz ticks
    uint32_t tl_row;
    unsigned long long *k3_stamps;   // diagnostics (normally null): per K3 wave {staging cycles, loop cycles, steps, iterations}
    // NCO pre-mix (optional): per-call segment table, per-channel fp32 phase, phase scratch and mixed-IQ buffer
    const sdrm_nco_seg *nco_segs;
    float *nco_phase_state;          // [C]
    float *nco_phase;                // [C][nco_phase_stride] phase of every sample of the call
    sdrm_f2 *nco_out;                // [C][nco_stride] input after the mix (what K1 reads for NCO channels)
    uint32_t nco_stride;
    uint32_t nco_phase_stride;       // padded off the power of two: all channels write the same column at the same time
    // launch geometry (host-computed maxima over the batch)
    uint32_t max_tiles;              // K1 grid.x for this call
    uint32_t max_symbols;            // upper bound of any channel's symbol count this call (k3_quantize grid)
    uint32_t t1_max, t2_max;         // size K1's LDS
    uint32_t dc_hx_cap, dc_l_cap;    // DC state geometry (sdrm_kernels.h, K2)
    uint32_t dc_group, dc_rpitch;    // channels per DC workgroup, floats between two of its delay rings in LDS
    uint32_t dc_lds;                 // dynamic LDS of the DC kernel
    int any_dc;
    int k3_carried_max;              // most samples a channel can carry between calls of the clock stage (sizes its ring)
    int quad_general;                // 1: the front-end's discriminator always takes its general form.  Nothing sets it any more (the
                                     //   SDRM_K1_QUAD switch went in round 6); it stays a launch parameter because the choice made at run time
                                     //   keeps the two forms apart in the compiler's eyes and the kernel at 118 registers -- 126 when it can
                                     //   prove the short form's table is always there, and four waves of 128 registers fill a SIMD: the
                                     //   front-end then runs 6 % slower beside the other stages (0.494 vs 0.465 ms, profiles/r06_ab.txt)
    uint32_t *placed;                // [2] DC / clock-stage workgroups that have started, cumulative over calls (k_hold_until)
    uint32_t *k3_done;               // clock-stage workgroups finished so far (all launches; the companion grid watches); nullptr: nobody is watching
    const int *gen_list;             // generic channels (sdrm_kernels.h): their indices, n_gen of them, and each one's state
    int n_gen;
    float *const *gen_state;         // [C] device pointers (null for the others)
    int k3_lanes, k3_ring, k3_plain; // clock-stage workgroup shape chosen for this batch (0: by channel count; SDRM_K3_LANES overrides both)
    uint32_t *counters;              // [16] batch-lifetime device counters: [0] channel-calls run by sdrm_k3_rescue (sdrm_batch_wild_calls),
                                     //   [1] in-call hand-off waits that ran into their bound (the call's results are void: its clock stage answers
                                     //   with the count SDRM_OUT_LEN_FAILED)
    // In-call hand-off (DESIGN.md "stages of one call overlap"): the three stages of ONE call are resident together, the DC
    // blocker starts on a channel's first finished front-end tiles, the clock stage on the first DC blocks.
    int handoff;                     // 0: stages ordered by stream events (each kernel finds its input complete)
    uint32_t epoch;                  // this call's stamp in hand_tiles / hand_prog (never 0)
    uint32_t *hand_tiles;            // [C][hand_tiles_cap]: == epoch once that front-end tile's outputs are in memory
    uint32_t hand_tiles_cap;
    unsigned long long *hand_prog;   // [C]: epoch << 32 | DC-blocker outputs of this call that are in memory
};
#define SDRM_OUT_LEN_FAILED 0xffffffffu  // == SDRM_COUNT_VOID of the public header
#define SDRM_HAND_MAX_LOOKS (1 << 21)  // bounded waits: ~2 s of looks 1 us apart, then the call fails loudly instead of hanging the device

// one kernel launch, described: what launch_* puts on a stream and what the explicitly built graph of the one-channel
// blocking call holds as a node (func == nullptr: nothing to launch this call)
struct KernelLaunch {
    const void *func = nullptr;
    dim3 grid = dim3(1), block = dim3(1);
    size_t lds = 0;
};
KernelLaunch describe_front(const DeviceBatch &b);      // args: DeviceBatch, const sdrm_f2 *d_in, size_t in_stride
KernelLaunch describe_dc(const DeviceBatch &b);         // args: DeviceBatch
KernelLaunch describe_clock(const DeviceBatch &b);      // args: DeviceBatch
KernelLaunch describe_quantize(const DeviceBatch &b);   // args: DeviceBatch (nothing to launch for the shapes that convert inside the clock stage)

size_t k1_lds_bytes(uint32_t t1_max, uint32_t t2_max);
bool front_waits_for_clock_start(int n_channels);
sdrm_k3_shape describe_shape(const DeviceBatch &b);  // the clock-stage shape the next launch takes
bool front_hold_is_forced();      // SDRM_FRONT_HOLD is set: measurements, not to be re-decided by the batch's calibration
bool k3_shape_is_forced();        // SDRM_K3_LANES likewise
void launch_hold_until(const uint32_t *counter, uint32_t target, int max_us, hipStream_t s);
unsigned dc_workgroups(const DeviceBatch &b);

void launch_nco_phase(const DeviceBatch &b, hipStream_t s);
void launch_nco_mix(const DeviceBatch &b, const sdrm_f2 *d_in, size_t in_stride, uint32_t max_len, hipStream_t s);
void launch_front(const DeviceBatch &b, const sdrm_f2 *d_in, size_t in_stride, hipStream_t s);
void launch_dc(const DeviceBatch &b, hipStream_t s);
void launch_dc_generic(const DeviceBatch &b, hipStream_t s);     // behind launch_dc, same stream (nothing without generic channels)
void launch_clock_generic(const DeviceBatch &b, hipStream_t s);  // behind launch_clock, same stream
// diagnostics buffer layout: 4 words per clock-stage workgroup (room for the smallest shape, 16 channels each), then 8
// words of the front-end, then 10 of the DC blocker
#define SDRM_STAMP_K3_WAVES(n_channels) (((n_channels) + 15) / 16)
void launch_clock(const DeviceBatch &b, hipStream_t s);
void launch_clock_company(const DeviceBatch &b, uint32_t target, int blocks, int max_rounds, int nops, hipStream_t s);
unsigned clock_workgroups(const DeviceBatch &b);
bool clock_shape_hands_off(const DeviceBatch &b);  // the in-call hand-off build of the clock stage exists for this batch's shape

// test probes
void launch_probe_boxcar_div(const float *d_sums, uint32_t length, float *d_out, size_t n, hipStream_t s);
void launch_probe_quad(const sdrm_f2 *d_y, size_t n, float gain, const float *d_tab, float *d_out, uint32_t *d_fast, hipStream_t s);
void launch_probe_atan2(const float *d_y, const float *d_x, const float *d_tab, float *d_out, size_t n, hipStream_t s);

}  // namespace sdrm

#endif
