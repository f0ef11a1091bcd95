// sdrm_plan.h -- host-side planning shared by the C-ABI (sdrm_batch.hip, sdrm_call.hip) and the CPU kernel emulation used by the
// CPU-only tests: packs the per-channel device parameters from the designs, and turns the per-channel input
// lengths of one call into the control blocks the kernels consume (decimation phase, tile counts, history parity).
// Plain C++, no HIP.
#ifndef SDRM_PLAN_H
#define SDRM_PLAN_H

#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "sdrm_design.h"
#include "sdrm_kernels.h"

namespace sdrm {

struct BatchPlan {
    std::vector<ChannelDesign> design;
    std::vector<sdrm_chan_params> params;
    std::vector<float> tap_pool;  // reversed taps of every distinct filter, 8-float aligned
    // geometry shared by the whole batch
    uint32_t t1_max = 0, t2_max = 0, hist_stride = 0, z_stride = 0, out_stride = 0, in_stride = 0;
    uint32_t dc_l_cap = 0, dc_hx_cap = 0;  // longest boxcar of the batch and its 2(L-1) carried samples: size every channel's DC state
    uint32_t dc_group = SDRM_K2_SLOTS;     // channels per DC workgroup: 16 unless the delay rings of that many would not fit in LDS
    size_t dc_state_floats = 0, dc_region_floats = 0;  // every channel owns a region of the largest size
    size_t private_taps_base = 0, private_taps_slot = 0;  // per-channel tap slots behind the shared pool (replan_channel)
    int any_dc = 0;
    // streaming bookkeeping that depends on input lengths only (kept on the host)
    std::vector<uint32_t> phase, parity, zbase;
    uint32_t clock_carried_max = 0;  // most samples a channel can carry between calls of the clock stage (< 1.01 sps + 8)
    size_t dc_lds_bytes() const;  // dynamic LDS of the DC kernel for this batch
};

// 0, or the error of design_channel() / -ENOTSUP for geometry the tiles cannot hold
int plan_batch(const sdrm_fsk_config *cfgs, size_t n, BatchPlan &plan);

// Give channel c a new configuration (same batch geometry: filters, DC length and buffer size no larger than what the
// batch was created for).  Fills `taps_slot` with the channel's private copy of both filters' reversed taps (to be put at
// tap_pool offset params[c].taps1_off) and resets the channel's streaming bookkeeping.  0, a design error, or -ENOTSUP.
int replan_channel(BatchPlan &plan, size_t c, const sdrm_fsk_config &cfg, std::vector<float> &taps_slot);

// The geometry a batch needs once channel c takes `cfg`: what of {longest filters, history, DC boxcar} would have to grow.
// The buffer length never grows (the input / output buffers and the pinned arena were handed out at that size).
struct GeometryGrowth {
    bool needed = false;
    uint32_t t1_max = 0, t2_max = 0, hist_stride = 0, dc_l_cap = 0;
    int any_dc = 0;
};
// 0 and `g` filled (g.needed false: the configuration fits as it is), a design error, or -ENOTSUP (buffer too long, or the
// grown DC blocker would not fit the device)
int plan_growth(const BatchPlan &plan, const sdrm_fsk_config &cfg, GeometryGrowth &g);
// Apply a growth: new maxima, strides, per-channel DC regions and private tap slots (every channel that lives in a private
// slot gets new tap offsets; `moved` lists them).  Device buffers are the caller's to re-lay-out.
void apply_growth(BatchPlan &plan, const GeometryGrowth &g, std::vector<size_t> &moved);

// Fill ctl[C] for one call and advance the bookkeeping.  lens[c] > max_len prints the reference's message
// (src/dsp/fir_filter.c:147-152) and is treated as an empty input.  Returns the largest tile count.
uint32_t plan_call(BatchPlan &plan, const size_t *lens, sdrm_chunk_ctl *ctl);

// Turn the caller's NCO segments (grouped by channel, consecutive, covering the channel's whole input of this call)
// into the device table and fill ctl[c].nco_off / nco_cnt.  Channels without segments get no NCO.  Returns 0, or -1
// when a channel's segments do not add up to its input length / are not grouped.
int plan_nco(const BatchPlan &plan, const sdrm_nco_segment *segs, size_t n_segs, sdrm_chunk_ctl *ctl,
             std::vector<sdrm_nco_seg> &table);

// Doppler batching of the reference (src/dsp/doppler.c:116-190) without its orbit model: the per-second shifts come
// from the caller (SGP4 stays on the host side of the boundary).
struct DopplerPlanner {
    uint64_t interval = 0, in_interval = 0;
    double cur = 0.0, next = 0.0, slope = 0.0;
    uint64_t second = 0;
    sdrm_doppler_shift_fn fn = nullptr;
    void *user = nullptr;
    size_t plan(uint32_t channel, size_t input_len, sdrm_nco_segment *out, size_t cap);
};

}  // namespace sdrm

#endif
