// sdrm_plan.h -- host-side planning shared by the C-ABI (sdrm_api.hip) and the CPU kernel emulation used by the
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
    uint32_t t1_max = 0, hist_stride = 0, z_stride = 0, out_stride = 0, in_stride = 0;
    uint32_t rx_cap = 64, rs_cap = 64;
    size_t dc_state_floats = 0;
    int any_dc = 0;
    // streaming bookkeeping that depends on input lengths only (kept on the host)
    std::vector<uint32_t> phase, parity, zbase;
};

// 0, or the error of design_channel() / -ENOTSUP for geometry the tiles cannot hold
int plan_batch(const sdrm_fsk_config *cfgs, size_t n, BatchPlan &plan);

// Fill ctl[C] for one call and advance the bookkeeping.  lens[c] > max_len prints the reference's message
// (src/dsp/fir_filter.c:147-152) and is treated as an empty input.  Returns the largest tile count.
uint32_t plan_call(BatchPlan &plan, const size_t *lens, sdrm_chunk_ctl *ctl);

}  // namespace sdrm

#endif
