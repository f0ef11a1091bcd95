// batcher_hip.cpp -- the one backend the shipped library has for the batcher: a sdrm_batch on a HIP device, driven
// through its pipelined host-buffer calls (sdrm_batch_arena / _submit / _collect).  No device, no batcher.
#include <stdlib.h>
#include <errno.h>
#include <stdio.h>

#include <vector>

#include "batcher.h"

namespace {

struct HipBackend : sdrm::BatchBackend {
    sdrm_batch *batch = nullptr;
    std::vector<uint32_t> maxlen;
    ~HipBackend() override { sdrm_batch_destroy(batch); }
    size_t channels() const override { return maxlen.size(); }
    uint32_t max_len(size_t c) const override { return maxlen[c]; }
    int arena(size_t slots, sdrm_cf32 **base, size_t *cs, size_t *ss) override {
        return sdrm_batch_arena(batch, slots, base, cs, ss);
    }
    int submit(size_t slot, const size_t *lens, const sdrm_nco_segment *segs, size_t n) override {
        return sdrm_batch_submit(batch, slot, lens, segs, n);
    }
    int collect(int8_t **outs, size_t *lens) override { return sdrm_batch_collect(batch, outs, lens); }
    int reset_channel(size_t c, const sdrm_fsk_config *cfg, int64_t pre_offset_hz) override {
        int code = sdrm_batch_reset_channel(batch, c, cfg);
        if (code == 0 && cfg != nullptr) {
            maxlen[c] = cfg->max_input_buffer_length;
        }
        if (code == 0 && pre_offset_hz != 0) {
            code = sdrm_batch_set_pre_offset(batch, c, pre_offset_hz);
        }
        return code;
    }
};

size_t plan_with_doppler(void *planner, uint32_t channel, size_t len, sdrm_nco_segment *segs, size_t cap) {
    return sdrm_doppler_plan(static_cast<sdrm_doppler *>(planner), channel, len, segs, cap);
}

}  // namespace

extern "C" int sdrm_batcher_create(const sdrm_fsk_config *cfgs, size_t n_channels, int device, const sdrm_batcher_config *cfg,
                                   sdrm_batcher **out) {
    if (cfgs == nullptr || n_channels == 0 || out == nullptr) {
        return -1;
    }
    std::unique_ptr<HipBackend> be(new HipBackend());
    // no creation-time calibration: the slots are placeholders, the real clients arrive one at a time with parameters of their own
    // (profiles/r05_node_schedule.txt: rules + online refinement tie with the calibrated schedule on a server's load)
    // (SDRM_BATCHER_CALIBRATE=1: measurements and the soak's regression -- the calibrated batch behind a batcher)
    const char *cal = getenv("SDRM_BATCHER_CALIBRATE");
    int code = sdrm_batch_create(cfgs, n_channels, device, cal != nullptr && atoi(cal) != 0 ? 0u : SDRM_FLAG_NO_CALIBRATION, &be->batch);
    if (code != 0) {
        return code;  // -ENODEV without a HIP device: there is no CPU path
    }
    for (size_t c = 0; c < n_channels; c++) {
        be->maxlen.push_back(cfgs[c].max_input_buffer_length);
    }
    const uint32_t slots = cfg ? cfg->slots : 4;
    const uint32_t wait_us = cfg ? cfg->max_wait_us : 2000;
    const bool blocking = cfg ? cfg->blocking : true;
    sdrm::Batcher *b = new sdrm::Batcher(std::move(be), slots, wait_us, blocking);
    code = b->init();
    if (code != 0) {
        delete b;
        return code;
    }
    *out = reinterpret_cast<sdrm_batcher *>(b);
    return 0;
}

extern "C" int sdrm_batcher_set_doppler(sdrm_batcher *b, size_t channel, sdrm_doppler *planner) {
    if (b == nullptr) {
        return -1;
    }
    reinterpret_cast<sdrm::Batcher *>(b)->set_doppler(channel, planner ? plan_with_doppler : nullptr, planner);
    return 0;
}
