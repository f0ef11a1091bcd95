// emu_batcher.cpp -- TEST INFRASTRUCTURE: the product's batcher host logic (sdr-modem_amd/host/batcher.cpp) on top of
// the kernel emulation, so that the CPU-only suite can exercise rounds, ordering, back-pressure, overwrite-newest and
// the poison pill without a GPU.  The product library has no such backend.
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <deque>
#include <vector>

#include "../../sdr-modem_amd/host/batcher.h"
#include "../../sdr-modem_amd/host/node.h"

struct EmuBatch;
extern "C" int emu_create(const sdrm_fsk_config *cfgs, size_t n, EmuBatch **out);
extern "C" void emu_destroy(EmuBatch *b);
extern "C" int emu_reset_channel(EmuBatch *b, size_t c, const sdrm_fsk_config *cfg);
extern "C" int emu_set_pre_offset(EmuBatch *b, size_t c, int64_t freq_hz);
extern "C" int emu_process(EmuBatch *b, const float *const *inputs, const size_t *lens, const int8_t **out8,
                           const float **outf, size_t *outlens);
extern "C" int emu_process_nco(EmuBatch *b, const float *const *inputs, const size_t *lens, const sdrm_nco_segment *segs,
                               size_t n_segs, const int8_t **out8, const float **outf, size_t *outlens);

#include <atomic>

namespace {

// fault injection: the n-th submit / collect from now fails with -EIO (0: never)
std::atomic<int> g_fail_submit_in{0}, g_fail_collect_in{0};
bool countdown(std::atomic<int> &n) {
    int v = n.load();
    while (v > 0 && !n.compare_exchange_weak(v, v - 1)) {
    }
    return v == 1;
}

// per virtual device (emu_node_create): the n-th submit from now on that device fails with -EIO (0: never)
std::atomic<int> g_device_fail_in[16];

struct EmuBackend : sdrm::BatchBackend {
    EmuBatch *emu = nullptr;
    int virtual_device = -1;
    std::vector<uint32_t> maxlen;
    std::vector<float> arena_mem;
    size_t stride = 0, slots = 0;
    unsigned delay_us = 0;
    struct Result {
        std::vector<std::vector<int8_t>> out;
    };
    std::deque<Result> done;
    ~EmuBackend() override { emu_destroy(emu); }
    size_t channels() const override { return maxlen.size(); }
    uint32_t max_len(size_t c) const override { return maxlen[c]; }
    int arena(size_t n_slots, sdrm_cf32 **base, size_t *cs, size_t *ss) override {
        for (uint32_t m : maxlen) stride = stride > m ? stride : m;
        slots = n_slots;
        arena_mem.assign(2 * slots * maxlen.size() * stride, 0.0f);
        *base = reinterpret_cast<sdrm_cf32 *>(arena_mem.data());
        *cs = stride;
        *ss = maxlen.size() * stride;
        return 0;
    }
    int submit(size_t slot, const size_t *lens, const sdrm_nco_segment *segs, size_t n_segs) override {
        if (done.size() >= 3) return -11;
        if (countdown(g_fail_submit_in)) return -5;
        if (virtual_device >= 0 && virtual_device < 16 && countdown(g_device_fail_in[virtual_device])) return -5;
        const size_t C = maxlen.size();
        std::vector<const float *> ins(C);
        for (size_t c = 0; c < C; c++) ins[c] = arena_mem.data() + 2 * ((slot * C + c) * stride);
        std::vector<const int8_t *> o8(C);
        std::vector<const float *> of(C);
        std::vector<size_t> ol(C);
        int code = emu_process_nco(emu, ins.data(), lens, segs, n_segs, o8.data(), of.data(), ol.data());
        if (code != 0) return code;
        Result r;
        r.out.resize(C);
        for (size_t c = 0; c < C; c++) r.out[c].assign(o8[c], o8[c] + ol[c]);
        done.push_back(std::move(r));
        return 0;
    }
    int reset_channel(size_t c, const sdrm_fsk_config *cfg, int64_t pre_offset_hz) override {
        int code = emu_reset_channel(emu, c, cfg);
        if (code == 0 && cfg != nullptr) maxlen[c] = cfg->max_input_buffer_length;
        if (code == 0 && pre_offset_hz != 0) code = emu_set_pre_offset(emu, c, pre_offset_hz);
        return code;
    }
    std::vector<std::vector<int8_t>> last;
    int collect(int8_t **outs, size_t *lens) override {
        if (done.empty()) return -1;
        if (delay_us) usleep(delay_us);
        if (countdown(g_fail_collect_in)) {
            done.pop_front();
            return -5;
        }
        last = std::move(done.front().out);
        done.pop_front();
        for (size_t c = 0; c < last.size(); c++) {
            outs[c] = last[c].data();
            lens[c] = last[c].size();
        }
        return 0;
    }
};

}  // namespace

extern "C" void emu_batcher_inject(int submit_in, int collect_in) {
    g_fail_submit_in = submit_in;
    g_fail_collect_in = collect_in;
}

static int emu_batcher_make(const sdrm_fsk_config *cfgs, size_t n, uint32_t slots, uint32_t max_wait_us, int blocking,
                            unsigned device_delay_us, int virtual_device, sdrm_batcher **out) {
    std::unique_ptr<EmuBackend> be(new EmuBackend());
    int code = emu_create(cfgs, n, &be->emu);
    if (code != 0) return code;
    for (size_t c = 0; c < n; c++) be->maxlen.push_back(cfgs[c].max_input_buffer_length);
    be->delay_us = device_delay_us;
    be->virtual_device = virtual_device;
    sdrm::Batcher *b = new sdrm::Batcher(std::move(be), slots, max_wait_us, blocking != 0);
    code = b->init();
    if (code != 0) {
        delete b;
        return code;
    }
    *out = reinterpret_cast<sdrm_batcher *>(b);
    return 0;
}

extern "C" int emu_batcher_create(const sdrm_fsk_config *cfgs, size_t n, uint32_t slots, uint32_t max_wait_us, int blocking,
                                  unsigned device_delay_us, sdrm_batcher **out) {
    return emu_batcher_make(cfgs, n, slots, max_wait_us, blocking, device_delay_us, -1, out);
}

// ---- the product's node front door (sdr-modem_amd/host/node.cpp) over VIRTUAL devices: every "device" is an emulation-backed
// batcher of its own; emu_node_fail_device makes one of them fail its n-th submit from now
static int emu_node_factory(void *, int device, const sdrm_fsk_config *cfgs, size_t n, const sdrm_batcher_config *cfg,
                            sdrm_batcher **out) {
    return emu_batcher_make(cfgs, n, cfg->slots, cfg->max_wait_us, cfg->blocking ? 1 : 0, 0, device, out);
}

extern "C" int emu_node_create(const sdrm_node_config *config, int virtual_devices, sdrm_node **node) {
    for (auto &f : g_device_fail_in) f = 0;
    sdrm::Node *n = new sdrm::Node(emu_node_factory, sdrm_batcher_destroy, nullptr);
    const int code = n->init(*config, virtual_devices);
    if (code != 0) {
        delete n;
        return code;
    }
    *node = reinterpret_cast<sdrm_node *>(n);
    return 0;
}

extern "C" void emu_node_fail_device(int virtual_device, int submit_in) {
    if (virtual_device >= 0 && virtual_device < 16) g_device_fail_in[virtual_device] = submit_in;
}
