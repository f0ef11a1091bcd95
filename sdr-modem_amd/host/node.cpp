// node.cpp -- client placement over the batchers of a multi-GPU node; design in node.h.
#include "node.h"

#include "../csrc/sdrm_core.h"  // SDRM_CLOCK_HCAP, SDRM_DC_MAX_LEN: the fast stages' range (sdrm_design.cpp, `generic`)

#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

namespace sdrm {

// the reference's tap-count rule, src/dsp/lpf_taps.c:33-40: (int) (53 fs / (22 tw)), made odd
static double taps_for(uint64_t fs, uint32_t transition_width) {
    const double tw = transition_width > 0 ? (double) transition_width : 1.0;
    int n = (int) (53.0 * (double) fs / (22.0 * tw));
    if ((n & 1) == 0) {
        n++;
    }
    return (double) n;
}

double channel_cost(const sdrm_fsk_config &cfg) {
    // fsk_demod.c:36-37: carson = abs(deviation) + baud / 2; LPF1's transition width = (uint32_t) (0.1f * carson)
    const double carson = fabs((double) cfg.deviation) + (double) cfg.baud_rate / 2.0;
    const uint32_t tw1 = (uint32_t) ((double) 0.1f * carson);
    const double t1 = taps_for(cfg.sampling_freq, tw1);
    const double t2 = taps_for(cfg.sampling_freq, cfg.transition_width);
    const double d = cfg.decimation > 0 ? (double) cfg.decimation : 1.0;
    return (double) cfg.sampling_freq * (4.0 * t1 + 2.0 * t2 / d);
}

// A client beyond the fast stages' range (more than ~228 samples per symbol, or a DC boxcar beyond 7712 samples: sdrm_design.cpp,
// "generic") runs the generic DC and clock stages: milliseconds per call on the DC stream, which every other client of the same
// batch then waits for.  Such clients are kept together, and away from the others while another batcher has room.
bool channel_is_slow(const sdrm_fsk_config &cfg) {
    if (cfg.baud_rate == 0 || cfg.decimation == 0) {
        return false;
    }
    const float sps = (float) ((double) cfg.sampling_freq / cfg.baud_rate / cfg.decimation);
    return sps * 1.01f + 24.0f > (float) (SDRM_CLOCK_HCAP - 1) || (cfg.use_dc_block && ceilf(sps * 32) > (float) SDRM_DC_MAX_LEN);
}

// how far above the least-loaded device's load (after placing the client) a device may be and still be preferred because
// it already serves the client's source
static const double SOURCE_AFFINITY_SLACK = 1.08;

Node::~Node() {
    for (Device &d : dev_) {
        if (d.batcher != nullptr) {
            destroy_(d.batcher);
        }
    }
}

int Node::init(const sdrm_node_config &cfg, int visible_devices) {
    size_t n = cfg.n_batchers;
    if (n == 0) {
        if (visible_devices <= 0) {
            fprintf(stderr, "<3>sdrmodem_hip: no HIP device available; this library has no CPU fallback\n");
            return -ENODEV;
        }
        n = (size_t) visible_devices;
    }
    if (cfg.slots_per_batcher == 0) {
        return -1;
    }
    std::vector<sdrm_fsk_config> table(cfg.slots_per_batcher, cfg.geometry);
    sdrm_batcher_config bc = cfg.batcher;
    if (bc.slots == 0) {
        bc.slots = 4;
        bc.max_wait_us = 2000;
        bc.blocking = true;
    }
    dev_.reserve(n);
    for (size_t i = 0; i < n; i++) {
        Device d;
        d.device = cfg.devices != nullptr ? cfg.devices[i] : (visible_devices > 0 ? (int) (i % (size_t) visible_devices) : (int) i);
        // per-device configuration "fan-out": one process, so each device gets the table by its own host-to-device copies
        const int code = make_(user_, d.device, table.data(), table.size(), &bc, &d.batcher);
        if (code != 0) {
            fprintf(stderr, "<3>sdrmodem_hip: node: batcher %zu on device %d could not be created (%d)\n", i, d.device, code);
            return code;  // the destructor releases the batchers made so far
        }
        d.used.assign(table.size(), 0);
        d.cost.assign(table.size(), 0.0);
        d.is_slow.assign(table.size(), 0);
        d.source.assign(table.size(), 0);
        // a round waits for every OPEN channel: slots without a client stay closed until the client's worker resets its
        // slot (sdrm_batcher_reset_channel reopens it), so a round goes as soon as the live clients have delivered
        for (size_t s = 0; s < table.size(); s++) {
            sdrm_batcher_abandon(d.batcher, s);
        }
        dev_.push_back(d);
    }
    return 0;
}

int Node::attach(const sdrm_fsk_config &client, uint64_t source_id, sdrm_node_slot *slot) {
    if (slot == nullptr) {
        return -1;
    }
    const double cost = channel_cost(client);
    const bool slow = channel_is_slow(client);
    std::lock_guard<std::mutex> g(m_);
    int best = -1, fellow = -1, first_error = 0;
    bool any_alive = false;
    for (size_t i = 0; i < dev_.size(); i++) {
        Device &d = dev_[i];
        const int err = sdrm_batcher_error(d.batcher);
        if (err != 0) {
            first_error = first_error ? first_error : err;
            continue;  // a failed device takes no new clients
        }
        any_alive = true;
        if (d.clients >= d.used.size()) {
            continue;  // full
        }
        // least loaded -- among the batchers of the client's kind (slow with slow, fast away from slow) while there is one
        const bool kind = slow ? d.slow > 0 : d.slow == 0;
        const bool best_kind = best >= 0 && (slow ? dev_[(size_t) best].slow > 0 : dev_[(size_t) best].slow == 0);
        if (best < 0 || (kind && !best_kind) || (kind == best_kind && d.load < dev_[(size_t) best].load)) {
            best = (int) i;
        }
        if (source_id != 0) {
            bool serves = false;
            for (size_t s = 0; s < d.used.size() && !serves; s++) {
                serves = d.used[s] && d.source[s] == source_id;
            }
            if (serves && (fellow < 0 || d.load < dev_[(size_t) fellow].load)) {
                fellow = (int) i;
            }
        }
    }
    if (best < 0) {
        if (!any_alive && first_error != 0) {
            return first_error;  // every device has failed
        }
        return -EBUSY;  // every slot of every healthy device is taken
    }
    int pick = best;
    if (fellow >= 0 && dev_[(size_t) fellow].load + cost <= (dev_[(size_t) best].load + cost) * SOURCE_AFFINITY_SLACK &&
        (dev_[(size_t) fellow].slow > 0) == (dev_[(size_t) best].slow > 0)) {
        pick = fellow;
    }
    Device &d = dev_[(size_t) pick];
    size_t s = 0;
    while (d.used[s]) {
        s++;
    }
    d.used[s] = 1;
    d.cost[s] = cost;
    d.is_slow[s] = slow ? 1 : 0;
    d.slow += slow ? 1 : 0;
    d.source[s] = source_id;
    d.load += cost;
    d.clients++;
    d.attached++;
    slot->batcher = d.batcher;
    slot->channel = s;
    slot->device = d.device;
    slot->batcher_index = (size_t) pick;
    return 0;
}

int Node::detach(const sdrm_node_slot &slot) {
    std::lock_guard<std::mutex> g(m_);
    if (slot.batcher_index >= dev_.size()) {
        return -1;
    }
    Device &d = dev_[slot.batcher_index];
    if (d.batcher != slot.batcher || slot.channel >= d.used.size() || !d.used[slot.channel]) {
        return -1;
    }
    // the client is gone: whatever it has not taken is nobody's, and the slot must not hold up the rounds of the others
    // while it waits for its next client (closed until that client's reset reopens it)
    sdrm_batcher_abandon(d.batcher, slot.channel);
    d.used[slot.channel] = 0;
    d.load -= d.cost[slot.channel];
    d.cost[slot.channel] = 0.0;
    d.slow -= d.is_slow[slot.channel];
    d.is_slow[slot.channel] = 0;
    d.source[slot.channel] = 0;
    d.clients--;
    if (d.clients == 0) {
        d.load = 0.0;  // no rounding residue from a long series of additions and subtractions
    }
    return 0;
}

int Node::stat(size_t index, sdrm_node_stat *out) const {
    std::lock_guard<std::mutex> g(m_);
    if (index >= dev_.size() || out == nullptr) {
        return -1;
    }
    const Device &d = dev_[index];
    out->device = d.device;
    out->batcher = d.batcher;
    out->slots = d.used.size();
    out->clients = d.clients;
    out->load = d.load;
    out->attached = d.attached;
    out->error = sdrm_batcher_error(d.batcher);
    return 0;
}

}  // namespace sdrm

// ---- C-ABI (include/sdrmodem_hip.h).  sdrm_node_create lives in node_hip.cpp: it is the only part that names a device API.

extern "C" double sdrm_channel_cost(const sdrm_fsk_config *config) { return config ? sdrm::channel_cost(*config) : 0.0; }

extern "C" int sdrm_node_attach(sdrm_node *node, const sdrm_fsk_config *client, uint64_t source_id, sdrm_node_slot *slot) {
    if (node == nullptr || client == nullptr) {
        return -1;
    }
    return reinterpret_cast<sdrm::Node *>(node)->attach(*client, source_id, slot);
}

extern "C" int sdrm_node_detach(sdrm_node *node, const sdrm_node_slot *slot) {
    if (node == nullptr || slot == nullptr) {
        return -1;
    }
    return reinterpret_cast<sdrm::Node *>(node)->detach(*slot);
}

extern "C" size_t sdrm_node_batchers(const sdrm_node *node) {
    return node ? reinterpret_cast<const sdrm::Node *>(node)->batchers() : 0;
}

extern "C" int sdrm_node_stat_read(const sdrm_node *node, size_t index, sdrm_node_stat *stat) {
    if (node == nullptr) {
        return -1;
    }
    return reinterpret_cast<const sdrm::Node *>(node)->stat(index, stat);
}

extern "C" void sdrm_node_destroy(sdrm_node *node) { delete reinterpret_cast<sdrm::Node *>(node); }
