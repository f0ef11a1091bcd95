// node.h -- the in-process front door of a multi-GPU node (SURVEY.md 8e in the reference's own process model).
//
// sdr-modem is ONE process: tcp_server.c:659 creates a dsp_worker per RX client, sdr_worker.c:25-55 feeds every worker of
// an SDR source from that source's thread, dsp_worker.c:188 starts the client's DSP thread.  Nothing in that model knows
// about devices.  A node owns one batcher (host/batcher.h: the queue + worker surface of many clients in front of one
// device batch) per GPU -- or several per GPU -- and PLACES each new client on one of them:
//   * cost of a client = fs x (4 T1 + 2 T2 / d), the front-end's multiply-adds per second of signal: the stage that bounds
//     a full GPU (the same function as sdr-modem_amd/shard.py channel_cost, which cuts a known table across ranks);
//   * least-loaded device first; a client of a source that already has clients on a device stays with them while that
//     device is within a few per cent of the least loaded one (a source's channels together);
//   * slots are recycled: a client that leaves frees its slot for the next one, on whichever device that turns out to be;
//   * a device whose batcher has failed (sdrm_batcher_error, sticky) takes no new clients; its own clients end through the
//     batcher's error path, the other devices' clients never notice.
// No collective: one process, so every device gets its table by plain host-to-device copies (last sentence).
//
// The batchers come from a factory so that the CPU test-suite can run this file over the kernel emulation with virtual
// devices (tests/emu); the shipped library only ever uses sdrm_batcher_create (HIP devices, no CPU path).
#ifndef SDRM_NODE_H
#define SDRM_NODE_H

#include <stddef.h>
#include <stdint.h>

#include <mutex>
#include <vector>

#include "../../include/sdrmodem_hip.h"

namespace sdrm {

typedef int (*batcher_factory)(void *user, int device, const sdrm_fsk_config *cfgs, size_t n, const sdrm_batcher_config *cfg,
                               sdrm_batcher **out);
typedef void (*batcher_deleter)(sdrm_batcher *b);

// fs x (4 T1 + 2 T2 / d) with the reference's tap-count rule (lpf_taps.c:33-40) for the two filters fsk_demod_create
// designs (fsk_demod.c:36-47)
double channel_cost(const sdrm_fsk_config &cfg);

class Node {
public:
    Node(batcher_factory make, batcher_deleter destroy, void *user) : make_(make), destroy_(destroy), user_(user) {}
    ~Node();
    int init(const sdrm_node_config &cfg, int visible_devices);
    int attach(const sdrm_fsk_config &client, uint64_t source_id, sdrm_node_slot *slot);
    int detach(const sdrm_node_slot &slot);
    size_t batchers() const { return dev_.size(); }
    int stat(size_t index, sdrm_node_stat *out) const;

private:
    struct Device {
        int device = 0;
        sdrm_batcher *batcher = nullptr;
        std::vector<uint8_t> used;
        std::vector<double> cost;      // cost of the client in each used slot
        std::vector<uint64_t> source;  // its source id
        std::vector<uint8_t> is_slow;  // it runs the generic DC / clock stages (channel_is_slow)
        size_t slow = 0;               // such clients here
        double load = 0.0;
        size_t clients = 0;
        uint64_t attached = 0;         // clients placed here since the node was created
    };
    batcher_factory make_;
    batcher_deleter destroy_;
    void *user_;
    std::vector<Device> dev_;
    mutable std::mutex m_;
};

}  // namespace sdrm

#endif
