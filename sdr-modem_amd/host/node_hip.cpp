// node_hip.cpp -- the one way the shipped library makes a node: one batcher per HIP device (sdrm_batcher_create).
// No device, no node (-ENODEV): there is no CPU path.
#include <errno.h>
#include <stdio.h>

#include "node.h"

namespace {

int make_hip_batcher(void *, int device, const sdrm_fsk_config *cfgs, size_t n, const sdrm_batcher_config *cfg, sdrm_batcher **out) {
    return sdrm_batcher_create(cfgs, n, device, cfg, out);
}

}  // namespace

extern "C" int sdrm_node_create(const sdrm_node_config *config, sdrm_node **node) {
    if (config == nullptr || node == nullptr) {
        return -1;
    }
    const int visible = sdrm_device_count();
    if (visible <= 0) {
        fprintf(stderr, "<3>sdrmodem_hip: no HIP device available; this library has no CPU fallback\n");
        return -ENODEV;
    }
    if (config->devices != nullptr) {
        for (size_t i = 0; i < config->n_batchers; i++) {
            if (config->devices[i] < 0 || config->devices[i] >= visible) {
                fprintf(stderr, "<3>sdrmodem_hip: node: device %d out of range (%d devices)\n", config->devices[i], visible);
                return -ENODEV;
            }
        }
    }
    sdrm::Node *n = new sdrm::Node(make_hip_batcher, sdrm_batcher_destroy, nullptr);
    const int code = n->init(*config, visible);
    if (code != 0) {
        delete n;
        return code;
    }
    *node = reinterpret_cast<sdrm_node *>(n);
    return 0;
}
