// host_stress.cpp -- TEST INFRASTRUCTURE: a threaded stress of the host-side code of the hot path's push/pull surface
// (sdr-modem_amd/host/queue.c and host/batcher.cpp over the kernel emulation), built and run under AddressSanitizer +
// UndefinedBehaviorSanitizer and under ThreadSanitizer by tests/san/run.sh (CPU build only: the GPU pool has no sanitizer
// support).  What the reference does with valgrind memcheck (test/resources/run_tests.sh:10).  Exit code 0 = clean run.
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <thread>
#include <vector>

#include "../../include/sdrmodem_hip.h"

extern "C" int emu_batcher_create(const sdrm_fsk_config *cfgs, size_t n, uint32_t slots, uint32_t max_wait_us, int blocking,
                                  unsigned device_delay_us, sdrm_batcher **out);

static int failures = 0;
#define CHECK(c)                                                       \
    do {                                                               \
        if (!(c)) {                                                    \
            fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); \
            failures++;                                                \
        }                                                              \
    } while (0)

// ---- queue: blocking and live (overwrite-newest) modes, poison pill, many buffers (reference test/test_queue.c)
static void queue_round(bool blocking, int n_buffers, int slots) {
    queue *q = nullptr;
    CHECK(create_queue(256, (uint16_t) slots, blocking, &q) == 0);
    std::atomic<int> taken{0};
    std::atomic<long> sum{0};
    std::thread consumer([&] {
        for (;;) {
            sdrm_cf32 *buf = nullptr;
            size_t len = 0;
            take_buffer_for_processing(&buf, &len, q);
            if (buf == nullptr) {
                break;
            }
            CHECK(len >= 1 && len <= 256);
            sum += (long) buf[0].re;
            taken++;
            if ((taken & 7) == 0) {
                usleep(50);
            }
            complete_buffer_processing(q);
        }
    });
    std::vector<sdrm_cf32> tmp(256);
    for (int i = 0; i < n_buffers; i++) {
        for (size_t k = 0; k < tmp.size(); k++) {
            tmp[k].re = (float) i;
            tmp[k].im = (float) k;
        }
        queue_put(tmp.data(), 1 + (size_t) (i % 256), q);
    }
    interrupt_waiting_the_data(q);
    consumer.join();
    if (blocking) {
        CHECK(taken == n_buffers);  // a file source loses nothing
        CHECK(sum == (long) n_buffers * (n_buffers - 1) / 2);
    } else {
        CHECK(taken >= 1 && taken <= n_buffers);
    }
    destroy_queue(q);
}

// ---- batcher: producers and consumers of several clients, a client that dies, a slot handed to a new client
static void batcher_round(bool blocking) {
    const size_t C = 5;
    std::vector<sdrm_fsk_config> cfgs(C);
    for (size_t c = 0; c < C; c++) {
        memset(&cfgs[c], 0, sizeof(cfgs[c]));
        cfgs[c].sampling_freq = 48000;
        cfgs[c].baud_rate = (c & 1) ? 4800 : 9600;
        cfgs[c].deviation = 5000;
        cfgs[c].decimation = (c & 1) ? 2 : 1;
        cfgs[c].transition_width = 2000;
        cfgs[c].use_dc_block = (c % 3) != 0;
        cfgs[c].max_input_buffer_length = 1024;
    }
    sdrm_batcher *bt = nullptr;
    CHECK(emu_batcher_create(cfgs.data(), C, 4, 500, blocking ? 1 : 0, 20, &bt) == 0);
    const int K = 40;
    std::vector<std::thread> th;
    std::atomic<int> delivered{0};
    for (size_t c = 0; c < C; c++) {
        th.emplace_back([&, c] {  // producer
            std::vector<sdrm_cf32> buf(1024);
            for (int k = 0; k < K; k++) {
                for (size_t i = 0; i < buf.size(); i++) {
                    buf[i].re = (float) ((i * 7 + k + c) % 13) - 6.0f;
                    buf[i].im = (float) ((i * 5 + k) % 11) - 5.0f;
                }
                sdrm_batcher_put(bt, c, buf.data(), 1 + (size_t) ((k * 97 + c * 31) % 1024));
                if (c == 4 && k == 10) {
                    break;  // client 4's producer stops early; its consumer dies below
                }
            }
            sdrm_batcher_interrupt(bt, c);  // poison pill behind the last buffer (client 4: behind the few it managed to put)
        });
        th.emplace_back([&, c] {  // consumer
            int got = 0;
            for (;;) {
                int8_t *soft = nullptr;
                size_t n = 0;
                sdrm_batcher_take(bt, c, &soft, &n);
                if (soft == nullptr) {
                    break;
                }
                CHECK(n <= 1024);
                long s = 0;
                for (size_t i = 0; i < n; i++) s += soft[i];  // touch every byte (ASan)
                (void) s;
                sdrm_batcher_complete(bt, c);
                delivered++;
                if (c == 4 && ++got == 3) {
                    sdrm_batcher_abandon(bt, c);  // "socket error": leaves with results still queued
                    break;
                }
            }
        });
    }
    for (auto &t : th) t.join();
    CHECK(delivered >= 3);
    // the dead client's slot serves a new client
    CHECK(sdrm_batcher_reset_channel(bt, 4, nullptr) == 0);
    std::vector<sdrm_cf32> buf(512);
    for (size_t i = 0; i < buf.size(); i++) {
        buf[i].re = (float) (i % 5);
        buf[i].im = 1.0f;
    }
    sdrm_batcher_put(bt, 4, buf.data(), buf.size());
    int8_t *soft = nullptr;
    size_t n = 0;
    sdrm_batcher_take(bt, 4, &soft, &n);
    CHECK(soft != nullptr);
    sdrm_batcher_complete(bt, 4);
    sdrm_batcher_destroy(bt);
}

int main() {
    for (int rep = 0; rep < 3; rep++) {
        queue_round(true, 400, 4);
        queue_round(false, 400, 3);
        queue_round(true, 50, 1);
        batcher_round(true);
        batcher_round(false);
    }
    printf("host_stress: %s\n", failures ? "FAILED" : "ok");
    return failures ? 1 : 0;
}
