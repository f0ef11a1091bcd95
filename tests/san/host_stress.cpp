// host_stress.cpp -- TEST INFRASTRUCTURE: a threaded stress of the host-side code of the hot path's push/pull surface
// (sdr-modem_amd/host/queue.c and host/batcher.cpp over the kernel emulation), built and run under AddressSanitizer +
// UndefinedBehaviorSanitizer and under ThreadSanitizer by tests/san/run.sh (CPU build only: the GPU pool has no sanitizer
// support).  What the reference does with valgrind memcheck (test/resources/run_tests.sh:10).  Exit code 0 = clean run.
#include "../../sdr-modem_amd/host/ledger.h"
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

extern "C" int emu_node_create(const sdrm_node_config *config, int virtual_devices, sdrm_node **node);
extern "C" void emu_node_fail_device(int virtual_device, int submit_in);

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

// ---- node front door (host/node.cpp) over three virtual devices: client threads attach, open their slot, stream, leave,
// come back; one device fails in the middle; placement, slot recycling and the error path run concurrently
static void node_round() {
    sdrm_node_config nc;
    memset(&nc, 0, sizeof(nc));
    nc.slots_per_batcher = 4;
    nc.geometry.sampling_freq = 48000;
    nc.geometry.baud_rate = 9600;
    nc.geometry.deviation = 5000;
    nc.geometry.decimation = 1;
    nc.geometry.transition_width = 2000;
    nc.geometry.use_dc_block = true;
    nc.geometry.max_input_buffer_length = 1024;
    nc.batcher.slots = 4;
    nc.batcher.max_wait_us = 300;
    nc.batcher.blocking = true;
    sdrm_node *node = nullptr;
    CHECK(emu_node_create(&nc, 3, &node) == 0);
    CHECK(sdrm_node_batchers(node) == 3);
    std::atomic<int> served{0}, refused{0}, ended_by_device{0};
    std::vector<std::thread> th;
    for (int t = 0; t < 10; t++) {
        th.emplace_back([&, t] {
            std::vector<sdrm_cf32> buf(1024);
            for (size_t i = 0; i < buf.size(); i++) {
                buf[i].re = (float) ((i * 3 + t) % 9) - 4.0f;
                buf[i].im = (float) ((i * 5 + t) % 7) - 3.0f;
            }
            for (int life = 0; life < 6; life++) {
                sdrm_fsk_config mine = nc.geometry;
                mine.baud_rate = (t + life) & 1 ? 4800 : 9600;
                mine.decimation = (t + life) & 1 ? 2 : 1;
                sdrm_node_slot slot;
                const int code = sdrm_node_attach(node, &mine, (uint64_t) (1 + t % 3), &slot);
                if (code != 0) {
                    refused++;  // -EBUSY: twelve slots (eight once a device is gone), ten clients: rare but legal
                    usleep(200);
                    continue;
                }
                if (sdrm_batcher_reset_channel(slot.batcher, slot.channel, &mine) != 0) {
                    CHECK(sdrm_batcher_error(slot.batcher) != 0);  // only a dead device refuses a slot it handed out
                    sdrm_node_detach(node, &slot);
                    continue;
                }
                bool alive = true;
                for (int k = 0; k < 5 && alive; k++) {
                    sdrm_batcher_put(slot.batcher, slot.channel, buf.data(), 200 + (size_t) ((k * 131 + t * 17) % 800));
                    int8_t *soft = nullptr;
                    size_t n = 0;
                    sdrm_batcher_take(slot.batcher, slot.channel, &soft, &n);
                    if (soft == nullptr) {
                        CHECK(sdrm_batcher_error(slot.batcher) != 0);
                        ended_by_device++;
                        alive = false;
                        break;
                    }
                    long s = 0;
                    for (size_t i = 0; i < n; i++) s += soft[i];
                    (void) s;
                    sdrm_batcher_complete(slot.batcher, slot.channel);
                }
                if (alive) {
                    sdrm_batcher_interrupt(slot.batcher, slot.channel);
                    served++;
                }
                CHECK(sdrm_node_detach(node, &slot) == 0);
                if (t == 0 && life == 2) {
                    emu_node_fail_device(1, 2);  // virtual device 1 fails its second call from now
                }
            }
        });
    }
    for (auto &t : th) t.join();
    CHECK(served > 20);
    size_t clients = 0;
    int dead = 0;
    for (size_t i = 0; i < 3; i++) {
        sdrm_node_stat st;
        CHECK(sdrm_node_stat_read(node, i, &st) == 0);
        clients += st.clients;
        dead += st.error != 0;
    }
    CHECK(clients == 0);
    CHECK(dead <= 1);
    sdrm_node_destroy(node);
}

// The device-wide ledger of waiting hand-off workgroups (sdr-modem_amd/host/ledger.cpp): 24 "batches" on their own threads ask for
// admission, arm their entry, let their "event" fire a little later (a flag another thread of the pair sets) and release -- some
// release themselves, some go quiet and are reaped by somebody else's admission, some are plain handles with a blocking call in
// flight.  Invariants: the workgroups listed at once never exceed the limit, every attempt is counted once, nothing is left behind.
static bool flag_fired(void *event) { return static_cast<std::atomic<bool> *>(event)->load(std::memory_order_acquire); }
static void ledger_round() {
    sdrm::WaitLedger ledger;
    const unsigned limit = 40;
    const int n_threads = 24, rounds = 400;
    std::atomic<int> listed{0};      // workgroups of admitted calls whose event has not fired yet (what really waits)
    std::atomic<long> attempts{0}, admitted{0};
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++) {
        th.emplace_back([&, t] {
            std::atomic<bool> event{false};
            int owner_tag = t;
            const bool plain = t % 3 == 0;
            const unsigned mine = plain ? 2u : 5u + (unsigned) (t % 7);
            unsigned seed = 99u + (unsigned) t;
            for (int r = 0; r < rounds; r++) {
                seed = seed * 1664525u + 1013904223u;
                if (plain) ledger.plain_begin();
                attempts++;
                event.store(false, std::memory_order_release);
                if (ledger.admit(&owner_tag, &event, mine, limit, plain, flag_fired)) {
                    admitted++;
                    const int now = listed.fetch_add((int) mine) + (int) mine;
                    CHECK(now <= (int) limit);
                    ledger.arm(&owner_tag);
                    if ((seed >> 8) % 4 == 0) std::this_thread::yield();
                    listed.fetch_sub((int) mine);                      // the call ends: its workgroups are gone BEFORE the event says so
                    event.store(true, std::memory_order_release);
                    if ((seed >> 12) % 3 != 0) ledger.release(&owner_tag);  // ... or the owner goes quiet: somebody else reaps the entry
                }
                if (plain) ledger.plain_end();
            }
            ledger.release(&owner_tag);  // going away: the event is about to be destroyed
        });
    }
    for (auto &t : th) t.join();
    uint64_t taken = 0, refused = 0;
    uint32_t peak = 0;
    ledger.stats(&taken, &refused, &peak);
    CHECK((long) (taken + refused) == attempts.load());
    CHECK((long) taken == admitted.load());
    CHECK(peak <= limit);
    CHECK(taken > 0 && refused > 0);
    // everything has fired or been released: a caller that wants the whole budget gets it
    std::atomic<bool> event{false};
    int tag = -1;
    CHECK(ledger.admit(&tag, &event, limit, limit, false, flag_fired));
    ledger.release(&tag);
}

int main() {
    for (int rep = 0; rep < 3; rep++) {
        ledger_round();
        queue_round(true, 400, 4);
        queue_round(false, 400, 3);
        queue_round(true, 50, 1);
        batcher_round(true);
        batcher_round(false);
        node_round();
    }
    printf("host_stress: %s\n", failures ? "FAILED" : "ok");
    return failures ? 1 : 0;
}
