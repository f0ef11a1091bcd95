// batcher.h -- per-GPU batcher behind many per-client DSP threads.
//
// The reference runs one queue + one DSP thread per client and calls fsk_demod_process once per buffer
// (src/dsp_worker.c:44-106, src/queue.c).  On a GPU one launch per client buffer wastes the device: the clock-recovery
// stage costs the same for 1 and for 1000 channels.  The batcher keeps the per-client surface (a producer thread puts
// IQ buffers, a consumer thread takes soft bits, in order, with the queue's blocking / overwrite-newest / poison-pill
// behaviour) and funnels everything into ONE batched call per round: producers copy straight into a slot of the pinned
// input arena, a round is launched when every open channel has delivered (or a deadline passes), up to three rounds are
// in flight on the device, and results are handed back per channel.
//
// The device side sits behind BatchBackend so that the host logic can be exercised on a machine without a GPU by the
// test-suite's kernel emulation; the shipped library only ever instantiates the HIP backend (sdrm_batcher_create).
#ifndef SDRM_BATCHER_H
#define SDRM_BATCHER_H

#include <stddef.h>
#include <stdint.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/sdrmodem_hip.h"

namespace sdrm {

struct BatchBackend {
    virtual ~BatchBackend() {}
    virtual size_t channels() const = 0;
    virtual uint32_t max_len(size_t channel) const = 0;
    // same contracts as sdrm_batch_arena / sdrm_batch_submit / sdrm_batch_collect
    virtual int arena(size_t slots, sdrm_cf32 **base, size_t *chan_stride, size_t *slot_stride) = 0;
    virtual int submit(size_t slot, const size_t *lens, const sdrm_nco_segment *segs, size_t n_segs) = 0;
    virtual int collect(int8_t **outputs, size_t *lens) = 0;
    // same contract as sdrm_batch_reset_channel (nothing is in flight when the batcher calls it), then sdrm_batch_set_pre_offset
    // when pre_offset_hz != 0 (the new client's file-source offset)
    virtual int reset_channel(size_t channel, const sdrm_fsk_config *cfg, int64_t pre_offset_hz) = 0;
};

typedef size_t (*doppler_plan_fn)(void *planner, uint32_t channel, size_t input_len, sdrm_nco_segment *segments, size_t cap);

class Batcher {
public:
    Batcher(std::unique_ptr<BatchBackend> backend, uint32_t slots, uint32_t max_wait_us, bool blocking);
    ~Batcher();
    int init();  // pins the arena, starts the batcher thread

    void put(size_t channel, const sdrm_cf32 *buf, size_t len);
    void take(size_t channel, int8_t **out, size_t *len);
    void complete(size_t channel);
    void interrupt(size_t channel);
    // the channel's consumer is gone (socket or disk error, client torn down): close the channel as interrupt() does
    // and throw away everything of it that nobody will take any more, including what the device has not finished yet,
    // so that its rounds retire and the other channels keep moving (a dead consumer in the reference only fills its own queue)
    void abandon(size_t channel);
    void set_doppler(size_t channel, doppler_plan_fn fn, void *planner);
    // hand the channel to a new client: waits until everything put on it has been consumed, drains the device, resets
    // the channel (cfg may be NULL = same configuration) and reopens it after a poison pill
    int reset_channel(size_t channel, const sdrm_fsk_config *cfg, int64_t pre_offset_hz = 0);
    // first device failure (submit or collect), 0 while the device answers.  Sticky: from then on nothing is delivered any
    // more -- take() returns NULL like after a poison pill, put() drops -- and the owners of the channels read the code
    // here (sdrm_batcher_error) to tell a dead device from a client that left.
    int error() const { return error_.load(std::memory_order_acquire); }
    size_t channels() const { return n_; }
    uint64_t rounds_launched() const { return launched_; }

private:
    void drop_done_locked(size_t channel);
    enum State { FREE, FILLING, SUBMITTED, DONE };
    struct Round {
        uint64_t id = 0;
        State state = FREE;
        std::vector<size_t> len;
        std::vector<uint8_t> has;
        size_t contributed = 0, unconsumed = 0;
        int writers = 0;
        std::chrono::steady_clock::time_point first;
        std::vector<int8_t> out;       // results, channel c at out_off_[c]
        std::vector<size_t> out_len;
    };
    Round &round(uint64_t r) { return rounds_[r % rounds_.size()]; }
    void run();
    void retire_locked();
    bool launchable_locked(const Round &rd, std::chrono::steady_clock::time_point now) const;

    std::unique_ptr<BatchBackend> be_;
    size_t n_ = 0;
    uint32_t max_wait_us_;
    bool blocking_;
    sdrm_cf32 *arena_ = nullptr;
    size_t chan_stride_ = 0, slot_stride_ = 0;
    std::vector<Round> rounds_;
    std::vector<size_t> out_off_;
    uint64_t fill_base_ = 0;    // oldest round not yet submitted
    uint64_t retire_base_ = 0;  // oldest round whose slot is not free yet
    std::deque<uint64_t> inflight_;
    std::vector<uint64_t> next_put_;
    std::vector<std::deque<uint64_t>> mine_;  // rounds holding an unconsumed buffer of the channel, oldest first
    std::vector<uint8_t> closed_;
    std::vector<uint8_t> abandoned_;  // closed, and its undelivered results are discarded as they arrive
    size_t open_ = 0;
    struct Doppler {
        doppler_plan_fn fn = nullptr;
        void *planner = nullptr;
    };
    std::vector<Doppler> doppler_;
    struct Reset {
        size_t channel;
        bool has_cfg;
        sdrm_fsk_config cfg;
        int64_t pre_offset_hz;
        bool done;
        int code;
    };
    std::deque<Reset *> resets_;  // pending channel resets, executed by the batcher thread with the device idle
    int reset_waiters_ = 0;       // resets waiting for their channel to run dry (complete() wakes them)
    size_t out_cap_ = 0;          // result bytes reserved per channel and round (the batch's largest buffer)
    bool stopping_ = false;
    std::atomic<int> error_{0};
    void fail_locked(int code);   // record the first device failure, close every channel, wake everybody
    uint64_t launched_ = 0;
    std::mutex m_;
    std::condition_variable cv_work_, cv_space_, cv_result_;
    std::thread thread_;
    bool started_ = false;
};

}  // namespace sdrm

#endif
