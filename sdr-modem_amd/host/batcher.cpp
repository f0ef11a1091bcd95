// batcher.cpp -- see batcher.h.  Queue semantics mirrored from the reference's src/queue.c (put :99-154, take
// :168-200, complete :202-213, interrupt :215-223): a blocking producer waits while no slot is free, a live producer
// overwrites its newest buffer that has not been handed to the device yet and logs "<3>queue is full"; buffers that
// were put before the interrupt are still delivered, then the consumer gets NULL.
#include "batcher.h"

#include <stdio.h>
#include <string.h>

#include <algorithm>

namespace sdrm {

static const size_t MAX_FLIGHT = 3;  // rounds on the device at once (sdrm_batch_submit accepts as many)

Batcher::Batcher(std::unique_ptr<BatchBackend> backend, uint32_t slots, uint32_t max_wait_us, bool blocking)
    : be_(std::move(backend)), max_wait_us_(max_wait_us), blocking_(blocking) {
    n_ = be_->channels();
    rounds_.resize(std::max<uint32_t>(slots, 4));
}

int Batcher::init() {
    int code = be_->arena(rounds_.size(), &arena_, &chan_stride_, &slot_stride_);
    if (code != 0) {
        return code;
    }
    // every channel gets room for the largest buffer of the batch: a channel may be given another configuration later
    for (size_t c = 0; c < n_; c++) {
        out_cap_ = std::max<size_t>(out_cap_, be_->max_len(c));
    }
    out_off_.assign(n_ + 1, 0);
    for (size_t c = 0; c < n_; c++) {
        out_off_[c + 1] = out_off_[c] + out_cap_;
    }
    for (Round &rd : rounds_) {
        rd.len.assign(n_, 0);
        rd.has.assign(n_, 0);
        rd.out.resize(out_off_[n_]);
        rd.out_len.assign(n_, 0);
    }
    next_put_.assign(n_, 0);
    mine_.assign(n_, std::deque<uint64_t>());
    closed_.assign(n_, 0);
    abandoned_.assign(n_, 0);
    doppler_.assign(n_, Doppler());
    open_ = n_;
    thread_ = std::thread(&Batcher::run, this);
    started_ = true;
    return 0;
}

Batcher::~Batcher() {
    {
        std::lock_guard<std::mutex> g(m_);
        stopping_ = true;
    }
    cv_work_.notify_all();
    cv_space_.notify_all();
    cv_result_.notify_all();
    if (started_) {
        thread_.join();
    }
}

void Batcher::set_doppler(size_t c, doppler_plan_fn fn, void *planner) {
    std::lock_guard<std::mutex> g(m_);
    if (c < n_) {
        doppler_[c].fn = fn;
        doppler_[c].planner = planner;
    }
}

void Batcher::put(size_t c, const sdrm_cf32 *buf, size_t len) {
    if (c >= n_ || buf == nullptr) {
        return;
    }
    if (len > be_->max_len(c)) {
        // the stage guard of the reference (src/dsp/fir_filter.c:147-152): message, nothing produced
        fprintf(stderr, "<3>requested buffer %zu is more than max: %u\n", len, be_->max_len(c));
        return;
    }
    std::unique_lock<std::mutex> lk(m_);
    uint64_t r = 0;
    bool overwrite = false;
    for (;;) {
        if (closed_[c] || stopping_ || error_.load(std::memory_order_relaxed) != 0) {
            return;
        }
        r = std::max(next_put_[c], fill_base_);
        if (r - retire_base_ < rounds_.size()) {
            break;
        }
        if (blocking_) {
            cv_space_.wait(lk);
            continue;
        }
        fprintf(stderr, "<3>queue is full\n");
        // live source: the newest buffer of this channel that the device has not been given yet is replaced
        if (next_put_[c] > fill_base_) {
            r = next_put_[c] - 1;
            overwrite = true;
            break;
        }
        return;  // everything pending is already on the device: the new buffer is dropped
    }
    Round &rd = round(r);
    if (rd.state == FREE) {
        rd.state = FILLING;
        rd.id = r;
        rd.contributed = rd.unconsumed = 0;
        std::fill(rd.has.begin(), rd.has.end(), 0);
        std::fill(rd.len.begin(), rd.len.end(), 0);
    }
    rd.writers++;
    lk.unlock();
    memcpy(arena_ + (r % rounds_.size()) * slot_stride_ + c * chan_stride_, buf, len * sizeof(sdrm_cf32));
    lk.lock();
    rd.writers--;
    rd.len[c] = len;
    if (!overwrite) {
        rd.has[c] = 1;
        rd.unconsumed++;
        if (rd.contributed++ == 0) {
            rd.first = std::chrono::steady_clock::now();
        }
        mine_[c].push_back(r);
        next_put_[c] = r + 1;
    }
    lk.unlock();
    cv_work_.notify_one();
}

void Batcher::take(size_t c, int8_t **out, size_t *len) {
    *out = nullptr;
    *len = 0;
    if (c >= n_) {
        return;
    }
    std::unique_lock<std::mutex> lk(m_);
    for (;;) {
        if (abandoned_[c] || error_.load(std::memory_order_relaxed) != 0) {
            return;  // the channel was given up, or the device failed (error()): nothing is delivered any more
        }
        if (!mine_[c].empty()) {
            Round &rd = round(mine_[c].front());
            if (rd.state == DONE) {
                *out = rd.out.data() + out_off_[c];
                *len = rd.out_len[c];
                return;
            }
        } else if (closed_[c] || stopping_) {
            return;  // poison pill, nothing left to deliver
        }
        if (stopping_) {
            return;
        }
        cv_result_.wait(lk);
    }
}

// the device failed: what the rounds in flight hold is lost and no later call can be trusted (the streams' state lives
// on the device).  Every channel is closed and given up, every waiter woken; take() and put() return at once from now on.
void Batcher::fail_locked(int code) {
    int expected = 0;
    error_.compare_exchange_strong(expected, code, std::memory_order_release);
    for (size_t c = 0; c < n_; c++) {
        if (!closed_[c]) {
            closed_[c] = 1;
            open_--;
        }
        abandoned_[c] = 1;
    }
    cv_work_.notify_all();
    cv_space_.notify_all();
    cv_result_.notify_all();
}

void Batcher::retire_locked() {
    bool freed = false;
    while (retire_base_ < fill_base_) {
        Round &rd = round(retire_base_);
        if (rd.state != DONE || rd.unconsumed != 0) {
            break;
        }
        rd.state = FREE;
        retire_base_++;
        freed = true;
    }
    if (freed) {
        cv_space_.notify_all();
    }
}

void Batcher::complete(size_t c) {
    if (c >= n_) {
        return;
    }
    std::lock_guard<std::mutex> g(m_);
    if (mine_[c].empty()) {
        return;
    }
    Round &rd = round(mine_[c].front());
    if (rd.state != DONE) {
        return;
    }
    mine_[c].pop_front();
    rd.unconsumed--;
    retire_locked();
    if (mine_[c].empty() && reset_waiters_ > 0) {
        cv_result_.notify_all();  // a reset of this channel may be waiting for it to run dry
    }
}

void Batcher::interrupt(size_t c) {
    if (c >= n_) {
        return;
    }
    {
        std::lock_guard<std::mutex> g(m_);
        if (!closed_[c]) {
            closed_[c] = 1;
            open_--;
        }
    }
    cv_work_.notify_all();  // rounds no longer wait for this channel
    cv_space_.notify_all();
    cv_result_.notify_all();
}

// finished rounds at the head of an abandoned channel's list count as consumed (the batcher thread calls this again
// whenever a round finishes)
void Batcher::drop_done_locked(size_t c) {
    bool dropped = false;
    while (!mine_[c].empty() && round(mine_[c].front()).state == DONE) {
        round(mine_[c].front()).unconsumed--;
        mine_[c].pop_front();
        dropped = true;
    }
    if (dropped) {
        retire_locked();
        if (mine_[c].empty() && reset_waiters_ > 0) {
            cv_result_.notify_all();
        }
    }
}

void Batcher::abandon(size_t c) {
    if (c >= n_) {
        return;
    }
    {
        std::lock_guard<std::mutex> g(m_);
        if (!closed_[c]) {
            closed_[c] = 1;
            open_--;
        }
        abandoned_[c] = 1;
        drop_done_locked(c);
    }
    cv_work_.notify_all();
    cv_space_.notify_all();
    cv_result_.notify_all();
}

int Batcher::reset_channel(size_t c, const sdrm_fsk_config *cfg, int64_t pre_offset_hz) {
    if (c >= n_) {
        return -1;
    }
    Reset req;
    req.channel = c;
    req.has_cfg = cfg != nullptr;
    if (cfg != nullptr) {
        req.cfg = *cfg;
    }
    req.pre_offset_hz = pre_offset_hz;
    req.done = false;
    req.code = 0;
    std::unique_lock<std::mutex> lk(m_);
    // a batcher whose device has failed hands out no slots: the streams' state lived on that device (sdrm_batcher_error)
    if (error_ != 0) {
        return error_;
    }
    // what was put for the previous client is still delivered to (and has to be consumed by) its consumer
    reset_waiters_++;
    while (!mine_[c].empty() && !stopping_ && error_ == 0) {
        cv_result_.wait(lk);
    }
    reset_waiters_--;
    if (stopping_) {
        return -1;
    }
    if (error_ != 0) {
        return error_;
    }
    resets_.push_back(&req);
    cv_work_.notify_all();
    while (!req.done && !stopping_) {
        cv_result_.wait(lk);
    }
    if (req.done && req.code == 0 && error_ != 0) {
        return error_;  // the device failed while the reset was queued: run() did not reopen the channel
    }
    return req.done ? req.code : -1;
}

bool Batcher::launchable_locked(const Round &rd, std::chrono::steady_clock::time_point now) const {
    if (rd.state != FILLING || rd.writers != 0 || rd.contributed == 0) {
        return false;
    }
    size_t open_in = 0;  // open channels that have delivered their buffer for this round
    for (size_t c = 0; c < n_; c++) {
        open_in += (rd.has[c] && !closed_[c]) ? 1 : 0;
    }
    if (open_in >= open_ || stopping_) {
        return true;
    }
    return now - rd.first >= std::chrono::microseconds(max_wait_us_);
}

void Batcher::run() {
    std::vector<size_t> lens(n_);
    std::vector<int8_t *> outs(n_);
    std::vector<size_t> olens(n_);
    std::vector<sdrm_nco_segment> segs;
    sdrm_nco_segment tmp[64];
    std::unique_lock<std::mutex> lk(m_);
    for (;;) {
        const auto now = std::chrono::steady_clock::now();
        Round &rd = round(fill_base_);
        const bool fresh = rd.state == FILLING && rd.id == fill_base_;
        if (!resets_.empty() && inflight_.empty()) {
            Reset *req = resets_.front();
            resets_.pop_front();
            lk.unlock();
            const int code = be_->reset_channel(req->channel, req->has_cfg ? &req->cfg : nullptr, req->pre_offset_hz);
            lk.lock();
            if (code == 0 && error_ == 0 && closed_[req->channel]) {
                closed_[req->channel] = 0;  // the slot serves a new client
                abandoned_[req->channel] = 0;
                open_++;
            }
            req->code = code;
            req->done = true;
            cv_result_.notify_all();
            continue;
        }
        if (resets_.empty() && fresh && inflight_.size() < MAX_FLIGHT && launchable_locked(rd, now)) {
            segs.clear();
            for (size_t c = 0; c < n_; c++) {
                // a client without a buffer in this round is ABSENT from the call: an empty call would make the clock
                // stage answer from its carried samples (it re-emits a symbol when samples/symbol >= 8) and the client's
                // stream would no longer be the one it put
                lens[c] = rd.has[c] ? rd.len[c] : SDRM_LEN_ABSENT;
                if (rd.has[c] && doppler_[c].fn != nullptr && lens[c] > 0) {
                    size_t k = doppler_[c].fn(doppler_[c].planner, (uint32_t) c, lens[c], tmp, 64);
                    segs.insert(segs.end(), tmp, tmp + k);
                }
            }
            rd.state = SUBMITTED;
            const uint64_t r = fill_base_++;
            inflight_.push_back(r);
            launched_++;
            lk.unlock();
            int code = be_->submit(r % rounds_.size(), lens.data(), segs.empty() ? nullptr : segs.data(), segs.size());
            if (code != 0) {
                fprintf(stderr, "<3>batcher: device call failed: %d\n", code);
            }
            lk.lock();
            if (code != 0) {  // no results will come: end every client instead of leaving it waiting (or fed empty buffers)
                Round &bad = round(r);
                std::fill(bad.out_len.begin(), bad.out_len.end(), 0);
                bad.state = DONE;
                inflight_.erase(std::find(inflight_.begin(), inflight_.end(), r));
                fail_locked(code);
                retire_locked();
                for (size_t c = 0; c < n_; c++) {
                    drop_done_locked(c);
                }
            }
            continue;
        }
        if (!inflight_.empty()) {
            const uint64_t r = inflight_.front();
            lk.unlock();
            int code = be_->collect(outs.data(), olens.data());
            Round &done = round(r);
            for (size_t c = 0; c < n_; c++) {
                const size_t n = (code == 0 && done.has[c]) ? std::min<size_t>(olens[c], out_cap_) : 0;
                if (n) {
                    memcpy(done.out.data() + out_off_[c], outs[c], n);
                }
                done.out_len[c] = n;
            }
            lk.lock();
            done.state = DONE;
            inflight_.pop_front();
            if (code != 0) {
                fprintf(stderr, "<3>batcher: device results could not be collected: %d\n", code);
                fail_locked(code);
            }
            cv_result_.notify_all();
            retire_locked();
            for (size_t c = 0; c < n_; c++) {
                if (abandoned_[c]) {
                    drop_done_locked(c);  // nobody will take these
                }
            }
            continue;
        }
        if (stopping_) {
            break;
        }
        if (fresh && rd.contributed > 0 && rd.writers == 0) {
#ifdef SDRM_TSAN_BUILD
            // gcc 11's ThreadSanitizer does not intercept pthread_cond_clockwait (what a steady_clock deadline becomes) and
            // then loses track of the mutex: the sanitizer build waits on the system clock instead (tests/san/run.sh)
            cv_work_.wait_until(lk, std::chrono::system_clock::now() + (rd.first + std::chrono::microseconds(max_wait_us_) - std::chrono::steady_clock::now()));
#else
            cv_work_.wait_until(lk, rd.first + std::chrono::microseconds(max_wait_us_));
#endif
        } else {
            cv_work_.wait(lk);
        }
    }
}

}  // namespace sdrm

// ---- C-ABI over a Batcher* (include/sdrmodem_hip.h); sdrm_batcher_create lives with the backend it instantiates

extern "C" void sdrm_batcher_put(sdrm_batcher *b, size_t channel, const sdrm_cf32 *buffer, size_t len) {
    if (b != nullptr) {
        reinterpret_cast<sdrm::Batcher *>(b)->put(channel, buffer, len);
    }
}

extern "C" void sdrm_batcher_take(sdrm_batcher *b, size_t channel, int8_t **output, size_t *output_len) {
    if (b != nullptr && output != nullptr && output_len != nullptr) {
        reinterpret_cast<sdrm::Batcher *>(b)->take(channel, output, output_len);
    }
}

extern "C" void sdrm_batcher_complete(sdrm_batcher *b, size_t channel) {
    if (b != nullptr) {
        reinterpret_cast<sdrm::Batcher *>(b)->complete(channel);
    }
}

extern "C" int sdrm_batcher_reset_channel(sdrm_batcher *b, size_t channel, const sdrm_fsk_config *config) {
    return b != nullptr ? reinterpret_cast<sdrm::Batcher *>(b)->reset_channel(channel, config) : -1;
}
extern "C" int sdrm_batcher_reset_channel_offset(sdrm_batcher *b, size_t channel, const sdrm_fsk_config *config, int64_t rx_offset_hz) {
    return b != nullptr ? reinterpret_cast<sdrm::Batcher *>(b)->reset_channel(channel, config, rx_offset_hz) : -1;
}

extern "C" void sdrm_batcher_interrupt(sdrm_batcher *b, size_t channel) {
    if (b != nullptr) {
        reinterpret_cast<sdrm::Batcher *>(b)->interrupt(channel);
    }
}

extern "C" void sdrm_batcher_abandon(sdrm_batcher *b, size_t channel) {
    if (b != nullptr) {
        reinterpret_cast<sdrm::Batcher *>(b)->abandon(channel);
    }
}

extern "C" int sdrm_batcher_error(const sdrm_batcher *b) {
    return b ? reinterpret_cast<const sdrm::Batcher *>(b)->error() : -1;
}

extern "C" size_t sdrm_batcher_channels(const sdrm_batcher *b) {
    return b ? reinterpret_cast<const sdrm::Batcher *>(b)->channels() : 0;
}

extern "C" uint64_t sdrm_batcher_rounds(const sdrm_batcher *b) {
    return b ? reinterpret_cast<const sdrm::Batcher *>(b)->rounds_launched() : 0;
}

extern "C" void sdrm_batcher_destroy(sdrm_batcher *b) { delete reinterpret_cast<sdrm::Batcher *>(b); }
