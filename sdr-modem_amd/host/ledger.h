// ledger.h -- device-wide admission of in-call hand-offs (host logic only; the device side is sdrm_kernels.hip, the caller sdrm_call.hip).
//
// A hand-off call's DC and clock-stage workgroups sit on their compute units and WAIT (for the front-end's first tiles, for the DC
// stage's first blocks).  The compute units they hold belong to the whole device, not to the batch: a server with one handle per
// client (reference src/dsp_worker.c:188, src/tcp_server.c:659) runs hundreds of one-channel batches side by side, and every one
// parks a clock-stage workgroup (141 KB of LDS: a whole CU) and a DC workgroup.  Admission per batch (round 5) let the waiting
// workgroups of different handles add up CU by CU.  So the count is kept per device, process-wide: a call takes the hand-off only
// while the workgroups waiting on the device, its own included, stay within the caller's limit -- otherwise it runs its stages in
// stream order like any call behind a running one, which is always safe.  An entry leaves the ledger when its call has finished:
// the owner says so when it sees the call's end, and anybody's admission reaps the entries whose completion event has fired, so
// that a client that goes quiet after a call holds nothing.
//
// Streams and hardware queues: every wait of a hand-off call looks at work enqueued BEFORE the waiting kernel by the same thread
// (front-end, then DC stage, then clock stage), queues are served in order, and the front-end waits for nobody -- the oldest
// unfinished kernel on the device is therefore always at the head of its queue and runnable as soon as it finds a CU, whatever
// shares its queue.  (The one look FORWARD is the placement hold in front of a multi-channel batch's front-end -- it lets the DC and
// clock workgroups take their CUs first -- and that is bounded at 0.4 ms, not at two seconds.)  What the ledger guarantees is the CU.
//
// The second count: blocking calls of plain handles (one-channel batches) in flight, with or without the hand-off.  Each is a chain
// of one-workgroup kernels on a stream of its own, and HIP serves the streams of a priority level from a handful of hardware
// queues: about three such chains run at once, whatever the number of handles (64 handles x 131072 samples: 1.4 ms per call
// against 3.9 alone; more queues -- GPU_MAX_HW_QUEUES -- make it worse, profiles/r06_handles.txt).  A hand-off call adds two
// streams and two spinning workgroups to that: with one handle calling it wins a millisecond per call (3.9 -> 2.9 ms); with two, only
// while the clock stage's side stream sits on the highest priority level (3.0 against 3.9 ms) -- and a process's first stream on that
// level brings a third pool of hardware queues to life, which cost 256 plain handles 15 % for the rest of their run; with both side
// streams on the middle level two handles collide (6.8 against 3.9 ms); from three on it loses or ties either way, and behind dozens
// of queued chains its front-end would wait for tens of milliseconds with its DC and clock workgroups holding CUs.  The hand-off is an
// optimisation of latency on a quiet device: a plain handle's call takes it only while NO other such call is in flight
// (SDRM_HAND_MAX_PLAIN = 1), on side streams of the middle level.
#ifndef SDRM_LEDGER_H
#define SDRM_LEDGER_H

#include <stdint.h>

#include <atomic>
#include <mutex>
#include <vector>

#define SDRM_HAND_MAX_PLAIN 1

namespace sdrm {

class WaitLedger {
  public:
    // has the event that was recorded behind an admitted call fired?  (hipEventQuery in the library, a flag in the tests)
    typedef bool (*DoneFn)(void *event);

    // `waiting` workgroups of owner's next call may wait on the device: true = listed (not armed yet), false = the budget is taken.
    // owner's previous entry, if any, is dropped first (the caller has seen that call end).  plain: owner is a plain handle whose own
    // blocking call is already counted by plain_begin().
    bool admit(const void *owner, void *event, unsigned waiting, unsigned limit, bool plain, DoneFn done);
    // owner's event has been recorded behind the call: from here on anybody's admission may find the call over and reap the entry
    void arm(const void *owner);
    // owner has seen its call end (or is going away: its event is about to be destroyed); tolerates a missing entry
    void release(const void *owner);
    // too many plain handles' calls in flight for anybody's hand-off?  (one atomic load: asked first, so that a crowded device's calls
    // are refused before they look at an event or a kernel's geometry; counted as a refusal)
    bool crowded(bool plain) {
        if (plain_calls_.load(std::memory_order_relaxed) - (plain ? 1 : 0) < SDRM_HAND_MAX_PLAIN) {
            return false;
        }
        crowded_refusals_.fetch_add(1, std::memory_order_relaxed);
        return true;
    }
    void plain_begin() { plain_calls_.fetch_add(1, std::memory_order_relaxed); }
    void plain_end() { plain_calls_.fetch_sub(1, std::memory_order_relaxed); }
    void stats(uint64_t *taken, uint64_t *refused, uint32_t *peak_waiting);

  private:
    struct Entry {
        const void *owner;
        void *event;
        unsigned waiting;
        bool armed;  // an event not yet recorded reads as complete: such an entry is never reaped
    };
    std::mutex m_;
    std::vector<Entry> held_;
    uint64_t taken_ = 0, refused_ = 0;
    unsigned peak_ = 0;
    std::atomic<int> plain_calls_{0};
    std::atomic<uint64_t> crowded_refusals_{0};
};

}  // namespace sdrm

#endif  // SDRM_LEDGER_H
