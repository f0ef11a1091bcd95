// ledger.cpp -- see ledger.h.  Plain C++: built into the library and, with the sanitizers, into tests/san/host_stress.
#include "ledger.h"

#include <algorithm>

namespace sdrm {

bool WaitLedger::admit(const void *owner, void *event, unsigned waiting, unsigned limit, bool plain, DoneFn done) {
    std::lock_guard<std::mutex> g(m_);
    unsigned sum = 0;
    for (size_t i = 0; i < held_.size();) {
        const Entry &e = held_[i];
        if (e.owner == owner || (e.armed && done(e.event))) {
            // (the owner's own entry: it has seen its previous call end.  Somebody else's: its owner finds out when it looks --
            // release() tolerates a missing entry)
            held_[i] = held_.back();
            held_.pop_back();
            continue;
        }
        sum += e.waiting;
        i++;
    }
    // (the owner's old entry is gone whatever the answer: after a refusal it holds nothing)
    if (plain_calls_.load(std::memory_order_relaxed) - (plain ? 1 : 0) >= SDRM_HAND_MAX_PLAIN || sum + waiting > limit) {
        refused_++;
        return false;
    }
    held_.push_back({owner, event, waiting, false});
    taken_++;
    peak_ = std::max(peak_, sum + waiting);
    return true;
}

void WaitLedger::arm(const void *owner) {
    std::lock_guard<std::mutex> g(m_);
    for (Entry &e : held_) {
        if (e.owner == owner) {
            e.armed = true;
        }
    }
}

void WaitLedger::release(const void *owner) {
    std::lock_guard<std::mutex> g(m_);
    for (size_t i = 0; i < held_.size(); i++) {
        if (held_[i].owner == owner) {
            held_[i] = held_.back();
            held_.pop_back();
            return;
        }
    }
}

void WaitLedger::stats(uint64_t *taken, uint64_t *refused, uint32_t *peak_waiting) {
    std::lock_guard<std::mutex> g(m_);
    if (taken) *taken = taken_;
    if (refused) *refused = refused_ + crowded_refusals_.load(std::memory_order_relaxed);
    if (peak_waiting) *peak_waiting = peak_;
}

}  // namespace sdrm
