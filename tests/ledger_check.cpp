// ledger_check.cpp -- TEST INFRASTRUCTURE (CPU suite): the device-wide ledger of waiting hand-off workgroups
// (sdr-modem_amd/host/ledger.cpp) through a deterministic scenario; the threaded stress is tests/san/host_stress.cpp.
// Built and run by tests/test_abi_cpu.py; prints "ledger ok" and exits 0 when every expectation holds.
#include <stdio.h>

#include <atomic>

#include "../sdr-modem_amd/host/ledger.h"

static int failures = 0;
#define EXPECT(c)                                                              \
    do {                                                                       \
        if (!(c)) {                                                            \
            fprintf(stderr, "EXPECT failed: %s (line %d)\n", #c, __LINE__);   \
            failures++;                                                        \
        }                                                                      \
    } while (0)

static bool fired(void *e) { return static_cast<std::atomic<bool> *>(e)->load(); }

int main() {
    sdrm::WaitLedger l;
    std::atomic<bool> ev[4];
    for (auto &e : ev) e = false;
    int a, b, c, d;  // four owners (only their addresses matter)
    // two batches of 80 waiting workgroups each fit a limit of 192, a third does not
    EXPECT(l.admit(&a, &ev[0], 80, 192, false, fired));
    EXPECT(l.admit(&b, &ev[1], 80, 192, false, fired));
    EXPECT(!l.admit(&c, &ev[2], 80, 192, false, fired));
    // a caller with a smaller limit of its own (a DC workgroup that fills its CU) is refused by the same total
    EXPECT(!l.admit(&d, &ev[3], 2, 16, false, fired));
    // an entry whose event has "fired" before it was armed is NOT reaped (an event not yet recorded reads as complete)
    ev[0] = true;
    EXPECT(!l.admit(&c, &ev[2], 80, 192, false, fired));
    l.arm(&a);
    EXPECT(l.admit(&c, &ev[2], 80, 192, false, fired));  // now it is: a's place went to c without a saying anything
    l.release(&a);                                       // ... and a's own release finds nothing, which is fine
    // an owner that comes again replaces its own entry (it has seen its previous call end)
    EXPECT(l.admit(&b, &ev[1], 100, 192, false, fired));
    EXPECT(!l.admit(&d, &ev[3], 20, 192, false, fired));  // 80 (c) + 100 (b) + 20 > 192
    l.release(&c);
    EXPECT(l.admit(&d, &ev[3], 20, 192, false, fired));
    l.release(&b);
    l.release(&d);
    // plain handles: a call is admitted while NO other plain call is in flight (SDRM_HAND_MAX_PLAIN = 1)
    l.plain_begin();                                      // a's own call
    EXPECT(l.admit(&a, &ev[0], 2, 192, true, fired));
    l.plain_begin();                                      // b's: a's is in flight
    EXPECT(!l.admit(&b, &ev[1], 2, 192, true, fired));
    EXPECT(l.crowded(true) && l.crowded(false));          // ... which the one-load test in front says too (two refusals counted)
    EXPECT(!l.admit(&d, &ev[3], 40, 192, false, fired)); // a batch sees two plain calls in flight
    l.plain_end();                                        // b's call ends
    EXPECT(!l.admit(&d, &ev[3], 40, 192, false, fired)); // one plain call in flight is still one too many for a batch
    l.plain_end();                                        // a's ends
    EXPECT(!l.crowded(false));
    EXPECT(l.admit(&d, &ev[3], 40, 192, false, fired));
    // a refusal leaves the owner nothing: d (40 workgroups listed) asks again while three plain calls are in flight, is refused --
    // and its old entry is gone with that, so that somebody else gets the whole budget (a and b, plain, released themselves)
    l.release(&a);
    l.release(&b);
    l.plain_begin();
    EXPECT(!l.admit(&d, &ev[3], 40, 192, false, fired));
    l.plain_end();
    EXPECT(l.admit(&c, &ev[2], 192, 192, false, fired));
    l.release(&c);
    uint64_t taken = 0, refused = 0;
    uint32_t peak = 0;
    l.stats(&taken, &refused, &peak);
    EXPECT(taken == 8 && refused == 10);
    EXPECT(peak == 192);
    printf(failures ? "ledger FAILED\n" : "ledger ok\n");
    return failures ? 1 : 0;
}
