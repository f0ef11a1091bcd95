"""Liveness soak of the batcher in LIVE mode (reference queue behaviour for SDR sources: a producer never blocks, the newest
undelivered buffer is overwritten when everything is taken) with slow consumers and poison pills at random moments.
Content cannot be compared (which buffers were dropped is the batcher's business); what is checked: nothing hangs, nothing
crashes, every take returns a plausible count, no client gets more results than it put buffers, consumers end at the pill.  python tools/soak_live.py [seconds] [first seed]"""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen
import orc
from test_gpu_fuzz import _cases

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
rounds = taken = 0


def fail(msg):
    print(msg, flush=True)
    os._exit(1)


while time.time() < t_end:
    rng = np.random.default_rng(seed)
    maxlen = 4096
    cfgs = [c + (maxlen,) for c in _cases(seed, int(rng.integers(2, 24)))]
    cfgs = [c for c in cfgs if orc.Fsk(*c).code == 0]
    bt = binding.Batcher(cfgs, slots=int(rng.integers(3, 6)), max_wait_us=int(rng.choice([200, 2000])), blocking=False)
    if bt.code != 0:
        seed += 1; continue
    K = int(rng.integers(5, 40))
    sig = [siggen.gmsk_channel(int(rng.integers(0, 1 << 30)), 3 * maxlen, fs=c[0], baud=c[1]) for c in cfgs]
    n_taken = [0] * len(cfgs)
    pause = [[float(rng.choice([0, 0, 0.0003, 0.002])) for _ in range(K)] for _ in cfgs]
    pill_at = [int(rng.integers(1, K + 1)) for _ in cfgs]

    def producer(c):
        for k in range(pill_at[c]):
            n = int([1, 100, 1999, maxlen][(k + c) % 4])
            bt.put(c, sig[c][k % 2 * 100:k % 2 * 100 + n])
        bt.interrupt(c)

    def consumer(c):
        while True:
            if pause[c][n_taken[c] % K]:
                time.sleep(pause[c][n_taken[c] % K])
            r = bt.take(c)
            if r is None:
                return
            if len(r) > maxlen:
                fail("IMPLAUSIBLE count %d: seed %d client %d" % (len(r), seed, c))
            n_taken[c] += 1

    th = [threading.Thread(target=f, args=(c,)) for c in range(len(cfgs)) for f in (producer, consumer)]
    for t in th:
        t.start()
    for t in th:
        t.join(60)
        if t.is_alive():
            fail("HANG: seed %d (live mode, %d clients)" % (seed, len(cfgs)))
    for c in range(len(cfgs)):
        if n_taken[c] > pill_at[c]:
            fail("MORE results than buffers: seed %d client %d: %d > %d" % (seed, c, n_taken[c], pill_at[c]))
    taken += sum(n_taken)
    bt.close()
    rounds += 1; seed += 1
print("live soak ok: %d rounds, %d buffers taken, %.0f s" % (rounds, taken, budget), flush=True)
