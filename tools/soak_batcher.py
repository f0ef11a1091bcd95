"""Soak of the batcher (sdrm_batcher_*): rounds of random client counts, configurations, buffer sizes and timing jitter, every
client with its own producer and consumer thread on one shared batch (blocking mode); each client's soft bits must equal
the oracle's for its own stream, bit for bit.  python tools/soak_batcher.py [seconds] [first seed]"""
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
rounds = buffers = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    maxlen = int(rng.choice([4096, 8192]))
    cfgs = [c + (maxlen,) for c in _cases(seed, int(rng.integers(2, 40)))]
    cfgs = [c for c in cfgs if orc.Fsk(*c).code == 0]
    K = int(rng.integers(2, 8))
    sizes = [[int(rng.choice([1, 17, 500, 3000, maxlen])) for _ in range(K)] for _ in cfgs]
    sigs = [siggen.gmsk_channel(int(rng.integers(0, 1 << 30)), sum(sz), fs=c[0], baud=c[1]) for c, sz in zip(cfgs, sizes)]
    chunks = [[s[sum(sz[:k]):sum(sz[:k + 1])] for k in range(K)] for s, sz in zip(sigs, sizes)]
    bt = binding.Batcher(cfgs, slots=int(rng.integers(3, 7)), max_wait_us=int(rng.choice([300, 2000, 50000])), blocking=True)
    if bt.code != 0:
        print("seed %d: batcher create %d, skipped" % (seed, bt.code)); seed += 1; continue
    got = [[] for _ in cfgs]
    jitter = [[float(rng.choice([0, 0, 0.0005, 0.003])) for _ in range(2 * K)] for _ in cfgs]

    def producer(c):
        for k in range(K):
            if jitter[c][k]:
                time.sleep(jitter[c][k])
            bt.put(c, chunks[c][k])

    def consumer(c):
        for k in range(K):
            if jitter[c][K + k]:
                time.sleep(jitter[c][K + k])
            got[c].append(bt.take(c))

    th = [threading.Thread(target=f, args=(c,)) for c in range(len(cfgs)) for f in (producer, consumer)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
        if t.is_alive():
            print("HANG: seed %d" % seed, flush=True); os._exit(2)
    for c, cfg in enumerate(cfgs):
        o = orc.Fsk(*cfg)
        for k in range(K):
            want = o.process(chunks[c][k])[0]
            if got[c][k] is None or not np.array_equal(got[c][k], want):
                print("MISMATCH batcher: seed %d client %d buffer %d cfg %s sizes %s got %s want %d" % (seed, c, k, cfg, sizes[c], None if got[c][k] is None else len(got[c][k]), len(want)), flush=True); sys.exit(1)
            buffers += 1
    for c in range(len(cfgs)):
        bt.interrupt(c)
    bt.close()
    rounds += 1; seed += 1
print("batcher soak ok: %d rounds, %d buffers, seeds up to %d, %.0f s" % (rounds, buffers, seed - 1, budget))
