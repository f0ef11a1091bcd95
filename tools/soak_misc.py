"""Soak of the channel hand-over paths (GPU box), bit-exact against the oracle or it stops:
  A. sdrm_batch_reset_channel: channels of a running batch are handed to new streams with new configurations at random points
  B. SDRM_SHARED_SLOTS: client threads create a plain fsk_demod handle with random parameters, push a few buffers through
     fsk_demod_process, destroy it and start over -- the handles of the process share one batcher whose slots change hands
python tools/soak_misc.py [seconds] [first seed]"""
import os, sys, time, threading
os.environ.setdefault("SDRM_SHARED_SLOTS", "12")
os.environ.setdefault("SDRM_SHARED_WAIT_US", "500")
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


def fail(msg):
    print(msg, flush=True)
    os._exit(1)


# ---- B first (the first handle of the process fixes the shared geometry): a long-filter, DC-on configuration
GEOM = (240000, 2400, 5000, 5, 1000, True, 8192)
keeper = binding.FskDemod(*GEOM)
assert keeper.code == 0
stop = False
counts = [0] * 10


def client(i):
    rng = np.random.default_rng(1000 + i)
    while not stop:
        cfg = _cases(int(rng.integers(0, 1 << 30)), 1)[0] + (int(rng.choice([2048, 4096, 8192])),)
        o = orc.Fsk(*cfg)
        if o.code != 0:
            continue
        d = binding.FskDemod(*cfg)
        if d.code != 0:
            continue
        sig = siggen.gmsk_channel(int(rng.integers(0, 1 << 30)), 6 * cfg[6], fs=cfg[0], baud=cfg[1])
        p = 0
        for k in range(int(rng.integers(1, 6))):
            n = int(rng.choice([1, 100, 1999, cfg[6]]))
            part = sig[p:p + n]; p += n
            if not np.array_equal(d.process(part), o.process(part)[0]):
                fail("MISMATCH shared handle: client %d cfg %s buffer %d len %d" % (i, cfg, k, n))
            counts[i] += 1
        d.close()


threads = [threading.Thread(target=client, args=(i,), daemon=True) for i in range(10)]
for t in threads:
    t.start()

# ---- A in the main thread meanwhile
rounds = resets = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    maxlen = 6000
    pool = [c + (maxlen,) for c in _cases(seed, 30)]
    pool = [c for c in pool if orc.Fsk(*c).code == 0]
    n_ch = int(rng.integers(2, 12))
    cfgs = [pool[int(rng.integers(0, len(pool)))] for _ in range(n_ch)]
    g = binding.Batch(cfgs)
    if g.code != 0:
        seed += 1; continue
    oracles = [orc.Fsk(*c) for c in cfgs]
    sigs = [siggen.gmsk_channel(int(rng.integers(0, 1 << 30)), 10 * maxlen, fs=c[0], baud=c[1]) for c in cfgs]
    pos = [0] * n_ch
    for call in range(8):
        for c in range(n_ch):
            if rng.random() < 0.2:  # hand the channel to a new stream, most of the time with new parameters
                new = pool[int(rng.integers(0, len(pool)))] if rng.random() < 0.7 else cfgs[c]
                code = g.reset_channel(c, new)
                if code == 0:
                    cfgs[c] = new
                    oracles[c] = orc.Fsk(*new)
                    sigs[c] = siggen.gmsk_channel(int(rng.integers(0, 1 << 30)), 10 * maxlen, fs=new[0], baud=new[1])
                    pos[c] = 0
                    resets += 1
        lens = [int(rng.choice([0, 100, 1999, 4096, maxlen])) for _ in range(n_ch)]
        parts = [s[p:p + n] for s, p, n in zip(sigs, pos, lens)]
        pos = [p + n for p, n in zip(pos, lens)]
        g8 = g.process(parts)
        for c in range(n_ch):
            if not np.array_equal(g8[c], oracles[c].process(parts[c])[0]):
                fail("MISMATCH after reset: seed %d call %d channel %d cfg %s" % (seed, call, c, cfgs[c]))
    g.close()
    rounds += 1; seed += 1
stop = True
for t in threads:
    t.join(60)
    if t.is_alive():
        fail("HANG: a shared-handle client did not finish")
keeper.close()
print("misc soak ok: %d reset rounds (%d hand-overs), %d buffers through shared handles, %.0f s" % (rounds, resets, sum(counts), budget), flush=True)
