"""Soak of PRIVATE plain handles calling at once with long buffers (GPU box), bit-exact against the oracle or it stops: the
reference's process model (one fsk_demod and one blocking call per client thread, src/dsp_worker.c:188 / :75) in the regime where
a call may take the in-call hand-off -- or be refused it by the device's ledger (sdr-modem_amd/host/ledger.cpp) because other
clients' calls are in flight.  A round: 1 .. 10 client threads, each with its own handle of a random configuration and buffer size
(13000 .. 140000 samples), 3 .. 6 calls of random lengths with random pauses between them, so that admitted and refused calls
alternate on the same handle (side streams and the handle's own stream in turn); every call's int8 soft bits against the oracle's.
python tools/soak_handles.py [seconds] [first seed]"""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen
import orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
KINDS = [(48000, 9600, 5000, 1, 2000, True), (48000, 4800, 5000, 2, 2000, True), (48000, 9600, 5000, 1, 2000, False),
         (240000, 19200, 5000, 5, 2000, True), (48000, 1200, 5000, 1, 2000, True), (192000, 40000, 5000, 1, 2000, True)]
failed = []


def client(i, cfg, chunks, pauses):
    d = binding.FskDemod(*cfg)
    o = orc.Fsk(*cfg)
    if d.code != 0 or o.code != 0:
        failed.append("create failed: client %d cfg %s (%d / %d)" % (i, cfg, d.code, o.code))
        return
    for k, part in enumerate(chunks):
        got = d.process(part)
        want = o.process(part)[0]
        if not np.array_equal(got, want):
            failed.append("MISMATCH plain handle: client %d cfg %s call %d len %d: %d symbols, the oracle has %d" % (i, cfg, k, len(part), len(got), len(want)))
            break
        if pauses[k] > 0:
            time.sleep(pauses[k])
    if binding.load().sdrm_fsk_demod_error(d.h) != 0:
        failed.append("handle in the error state: client %d cfg %s" % (i, cfg))
    d.close()


rounds = calls = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 11))
    work = []
    for i in range(n):
        kind = KINDS[int(rng.integers(0, len(KINDS)))]
        maxlen = int(rng.choice([13000, 20000, 32768, 65536, 100000, 131072, 140000]))
        cfg = kind + (maxlen,)
        k = int(rng.integers(3, 7))
        lens = [int(rng.choice([maxlen, maxlen, maxlen - 7, max(1, maxlen // 2), 4096, 1])) for _ in range(k)]
        sig = siggen.gmsk_channel(int(rng.integers(0, 1 << 30)), sum(lens), fs=cfg[0], baud=cfg[1])
        pos = np.cumsum([0] + lens)
        chunks = [sig[pos[j]:pos[j + 1]] for j in range(k)]
        pauses = [float(rng.choice([0.0, 0.0, 0.001, 0.004])) for _ in range(k)]
        work.append((i, cfg, chunks, pauses))
    th = [threading.Thread(target=client, args=w) for w in work]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
        if t.is_alive():
            print("HANG in round seed %d" % seed, flush=True)
            os._exit(2)
    if failed:
        print("%s (round seed %d, %d clients)" % (failed[0], seed, n), flush=True)
        os._exit(1)
    rounds += 1
    calls += sum(len(w[2]) for w in work)
    seed += 1
taken, refused, peak = binding.handoff_stats()
print("handles soak ok: %d rounds, %d blocking calls of private handles (hand-off taken %d, refused %d, peak %d waiting workgroups), seeds up to %d, %.0f s"
      % (rounds, calls, taken, refused, peak, seed - 1, budget), flush=True)
