"""Soak of the dsp_worker surface (create / put / destroy, file sinks): rounds of several workers at once, each fed by its own
thread with random buffer sizes and pauses -- private demodulators (blocking file-source queues), workers bound to a shared
batcher, and (round 4) workers placed by a node of two batchers on the device, some with the file source's frequency offset,
some with symbols beyond the fast stages' range -- then destroyed; every rx.demod2client.<id>.s8 must be byte-identical to the oracle's soft bits of that worker's
stream and every rx.sdr2demod.<id>.cf32 to what was put.  python tools/soak_workers.py [seconds] [first seed]"""
import ctypes as C, os, sys, time, threading, tempfile
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
L = binding.load()
rounds = files = 0


def fail(msg):
    print(msg, flush=True)
    os._exit(1)


while time.time() < t_end:
    rng = np.random.default_rng(seed)
    bufsize = int(rng.choice([2048, 4096]))
    cfgs = [c for c in _cases(seed, int(rng.integers(1, 10))) if orc.Fsk(*c, bufsize).code == 0]
    layout = int(rng.integers(0, 3))  # 0 private, 1 one shared batcher, 2 a node of two batchers
    shared = layout == 1 and len(cfgs) > 1
    if layout == 2 and seed % 4 == 0:
        cfgs.append((240000, 600, 5000, 1, 2000, bool(rng.integers(0, 2))))  # 400 samples per symbol: the generic stages
    offsets = [int(rng.choice([0, 0, 1000, -2500])) for _ in cfgs]
    with tempfile.TemporaryDirectory() as tmp:
        bt = None
        node = None
        if layout == 2:
            node = binding.Node(cfgs[0] + (bufsize,), max(2, (len(cfgs) + 1) // 2 + 1), n_batchers=2, devices=[0, 0],
                                batcher=(4, int(rng.choice([300, 3000])), True))
            if node.code != 0:
                fail("node create failed: %d (seed %d)" % (node.code, seed))
        if shared:  # geometry of the batcher = the configuration with the longest filters, so that every client fits
            order = sorted(range(len(cfgs)), key=lambda i: (-cfgs[i][0] / max(cfgs[i][4], 1), -cfgs[i][0] / cfgs[i][1]))
            geom = cfgs[order[0]] + (bufsize,)
            bt = binding.Batcher([geom] * len(cfgs), slots=4, max_wait_us=int(rng.choice([300, 3000])), blocking=True)
            if bt.code != 0:
                bt, shared = None, False
        workers, streams = [], []
        for i, c in enumerate(cfgs):
            wc = binding.WorkerConfig(c[0], c[1], c[2], c[3], c[4], c[5], True, 0, bufsize, int(rng.integers(2, 6)), True, tmp.encode())
            if shared:
                wc.batcher = bt.h
                wc.batcher_channel = i
            if node is not None:
                wc.node = node.h
                wc.source_id = 1 + i % 2
            wc.rx_offset_hz = offsets[i]
            w = C.c_void_p()
            code = L.dsp_worker_create(100 + i, -1, C.byref(wc), C.byref(w))
            if code != 0:
                workers.append(None); streams.append(None)
                continue
            workers.append(w)
            n_buf = int(rng.integers(1, 12))
            sizes = [int(rng.choice([1, 100, 1999, bufsize])) for _ in range(n_buf)]
            sig = siggen.gmsk_channel(int(rng.integers(0, 1 << 30)), sum(sizes), fs=c[0], baud=c[1])
            streams.append((sig, sizes, [float(rng.choice([0, 0, 0.0005, 0.002])) for _ in sizes]))

        def feed(i):
            sig, sizes, pauses = streams[i]
            p = 0
            for n, dt in zip(sizes, pauses):
                if dt:
                    time.sleep(dt)
                part = np.ascontiguousarray(sig[p:p + n]).view(np.float32)
                L.dsp_worker_put(part.ctypes.data, n, workers[i])
                p += n
            L.dsp_worker_destroy(workers[i])  # the pill goes in behind the queued buffers

        th = [threading.Thread(target=feed, args=(i,)) for i in range(len(cfgs)) if workers[i] is not None]
        for t in th:
            t.start()
        for t in th:
            t.join(120)
            if t.is_alive():
                fail("HANG: seed %d (%s workers)" % (seed, "batcher-bound" if shared else "private"))
        for i, c in enumerate(cfgs):
            if workers[i] is None:
                continue
            sig, sizes, _ = streams[i]
            o = orc.Fsk(*c, bufsize)
            osc = orc.Nco(1.0, c[0], bufsize) if offsets[i] else None
            want, p = [], 0
            for n in sizes:
                part = sig[p:p + n]
                if osc is not None:
                    part = osc.multiply(offsets[i], np.ascontiguousarray(part).view(np.float32))
                want.append(o.process(part)[0]); p += n
            want = np.concatenate(want) if want else np.zeros(0, np.int8)
            got = np.fromfile(os.path.join(tmp, "rx.demod2client.%d.s8" % (100 + i)), dtype=np.int8)
            dump = np.fromfile(os.path.join(tmp, "rx.sdr2demod.%d.cf32" % (100 + i)), dtype=np.complex64)
            if not np.array_equal(dump, sig[:p]):
                fail("MISMATCH iq dump: seed %d worker %d (%s)" % (seed, i, "batcher-bound" if shared else "private"))
            if not np.array_equal(got, want):
                fail("MISMATCH soft bits: seed %d worker %d cfg %s sizes %s offset %d (%s): %d vs %d bytes" % (
                    seed, i, c, sizes, offsets[i], ("private", "batcher-bound", "node-placed")[layout], len(got), len(want)))
            files += 1
        if bt is not None:
            bt.close()
        if node is not None:
            if sum(node.stat(k).clients for k in range(2)) != 0:
                fail("node still counts clients after every worker is gone: seed %d" % seed)
            node.close()
    rounds += 1; seed += 1
print("worker soak ok: %d rounds, %d output files identical, %.0f s" % (rounds, files, budget), flush=True)
