"""Soak: seeded differential fuzz of the HIP path against the oracle for a given number of seconds (GPU box).
Every round draws a batch of random configurations (as tests/test_gpu_fuzz.py), random signals -- GMSK, white noise over
many decades, silence, denormal-scale and constant stretches spliced in -- and random call lengths up to 20000 samples;
every fifth round adds channels of 267 to 400 samples per symbol (generic stages); every second round channels with a
deviation of 1 .. 1000 Hz (timing loops outside their tame range: the clock stage's global-memory form);
every third round also drives one plain fsk_demod handle with repeated lengths (the graph replay), every fourth the
pinned-arena pipeline with three calls in flight, every fifth device-resident calls queued back to back, every 25th a batch
of 400 to 2100 channels; every third round forces a clock-stage workgroup shape.  Bit-exact or it stops.
python tools/soak_fuzz.py [seconds] [first seed]"""
import os, sys, time
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
rounds = calls = 0


def signal(rng, cfg, n, i):
    kind = rng.integers(0, 6)
    if kind <= 2:
        s = siggen.gmsk_channel(int(rng.integers(0, 1 << 30)), n, fs=cfg[0], baud=cfg[1])
    elif kind == 3:
        s = (10.0 ** rng.uniform(-6, 6, n) * np.exp(1j * rng.uniform(-np.pi, np.pi, n))).astype(np.complex64)
    elif kind == 4:
        s = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64) * np.float32(10.0 ** rng.uniform(-30, 3))
    else:
        s = np.full(n, np.complex64(1 + 0j))
    s = s.copy()
    for _ in range(int(rng.integers(0, 4))):  # splice in silence / denormal-scale / constant stretches
        a = int(rng.integers(0, n)); b = min(n, a + int(rng.integers(1, 3000)))
        what = rng.integers(0, 3)
        s[a:b] = 0 if what == 0 else (s[a:b] * np.float32(1e-38) if what == 1 else s[a])
    return s


while time.time() < t_end:
    rng = np.random.default_rng(seed)
    maxlen = int(rng.choice([6000, 20000]))
    # every third round forces a clock-stage workgroup shape (read per launch): small batches then run the shapes of large ones --
    # 32x512 takes the in-call hand-off like the default 16x1024
    if seed % 3 == 0:
        os.environ["SDRM_K3_LANES"] = str(np.random.default_rng(seed + 7).choice(["32x512", "32x512", "16x512", "64x256p", "32x256"]))
    else:
        os.environ.pop("SDRM_K3_LANES", None)
    cfgs = [c + (maxlen,) for c in _cases(seed, int(rng.integers(3, 40)))]
    if seed % 7 == 2:  # round 3: long symbols (up to 240 samples each, DC boxcars up to 7680) inside an ordinary batch
        cfgs += [c + (maxlen,) for c in [(240000, 1200, 5000, 1, 2000, True), (192000, 1200, 2400, 1, 4000, bool(rng.integers(0, 2))),
                                         (240000, 1000, 5000, 1, 2000, True), (96000, 1200, 7500, 1, 1000, False)]]
    if seed % 5 == 1:  # round 4: symbols beyond the fast stages' range (generic DC / clock stages) inside an ordinary batch
        cfgs += [c + (maxlen,) for c in [(240000, 600, 5000, 1, 2000, True), (240000, 900, 2400, 1, 1000, bool(rng.integers(0, 2))),
                                         (480000, 1200, 5000, 1, 2000, False), (96000, 300, 5000, 1, 2000, True)]]
    if seed % 2 == 1:  # round 5: deviations of 1 .. 1000 Hz (gains up to 38000: timing loops that stand still or walk backwards)
        for _ in range(int(rng.integers(2, 12))):
            fs, baud = [(48000, 9600), (48000, 4800), (240000, 19200), (48000, 19200), (192000, 40000), (48000, 1200), (96000, 9600)][rng.integers(7)]
            cfgs.append((fs, baud, int(np.exp(rng.uniform(0, np.log(1000)))) * int(rng.choice([1, 1, 1, -1])), int(rng.choice([1, 1, 2, 4, 5, 8])),
                         2000, bool(rng.integers(2)), maxlen))
    oracles = [orc.Fsk(*c) for c in cfgs]
    keep = [o.code == 0 for o in oracles]
    cfgs = [c for c, k in zip(cfgs, keep) if k]; oracles = [o for o, k in zip(oracles, keep) if k]
    keep_soft = bool(rng.integers(0, 2))
    g = binding.Batch(cfgs, keep_soft=keep_soft)
    if g.code != 0:
        print("seed %d: batch create %d, skipped" % (seed, g.code)); seed += 1; continue
    total = 9 * maxlen  # the pinned-arena and device-resident rounds below make up to eight calls of up to maxlen samples from position 0
    sigs = [signal(rng, c, total, i) for i, c in enumerate(cfgs)]
    pos = [0] * len(cfgs)
    for call in range(6):
        lens = [int(rng.choice([0, 1, 7, 100, 257, 1999, 4096, maxlen])) for _ in cfgs]
        parts = [s[p:p + n] for s, p, n in zip(sigs, pos, lens)]
        pos = [p + n for p, n in zip(pos, lens)]
        g8 = g.process(parts)
        for i, o in enumerate(oracles):
            o8, of = o.process(parts[i])
            if not np.array_equal(g8[i], o8):
                print("MISMATCH int8: seed %d call %d channel %d cfg %s len %d" % (seed, call, i, cfgs[i], lens[i])); sys.exit(1)
            if keep_soft:
                gf = g.last_soft(i)
                nn = np.isnan(of) & np.isnan(gf)
                if not np.array_equal(gf.view(np.uint32)[~nn], of.view(np.uint32)[~nn]):
                    print("MISMATCH float: seed %d call %d channel %d cfg %s" % (seed, call, i, cfgs[i])); sys.exit(1)
        calls += 1
    g.close()
    if seed % 3 == 0:  # one plain handle, repeated lengths: graph replay
        c = cfgs[0]
        d = binding.FskDemod(*c); o = orc.Fsk(*c)
        s = signal(rng, c, 40000, 0); p = 0
        for n in [4096] * 4 + [1000] * 3 + [4096] * 2 + [c[6] // 3] * 3:
            part = s[p:p + n]; p += n
            if not np.array_equal(d.process(part), o.process(part)[0]):
                print("MISMATCH handle: seed %d cfg %s len %d" % (seed, c, n)); sys.exit(1)
        d.close()
    if seed % 4 == 3:  # pinned arena, three calls in flight (copies and stages of consecutive calls overlap)
        n_ch = len(cfgs)
        g = binding.Batch(cfgs)
        arena = g.arena(4)
        lens_plan = [[int(rng.choice([0, 100, 1999, 4096, maxlen])) for _ in cfgs] for _ in range(7)]
        pos = [0] * n_ch; got = [[] for _ in cfgs]; pending = 0
        for k, lens in enumerate(lens_plan):
            slot = k % 4
            for c in range(n_ch):
                part = sigs[c][pos[c]:pos[c] + lens[c]].view(np.float32)
                arena[slot, c, :len(part)] = part
                pos[c] += lens[c]
            if g.submit(slot, lens) != 0:
                for c, o8 in enumerate(g.collect()):
                    got[c].append(o8)
                pending -= 1
                assert g.submit(slot, lens) == 0
            pending += 1
        while pending:
            for c, o8 in enumerate(g.collect()):
                got[c].append(o8)
            pending -= 1
        for c in range(n_ch):
            o = orc.Fsk(*cfgs[c]); exp = []; p0 = 0
            for lens in lens_plan:
                exp.append(o.process(sigs[c][p0:p0 + lens[c]])[0]); p0 += lens[c]
            if not np.array_equal(np.concatenate(got[c]), np.concatenate(exp)):
                print("MISMATCH pipelined: seed %d channel %d cfg %s" % (seed, c, cfgs[c])); sys.exit(1)
        g.close()
    if seed % 5 == 1:  # device-resident input, several calls in flight on a side stream, only the last one fetched
        import torch
        n_ch = len(cfgs)
        g = binding.Batch(cfgs)
        stride = maxlen
        lens_plan = [[int(rng.choice([0, 100, 1999, 4096, maxlen])) for _ in cfgs] for _ in range(int(rng.integers(2, 9)))]
        bufs, pos = [], [0] * n_ch
        for lens in lens_plan:
            host = np.zeros((n_ch, 2 * stride), np.float32)
            for c in range(n_ch):
                part = sigs[c][pos[c]:pos[c] + lens[c]].view(np.float32)
                host[c, :len(part)] = part
                pos[c] += lens[c]
            bufs.append(torch.from_numpy(host).cuda())
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        for t, lens in zip(bufs, lens_plan):
            g.process_device(t.data_ptr(), stride, lens, side.cuda_stream)
        data, got_lens = g.fetch(maxlen)
        for c in range(n_ch):
            o = orc.Fsk(*cfgs[c]); p0 = 0; last = None
            for lens in lens_plan:
                last = o.process(sigs[c][p0:p0 + lens[c]])[0]; p0 += lens[c]
            if got_lens[c] != len(last) or not np.array_equal(data[c, :got_lens[c]], last):
                print("MISMATCH device-resident: seed %d channel %d cfg %s calls %d" % (seed, c, cfgs[c], len(lens_plan))); sys.exit(1)
        g.close()
        del bufs
    if seed % 25 == 0:  # many channels: the 64-channel clock-stage workgroups, the DC / front-end placement holds
        n_big = int(rng.choice([400, 1100, 2100, 2700]))
        pool = cfgs[:6]
        big = [pool[i % len(pool)] for i in range(n_big)]
        g = binding.Batch(big)
        if g.code == 0:
            obig = [orc.Fsk(*c) for c in big]
            pos = [int(rng.integers(0, 1000)) for _ in big]
            for call in range(4):
                lens = [int(rng.choice([0, 100, 1999, 4096, min(maxlen, 6000)])) for _ in big]
                parts = [sigs[i % len(pool)][p:p + n] for i, (p, n) in enumerate(zip(pos, lens))]
                pos = [p + n for p, n in zip(pos, lens)]
                g8 = g.process(parts)
                for i, o in enumerate(obig):
                    if not np.array_equal(g8[i], o.process(parts[i])[0]):
                        print("MISMATCH big batch: seed %d channels %d call %d channel %d cfg %s" % (seed, n_big, call, i, big[i])); sys.exit(1)
                calls += 1
            g.close()
    rounds += 1; seed += 1
print("soak ok: %d rounds, %d batch calls, seeds up to %d, %.0f s" % (rounds, calls, seed - 1, budget))
