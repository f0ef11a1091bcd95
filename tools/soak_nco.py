"""Soak of the Doppler pre-correction (NCO phase recursion + mix) against the oracle: random channel counts (one to three
phase workgroups, partly filled), sampling rates, ragged inputs, batches that end anywhere, empty batches, channels that skip
the correction, shifts beyond the sampling rate.  The corrected IQ must equal the oracle's within 1e-6 and the number of floats that differ at all is counted (the
oscillator samples are the correctly rounded floats of the exact cos/sin: a libm differs on ~1e-9 of them).
python tools/soak_nco.py [seconds] [first seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding
import orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
rounds = total = same = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    n_ch = int(rng.choice([1, 3, 17, 64, 65, 70, 130, 200]))
    maxlen = int(rng.choice([3000, 5000, 9000]))
    fs, baud = [(48000, 9600), (240000, 19200), (96000, 4800)][int(rng.integers(0, 3))]
    g = binding.Batch([(fs, baud, 5000, 1, 2000, bool(rng.integers(0, 2)), maxlen)] * n_ch)
    if g.code != 0:
        seed += 1; continue
    ncos = [orc.Nco(1.0, fs, maxlen) for _ in range(n_ch)]
    for call in range(int(rng.integers(2, 7))):
        parts, segs, want = [], [], []
        for c in range(n_ch):
            n = int(rng.choice([0, 1, 3, 63, 64, 65, 127, 128, 129, 1000, maxlen])) if rng.random() < 0.5 else int(rng.integers(0, maxlen + 1))
            x = (rng.standard_normal(2 * n) * 0.5).astype(np.float32)
            parts.append(x.view(np.complex64))
            if n == 0 or rng.random() < 0.1:
                want.append(None)
                continue
            k = int(rng.integers(1, 7))
            cuts = np.sort(rng.integers(0, n + 1, size=k - 1))
            lens = np.diff(np.concatenate([[0], cuts, [n]])).astype(int)
            ref, off = [], 0
            for ln in lens:
                f = int(rng.choice([fs + 12000, -fs - 22000, 2 * fs + 1457])) if rng.random() < 0.08 else int(rng.integers(-20000, 20001))
                segs.append((c, int(ln), f))
                if ln:
                    ref.append(ncos[c].multiply(f, x[2 * off:2 * (off + ln)]))
                off += ln
            want.append(np.concatenate(ref))
        g.process_nco(parts, segs)
        for c in range(n_ch):
            if want[c] is None and not segs:
                continue  # no correction requested anywhere yet: the batch has no NCO buffers to read back
            got = g.last_mixed(c)
            if want[c] is None:
                if len(got) != 0:
                    print("MISMATCH nco: seed %d call %d channel %d: uncorrected channel has mixed output" % (seed, call, c), flush=True); os._exit(1)
                continue
            if len(got) != len(want[c]) or np.abs(got - want[c]).max() >= 1e-6:
                print("MISMATCH nco: seed %d call %d channel %d of %d (fs %d): len %d vs %d, max diff %g" % (
                    seed, call, c, n_ch, fs, len(got), len(want[c]), np.abs(got[:min(len(got), len(want[c]))] - want[c][:min(len(got), len(want[c]))]).max()), flush=True)
                os._exit(1)
            total += len(got); same += int(np.sum(got.view(np.uint32) == want[c].view(np.uint32)))
    g.close()
    rounds += 1; seed += 1
print("nco soak ok: %d rounds, %.1f M corrected floats (%.1f M samples), %d differ from the oracle's (%.2e), %.0f s" % (
    rounds, total / 1e6, total / 2e6, total - same, (total - same) / max(total, 1), budget), flush=True)
