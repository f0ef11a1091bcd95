"""Latency of the reference-shaped call: one handle, fsk_demod_process on an n-sample host buffer (perf_fsk_modem.c: 4096),
checked against the oracle on the last call.  SDRM_HANDOFF=0 switches the in-call hand-off off (before / after)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen
import orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cfg = (48000, 4800, 5000, 2, 2000, True, 2016000)
sig = siggen.gmsk_channel(1, 4 * n, fs=48000, baud=4800)
d = binding.FskDemod(*cfg)
o = orc.Fsk(*cfg)
got = want = None
for k in range(20):
    part = sig[(k % 4) * n:(k % 4 + 1) * n]
    got = d.process(part); want, _ = o.process(part)
ts = []
for k in range(calls):
    part = sig[(k % 4) * n:(k % 4 + 1) * n]
    t0 = time.perf_counter()
    got = d.process(part)
    ts.append(time.perf_counter() - t0)
    want, _ = o.process(part)
ok = np.array_equal(np.asarray(got), want)
ts = np.array(ts) * 1e6
print("fsk_demod_process(%d samples), SDRM_HANDOFF=%s: median %.1f us per call, min %.1f (one handle = one stream); last call %s the oracle" %
      (n, os.environ.get("SDRM_HANDOFF", "default"), np.median(ts), ts.min(), "==" if ok else "!="))
