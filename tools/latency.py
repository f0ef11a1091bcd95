"""Latency of the reference-shaped call: one handle, fsk_demod_process on a 4096-sample host buffer (perf_fsk_modem.c)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ramp = np.zeros(n, dtype=np.complex64)
ramp.real = (np.arange(n) % 256).astype(np.float32)
d = binding.FskDemod(48000, 4800, 5000, 2, 2000, True, 2016000)
for _ in range(20):
    d.process(ramp)
t0 = time.perf_counter()
for _ in range(200):
    d.process(ramp)
dt = (time.perf_counter() - t0) / 200
print("fsk_demod_process(%d samples): %.1f us per call (%s)" % (n, dt * 1e6, "one handle = one stream"))
