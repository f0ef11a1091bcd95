"""Diagnostic: the clock stage's cycles inside BASELINE configs[4]'s mix (256 channels, 240 kHz / 19200 / d5 interleaved with
48 kHz / 1200 / d8, per-channel Doppler): python tools/k3_probe_c5.py [channels] [interleave 0/1] [nco 0/1]"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen
import bench
Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 256
inter = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nco = int(sys.argv[3]) if len(sys.argv) > 3 else 1
N = 131072
heavy, light = (240000, 19200, 5000, 5, 2000, True, N), (48000, 1200, 5000, 8, 2000, True, N)
cfgs = [heavy if (c % 2 == 0 if inter else c < Cn // 2) else light for c in range(Cn)]
mine = bench.config5_segments(range(Cn), N) if nco else np.zeros((0, 3), dtype=np.int64)
torch.cuda.set_device(0)
b, x, step = bench.config5(torch, binding, siggen, torch.device("cuda", 0), cfgs, N, plan_step=lambda: mine)
L = binding.load()
L.sdrm_batch_k3_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
NCALLS, MID = 12, 8
L.sdrm_batch_k3_stamps(b.h, MID, None, 0)
import time
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(NCALLS):
    step(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / NCALLS
km = [b.timing_read(w) for w in range(3)]
print("%d channels, %s, nco %d: %.3f ms per step, kernels %s" % (Cn, "interleaved" if inter else "blocks", nco, dt * 1e3,
      [round(t / max(n, 1), 3) for t, n in km]))
slots = (Cn + 15) // 16
waves = slots
out = np.zeros(slots * 4 + 24, dtype=np.uint64)
L.sdrm_batch_k3_stamps(b.h, 1, out.ctypes.data, slots)
out = out[:slots * 4].reshape(slots, 4)
starts = [int(out[w][0]) >> 32 for w in range(waves)]
print("loop start of each consumer wave, us after the first: %s" % " ".join("%.0f" % ((s - min(starts)) / 100.0) for s in starts))
for w in range(waves):
    out[w][3] = int(out[w][3]) & 0xffffffff
for w in list(range(min(waves, 3))) + [waves - 1]:
    stg, drn, packed, it = [int(v) for v in out[w]]
    stg &= 0xffffffff
    nb, ticks = packed & 0xffffffff, packed >> 32
    print("wave %d: wait-for-producer %.0f cyc/step, loops %.0f cyc/step, %d steps, %.1f iterations/step, %.0f cyc/iteration, "
          "%.3f ms at %.0f MHz" % (w, stg / max(nb, 1), drn / max(nb, 1), nb, it / max(nb, 1), drn / max(it, 1), ticks / 1e5, (stg + drn) / max(ticks, 1) * 100))
print("cycles per iteration, all %d consumer waves: %s" % (waves, " ".join("%.0f" % (int(out[w][1]) / max(int(out[w][3]), 1)) for w in range(waves))))
