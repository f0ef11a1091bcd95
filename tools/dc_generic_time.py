"""The generic DC stage's cost: a batch of ordinary channels plus generic ones (48 kHz / 150 baud: 320 samples per symbol, a boxcar
of 10240 samples), 131072-sample calls; ms per step and the DC stream's time per call.  python tools/dc_generic_time.py [generic channels]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen
import orc
n_gen = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = 131072
fast = (48000, 9600, 5000, 1, 2000, True, N)
slow = (48000, 150, 5000, 1, 2000, True, N)
cfgs = [fast] * 64 + [slow] * n_gen
sig = np.stack([siggen.gmsk_channel(i, 2 * N, fs=c[0], baud=c[1]) for i, c in enumerate(cfgs)])
x = torch.from_numpy(sig.view(np.float32)).cuda()
b = binding.Batch(cfgs)
assert b.code == 0
st = torch.cuda.current_stream().cuda_stream
lens = [N] * len(cfgs)
for i in range(4):
    b.process_device(x.data_ptr() + (i % 2) * N * 8, 2 * N, lens, st)
b.sync()
b.timing_enable(True)
t0 = time.perf_counter()
K = 12
for i in range(K):
    b.process_device(x.data_ptr() + (i % 2) * N * 8, 2 * N, lens, st)
b.sync()
dt = (time.perf_counter() - t0) / K
km = [b.timing_read(w) for w in range(3)]
print("64 ordinary + %d generic channels: %.3f ms per step; front / dc / clock streams %s ms per call" %
      (n_gen, dt * 1e3, [round(m / max(n, 1), 3) for m, n in km]))
data, got = b.fetch(N)
o = orc.Fsk(*slow)
want = None
for i in range(4 + K):
    want = o.process(sig[64][(i % 2) * N:(i % 2) * N + N])[0]
print("generic channel's last call %s the oracle (%d symbols)" % ("==" if np.array_equal(data[64][:got[64]], want) else "!=", len(want)))
