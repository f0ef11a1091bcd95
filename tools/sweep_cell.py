"""one cell of the heuristics sweep: python tools/sweep_cell.py <channels> <chunk> [steps] [batch buffer length] -> "<ms per step> <Msamples/s> <kernel ms x3>"
(the stream holds, the companion grid and the clock stage's shape are chosen by the library from the batch, or forced
through SDRM_FRONT_HOLD / SDRM_DC_FIRST / SDRM_K3_COMPANY / SDRM_K3_LANES; every variant is its own process because the
library reads those once)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen
Cn, N = int(sys.argv[1]), int(sys.argv[2])
K = int(sys.argv[3]) if len(sys.argv) > 3 else max(12, min(200, int(3.0e9 / (Cn * N))))
MAXN = int(sys.argv[4]) if len(sys.argv) > 4 else N  # the batch's buffer length (calls of N samples on a longer-buffer batch)
cfg = (48000, 9600, 5000, 1, 2000, True, MAXN)
base = np.stack([siggen.gmsk_channel(i, 2 * N) for i in range(8)]).view(np.float32)
bt = torch.from_numpy(base).cuda()
x = torch.empty((Cn, 4 * N), dtype=torch.float32, device="cuda")
for c in range(Cn):
    x[c] = torch.roll(bt[c % 8], shifts=2 * 977 * (c // 8))
b = binding.Batch([cfg] * Cn)
assert b.code == 0
st = torch.cuda.current_stream().cuda_stream
lens = (binding.C.c_size_t * Cn)(*([N] * Cn))
for i in range(192):  # fill + (calls the calibration did not cover) the online refinement
    b.process_device(x.data_ptr() + (i % 2) * N * 8, 2 * N, lens, st)
torch.cuda.synchronize()
b.timing_enable(True)
t0 = time.perf_counter()
for i in range(K):
    b.process_device(x.data_ptr() + (i % 2) * N * 8, 2 * N, lens, st)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
km = [b.timing_read(w)[0] / max(b.timing_read(w)[1], 1) for w in range(3)]
print("%.3f %.0f %.3f %.3f %.3f" % (dt * 1e3, Cn * N / dt / 1e6, km[0], km[1], km[2]))
