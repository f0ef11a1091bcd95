import os, sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import torch, sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen
Cn, N = int(sys.argv[1]), 131072
cfg = (48000, 9600, 5000, 1, 2000, True, N)
base = np.stack([siggen.gmsk_channel(i, 2 * N) for i in range(32)])
x = torch.from_numpy(np.tile(base, (Cn // 32, 1)).view(np.float32)).cuda()
b = binding.Batch([cfg] * Cn)
st = torch.cuda.current_stream().cuda_stream
for i in range(4):
    b.process_device(x.data_ptr() + (i % 2) * N * 8, 2 * N, [N] * Cn, st)
torch.cuda.synchronize()
import ctypes as C
L = binding.load()
L.sdrm_batch_timeline.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
L.sdrm_batch_timeline(b.h, 1, None, 0)
b.timing_enable(True)
t0 = time.perf_counter()
K = 24
for i in range(K):
    b.process_device(x.data_ptr() + (i % 2) * N * 8, 2 * N, [N] * Cn, st)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("channels %d: %.3f ms per step, %.0f Msamples/s, kernels %s" % (Cn, dt * 1e3, Cn * N / dt / 1e6,
      [round(b.timing_read(w)[0] / max(b.timing_read(w)[1], 1), 3) for w in range(3)]))
tl = np.zeros(64 * 6, dtype=np.uint64)
rows = L.sdrm_batch_timeline(b.h, 0, tl.ctypes.data, 64)
tl = tl[:rows * 6].reshape(rows, 6).astype(np.float64)
t_ref = tl[8, 0]
print("call: front [start, end]  dc [start, end]  clock [start, end]   (ms, device clock)")
for r in range(8, min(rows, 16)):
    print("%2d: " % r + "  ".join("[%7.3f, %7.3f]" % ((tl[r, 2 * k] - t_ref) / 1e5, (tl[r, 2 * k + 1] - t_ref) / 1e5) for k in range(3)))
