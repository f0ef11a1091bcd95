"""Why is config 5 slower inside bench.py than in tools/config5.py?  Modes: plain | numpy (segments as an (n, 3) array per
call, as bench.py passes them) | live (a 256-channel headline batch created and stepped first, kept alive) | both"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
Cn, N = 256, 131072
st = torch.cuda.current_stream().cuda_stream
keep = None
if mode in ("live", "both"):
    base = np.stack([siggen.gmsk_channel(i, 2 * N) for i in range(8)])
    x0 = torch.from_numpy(np.tile(base, (Cn // 8, 1)).view(np.float32)).cuda()
    keep = binding.Batch([(48000, 9600, 5000, 1, 2000, True, N)] * Cn)
    for i in range(40):
        keep.process_device(x0.data_ptr() + (i % 2) * N * 8, 2 * N, [N] * Cn, st)
    torch.cuda.synchronize()
cfgs = [(240000, 19200, 5000, 5, 2000, True, N) if c % 2 == 0 else (48000, 1200, 5000, 8, 2000, True, N) for c in range(Cn)]
a = siggen.gmsk_channel(1, 2 * N, fs=240000, baud=19200)
b_ = siggen.gmsk_channel(2, 2 * N, fs=48000, baud=1200)
x = torch.from_numpy(np.stack([a if c % 2 == 0 else b_ for c in range(Cn)]).view(np.float32)).cuda()
b = binding.Batch(cfgs)
lens = (C.c_size_t * Cn)(*([N] * Cn))
tuples = [(c, n, -10000 + (80 * c) % 20000 + 500 * k) for c in range(Cn) for k, n in enumerate((40000, 40000, N - 80000))]
segs = (binding.NcoSegment * (3 * Cn))(*[binding.NcoSegment(*t) for t in tuples])
arr = np.array(tuples, dtype=np.int64).reshape(-1, 3)
def call(i):
    b.process_device_nco(x.data_ptr() + (i % 2) * N * 8, 2 * N, lens, arr if mode in ("numpy", "both") else segs, st)
for i in range(4):
    call(i)
torch.cuda.synchronize()
b.timing_enable(True)
K = 24
t0 = time.perf_counter()
for i in range(K):
    call(i)
t_host = (time.perf_counter() - t0) / K
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("%-6s: %.2f ms per step (host side %.2f ms per call), %.0f Msamples/s, kernels %s" % (
    mode, dt * 1e3, t_host * 1e3, Cn * N / dt / 1e6, [round(b.timing_read(w)[0] / max(b.timing_read(w)[1], 1), 3) for w in range(3)]))
