"""BASELINE configs[4] in one GPU's share: half 240 kHz / 19200 baud (decimation 5), half 48 kHz / 1200 baud (decimation 8)
channels, every channel with its own Doppler ramp (three NCO batches per channel and call), device-resident input.
python tools/config5.py [channels]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen
Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = 131072
cfgs = [(240000, 19200, 5000, 5, 2000, True, N) if c % 2 == 0 else (48000, 1200, 5000, 8, 2000, True, N) for c in range(Cn)]
a = siggen.gmsk_channel(1, 2 * N, fs=240000, baud=19200)
b_ = siggen.gmsk_channel(2, 2 * N, fs=48000, baud=1200)
x = torch.from_numpy(np.stack([a if c % 2 == 0 else b_ for c in range(Cn)]).view(np.float32)).cuda()
L = binding.load()
L.sdrm_batch_process_device_nco.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(binding.NcoSegment),
                                            C.c_size_t, C.c_void_p]
b = binding.Batch(cfgs)
assert b.code == 0
st = torch.cuda.current_stream().cuda_stream
lens = (C.c_size_t * Cn)(*([N] * Cn))
segs = (binding.NcoSegment * (3 * Cn))(*[binding.NcoSegment(c, n, -10000 + (80 * c) % 20000 + 500 * k)
                                         for c in range(Cn) for k, n in enumerate((40000, 40000, N - 80000))])
def call(i):
    assert L.sdrm_batch_process_device_nco(b.h, C.c_void_p(x.data_ptr() + (i % 2) * N * 8), 2 * N, lens, segs, 3 * Cn, C.c_void_p(st)) == 0
WARM = int(sys.argv[2]) if len(sys.argv) > 2 else 192
for i in range(WARM):  # the pipeline's fill + the online refinement for calls with NCO batches (~130 calls from the 17th on)
    call(i)
torch.cuda.synchronize()
b.timing_enable(True)
K = 96  # the timed region ends with the pipeline's drain (about two steps' worth)
t0 = time.perf_counter()
for i in range(K):
    call(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("mixed 240k/19200/d5 + 48k/1200/d8 with per-channel Doppler, %d channels: %.2f ms per step, %.0f Msamples/s, kernels (front, dc, clock) %s" % (
    Cn, dt * 1e3, Cn * N / dt / 1e6, [round(b.timing_read(w)[0] / max(b.timing_read(w)[1], 1), 3) for w in range(3)]))
