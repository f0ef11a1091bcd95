"""One cell of an A/B between library builds: python tools/ab_cell.py <lib.so> <channels> <chunk> [steps]
-> "<ms per step> <Msamples/s> <kernel ms front / dc / clock>".  Binds only the six entry points it needs, so that builds of
earlier rounds (which lack newer symbols) can be measured beside HEAD on the same box in the same session."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import siggen
from sdr_modem_amd.binding import FskConfig
lib, Cn, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
K = int(sys.argv[4]) if len(sys.argv) > 4 else max(12, min(200, int(3.0e9 / (Cn * N))))
L = C.CDLL(os.path.abspath(lib))
vp = C.c_void_p
L.sdrm_batch_create.argtypes = [C.POINTER(FskConfig), C.c_size_t, C.c_int, C.c_uint32, C.POINTER(vp)]
L.sdrm_batch_process_device.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t), vp]
L.sdrm_batch_timing_enable.argtypes = [vp, C.c_int]
L.sdrm_batch_timing_read.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
L.sdrm_batch_destroy.argtypes = [vp]
L.sdrm_batch_destroy.restype = None
cfgs = (FskConfig * Cn)(*[FskConfig(48000, 9600, 5000, 1, 2000, True, N) for _ in range(Cn)])
base = np.stack([siggen.gmsk_channel(i, 2 * N) for i in range(8)]).view(np.float32)
bt = torch.from_numpy(base).cuda()
x = torch.empty((Cn, 4 * N), dtype=torch.float32, device="cuda")
for c in range(Cn):
    x[c] = torch.roll(bt[c % 8], shifts=2 * 977 * (c // 8))
h = vp()
assert L.sdrm_batch_create(cfgs, Cn, -1, 0, C.byref(h)) == 0
st = torch.cuda.current_stream().cuda_stream
lens = (C.c_size_t * Cn)(*([N] * Cn))


def call(i):
    assert L.sdrm_batch_process_device(h, vp(x.data_ptr() + (i % 2) * N * 8), 2 * N, lens, vp(st)) == 0


for i in range(6):
    call(i)
torch.cuda.synchronize()
L.sdrm_batch_timing_enable(h, 1)
t0 = time.perf_counter()
for i in range(K):
    call(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
km = []
for w in range(3):
    ms, n = C.c_double(), C.c_uint64()
    L.sdrm_batch_timing_read(h, w, C.byref(ms), C.byref(n))
    km.append(ms.value / max(n.value, 1))
print("%.3f %.0f %.3f %.3f %.3f" % (dt * 1e3, Cn * N / dt / 1e6, km[0], km[1], km[2]))
L.sdrm_batch_destroy(h)
