"""Cost of the Doppler pre-correction (K0) in front of the demodulator: the same device-resident calls with and without
one constant-frequency NCO batch per channel and call.  python tools/nco_times.py [channels] [chunk]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen
Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
cfg = (48000, 9600, 5000, 1, 2000, True, N)
base = np.stack([siggen.gmsk_channel(i, 2 * N) for i in range(8)])
x = torch.from_numpy(np.tile(base, (Cn // 8, 1)).view(np.float32)).cuda()
L = binding.load()
L.sdrm_batch_process_device_nco.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(binding.NcoSegment),
                                            C.c_size_t, C.c_void_p]
st = torch.cuda.current_stream().cuda_stream
lens = (C.c_size_t * Cn)(*([N] * Cn))
for with_nco in (False, True):
    b = binding.Batch([cfg] * Cn)
    # three batches per channel and call, like a pass with the shift changing inside the buffer
    segs = (binding.NcoSegment * (3 * Cn))(*[binding.NcoSegment(c, n, 1500 - 7 * c + 100 * k)
                                             for c in range(Cn) for k, n in enumerate((48000, 48000, N - 96000))])
    def call(i):
        ptr = C.c_void_p(x.data_ptr() + (i % 2) * N * 8)
        if with_nco:
            assert L.sdrm_batch_process_device_nco(b.h, ptr, 2 * N, lens, segs, 3 * Cn, C.c_void_p(st)) == 0
        else:
            assert L.sdrm_batch_process_device(b.h, ptr, 2 * N, lens, C.c_void_p(st)) == 0
    for i in range(4):
        call(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 24
    for i in range(K):
        call(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("%s Doppler correction: %.2f ms per call of %d x %d samples = %.0f Msamples/s" % (
        "with" if with_nco else "without", dt * 1e3, Cn, N, Cn * N / dt / 1e6))
    b.close()
