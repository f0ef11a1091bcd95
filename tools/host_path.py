"""PCIe-inclusive rate of the host-buffer boundary (sdrm_batch_process): pageable vs pinned caller buffers.
python tools/host_path.py [channels] [chunk]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen
Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
cfg = (48000, 9600, 5000, 1, 2000, True, N)
base = np.stack([siggen.gmsk_channel(i, N) for i in range(8)]).view(np.float32)
L = binding.load()
for kind in ("pageable", "pinned"):
    b = binding.Batch([cfg] * Cn)
    host = torch.from_numpy(np.tile(base, (Cn // 8, 1)).copy())
    if kind == "pinned":
        host = host.pin_memory()
    ptrs = (C.c_void_p * Cn)(*[host[c].data_ptr() for c in range(Cn)])
    lens = (C.c_size_t * Cn)(*([N] * Cn))
    outs = (binding.i8p * Cn)()
    olens = (C.c_size_t * Cn)()
    for i in range(2):
        assert L.sdrm_batch_process(b.h, ptrs, lens, outs, olens) == 0
    t0 = time.perf_counter()
    K = 6
    for i in range(K):
        L.sdrm_batch_process(b.h, ptrs, lens, outs, olens)
    dt = (time.perf_counter() - t0) / K
    print("%s caller buffers: %.2f ms per call of %d x %d samples = %.0f Msamples/s (%.1f GB/s of IQ over PCIe)" % (
        kind, dt * 1e3, Cn, N, Cn * N / dt / 1e6, Cn * N * 8 / dt / 1e9))
    del b
