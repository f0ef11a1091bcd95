"""K3 time with / without the DC stage in the pipeline: python tools/k3_ab.py [channels]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen
Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = 131072
base = np.stack([siggen.gmsk_channel(i, 2 * N) for i in range(32)])
x = torch.from_numpy(np.stack([np.roll(base[c % 32], 977 * (c // 32)) for c in range(Cn)]).view(np.float32)).cuda()
st = torch.cuda.current_stream().cuda_stream
for dc in (True, False):
    b = binding.Batch([(48000, 9600, 5000, 1, 2000, dc, N)] * Cn)
    for i in range(6):
        b.process_device(x.data_ptr() + (i % 2) * N * 8, 2 * N, [N] * Cn, st)
    torch.cuda.synchronize()
    b.timing_enable(True)
    t0 = time.perf_counter()
    for i in range(48):
        b.process_device(x.data_ptr() + (i % 2) * N * 8, 2 * N, [N] * Cn, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 48
    print("dc=%s: %.3f ms per step, kernels %s" % (dc, dt * 1e3, [round(b.timing_read(w)[0] / max(b.timing_read(w)[1], 1), 3) for w in range(3)]))
    b.close()
