"""Clock-stage time with long calls (the companion grid must live as long as the stage): python tools/long_chunk.py [channels] [samples]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen
Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1048576
base = np.stack([siggen.gmsk_channel(i, N) for i in range(8)])
x = torch.from_numpy(np.tile(base, (Cn // 8, 1)).view(np.float32)).cuda()
st = torch.cuda.current_stream().cuda_stream
b = binding.Batch([(48000, 9600, 5000, 1, 2000, True, N)] * Cn)
for i in range(3):
    b.process_device(x.data_ptr(), N, [N] * Cn, st)
torch.cuda.synchronize()
b.timing_enable(True)
t0 = time.perf_counter()
K = 8
for i in range(K):
    b.process_device(x.data_ptr(), N, [N] * Cn, st)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("%d channels x %d samples: %.3f ms per call, kernels %s" % (Cn, N, dt * 1e3, [round(b.timing_read(w)[0] / max(b.timing_read(w)[1], 1), 3) for w in range(3)]))
