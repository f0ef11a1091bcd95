"""What do the per-kernel HIP timing events cost the pipeline?  256 channels x 131072, steps with timing off / on"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen
Cn, N = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 131072
base = np.stack([siggen.gmsk_channel(i, 2 * N) for i in range(32)])
x = torch.from_numpy(np.tile(base, (Cn // 32, 1)).view(np.float32)).cuda()
b = binding.Batch([(48000, 9600, 5000, 1, 2000, True, N)] * Cn)
st = torch.cuda.current_stream().cuda_stream
def run(K):
    t0 = time.perf_counter()
    for i in range(K):
        b.process_device(x.data_ptr() + (i % 2) * N * 8, 2 * N, [N] * Cn, st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K
run(16)
for rep in range(3):
    for on in (False, True):
        b.timing_enable(on)
        dt = run(64)
        print("timing %-5s: %.4f ms per step, %.0f Msamples/s" % (on, dt * 1e3, Cn * N / dt / 1e6))
