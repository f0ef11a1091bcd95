"""A batch sized for many clients with few of them live (a server's batcher off-peak): `present` of `channels` channels take part in
every call, the rest are ABSENT.  ms per call (pipelined, device-resident) and the device timeline of a few calls.
python tools/sparse_batch.py [channels] [present] [samples]     env: SDRM_K3_COMPANY, SDRM_AUTOTUNE, SDRM_HANDOFF ..."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen
Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 512
P = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N = int(sys.argv[3]) if len(sys.argv) > 3 else 131072
base = np.stack([siggen.gmsk_channel(i, 2 * N) for i in range(8)])
x = torch.from_numpy(np.tile(base, (Cn // 8, 1)).view(np.float32)).cuda()
b = binding.Batch([(48000, 9600, 5000, 1, 2000, True, N)] * Cn)
st = torch.cuda.current_stream().cuda_stream
ABSENT = binding.C.c_size_t(-1).value
lens = (binding.C.c_size_t * Cn)(*([N] * P + [ABSENT] * (Cn - P)))
def call(i): b.process_device(x.data_ptr() + (i % 2) * N * 8, 2 * N, lens, st)
for i in range(8): call(i)
torch.cuda.synchronize()
b.timing_enable(True)
t0 = time.perf_counter()
for i in range(40): call(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 40 * 1e3
km = [b.timing_read(w) for w in range(3)]
b.timing_enable(False)
b.timeline_begin()
for i in range(8): call(i)
torch.cuda.synchronize()
tl = b.timeline_read()
print("%d channels, %d present: %.3f ms per call; kernels front %.3f dc %.3f clock %.3f; schedule %s" % (
    Cn, P, dt, *[m / max(n, 1) for m, n in km], {k: b.schedule().get(k) for k in ("clock_stage", "company_blocks", "front_hold")}))
for r in range(2, min(len(tl), 6)):
    t_ref = tl[2, 0]
    print("   call %d: " % r + "  ".join("[%7.3f, %7.3f]" % (tl[r, 2 * k] - t_ref, tl[r, 2 * k + 1] - t_ref) for k in range(3)))
