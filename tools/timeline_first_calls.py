"""Diagnostic: the device timeline of a batch's first 60 calls (short calls on a long-buffer batch):
python tools/timeline_first_calls.py <channels> <chunk> <batch buffer length>"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen
Cn, N, MAXN = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = (48000, 9600, 5000, 1, 2000, True, MAXN)
base = np.stack([siggen.gmsk_channel(i, 2 * N) for i in range(8)]).view(np.float32)
x = torch.from_numpy(np.tile(base, (Cn // 8, 1))).cuda()
torch.cuda.synchronize()
b = binding.Batch([cfg] * Cn)
assert b.code == 0
L = binding.load()
L.sdrm_batch_timeline.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
L.sdrm_batch_timeline(b.h, 1, None, 0)
st = torch.cuda.current_stream().cuda_stream
lens = (C.c_size_t * Cn)(*([N] * Cn))
for i in range(60):
    b.process_device(x.data_ptr() + (i % 2) * N * 8, 2 * N, lens, st)
torch.cuda.synchronize()
tl = np.zeros(64 * 6, dtype=np.uint64)
rows = L.sdrm_batch_timeline(b.h, 0, tl.ctypes.data, 64)
tl = tl[:rows * 6].reshape(rows, 6).astype(np.float64)
t_ref = tl[0, 0]
print("call: front [start, end]  dc [start, end]  clock [start, end]   (ms, device clock); clock end - previous clock end")
prev = None
for r in range(rows):
    d = (tl[r, 5] - prev) / 1e5 if prev is not None else 0.0
    prev = tl[r, 5]
    print("%2d: " % r + "  ".join("[%8.3f, %8.3f]" % ((tl[r, 2 * k] - t_ref) / 1e5, (tl[r, 2 * k + 1] - t_ref) / 1e5) for k in range(3)) + "   %.3f" % d)
print(b.schedule())
