"""Diagnostic: the device timeline of a batch's first 60 calls (short calls on a long-buffer batch):
python tools/timeline_first_calls.py <channels> <chunk> <batch buffer length>"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen
Cn, N, MAXN = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
MIX = os.environ.get("MIX") is not None  # BASELINE configs[4]'s mix with three NCO batches per channel and call
FIRST = int(os.environ.get("FIRST", "0"))  # calls made before the table is attached
if MIX:
    import bench
    cfgs = [(240000, 19200, 5000, 5, 2000, True, MAXN) if c % 2 == 0 else (48000, 1200, 5000, 8, 2000, True, MAXN) for c in range(Cn)]
    a = siggen.gmsk_channel(1, 2 * N, fs=240000, baud=19200)
    b_ = siggen.gmsk_channel(2, 2 * N, fs=48000, baud=1200)
    x = torch.from_numpy(np.stack([a if c % 2 == 0 else b_ for c in range(Cn)]).view(np.float32)).cuda()
    segs = bench.config5_segments(range(Cn), N)
else:
    cfgs = [(48000, 9600, 5000, 1, 2000, True, MAXN)] * Cn
    base = np.stack([siggen.gmsk_channel(i, 2 * N) for i in range(8)]).view(np.float32)
    x = torch.from_numpy(np.tile(base, (Cn // 8, 1))).cuda()
torch.cuda.synchronize()
b = binding.Batch(cfgs)
assert b.code == 0
L = binding.load()
L.sdrm_batch_timeline.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
st = torch.cuda.current_stream().cuda_stream
lens = (C.c_size_t * Cn)(*([N] * Cn))
def call(i):
    if MIX:
        b.process_device_nco(x.data_ptr() + (i % 2) * N * 8, 2 * N, lens, segs, st)
    else:
        b.process_device(x.data_ptr() + (i % 2) * N * 8, 2 * N, lens, st)
for i in range(FIRST):
    call(i)
L.sdrm_batch_timeline(b.h, 1, None, 0)
for i in range(FIRST, FIRST + 60):
    call(i)
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
