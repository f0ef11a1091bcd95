"""One BLOCKING call of the batch at a time (what the reference's caller does: src/dsp_worker.c:75 waits for fsk_demod_process):
device-resident input, sdrm_batch_process_device + sdrm_batch_sync per call.  Reports ms per call and, with the device
timeline, when each stage started and ended inside the call.  SDRM_HANDOFF=0/1 switches the in-call hand-off.
python tools/blocking_call.py [channels] [samples] [calls]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen

channels = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 30
cfg = (48000, 9600, 5000, 1, 2000, True)
sig = siggen.gmsk_batch(min(channels, 32), 2 * n)
rows = np.stack([sig[c % len(sig)] for c in range(channels)])
x = torch.from_numpy(rows.view(np.float32)).to("cuda:0")
b = binding.Batch([cfg + (n,)] * channels, device=0)
assert b.code == 0
st = torch.cuda.current_stream().cuda_stream
lens = (binding.C.c_size_t * channels)(*([n] * channels))
for i in range(5):
    b.process_device(x.data_ptr() + (i % 2) * n * 8, 2 * n, lens, st); b.sync()
torch.cuda.synchronize()
ts = []
for i in range(calls):
    t0 = time.perf_counter()
    b.process_device(x.data_ptr() + (i % 2) * n * 8, 2 * n, lens, st)
    b.sync()
    ts.append((time.perf_counter() - t0) * 1e3)
ts = np.array(ts)
if os.environ.get("SDRM_TIMELINE"):
    import ctypes as C
    L = binding.load()
    L.sdrm_batch_timeline.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    L.sdrm_batch_timeline(b.h, 1, None, 0)
    for i in range(6):
        b.process_device(x.data_ptr() + (i % 2) * n * 8, 2 * n, lens, st); b.sync()
    tl = np.zeros(64 * 6, dtype=np.uint64)
    rows_ = L.sdrm_batch_timeline(b.h, 0, tl.ctypes.data, 64)
    tl = tl[:rows_ * 6].reshape(rows_, 6).astype(np.float64)
    print("device timeline per call (ms from the call's first front-end workgroup): front [start, end]  dc [start, end]  clock [start, end]")
    for r in range(rows_):
        t_ref = tl[r, 0]
        print("  %d: " % r + "  ".join("[%7.3f, %7.3f]" % ((tl[r, 2 * k] - t_ref) / 1e5, (tl[r, 2 * k + 1] - t_ref) / 1e5) for k in range(3)))
print("blocking call, %d channels x %d samples, SDRM_HANDOFF=%s: median %.3f ms, min %.3f, max %.3f (%d calls)" %
      (channels, n, os.environ.get("SDRM_HANDOFF", "default"), np.median(ts), ts.min(), ts.max(), calls))
if os.environ.get("SDRM_VERIFY"):
    import orc
    lens_py = [n] * channels
    data, got = b.fetch(n)
    outs = [data[c][:got[c]] for c in range(channels)]
    # replay channel 0 and the last channel through the oracle over the same sequence of calls
    for c in (0, channels - 1):
        o = orc.Fsk(*cfg, n)
        for i in list(range(5)) + list(range(calls)) + (list(range(6)) if os.environ.get("SDRM_TIMELINE") else []):
            part = rows[c][(i % 2) * n:(i % 2) * n + n]
            w8, _ = o.process(part)
        g = outs[c] if outs is not None else None
        print("channel %d: last call %s the oracle (%d symbols)" % (c, "==" if g is not None and np.array_equal(g, w8) else "!=", len(w8)))
b.close()
