"""bench.py's own config5 block, alone and behind its end_to_end block"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen
import bench
dev = torch.device("cuda:0")
order = sys.argv[1:] or ["c5"]
for what in order:
    if what == "c5":
        r = bench.config5_single(torch, binding, siggen, dev, 256, 131072)
        print("config5:", r["value"], r["ms_per_step"], r["kernel_ms"])
    elif what == "c5stamp":
        import ctypes as C, numpy as np
        Cn, N = 256, 131072
        cfgs = [(240000, 19200, 5000, 5, 2000, True, N) if c % 2 == 0 else (48000, 1200, 5000, 8, 2000, True, N) for c in range(Cn)]
        mine = bench.config5_segments(range(Cn), N)
        b, x, step = bench.config5(torch, binding, siggen, dev, cfgs, N, 24, plan_step=lambda: mine)
        L = binding.load()
        L.sdrm_batch_k3_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.sdrm_batch_k3_stamps(b.h, 8, None, 0)   # stamp the 8th of the following calls
        for i in range(12):
            step(i)
        torch.cuda.synchronize()
        slots = Cn // 16
        out = np.zeros(slots * 4 + 24, dtype=np.uint64)
        L.sdrm_batch_k3_stamps(b.h, 1, out.ctypes.data, slots)
        o = out[:slots * 4].reshape(slots, 4)
        for w in range(2):
            stg, drn, packed, it = [int(v) for v in o[w]]
            stg &= 0xffffffff; it &= 0xffffffff
            nb, ticks = packed & 0xffffffff, packed >> 32
            print("wave %d: wait %.0f cyc/step, loops %.0f cyc/step, %d steps, %.1f it/step, %.0f cyc/it, %.3f ms at %.0f MHz" % (
                w, stg / nb, drn / nb, nb, it / nb, drn / max(it, 1), ticks / 1e5, (stg + drn) / max(ticks, 1) * 100))
        b.close()
    elif what == "c5tl":
        import ctypes as C, numpy as np
        Cn, N = 256, 131072
        cfgs = [(240000, 19200, 5000, 5, 2000, True, N) if c % 2 == 0 else (48000, 1200, 5000, 8, 2000, True, N) for c in range(Cn)]
        mine = bench.config5_segments(range(Cn), N)
        b, x, step = bench.config5(torch, binding, siggen, dev, cfgs, N, 24, plan_step=lambda: mine)
        L = binding.load()
        L.sdrm_batch_timeline.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.sdrm_batch_timeline(b.h, 1, None, 0)
        for i in range(20):
            step(i)
        torch.cuda.synchronize()
        tl = np.zeros(64 * 6, dtype=np.uint64)
        rows = L.sdrm_batch_timeline(b.h, 0, tl.ctypes.data, 64)
        tl = tl[:rows * 6].reshape(rows, 6).astype(np.float64)
        t_ref = tl[8, 0]
        print("call: front [start, end]  dc [start, end]  clock [start, end]   (ms, device clock)")
        for r in range(8, min(rows, 16)):
            print("%2d: " % r + "  ".join("[%7.3f, %7.3f]" % ((tl[r, 2 * k] - t_ref) / 1e5, (tl[r, 2 * k + 1] - t_ref) / 1e5) for k in range(3)))
        print("kernel_ms", [round(b.timing_read(w)[0] / max(b.timing_read(w)[1], 1), 3) for w in range(3)])
        b.close()
    elif what.startswith("sleep"):
        import time
        time.sleep(float(what[5:]))
    elif what == "sync":
        torch.cuda.synchronize()
        import gc; gc.collect(); torch.cuda.empty_cache()
    elif what in ("arena", "one", "proc", "hostproc"):
        import numpy as np
        N = 131072
        b = binding.Batch([(48000, 9600, 5000, 1, 2000, True, N)] * 256)
        if what == "arena":
            arena = b.arena(4)
        elif what == "one":
            arena = b.arena(4)
            arena[0, :, :] = 0.25
            assert b.submit(0, [N] * 256) == 0
            b.collect(copy=False)
        elif what == "hostproc":
            x = np.zeros((256, N), dtype=np.complex64)
            x[:] = 0.25
            b.process([x[i] for i in range(256)])
        else:
            x = torch.zeros(256, 2 * N, device="cuda")
            b.process_device(x.data_ptr(), N, [N] * 256, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        b.close()
        print(what, "done")
    elif what == "e2e":
        r = bench.end_to_end(binding, siggen, 256, 131072)
        print("end_to_end:", r["value"])
