"""The companion grid's footprint (DESIGN.md, K3 'Company'): python tools/company_sweep.py [channels] [chunk]
For grids of 0 / 64 / 256 / 1024 / 4096 one-wave workgroups and 1 / 4 / 16 / 64 s_nop 7 between two vector instructions of a
companion wave: ms per step, the three stages' kernel times, the clock stage's cycles per symbol (from its in-kernel stamps)
and the board's average power over the timed region (hwmon power1_average, sampled every 20 ms -- not under a profiler).
Every cell is a fresh batch in this process (SDRM_K3_COMPANY is read when a batch is created); the self-calibration is off."""
import glob, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SDRM_AUTOTUNE"] = "0"
import torch
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen
Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
K = 60
cfg = (48000, 9600, 5000, 1, 2000, True, N)
base = np.stack([siggen.gmsk_channel(i, 2 * N) for i in range(8)]).view(np.float32)
bt = torch.from_numpy(base).cuda()
x = torch.empty((Cn, 4 * N), dtype=torch.float32, device="cuda")
for c in range(Cn):
    x[c] = torch.roll(bt[c % 8], shifts=2 * 977 * (c // 8))
st = torch.cuda.current_stream().cuda_stream
lens = (binding.C.c_size_t * Cn)(*([N] * Cn))
power_files = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")) or sorted(glob.glob("/sys/class/hwmon/hwmon*/power1_average"))


def read_power():
    try:
        return max(int(open(p).read()) for p in power_files) / 1e6
    except Exception:
        return float("nan")


print("power source: %s" % (power_files if power_files else "none readable"))
print("%-6s %-5s %9s %8s %8s %8s %9s" % ("grid", "nops", "ms/step", "front", "dc", "clock", "watts"))
for grid in (0, 64, 256, 1024, 4096):
    for nops in ((1,) if grid == 0 else (1, 4, 16, 64)):
        os.environ["SDRM_K3_COMPANY"] = "%d,0,1000000,%d" % (grid, nops)
        b = binding.Batch([cfg] * Cn)
        assert b.code == 0
        for i in range(8):
            b.process_device(x.data_ptr() + (i % 2) * N * 8, 2 * N, lens, st)
        torch.cuda.synchronize()
        b.timing_enable(True)
        watts, stop = [], threading.Event()

        def sampler():
            while not stop.is_set():
                watts.append(read_power())
                time.sleep(0.02)
        th = threading.Thread(target=sampler)
        th.start()
        t0 = time.perf_counter()
        for i in range(K):
            b.process_device(x.data_ptr() + (i % 2) * N * 8, 2 * N, lens, st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        stop.set()
        th.join()
        km = [b.timing_read(w)[0] / max(b.timing_read(w)[1], 1) for w in range(3)]
        b.close()
        print("%-6d %-5d %9.3f %8.3f %8.3f %8.3f %9.1f" % (grid, nops, dt * 1e3, km[0], km[1], km[2], float(np.nanmean(watts)) if watts else float("nan")), flush=True)
