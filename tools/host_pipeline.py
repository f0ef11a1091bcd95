"""PCIe-inclusive rate of the pipelined host-buffer path (sdrm_batch_arena/_submit/_collect): the arena slots hold
synthetic IQ; every call copies its slot host->device, runs the path and copies the soft bits back.
python tools/host_pipeline.py [channels] [chunk] [calls]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (one HIP runtime per process)
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen


def measure(Cn, N, K, slots=4):
    cfg = (48000, 9600, 5000, 1, 2000, True, N)
    b = binding.Batch([cfg] * Cn)
    arena = b.arena(slots)
    base = np.stack([siggen.gmsk_channel(i, N) for i in range(8)]).view(np.float32)
    for s in range(slots):
        arena[s, :, :2 * N] = np.tile(np.roll(base, 2 * 977 * s, axis=1), (Cn // 8, 1))
    lens = [N] * Cn
    F = 3  # calls kept in flight
    for k in range(4):
        assert b.submit(k % slots, lens) == 0
        b.collect(copy=False)
    t0 = time.perf_counter()
    for k in range(K):
        if k >= F:
            b.collect(copy=False)
        assert b.submit(k % slots, lens) == 0
    for k in range(min(F, K)):
        n = b.collect(copy=False)
    dt = (time.perf_counter() - t0) / K
    b.close()
    return dt, n


if __name__ == "__main__":
    Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    dt, n = measure(Cn, N, K)
    print("pipelined host path: %.2f ms per call of %d x %d samples = %.0f Msamples/s (%.1f GB/s of IQ over the host link), "
          "%d symbols per channel" % (dt * 1e3, Cn, N, Cn * N / dt / 1e6, Cn * N * 8 / dt / 1e9, n[0]))
