"""Diagnostic: where do K3's cycles go (staging vs symbol loops)?  python tools/k3_probe.py [channels]"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen
Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = 131072
cfg = (48000, 9600, 5000, 1, 2000, True, N)
D = int(os.environ.get('DISTINCT', '8'))  # distinct waveforms (the rest are copies): 8 = little LDS bank conflict
base = np.stack([siggen.gmsk_channel(i, 2 * N) for i in range(D)])
x = torch.from_numpy(np.tile(base, (Cn // D, 1)).view(np.float32)).cuda()
b = binding.Batch([cfg] * Cn)
L = binding.load()
L.sdrm_batch_k3_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
NCALLS = int(os.environ.get("CALLS", "4"))  # CALLS=12 MID=8: look at the 8th of 12 back-to-back (overlapped) calls
L.sdrm_batch_k3_stamps(b.h, int(os.environ.get("MID", "1")), None, 0)
st = torch.cuda.current_stream().cuda_stream
for i in range(NCALLS):
    b.process_device(x.data_ptr() + (i % 2) * N * 8, 2 * N, [N] * Cn, st)
    if os.environ.get("BLOCKING"):  # one call at a time: what the in-call hand-off (SDRM_HANDOFF) is for
        b.sync()
if os.environ.get("LOAD"):  # keep the rest of the chip busy (fp32 GEMMs on a side stream) while the last call runs
    side = torch.cuda.Stream()
    a = torch.randn(8192, 8192, device="cuda")
    with torch.cuda.stream(side):
        for _ in range(int(os.environ["LOAD"])):
            a = (a @ a) * 1e-4
torch.cuda.synchronize()
lanes = int(os.environ.get('SDRM_K3_LANES', '16' if Cn <= 1280 else ('32' if Cn <= 2560 else '64')).split('x')[0])
slots = (Cn + 15) // 16  # record slots (sized for the smallest workgroup shape)
waves = (Cn + lanes - 1) // lanes
out = np.zeros(slots * 4 + 24, dtype=np.uint64)
L.sdrm_batch_k3_stamps(b.h, 1, out.ctypes.data, slots)
k1 = [int(v) for v in out[slots * 4:slots * 4 + 5]]
k2 = [int(v) for v in out[slots * 4 + 8:slots * 4 + 18]]
k3p = [int(v) for v in out[slots * 4 + 18:slots * 4 + 22]]
out = out[:slots * 4].reshape(slots, 4)
if k1[4]:
    print("K1 per workgroup (cycles): load %.0f, lpf1 %.0f, quad %.0f, lpf2+store %.0f  (%d workgroups)" % (
        k1[0] / k1[4], k1[1] / k1[4], k1[2] / k1[4], k1[3] / k1[4], k1[4]))
starts = [int(out[w][0]) >> 32 for w in range(waves)]
print("loop start of each consumer wave, us after the first: %s" % " ".join("%.0f" % ((s - min(starts)) / 100.0) for s in starts))
hw = [(int(out[w][3]) >> 32) & 0xffff for w in range(waves)]
hwp = [(int(out[w][3]) >> 48) & 0xffff for w in range(waves)]
for w in range(waves):
    out[w][3] = int(out[w][3]) & 0xffffffff
print("consumer/producer SIMD, CU, SE per workgroup: %s" % " ".join("%d/%d:%d:%d" % ((hw[w] >> 4) & 3, (hwp[w] >> 4) & 3, (hw[w] >> 8) & 15, (hw[w] >> 13) & 7) for w in range(waves)))
for w in range(min(waves, 2)):
    stg, drn, packed, it = [int(v) for v in out[w]]
    stg &= 0xffffffff
    nb, ticks = packed & 0xffffffff, packed >> 32
    print("wave %d: wait-for-producer %.0f cyc/step, loops %.0f cyc/step, %d steps, %.1f iterations/step, %.0f cyc/iteration, "
          "%.3f ms at %.0f MHz" % (w, stg / nb, drn / nb, nb, it / nb, drn / max(it, 1), ticks / 1e5, (stg + drn) / max(ticks, 1) * 100))
if k3p[0]:
    nb0 = (int(out[0][2]) & 0xffffffff) or 1
    print("staging wave of workgroup 0, cycles per step: ring writes %.0f, int8 conversion %.0f, next block's fetch %.0f, at the barrier %.0f" % tuple(v / nb0 for v in k3p))
print("cycles per iteration, all %d consumer waves: %s" % (waves, " ".join("%.0f" % (int(out[w][1]) / max(int(out[w][3]), 1)) for w in range(waves))))
if k2[9]:
    print("K2 workgroup 0: %d iterations, %.0f cycles each; busy per iteration: chain %.0f, feeder %.0f, stages %.0f %.0f %.0f, output %.0f" % (
        k2[9], k2[8] / k2[9], *[k2[w] / k2[9] for w in range(6)]))
    if k2[6]:
        print("   stage 0 segments (debug build): reads %.0f, quotients %.0f, ring store %.0f, terms + store %.0f" % (
            (k2[6] & 0xffffffff) / k2[9], (k2[6] >> 32) / k2[9], (k2[7] & 0xffffffff) / k2[9], (k2[7] >> 32) / k2[9]))
