"""What the companion grid beside the clock stage changes (review of round 4, item 6): one session, 256 channels x 131072 samples,
pipelined calls; per variant the step, the clock stage's time (HIP events) and, from the in-kernel stamps of one call, what a
consumer wave sees: shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime) of its symbol loops -> cycles per symbol and the
clock those cycles ran at.  python tools/company_mechanism.py   (each variant is a fresh process: the switches are read at creation)"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CELL = r'''
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(HERE))
import torch
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen
Cn, N = 256, 131072
base = np.stack([siggen.gmsk_channel(i, 2 * N) for i in range(8)])
x = torch.from_numpy(np.tile(base, (Cn // 8, 1)).view(np.float32)).cuda()
b = binding.Batch([(48000, 9600, 5000, 1, 2000, True, N)] * Cn)
st = torch.cuda.current_stream().cuda_stream
lens = (C.c_size_t * Cn)(*([N] * Cn))
def call(i): b.process_device(x.data_ptr() + (i % 2) * N * 8, 2 * N, lens, st)
for i in range(12): call(i)
torch.cuda.synchronize()
b.timing_enable(True)
t0 = time.perf_counter()
for i in range(48): call(i)
torch.cuda.synchronize()
step = (time.perf_counter() - t0) / 48 * 1e3
km = [b.timing_read(w) for w in range(3)]
b.timing_enable(False)
L = binding.load()
L.sdrm_batch_k3_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
L.sdrm_batch_k3_stamps(b.h, 8, None, 0)   # the 8th call from now writes the stamps
for i in range(12): call(i)
torch.cuda.synchronize()
out = np.zeros(16 * 4 + 24, dtype=np.uint64)
L.sdrm_batch_k3_stamps(b.h, 1, out.ctypes.data, 16)
w = out[:64].reshape(16, 4)
cyc = np.array([int(r[1]) for r in w], dtype=np.float64)                      # cycles in the symbol loops
wait = np.array([int(r[0]) & 0xffffffff for r in w], dtype=np.float64)        # cycles at the hand-over
ticks = np.array([int(r[2]) >> 32 for r in w], dtype=np.float64)              # 100 MHz ticks of the whole loop
its = np.array([int(r[3]) & 0xffffffff for r in w], dtype=np.float64)
print("%-34s step %.3f ms  clock stage %.3f ms | consumer waves: %.1f cycles per symbol (min %.1f max %.1f), %.0f cycles per step at the hand-over, "
      "loop %.3f ms, shader clock seen %.0f MHz" % (os.environ.get("LABEL", ""), step, km[2][0] / max(km[2][1], 1), (cyc / its).mean(), (cyc / its).min(), (cyc / its).max(),
      (wait / 512).mean(), ticks.mean() / 1e5, ((cyc + wait) / ticks).mean() * 100))
if os.environ.get("PER_WAVE"):
    print("    per workgroup (workgroup i sits on XCD i % 8), cycles per symbol: " + " ".join("%.0f" % v for v in cyc / its))
'''
if len(sys.argv) > 1 and sys.argv[1] == "xcd":
    # companions on every CU of XCD 0 and on ONE CU (31) of each other XCD (an XCD cannot be left out: no bit = all of its CUs)
    words = [0] * 8
    for cu in range(32):
        bit = cu * 8 + 0
        words[bit // 32] |= 1 << (bit % 32)
    for x in range(1, 8):
        bit = 31 * 8 + x
        words[bit // 32] |= 1 << (bit % 32)
    mask = ",".join("%x" % w for w in words)
    for rep in range(2):
        for label, env in [("no companion grid", {"SDRM_K3_COMPANY": "0,0,0"}), ("4096 x 1 (default)", {"SDRM_K3_COMPANY": "4096,0,100000,1"}),
                           ("4096 x 1 on XCD 0 (+ CU 31 elsewhere)", {"SDRM_K3_COMPANY": "4096,0,100000,1", "SDRM_K3_COMPANY_CUMASK": mask})]:
            subprocess.call([sys.executable, "-c", "HERE=%r\n" % HERE + CELL], env=dict(os.environ, SDRM_AUTOTUNE="0", LABEL=label, PER_WAVE="1", **env))
    sys.exit(0)
variants = [("no companion grid", {"SDRM_K3_COMPANY": "0,0,0"}),
            ("4096 x 1 (default)", {"SDRM_K3_COMPANY": "4096,0,100000,1"}),
            ("4096 x 1, CUs 0-1 of every XCD", {"SDRM_K3_COMPANY": "4096,0,100000,1", "SDRM_K3_COMPANY_CUMASK": "ffff"}),
            ("4096 x 1, CUs 16-31 of every XCD", {"SDRM_K3_COMPANY": "4096,0,100000,1", "SDRM_K3_COMPANY_CUMASK": "0,0,0,0,ffffffff,ffffffff,ffffffff,ffffffff"}),
            ("4096 x 1, CUs 2-31 of every XCD", {"SDRM_K3_COMPANY": "4096,0,100000,1", "SDRM_K3_COMPANY_CUMASK": "ffff0000,ffffffff,ffffffff,ffffffff,ffffffff,ffffffff,ffffffff,ffffffff"}),
            ("256 x 1 (one per CU)", {"SDRM_K3_COMPANY": "256,0,100000,1"}),
            ("4096 x 64 (sparse)", {"SDRM_K3_COMPANY": "4096,0,100000,64"})]
for rep in range(2):
    for label, env in variants:
        e = dict(os.environ, SDRM_AUTOTUNE="0", LABEL=label, **env)
        subprocess.call([sys.executable, "-c", "HERE=%r\n" % HERE + CELL], env=e)
