"""Per-kernel device time with the stages NOT overlapped (host sync after every call).
python tools/stage_times.py [channels] [chunk]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen
Cn = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
cfg = (48000, 9600, 5000, 1, 2000, True, N)
base = np.stack([siggen.gmsk_channel(i, 2 * N) for i in range(8)])
x = torch.from_numpy(np.tile(base, (Cn // 8, 1)).view(np.float32)).cuda()
b = binding.Batch([cfg] * Cn)
st = torch.cuda.current_stream().cuda_stream
for i in range(2):
    b.process_device(x.data_ptr() + (i % 2) * N * 8, 2 * N, [N] * Cn, st); b.sync()
b.timing_enable(True)
for i in range(6):
    b.process_device(x.data_ptr() + (i % 2) * N * 8, 2 * N, [N] * Cn, st); b.sync()
names = ["front", "dc", "clock"]
out = []
for w in range(3):
    ms, n = b.timing_read(w)
    out.append("%s %.3f ms" % (names[w], ms / max(n, 1)))
print("channels %d chunk %d (serialised): " % (Cn, N) + ", ".join(out))
