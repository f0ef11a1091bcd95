"""debug: the same calls through a batch with the in-call hand-off and one without; where do the soft bits differ?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen
channels = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 4
keep = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
cfg = (48000, 9600, 5000, 1, 2000, True)
sig = siggen.gmsk_batch(min(channels, 32), 2 * n)
rows = np.stack([sig[c % len(sig)] for c in range(channels)])
x = torch.from_numpy(rows.view(np.float32)).to("cuda:0")
os.environ["SDRM_HANDOFF"] = "1"
a = binding.Batch([cfg + (n,)] * channels, device=0, keep_soft=keep)
os.environ["SDRM_HANDOFF"] = "0"
b = binding.Batch([cfg + (n,)] * channels, device=0, keep_soft=keep)
st = torch.cuda.current_stream().cuda_stream
lens = (binding.C.c_size_t * channels)(*([n] * channels))
for i in range(calls):
    for q in (a, b):
        q.process_device(x.data_ptr() + (i % 2) * n * 8, 2 * n, lens, st); q.sync()
    da, la = a.fetch(n); db, lb = b.fetch(n)
    bad = [c for c in range(channels) if la[c] != lb[c] or not np.array_equal(da[c][:la[c]], db[c][:lb[c]])]
    print("call %d: %d channels differ" % (i, len(bad)), bad[:20])
    for c in bad[:6]:
        m = min(la[c], lb[c])
        d = np.nonzero(da[c][:m] != db[c][:m])[0]
        print("   channel %d: lens %d / %d, %d symbols differ, first at %s, last at %s" % (c, la[c], lb[c], len(d), d[:5], d[-3:]))
print("wild calls", a.wild_calls(), b.wild_calls())
