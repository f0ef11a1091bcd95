import os, sys, json
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import orc, sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding
G = os.path.join(R, "tests", "golden")
D = json.load(open(os.path.join(G, "doppler_shifts_lucky7.json")))
iq = np.fromfile(os.path.join(G, "lucky7.cf32"), dtype=np.complex64)
chunk = 47000
g = binding.Batch([(48000, 4800, 5000, 2, 2000, True, chunk)])
pl = binding.DopplerPlanner(48000, lambda k: D["shifts_hz"][min(k, 7)])
o = orc.Doppler(48000, D["shifts_hz"], chunk)
for off in range(0, len(iq), chunk):
    part = iq[off:off + chunk]
    segs = pl.plan(0, len(part))
    g.process_nco([part], segs)
    a = g.last_mixed(0); b = o.process(part.view(np.float32))
    d = np.abs(a - b)
    bad = np.nonzero(d > 1e-6)[0]
    print(off, segs, "first bad", bad[:3], "max", d.max(), "n bad", len(bad))
    if len(bad):
        i = bad[0] // 2
        print("  sample", i, a[2*i:2*i+2], b[2*i:2*i+2], "in", part[i])
