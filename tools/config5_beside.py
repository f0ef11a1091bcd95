"""Why BASELINE configs[4]'s mix reads lower inside the bench line than alone (round-3 review, weak item 9): the same config5
run (bench.config5_single) three times in one process -- fresh; right after a 256-channel headline run of 60 steps whose batch
has been closed; and again after two idle seconds.  python tools/config5_beside.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, siggen
import bench
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
N = 131072


def c5(tag):
    r = bench.config5_single(torch, binding, siggen, dev, 256, N, steps=24, verify=False)
    print("%-44s %.3f ms per step, %.0f Msamples/s, kernels %s" % (tag, r["ms_per_step"], r["value"], r["kernel_ms"]), flush=True)


c5("config5, fresh process")
c5("config5 again (second batch of the process)")
rig = bench.Rig(torch, binding, siggen, dev, 0, [(48000, 9600, 5000, 1, 2000, True, N)] * 256, 0, N, 4)
for i in range(60):
    rig.step(i)
torch.cuda.synchronize()
c5("config5 beside the LIVE (idle) headline batch")
rig.close()
c5("config5 after the headline batch was closed")
time.sleep(2.0)
c5("config5 after two idle seconds")
