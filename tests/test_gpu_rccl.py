"""`-m gpu`: RCCL on gfx950, as far as one GPU allows.  The pool's boxes have one device, so no scaling curve can be measured here
(8-GPU runs are the driver's); what CAN be shown is that the rank code's collectives execute through RCCL on device tensors:
`init_process_group("nccl", world_size=1, device_id=cuda:0)`, the channel table's broadcast (shard.fanout_configs), the per-call
NCO batch fan-out in its fixed-capacity form (shard.fanout_nco_segments), bench.py's all-reduce and barrier -- and that the shard
such a rank is given demodulates to the oracle's bits.  In a child process: one process group per process.
Reference: none (sdr-modem is one process, one thread per client: src/dsp_worker.c:188); partitioning per SURVEY 8e."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, shard, siggen
import orc
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", world_size=1, rank=0, device_id=dev)
assert dist.get_backend() == "nccl"
N = 16384
table = [(48000, 9600, 5000, 1, 2000, True, N)] * 5 + [(48000, 4800, 5000, 2, 2000, False, N)] * 3
part = shard.fanout_configs(table, len(table), device=dev)              # broadcast of a device tensor through RCCL
assert (part.lo, part.hi) == (0, len(table)) and part.cfgs == table
cost = shard.fanout_configs(table, len(table), device=dev, balance="cost")
assert len(cost) == len(table)
rows = np.array([(c, n, 100 * c - 300 + 50 * k) for c in range(len(table)) for k, n in enumerate((6000, N - 6000))], dtype=np.int64)
mine = shard.fanout_nco_segments(rows, part, device=dev, as_array=True, capacity=4 * len(table))
assert np.array_equal(mine, rows)
try:
    shard.fanout_nco_segments(rows, part, device=dev, as_array=True, capacity=3)   # the overflow travels through the collective
    raise SystemExit("capacity overflow not raised")
except ValueError:
    pass
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)                                # bench.py's max-over-ranks
dist.barrier()
assert float(t.item()) == 1.25
# the shard this rank was given, one call with its NCO batches, against the oracle
sig = [siggen.gmsk_channel(40 + c, N, fs=cf[0], baud=cf[1]) for c, cf in enumerate(part.cfgs)]
b = binding.Batch(part.cfgs, device=0)
assert b.code == 0
got = b.process_nco(sig, [(int(c), int(n), int(f)) for c, n, f in mine])
for c, cf in enumerate(part.cfgs):
    nco, fsk = orc.Nco(1.0, cf[0], N), orc.Fsk(*cf)
    x, pos, parts = sig[c].view(np.float32), 0, []
    for _, n, f in rows[rows[:, 0] == c]:
        parts.append(nco.multiply(int(f), x[2 * pos:2 * (pos + int(n))]))
        pos += int(n)
    want, _ = fsk.process(np.concatenate(parts))
    assert np.array_equal(got[c], want), c
b.close()
dist.destroy_process_group()
print("RCCL ok: nccl", ".".join(str(v) for v in torch.cuda.nccl.version()))
'''


def test_rccl_initialises_and_runs_the_rank_codes_collectives_on_the_device():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0",
                "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    out = subprocess.run([sys.executable, "-c", RANK % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "RCCL ok: nccl" in out.stdout
