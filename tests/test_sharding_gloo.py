"""N>1 path on CPU: world_size 2 over gloo.  Rank 0 owns the channel table, broadcasts it (RCCL on GPUs), each rank
demodulates only its shard (here: through the host-driven kernel emulation) and the union equals the oracle."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sdrm_pkg
    sdrm_pkg.load()
    from sdr_modem_amd import shard, siggen
    import emu_api
    import orc
    total = 7  # not divisible by 2: ragged shards
    table = None
    if rank == 0:
        table = [(48000, 9600, 5000, 1, 2000, True, 4096)] * 4 + [(48000, 4800, 5000, 2, 2000, False, 4096)] * 3
    cfgs, lo, hi = shard.fanout_configs(table, total)
    ok = len(cfgs) == hi - lo
    e = emu_api.EmuBatch(cfgs)
    ok = ok and e.code == 0
    sigs = [siggen.gmsk_channel(lo + i, 4096, fs=c[0], baud=c[1]) for i, c in enumerate(cfgs)]
    got8, _ = e.process(sigs)
    for i, c in enumerate(cfgs):
        want8, _ = orc.demod_stream(c[:6], sigs[i], 4096)
        ok = ok and np.array_equal(want8, got8[i])
    # Doppler batches for one call, planned on rank 0 for the whole node, fanned out and applied per shard
    n = 4096
    segs0 = None
    if rank == 0:
        segs0 = []
        for c in range(total):
            if c % 2 == 0:  # every other channel is corrected, in two batches
                segs0 += [(c, 1000, 1500 - 100 * c), (c, n - 1000, 1490 - 100 * c)]
    mine = shard.fanout_nco_segments(segs0, total)
    ok = ok and all(0 <= ch < hi - lo for ch, _, _ in mine)
    ok = ok and sorted(set(ch + lo for ch, _, _ in mine)) == [c for c in range(lo, hi) if c % 2 == 0]
    e2 = emu_api.EmuBatch(cfgs)
    got8, _ = e2.process(sigs, segments=mine)
    for i, c in enumerate(cfgs):
        x = sigs[i].view(np.float32)
        if (lo + i) % 2 == 0:
            osc = orc.Nco(1.0, c[0], n)
            x = np.concatenate([osc.multiply(1500 - 100 * (lo + i), x[:2000]), osc.multiply(1490 - 100 * (lo + i), x[2000:])])
        want8, _ = orc.demod_stream(c[:6], x.view(np.complex64), n)
        ok = ok and np.array_equal(want8, got8[i])
    # every rank reports its range; rank 0 checks the shards tile [0, total)
    ranges = [None] * world
    dist.all_gather_object(ranges, (lo, hi, bool(ok)))
    if rank == 0:
        ret["ranges"] = ranges
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_config_fanout():
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    ranges = ret["ranges"]
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == 7
    assert ranges[0][1] - ranges[0][0] == 4
    assert all(r[2] for r in ranges)


def test_shard_range_tiles_everything():
    sys.path.insert(0, ROOT)
    import sdrm_pkg
    sdrm_pkg.load()
    from sdr_modem_amd import shard
    for total in (1, 7, 256, 4096, 4099):
        for world in (1, 2, 4, 8):
            spans = [shard.shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
