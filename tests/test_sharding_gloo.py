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
    part = shard.fanout_configs(table, total)
    cfgs, lo, hi = part
    ok = len(cfgs) == hi - lo == len(part)
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
    mine = shard.fanout_nco_segments(segs0, part)
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
    # cost-balanced shards of a mixed-rate table: every rank derives the same cuts from the broadcast table
    mix = None
    if rank == 0:
        mix = [(240000, 19200, 5000, 5, 2000, True, 4096)] * 6 + [(48000, 1200, 5000, 8, 2000, True, 4096)] * 18
    part_c = shard.fanout_configs(mix, 24, balance="cost")
    cfgs_c, lo_c, hi_c = part_c
    # the NCO fan-out takes the partition itself: with a cost-balanced table every rank keeps the batches of exactly
    # the channels it was given (equal-count cuts would hand rank 1 the batches of channels rank 0 demodulates)
    segs_c = [(c, 4096, 100 * c) for c in range(24)] if rank == 0 else None
    mine_c = shard.fanout_nco_segments(segs_c, part_c)
    one_shot = shard.fanout_nco_segments(np.array(segs_c) if rank == 0 else None, part_c, as_array=True, capacity=40)
    ok = ok and one_shot.tolist() == [list(t) for t in mine_c]  # one fixed-size collective per call: the same batches
    ok = ok and [(ch + lo_c, f) for ch, _, f in mine_c] == [(c, 100 * c) for c in range(lo_c, hi_c)]
    try:
        shard.fanout_nco_segments(segs_c, 24)
        ok = False  # a bare channel count is no partition
    except TypeError:
        pass
    # more batches than the agreed table holds: EVERY rank raises after the collective (rank 0 raising alone would leave
    # the others blocked in the broadcast until the backend's timeout), and the next collective still lines up
    try:
        shard.fanout_nco_segments(segs_c, part_c, as_array=True, capacity=10)
        ok = False
    except ValueError as exc:
        ok = ok and "24 NCO batches exceed the agreed capacity of 10" in str(exc)
    again = shard.fanout_nco_segments(np.array(segs_c) if rank == 0 else None, part_c, as_array=True, capacity=24)
    ok = ok and again.tolist() == one_shot.tolist()
    spans = [None] * world
    dist.all_gather_object(spans, (lo_c, hi_c, sum(shard.channel_cost(c) for c in cfgs_c)))
    ok = ok and spans[0][0] == 0 and spans[0][1] == spans[1][0] and spans[1][1] == 24
    ok = ok and max(s[2] for s in spans) / min(s[2] for s in spans) <= 1.15 and spans[0][1] - spans[0][0] < 12
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


def _worker_node(rank, world, port, ret):
    """BASELINE configs[3] (4096 channels) and configs[4] (2048 mixed-rate channels, cost-balanced) fanned out over `world` ranks"""
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
    ok = True
    report = {}
    n = 2048  # samples per call here: the tables are the node's, the calls are short (CPU emulation)
    for name, total, balance, make in (
            ("configs3", 4096, "count", lambda: [(48000, 9600, 5000, 1, 2000, True, n)] * 4096),
            ("configs4", 2048, "cost", lambda: [(240000, 19200, 5000, 5, 2000, True, n)] * 1024 + [(48000, 1200, 5000, 8, 2000, True, n)] * 1024)):
        part = shard.fanout_configs(make() if rank == 0 else None, total, balance=balance)
        cfgs, lo, hi = part
        ok = ok and len(cfgs) == hi - lo
        # this rank demodulates the first and the last channel of its shard (its whole shard on a GPU)
        spots = sorted(set([0, len(cfgs) - 1])) if cfgs else []
        if spots:
            e = emu_api.EmuBatch([cfgs[i] for i in spots])
            sigs = [siggen.gmsk_channel(lo + i, n, fs=cfgs[i][0], baud=cfgs[i][1]) for i in spots]
            got8, _ = e.process(sigs)
            for k, i in enumerate(spots):
                ok = ok and np.array_equal(orc.demod_stream(cfgs[i][:6], sigs[k], n)[0], got8[k])
        # the call's Doppler batches: three per channel, planned on rank 0 for the node, one fixed-size collective
        segs = np.array([(c, ln, -10000 + 7 * c + 100 * k) for c in range(total) for k, ln in enumerate((700, 700, n - 1400))],
                        dtype=np.int64) if rank == 0 else None
        mine = shard.fanout_nco_segments(segs, part, as_array=True, capacity=3 * total)
        ok = ok and len(mine) == 3 * (hi - lo) and (len(mine) == 0 or (mine[0, 0] == 0 and mine[-1, 0] == hi - lo - 1))
        ok = ok and all(int(f) == -10000 + 7 * (int(c) + lo) + 100 * (j % 3) for j, (c, _, f) in enumerate(mine.tolist()))
        try:  # more batches than agreed: every rank raises, behind the collective
            shard.fanout_nco_segments(segs, part, as_array=True, capacity=total)
            ok = False
        except ValueError as exc:
            ok = ok and "exceed the agreed capacity" in str(exc)
        again = shard.fanout_nco_segments(segs, part, as_array=True, capacity=3 * total)  # ... and the next one lines up
        ok = ok and np.array_equal(again, mine)
        spans = [None] * world
        dist.all_gather_object(spans, (lo, hi, sum(shard.channel_cost(c) for c in cfgs)))
        report[name] = spans
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    if rank == 0:
        ret["report"] = report
        ret["ok"] = flags
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("world", [4, 8])
def test_node_sized_tables_over_four_and_eight_ranks(world):
    """SURVEY 8e at the node's size, on CPU over gloo (the RCCL leg needs the 8-GPU node): BASELINE configs[3]'s 4096 channels in
    equal blocks and configs[4]'s 2048 mixed-rate channels in cost-balanced blocks, the channel table broadcast from rank 0,
    per-call Doppler batches fanned out with the same partition through one fixed-size collective (and its overflow raised on
    every rank).  Blocks tile the table, the cost per rank is within 15 %, spot channels of every rank match the oracle."""
    port = 31500 + (os.getpid() % 2000) + world
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_node, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret["ok"]), ret["ok"]
    for name, total in (("configs3", 4096), ("configs4", 2048)):
        spans = ret["report"][name]
        assert spans[0][0] == 0 and spans[-1][1] == total and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        costs = [s[2] for s in spans]
        assert max(costs) / min(costs) <= 1.15, (name, world, costs)
    assert [s[1] - s[0] for s in ret["report"]["configs3"]] == [4096 // world] * world
    counts4 = [s[1] - s[0] for s in ret["report"]["configs4"]]
    assert counts4[0] < 2048 // world < counts4[-1]  # the heavy channels come first: fewer of them per rank


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


def test_cost_balanced_shards_of_the_config5_mix():
    """BASELINE configs[4]: 240 kHz / 19200 baud and 48 kHz / 1200 baud channels cost 5-10x apart; shards are balanced by
    fs * (4 T1 + 2 T2 / d), contiguous, and tile the table."""
    sys.path.insert(0, ROOT)
    import sdrm_pkg
    sdrm_pkg.load()
    from sdr_modem_amd import shard
    assert shard.filter_lengths((48000, 9600, 5000, 1, 2000, True, 1)) == (117, 57)
    assert shard.filter_lengths((240000, 19200, 5000, 1, 2000, True, 1)) == (397, 289)
    assert shard.filter_lengths((48000, 1200, 5000, 1, 2000, True, 1)) == (207, 57)
    assert shard.filter_lengths((192000, 40000, 5000, 1, 2000, True, 1)) == (185, 231)
    for decims in ((5, 8), (1, 1)):
        for total in (4096, 512, 100):
            half = total // 2
            cfgs = [(240000, 19200, 5000, decims[0], 2000, True, 131072)] * half + \
                   [(48000, 1200, 5000, decims[1], 2000, True, 131072)] * (total - half)
            for world in (2, 4, 8):
                spans = shard.shard_by_cost(cfgs, world)
                assert spans[0][0] == 0 and spans[-1][1] == total
                assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
                costs = [sum(shard.channel_cost(c) for c in cfgs[a:b]) for a, b in spans]
                # 100 channels over 8 ranks: one heavy channel is a seventh of a rank's share, granularity bounds the balance
                assert max(costs) / min(costs) <= (1.15 if total >= 512 else 1.35), (decims, total, world, costs)
                by_count = [sum(shard.channel_cost(c) for c in cfgs[slice(*shard.shard_range(total, world, r))]) for r in range(world)]
                assert max(by_count) / min(by_count) > 3.0  # what equal counts would have given
    # uniform tables fall back to equal counts; more ranks than channels leaves empty shards at the end
    uni = [(48000, 9600, 5000, 1, 2000, True, 4096)] * 4096
    assert shard.shard_by_cost(uni, 8) == [(512 * r, 512 * (r + 1)) for r in range(8)]
    assert shard.shard_by_cost(uni[:3], 8)[:3] == [(0, 1), (1, 2), (2, 3)]


def _worker_alone(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sdrm_pkg
    sdrm_pkg.load()
    from sdr_modem_amd import shard
    calls = {"n": 0}
    real = dist.broadcast

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    dist.broadcast = counting
    table = [(48000, 9600, 5000, 1, 2000, True, 4096)] * 3 + [(240000, 19200, 5000, 5, 2000, True, 4096)] * 2
    part = shard.fanout_configs(table, len(table))
    cost = shard.fanout_configs(table, len(table), balance="cost")
    rows = np.array([(c, 4096, 10 * c) for c in range(len(table))], dtype=np.int64)
    fixed = shard.fanout_nco_segments(rows, part, as_array=True, capacity=16)
    loose = shard.fanout_nco_segments(rows, part, as_array=True)
    dist.broadcast = real
    ret["alone"] = (part.cfgs == table and (part.lo, part.hi) == (0, len(table)) and len(cost) == len(table)
                    and np.array_equal(fixed, rows) and np.array_equal(loose, rows), calls["n"])
    dist.destroy_process_group()


def test_a_world_of_one_still_runs_its_collectives():
    """shard.py broadcasts whenever a process group exists -- a world of one included -- so that on a one-GPU box the rank code's
    collectives really go through the backend (RCCL in tests/test_gpu_rccl.py, gloo here): five broadcasts for two table fan-outs, one
    fixed-capacity and one count-then-rows NCO fan-out."""
    port = 31500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_alone, args=(1, port, ret), nprocs=1, join=True)
    ok, broadcasts = ret["alone"]
    assert ok and broadcasts == 5
