"""Channel sharding across the GPUs of a node.

Every RX channel is an independent stream, so the path shards by channel with no data-path collective: rank r owns
the contiguous block of channels [r*per_rank, (r+1)*per_rank).  The only collective is the fan-out of the channel
table (the fsk_demod_create() arguments of every channel) from rank 0 -- RCCL on GPUs (backend "nccl"), gloo in the
CPU tests."""
import torch
import torch.distributed as dist

FIELDS = 7  # sampling_freq, baud_rate, deviation, decimation, transition_width, use_dc_block, max_input_buffer_length


def encode(cfgs):
    return torch.tensor([[int(v) for v in c] for c in cfgs], dtype=torch.int64).reshape(len(cfgs), FIELDS)


def decode(table):
    return [(int(r[0]), int(r[1]), int(r[2]), int(r[3]), int(r[4]), bool(r[5]), int(r[6])) for r in table.tolist()]


def shard_range(total, world, rank):
    """contiguous blocks; the first `total % world` ranks take one extra channel"""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def filter_lengths(cfg):
    """(T1, T2) of a channel: the reference's tap-count rule (src/dsp/lpf_taps.c:33-40, `53 fs / (22 tw)` made odd) applied
    to the two filters fsk_demod_create designs (src/dsp/fsk_demod.c:36-47: LPF1 with a transition width of a tenth of the
    Carson bandwidth -- the reference multiplies by the float constant 0.1f --, LPF2 with the request's)."""
    fs, baud, dev, _decim, tw = int(cfg[0]), int(cfg[1]), int(cfg[2]), int(cfg[3]), int(cfg[4])
    carson = abs(dev) + baud / 2.0
    tw1 = int(float(torch.tensor(0.1, dtype=torch.float32)) * carson)

    def ntaps(width):
        n = int(53.0 * fs / (22.0 * max(width, 1)))
        return n if n % 2 else n + 1
    return ntaps(tw1), ntaps(tw)


def channel_cost(cfg):
    """Relative work of one channel per second of signal: the front-end's multiply-adds,
    fs * (4 T1 + 2 T2 / d) -- complex LPF1 at the input rate plus real LPF2 at the decimated one."""
    t1, t2 = filter_lengths(cfg)
    return int(cfg[0]) * (4.0 * t1 + 2.0 * t2 / max(int(cfg[3]), 1))


def shard_by_cost(cfgs, world):
    """Contiguous channel blocks, one per rank, with the LARGEST block cost as small as contiguity allows (a mixed-rate
    batch: a 240 kHz / 397-tap channel costs ten times a 48 kHz / 207-tap one, so equal counts are not equal work).
    Classic linear partition: binary search on the bound, greedy fill; then the cuts are moved right while that lowers
    the heavier neighbour without raising the bound, which evens out the light ranks.  Returns [(lo, hi)] * world."""
    n = len(cfgs)
    cost = [channel_cost(c) for c in cfgs]
    if world <= 1 or n == 0:
        return [(0, n)] + [(n, n)] * (max(world, 1) - 1)

    def cuts_for(bound):
        cuts, acc = [0], 0.0
        for i, w in enumerate(cost):
            if acc + w > bound and acc > 0.0:
                cuts.append(i)
                acc = 0.0
            acc += w
        cuts.append(n)
        return cuts
    lo_b, hi_b = max(cost), sum(cost)
    for _ in range(60):
        mid = 0.5 * (lo_b + hi_b)
        if len(cuts_for(mid)) - 1 <= world:
            hi_b = mid
        else:
            lo_b = mid
    cuts = cuts_for(hi_b)
    while len(cuts) - 1 < world:  # fewer blocks than ranks: split the heaviest block that can be split
        sums = [sum(cost[cuts[i]:cuts[i + 1]]) for i in range(len(cuts) - 1)]
        order = sorted(range(len(sums)), key=lambda i: -sums[i])
        for i in order:
            if cuts[i + 1] - cuts[i] > 1:
                half, acc, j = sums[i] / 2.0, 0.0, cuts[i]
                while j < cuts[i + 1] - 1 and acc + cost[j] <= half:
                    acc += cost[j]
                    j += 1
                cuts.insert(i + 1, max(j, cuts[i] + 1))
                break
        else:
            cuts.insert(len(cuts) - 1, cuts[-1])  # nothing left to split: empty rank
    # greedy fill leaves the last rank light: shift work towards it while the maximum does not grow
    prefix = [0.0]
    for w in cost:
        prefix.append(prefix[-1] + w)
    moved = True
    while moved:
        moved = False
        for i in range(len(cuts) - 2, 0, -1):
            left = prefix[cuts[i]] - prefix[cuts[i - 1]]
            right = prefix[cuts[i + 1]] - prefix[cuts[i]]
            if cuts[i] - cuts[i - 1] > 1:
                w = cost[cuts[i] - 1]
                if max(left - w, right + w) < max(left, right):
                    cuts[i] -= 1
                    moved = True
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


class Shard:
    """This rank's part of a fanned-out channel table: its configs and the global range [lo, hi) they came from.
    Unpacks as (cfgs, lo, hi).  The NCO fan-out takes the SAME object, so that the batches a rank keeps are always
    those of the channels it was given (a cost-balanced table has other cuts than an equal-count one)."""

    def __init__(self, cfgs, lo, hi, total, world, rank, balance):
        self.cfgs, self.lo, self.hi = cfgs, lo, hi
        self.total, self.world, self.rank, self.balance = total, world, rank, balance

    def __iter__(self):
        return iter((self.cfgs, self.lo, self.hi))

    def __len__(self):
        return self.hi - self.lo


def fanout_configs(cfgs_rank0, total, device="cpu", balance="count"):
    """Broadcast the channel table from rank 0; return this rank's Shard (unpacks as configs, lo, hi).  balance="cost":
    shards of equal front-end work (shard_by_cost, computed by every rank from the same broadcast table) instead of
    equal counts."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    table = torch.zeros((total, FIELDS), dtype=torch.int64, device=device)
    if rank == 0:
        table.copy_(encode(cfgs_rank0))
    if dist.is_initialized():  # (a world of one included: the collective then still runs through the backend -- RCCL on a GPU)
        dist.broadcast(table, src=0)
    if balance == "cost":
        lo, hi = shard_by_cost(decode(table.cpu()), world)[rank]
    elif balance == "count":
        lo, hi = shard_range(total, world, rank)
    else:
        raise ValueError("balance must be 'count' or 'cost'")
    return Shard(decode(table[lo:hi].cpu()), lo, hi, total, world, rank, balance)


def fanout_nco_segments(segments_rank0, shard, device="cpu", as_array=False, capacity=None):
    """Doppler pre-correction at node scale: rank 0 runs the orbit model and the planner for every channel
    of the node and holds the call's NCO batches as (global_channel, len, freq_hz), grouped by channel -- a list of
    tuples or an (n, 3) int64 array; a broadcast (RCCL on GPUs, KB-sized) gives every rank the batches of its own
    channels, with the channel index rebased to the rank's shard.  `shard` is what fanout_configs returned on this rank
    (required: the batches kept are those of exactly the channels the rank was given).  With `capacity` (rows, agreed
    by all ranks at setup) the count travels in row 0 of ONE fixed-size table -- one collective per call; without it a
    count is broadcast first.  Returns [(local_channel, len, freq_hz)], or with as_array the same as an (m, 3) int64
    numpy array (what binding.Batch.process_device_nco takes without a Python loop: this runs before every call)."""
    import numpy as np
    if not isinstance(shard, Shard):
        raise TypeError("fanout_nco_segments needs the Shard returned by fanout_configs")
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    rows0 = None
    if rank == 0:
        rows0 = np.ascontiguousarray(np.asarray(segments_rank0, dtype=np.int64).reshape(-1, 3))
    if capacity is not None:
        table = torch.zeros((capacity + 1, 3), dtype=torch.int64, device=device)
        if rank == 0:
            # too many batches for the agreed table: the failure travels THROUGH the collective (count -1 in row 0, the real
            # count beside it) so that every rank raises after the broadcast -- raising on rank 0 alone would leave the others
            # blocked in it until the backend's timeout
            fits = len(rows0) <= capacity
            table[0, 0] = len(rows0) if fits else -1
            table[0, 1] = len(rows0)
            if fits and len(rows0):
                table[1:1 + len(rows0)].copy_(torch.from_numpy(rows0))
        if dist.is_initialized():
            dist.broadcast(table, src=0)
        host = table.cpu().numpy()
        if int(host[0, 0]) < 0:
            raise ValueError("%d NCO batches exceed the agreed capacity of %d" % (int(host[0, 1]), capacity))
        rows = host[1:1 + int(host[0, 0])]
    else:
        count = torch.zeros(1, dtype=torch.int64, device=device)
        if rank == 0:
            count[0] = len(rows0)
        if dist.is_initialized():
            dist.broadcast(count, src=0)
        n = int(count.item())
        table = torch.zeros((max(n, 1), 3), dtype=torch.int64, device=device)
        if rank == 0 and n:
            table[:n].copy_(torch.from_numpy(rows0))
        if dist.is_initialized():
            dist.broadcast(table, src=0)
        rows = table[:n].cpu().numpy()
    mine = rows[(rows[:, 0] >= shard.lo) & (rows[:, 0] < shard.hi)].copy()
    mine[:, 0] -= shard.lo
    return mine if as_array else [(int(c), int(ln), int(f)) for c, ln, f in mine.tolist()]
