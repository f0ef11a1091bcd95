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


def fanout_configs(cfgs_rank0, total, device="cpu"):
    """Broadcast the channel table from rank 0; return (this rank's configs, lo, hi)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    table = torch.zeros((total, FIELDS), dtype=torch.int64, device=device)
    if rank == 0:
        table.copy_(encode(cfgs_rank0))
    if world > 1:
        dist.broadcast(table, src=0)
    lo, hi = shard_range(total, world, rank)
    return decode(table[lo:hi].cpu()), lo, hi


def fanout_nco_segments(segments_rank0, total, device="cpu"):
    """Doppler pre-correction at node scale: rank 0 runs the orbit model and the planner for every channel
    of the node and holds the call's NCO batches as (global_channel, len, freq_hz), grouped by channel; one broadcast
    (a count, then an int64 table -- RCCL on GPUs, KB-sized) gives every rank the batches of its own channels, with
    the channel index rebased to the rank's shard.  Returns this rank's list of (local_channel, len, freq_hz)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    count = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == 0:
        count[0] = len(segments_rank0)
    if world > 1:
        dist.broadcast(count, src=0)
    n = int(count.item())
    table = torch.zeros((max(n, 1), 3), dtype=torch.int64, device=device)
    if rank == 0 and n:
        table[:n].copy_(torch.tensor([[int(c), int(ln), int(f)] for c, ln, f in segments_rank0], dtype=torch.int64))
    if world > 1:
        dist.broadcast(table, src=0)
    lo, hi = shard_range(total, world, rank)
    rows = table[:n].cpu().tolist()
    return [(int(c) - lo, int(ln), int(f)) for c, ln, f in rows if lo <= c < hi]
