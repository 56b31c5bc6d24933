"""Multi-GPU split of the demod workload.

Channels are independent, so the hot path has NO collective: every rank demodulates its own
channels out of its own HBM. The split is by SF bucket (a launch is uniform in N), each
bucket cut into `world_size` contiguous ranges; the at most world_size-1 left-over channels
of a bucket go to the ranks that currently hold the fewest bytes (weight 8*2^SF+14 per
symbol). torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only after the
run, to gather the int16 symbols / counters.
"""
import numpy as np


def bytes_per_symbol(sf, fft_out=False, dec_out=False):
    """Algorithmic HBM bytes of one demodulated symbol window:
    8*2^SF of cf32 IQ + 2 (uint16 symbol) + 12 (power, powerAvg, fIndex); each optional debug
    output adds another 8*2^SF."""
    n = 8 << sf
    return n + 14 + (n if fft_out else 0) + (n if dec_out else 0)


def shard_channels(sfs, world_size):
    """Assign channel c (spreading factor sfs[c]) to a rank.

    Returns a list of `world_size` int64 arrays of channel indices; within a rank the channels
    are ordered by SF, then by channel number, so each rank runs one launch per SF bucket.
    """
    sfs = np.asarray(sfs, dtype=np.int64).reshape(-1)
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    parts = [[] for _ in range(world_size)]
    load = np.zeros(world_size, dtype=np.int64)
    for sf in sorted(set(sfs.tolist()), reverse=True):          # big windows first: better balance
        chans = np.nonzero(sfs == sf)[0]
        base, rem = divmod(chans.size, world_size)
        counts = np.full(world_size, base, dtype=np.int64)
        if rem:
            # left-overs to the least-loaded ranks (stable: lowest rank wins ties)
            order = np.argsort(load + counts * bytes_per_symbol(sf), kind="stable")
            counts[order[:rem]] += 1
        start = 0
        for r in range(world_size):
            parts[r].append((sf, chans[start:start + counts[r]]))
            start += counts[r]
            load[r] += counts[r] * bytes_per_symbol(sf)
    out = []
    for r in range(world_size):
        pieces = [c for _, c in sorted(parts[r], key=lambda p: p[0])]
        out.append(np.concatenate(pieces) if pieces else np.zeros(0, np.int64))
    return out


def gather_symbols(local_sym, local_channels, n_channels, group=None):
    """After the run: put every rank's (channels_r, S) int16 symbols into the global
    (n_channels, S) array on every rank. One all_gather of a few MiB -- not on the hot path."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = local_sym.device
    S = local_sym.shape[1]
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([local_sym.shape[0]], dtype=torch.int64, device=dev), group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(counts) if counts else 0
    pad_sym = torch.zeros((cap, S), dtype=local_sym.dtype, device=dev)
    pad_sym[:local_sym.shape[0]] = local_sym
    pad_ch = torch.full((cap,), -1, dtype=torch.int64, device=dev)
    pad_ch[:local_sym.shape[0]] = torch.as_tensor(local_channels, dtype=torch.int64, device=dev)
    # 16-bit integers are not a collective dtype in RCCL/gloo: ship the payload as bytes
    raw = pad_sym.contiguous().view(torch.uint8)
    all_raw = [torch.empty_like(raw) for _ in range(world)]
    all_ch = [torch.empty_like(pad_ch) for _ in range(world)]
    dist.all_gather(all_raw, raw, group=group)
    dist.all_gather(all_ch, pad_ch, group=group)
    all_sym = [r.view(local_sym.dtype).reshape(cap, S) for r in all_raw]
    out = torch.zeros((n_channels, S), dtype=local_sym.dtype, device=dev)
    for r in range(world):
        k = counts[r]
        if k:
            out[all_ch[r][:k]] = all_sym[r][:k]
    return out
