"""CPU: the multi-GPU split (lora_sdr_amd/shard.py) and the post-run gather over gloo, world_size 2."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

from lora_sdr_amd.shard import bytes_per_symbol, shard_channels


def test_bytes_per_symbol_matches_survey():
    assert [bytes_per_symbol(sf) for sf in range(7, 13)] == [1038, 2062, 4110, 8206, 16398, 32782]
    assert bytes_per_symbol(7, fft_out=True, dec_out=True) == 1038 + 2 * 1024


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_shard_is_a_balanced_partition(world):
    sfs = 7 + np.arange(16384) % 6          # BASELINE configs[3]
    parts = shard_channels(sfs, world)
    allc = np.concatenate(parts)
    assert np.array_equal(np.sort(allc), np.arange(16384))       # every channel exactly once
    loads = [sum(bytes_per_symbol(int(s)) for s in sfs[p]) for p in parts]
    assert max(loads) - min(loads) <= bytes_per_symbol(12)        # within one SF12 window
    for p in parts:
        assert np.all(np.diff(sfs[p]) >= 0)                       # bucketed by SF inside a rank


def test_shard_ragged_and_empty():
    assert [len(p) for p in shard_channels([12, 7, 7], 4)] == [1, 1, 1, 0]
    assert all(len(p) == 0 for p in shard_channels([], 3))
    parts = shard_channels([7] * 5 + [12] * 3, 2)
    assert sorted(np.concatenate(parts).tolist()) == list(range(8))
    with pytest.raises(ValueError):
        shard_channels([7], 0)


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from lora_sdr_amd.shard import gather_symbols, shard_channels
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_ch, S = 37, 5
    sfs = 7 + np.arange(n_ch) % 6
    mine = shard_channels(sfs, world)[rank]
    # stand-in for the per-rank demod result: symbol of (channel c, window k) = 100*c + k
    local = torch.tensor([[100 * int(c) + k for k in range(S)] for c in mine], dtype=torch.int16).reshape(len(mine), S)
    full = gather_symbols(local, mine, n_ch)
    expect = torch.tensor([[100 * c + k for k in range(S)] for c in range(n_ch)], dtype=torch.int16)
    t = torch.tensor([float(len(mine))])
    dist.all_reduce(t)
    q.put((rank, bool(torch.equal(full, expect)), float(t[0])))
    dist.destroy_process_group()


def test_gather_over_gloo_world2():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0, "gloo worker failed"
    res = [q.get(timeout=10) for _ in procs]
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res)
    assert all(r[2] == 37.0 for r in res)


def _mixed_worker(rank, world, port, q):
    """BASELINE configs[3] end to end minus the GPU: the shard plan, the per-bucket result layout, the gather and the check that
    bench.py --config mixed performs, with the demodulator replaced by its defining property (symbol s -> bin s + 1)"""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from lora_sdr_amd import workloads as WL
    from lora_sdr_amd.shard import gather_symbols, shard_channels
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_ch, S = 16384, 16
    sfs = WL.mixed_sf_channels(n_ch)
    mine = shard_channels(sfs, world)[rank]
    sent = WL.mixed_sent(sfs, S)
    parts, order = [], []
    for sf in range(7, 13):                                    # one "launch" per SF bucket, like the bench
        ch = mine[sfs[mine] == sf]
        if ch.size == 0:
            continue
        got = (sent[torch.from_numpy(ch)] + 1) % (1 << sf)
        parts.append(got.to(torch.int16))
        order.append(ch)
    full = gather_symbols(torch.cat(parts), np.concatenate(order), n_ch)
    bad = WL.mixed_errors(full, sfs, S)
    # and a corrupted result must be caught
    full2 = full.clone()
    full2[5, 3] ^= 1
    q.put((rank, bad, WL.mixed_errors(full2, sfs, S), int(mine.size)))
    dist.destroy_process_group()


def test_mixed_sf_config_structure_over_gloo_world2():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mixed_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0, "gloo worker failed"
    res = sorted(q.get(timeout=10) for _ in procs)
    assert [r[1] for r in res] == [0, 0] and [r[2] for r in res] == [1, 1]
    assert sum(r[3] for r in res) == 16384
