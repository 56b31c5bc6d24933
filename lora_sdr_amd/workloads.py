"""Synthetic workloads of the demod hot path, generated in HBM through the C ABI (bench.py, tools/, tests).

Nothing here computes a result: these are the input generators (the library's batched genChirp / LoRaMod / AWGN kernels,
ChirpGenerator.hpp:22-47, LoRaMod.cpp:109-238) arranged into the shapes BASELINE.json's configs name."""
import numpy as np


def default_geometry(sf):
    """channels x windows per step of the steady-state batch: BASELINE.json configs[1] (4096 channels SF7) and configs[2]
    (1024 channels SF12); in between the same 1 GiB of IQ per step"""
    channels = {7: 4096, 8: 4096, 9: 2048, 10: 2048, 11: 1024, 12: 1024}.get(sf, 4096)
    symbols = (1 << 30) // (channels * (8 << sf))
    return channels, max(symbols, 1)


# channels of the level-3 (whole LoRaDemod block) workload: enough channels to fill 256 CUs with the channel-per-wave-group /
# channel-per-workgroup mapping of the streaming kernels
LEVEL3_CHANNELS = {6: 16384, 7: 16384, 8: 8192, 9: 8192, 10: 4096, 11: 2048, 12: 1024}


def frame_streams(ctx, n_channels, n_frames=4, nsyms=48, sigma=0.05, sync=0x12, seed=1, distinct=64, stagger=True):
    """(n_channels, samples) complex64 device tensor: every channel carries n_frames LoRa frames (10 up-chirps, the two
    sync-word chirps, 2 1/4 down-chirps, nsyms data symbols; LoRaMod.cpp:135-229) separated by silence, plus AWGN.
    `distinct` different payload sets are tiled over the channels; with `stagger` the channels start at 16 different times
    spread over half a frame, so that the channels sharing a wavefront are NOT in the same receiver state at the same time
    (synchronised channels would flatter the streaming kernels, which take shortcuts when a whole wave is in DATASYMBOLS).
    Returns (iq, data) with data[(c % distinct), frame, :] the sent symbols."""
    import torch
    sf, N = ctx.sf, ctx.N
    dev = torch.device("cuda", ctx.device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    V = min(distinct, n_channels)
    per_frame = 10 + 2 + nsyms
    data = torch.randint(0, N, (V, n_frames, nsyms), generator=g, device=dev, dtype=torch.int32)
    syms = torch.zeros((V, n_frames, per_frame), dtype=torch.int32, device=dev)
    syms[:, :, 10] = (sync >> 4) * 8
    syms[:, :, 11] = (sync & 0xf) * 8                                                    # LoRaMod.cpp:150-169
    syms[:, :, 12:] = data
    up = ctx.synth_symbols(syms.reshape(-1).to(torch.int16)).reshape(V, n_frames, per_frame, N)
    down = torch.conj(ctx.synth_symbols(torch.zeros(1, dtype=torch.int16, device=dev)))  # LoRaMod.cpp:172-197
    parts = [torch.zeros((V, N // 2 + 5), dtype=torch.complex64, device=dev)]
    for f in range(n_frames):
        parts += [up[:, f, :12].reshape(V, -1), down.repeat(V, 2), down[: N // 4].repeat(V, 1), up[:, f, 12:].reshape(V, -1),
                  torch.zeros((V, 3 * N), dtype=torch.complex64, device=dev)]
    base = torch.cat(parts, dim=1)
    del parts, up
    if stagger:
        L0 = base.shape[1]
        groups = 16
        step = (per_frame * N // 2) // groups                       # leads 0 .. half a frame
        iq = torch.zeros((n_channels, L0 + groups * step), dtype=torch.complex64, device=dev)
        rows = torch.arange(n_channels, device=dev)
        for gi in range(groups):
            sel = rows[(rows * 7 + rows // 8) % groups == gi]       # neighbouring channels (one wave holds 1-16 of them) get different leads
            if sel.numel():
                lead = gi * step + 3 * gi
                iq[sel, lead:lead + L0] = base[sel % V]
    else:
        iq = base.repeat((n_channels + V - 1) // V, 1)[:n_channels].contiguous()
    del base
    if iq.shape[1] % 16:
        # rows of whole 128-byte lines (16 samples): what the resident receiver asks of its input (lorahip_demod_receive, async = 3)
        iq = torch.nn.functional.pad(iq, (0, -iq.shape[1] % 16)).contiguous()
    if sigma:
        ctx.add_awgn(iq, sigma, seed=seed)
    torch.cuda.synchronize(dev)
    return iq, data


def check_frame_packets(packets, data, N, nsyms, limit=None):
    """packets: LoRaDemod.packets() of a frame_streams() run. Every packet must carry the sent symbols up to one constant bin
    offset per packet (the frame sync removes genChirp's one-sample phase lead). Returns (checked, ok)."""
    V = data.shape[0]
    want_all = data.cpu().numpy()
    seen = {}
    ok = n = 0
    for ch, _rd, s in packets if limit is None else packets[:limit]:
        f = seen.get(ch, 0)
        seen[ch] = f + 1
        n += 1
        if f >= want_all.shape[1] or len(s) != nsyms:
            continue
        want = want_all[ch % V, f]
        diff = (s.astype(np.int64) - want) % N
        ok += int(np.all(diff == diff[0]))
    return n, ok


def mixed_sf_channels(n_channels=16384):
    """BASELINE.json configs[3]: SF(c) = 7 + (c mod 6)"""
    return 7 + (np.arange(n_channels) % 6)


def mixed_sent(sfs, S, device="cpu"):
    """the symbols "sent" on every channel of the mixed-SF workload: (n_channels, S) int32, symbol (c, k) < 2^SF(c). A counter
    hash in plain integer tensor arithmetic, so every rank -- and the CPU-side structure test -- derives the same truth."""
    import torch
    n = len(sfs)
    idx = torch.arange(n * S, dtype=torch.int64, device=device).reshape(n, S)
    h = (idx * 2654435761 + 0x9E3779B9) & 0xFFFFFFFF
    h = (h ^ (h >> 15)) * 2246822519 & 0xFFFFFFFF
    h = h ^ (h >> 13)
    mask = torch.as_tensor((1 << np.asarray(sfs, np.int64)) - 1, dtype=torch.int64, device=device).reshape(n, 1)
    return (h & mask).to(torch.int32)


def mixed_errors(full, sfs, S, bin_offset=1):
    """gathered (n_channels, S) symbols against mixed_sent(): demodulating a window-aligned genChirp symbol s yields bin
    s + 1. Returns the number of symbols that differ."""
    import torch
    dev = full.device
    sent = mixed_sent(sfs, S, dev)
    n = torch.as_tensor(1 << np.asarray(sfs, np.int64), dtype=torch.int64, device=dev).reshape(-1, 1)
    got = full.to(torch.int64) & 0xffff
    return int((((got - sent.to(torch.int64)) % n) != bin_offset).sum())
