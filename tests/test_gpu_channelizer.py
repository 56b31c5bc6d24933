"""GPU: the front-end channeliser (lorahip_channelizer_*). The reference has no such block (parity
unpinned, see oracle/channelizer.py): the fp32 kernel is held to the float64 restatement of its definition within a stated
tolerance, to bit-exact chunk invariance (a stream cut into ragged pieces == one call), and to the property that matters: a
wideband capture of several LoRa channels, channelised and handed to the batched demodulator, yields the symbols that were sent."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _stream(rng, n):
    return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)


# fp32 evaluation of an L-term sum of products with unit-scale inputs: the error bound is ~L * 2^-24 * sum|h| * max|x|; the
# measured error is far smaller (random-walk growth). Tolerance = 4e-6 * sum|h| * max|x|, stated relative to the output scale.
TOL = 4e-6


@pytest.mark.parametrize("K,D,L", [(8, 8, 64), (3, 5, 37), (19, 16, 128), (1, 1, 1), (2, 1, 9), (9, 64, 256), (5, 72, 300), (4, 10, 3)])
def test_against_float64_definition(gpu, K, D, L):
    import torch
    import lora_sdr_amd as Lh
    from oracle import channelizer as oc
    rng = np.random.default_rng(K * 1000 + D)
    n = 20000 + 7
    x = _stream(rng, n)
    freqs = rng.uniform(-0.5, 0.5, K)
    freqs[0] = 0.0
    h = oc.design_lowpass(D, L) if L > 1 else np.ones(1, np.float32)
    h = (h * rng.uniform(0.5, 1.5, L)).astype(np.float32)          # not symmetric: the tap order matters
    want = oc.channelize(x, freqs, D, h)
    with Lh.Context(7) as ctx:
        ch = Lh.Channelizer(ctx, freqs, D, h)
        got = ch.run(torch.from_numpy(x).cuda()).cpu().numpy()
        ch.close()
    assert got.shape == want.shape == (K, n // D)
    scale = float(np.abs(h).sum() * np.abs(x).max())
    err = float(np.abs(got - want).max())
    assert err <= TOL * scale, (err, scale)
    # and it is not trivially small: the outputs carry signal
    assert float(np.abs(want).max()) > 0.05 * scale / max(1.0, np.sqrt(L))


def test_phase_counter_matches_and_never_drifts(gpu):
    """a tone at a channel's centre comes out as DC of gain sum(h), also two billion samples into the stream (the 64-bit phase
    counter wraps exactly; a float phase accumulator would have lost the tone long before)"""
    import torch
    import lora_sdr_amd as Lh
    from oracle import channelizer as oc
    lib = Lh.load()
    for f in (0.1, -0.25, 0.4999, 1.0 / 3.0, -7.3):
        assert int(lib.lorahip_channelizer_phase_inc(f)) == oc.phase_inc(f)
    D, L, f0 = 8, 64, 0.1234567
    h = oc.design_lowpass(D, L)
    w = oc.phase_inc(f0)
    with Lh.Context(7) as ctx:
        ch = Lh.Channelizer(ctx, [f0, f0 + 0.25], D, h)
        # skip ahead: feed zeros in big chunks (cheap), then the tone with the phase the absolute sample index implies
        zeros = torch.zeros(1 << 24, dtype=torch.complex64, device="cuda")
        n0 = 0
        for _ in range(128):
            ch.run(zeros)
            n0 += zeros.numel()
        n = 4096
        idx = (np.arange(n, dtype=np.uint64) + np.uint64(n0))
        with np.errstate(over="ignore"):
            turns = (np.uint64(w) * idx).astype(np.int64).astype(np.float64) * 2.0 ** -64
        tone = np.exp(2j * np.pi * turns).astype(np.complex64)
        y = ch.run(torch.from_numpy(tone).cuda()).cpu().numpy()
        ch.close()
    settled = y[0, L // D + 1:]
    assert np.abs(settled - 1.0).max() < 2e-6            # sum(h) = 1, phase 0
    assert np.abs(y[1, L // D + 1:]).max() < 1e-3        # the channel a quarter of the band away sees only the stop band


def test_chunked_stream_is_bit_identical(gpu):
    import torch
    import lora_sdr_amd as Lh
    from oracle import channelizer as oc
    rng = np.random.default_rng(5)
    K, D, L, n = 11, 12, 100, 50000
    x = torch.from_numpy(_stream(rng, n)).cuda()
    freqs = rng.uniform(-0.5, 0.5, K)
    h = oc.design_lowpass(D, L)
    with Lh.Context(7) as ctx:
        ch = Lh.Channelizer(ctx, freqs, D, h)
        whole = ch.run(x).cpu().numpy()
        ch.reset()
        parts, pos = [], 0
        sizes = [1, 3, 0, 11, 12, 13, 1, 1, 1, 200, 5, 4096, 7, 111, 2, 10000]     # shorter than D, than the history, empty, long
        while pos < n:
            s = min(sizes[len(parts) % len(sizes)], n - pos)
            assert ch.out_count(s) == (pos + s) // D - pos // D
            parts.append(ch.run(x[pos:pos + s]).cpu().numpy())
            pos += s
        again = ch.run(x[:0])
        assert again.shape == (K, 0)
        ch.close()
    glued = np.concatenate(parts, axis=1)
    assert glued.shape == whole.shape
    assert np.array_equal(glued.view(np.uint32), whole.view(np.uint32))


def test_batch_of_captures_equals_fresh_streams(gpu):
    import torch
    import lora_sdr_amd as Lh
    from oracle import channelizer as oc
    rng = np.random.default_rng(9)
    K, D, L, S, n = 10, 8, 64, 5, 7001
    x = torch.from_numpy(np.stack([_stream(rng, n) for _ in range(S)])).cuda()
    freqs = rng.uniform(-0.5, 0.5, K)
    h = oc.design_lowpass(D, L)
    with Lh.Context(7) as ctx:
        ch = Lh.Channelizer(ctx, freqs, D, h)
        ch.run(x[0, :1234])                                  # leave the stream in some state: run_captures must not care
        batch = ch.run_captures(x).cpu().numpy()
        assert ch.out_count(D) == (1234 + D) // D - 1234 // D  # ... nor change it
        singles = []
        for s_ in range(S):
            ch.reset()
            singles.append(ch.run(x[s_]).cpu().numpy())
        ch.close()
    assert batch.shape == (S, K, n // D)
    for s_ in range(S):
        assert np.array_equal(batch[s_].view(np.uint32), singles[s_].view(np.uint32))


def test_argument_checks(gpu):
    import lora_sdr_amd as Lh
    h = np.ones(8, np.float32)
    with Lh.Context(7) as ctx:
        with pytest.raises(Lh.LoraHipError):
            Lh.Channelizer(ctx, [0.0], 0, h)
        with pytest.raises(Lh.LoraHipError):
            Lh.Channelizer(ctx, [], 4, h)
        with pytest.raises(Lh.LoraHipError):
            Lh.Channelizer(ctx, [0.0], 4, np.zeros(0, np.float32))
        with pytest.raises(Lh.LoraHipError):
            Lh.Channelizer(ctx, [0.0], 256, np.ones(65536, np.float32))      # tile does not fit the LDS


@pytest.mark.parametrize("sf", [7, 9])
def test_wideband_capture_to_symbols(gpu, sf):
    """8 LoRa channels on a 0.1-cycle grid in one wideband stream at 16x the channel rate (what a 2 MHz capture of 125 kHz
    channels at 200 kHz spacing looks like), plus noise: channeliser -> streaming demodulator returns every frame with the
    symbols that were modulated (up to each frame's constant bin offset, as for a directly fed demodulator)."""
    import torch
    import lora_sdr_amd as Lh
    K, D, L, nsyms, N = 8, 16, 128, 24, 1 << sf
    g = torch.Generator(device="cuda"); g.manual_seed(sf)
    freqs = (np.arange(K) - 3.5) * 0.1
    with Lh.Context(sf) as ctx:
        sent = torch.randint(0, N, (K, nsyms), generator=g, device="cuda", dtype=torch.int32)
        base = ctx.mod_frames(sent.to(torch.int16), sync=0x12, ampl=1.0, padding=1, lead=N // 2 + 5, tail=3 * N)   # (K, T) at the channel rate
        T = base.shape[1]
        # ideal interpolation by D (zero-padded spectrum), then each channel moved to its centre frequency and summed
        spec = torch.fft.fft(base, dim=1)
        wide_spec = torch.zeros((K, T * D), dtype=torch.complex64, device="cuda")
        half = T // 2
        wide_spec[:, :half] = spec[:, :half]
        wide_spec[:, -(T - half):] = spec[:, half:]
        up = torch.fft.ifft(wide_spec, dim=1) * D
        n = torch.arange(T * D, device="cuda", dtype=torch.float64)
        carriers = torch.exp(2j * np.pi * torch.from_numpy(freqs).cuda()[:, None] * n[None, :]).to(torch.complex64)
        wide = (up * carriers).sum(dim=0).contiguous()
        ctx.add_awgn(wide, 0.2, seed=3)                         # wideband noise; ~20 dB of SNR left in each channel after the filter
        ch = Lh.Channelizer(ctx, freqs, D, Lh.design_lowpass(D, L, cutoff=0.6 / D))
        narrow = ch.run(wide)
        ch.close()
        assert narrow.shape == (K, T)
        d = Lh.LoRaDemod(sf, n_channels=K); d.set_mode(1); d.setMTU(nsyms)
        d.work(narrow.contiguous())                              # no host sync: channeliser and demodulator share torch's stream
        pk = d.packets()
        d.close()
    sent_h = sent.cpu().numpy().astype(np.int64)
    found = {}
    for c, _, s in pk:
        found.setdefault(c, s)
    assert sorted(found) == list(range(K))
    for c in range(K):
        s = found[c].astype(np.int64)
        assert s.size == nsyms
        diff = (s - sent_h[c]) % N
        assert np.all(diff == diff[0]), (c, diff)


def test_running_chain_channeliser_into_segments(gpu):
    """The receiver as it runs: the wideband stream arrives in buffers of arbitrary length, the channeliser
    appends each buffer's output to the columns of one (K, capacity) device buffer, and the demodulator is given, per channel, the
    samples it has not consumed yet (lorahip_demod_run_device_segments) -- no copy of the remainders, no loss at the buffer
    boundaries. The packets are those of channelising the whole capture at once and demodulating it in one work()."""
    import torch
    import lora_sdr_amd as Lh
    sf = 8
    K, D, L, nsyms, N = 4, 8, 96, 10, 1 << sf
    g = torch.Generator(device="cuda"); g.manual_seed(21)
    rng = np.random.default_rng(21)
    freqs = (np.arange(K) - 1.5) * 0.2
    with Lh.Context(sf) as ctx:
        frames = []
        for f in range(3):
            sent = torch.randint(0, N, (K, nsyms), generator=g, device="cuda", dtype=torch.int32)
            frames.append(ctx.mod_frames(sent.to(torch.int16), sync=0x12, ampl=1.0, padding=1, lead=N // 2 + 5 + 37 * f, tail=2 * N))
        base = torch.cat(frames, dim=1)
        T = base.shape[1]
        spec = torch.fft.fft(base, dim=1)
        wide_spec = torch.zeros((K, T * D), dtype=torch.complex64, device="cuda")
        half = T // 2
        wide_spec[:, :half] = spec[:, :half]
        wide_spec[:, -(T - half):] = spec[:, half:]
        up = torch.fft.ifft(wide_spec, dim=1) * D
        n = torch.arange(T * D, device="cuda", dtype=torch.float64)
        carriers = torch.exp(2j * np.pi * torch.from_numpy(freqs).cuda()[:, None] * n[None, :]).to(torch.complex64)
        wide = (up * carriers).sum(dim=0).contiguous()
        ctx.add_awgn(wide, 0.1, seed=5)
        taps = Lh.design_lowpass(D, L, cutoff=0.6 / D)
        # one shot
        ch = Lh.Channelizer(ctx, freqs, D, taps)
        narrow = ch.run(wide).contiguous()
        ch.close()
        d = Lh.LoRaDemod(sf, n_channels=K); d.set_mode(1); d.setMTU(nsyms)
        d.work(narrow)
        want = sorted((c, s.tolist()) for c, _, s in d.packets())
        want_consumed = [d.consumed(c) for c in range(K)]
        d.close()
        assert len(want) == 3 * K
        # running
        cap = narrow.shape[1] + 8
        ring = torch.zeros((K, cap), dtype=torch.complex64, device="cuda")
        ch = Lh.Channelizer(ctx, freqs, D, taps)
        d = Lh.LoRaDemod(sf, n_channels=K); d.set_mode(1); d.setMTU(nsyms)
        read = np.zeros(K, np.int64)
        w, fed, got = 0, 0, []
        while fed < wide.numel():
            n_in = min(wide.numel() - fed, int(rng.integers(D * N // 3, 5 * D * N)))
            out = ch.run(wide[fed:fed + n_in], out=ring[:, w:])
            fed += n_in
            w += out.shape[1]
            d.work_segments(ring, np.arange(K) * cap + read, w - read)
            got += [(c, s.tolist()) for c, _, s in d.packets()]
            read += d.consumed_all()
        assert w == narrow.shape[1]
        assert torch.equal(ring[:, :w], narrow)                  # the chunked channeliser is bit-identical (tested above too)
        assert sorted(got) == want
        assert read.tolist() == want_consumed
        d.close(); ch.close()
