"""GPU: B channels of the LoRaDemod block (level 3 of the C ABI) against the CPU oracle's
restated block -- which is itself pinned against the verbatim LoRaDemod.cpp (test_oracle_vs_ref)
and the committed golden stream -- and the BASELINE.json configs that are parity cases."""
import numpy as np
import pytest

from conftest import same_values

pytestmark = pytest.mark.gpu

TOL_DB = 2e-5


def frames(oracle, rng, sf, n_frames, nsyms, off=0.0, noise=0.05, sync=0x12, lead=None):
    N = 1 << sf
    syms = [rng.integers(0, N, nsyms).astype(np.uint16) for _ in range(n_frames)]
    parts = [np.zeros(N // 2 + 5 if lead is None else lead, np.complex64)]
    for s in syms:
        parts.append(oracle.mod_frame(sf, s, sync=sync, padding=3))
    parts.append(np.zeros(3 * N, np.complex64))
    st = np.concatenate(parts)
    st = (st * np.exp(2j * np.pi * off / N * np.arange(st.size))).astype(np.complex64)
    st += (noise * (rng.standard_normal(st.size) + 1j * rng.standard_normal(st.size))).astype(np.complex64)
    return st, syms


def compare_channel(tr, ref_calls):
    assert len(tr) == len(ref_calls)
    for i, (a, b) in enumerate(zip(tr, ref_calls)):
        assert a["consumed"] == b["consumed"], "call %d consumed" % i
        assert a["state_before"] == b["state"], "call %d state" % i
        assert a["value"] == b["value"], "call %d value" % i
        assert abs(a["f_index"] - b["fIndex"]) <= 2e-6
        if np.isfinite(b["power"]):
            assert abs(a["power"] - b["power"]) <= TOL_DB


MODES = [1, 2]     # 1 = streaming kernel (frame machine on the device), 2 = host-driven lock-step rounds


@pytest.mark.parametrize("mode", MODES)
def test_config1_single_channel_sf7_loopback(gpu, oracle, mode):
    """BASELINE configs[0]: single channel SF=7, modulator frame -> demod -> sent symbols (no noise)"""
    import lora_sdr_amd as L
    rng = np.random.default_rng(1)
    st, syms = frames(oracle, rng, 7, 1, 24, noise=0.0)
    d = L.LoRaDemod(7)
    d.set_mode(mode)
    d.setMTU(24)
    d.set_trace(True)
    d.work([st])
    pk = d.packets()
    assert len(pk) == 1 and np.array_equal(pk[0][2], syms[0].astype(np.int16))
    r = oracle.demod_run(7, st, mtu=24)
    compare_channel(d.trace(0), r["calls"])
    assert [p[1] for p in pk] == [c for c, _ in r["packets"]]


@pytest.mark.parametrize("mode", MODES)
def test_golden_stream_through_demod(gpu, golden, mode):
    """the committed stream recorded from the verbatim LoRaDemod.cpp: same consumption, same packets"""
    import lora_sdr_amd as L
    g = golden("demod_stream.npz")
    for sf in (7, 9):
        d = L.LoRaDemod(sf)
        d.set_mode(mode)
        d.setMTU(int(g["mtu_%d" % sf]))
        d.set_trace(True)
        d.work([g["iq_%d" % sf]])
        tr = d.trace(0)
        assert [t["consumed"] for t in tr] == g["consumed_%d" % sf].tolist()
        assert d.labels(0) == g["labels_%d" % sf].tolist()          # the block's stream labels, rebuilt from the trace
        pk = d.packets()
        assert [p[1] for p in pk] == g["packet_calls_%d" % sf].tolist()
        assert np.array_equal(np.stack([p[2] for p in pk]), g["packets_%d" % sf])
        sig = [(t["sig_error"], t["sig_power"], t["sig_snr"]) for t in tr if t["signals"]]
        assert np.allclose(np.array(sig).reshape(-1), g["signals_%d" % sf], rtol=0, atol=TOL_DB)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("sf", [6, 7, 8, 9, 10, 11, 12])
def test_many_channels_lockstep(gpu, oracle, sf, mode):
    """channels with different lengths, frequency offsets, sync alignment and noise, one of them pure
    noise and one too short to work at all: every channel must follow its own oracle block"""
    import lora_sdr_amd as L
    rng = np.random.default_rng(sf)
    N = 1 << sf
    B = 12
    streams, sent = [], []
    for c in range(B):
        if c == 3:
            st = (0.3 * (rng.standard_normal(9 * N) + 1j * rng.standard_normal(9 * N))).astype(np.complex64)
            sy = []
        elif c == 7:
            st, sy = np.zeros(2 * N - 1, np.complex64), []      # < 2N: never works (LoRaDemod.cpp:148)
        else:
            st, sy = frames(oracle, rng, sf, 1 + c % 3, 6 + c, off=rng.uniform(-0.45, 0.45),
                            noise=0.02 + 0.02 * c, lead=int(rng.integers(0, 2 * N)))
        streams.append(st)
        sent.append(sy)
    d = L.LoRaDemod(sf, n_channels=B)
    d.set_mode(mode)
    d.setMTU(64)
    d.set_trace(True)
    d.work(streams)
    pk = d.packets()
    total_calls = 0
    for c in range(B):
        r = oracle.demod_run(sf, streams[c], mtu=64)
        compare_channel(d.trace(c), r["calls"])
        total_calls += len(r["calls"])
        mine = [p[2] for p in pk if p[0] == c]
        assert len(mine) == len(r["packets"])
        for a, (_, b) in zip(mine, r["packets"]):
            assert np.array_equal(a, b)
    assert d.work_calls() == total_calls
    assert len(d.trace(7)) == 0


@pytest.mark.parametrize("mode", [0] + MODES)
def test_sf6_demodulator(gpu, oracle, mode):
    """SF6 (64-point windows, 4 lanes each): the streaming kernel, host-driven rounds and the automatic choice all follow the oracle"""
    import lora_sdr_amd as L
    rng = np.random.default_rng(66)
    sf, N = 6, 64
    streams = [frames(oracle, rng, sf, 2, 9 + c, off=rng.uniform(-0.4, 0.4), noise=0.02, lead=int(rng.integers(0, 2 * N)))[0] for c in range(21)]
    d = L.LoRaDemod(sf, n_channels=21); d.set_mode(mode); d.setMTU(64); d.set_trace(True)
    d.work(streams)
    pk = d.packets()
    for c in range(21):
        r = oracle.demod_run(sf, streams[c], mtu=64)
        compare_channel(d.trace(c), r["calls"])
        mine = [p[2] for p in pk if p[0] == c]
        assert len(mine) == len(r["packets"]) and all(np.array_equal(a, b) for a, (_, b) in zip(mine, r["packets"]))
    d.close()


@pytest.mark.parametrize("mode", MODES)
def test_device_resident_streams(gpu, oracle, mode):
    """lorahip_demod_run_device: the streams are already one (B, samples) tensor in HBM"""
    import lora_sdr_amd as L
    sf, B = 8, 5
    rng = np.random.default_rng(42)
    sts = [frames(oracle, rng, sf, 2, 10, off=0.2 * c - 0.4, noise=0.05, lead=100)[0] for c in range(B)]
    n = min(len(s) for s in sts)
    arr = np.stack([s[:n] for s in sts])
    d = L.LoRaDemod(sf, n_channels=B)
    d.set_mode(mode)
    d.setMTU(10)
    d.set_trace(True)
    d.work(gpu.from_numpy(arr).cuda())
    for c in range(B):
        compare_channel(d.trace(c), oracle.demod_run(sf, arr[c], mtu=10)["calls"])


def test_config5_sf10_awgn_minus10db(gpu, oracle):
    """BASELINE configs[4]: SF=10, AWGN at SNR=-10 dB per sample (sigma^2 = 5 per component), 8192
    channels: symbol error rate of the HIP path == that of the CPU reference on identical IQ
    (identical indices on a CPU-checked subset, SER ~ 0 on the full batch)."""
    import lora_sdr_amd as L
    torch = gpu
    sf, N, B, S = 10, 1024, 8192, 4
    ctx = L.Context(sf)
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    sym = torch.randint(0, N, (B * S,), generator=g, device="cuda", dtype=torch.int32).to(torch.int16)
    iq = ctx.synth_symbols(sym, ampl=1.0, noise_sigma=float(np.sqrt(5.0)), seed=77)
    r = ctx.detect_batch(iq)
    torch.cuda.synchronize()
    got = r["sym"].cpu().numpy().view(np.uint16).astype(np.int64)
    sent = sym.cpu().numpy().view(np.uint16).astype(np.int64)
    ser = float((((got - sent) % N) != 1).mean())            # genChirp symbol s -> bin s+1
    assert ser < 1e-3, ser
    k = 1500
    o = oracle.detect_batch(sf, iq[:k * N].cpu().numpy(), nthreads=8)
    assert np.array_equal(o["sym"], got[:k].astype(np.uint16))
    ser_cpu = float((((o["sym"].astype(np.int64) - sent[:k]) % N) != 1).mean())
    assert ser_cpu == float((((got[:k] - sent[:k]) % N) != 1).mean())
    assert np.abs(r["power"][:k].cpu().numpy() - o["power"]).max() <= TOL_DB


def test_config4_mixed_sf_sharded(gpu, oracle):
    """BASELINE configs[3] (scaled down to one GPU's worth of one rank): mixed SF 7..12 channels, bucketed
    by SF and sharded over 8 ranks; this process plays rank 3. Indices equal the oracle's."""
    import lora_sdr_amd as L
    torch = gpu
    n_ch, S, world, rank = 768, 4, 8, 3
    sfs = 7 + np.arange(n_ch) % 6
    mine = L.shard_channels(sfs, world)[rank]
    assert len(mine) == n_ch // world
    rng = np.random.default_rng(4)
    for sf in range(7, 13):
        N = 1 << sf
        chans = mine[sfs[mine] == sf]
        if len(chans) == 0:
            continue
        ctx = L.Context(sf)
        sym = torch.from_numpy(rng.integers(0, N, len(chans) * S).astype(np.int16)).cuda()
        iq = ctx.synth_symbols(sym, noise_sigma=1.0, seed=1000 + sf)
        r = ctx.detect_batch(iq)
        o = oracle.detect_batch(sf, iq.cpu().numpy(), nthreads=8)
        assert np.array_equal(r["sym"].cpu().numpy().view(np.uint16), o["sym"])


def ulp_diff(a, b):
    """distance in float32 ulps between two float arrays (same sign assumed where it matters)"""
    ia = np.ascontiguousarray(a).view(np.int32).astype(np.int64)
    ib = np.ascontiguousarray(b).view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia)
    ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


@pytest.mark.parametrize("sf,nsyms,sync,ampl,padding", [(7, 11, 0x12, 1.0, 1), (8, 6, 0x34, 0.5, 3), (10, 5, 0x8e, 1.0, 0), (12, 3, 0x12, 2.0, 2)])
def test_batched_modulator_matches_loramod(gpu, oracle, sf, nsyms, sync, ampl, padding):
    """lorahip_mod_frames against the LoRaMod frame of the oracle (itself bit-exact with the verbatim LoRaMod.cpp):
    same length, same zero padding, every sample within 1 float ulp. The float frequency / phase recurrence is reproduced
    exactly (an error there would grow along the frame); what differs is polar()'s cosf/sinf: the reference takes them
    from the platform's libm (glibc: up to 0.56 ulp), the kernel rounds the fp64 value once (<= 0.5 ulp), so a percent
    or two of the samples sit one ulp apart."""
    import lora_sdr_amd as L
    torch = gpu
    rng = np.random.default_rng(sf)
    F = 70                                               # more than one wavefront, ragged
    syms = rng.integers(0, 1 << sf, (F, nsyms)).astype(np.uint16)
    ctx = L.Context(sf)
    iq = ctx.mod_frames(torch.from_numpy(syms.view(np.int16)).cuda(), sync=sync, ampl=ampl, padding=padding, lead=5, tail=3)
    torch.cuda.synchronize()
    got = iq.cpu().numpy()
    assert got.shape[1] == 5 + ctx.mod_frame_len(nsyms, padding) + 3
    assert not got[:, :5].any() and not got[:, -3:].any()
    worst, differ, total = 0, 0, 0
    for f in (0, 1, 63, 64, F - 1):
        ref = oracle.mod_frame(sf, syms[f], sync=sync, ampl=ampl, padding=padding)
        mine = got[f, 5:-3]
        assert mine.size == ref.size
        d = ulp_diff(mine.view(np.float32), ref.view(np.float32))
        small = np.abs(ref.view(np.float32)) < 1e-6 * ampl           # near a zero crossing an ulp is meaningless: absolute check
        assert np.abs(mine.view(np.float32) - ref.view(np.float32))[small].max(initial=0) < 1e-7 * ampl
        worst = max(worst, int(d[~small].max()))
        differ += int((d[~small] > 0).sum())
        total += int((~small).sum())
    assert worst <= 1, "more than one ulp from the reference modulator: %d" % worst
    assert differ <= 5e-2 * total, "%d of %d samples differ in the last ulp" % (differ, total)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("sf", [7, 9, 10, 11, 12])
def test_loopback_mod_noise_demod(gpu, sf, mode):
    """TestLoopback.cpp's chain without the codec: symbols -> modulator -> AWGN -> demodulator -> the same symbols,
    for many channels at once, everything on the device"""
    import lora_sdr_amd as L
    torch = gpu
    N, B, nsyms, frames = 1 << sf, (300 if sf <= 10 else 70), 20, 2
    g = torch.Generator(device="cuda")
    g.manual_seed(sf)
    syms = torch.randint(0, N, (B * frames, nsyms), generator=g, device="cuda", dtype=torch.int32).to(torch.int16)
    ctx = L.Context(sf)
    iq = ctx.mod_frames(syms, padding=2, lead=N // 2 + 7, tail=0)                 # (B*frames, row)
    iq = iq.reshape(B, -1)
    iq = torch.cat([iq, torch.zeros((B, 3 * N), dtype=torch.complex64, device="cuda")], dim=1).contiguous()
    ctx.add_awgn(iq, sigma=0.2, seed=99)
    d = L.LoRaDemod(sf, n_channels=B)
    d.set_mode(mode)
    d.setMTU(nsyms)
    d.work(iq)
    pk = d.packets()
    assert len(pk) == B * frames
    sent = syms.cpu().numpy().reshape(B, frames, nsyms)
    seen = np.zeros(B, np.int64)
    for ch, _, s in pk:
        assert np.array_equal(s, sent[ch, seen[ch]]), "channel %d frame %d" % (ch, seen[ch])
        seen[ch] += 1
    assert (seen == frames).all()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("sf", [7, 9, 11])
def test_chunked_streaming_equals_one_shot(gpu, oracle, sf, mode):
    """a running receiver: the stream arrives in chunks of arbitrary size, the block keeps its state between work() calls
    (fine-tune index, frequency error, symbols of a half-received packet) and the caller re-presents what was not
    consumed -- the packets must be those of one work() over the whole stream, and of the oracle's block"""
    import lora_sdr_amd as L
    rng = np.random.default_rng(100 + sf)
    N = 1 << sf
    B = 5
    streams = [frames(oracle, rng, sf, 3, 9 + c, off=rng.uniform(-0.4, 0.4), noise=0.05, lead=int(rng.integers(0, N)))[0] for c in range(B)]
    one = L.LoRaDemod(sf, n_channels=B); one.set_mode(mode); one.setMTU(9)
    one.work(streams)
    want = [[p[2] for p in one.packets(clear=False) if p[0] == c] for c in range(B)]
    for c in range(B):
        ref = oracle.demod_run(sf, streams[c], mtu=9)["packets"]
        assert len(ref) == len(want[c]) and all(np.array_equal(a, b) for a, (_, b) in zip(want[c], ref))
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(mode); d.setMTU(9)
    got = [[] for _ in range(B)]
    fed = [0] * B                       # samples handed over so far
    rest = [np.zeros(0, np.complex64) for _ in range(B)]
    while any(fed[c] < len(streams[c]) or len(rest[c]) >= 2 * N for c in range(B)):
        bufs = []
        for c in range(B):
            n = int(rng.integers(N // 3, 5 * N))
            bufs.append(np.concatenate([rest[c], streams[c][fed[c]:fed[c] + n]]))
            fed[c] = min(len(streams[c]), fed[c] + n)
        d.work(bufs)
        for ch, _, s in d.packets():
            got[ch].append(s)
        progressed = False
        for c in range(B):
            k = d.consumed(c)
            progressed |= k > 0
            rest[c] = bufs[c][k:]
        if not progressed and all(fed[c] >= len(streams[c]) for c in range(B)):
            break
    for c in range(B):
        assert len(got[c]) == len(want[c]), "channel %d: %d packets, expected %d" % (c, len(got[c]), len(want[c]))
        assert all(np.array_equal(a, b) for a, b in zip(got[c], want[c]))


@pytest.mark.parametrize("sf", [7, 10, 12])
def test_stream_kernel_resumes_when_its_record_buffer_fills(gpu, oracle, sf):
    """the streaming launch stops a channel when its per-launch record buffer is full and is relaunched from the saved state:
    with a capacity of 5 calls per launch every frame is cut many times, packets and traces must not change"""
    import lora_sdr_amd as L
    rng = np.random.default_rng(7 * sf)
    B = 6
    streams = [frames(oracle, rng, sf, 2, 7 + c, off=rng.uniform(-0.4, 0.4), noise=0.03, lead=int(rng.integers(0, 100)))[0] for c in range(B)]
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(7); d.set_trace(True)
    d.set_record_capacity(5)
    d.work(streams)
    pk = d.packets()
    for c in range(B):
        r = oracle.demod_run(sf, streams[c], mtu=7)
        compare_channel(d.trace(c), r["calls"])
        mine = [p[2] for p in pk if p[0] == c]
        assert len(mine) == len(r["packets"]) and all(np.array_equal(a, b) for a, (_, b) in zip(mine, r["packets"]))


@pytest.mark.parametrize("mode", MODES)
def test_sync_word_threshold_and_mtu_settings(gpu, oracle, mode):
    """the block's three setters: a non-default sync word (and the wrong one: no lock), a threshold high enough that the
    padding squelches the packet early, MTU 4 / 1 / 0 (a packet per symbol) -- every call and packet as the oracle's block"""
    import lora_sdr_amd as L
    rng = np.random.default_rng(7)
    sf, N = 8, 256
    syms = rng.integers(0, N, 9).astype(np.uint16)
    fr = oracle.mod_frame(sf, syms, sync=0x34, padding=4)
    st = np.concatenate([fr, fr, np.zeros(2 * N, np.complex64)])
    st += (0.01 * (rng.standard_normal(st.size) + 1j * rng.standard_normal(st.size))).astype(np.complex64)
    cases = ((0x34, 64, 10.0), (0x34, 4, 10.0), (0x12, 64, 10.0), (0x34, 1, -30.0), (0x34, 0, -30.0), (0x34, 64, 60.0))
    d = L.LoRaDemod(sf, n_channels=len(cases))       # one channel per setting would need one block each: run them one by one
    d.close()
    for sync, mtu, thresh in cases:
        d = L.LoRaDemod(sf)
        d.set_mode(mode); d.setSync(sync); d.setMTU(mtu); d.setThreshold(thresh); d.set_trace(True)
        d.work([st])
        r = oracle.demod_run(sf, st, sync=sync, mtu=mtu, thresh=thresh)
        compare_channel(d.trace(0), r["calls"])
        pk = d.packets()
        assert [p[1] for p in pk] == [c for c, _ in r["packets"]], (sync, mtu, thresh)
        assert all(np.array_equal(p[2], q) for p, (_, q) in zip(pk, r["packets"]))
        d.close()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("sf", [7, 9, 11])
def test_level3_debug_ports_and_labels(gpu, oracle, golden, sf, mode):
    """the block's raw / dec / fft outputs and stream labels at level 3 of the C ABI (lorahip_demod_set_ports,
    lorahip_demod_get_labels; LoRaDemod.cpp:81-83,163-164,172,314-324): bit-identical to the restated block, which is pinned to the
    verbatim LoRaDemod.cpp; for SF7 / SF9 also against the checksums recorded from the verbatim block"""
    import lora_sdr_amd as L
    rng = np.random.default_rng(100 + sf)
    N = 1 << sf
    if sf in (7, 9):
        g = golden("demod_stream.npz")
        st0, mtu = g["iq_%d" % sf], int(g["mtu_%d" % sf])
    else:
        st0, _ = frames(oracle, rng, sf, 1, 6, off=0.3)
        mtu = 6
    st1, _ = frames(oracle, rng, sf, 2, mtu, off=-0.4, lead=N // 3)
    n = max(st0.size, st1.size)
    streams = np.zeros((2, n), np.complex64)
    streams[0, :st0.size] = st0
    streams[1, :st1.size] = st1
    d = L.LoRaDemod(sf, n_channels=2)
    d.set_mode(mode)
    d.setMTU(mtu)
    d.set_trace(True)
    cap_frames = n // (N // 4) + 8
    d.set_ports(fft_frames=cap_frames, dec_samples=n, raw_samples=n)
    d.work(gpu.from_numpy(streams).to("cuda:0"))
    for c in range(2):
        r = oracle.demod_run(sf, streams[c], mtu=mtu)
        consumed = np.array([k["consumed"] for k in r["calls"]])
        p = d.ports(c)
        assert p["produced"] == dict(fft=len(consumed), dec=int(consumed.sum()), raw=int(consumed.sum()))
        assert same_values(p["raw"].cpu().numpy(), streams[c][:int(consumed.sum())])
        assert same_values(p["fft"].cpu().numpy(), np.stack(r["fft"]))
        want_dec = np.concatenate([r["dec"][k][:int(consumed[k])] for k in range(len(consumed))])
        assert same_values(p["dec"].cpu().numpy(), want_dec)
        assert d.labels(c) == [k["label"] for k in r["calls"]]
        if c == 0 and sf in (7, 9):
            fft = p["fft"].cpu().numpy()
            assert np.array_equal(np.abs(fft).argmax(axis=1), g["fft_peak_%d" % sf])
            assert same_values(fft.sum(axis=1), g["fft_sum_%d" % sf])
            starts = np.concatenate([[0], np.cumsum(consumed)[:-1]])
            dec = p["dec"].cpu().numpy()
            for j, k in enumerate([0, 5, 12, 20]):
                m = min(N, int(consumed[k]))
                assert same_values(dec[starts[k]:starts[k] + m], g["dec_first_%d" % sf][j][:m])
    # ports off again: nothing is produced, the run is the plain one
    d.set_ports()
    d.activate()
    d.work(gpu.from_numpy(streams).to("cuda:0"))
    assert d.ports(0)["produced"] == dict(fft=0, dec=0, raw=0)
    d.close()


@pytest.mark.parametrize("mode", MODES)
def test_ports_without_a_trace_keep_no_trace(gpu, oracle, mode):
    """debug ports on, tracing never asked for: the library needs a per-call trace internally for the port replay, but it lives for
    one run only -- nothing accumulates over the runs of a long-lived receiver and nothing leaks into the trace / label accessors --
    while the ports still carry the block's outputs"""
    import lora_sdr_amd as L
    sf, N, mtu = 8, 256, 7
    rng = np.random.default_rng(88)
    st, _ = frames(oracle, rng, sf, 2, mtu, off=0.3, lead=N // 3)
    streams = st.reshape(1, -1)
    d = L.LoRaDemod(sf, n_channels=1)
    d.set_mode(mode)
    d.setMTU(mtu)
    d.set_ports(fft_frames=st.size // (N // 4) + 8, dec_samples=st.size, raw_samples=st.size)
    r = oracle.demod_run(sf, st, mtu=mtu)
    consumed = np.array([k["consumed"] for k in r["calls"]])
    for _ in range(3):
        d.activate()
        d.work(gpu.from_numpy(streams).to("cuda:0"))
        assert d.trace(0) == [] and d.labels(0) == []
        p = d.ports(0)
        assert p["produced"] == dict(fft=len(consumed), dec=int(consumed.sum()), raw=int(consumed.sum()))
        assert same_values(p["fft"].cpu().numpy(), np.stack(r["fft"]))
        assert len(d.packets()) == len(r["packets"])
        # the zero start state for the next pass (activate() alone keeps the fine-tune members, like the reference)
        d.close()
        d = L.LoRaDemod(sf, n_channels=1)
        d.set_mode(mode); d.setMTU(mtu)
        d.set_ports(fft_frames=st.size // (N // 4) + 8, dec_samples=st.size, raw_samples=st.size)
    # one object, several runs: still no trace
    for _ in range(3):
        d.work(gpu.from_numpy(streams).to("cuda:0"))
        assert d.trace(0) == []
    d.close()


@pytest.mark.parametrize("sf", [7, 10, 12])
def test_packets_packed_on_the_device_equal_the_host_queue(gpu, oracle, sf):
    """lorahip_demod_packets_to_device straight after a streaming run packs the decoder's input rows from the kernel's records on
    the device (no host round trip); the same packets must come out of the host queue. Then streams cut in the middle of a
    packet: the run that ends inside a packet keeps its symbols, the next one completes it (host-queue path)."""
    import lora_sdr_amd as L
    rng = np.random.default_rng(500 + sf)
    N = 1 << sf
    B = 9
    mtu = 11
    sts, want = [], []
    for c in range(B):
        st, syms = frames(oracle, rng, sf, 1 + c % 3, mtu, off=0.2 * (c % 4), lead=N // 2 + 3 * c)
        sts.append(st)
    n = max(s.size for s in sts) + 2 * N
    streams = np.zeros((B, n), np.complex64)
    for c, s in enumerate(sts):
        streams[c, :s.size] = s
        want.append([p for _k, p in oracle.demod_run(sf, streams[c], mtu=mtu, keep=False)["packets"]])
    dev = gpu.from_numpy(streams).to("cuda:0")
    d = L.LoRaDemod(sf, n_channels=B)
    d.set_mode(1)
    d.setMTU(mtu)
    d.work(dev)
    ps, pn, pc = d.packets_device(clear=False)                   # device path: rows by channel, then time
    ps, pn, pc = ps.cpu().numpy(), pn.cpu().numpy(), pc.cpu().numpy()
    assert ps.shape == (sum(len(w) for w in want), mtu)
    assert pc.tolist() == [c for c in range(B) for _ in want[c]]
    row = 0
    for c in range(B):
        for p in want[c]:
            assert pn[row] == len(p) and np.array_equal(ps[row, :len(p)], p) and not ps[row, len(p):].any()
            row += 1
    host = d.packets(clear=False)                                # the host queue holds the same packets (round-major order)
    assert sorted((c, tuple(s.tolist())) for c, _r, s in host) == sorted((int(pc[i]), tuple(ps[i, :pn[i]].tolist())) for i in range(len(pn)))
    ps2, pn2, pc2 = d.packets_device()                           # now from the host queue: its order
    assert [int(x) for x in pc2.cpu()] == [c for c, _r, _s in host]
    assert all(np.array_equal(ps2[i, :len(s)].cpu().numpy(), s) for i, (_c, _r, s) in enumerate(host))
    # ---- cut every stream inside its last packet: run 1 ends with open packets, run 2 completes them ----
    d.activate()
    cut = n - 2 * N - (3 * N + N // 2) - 4 * N                    # before the end of the last frame's data symbols
    d.work(dev[:, :cut].contiguous())
    first = d.packets()                                           # drains (channels are inside a packet)
    used = [d.consumed(c) for c in range(B)]
    rest = max(n - u for u in used)
    tail = np.zeros((B, rest), np.complex64)
    for c in range(B):
        tail[c, :n - used[c]] = streams[c, used[c]:]
    d.work(gpu.from_numpy(tail).to("cuda:0"))
    ps3, pn3, pc3 = d.packets_device(clear=False)                 # a packet begun in run 1: host-queue path
    second = d.packets()
    got = {c: [] for c in range(B)}
    for c, _r, s in first + second:
        got[c].append(s)
    for c in range(B):
        assert len(got[c]) == len(want[c]) and all(np.array_equal(a, b) for a, b in zip(got[c], want[c])), c
    assert [int(x) for x in pc3.cpu()] == [c for c, _r, _s in second]
    d.close()


@pytest.mark.parametrize("sf", [7, 9, 10, 11, 12])
def test_squelch_decisions_without_trace_at_the_threshold(gpu, oracle, sf):
    """Without a trace the streaming kernels take the squelch decision (LoRaDemod.cpp:173-174) of DATASYMBOLS and FRAMESYNC
    windows from a quick estimate, falling back to the exact chain within 0.01 dB of the threshold, and evaluate fIndex only for
    unsquelched FRAMESYNC windows. Thresholds placed EXACTLY on snr values the reference computes for data symbols and for
    windows of the sync search (and a hair beside them) must give the reference's packets: which symbol is squelched first
    decides a packet's length, which preamble window is squelched decides whether (and where) the frame is found at all."""
    import lora_sdr_amd as L
    rng = np.random.default_rng(900 + sf)
    N = 1 << sf
    st, _ = frames(oracle, rng, sf, 3, 14, off=0.15, noise=0.4)
    r0 = oracle.demod_run(sf, st, mtu=64, thresh=-30.0, keep=False)
    snrs = sorted(c["snr"] for c in r0["calls"] if c["state"] == 4 and np.isfinite(c["snr"]))
    picks = [snrs[len(snrs) // 4], snrs[len(snrs) // 2], snrs[(3 * len(snrs)) // 4]]
    snrs0 = sorted(c["snr"] for c in r0["calls"] if c["state"] == 0 and np.isfinite(c["snr"]) and c["snr"] > -20.0)
    assert len(snrs0) >= 4
    picks += [snrs0[len(snrs0) // 3], snrs0[(2 * len(snrs0)) // 3], snrs0[-1]]
    threshs = []
    for v in picks:
        v = np.float32(v)
        threshs += [float(v), float(np.nextafter(v, np.float32(np.inf))), float(np.nextafter(v, np.float32(-np.inf))), float(v) + 0.004, float(v) - 0.004]
    threshs += [-30.0, 3.0, 40.0]
    dev = gpu.from_numpy(np.tile(st, (3, 1))).to("cuda:0")
    d = L.LoRaDemod(sf, n_channels=3)
    d.set_mode(1)
    d.setMTU(64)
    lens = set()
    for th in threshs:
        want = [p for _k, p in oracle.demod_run(sf, st, mtu=64, thresh=th, keep=False)["packets"]]
        d.setThreshold(th)
        d.activate()
        d.work(dev)                                              # no trace: the quick path
        got = {c: [] for c in range(3)}
        for c, _r, s in d.packets():
            got[c].append(s)
        for c in range(3):
            assert len(got[c]) == len(want) and all(np.array_equal(a, b) for a, b in zip(got[c], want)), (sf, th, c)
        lens.add(tuple(len(p) for p in want))
        # restore the state the next run starts from like the oracle's fresh block: a squelched window resets the tracking
        d.close()
        d = L.LoRaDemod(sf, n_channels=3)
        d.set_mode(1)
        d.setMTU(64)
    assert len(lens) >= 3                                        # the thresholds really moved the packet boundaries
    d.close()


def test_demod_on_a_second_device_while_the_first_is_current(gpu, oracle):
    """a demodulator created for device 1 while device 0 is current (its staging buffers must live on ITS device; create / destroy
    must leave the caller's current device alone); host-driven rounds and the streaming kernel -- needs two GPUs"""
    import lora_sdr_amd as L
    if gpu.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    rng = np.random.default_rng(4)
    st, syms = frames(oracle, rng, 7, 1, 12)
    gpu.cuda.set_device(0)
    for mode in MODES:
        d = L.LoRaDemod(7, n_channels=1, device=1)
        assert gpu.cuda.current_device() == 0
        d.set_mode(mode)
        d.setMTU(12)
        d.work([st])
        pk = d.packets()
        assert len(pk) == 1 and np.array_equal(pk[0][2], syms[0].astype(np.int16))
        d.close()
        assert gpu.cuda.current_device() == 0


@pytest.mark.parametrize("sf", [7, 9, 11])
def test_untraced_prefix_leaves_the_state_a_traced_run_has(gpu, oracle, sf):
    """Without a trace the streaming kernels evaluate only what the frame machine consumes (a quick squelch estimate, fIndex only
    where FRAMESYNC adds it to _finefreqError). Whatever they skip must not leak into the state: a stream is run (a) traced from
    the start and (b) untraced up to a cut, traced from there -- for cuts all along the stream -- and (b)'s traced calls must be
    the tail of (a)'s, including the fine-tune index and error each call starts from. The streams hold the case the shortcut must
    not get wrong: a FRAMESYNC call whose first sync word matches and whose SECOND window is squelched noise -- the reference adds
    that window's fIndex without looking at its snr (LoRaDemod.cpp:203, :217-221); the threshold is set so that noise IS squelched."""
    import lora_sdr_amd as L
    rng = np.random.default_rng(4242 + sf)
    N = 1 << sf
    B = 4
    streams = []
    for c in range(B):
        full = oracle.mod_frame(sf, rng.integers(0, N, 6).astype(np.uint16), sync=0x12, padding=0)
        cut_at = (10 + 1) * N + int(rng.integers(-N // 8, N // 8))               # preamble + the first sync chirp, then noise instead of the second
        good = oracle.mod_frame(sf, rng.integers(0, N, 6).astype(np.uint16), sync=0x12, padding=2)
        st = np.concatenate([np.zeros(int(rng.integers(5, N)), np.complex64), full[:cut_at], np.zeros(N + N // 3, np.complex64), good,
                             np.zeros(3 * N, np.complex64)])
        st = (st * np.exp(2j * np.pi * rng.uniform(-0.3, 0.3) / N * np.arange(st.size))).astype(np.complex64)
        st += (0.05 * (rng.standard_normal(st.size) + 1j * rng.standard_normal(st.size))).astype(np.complex64)
        streams.append(st)
    thresh = -6.0                                                                  # noise windows sit near -14 dB, chirps far above
    ref, has_case = [], []
    a = L.LoRaDemod(sf, n_channels=B); a.set_mode(1); a.setMTU(6); a.setThreshold(thresh); a.set_trace(True)
    a.work(streams)
    for c in range(B):
        tr = a.trace(c)
        want = oracle.demod_run(sf, streams[c], mtu=6, thresh=thresh)["calls"]
        compare_channel(tr, want)
        ref.append(tr)
        # the case is really in the stream: a FRAMESYNC call that is sync'd (the call before it peaked at bin 0, unsquelched), matches
        # the first sync word (0x12: (value + 4) / 8 == 1) and still consumes N - value -- its second window did not match
        has_case.append(any(x["state"] == 0 and (x["value"] + 4) // 8 == 0 and y["state"] == 0 and y["snr"] >= thresh and (y["value"] + 4) // 8 == 1
                            and y["consumed"] == N - y["value"] for x, y in zip(want, want[1:])))
    a.close()
    assert sum(has_case) >= 2                                                      # (where the cut came late the second sync word still matched)
    total = min(len(s) for s in streams)
    for cut in range(2 * N, total - 2 * N, (3 * N) // 2 + 7):
        d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(6); d.setThreshold(thresh)
        d.work([s[:cut] for s in streams])                                         # untraced: the shortcuts
        used = [d.consumed(c) for c in range(B)]
        n_before = []
        for c in range(B):
            pos, k = 0, 0
            while k < len(ref[c]) and pos < used[c]:
                pos += ref[c][k]["consumed"]; k += 1
            assert pos == used[c], "channel %d: the untraced run stopped inside a reference call" % c
            n_before.append(k)
        d.clear_packets()
        d.set_trace(True)
        d.work([s[u:] for s, u in zip(streams, used)])                             # traced from where it stopped
        for c in range(B):
            got, want = d.trace(c), ref[c][n_before[c]:]
            assert len(got) == len(want), (cut, c)
            for g, w in zip(got, want):
                for k in ("consumed", "state_before", "value", "fine_idx_before", "fine_idx_after", "packet_len"):
                    assert g[k] == w[k], (cut, c, k)
                assert np.float32(g["fine_err_before"]).tobytes() == np.float32(w["fine_err_before"]).tobytes(), (cut, c)
        d.close()


def test_state_stays_on_the_device_between_streaming_runs(gpu, oracle):
    """Streaming mode keeps the frame-machine state on the device from run to run (no per-run upload; the host mirrors are brought
    up to date lazily). A receiver fed chunk by chunk, switching between the streaming kernel and host-driven rounds, with an
    activate() in the middle, must deliver what a receiver that runs host-driven rounds throughout delivers: the same packets,
    the same consumption per chunk. The first chunk is handed over as a device tensor (equal lengths: the uniform placement the
    kernel computes itself), the later ones as per-channel host buffers of different lengths."""
    import lora_sdr_amd as L
    sf, N, B, mtu = 8, 256, 6, 10
    rng = np.random.default_rng(4242)
    streams = [frames(oracle, rng, sf, 4, mtu, off=rng.uniform(-0.4, 0.4), noise=0.05, lead=int(rng.integers(0, N)))[0] for _ in range(B)]
    plan = [(1, False), (1, False), (2, False), (1, True), (1, False), (0, False), (1, False), (1, False)]     # (mode, activate() first)
    n = min(len(s) for s in streams)
    chunk = n // len(plan)

    def run(modes):
        d = L.LoRaDemod(sf, n_channels=B)
        d.setMTU(mtu)
        rest = [np.zeros(0, np.complex64) for _ in range(B)]
        out, consumed = [[] for _ in range(B)], []
        for k, (mode, act) in enumerate(modes):
            bufs = [np.concatenate([rest[c], streams[c][k * chunk:(k + 1) * chunk if k + 1 < len(modes) else len(streams[c])]]) for c in range(B)]
            d.set_mode(mode)
            if act:
                d.activate()
            if k == 0:
                d.work(gpu.from_numpy(np.stack(bufs)).to("cuda:0"))
            else:
                d.work(bufs)
            for ch, _r, s in d.packets():
                out[ch].append(s)
            consumed.append([d.consumed(c) for c in range(B)])
            for c in range(B):
                rest[c] = bufs[c][d.consumed(c):]
        d.close()
        return out, consumed

    want, cw = run([(2, a) for _m, a in plan])
    got, cg = run(plan)
    assert cg == cw
    for c in range(B):
        assert len(got[c]) == len(want[c]) >= 2 and all(np.array_equal(a, b) for a, b in zip(got[c], want[c]))


@pytest.mark.parametrize("sf", [7, 11])
def test_near_threshold_counters(gpu, oracle, sf):
    """lorahip_demod_near_threshold: decisions within float rounding of their boundary are COUNTED (never altered). With the
    threshold placed exactly on the lowest snr a squelch decision sees, the call that consumed it is counted, by the streaming kernel (traced and
    untraced) and by the host-driven rounds alike; at the block's default threshold nothing is near; activate() restarts the count."""
    import lora_sdr_amd as L
    rng = np.random.default_rng(31 + sf)
    st, _ = frames(oracle, rng, sf, 2, 12, off=0.2, noise=0.3)
    r0 = oracle.demod_run(sf, st, mtu=64, keep=False)
    # the LOWEST snr any call sees whose squelch decision is consumed (FRAMESYNC / DATASYMBOLS): with the threshold exactly there
    # that call sits on the boundary (`snr < thresh` is false at equality) and every other decision keeps its side
    th = float(np.float32(min(c["snr"] for c in r0["calls"] if c["state"] in (0, 4) and np.isfinite(c["snr"]))))
    dev = gpu.from_numpy(st.reshape(1, -1)).to("cuda:0")
    seen = []
    for mode, trace in ((1, False), (1, True), (2, False)):
        d = L.LoRaDemod(sf, n_channels=1)
        d.set_mode(mode); d.setMTU(64); d.set_trace(trace)
        d.work(dev)
        assert d.near_threshold()[0] == 0                        # default threshold -30 dB: nothing near
        step0 = d.near_threshold()[1]
        d.close()
        d = L.LoRaDemod(sf, n_channels=1)
        d.set_mode(mode); d.setMTU(64); d.set_trace(trace); d.setThreshold(th)
        d.work(dev)
        ns, nstep = d.near_threshold()
        assert ns >= 1
        seen.append((ns, nstep, step0))
        d.activate()
        assert d.near_threshold() == (0, 0)
        d.close()
    assert seen[0] == seen[1] == seen[2]


@pytest.mark.parametrize("sf,sigma,pct", [(7, 2.0, None), (8, 1.2, 8), (9, 3.5, None), (10, 2.2, 5), (11, 3.5, 10)])
def test_many_noisy_channels_against_the_reference_runner(gpu, oracle, sf, sigma, pct):
    """Level-3 identity away from the clean bench workload: hundreds of channels with heavy noise (down to the sensitivity limit of
    the SF, where the sync search mis-fires, squelch decisions sit close to a raised threshold and packets are cut short), random
    leads and carrier offsets, the default threshold or one placed at a low percentile of the data symbols' own snr (so that
    squelch decisions fall on both sides all the time) -- every channel's packets AND every work() call's consumption / label kind
    against the all-channel CPU runner (the verbatim LoRaDemod.cpp where oracle/_ref travelled, else the pinned restatement),
    traced and untraced; the near-threshold counters of the two passes agree."""
    import lora_sdr_amd as L
    from oracle.oracle import Ref
    impl = Ref() if Ref.available() else oracle
    rng = np.random.default_rng(1234 + sf)
    N, mtu = 1 << sf, 16
    B = 384 if sf <= 9 else 96
    sts = []
    for c in range(B):
        st, _ = frames(oracle, rng, sf, 2, mtu, off=rng.uniform(-0.45, 0.45), noise=0.0, lead=int(rng.integers(0, 2 * N)))
        sts.append(st)
    n = max(len(s) for s in sts) + N
    x = np.zeros((B, n), np.complex64)
    for c, st in enumerate(sts):
        x[c, :len(st)] = st
    x += (sigma * (rng.standard_normal(x.shape) + 1j * rng.standard_normal(x.shape))).astype(np.complex64)
    thresh = -30.0
    if pct is not None:
        snr = [k["snr"] for c in range(3) for k in oracle.demod_run(sf, x[c], mtu=mtu, keep=False)["calls"] if k["state"] == 4]
        thresh = float(np.float32(np.percentile(snr, pct)))
    r = impl.demod_run_many(sf, x, thresh=thresh, mtu=mtu, nthreads=8, calls=True)
    dev = gpu.from_numpy(x).to("cuda:0")
    near = []
    for trace in (False, True):
        d = L.LoRaDemod(sf, n_channels=B)
        d.set_mode(1); d.setMTU(mtu); d.setThreshold(thresh); d.set_trace(trace)
        d.work(dev)
        near.append(d.near_threshold())
        ch, _rd, ln, sy = d.packets_arrays()
        start = np.concatenate([[0], np.cumsum(ln)])[:-1]
        for c in range(B):
            mine = np.nonzero(ch == c)[0]
            assert len(mine) == r["n_packets"][c], (sf, c, trace)
            at = 0
            for j, p in enumerate(mine):
                m = int(ln[p])
                assert m == r["pkt_lens"][c, j]
                assert np.array_equal(sy[start[p]:start[p] + m], r["pkt_syms"][c, at:at + m]), (sf, c, j, trace)
                at += m
        if trace:
            th = np.float32(thresh)
            for c in range(B):
                t = d.trace_array(c)
                k = int(r["n_calls"][c])
                assert t.size == k, (sf, c)
                st_, co = t["state_before"], t["consumed"]
                cls = np.where(st_ == 0, np.where(co == 2 * N, 1, np.where(~(t["snr"] < th), 2, 0)),
                               np.where(st_ == 1, 3, np.where(st_ == 2, 0, np.where(st_ == 3, 4, 5))))
                assert np.array_equal(co, r["consumed"][c, :k]) and np.array_equal(cls, r["cls"][c, :k]), (sf, c)
        d.close()
    assert near[0] == near[1]
    lens = r["pkt_lens"][r["pkt_lens"] > 0]
    assert lens.size >= B // 4 and (pct is None or np.unique(lens).size >= 8)     # hard, not hopeless; thresholds really cut packets short


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("sf", [7, 10, 11])
def test_streams_with_non_finite_and_extreme_samples(gpu, oracle, sf, mode):
    """A front end that hands over a NaN, an infinity, or a burst that overflows |X|^2. In the down-chirp, quarter-chirp and data
    states the reference's block (its -fcx-limited-range build: the textbook complex product, see
    tests/test_oracle_vs_ref.py::test_non_finite_samples_where_the_reference_is_defined) goes through it by the rules of IEEE
    arithmetic (NaN never wins the arg-max: index 0; snr NaN is not < threshold: NOT squelched) and so must the device's frame machine -- call by call with a trace, and packet by packet without
    one, where the squelch decision comes from the quick estimate and has to fall back to the exact chain on such windows.
    In FRAMESYNC the reference has NO defined behaviour for such a window: fIndex is NaN, `_finefreqError += fIndex`
    (LoRaDemod.cpp:221) makes the step NaN, `_fineTuneIndex -= NaN` (:162) converts NaN to int and the next table read is out
    of bounds -- the reference block (and the restatement) segfault on channels 1, 2 and 6 of this test, so nothing is compared
    there. What IS required of them: the launch completes, and the channels that share their wavefront are untouched."""
    import lora_sdr_amd as L
    rng = np.random.default_rng(900 + sf)
    N = 1 << sf
    B = 10
    streams = []
    for c in range(B):
        st, _ = frames(oracle, rng, sf, 2, 9, off=rng.uniform(-0.4, 0.4), noise=0.05, lead=int(rng.integers(0, N)))
        lead = st.size - (2 * (N * (10 + 2 + 2 + 9 + 3) + N // 4)) - 3 * N        # not exact, only to aim the faults into the frames
        first = max(lead, 0)
        spots = {0: [], 1: [first + 3 * N + 11], 2: [first + 11 * N + 5], 3: [first + 13 * N + 77], 4: [first + 16 * N + 1, first + 18 * N + 9],
                 5: [first + 15 * N + 3], 6: [first + 2 * N, first + 40 * N], 7: [first + 17 * N], 8: [5], 9: [first + 14 * N + N // 3]}[c]
        for k, p in enumerate(spots):
            p = min(p, st.size - 1)
            if c in (1, 2, 3, 4):
                st[p] = np.nan if (c + k) % 2 else complex(0.0, np.nan)
            elif c in (5, 6):
                st[p] = complex(np.inf, 0.0) if k == 0 else complex(-np.inf, np.inf)
            elif c == 7:
                st[p:p + N] *= np.float32(3e19)                    # |X|^2 overflows to +Inf in every bin's neighbourhood
            elif c == 8:
                st[:] *= np.float32(1e-21)                          # subnormal squares throughout
            elif c == 9:
                st[p] = complex(3e38, -3e38)
        streams.append(st)
    undefined = (1, 2, 6)                                           # the fault lands in FRAMESYNC: see above
    refs = []
    with np.errstate(all="ignore"):
        for c in range(B):
            refs.append(None if c in undefined else oracle.demod_run(sf, streams[c], mtu=12))
    assert sum(1 for r in refs if r is not None for x in r["calls"] if not np.isfinite(x["power"]) or not np.isfinite(x["fIndex"])) >= 6
    for traced in (True, False):
        d = L.LoRaDemod(sf, n_channels=B)
        d.set_mode(mode)
        d.setMTU(12)
        d.set_trace(traced)
        d.work(streams)
        pk = d.packets()
        for c in range(B):
            r = refs[c]
            if r is None:
                continue
            where = "sf%d mode %d channel %d traced %d" % (sf, mode, c, traced)
            if traced:
                tr = d.trace(c)
                assert len(tr) == len(r["calls"]), where
                for i, (a, b) in enumerate(zip(tr, r["calls"])):
                    assert (a["consumed"], a["state_before"], a["value"]) == (b["consumed"], b["state"], b["value"]), "%s call %d" % (where, i)
                    for ka, kb, tol in (("power", "power", TOL_DB), ("f_index", "fIndex", 2e-6)):
                        va, vb = float(a[ka]), float(b[kb])
                        assert (np.isnan(va), np.isposinf(va), np.isneginf(va)) == (np.isnan(vb), np.isposinf(vb), np.isneginf(vb)), "%s call %d %s" % (where, i, ka)
                        if np.isfinite(vb):
                            assert abs(va - vb) <= max(tol, 2 * float(np.spacing(np.float32(abs(vb))))), "%s call %d %s" % (where, i, ka)
            assert d.consumed(c) == int(sum(x["consumed"] for x in r["calls"])), where
            mine = [p[2] for p in pk if p[0] == c]
            assert len(mine) == len(r["packets"]), where
            for a, (_, b) in zip(mine, r["packets"]):
                assert np.array_equal(a, b), where
        assert all(d.consumed(c) > streams[c].size - 2 * N for c in undefined)      # they ran to the end of their streams


@pytest.mark.parametrize("sf", [10, 12])
def test_streams_beyond_2_pow_31_samples(gpu, oracle, sf):
    """Three channels of 2^30 + 2^26 samples each in one device buffer (27 GB of the 288): the third channel starts beyond sample
    2^31 of the buffer and every channel's own position passes 2^30 -- the streaming kernels' 64-bit positions and addresses. The
    streams are zeros with two frames at the very end; an all-zero window is not squelched (snr = -inf - -inf = NaN is not < thresh),
    does not sync (value 0 does not match the sync word) and consumes N - 0 samples (LoRaDemod.cpp:219), so the reference on the
    last part of the stream, started on a window boundary, sees exactly what the device sees there."""
    import lora_sdr_amd as L
    if gpu.cuda.get_device_properties(0).total_memory < 64 * 2**30:
        pytest.skip("needs 27 GB of device memory")
    rng = np.random.default_rng(31 + sf)
    N = 1 << sf
    LEN = 2**30 + 2**26
    B = 3
    buf = gpu.zeros((B, LEN), dtype=gpu.complex64, device="cuda")
    tails, zeros = [], []
    for c in range(B):
        st, _ = frames(oracle, rng, sf, 2, 7 + c, off=0.3 * c - 0.2, noise=0.02, lead=N // 3 + 17 * c)
        z = ((LEN - st.size) // N) * N                       # a whole number of windows of zeros in front
        tail = np.concatenate([st, np.zeros(LEN - z - st.size, np.complex64)])
        buf[c, z:] = gpu.from_numpy(tail).cuda()
        tails.append(tail)
        zeros.append(z)
    assert 2 * LEN > 2**31
    d = L.LoRaDemod(sf, n_channels=B)
    d.set_mode(1)
    d.setMTU(7)
    d.work(buf)
    pk = d.packets()
    calls = 0
    for c in range(B):
        r = oracle.demod_run(sf, tails[c], mtu=7)
        assert len(r["packets"]) >= 2
        mine = [p[2] for p in pk if p[0] == c]
        assert len(mine) == len(r["packets"]), "channel %d" % c
        for a, (_, b) in zip(mine, r["packets"]):
            assert np.array_equal(a, b), "channel %d" % c
        assert d.consumed(c) == zeros[c] + int(sum(x["consumed"] for x in r["calls"])), "channel %d" % c
        calls += zeros[c] // N + len(r["calls"])
    assert d.work_calls() == calls


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("sf", [8, 11])
def test_extreme_settings(gpu, oracle, sf, mode):
    """The setters are plain stores without validation (LoRaDemod.cpp:124-137): sync words 0x00 (the preamble itself matches the
    first nibble), 0xff, 0x0f, 0xf0; thresholds -inf (nothing squelched), +inf (everything with a finite snr squelched), NaN (`snr
    < NaN` is false: nothing squelched), +-1e30; an MTU larger than the stream has symbols. With a trace (exact chain) and without
    (quick squelch estimate) the calls, packets and consumption are the reference's."""
    import lora_sdr_amd as L
    rng = np.random.default_rng(77 + sf)
    N = 1 << sf
    streams = {}
    for sync in (0x00, 0xff, 0x0f, 0xf0, 0x12):
        syms = rng.integers(0, N, 7).astype(np.uint16)
        fr = oracle.mod_frame(sf, syms, sync=sync, padding=3)
        st = np.concatenate([np.zeros(N // 2 + 9, np.complex64), fr, fr, np.zeros(2 * N, np.complex64)])
        st = (st * np.exp(2j * np.pi * 0.21 / N * np.arange(st.size))).astype(np.complex64)
        st += (0.02 * (rng.standard_normal(st.size) + 1j * rng.standard_normal(st.size))).astype(np.complex64)
        streams[sync] = st
    inf = float("inf")
    cases = [(0x00, 64, 3.0), (0xff, 64, 3.0), (0x0f, 5, 3.0), (0xf0, 64, 3.0), (0x12, 5, -inf), (0x12, 64, inf), (0x12, 5, float("nan")),
             (0x12, 3, 1e30), (0x12, 6, -1e30), (0x12, 100000, 3.0), (0x00, 1, inf)]
    for sync, mtu, thresh in cases:
        st = streams[sync]
        with np.errstate(all="ignore"):
            r = oracle.demod_run(sf, st, sync=sync, mtu=mtu, thresh=thresh)
        for traced in (True, False):
            d = L.LoRaDemod(sf)
            d.set_mode(mode); d.setSync(sync); d.setMTU(mtu); d.setThreshold(thresh); d.set_trace(traced)
            d.work([st])
            where = "sf%d mode %d sync %#x mtu %d thresh %r traced %d" % (sf, mode, sync, mtu, thresh, traced)
            if traced:
                compare_channel(d.trace(0), r["calls"])
            assert d.work_calls() == len(r["calls"]), where
            assert d.consumed(0) == int(sum(x["consumed"] for x in r["calls"])), where
            pk = d.packets()
            assert len(pk) == len(r["packets"]), where
            assert all(np.array_equal(p[2], q) for p, (_, q) in zip(pk, r["packets"])), where
            d.close()


@pytest.mark.parametrize("sf", [7, 11])
def test_record_capacity_follows_what_a_receiver_needs(gpu, oracle, sf):
    """A receiver idling on noise above its threshold makes a work() call per N - value samples, about two per N (LoRaDemod.cpp:219)
    -- more than the per-launch record buffers hold at first, so the first run is resumed (several launches, records drained in
    between: correct, slower); the capacity then follows what the run needed and the next runs take ONE launch. The first run
    (fresh blocks) is compared with the reference; activate() keeps everything but the state and the table (LoRaDemod.cpp:139-143),
    so the later runs are compared with the host-driven mode of a second object that has been through the same runs."""
    import lora_sdr_amd as L
    rng = np.random.default_rng(5 + sf)
    N, B, S = 1 << sf, 6, 400 * (1 << sf) + 77
    x = (rng.standard_normal((B, S)) + 1j * rng.standard_normal((B, S))).astype(np.complex64)
    refs = [oracle.demod_run(sf, x[c], mtu=16) for c in range(B)]
    assert max(len(r["calls"]) for r in refs) > 1.5 * (S // N)
    d, h = L.LoRaDemod(sf, n_channels=B), L.LoRaDemod(sf, n_channels=B)
    d.set_mode(1); h.set_mode(2)
    launches = []
    for run in range(3):
        got = []
        for o in (d, h):
            o.setMTU(16)
            o.activate()
            o.work(gpu.from_numpy(x).cuda())
            got.append((o.work_calls(), [o.consumed(c) for c in range(B)], [(p[0], p[2].tolist()) for p in sorted(o.packets(), key=lambda p: (p[0], p[1]))]))
            o.clear_packets()
        launches.append(d.last_launches())
        assert got[0] == got[1], run
        assert len(got[0][2]) > 0 or sf > 7                 # noise makes false packets at SF7 (the test compares them too)
        if run == 0:
            assert got[0][0] == sum(len(r["calls"]) for r in refs)
            assert got[0][1] == [int(sum(k["consumed"] for k in r["calls"])) for r in refs]
            assert got[0][2] == [(c, q.tolist()) for c in range(B) for _, q in refs[c]["packets"]]
    assert launches[0] > 1 and launches[1] == 1 and launches[2] == 1, launches
    assert h.last_launches() == 0 and h.kernel_ms() == 0.0


@pytest.mark.parametrize("via", ["host queue", "device rows", "switching", "packets longer than the carry rows"])
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("sf", [7, 11])
def test_running_receiver_on_segments_of_one_device_buffer(gpu, oracle, sf, mode, via):
    """lorahip_demod_run_device_segments: the receiver behind a channeliser. All channels live in one (channels, capacity) device
    buffer that fills chunk by chunk; every work() is given, per channel, the samples between what that channel has consumed so far
    and what has been written so far -- the remainder of the last call (up to 2N - 1 samples, different from channel to channel)
    together with the new chunk, nothing copied. Packets, and the total consumption, are those of one work() over the whole stream
    and of the reference; with ports on, the traced consumption is too."""
    import lora_sdr_amd as L
    rng = np.random.default_rng(300 + sf)
    N, B = 1 << sf, 6
    streams = [frames(oracle, rng, sf, 3, 8 + c, off=rng.uniform(-0.4, 0.4), noise=0.05, lead=int(rng.integers(0, 2 * N)))[0] for c in range(B)]
    cap = max(s.size for s in streams)
    host = np.zeros((B, cap), np.complex64)
    for c, s in enumerate(streams):
        host[c, :s.size] = s
    # an MTU beyond 4096 symbols: open packets are handed from run to run through the host (lorahip.h); the packets end where the
    # padding is squelched (threshold 10 dB), several work() calls after they began
    mtu, thresh = (5000, 10.0) if via == "packets longer than the carry rows" else (8, -30.0)
    refs = [oracle.demod_run(sf, host[c], mtu=mtu, thresh=thresh) for c in range(B)]
    buf = gpu.zeros((B, cap), dtype=gpu.complex64, device="cuda")
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(mode); d.setMTU(mtu); d.setThreshold(thresh)
    read = np.zeros(B, np.int64)
    written, got, calls, step = 0, [[] for _ in range(B)], 0, 0
    with pytest.raises(ValueError):
        d.work_segments(buf, np.arange(B) * cap, np.full(B, cap + 1))
    while written < cap:
        n = min(cap - written, int(rng.integers(N // 2, 6 * N)))
        buf[:, written:written + n] = gpu.from_numpy(host[:, written:written + n]).cuda()      # "the channeliser's next chunk"
        written += n
        if via == "switching":
            # a packet that is open across work() calls is handed on by whichever side holds its symbols: the device's copy (streaming
            # mode), the mirrors (host-driven mode, traced runs, the host queue) -- every transition between them, mid-packet
            step += 1
            d.set_mode([1, 1, 2, 1, 2, 2, 1][step % 7] if mode == 1 else [2, 1, 1, 2][step % 4])
            d.set_trace(step % 5 == 3)
            if step % 6 == 4:
                d.setMTU(mtu)                                   # a setter in between: nothing changes
        d.work_segments(buf, np.arange(B) * cap + read, written - read)
        if via in ("device rows", "packets longer than the carry rows") or (via == "switching" and step % 3):
            # the decoder's input, packed on the device: in streaming mode the packets -- those that began in an earlier work() too --
            # never visit the host
            sy, ns, chn = d.packets_device(stride=64)
            sy, ns, chn = sy.cpu().numpy(), ns.cpu().numpy(), chn.cpu().numpy()
            for i in range(len(ns)):
                got[int(chn[i])].append(sy[i, :ns[i]].copy())
        else:
            for ch, _, s in d.packets():
                got[ch].append(s)
        assert d.consumed_all().tolist() == [d.consumed(c) for c in range(B)]
        for c in range(B):
            k = d.consumed(c)
            assert 0 <= k <= written - read[c]
            assert written - read[c] - k < 2 * N or k == 0 and written - read[c] < 2 * N      # LoRaDemod.cpp:148
            read[c] += k
        calls = d.work_calls()
    for c in range(B):
        r = refs[c]
        assert len(got[c]) == len(r["packets"]), "channel %d" % c
        assert len(got[c]) >= 3 or mtu > 8, "channel %d" % c
        assert all(np.array_equal(a, b) for a, (_, b) in zip(got[c], r["packets"])), "channel %d" % c
        assert read[c] == int(sum(k["consumed"] for k in r["calls"])), "channel %d" % c
    assert calls == sum(len(r["calls"]) for r in refs)
    assert sum(len(g) for g in got) >= 6
