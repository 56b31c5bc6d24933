"""GPU: the streaming kernels with more lanes per channel (lorahip_stream_lanes.hip, lorahip_demod_set_stream_lanes) -- the instances a
receiver with fewer channels than the device holds wavefronts runs on. Scheduling only: per-call traces, packets, signals, read
positions and chunked receiver steps must equal the CPU oracle's restated block (pinned to the verbatim LoRaDemod.cpp) and the
default 16-points-per-lane instances, bit for bit."""
import numpy as np
import pytest

from test_gpu_demod import frames, compare_channel

pytestmark = pytest.mark.gpu

AHEAD = 16                                                  # lorahip.h: LORAHIP_LANES_AHEAD | log2 lanes of one window -- two windows per channel and pass
INSTANCES = [(7, 4), (7, 5), (8, 5), (8, 6), (9, 6),        # (SF, log2 lanes per channel)
             (7, AHEAD | 3), (7, AHEAD | 4), (7, AHEAD | 5), (8, AHEAD | 4), (8, AHEAD | 5), (9, AHEAD | 5)]


def _host(oracle, rng, sf, B, n_frames, nsyms):
    N = 1 << sf
    streams = [frames(oracle, rng, sf, n_frames, nsyms + c % 4, off=rng.uniform(-0.45, 0.45), noise=0.05, lead=int(rng.integers(0, 2 * N)))[0] for c in range(B)]
    cap = max(s.size for s in streams)
    host = np.zeros((B, cap), np.complex64)
    for c, s in enumerate(streams):
        host[c, :s.size] = s
    return host


@pytest.mark.parametrize("sf,lanes", INSTANCES)
def test_traces_packets_and_signals_equal_the_reference(gpu, oracle, sf, lanes):
    """every work() call of every channel: consumption, state, value, fIndex, power; the packets; the signals of an untraced run"""
    import lora_sdr_amd as L
    rng = np.random.default_rng(5000 + 16 * sf + lanes)
    B = 11                                                    # not a multiple of the channels per wavefront / workgroup
    host = _host(oracle, rng, sf, B, 3, 9)
    refs = [oracle.demod_run(sf, host[c], mtu=10) for c in range(B)]
    iq = gpu.from_numpy(host).cuda()
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(10); d.set_stream_lanes(lanes); d.set_trace(True)
    assert d.stream_lanes() == lanes                          # the build holds the instance
    d.work(iq)
    pk = d.packets()
    for c in range(B):
        compare_channel(d.trace(c), refs[c]["calls"])
        mine = [p[2] for p in pk if p[0] == c]
        assert len(mine) == len(refs[c]["packets"]) >= 3 and all(np.array_equal(a, b) for a, (_, b) in zip(mine, refs[c]["packets"]))
    sig_traced = {}
    for c in range(B):
        tr = d.trace_array(c)
        sig_traced[c] = tr[tr["signals"] != 0]
    d.close()
    # untraced (the quick squelch estimate, fIndex only where consumed), with signals kept: same packets, same signal values
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(10); d.set_stream_lanes(lanes); d.set_signals(True)
    d.work(iq)
    pk2 = d.packets(clear=False)
    assert len(pk2) == len(pk) and all(a[0] == b[0] and np.array_equal(a[2], b[2]) for a, b in zip(pk, pk2))
    ch, rd, er, pw, sn = d.signals()
    for c in range(B):
        mine, want = ch == c, sig_traced[c]
        assert mine.sum() == want.size >= 3
        assert er[mine].tolist() == want["sig_error"].tolist() == [g[0] for g in refs[c]["signals"]]
        assert np.array_equal(pw[mine], want["sig_power"]) and np.array_equal(sn[mine], want["sig_snr"])
    for c in range(B):
        assert d.consumed(c) == int(sum(q["consumed"] for q in refs[c]["calls"]))
    d.close()


@pytest.mark.parametrize("async_", [True, 2])
@pytest.mark.parametrize("sf,lanes", INSTANCES)
def test_chunked_receiver_steps(gpu, oracle, sf, lanes, async_):
    """the running receiver on these instances: chunks of a few windows, packets open across chunks (carry rows), a small record
    capacity so that launches are resumed; all steps together give the reference's packets and call count"""
    import lora_sdr_amd as L
    rng = np.random.default_rng(6000 + 16 * sf + lanes)
    N, B = 1 << sf, 9
    host = _host(oracle, rng, sf, B, 3, 20)
    cap = host.shape[1]
    refs = [oracle.demod_run(sf, host[c], mtu=24) for c in range(B)]
    iq = gpu.from_numpy(host).cuda()
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(24); d.set_stream_lanes(lanes)
    d.set_record_capacity(9)
    rows = d.receiver_rows(cap_packets=8 * B, stride=32)
    got, calls, w = [[] for _ in range(B)], 0, 0

    def take(n):
        gpu.cuda.synchronize()
        sy, ns, chn = rows[0][:n].cpu().numpy(), rows[1][:n].cpu().numpy(), rows[2][:n].cpu().numpy()
        for i in range(n):
            got[int(chn[i])].append(sy[i, :ns[i]].copy())
    while w < cap:
        w = min(cap, w + int(rng.integers(N, 7 * N)))
        n, c_ = d.receive(iq, w, rows, async_=async_)
        take(n); calls += c_
    if async_ == 2:
        n, c_ = d.receive_flush(rows)
        take(n); calls += c_
    for c in range(B):
        r = refs[c]
        assert len(got[c]) == len(r["packets"]) >= 3, "channel %d" % c
        assert all(np.array_equal(a, b) for a, (_, b) in zip(got[c], r["packets"])), "channel %d" % c
        assert d.consumed(c) == int(sum(q["consumed"] for q in r["calls"]))
    assert calls == sum(len(r["calls"]) for r in refs) == d.work_calls()
    d.close()


@pytest.mark.parametrize("sf", [7, 8, 9])
def test_every_lane_choice_gives_the_same_records(gpu, oracle, sf):
    """one workload through the default choice, the forced 16-points-per-lane geometry and every wider instance of the SF: the packets,
    the per-channel call counts and the near-threshold counters are identical (the choice is scheduling, nothing else)"""
    import lora_sdr_amd as L
    rng = np.random.default_rng(7000 + sf)
    B = 37
    host = _host(oracle, rng, sf, B, 2, 12)
    iq = gpu.from_numpy(host).cuda()
    outs = []
    for lanes in [0, -1] + [l for s_, l in INSTANCES if s_ == sf]:
        d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(12); d.set_stream_lanes(lanes)
        d.work(iq)
        ch, rd, ln, sy = d.packets_arrays()
        outs.append((ch.tolist(), rd.tolist(), ln.tolist(), sy.tolist(), d.consumed_all().tolist(), d.work_calls(), d.near_threshold()))
        d.close()
    assert all(o == outs[0] for o in outs[1:])
    assert len(outs[0][0]) >= 2 * B


def test_the_choice_follows_the_channel_count(gpu):
    """16 points per lane once the channels fill the device's wavefront slots (two per SIMD), more lanes per channel below that"""
    import lora_sdr_amd as L
    import torch
    slots = torch.cuda.get_device_properties(0).multi_processor_count * 8
    for sf, base in ((7, 3), (8, 4), (9, 5), (10, 6), (12, 8)):
        per_wave = max(1, 64 >> base)
        full = L.LoRaDemod(sf, n_channels=slots * per_wave)
        assert full.stream_lanes() == base
        full.close()
        few = L.LoRaDemod(sf, n_channels=max(1, slots * per_wave // 8))
        assert few.stream_lanes() == (min(base + 2, 6) if sf <= 9 else base)
        few.set_stream_lanes(-1)
        assert few.stream_lanes() == base
        few.set_stream_lanes(6)
        assert few.stream_lanes() == (6 if sf in (8, 9) else base)
        few.close()
    # a sixteenth of the slots at SF7: two groups of 32 lanes per channel, the second a window ahead (lorahip_stream_pairs.hip)
    fewer = L.LoRaDemod(7, n_channels=slots // 2)
    assert fewer.stream_lanes() == (AHEAD | 5)
    fewer.set_stream_lanes(5)
    assert fewer.stream_lanes() == 5
    fewer.close()


def test_the_parts_of_a_mixed_object_count_their_siblings(gpu):
    """a part of lorahip_demod_create_mixed shares the device with the other parts: wider lanes only into slots nobody else takes"""
    import lora_sdr_amd as L
    import torch
    slots = torch.cuda.get_device_properties(0).multi_processor_count * 8
    # BASELINE configs[3]: 16384 channels, SF = 7 + c mod 6 -- the SF10-12 parts alone fill the device several times over
    big = L.LoRaDemod(channel_sf=(7 + np.arange(16384) % 6).astype(np.int32), devices=[0])
    assert [p[1] for p in big.parts] == [7, 8, 9, 10, 11, 12]
    assert big.part_stream_lanes() == [3, 4, 5, 6, 7, 8]
    big.set_stream_lanes(5)                                              # asked for: as asked
    assert big.part_stream_lanes()[0] == 5
    big.set_stream_lanes(0)
    assert big.part_stream_lanes()[0] == 3
    big.close()
    # a handful of channels per SF: every part as wide as it would be alone
    n = max(6, slots // 64 * 6)
    small = L.LoRaDemod(channel_sf=(7 + np.arange(n) % 3).astype(np.int32), devices=[0])
    assert small.part_stream_lanes() == [AHEAD | 5, 6, 6]
    small.close()
    # the same results either way (scheduling only): the mixed object against its parts alone is tests/test_gpu_mixed.py
