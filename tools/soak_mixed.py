"""Randomised soak of the MIXED object (lorahip_demod_create_mixed: one handle, a part per device entry and SF, a stream and a host thread
each) against the CPU oracle: random SF per channel, channel count, device list (entries of the one device repeated: several shards),
MTU, threshold, sync word, segment placement in ONE device buffer, forced lanes, signals; every channel's packets, read position, the
call total -- and signals where kept -- must be the reference's, as if each channel were its own LoRaDemod block (LoRaDemod.cpp:119-122).
    python tools/soak_mixed.py [seconds] [first seed]
(tests/test_gpu_mixed.py and test_gpu_receiver.py hold the fixed cases; tools/soak_level3.py is the single-SF soak in the four receiver modes.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import lora_sdr_amd as L
from oracle.oracle import Oracle
from test_gpu_demod import frames

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
oracle = Oracle()
t_end = time.time() + budget
cases = calls_total = packets_total = signals_total = parts_total = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    sfs_pool = rng.choice(np.arange(7, 13), size=int(rng.integers(1, 7)), replace=False)
    B = int(rng.integers(1, 40))
    sf_of = rng.choice(sfs_pool, size=B).astype(np.int32)
    devices = [0] * int(rng.integers(1, 4))
    mtu = int(rng.integers(3, 40)); thresh = float(rng.uniform(-40, -5)); sync = int(rng.integers(0, 256)) if rng.random() < 0.3 else 0x12
    streams = []
    for c in range(B):
        sf = int(sf_of[c]); N = 1 << sf
        s, _ = frames(oracle, rng, sf, int(rng.integers(1, 3)), int(rng.integers(2, 20)), off=float(rng.uniform(-0.45, 0.45)), noise=float(rng.uniform(0.0, 0.3)),
                      sync=sync, lead=int(rng.integers(0, 3 * N)))
        streams.append(s.astype(np.complex64))
    refs = [oracle.demod_run(int(sf_of[c]), streams[c], sync=sync, thresh=thresh, mtu=mtu) for c in range(B)]
    # the segments in one buffer, in a random order, with random gaps (what a channeliser's output rows look like)
    order = rng.permutation(B)
    first = np.zeros(B, np.int64); cnt = np.zeros(B, np.uint64); at = int(rng.integers(0, 64))
    for c in order:
        first[c] = at; cnt[c] = streams[c].size; at += streams[c].size + int(rng.integers(0, 200))
    host = (rng.standard_normal(at) + 1j * rng.standard_normal(at)).astype(np.complex64)       # the gaps hold noise nobody may read
    for c in range(B): host[first[c]:first[c] + streams[c].size] = streams[c]
    buf = torch.from_numpy(host).cuda()
    d = L.LoRaDemod(channel_sf=sf_of, devices=devices)
    d.setMTU(mtu); d.setThreshold(thresh); d.setSync(sync)
    d.set_stream_lanes(int(rng.choice([0, 0, -1, 5, 21])))
    sigs = bool(rng.random() < 0.5)
    d.set_signals(sigs)
    passes = 1                                           # (a second run of the same object starts from the members the first left behind -- _prevValue, the fine-tune state: activate() resets the state only, LoRaDemod.cpp:139-143 -- which a fresh reference block does not have)
    calls_before = 0
    for p in range(passes):
        d.clear_packets(); d.activate()
        if len(devices) == 1: d.work_segments(buf, first, cnt)
        else: d.work_segments_multi([buf] * len(devices), first, cnt)        # (every entry is the one device: the one buffer)
        got_sig = [[] for _ in range(B)]
        if sigs:
            ch_, _rd, er_, po_, sn_ = d.signals()
            for i in range(len(ch_)): got_sig[int(ch_[i])].append((int(er_[i]), float(po_[i]), float(sn_[i])))
        got = [[] for _ in range(B)]
        for ch, _r, q in d.packets(): got[ch].append(q)
        want_calls = sum(len(r["calls"]) for r in refs)
        assert d.work_calls() - calls_before == want_calls, ("calls", seed, p, d.work_calls() - calls_before, want_calls)   # (the count runs on over the object's life)
        calls_before = d.work_calls()
        for c, r in enumerate(refs):
            assert len(got[c]) == len(r["packets"]), ("packet count", seed, p, c, int(sf_of[c]))
            assert all(np.array_equal(a, b) for a, (_, b) in zip(got[c], r["packets"])), ("packet symbols", seed, p, c, int(sf_of[c]))
            assert d.consumed(c) == int(sum(k_["consumed"] for k_ in r["calls"])), ("consumed", seed, p, c, int(sf_of[c]))
            if sigs:
                assert [g[0] for g in got_sig[c]] == [int(q[0]) for q in r["signals"]], ("signal errors", seed, p, c)
                if got_sig[c]:
                    assert np.allclose(np.array([g[1:] for g in got_sig[c]], np.float64), np.array([q[1:] for q in r["signals"]], np.float64), rtol=0, atol=2e-5, equal_nan=True), \
                        ("signal values", seed, p, c)
    signals_total += sum(len(g) for g in got_sig)
    parts_total += len(d.parts)
    d.close()
    cases += 1; calls_total += want_calls; packets_total += sum(len(r["packets"]) for r in refs)
    seed += 1
print("mixed-object soak: %d random cases (seeds up to %d), %d parts, %d work() calls, %d packets, %d signals: every channel's packets, read position, "
      "the call total -- and signals where kept -- equal the reference's" % (cases, seed - 1, parts_total, calls_total, packets_total, signals_total))
