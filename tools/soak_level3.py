"""Randomised level-3 soak against the CPU oracle (the restated block, pinned to the verbatim LoRaDemod.cpp): random SF, channel
count, MTU, threshold, sync word, carrier offset, noise, stream grid, lanes per channel and chunking; packets, call counts and read positions must be
the reference's in every case.   python tools/soak_level3.py [seconds] [first seed]
(tests/ hold the fixed-seed versions of these cases; this is the same comparison over many more parameter combinations.)"""
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
cases = calls_total = packets_total = signals_total = 0
per_how = {}
per_sf = {}
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    sf = int(rng.integers(7, 13)); N = 1 << sf
    if "SOAK_SFS" in os.environ: sf = int(rng.choice([int(x) for x in os.environ["SOAK_SFS"].split(",")])); N = 1 << sf      # e.g. SOAK_SFS=7,8,9 with SOAK_LANES_SET=19,20,21
    B = int(rng.integers(1, 24 if sf < 11 else 10))
    mtu = int(rng.integers(3, 40)); thresh = float(rng.uniform(-40, -5)); sync = int(rng.integers(0, 256)) if rng.random() < 0.3 else 0x12
    streams = []
    for c in range(B):
        s, _ = frames(oracle, rng, sf, int(rng.integers(1, 4)), int(rng.integers(2, 30)), off=float(rng.uniform(-0.45, 0.45)), noise=float(rng.uniform(0.0, 0.3)),
                      sync=sync, lead=int(rng.integers(0, 3 * N)))
        streams.append(s)
    cap = max(s.size for s in streams)
    cap += -cap % 16                                     # rows of whole 128-byte lines (what the resident receiver asks for)
    host = np.zeros((B, cap), np.complex64)
    for c, s in enumerate(streams): host[c, :s.size] = s
    refs = [oracle.demod_run(sf, host[c], sync=sync, thresh=thresh, mtu=mtu) for c in range(B)]
    iq = torch.from_numpy(host).cuda()
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(mtu); d.setThreshold(thresh); d.setSync(sync)
    d.set_stream_grid(int(rng.choice([0, -1, 1, 2, 5])))
    # SF7-9: more lanes per channel (lorahip_stream_lanes.hip), and 16 | l: two windows of 2^l lanes per channel, the second one ahead (lorahip_stream_pairs.hip)
    lanes_set = [int(x) for x in os.environ["SOAK_LANES_SET"].split(",")] if "SOAK_LANES_SET" in os.environ else [0, -1, 4, 5, 6, 19, 20, 21]
    lanes = int(rng.choice(lanes_set)) if os.environ.get("SOAK_LANES", "1") != "0" else 0
    d.set_stream_lanes(lanes)
    how = int(rng.integers(0, 4)) if "SOAK_HOW" not in os.environ else (int(rng.integers(0, 4)), int(os.environ["SOAK_HOW"]))[1]                        # 0 one shot, 1 sequential steps, 2 pipelined, 3 resident
    sigs = rng.random() < 0.5                            # the block's signals (error / power / snr at DOWNCHIRP1) kept and compared too
    got = [[] for _ in range(B)]
    got_sig = [[] for _ in range(B)]
    d.set_signals(bool(sigs))
    if how == 0:
        d.work(iq)
        if sigs:
            ch_, _rd, er_, po_, sn_ = d.signals()
            for i in range(len(ch_)): got_sig[int(ch_[i])].append((int(er_[i]), float(po_[i]), float(sn_[i])))
        for ch, _r, q in d.packets(): got[ch].append(q)
        ncalls = d.work_calls()
    else:
        depth = int(rng.integers(1, 4)) if how == 3 else 1  # resident: steps in flight (the caller cycles depth + 1 sets of rows)
        NS = depth + 1
        rows = [d.receiver_rows(cap_packets=B * 40, stride=max(mtu, 8)) for _ in range(NS)]
        pin_ = bool(rng.random() < 0.5)
        srows = [d.receiver_signal_rows(B * 48, pinned_host=pin_) for _ in range(NS)] if sigs else None      # cycling like the packet rows
        w = k = ncalls = 0
        pending = []
        def take(n, r, srow=None, m=0):
            # (never a DEVICE-wide synchronise while the resident kernel is on the device: it would wait for the flush -- and the rows of a
            # resident step are complete in memory when the call that reports them returns)
            if not (how == 3 and d.resident_active()): torch.cuda.synchronize()
            sy, ns, chn = r[0][:n].cpu().numpy(), r[1][:n].cpu().numpy(), r[2][:n].cpu().numpy()
            for i in range(n): got[int(chn[i])].append(sy[i, :ns[i]].copy())
            if sigs and m:
                sc, se, sp, ss = (np.asarray(t_[:m].cpu() if hasattr(t_, "cpu") else t_[:m]) for t_ in srow)
                for i in range(m): got_sig[int(sc[i])].append((int(se[i]), float(sp[i]), float(ss[i])))
        while w < cap:
            w = min(cap, w + int(rng.integers(N // 2, 9 * N)))
            j = k % NS
            if sigs: d.register_signal_rows(srows[j])
            n, c_ = d.receive(iq, w, rows[j], async_=(how if how in (2, 3) else True), depth=depth)
            ncalls += c_; k += 1
            if how == 3 and d.resident_active():
                # a resident step fills the rows that came with ITS call and is reported `depth` calls later
                pending.append(j)
                for pk_, sg_ in d.last_steps():
                    jj = pending.pop(0); take(pk_, rows[jj], srows[jj] if sigs else None, sg_)
            else:
                take(n, rows[j], srows[j] if sigs else None, d.last_signals() if sigs else 0)
        if how in (2, 3):
            j = k % NS
            if sigs: d.register_signal_rows(srows[j])
            n, c_ = d.receive_flush(rows[j]); ncalls += c_
            steps_ = d.last_steps() if pending else []
            for pk_, sg_ in steps_:
                jj = pending.pop(0); take(pk_, rows[jj], srows[jj] if sigs else None, sg_)
            take(n - sum(p_ for p_, _ in steps_), rows[j], srows[j] if sigs else None, (d.last_signals() if sigs else 0) - sum(s_ for _, s_ in steps_))
        if sigs:
            d.receiver_signal_rows(0)
    print("case seed %d: SF%d, %d channels, mtu %d, mode %d, signals %s" % (seed, sf, B, mtu, how, bool(sigs)), file=sys.stderr, flush=True) if os.environ.get("SOAK_VERBOSE") else None
    want_calls = sum(len(r["calls"]) for r in refs)
    assert ncalls == want_calls, ("calls", seed, sf, B, how, ncalls, want_calls)
    for c, r in enumerate(refs):
        assert len(got[c]) == len(r["packets"]), ("packet count", seed, sf, c, how)
        assert all(np.array_equal(a, b) for a, (_, b) in zip(got[c], r["packets"])), ("packet symbols", seed, sf, c, how)
        assert d.consumed(c) == int(sum(k_["consumed"] for k_ in r["calls"])), ("consumed", seed, sf, c, how)
        if sigs:
            assert [g[0] for g in got_sig[c]] == [int(q[0]) for q in r["signals"]], ("signal errors", seed, sf, c, how)
            if got_sig[c]:
                # (a silent window at DOWNCHIRP1 gives power -inf and snr NaN in the reference too: equal non-finite values are equal)
                assert np.allclose(np.array([g[1:] for g in got_sig[c]], np.float64), np.array([q[1:] for q in r["signals"]], np.float64), rtol=0, atol=2e-5, equal_nan=True), \
                    ("signal values", seed, sf, c, how, [g[1:] for g in got_sig[c]], [tuple(q[1:]) for q in r["signals"]])
            signals_total += len(got_sig[c])
    d.close()
    cases += 1; calls_total += want_calls; packets_total += sum(len(r["packets"]) for r in refs)
    per_sf[sf] = per_sf.get(sf, 0) + 1
    per_how[how] = per_how.get(how, 0) + 1
    seed += 1
print("level-3 soak: %d random cases (seeds up to %d; per SF %s; per mode [one shot, sequential, pipelined, resident] %s), %d work() calls, %d packets, %d signals: every "
      "channel's packets, call count, read position -- and signals where kept -- equal the reference's"
      % (cases, seed - 1, dict(sorted(per_sf.items())), [per_how.get(i, 0) for i in range(4)], calls_total, packets_total, signals_total))
