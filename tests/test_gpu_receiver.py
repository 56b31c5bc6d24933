"""GPU: the round-4 additions to level 3 of the C ABI -- signals without a trace (lorahip_demod_set_signals), the running receiver in
one call (lorahip_demod_run_device_append / lorahip_demod_receive) and ONE object over mixed spreading factors and several devices
(lorahip_demod_create_mixed) -- against the CPU oracle's restated block (pinned to the verbatim LoRaDemod.cpp) and against the
paths that were there before."""
import numpy as np
import pytest

from test_gpu_demod import frames, MODES, TOL_DB

pytestmark = pytest.mark.gpu


def _streams(oracle, rng, sf, B, n_frames=3, nsyms0=8):
    N = 1 << sf
    streams = [frames(oracle, rng, sf, n_frames, nsyms0 + c % 5, off=rng.uniform(-0.4, 0.4), noise=0.05, lead=int(rng.integers(0, 2 * N)))[0]
               for c in range(B)]
    cap = max(s.size for s in streams)
    host = np.zeros((B, cap), np.complex64)
    for c, s in enumerate(streams):
        host[c, :s.size] = s
    return host


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("sf", [7, 9, 10, 11, 12])
def test_signals_without_a_trace(gpu, oracle, sf, mode):
    """error / power / snr at DOWNCHIRP1 (LoRaDemod.cpp:267-269) from the kernels' per-emission records: equal to the traced run's
    sig_* fields and to the reference's values, and nothing else about the run changes (packets, consumption, call count)"""
    import lora_sdr_amd as L
    rng = np.random.default_rng(500 + sf)
    B = 9
    host = _streams(oracle, rng, sf, B)
    iq = gpu.from_numpy(host).cuda()
    refs = [oracle.demod_run(sf, host[c], mtu=8) for c in range(B)]
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(mode); d.setMTU(8)
    d.set_signals(True)
    d.work(iq)
    ch, rd, er, pw, sn = d.signals()
    pk = d.packets(clear=False)
    assert len(d.signals()[0]) == len(ch)                           # reading does not consume
    d.clear_packets()
    assert len(d.signals()[0]) == 0                                 # ... clearing does
    calls_plain = d.work_calls()
    # the traced run of the same object (a fresh activation on the same streams)
    t = L.LoRaDemod(sf, n_channels=B); t.set_mode(mode); t.setMTU(8); t.set_trace(True)
    t.work(iq)
    assert t.work_calls() == calls_plain == sum(len(r["calls"]) for r in refs)
    n_sig = 0
    for c in range(B):
        tr = t.trace_array(c)
        want = tr[tr["signals"] != 0]
        mine = ch == c
        assert mine.sum() == want.size >= 3
        assert rd[mine].tolist() == np.nonzero(tr["signals"])[0].tolist()       # the call (round) that emitted
        assert er[mine].tolist() == want["sig_error"].tolist()
        assert np.array_equal(pw[mine], want["sig_power"]) and np.array_equal(sn[mine], want["sig_snr"])
        ref_sig = refs[c]["signals"]                                # (error, power, snr) per emission
        assert er[mine].tolist() == [g[0] for g in ref_sig]
        assert np.allclose(pw[mine], [g[1] for g in ref_sig], rtol=0, atol=TOL_DB)
        assert np.allclose(sn[mine], [g[2] for g in ref_sig], rtol=0, atol=2 * TOL_DB)
        n_sig += int(mine.sum())
        assert [q.tolist() for cc, _, q in pk if cc == c] == [q.tolist() for _, q in refs[c]["packets"]]
    assert n_sig == ch.size
    assert t.signals()[0].size == 0                                 # signals are kept only when asked for
    d.close(); t.close()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("sf", [7, 10, 11])
def test_append_runs_continue_every_channel(gpu, oracle, sf, mode):
    """lorahip_demod_run_device_append: the (channels, capacity) buffer fills chunk by chunk, each run is given the new fill level and
    nothing else; every channel continues at its own read position. Packets, consumption and calls are the one-shot run's and the
    reference's; a rewind starts over and gives the same again."""
    import lora_sdr_amd as L
    rng = np.random.default_rng(600 + sf)
    N, B = 1 << sf, 7
    host = _streams(oracle, rng, sf, B)
    cap = host.shape[1]
    refs = [oracle.demod_run(sf, host[c], mtu=8) for c in range(B)]
    buf = gpu.zeros((B, cap + 3), dtype=gpu.complex64, device="cuda")       # a row stride that is not the fill level
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(mode); d.setMTU(8)
    for rep in range(2):
        written, got = 0, [[] for _ in range(B)]
        c0 = d.work_calls()
        while written < cap:
            n = min(cap - written, int(rng.integers(N // 2, 7 * N)))
            buf[:, written:written + n] = gpu.from_numpy(host[:, written:written + n]).cuda()
            written += n
            d.work_append(buf, written)
            for ch, _, s in d.packets():
                got[ch].append(s)
            pos = d.consumed_all()
            assert ((written - pos < 2 * N) & (pos >= 0)).all()             # LoRaDemod.cpp:148: fewer than 2N left everywhere
        for c in range(B):
            r = refs[c]
            assert len(got[c]) == len(r["packets"]) >= 3
            assert all(np.array_equal(a, b) for a, (_, b) in zip(got[c], r["packets"]))
            assert d.consumed(c) == int(sum(k["consumed"] for k in r["calls"]))
        if rep == 0:
            # (a re-activated receiver starts from what the last stream left in _prevValue / _finefreqError, like the reference block
            # would: the packets are the same, the number of calls it takes to lock need not be)
            assert d.work_calls() - c0 == sum(len(r["calls"]) for r in refs)
        with pytest.raises(L.LoraHipError):
            d.work_append(buf, written - 1)                                 # the fill level cannot shrink
        d.rewind(); d.activate()
    d.close()


@pytest.mark.parametrize("sf", [7, 10, 12])
def test_receive_packs_the_packets_of_every_step_on_the_device(gpu, oracle, sf):
    """lorahip_demod_receive: append run + device-side packing (rows numbered on the device) + clear, one call per chunk; the rows of
    all steps together are the reference's packets, those that span chunks included"""
    import lora_sdr_amd as L
    rng = np.random.default_rng(700 + sf)
    N, B = 1 << sf, 11
    host = _streams(oracle, rng, sf, B, n_frames=4)
    cap = host.shape[1]
    refs = [oracle.demod_run(sf, host[c], mtu=9) for c in range(B)]
    iq = gpu.from_numpy(host).cuda()
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(9)
    rows = d.receiver_rows(cap_packets=64, stride=16)
    got, calls, w = [[] for _ in range(B)], 0, 0
    while w < cap:
        w = min(cap, w + int(rng.integers(N, 9 * N)))
        n, k = d.receive(iq, w, rows, async_=bool(rng.integers(0, 2)))
        gpu.cuda.synchronize()
        calls += k
        sy, ns, chn = rows[0][:n].cpu().numpy(), rows[1][:n].cpu().numpy(), rows[2][:n].cpu().numpy()
        assert (np.diff(chn) >= 0).all()                                    # rows: channels ascending
        for i in range(n):
            assert (sy[i, ns[i]:] == 0).all()
            got[int(chn[i])].append(sy[i, :ns[i]].copy())
        assert d.packets() == []                                            # the queue is cleared by the call
    for c in range(B):
        r = refs[c]
        assert len(got[c]) == len(r["packets"]) >= 4
        assert all(np.array_equal(a, b) for a, (_, b) in zip(got[c], r["packets"]))
    assert calls == sum(len(r["calls"]) for r in refs) == d.work_calls()
    # too few rows: refused, the packets stay queued
    d.rewind(); d.activate()
    small = d.receiver_rows(cap_packets=1, stride=16)
    with pytest.raises(L.LoraHipError):
        d.receive(iq, cap, small)
    assert len(d.packets()) == sum(len(r["packets"]) for r in refs)
    d.close()


@pytest.mark.parametrize("sf", [7, 11, 12])
def test_stream_grid_is_scheduling_only(gpu, oracle, sf):
    """lorahip_demod_set_stream_grid: however the channels are spread over workgroups -- one workgroup per channel, a few workgroups
    each walking channel after channel (the default at SF11 once the channels outnumber the resident workgroups), chunked or in one
    run -- packets, signals, call counts and read positions are the reference's."""
    import lora_sdr_amd as L
    rng = np.random.default_rng(900 + sf)
    N, B = 1 << sf, 11
    host = _streams(oracle, rng, sf, B, n_frames=3)
    cap = host.shape[1]
    refs = [oracle.demod_run(sf, host[c], mtu=9) for c in range(B)]
    iq = gpu.from_numpy(host).cuda()
    want_calls = sum(len(r["calls"]) for r in refs)
    for grid in (0, -1, 1, 3, 8, 64):
        d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(9); d.set_signals(True)
        d.set_stream_grid(grid)
        d.work(iq)
        assert d.work_calls() == want_calls, "grid %d" % grid
        pk = d.packets(clear=False)
        ch, rd, er, pw, sn = d.signals()
        for c in range(B):
            assert [q.tolist() for cc, _, q in pk if cc == c] == [q.tolist() for _, q in refs[c]["packets"]], "grid %d channel %d" % (grid, c)
            assert d.consumed(c) == int(sum(k["consumed"] for k in refs[c]["calls"]))
            assert er[ch == c].tolist() == [g[0] for g in refs[c]["signals"]]
        d.close()
    # the running receiver in chunks, three workgroups for eleven channels: open packets cross both chunk and channel-walk boundaries
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(9); d.set_stream_grid(3)
    rows = d.receiver_rows(cap_packets=64, stride=16)
    got, calls, w = [[] for _ in range(B)], 0, 0
    while w < cap:
        w = min(cap, w + int(rng.integers(N, 7 * N)))
        n, c_ = d.receive(iq, w, rows)
        calls += c_
        sy, ns, chn = rows[0][:n].cpu().numpy(), rows[1][:n].cpu().numpy(), rows[2][:n].cpu().numpy()
        for i in range(n):
            got[int(chn[i])].append(sy[i, :ns[i]].copy())
    assert calls == want_calls
    for c, r in enumerate(refs):
        assert len(got[c]) == len(r["packets"]) >= 3
        assert all(np.array_equal(a_, b_) for a_, (_, b_) in zip(got[c], r["packets"]))
    d.close()


def test_sf11_beyond_the_resident_set(gpu):
    """More SF11 channels than the device holds workgroups: the library's own grid (the resident number of workgroups, each walking
    channel after channel -- lorahip_wide.hip) against one workgroup per channel: the same calls, read positions and packets, and the
    packets carry the sent symbols."""
    import lora_sdr_amd as L
    from lora_sdr_amd import workloads as WL
    ctx = L.Context(11)
    B = 1100
    iq, data = WL.frame_streams(ctx, B, 2, 12, sigma=0.05)
    res = []
    for grid in (0, -1):
        d = L.LoRaDemod(11, n_channels=B); d.set_mode(1); d.setMTU(12); d.set_stream_grid(grid)
        d.work(iq)
        # 2200 packets of 1100 channels packed on the device (row numbering by prefix sum over the channels): rows by channel, then
        # time, zero padded
        ps, pn, pc = d.packets_device(stride=16, clear=False)
        pk = d.packets()
        by_channel = sorted(pk, key=lambda e: (e[0], e[1]))
        assert pc.cpu().tolist() == [c for c, _, _ in by_channel] and pn.cpu().tolist() == [len(q) for _, _, q in by_channel]
        rows_ = ps.cpu().numpy()
        assert all(np.array_equal(rows_[i, :len(q)], q) and not rows_[i, len(q):].any() for i, (_, _, q) in enumerate(by_channel))
        res.append((d.work_calls(), [d.consumed(c) for c in range(0, B, 53)], sorted((c, r, tuple(q.tolist())) for c, r, q in pk)))
        if grid == 0:
            n, ok = WL.check_frame_packets(pk, data, 1 << 11, 12)
            assert n == 2 * B and ok >= n - 4                       # (a noise-triggered early end is the reference's behaviour too)
        d.close()
    assert res[0] == res[1]
    ctx.close()


@pytest.mark.parametrize("sf", [7, 9, 10, 12])
def test_pipelined_receiver_delivers_every_packet_one_step_late(gpu, oracle, sf):
    """lorahip_demod_receive with async = 2: step k's kernel is launched before step k-1's summary is read; the packets of a step
    arrive with the next call (the last ones with receive_flush). All steps together: the reference's packets, calls and read
    positions -- whatever the chunking, including chunks so small that a step posts nothing. While a step is in flight everything
    else is refused; after the flush the object is an ordinary one again."""
    import lora_sdr_amd as L
    rng = np.random.default_rng(800 + sf)
    N, B = 1 << sf, 13
    host = _streams(oracle, rng, sf, B, n_frames=4)
    cap = host.shape[1]
    refs = [oracle.demod_run(sf, host[c], mtu=9) for c in range(B)]
    iq = gpu.from_numpy(host).cuda()
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(9)
    rows = [d.receiver_rows(cap_packets=64, stride=16) for _ in range(2)]
    # the block's signals (error / power / snr at DOWNCHIRP1, LoRaDemod.cpp:267-269) travel with the packets, one step late like them:
    # into pinned host rows the device writes directly (what a host consumer -- a block's emitSignal -- reads)
    d.set_signals(True)
    sig_rows = d.receiver_signal_rows(64 + B, pinned_host=True)
    got, calls, w, k = [[] for _ in range(B)], 0, 0, 0
    got_sig = [[] for _ in range(B)]

    def take(n, r):
        gpu.cuda.synchronize()
        sy, ns, chn = r[0][:n].cpu().numpy(), r[1][:n].cpu().numpy(), r[2][:n].cpu().numpy()
        for i in range(n):
            got[int(chn[i])].append(sy[i, :ns[i]].copy())
        for i in range(d.last_signals()):
            got_sig[int(sig_rows[0][i])].append((int(sig_rows[1][i]), float(sig_rows[2][i]), float(sig_rows[3][i])))
    piped = 0
    while w < cap:
        w = min(cap, w + int(rng.integers(N // 2, 6 * N)))
        n, c_ = d.receive(iq, w, rows[k & 1], async_=2)
        take(n, rows[k & 1])
        calls += c_
        k += 1
        if k == 3:
            # a step is in flight: runs and accessors say so instead of racing it
            for fn in (lambda: d.work(iq), lambda: d.packets(), lambda: d.activate(), lambda: d.set_trace(True), lambda: d.receive(iq, w, rows[0], async_=True)):
                with pytest.raises(L.LoraHipError):
                    fn()
            # (the read positions may be asked for: the copy is ordered behind the step in flight)
            pos = d.consumed_all()
            assert ((w - pos < 2 * N) & (pos >= 0)).all()
            piped += 1
    n, c_ = d.receive_flush(rows[k & 1])
    take(n, rows[k & 1])
    calls += c_
    assert piped == 1 and k > 8
    for c in range(B):
        r = refs[c]
        assert len(got[c]) == len(r["packets"]) >= 4, "channel %d" % c
        assert all(np.array_equal(a, b) for a, (_, b) in zip(got[c], r["packets"])), "channel %d" % c
        assert d.consumed(c) == int(sum(q["consumed"] for q in r["calls"]))
        # every emission of the reference block, in order: error exactly, power / snr within the level-3 tolerance
        assert len(got_sig[c]) == len(r["signals"]) >= 4, "channel %d" % c
        assert [g[0] for g in got_sig[c]] == [int(q[0]) for q in r["signals"]]
        assert np.allclose([g[1:] for g in got_sig[c]], [q[1:] for q in r["signals"]], rtol=0, atol=2e-5)
    assert calls == sum(len(r["calls"]) for r in refs) == d.work_calls()
    assert d.receive_flush() == (0, 0)                                      # nothing in flight: a no-op
    assert d.last_signals() == 0
    # the flushed object is an ordinary one: a rewound one-shot run gives the same again
    d.rewind(); d.activate()
    refs2 = None
    d.work_append(iq, cap)
    pk = d.packets()
    assert sum(len(r["packets"]) for r in refs) == len(pk)
    d.close()


@pytest.mark.parametrize("sf,B", [(7, 24), (10, 6), (12, 4)])
def test_resident_rows_too_small_drop_the_excess_loudly(gpu, oracle, sf, B):
    """The resident receiver writes a step's packets while the step runs: rows that cannot hold them lose the excess -- counted, and the
    call that reports the step fails with LORAHIP_E_INVALID and the number of rows the step needed (include/lorahip.h). What DID fit is
    whole packets of the reference, every later step is unaffected, the counts add up to the reference's, and the flushed object is an
    ordinary one."""
    import lora_sdr_amd as L
    rng = np.random.default_rng(1500 + sf)
    N = 1 << sf
    host = _streams(oracle, rng, sf, B, n_frames=3)
    host = np.pad(host, ((0, 0), (0, -host.shape[1] % 16)))
    cap = host.shape[1]
    refs = [oracle.demod_run(sf, host[c], mtu=9) for c in range(B)]
    total = sum(len(r["packets"]) for r in refs)
    iq = gpu.from_numpy(host).cuda()
    gpu.cuda.synchronize()
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(9)
    small = 2
    rows = [d.receiver_rows(cap_packets=small, stride=16) for _ in range(2)]
    reported = failed = kept = 0
    w = k = 0
    pending = None
    step = 40 * N                                                      # long steps: several channels finish a packet in each

    def check_rows(r, n):
        nonlocal kept
        sy, ns, chn = r[0][:n].cpu().numpy(), r[1][:n].cpu().numpy(), r[2][:n].cpu().numpy()
        for i in range(n):
            assert any(np.array_equal(sy[i, :ns[i]], q) for _, q in refs[int(chn[i])]["packets"]), "row %d is no packet of channel %d" % (i, int(chn[i]))
            kept += 1
    while w < cap:
        w = min(cap, w + (step if k else 2 * N))                        # (the first call is an ordinary step: too short for a packet)
        try:
            n, _ = d.receive(iq, w, rows[k & 1], async_=3)
        except L.LoraHipError as e:
            n = e.n_packets
            failed += 1
            assert n > small and d.resident_active()                    # the step needed more rows than it had; the kernel stays
        if k == 0:
            assert n == 0
        else:
            assert d.resident_active()
            if pending is not None: check_rows(rows[pending], min(n, small))
            pending = k & 1
        reported += n
        k += 1
    try:
        n, _ = d.receive_flush(rows[k & 1])
    except L.LoraHipError as e:
        n = e.n_packets
        failed += 1
    st = d.last_steps()
    if pending is not None and st: check_rows(rows[pending], min(st[0][0], small))
    reported += n
    assert not d.resident_active()
    assert failed >= 1 and reported == total and kept >= failed * small
    d.rewind(); d.activate()
    d.work_append(iq, cap)
    assert len(d.packets()) == total
    d.close()


@pytest.mark.parametrize("sf,B,depth", [(7, 40, 1), (8, 13, 1), (9, 21, 2), (10, 9, 3), (7, 70, 3), (11, 7, 1), (12, 5, 2), (11, 3, 3)])
def test_resident_receiver_equals_the_reference(gpu, oracle, sf, B, depth):
    """lorahip_demod_receive with async = 3: ONE kernel launch stays on the device, the steps arrive as messages, the kernel packs every
    step's packets and signals into the rows that came with the step's call; a step is reported `depth` calls later (default 1: the
    caller cycles depth + 1 sets of rows), the flush reports what is left and ends the kernel. All steps together, whatever the chunking
    (chunks so small that a step posts nothing included): the reference's packets, signals, call counts and read positions. While the
    kernel is resident everything else is refused; afterwards the object is an ordinary one."""
    import lora_sdr_amd as L
    rng = np.random.default_rng(1300 + sf + depth)
    N = 1 << sf
    host = _streams(oracle, rng, sf, B, n_frames=4)
    host = np.pad(host, ((0, 0), (0, -host.shape[1] % 16)))       # rows of whole 128-byte lines: what the resident mode asks for
    cap = host.shape[1]
    refs = [oracle.demod_run(sf, host[c], mtu=9) for c in range(B)]
    iq = gpu.from_numpy(host).cuda()
    gpu.cuda.synchronize()
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(9)
    d.set_signals(True)
    NS = depth + 1
    rows = [d.receiver_rows(cap_packets=4 * B, stride=16) for _ in range(NS)]
    # as many sets of signal rows, cycling with the packet rows (a step's signals go where its packets go: with ITS call's registration)
    sigs = [d.receiver_signal_rows(8 * B, pinned_host=True) for _ in range(NS)]
    got, got_sig, calls, w, k = [[] for _ in range(B)], [[] for _ in range(B)], 0, 0, 0
    pending = []                                                 # row sets of the resident steps rung and not yet reported, oldest first

    def take(n, r, sig, nsig):
        sy, ns, chn = r[0][:n].cpu().numpy(), r[1][:n].cpu().numpy(), r[2][:n].cpu().numpy()
        for i in range(n):
            got[int(chn[i])].append(sy[i, :ns[i]].copy())
        for i in range(nsig):
            got_sig[int(sig[0][i])].append((int(sig[1][i]), float(sig[2][i]), float(sig[3][i])))
    resident = 0
    while w < cap:
        w = min(cap, w + int(rng.integers(N // 2, 6 * N)))
        j = k % NS
        d.register_signal_rows(sigs[j])
        n, c_ = d.receive(iq, w, rows[j], async_=3, depth=depth)
        calls += c_
        if d.resident_active():
            pending.append(j)                                    # this call rang a step into rows[j]
            steps = d.last_steps()
            assert sum(p for p, _ in steps) == n and len(steps) <= 1
            for pk_, sg_ in steps:                               # ... and reported the one rung `depth` calls ago
                jj = pending.pop(0)
                take(pk_, rows[jj], sigs[jj], sg_)
        else:
            take(n, rows[j], sigs[j], d.last_signals())         # an ordinary step (the first): its own rows, at once
        k += 1
        if k == 4:
            for fn in (lambda: d.work(iq), lambda: d.packets(), lambda: d.activate(), lambda: d.receive(iq, w, rows[0], async_=True),
                       lambda: d.receive(iq, w, rows[0], async_=2), lambda: d.consumed_all()):
                with pytest.raises(L.LoraHipError):
                    fn()
            assert d.resident_active() and len(pending) == min(depth, 3)
            resident += 1
    n, c_ = d.receive_flush(rows[k % NS])
    calls += c_
    steps = d.last_steps()
    assert len(steps) == len(pending) == depth and not d.resident_active()
    for pk_, sg_ in steps:
        jj = pending.pop(0)
        take(pk_, rows[jj], sigs[jj], sg_)
    take(n - sum(p for p, _ in steps), rows[k % NS], sigs[k % NS], 0)      # (what an ordinary step at the flush added, if any)
    assert resident == 1 and k > 8
    for c in range(B):
        r = refs[c]
        # a channel's packets arrive in time order (rows of a step are not sorted by channel, the steps are in order)
        assert len(got[c]) == len(r["packets"]) >= 4, "channel %d" % c
        assert all(np.array_equal(a, b) for a, (_, b) in zip(got[c], r["packets"])), "channel %d" % c
        assert d.consumed(c) == int(sum(q["consumed"] for q in r["calls"]))
        assert [g[0] for g in got_sig[c]] == [int(q[0]) for q in r["signals"]] and len(got_sig[c]) >= 4, "channel %d" % c
        assert np.allclose([g[1:] for g in got_sig[c]], [q[1:] for q in r["signals"]], rtol=0, atol=2e-5, equal_nan=True)
    assert calls == sum(len(r["calls"]) for r in refs) == d.work_calls()
    assert d.receive_flush() == (0, 0)
    # the flushed object is an ordinary one: a rewound one-shot run gives the same again
    d.set_signals(False); d.receiver_signal_rows(0)
    d.rewind(); d.activate()
    d.work_append(iq, cap)
    assert sum(len(r["packets"]) for r in refs) == len(d.packets())
    d.close()


@pytest.mark.parametrize("async_", [False, True, 2])
def test_receiver_steps_deliver_the_signals_and_lose_none_on_small_rows(gpu, oracle, async_):
    """Signals in a running receiver, every kind of step (waiting, stream-ordered, pipelined): device rows this time. Signal rows that
    cannot hold what is due are an error like packet rows that are too small -- nothing is lost: registering larger rows and calling
    again delivers everything."""
    import lora_sdr_amd as L
    sf, B = 8, 9
    rng = np.random.default_rng(4242)
    N = 1 << sf
    host = _streams(oracle, rng, sf, B, n_frames=3)
    cap = host.shape[1]
    refs = [oracle.demod_run(sf, host[c], mtu=9) for c in range(B)]
    iq = gpu.from_numpy(host).cuda()
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(9)
    d.set_signals(True)
    rows = d.receiver_rows(cap_packets=64, stride=16)
    sig = d.receiver_signal_rows(2)                                         # far too small: the first step with > 2 emissions must refuse
    got, got_sig, refused = [[] for _ in range(B)], [[] for _ in range(B)], 0

    def take(n):
        gpu.cuda.synchronize()
        sy, ns, chn = rows[0][:n].cpu().numpy(), rows[1][:n].cpu().numpy(), rows[2][:n].cpu().numpy()
        for i in range(n):
            got[int(chn[i])].append(sy[i, :ns[i]].copy())
        m = d.last_signals()
        sc, se, sp, ss = (t[:m].cpu().numpy() for t in sig)
        for i in range(m):
            got_sig[int(sc[i])].append((int(se[i]), float(sp[i]), float(ss[i])))
    w = 0
    while w < cap:
        w = min(cap, w + 5 * N)
        try:
            n, _ = d.receive(iq, w, rows, async_=async_)
        except L.LoraHipError:
            refused += 1
            assert d.last_signals() > 2 or async_ != 2                      # (pipelined: what is due is reported)
            sig = d.receiver_signal_rows(64 + B)
            if async_ == 2:
                n, _ = d.receive(iq, w, rows, async_=2)                     # the held step first, then this one's kernel
            else:
                # an ordinary step's packets and signals stayed queued: the accessors still have them
                pk = d.packets(clear=False)
                ch_, _, er_, po_, sn_ = d.signals()
                for c_, _, sy_ in pk:
                    got[c_].append(np.asarray(sy_))
                for i in range(len(ch_)):
                    got_sig[int(ch_[i])].append((int(er_[i]), float(po_[i]), float(sn_[i])))
                d.clear_packets()
                continue
        take(n)
    if async_ == 2:
        n, _ = d.receive_flush(rows)
        take(n)
    assert refused == 1
    for c in range(B):
        r = refs[c]
        assert len(got[c]) == len(r["packets"]) >= 3 and all(np.array_equal(a, b) for a, (_, b) in zip(got[c], r["packets"])), c
        assert [g[0] for g in got_sig[c]] == [int(q[0]) for q in r["signals"]] and len(got_sig[c]) >= 3, c
        assert np.allclose([g[1:] for g in got_sig[c]], [q[1:] for q in r["signals"]], rtol=0, atol=2e-5)
    d.close()


def test_configs3_over_eight_device_entries_equals_six_single_sf_objects(gpu):
    """BASELINE configs[3] through ONE handle with the device list of an 8-GPU node -- here 0,0,0,0,0,0,0,0: lorahip_shard_plan's split,
    48 (device, SF) parts, 48 host threads and streams are the real ones, only the GPU is the one a test box has. 16384 channels,
    SF = 7 + c mod 6, every channel one frame of 16 symbols in one device buffer (lorahip_demod_run_device_segments). Every channel's
    packets, call count and read position equal what six single-SF objects make of the same samples. A CODE-PATH check of the
    multi-device object (DeviceGuard, per-part threads, per-device kernel attributes), not a scaling measurement."""
    import lora_sdr_amd as L
    from lora_sdr_amd import workloads as WL
    torch = gpu
    n_channels, nsyms = 16384, 16
    sfs = WL.mixed_sf_channels(n_channels)
    parts, first, cnt, at, per_sf = [], np.zeros(n_channels, np.int64), np.zeros(n_channels, np.uint64), 0, {}
    for sf in range(7, 13):
        local = np.nonzero(sfs == sf)[0]
        ctx = L.Context(sf)
        iq, _ = WL.frame_streams(ctx, local.size, 1, nsyms, sigma=0.05, seed=3 + sf)
        ctx.close()
        n = int(iq.shape[1])
        first[local] = at + np.arange(local.size, dtype=np.int64) * n
        cnt[local] = n
        at += local.size * n
        parts.append(iq.reshape(-1))
        per_sf[sf] = (local, iq)
    buf = torch.cat(parts)
    del parts
    m = L.LoRaDemod(channel_sf=sfs, devices=[0] * 8)
    assert len(m.parts) == 48 and sum(p[2] for p in m.parts) == n_channels
    m.setMTU(nsyms)
    m.work_segments_multi([buf] * 8, first, cnt)           # one buffer per entry of the device list (here the same one eight times)
    ch_, _rd, ln_, sy_ = m.packets_arrays()
    starts = np.concatenate([[0], np.cumsum(ln_)])[:-1]
    mine = {}
    for i in np.argsort(ch_, kind="stable").tolist():
        mine.setdefault(int(ch_[i]), []).append(sy_[starts[i]:starts[i] + ln_[i]])
    consumed_m = m.consumed_all()
    total_calls = 0
    for sf, (local, iq) in per_sf.items():
        d = L.LoRaDemod(sf, n_channels=local.size); d.set_mode(1); d.setMTU(nsyms)
        d.work(iq)
        total_calls += d.work_calls()
        c1, _r1, l1, s1 = d.packets_arrays()
        st1 = np.concatenate([[0], np.cumsum(l1)])[:-1]
        want = {}
        for i in np.argsort(c1, kind="stable").tolist():
            want.setdefault(int(local[c1[i]]), []).append(s1[st1[i]:st1[i] + l1[i]])
        assert len(want) >= local.size - 2                                        # (practically every channel posts its packet)
        for g in local.tolist():
            a, b = mine.get(g, []), want.get(g, [])
            assert len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b)), "channel %d (SF%d)" % (g, sf)
        assert np.array_equal(consumed_m[local], d.consumed_all())
        d.close()
    assert m.work_calls() == total_calls
    m.close()


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_mixed_object_equals_single_sf_objects(gpu, oracle, devices):
    """lorahip_demod_create_mixed: 19 channels with SF = 7 + c mod 6 behind ONE handle, over one / two / three "devices" (all device
    0: the split, the parts' threads and streams are real, the GPU is the one a test box has). Packets, signals, consumption, call
    counts and per-call traces with GLOBAL channel numbers equal six single-SF objects' bit for bit, and the reference's."""
    import lora_sdr_amd as L
    rng = np.random.default_rng(900 + len(devices))
    B = 19
    sfs = 7 + np.arange(B) % 6
    streams, refs = [], []
    for c in range(B):
        sf = int(sfs[c])
        N = 1 << sf
        st, _ = frames(oracle, rng, sf, 2, 6 + c % 3, off=rng.uniform(-0.4, 0.4), noise=0.05, lead=int(rng.integers(0, 2 * N)))
        streams.append(st)
        refs.append(oracle.demod_run(sf, st, mtu=7))
    m = L.LoRaDemod(channel_sf=sfs, devices=devices)
    assert m.n_channels == B and len(m.parts) == 6 * len(devices) and sorted(set(p[1] for p in m.parts)) == [7, 8, 9, 10, 11, 12]
    assert sum(p[2] for p in m.parts) == B
    m.setMTU(7); m.set_signals(True); m.set_trace(True)
    m.work(streams)                                                         # host buffers, one per channel
    pk = m.packets(clear=False)
    sg = m.signals()
    singles = {}
    for sf in range(7, 13):
        idx = np.nonzero(sfs == sf)[0]
        s1 = L.LoRaDemod(sf, n_channels=idx.size); s1.setMTU(7); s1.set_signals(True); s1.set_trace(True)
        s1.work([streams[c] for c in idx])
        singles[sf] = (idx, s1)
    for sf, (idx, s1) in singles.items():
        pk1 = s1.packets(clear=False)
        sg1 = s1.signals()
        for j, c in enumerate(idx):
            assert [q.tolist() for cc, _, q in pk if cc == c] == [q.tolist() for cc, _, q in pk1 if cc == j] == [q.tolist() for _, q in refs[c]["packets"]]
            assert len(refs[c]["packets"]) == 2
            a, b = sg[0] == c, sg1[0] == j
            for f in range(1, 5):
                assert np.array_equal(sg[f][a], sg1[f][b])
            assert m.consumed(int(c)) == s1.consumed(j) == int(sum(k["consumed"] for k in refs[c]["calls"]))
            assert np.array_equal(m.trace_array(int(c)), s1.trace_array(j))
            assert m.labels(int(c)) == s1.labels(j)
    assert m.consumed_all().tolist() == [m.consumed(c) for c in range(B)]
    assert m.work_calls() == sum(s1.work_calls() for _, s1 in singles.values()) == sum(len(r["calls"]) for r in refs)
    assert m.near_threshold() == tuple(sum(s1.near_threshold()[i] for _, s1 in singles.values()) for i in range(2))
    m.clear_packets()
    assert m.packets() == [] and m.signals()[0].size == 0

    # the same channels as segments of device buffers: one buffer per device slot, every channel where its slot says
    slot = m.device_slot_of()
    bufs, first, cnt = [], np.zeros(B, np.int64), np.zeros(B, np.uint64)
    for s in range(len(devices)):
        mine = np.nonzero(slot == s)[0]
        at, parts = 5, [np.zeros(5, np.complex64)]
        for c in mine:
            first[c], cnt[c] = at, streams[c].size
            parts.append(streams[c]); at += streams[c].size
        bufs.append(gpu.from_numpy(np.concatenate(parts)).cuda())
    # (a fresh object: a re-activated one starts from the _finefreqError / _prevValue the last stream left, like the reference block)
    m.close()
    m = L.LoRaDemod(channel_sf=sfs, devices=devices)
    m.setMTU(7)
    if len(devices) == 1:
        m.work_segments(bufs[0], first, cnt)
    else:
        with pytest.raises(L.LoraHipError):
            m.work_segments(bufs[0], first, cnt)                            # one buffer cannot serve several device slots
        m.work_segments_multi(bufs, first, cnt)
    pk2 = m.packets()
    for c in range(B):
        assert [q.tolist() for cc, _, q in pk2 if cc == c] == [q.tolist() for _, q in refs[c]["packets"]]
    # what a mixed object does not offer says so
    with pytest.raises(L.LoraHipError):
        m.work(gpu.zeros((B, 64), dtype=gpu.complex64, device="cuda"))      # lorahip_demod_run_device: one uniform array
    with pytest.raises(ValueError):
        m.set_ports(fft_frames=4)                                           # one SF only
    for _, s1 in singles.values():
        s1.close()
    m.close()


def test_mixed_object_rejects_bad_arguments(gpu):
    import lora_sdr_amd as L
    with pytest.raises(L.LoraHipError):
        L.LoRaDemod(channel_sf=[7, 13], devices=[0])
    with pytest.raises(L.LoraHipError):
        L.LoRaDemod(channel_sf=[7, 8], devices=[99])
    m = L.LoRaDemod(channel_sf=[9, 9, 9, 9, 9], devices=[0, 0])             # one SF over two "devices": debug ports are offered
    assert m.sf == 9 and len(m.parts) == 2
    m.close()


@pytest.mark.parametrize("sf", [7, 12])
def test_pipelined_receiver_with_rows_that_are_too_small_loses_nothing(gpu, oracle, sf):
    """Pipelined steps (async = 2) whose rows cannot hold the packets that are due: LORAHIP_E_INVALID with the rows needed, the packets
    kept in the step's record set, delivered -- with the packets of the step launched in between -- by the next call whose rows hold
    them; a flush with too few rows likewise. Every packet of the reference arrives, in order, with the reference's call count."""
    import lora_sdr_amd as L
    rng = np.random.default_rng(1200 + sf)
    N, B = 1 << sf, 11
    host = _streams(oracle, rng, sf, B, n_frames=4)
    cap = host.shape[1]
    refs = [oracle.demod_run(sf, host[c], mtu=9) for c in range(B)]
    iq = gpu.from_numpy(host).cuda()
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(9)
    small, big = d.receiver_rows(cap_packets=2, stride=16), d.receiver_rows(cap_packets=4 * B, stride=16)
    got, calls, refused = [[] for _ in range(B)], 0, 0

    def take(n, r):
        gpu.cuda.synchronize()
        sy, ns, chn = r[0][:n].cpu().numpy(), r[1][:n].cpu().numpy(), r[2][:n].cpu().numpy()
        for i in range(n):
            got[int(chn[i])].append(sy[i, :ns[i]].copy())

    def step(w):
        """one receiver step the way a caller with too few rows experiences it: refused (nothing lost), then repeated with enough"""
        nonlocal calls, refused
        try:
            n, c_ = d.receive(iq, w, small, async_=2)
            take(n, small)
        except L.LoraHipError as e:
            assert e.n_packets > 2                         # what is needed
            refused += 1
            with pytest.raises(L.LoraHipError):            # still too small: still nothing lost, nothing launched
                d.receive(iq, w, small, async_=2)
            n, c_ = d.receive(iq, w, big, async_=2)        # both steps' packets
            assert n >= e.n_packets
            take(n, big)
        calls += c_
    w = 0
    while w < cap:
        w = min(cap, w + int(rng.integers(4 * N, 12 * N)))
        step(w)
    try:
        n, c_ = d.receive_flush(small)
        take(n, small)
    except L.LoraHipError as e:
        refused += 1
        n, c_ = d.receive_flush(big)
        assert n == e.n_packets
        take(n, big)
    calls += c_
    assert refused >= 2
    for c in range(B):
        r = refs[c]
        assert len(got[c]) == len(r["packets"]) >= 4, "channel %d" % c
        assert all(np.array_equal(a, b) for a, (_, b) in zip(got[c], r["packets"])), "channel %d" % c
        assert d.consumed(c) == int(sum(q["consumed"] for q in r["calls"]))
    assert calls == sum(len(r["calls"]) for r in refs) == d.work_calls()
    d.close()


@pytest.mark.parametrize("sf", [7, 11])
def test_pipelined_flush_resumes_a_channel_whose_records_were_full(gpu, oracle, sf):
    """The last pipelined step fills the channels' per-launch record capacity (bounded to 6 calls here): the flush resumes them until no
    channel has 2N samples left and delivers those packets too -- the object is left where a one-shot run leaves it."""
    import lora_sdr_amd as L
    rng = np.random.default_rng(1300 + sf)
    N, B = 1 << sf, 7
    host = _streams(oracle, rng, sf, B, n_frames=3)
    cap = host.shape[1]
    refs = [oracle.demod_run(sf, host[c], mtu=9) for c in range(B)]
    iq = gpu.from_numpy(host).cuda()
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(9)
    d.set_record_capacity(6)
    rows = d.receiver_rows(cap_packets=8 * B, stride=16)
    got, calls = [[] for _ in range(B)], 0

    def take(n):
        gpu.cuda.synchronize()
        sy, ns, chn = rows[0][:n].cpu().numpy(), rows[1][:n].cpu().numpy(), rows[2][:n].cpu().numpy()
        for i in range(n):
            got[int(chn[i])].append(sy[i, :ns[i]].copy())
    for w in (4 * N, 8 * N, cap):                           # the last step covers most of the capture: 6 calls per channel are not enough
        n, c_ = d.receive(iq, w, rows, async_=2)
        take(n); calls += c_
    n, c_ = d.receive_flush(rows)
    take(n); calls += c_
    for c in range(B):
        r = refs[c]
        assert len(got[c]) == len(r["packets"]) >= 3, "channel %d" % c
        assert all(np.array_equal(a, b) for a, (_, b) in zip(got[c], r["packets"])), "channel %d" % c
        assert d.consumed(c) == int(sum(q["consumed"] for q in r["calls"]))
    assert calls == sum(len(r["calls"]) for r in refs) == d.work_calls()
    d.close()


@pytest.mark.parametrize("async_", [True, 2])
@pytest.mark.parametrize("sf", [11, 12])
def test_wide_kernels_carry_long_open_packets_across_chunks(gpu, oracle, sf, async_):
    """SF11 / SF12: a channel is a workgroup of 2 / 4 wavefronts, and a packet that is open when a launch ends is saved to the carry
    rows by ALL of them while its last symbol was stored by the first -- packets of 100 symbols cut by chunks of a few windows, so
    that most launches end inside a packet of more than 64 symbols (the barrier before the save, lorahip_framemachine.h::carryOut)."""
    import lora_sdr_amd as L
    rng = np.random.default_rng(1400 + sf)
    N, B, nsyms = 1 << sf, 5, 100
    streams = [frames(oracle, rng, sf, 2, nsyms, off=rng.uniform(-0.4, 0.4), noise=0.05, lead=int(rng.integers(0, 2 * N)))[0] for c in range(B)]
    cap = max(s.size for s in streams)
    host = np.zeros((B, cap), np.complex64)
    for c, s in enumerate(streams):
        host[c, :s.size] = s
    refs = [oracle.demod_run(sf, host[c], mtu=nsyms) for c in range(B)]
    iq = gpu.from_numpy(host).cuda()
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(nsyms)
    rows = d.receiver_rows(cap_packets=4 * B, stride=128)
    got, w = [[] for _ in range(B)], 0

    def take(n):
        gpu.cuda.synchronize()
        sy, ns, chn = rows[0][:n].cpu().numpy(), rows[1][:n].cpu().numpy(), rows[2][:n].cpu().numpy()
        for i in range(n):
            got[int(chn[i])].append(sy[i, :ns[i]].copy())
    while w < cap:
        w = min(cap, w + int(rng.integers(2 * N, 5 * N)))
        take(d.receive(iq, w, rows, async_=async_)[0])
    if async_ == 2:
        take(d.receive_flush(rows)[0])
    for c in range(B):
        r = refs[c]
        assert len(got[c]) == len(r["packets"]) == 2, "channel %d" % c
        assert all(np.array_equal(a, b) for a, (_, b) in zip(got[c], r["packets"])), "channel %d" % c
    d.close()


def test_pipelined_rows_are_ordered_behind_a_consumer_on_the_launch_stream(gpu, oracle):
    """ONE set of rows, short steps (their packets are packed on the side stream beside the running kernel): torch's stream is made
    to wait for each call's packing (lorahip_demod_stream_wait) and the copy queued on it -- the 'decoder' -- sees every step's
    rows before the next step overwrites them; no host synchronise between the steps."""
    import lora_sdr_amd as L
    sf = 7
    rng = np.random.default_rng(1500)
    N, B = 1 << sf, 64
    host = _streams(oracle, rng, sf, B, n_frames=6)
    cap = host.shape[1]
    refs = [oracle.demod_run(sf, host[c], mtu=9) for c in range(B)]
    iq = gpu.from_numpy(host).cuda()
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(9)
    rows = d.receiver_rows(cap_packets=4 * B, stride=16)
    kept, w = [], 0
    while w < cap:
        w = min(cap, w + 8 * N)
        n, _ = d.receive(iq, w, rows, async_=2)
        if n:
            kept.append((n, rows[0][:n].clone(), rows[1][:n].clone(), rows[2][:n].clone()))     # on torch's stream, not waited for
    n, _ = d.receive_flush(rows)
    kept.append((n, rows[0][:n].clone(), rows[1][:n].clone(), rows[2][:n].clone()))
    gpu.cuda.synchronize()
    got = [[] for _ in range(B)]
    for n, sy, ns, chn in kept:
        sy, ns, chn = sy.cpu().numpy(), ns.cpu().numpy(), chn.cpu().numpy()
        for i in range(n):
            got[int(chn[i])].append(sy[i, :ns[i]].copy())
    for c in range(B):
        r = refs[c]
        assert len(got[c]) == len(r["packets"]) >= 6, "channel %d" % c
        assert all(np.array_equal(a, b) for a, (_, b) in zip(got[c], r["packets"])), "channel %d" % c
    d.close()


@pytest.mark.parametrize("sf,B", [(7, 2500), (10, 1030), (7, 40000)])
def test_summary_and_row_numbers_over_several_workgroups(gpu, oracle, sf, B):
    """Many channels: the packets' rows are numbered by one workgroup per 1024 channels, and beyond 32768 channels the summary of a streaming
    launch is the sum of one workgroup's record per 4096 channels (a second launch adds them up). Channels with 0 / 1 / 2 / 3 frames in an
    order that puts different packet counts on both sides of every workgroup boundary: the call and packet totals, the device-packed rows
    (channel-major, time ascending) and the chunked receiver's per-step totals equal the reference's."""
    import lora_sdr_amd as L
    rng = np.random.default_rng(1600 + sf)
    N, K = 1 << sf, 8
    kinds = []
    for k in range(K):
        st = frames(oracle, rng, sf, k % 4, 6, off=rng.uniform(-0.4, 0.4), noise=0.05, lead=int(rng.integers(0, 2 * N)))[0] if k % 4 else \
            (0.05 * (rng.standard_normal(6 * N) + 1j * rng.standard_normal(6 * N))).astype(np.complex64)
        kinds.append(st)
    cap = max(s.size for s in kinds)
    host = np.zeros((K, cap), np.complex64)
    for k, s_ in enumerate(kinds):
        host[k, :s_.size] = s_
    refs = [oracle.demod_run(sf, host[k], mtu=6) for k in range(K)]
    kind_of = (np.arange(B) * 5 + np.arange(B) // 1024) % K
    iq = gpu.from_numpy(host).cuda()[gpu.from_numpy(kind_of).cuda()]
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(6)
    d.work(iq)
    assert d.work_calls() == sum(len(refs[k]["calls"]) for k in kind_of)
    sy, ns, ch = d.packets_device(stride=8, clear=False)
    sy, ns, ch = sy.cpu().numpy(), ns.cpu().numpy(), ch.cpu().numpy()
    want = [(c, p) for c in range(B) for _, p in refs[kind_of[c]]["packets"]]
    assert len(want) == ns.size > B // 2
    assert ch.tolist() == [c for c, _ in want]
    assert all(np.array_equal(sy[i, :ns[i]], p) for i, (_, p) in enumerate(want))
    assert d.consumed_all().tolist() == [int(sum(q["consumed"] for q in refs[k]["calls"])) for k in kind_of]
    # the running receiver: every step's summary (calls, packets) through the same kernels (a fresh block: activate() keeps the frequency
    # estimate and the previous value of the run before, LoRaDemod.cpp:139-143)
    d.close()
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(6)
    rows = d.receiver_rows(cap_packets=4 * B, stride=8)
    calls = npk = w = 0
    got = [[] for _ in range(B)]
    while w < cap:
        w = min(cap, w + 3 * N)
        n, c_ = d.receive(iq, w, rows, async_=True)
        gpu.cuda.synchronize()
        r_sy, r_ns, r_ch = rows[0][:n].cpu().numpy(), rows[1][:n].cpu().numpy(), rows[2][:n].cpu().numpy()
        assert (np.diff(r_ch) >= 0).all()                           # channel-major inside a step
        for i in range(n):
            got[int(r_ch[i])].append(r_sy[i, :r_ns[i]].copy())
        calls += c_; npk += n
    assert calls == d.work_calls() == sum(len(refs[k]["calls"]) for k in kind_of)
    assert npk == len(want)
    for c in range(0, B, 37):
        assert all(np.array_equal(a, b) for a, (_, b) in zip(got[c], refs[kind_of[c]]["packets"])) and len(got[c]) == len(refs[kind_of[c]]["packets"])
    d.close()


def test_random_mixed_objects_equal_their_reference_blocks(gpu):
    """a few seconds of tools/soak_mixed.py (fixed seeds): random SF mixes, device lists, segment placements, settings, lanes, signals --
    every channel of the mixed object equal to its own reference block (LoRaDemod.cpp:119-122: one block per channel)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_mixed.py"), "5", "424200"], capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0 and "equal the reference's" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
