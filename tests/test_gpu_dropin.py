"""The drop-in, compiled and run: oracle/_ref/libloradrop.so holds

  (a) the reference's own LoRaDemod.cpp with the two-line patch of INTEGRATION.md section 1 (LoRaDetector<float> ->
      LoRaDetectorHip<float>), built against the fake Pothos of oracle/stub and linked with liblorahip.so;
  (b) lora_sdr_amd/pothos/LoRaDemodBatch.cpp, the multi-channel Pothos block on level 3 of the C ABI.

Both are compared with the UNPATCHED reference block (oracle/_ref/libloraref.so: verbatim LoRaDemod.cpp on the reference's own
LoRaDetector / kissfft) on the same streams: consumption, labels, packets, signals, the fft / dec / raw ports. The libraries
are built by oracle/Makefile where /root/reference exists and travel to the GPU box as binaries."""
import os

import numpy as np
import pytest

from conftest import same_values

TOL_DB = 2e-5


def _need(path):
    if not os.path.exists(path):
        pytest.skip("%s not built (needs /root/reference at build time)" % os.path.basename(path))


def test_dropin_library_loads_and_fails_loudly_without_a_gpu():
    """CPU: the drop-in library resolves against liblorahip.so; without a gfx950 device the batch block's constructor throws
    (no silent CPU path) and the patched reference block cannot be made either"""
    import torch
    from oracle.oracle import DROPIN_SO, DropInBatch
    _need(DROPIN_SO)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError):
        DropInBatch(7, 2)


def base_stream(golden, oracle, sf):
    """(stream, mtu): the golden demodulator stream where the fixture holds one (SF7, SF9); for the long windows (SF11, SF12: the
    wide kernels, demodStreamWide) a two-frame stream with a fractional frequency offset and noise made here -- the comparison is
    against the reference block run live on the same samples either way"""
    if sf in (7, 9):
        g = golden("demod_stream.npz")
        return g["iq_%d" % sf], int(g["mtu_%d" % sf])
    rng = np.random.default_rng(1000 + sf)
    N, mtu = 1 << sf, 9
    parts = [np.zeros(N // 3 + 5, np.complex64)]
    for _ in range(2):
        parts.append(oracle.mod_frame(sf, rng.integers(0, N, mtu).astype(np.uint16), padding=3))
    x = np.concatenate(parts + [np.zeros(2 * N, np.complex64)])
    x = (x * np.exp(2j * np.pi * (-0.29) / N * np.arange(x.size))).astype(np.complex64)
    x += (0.05 * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))).astype(np.complex64)
    return x, mtu


def streams_for(golden, oracle, sf, rng):
    """three channels of equal length: the base stream, the same delayed, and a fresh two-frame stream with an offset"""
    base, mtu = base_stream(golden, oracle, sf)
    N = 1 << sf
    st2 = np.concatenate([np.zeros(37, np.complex64), base])
    syms = [rng.integers(0, N, mtu).astype(np.uint16) for _ in range(2)]
    parts = [np.zeros(N // 2 + 11, np.complex64)]
    for s in syms:
        parts.append(oracle.mod_frame(sf, s, padding=3))
    st3 = np.concatenate(parts)
    st3 = (st3 * np.exp(2j * np.pi * 0.37 / N * np.arange(st3.size))).astype(np.complex64)
    st3 += (0.03 * (rng.standard_normal(st3.size) + 1j * rng.standard_normal(st3.size))).astype(np.complex64)
    n = max(base.size, st2.size, st3.size) + N
    out = np.zeros((3, n), np.complex64)
    for i, s in enumerate((base, st2, st3)):
        out[i, :s.size] = s
    return out, mtu


@pytest.mark.gpu
@pytest.mark.parametrize("sf", [7, 9, 11, 12])
def test_patched_reference_block_on_the_hip_detector(gpu, golden, oracle, sf):
    """(a): /root/reference/LoRaDemod.cpp itself, its detector swapped for LoRaDetectorHip, against the unpatched block"""
    from oracle.oracle import Ref, REF_VARIANTS
    _need(REF_VARIANTS["dropin"])
    _need(REF_VARIANTS["-O2"])
    iq, mtu = base_stream(golden, oracle, sf)
    want = Ref("-O2").demod_run(sf, iq, mtu=mtu)
    got = Ref("dropin").demod_run(sf, iq, mtu=mtu)
    assert got["consumed"].tolist() == want["consumed"].tolist()
    if sf in (7, 9):
        assert got["consumed"].tolist() == golden("demod_stream.npz")["consumed_%d" % sf].tolist()
    assert len(want["packets"]) >= 2
    assert got["labels"] == want["labels"]
    assert same_values(got["fft"], want["fft"])                 # the fft port: every bin of every call, bit for bit
    assert same_values(got["dec"], want["dec"])                 # the dec port is the block's own arithmetic either way
    assert [c for c, _ in got["packets"]] == [c for c, _ in want["packets"]]
    assert all(np.array_equal(a, b) for (_, a), (_, b) in zip(got["packets"], want["packets"]))
    assert [n for n, _ in got["signals"]] == [n for n, _ in want["signals"]]
    assert np.allclose([v for _, v in got["signals"]], [v for _, v in want["signals"]], rtol=0, atol=TOL_DB)


@pytest.mark.gpu
@pytest.mark.parametrize("sf,devices", [(7, None), (9, None), (11, None), (12, None), (7, "0,0"), (11, "0, 0,0")])
def test_batch_block_posts_what_the_reference_block_posts(gpu, golden, oracle, sf, devices):
    """(b): LoRaDemodBatch with three channels against three runs of the unpatched reference block. SF11 / SF12 run the streaming
    kernel of the long windows (demodStreamWide). With setDevices the block spreads its channels over several level-3 objects, one
    per device and host thread -- here every "device" is device 0, the one GPU a test box has: what each
    channel's ports carry must not depend on the split."""
    from oracle.oracle import Ref, REF_VARIANTS, DropInBatch
    _need(REF_VARIANTS["dropin"])
    _need(REF_VARIANTS["-O2"])
    rng = np.random.default_rng(sf)
    iq, mtu = streams_for(golden, oracle, sf, rng)
    N = 1 << sf
    blk = DropInBatch(sf, 3, max_windows=16)                    # small buffers: several work() calls of the block per stream
    if devices:
        assert blk.set_string("setDevices", "0,x") == -2        # Pothos::InvalidArgumentException, like the sibling blocks' setters
        assert blk.set_string("setDevices", "0,99") == -2       # a device that does not exist: the block keeps what it had ...
        assert blk.set_string("setDevices", devices) == 0       # ... and can still be configured
    blk.set("setDebugPorts", 1)                                 # the reference's raw / dec / fft outputs are opt-in here
    blk.set("setMTU", mtu)
    chans, signals, works = blk.run(iq)
    assert works > 1
    ref = Ref("-O2")
    sig_by_channel = {}
    cur = None
    for name, v in signals:
        if name == "channel":
            cur = int(v)
        else:
            sig_by_channel.setdefault(cur, []).append((name, v))
    for c in range(3):
        want = ref.demod_run(sf, iq[c], mtu=mtu)
        got = chans[c]
        consumed = want["consumed"]
        assert got["consumed"] == int(consumed.sum())
        starts = np.concatenate([[0], np.cumsum(consumed)[:-1]])
        # raw = the samples consumed; dec = `total` dechirped samples per call; fft = N bins per call
        assert same_values(got["raw"], iq[c][:int(consumed.sum())])
        want_dec = np.concatenate([want["dec"][k, :int(consumed[k])] for k in range(len(consumed))])
        assert same_values(got["dec"], want_dec)
        assert same_values(got["fft"], want["fft"].reshape(-1))
        lab = [(int(starts[k]), l) for k, l in enumerate(want["labels"]) if l]
        assert got["raw_labels"] == lab and got["dec_labels"] == lab
        assert got["fft_labels"] == [(k * N, l) for k, l in enumerate(want["labels"]) if l]
        assert len(got["packets"]) == len(want["packets"]) >= 2
        assert all(np.array_equal(a, b) for a, (_, b) in zip(got["packets"], want["packets"]))
        ws = want["signals"]
        gs = sig_by_channel.get(c, [])
        assert [n for n, _ in gs] == [n for n, _ in ws]
        assert np.allclose([v for _, v in gs], [v for _, v in ws], rtol=0, atol=TOL_DB)
    blk.close()


@pytest.mark.gpu
def test_oracle_equals_the_reference_on_this_box(gpu, oracle, ref):
    """The GPU parity tests compare the HIP path with the plain-C restatement (oracle/lora_oracle.c); this closes the chain ON THE
    GPU BOX: the restatement against the reference compiled in place (oracle/_ref/libloraref.so travels as a binary) -- detector
    outputs and FFT bins for every SF, whole demodulator runs, the decoder -- by running tests/test_oracle_vs_ref.py's checks in
    the -m gpu set."""
    import test_oracle_vs_ref as T
    for sf in range(6, 13):
        T.test_detect_bit_exact(oracle, ref, sf)
    T.test_detector_sweep_n1024(oracle, ref)
    for sf, off in ((7, 0.3), (8, -0.4), (10, 0.25)):
        T.test_demod_block_identical(oracle, ref, sf, off)
    T.test_demod_sync_word_and_squelch(oracle, ref)
    for sf in (7, 10, 12):
        T.test_decoder_matches_verbatim_block(oracle, ref, sf)


@pytest.mark.gpu
@pytest.mark.parametrize("sf,devices", [(7, None), (10, "0,0"), (12, None)])
def test_batch_block_without_debug_ports(gpu, golden, oracle, sf, devices):
    """The block as a receiver -- its default: no raw / dec / fft outputs, no per-call trace, nothing per call across PCIe. What it
    consumes, the packets it posts and the signals it emits (from the kernels' per-packet records, lorahip_demod_set_signals) equal
    the unpatched reference block's; the stream outputs stay untouched."""
    from oracle.oracle import Ref, REF_VARIANTS, DropInBatch
    _need(REF_VARIANTS["dropin"])
    _need(REF_VARIANTS["-O2"])
    rng = np.random.default_rng(100 + sf)
    if sf == 10:
        N, mtu = 1 << sf, 11
        parts = [np.zeros(N // 3 + 5, np.complex64)]
        for _ in range(3):
            parts.append(oracle.mod_frame(sf, rng.integers(0, N, mtu).astype(np.uint16), padding=3))
        x = np.concatenate(parts + [np.zeros(2 * N, np.complex64)])
        x = (x * np.exp(2j * np.pi * 0.21 / N * np.arange(x.size))).astype(np.complex64)
        x += (0.05 * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))).astype(np.complex64)
        iq = np.stack([x, np.roll(x, 57), np.roll(x, 411)])
    else:
        iq, mtu = streams_for(golden, oracle, sf, rng)
    blk = DropInBatch(sf, 3, max_windows=16)
    if devices:
        assert blk.set_string("setDevices", devices) == 0
    blk.set("setMTU", mtu)
    chans, signals, works = blk.run(iq)
    assert works == 1                                           # everything an input holds in ONE work() of the block
    ref = Ref("-O2")
    sig_by_channel = {}
    cur = None
    for name, v in signals:
        if name == "channel":
            cur = int(v)
        else:
            sig_by_channel.setdefault(cur, []).append((name, v))
    for c in range(3):
        want = ref.demod_run(sf, iq[c], mtu=mtu)
        got = chans[c]
        assert got["consumed"] == int(want["consumed"].sum())
        assert got["raw"].size == 0 and got["dec"].size == 0 and got["fft"].size == 0
        assert got["raw_labels"] == [] and got["fft_labels"] == []
        assert len(got["packets"]) == len(want["packets"]) >= 2
        assert all(np.array_equal(a, b) for a, (_, b) in zip(got["packets"], want["packets"]))
        ws = want["signals"]
        gs = sig_by_channel.get(c, [])
        assert [n for n, _ in gs] == [n for n, _ in ws]
        assert np.allclose([v for _, v in gs], [v for _, v in ws], rtol=0, atol=TOL_DB)
    # the same samples arriving in pieces (the receiver shape the bench times): same packets, same consumption
    blk2 = DropInBatch(sf, 3, max_windows=16)
    blk2.set("setMTU", mtu)
    r = blk2.bench(iq, 5 << sf)
    assert r["works"] > 2
    assert r["packets"] == sum(len(ch["packets"]) for ch in chans)
    assert r["consumed"] == sum(ch["consumed"] for ch in chans)
    assert r["signals"] == len(signals)
    # a device list cannot change under a running receiver (the channels' state would be lost): refused, block intact
    assert blk2.set_string("setDevices", "0,0") == -2
    blk.close()
    blk2.close()


@pytest.mark.gpu
def test_batch_block_with_a_spreading_factor_per_channel(gpu, oracle):
    """setSpreadFactors: six channels SF7..12 behind ONE block (lorahip_demod_create_mixed inside), over two "devices"; every
    channel's packets, signals and consumption equal a reference block of that channel's SF"""
    from oracle.oracle import Ref, REF_VARIANTS, DropInBatch
    _need(REF_VARIANTS["dropin"])
    _need(REF_VARIANTS["-O2"])
    rng = np.random.default_rng(77)
    sfs = [7, 8, 9, 10, 11, 12]
    mtu = 6
    streams = []
    for sf in sfs:
        N = 1 << sf
        parts = [np.zeros(N // 2 + 3, np.complex64)]
        for _ in range(2):
            parts.append(oracle.mod_frame(sf, rng.integers(0, N, mtu).astype(np.uint16), padding=3))
        x = np.concatenate(parts + [np.zeros(2 * N, np.complex64)])
        x = (x * np.exp(2j * np.pi * 0.17 / N * np.arange(x.size))).astype(np.complex64)
        x += (0.04 * (rng.standard_normal(x.size) + 1j * rng.standard_normal(x.size))).astype(np.complex64)
        streams.append(x)
    n = max(s.size for s in streams)
    iq = np.zeros((6, n), np.complex64)
    for i, s in enumerate(streams):
        iq[i, :s.size] = s
    blk = DropInBatch(7, 6, max_windows=16)
    assert blk.set_string("setSpreadFactors", "7,8,x") == -2
    assert blk.set_string("setSpreadFactors", "7,8,9,10,11,12") == 0
    assert blk.set_string("setDevices", "0,0") == 0
    assert blk.set("setDebugPorts", 1, may_fail=True) == -2       # one SF only
    blk.set("setMTU", mtu)
    chans, signals, works = blk.run(iq)
    ref = Ref("-O2")
    sig_by_channel = {}
    cur = None
    for name, v in signals:
        if name == "channel":
            cur = int(v)
        else:
            sig_by_channel.setdefault(cur, []).append((name, v))
    for c, sf in enumerate(sfs):
        want = ref.demod_run(sf, iq[c], mtu=mtu)
        got = chans[c]
        assert got["consumed"] == int(want["consumed"].sum())
        assert len(got["packets"]) == len(want["packets"]) == 2
        assert all(np.array_equal(a, b) for a, (_, b) in zip(got["packets"], want["packets"]))
        gs = sig_by_channel.get(c, [])
        assert [n_ for n_, _ in gs] == [n_ for n_, _ in want["signals"]]
        assert np.allclose([v for _, v in gs], [v for _, v in want["signals"]], rtol=0, atol=TOL_DB)
    blk.close()


@pytest.mark.gpu
def test_decoder_batch_block_equals_the_verbatim_decoder_block(gpu, ref, golden):
    """lora_sdr_amd/pothos/LoRaDecoderBatch.cpp (/lora/lora_decoder_batch: the LoRaDecoder setters, B message inputs, one launch per
    work()) compiled against the fake Pothos: golden symbol packets with fresh random damage on 7 channels, per decoder configuration
    -- every message it posts, per channel and in order, and every value of its "dropped" signal equal what the verbatim
    LoRaDecoder.cpp block posts and emits for the same packets one at a time."""
    from oracle.oracle import DropInDecoder
    if not DropInDecoder.available():
        pytest.skip("oracle/_ref/libloradrop.so not built")
    RDD_TO_CR = {0: "4/4", 1: "4/5", 2: "4/6", 3: "4/7", 4: "4/8"}
    g = golden("codec_kat.npz")
    rng = np.random.default_rng(77)
    B, seen, n_msgs, n_drops = 7, set(), 0, 0
    for i in range(int(g["count"])):
        cfg = tuple(int(v) for v in g["cfg_%d" % i])
        if cfg in seen or len(seen) >= 24 or int(g["res_%d" % i][2]) != 0:
            continue
        seen.add(cfg)
        sf, ppm, rdd, crcc, inter, ec, explicit, hdr, dlen = cfg
        kw = dict(ppm=ppm, cr=RDD_TO_CR[rdd], crcc=bool(crcc), interleaving=bool(inter), error_check=bool(ec), explicit=bool(explicit), hdr=bool(hdr), data_length=dlen)
        base = g["syms_%d" % i]
        blk = DropInDecoder(B)
        assert blk.configure(sf, **kw) == 0
        assert blk.activate() == 0
        want = [[] for _ in range(B)]
        drops = 0
        for k in range(60):
            s = base.copy()
            for j in rng.integers(0, s.size, int(rng.integers(0, 4))):
                s[int(j)] = int(rng.integers(0, 1 << sf)) if rng.random() < 0.5 else (int(s[int(j)]) ^ (1 << int(rng.integers(0, sf))))
            if rng.random() < 0.15:
                s = s[:int(rng.integers(0, s.size))]                  # truncated (shorter than a header: nothing is posted)
            c = int(rng.integers(0, B))
            blk.push(c, s)
            o, d = ref.decode(sf, s, **kw)                            # a fresh verbatim block per packet: d = 1 if it called drop()
            drops += d
            if o is not None:
                want[c].append(o)
        assert blk.work() == 0
        for c in range(B):
            got = blk.outputs(c, interleaving=bool(inter))
            assert len(got) == len(want[c]), (cfg, c)
            assert all(np.array_equal(a, b) for a, b in zip(got, want[c])), (cfg, c)
            n_msgs += len(got)
        # activate() emits 0 (LoRaDecoder.cpp:192), every drop() the running count (:403-404)
        assert blk.dropped_signals() == list(range(0, drops + 1)), cfg
        n_drops += drops
        assert blk.work() == 0 and sum(len(blk.outputs(c)) for c in range(B)) == sum(len(w) for w in want)     # nothing waiting: nothing posted
        blk.close()
    assert len(seen) >= 16 and n_msgs > 500 and n_drops > 20
    # the setters' error behaviour is the reference's
    blk = DropInDecoder(2)
    assert blk.configure(7, cr="4/9") == -2                           # Pothos::InvalidArgumentException (LoRaDecoder.cpp:150)
    assert blk.configure(7, ppm=9) == 0
    blk.push(0, np.zeros(16, np.uint16))
    assert blk.work() == -2                                           # Pothos::Exception "failed check: PPM <= SF" (:201)
    blk.close()


@pytest.mark.gpu
def test_demod_block_mtu_1024_into_the_decoder_block(gpu, ref, oracle):
    """A demodulator with MTU 1024 (setMTU is an unchecked size_t, LoRaDemod.cpp:134-137) feeds the decoder block messages of 1024
    symbols -- the packet (600 symbols: 255 bytes at SF7, 4/8) and what followed it up to the MTU. /lora/lora_demod_batch -> /lora/
    lora_decoder_batch post the bytes the verbatim LoRaDemod.cpp -> LoRaDecoder.cpp post; nothing is lost silently (round 5: out_len -2,
    no message, no drop counted)."""
    from oracle.oracle import DropInBatch, DropInDecoder, REF_VARIANTS
    _need(REF_VARIANTS["dropin"])
    if not DropInDecoder.available():
        pytest.skip("oracle/_ref/libloradrop.so not built")
    sf, mtu, B = 7, 1024, 3
    N = 1 << sf
    rng = np.random.default_rng(91)
    datas = [rng.integers(0, 256, 255).astype(np.uint8) for _ in range(B)]
    streams = []
    for c in range(B):
        syms = ref.encode(sf, datas[c], cr="4/8")
        assert syms.size > 512
        tail = rng.integers(0, N, mtu - syms.size + 4).astype(np.uint16)              # the sender keeps transmitting: the demod fills its MTU
        frame = ref.mod_frame(sf, np.concatenate([syms, tail]), padding=4)
        st = np.concatenate([np.zeros(N // 2 + 9 * c, np.complex64), frame, np.zeros(3 * N, np.complex64)])
        streams.append(st + (0.02 * (rng.standard_normal(st.size) + 1j * rng.standard_normal(st.size))).astype(np.complex64))
    n = max(s_.size for s_ in streams)
    iq = np.zeros((B, n), np.complex64)
    for c, s_ in enumerate(streams):
        iq[c, :s_.size] = s_
    blk = DropInBatch(sf, B, max_windows=64)
    blk.set("setMTU", mtu)
    chans, _, _ = blk.run(iq)
    dec = DropInDecoder(B)
    assert dec.configure(sf, cr="4/8", crcc=True) == 0 and dec.activate() == 0
    want = []
    for c in range(B):
        r = ref.demod_run(sf, iq[c], mtu=mtu)
        assert len(chans[c]["packets"]) == len(r["packets"]) >= 1 and all(np.array_equal(a, b) for a, (_, b) in zip(chans[c]["packets"], r["packets"]))
        assert chans[c]["packets"][0].size == mtu
        outs = []
        for pkt in chans[c]["packets"]:
            dec.push(c, pkt.astype(np.uint16))
            o, _ = ref.decode(sf, pkt.astype(np.uint16), cr="4/8", crcc=True)
            if o is not None:
                outs.append(o)
        want.append(outs)
    assert dec.work() == 0
    for c in range(B):
        got = dec.outputs(c)
        assert len(got) == len(want[c]) >= 1 and all(np.array_equal(a, b) for a, b in zip(got, want[c]))
        assert np.array_equal(got[0], datas[c])
    blk.close(); dec.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sf", [7, 10, 12])
def test_batch_block_through_its_own_pinned_input_slabs(gpu, golden, oracle, sf):
    """The block's getInputBufferManager() (the counterpart of LoRaDemod.cpp:346-357): the driver -- playing the framework -- takes
    every input's buffers from the manager the block returned (pinned slabs, all ports' slabs in one allocation), writes each arrival
    behind the unconsumed remainder, and the block uploads a work()'s inputs as rows of that block of memory
    (lorahip_demod_run_host_rows). Consumption, packets and signals equal the unpatched reference block's."""
    from oracle.oracle import Ref, REF_VARIANTS, DropInBatch
    _need(REF_VARIANTS["dropin"])
    _need(REF_VARIANTS["-O2"])
    rng = np.random.default_rng(300 + sf)
    iq, mtu = streams_for(golden, oracle, sf, rng)
    ref = Ref("-O2")
    for max_windows in (6, 64):                                 # arrivals of 6 windows (many work() calls, remainders carried over) and of everything at once
        blk = DropInBatch(sf, 3, max_windows=max_windows)
        blk.use_input_slabs(True)
        blk.set("setMTU", mtu)
        chans, signals, works = blk.run(iq)
        assert blk.input_slabs_active() and works >= 1
        # every work() that ran took the one-DMA path: the ports walk the pool's generations in step (slabs handed out in slab order)
        assert blk.get("workRuns") >= 1 and blk.get("slabRowRuns") == blk.get("workRuns")
        sig_by_channel, cur = {}, None
        for name, v in signals:
            if name == "channel":
                cur = int(v)
            else:
                sig_by_channel.setdefault(cur, []).append((name, v))
        for c in range(3):
            want = ref.demod_run(sf, iq[c], mtu=mtu)
            got = chans[c]
            assert got["consumed"] == int(want["consumed"].sum()), (max_windows, c)
            assert len(got["packets"]) == len(want["packets"]) >= 2
            assert all(np.array_equal(a, b) for a, (_, b) in zip(got["packets"], want["packets"]))
            gs, ws = sig_by_channel.get(c, []), want["signals"]
            assert [n for n, _ in gs] == [n for n, _ in ws]
            assert np.allclose([v for _, v in gs], [v for _, v in ws], rtol=0, atol=TOL_DB)
        blk.close()


@pytest.mark.gpu
def test_batch_block_pinned_pool_policy(gpu, golden, oracle):
    """The pinned pool is bounded: a block whose slabs would not fit setPinnedInputLimit (MiB), and a mixed-SF block (one row stride
    would pin the largest SF's slab for every port), answer getInputBufferManager() like Pothos::Block does -- the framework's default
    buffers -- and run through lorahip_demod_run instead; results unchanged."""
    from oracle.oracle import Ref, REF_VARIANTS, DropInBatch
    _need(REF_VARIANTS["dropin"])
    _need(REF_VARIANTS["-O2"])
    rng = np.random.default_rng(77)
    iq, mtu = streams_for(golden, oracle, 8, rng)
    want = [Ref("-O2").demod_run(8, iq[c], mtu=mtu) for c in range(3)]
    for limit, mixed, active in ((0, False, False), (1, False, True), (8192, True, False)):
        blk = DropInBatch(8, 3, max_windows=64)
        blk.set("setPinnedInputLimit", limit)
        if mixed:
            assert blk.set_string("setSpreadFactors", "8,8,9") == 0
        blk.use_input_slabs(True)
        blk.set("setMTU", mtu)
        chans, _, works = blk.run(iq)
        assert blk.input_slabs_active() == active, (limit, mixed)
        assert (blk.get("slabRowRuns") > 0) == active
        if not mixed:
            for c in range(3):
                assert chans[c]["consumed"] == int(want[c]["consumed"].sum())
                assert all(np.array_equal(a, b) for a, (_, b) in zip(chans[c]["packets"], want[c]["packets"]))
        blk.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sf", [8, 11])
def test_run_host_rows_equals_run(gpu, oracle, sf):
    """lorahip_demod_run_host_rows: per-channel segments of the rows of one pinned host block (ragged starts and lengths, an empty
    channel), one strided copy -- the packets, call counts and read positions of lorahip_demod_run on the same samples"""
    import lora_sdr_amd as L
    from test_gpu_demod import frames
    rng = np.random.default_rng(400 + sf)
    N, B = 1 << sf, 6
    streams = [frames(oracle, rng, sf, 2, 8, off=rng.uniform(-0.4, 0.4), noise=0.05, lead=int(rng.integers(0, 2 * N)))[0] for _ in range(B)]
    streams[3] = streams[3][:0]                                 # a channel with nothing
    first = rng.integers(0, 3 * N, B).astype(np.int64)
    stride = int(max(f + s.size for f, s in zip(first, streams))) + 7
    rows = L.pinned_empty((B, stride))
    rows[...] = 0
    for c, s in enumerate(streams):
        rows[c, first[c]:first[c] + s.size] = s
    a = L.LoRaDemod(sf, n_channels=B); a.set_mode(1); a.setMTU(8)
    a.work(streams)
    b = L.LoRaDemod(sf, n_channels=B); b.set_mode(1); b.setMTU(8)
    b.work_host_rows(rows, first, [s.size for s in streams])
    pa, pb = a.packets(), b.packets()
    assert len(pa) == len(pb) >= 2 * (B - 1) and all(x[0] == y[0] and np.array_equal(x[2], y[2]) for x, y in zip(pa, pb))
    assert a.work_calls() == b.work_calls() and a.consumed_all().tolist() == b.consumed_all().tolist()
    # a mixed object over two parts takes the rows one by one (same results); bad geometry is refused
    m = L.LoRaDemod(channel_sf=[sf] * B, devices=[0, 0]); m.setMTU(8)
    m.work_host_rows(rows, first, [s.size for s in streams])
    pm = m.packets()
    assert sorted((x[0], x[2].tolist()) for x in pm) == sorted((x[0], x[2].tolist()) for x in pa)
    with pytest.raises(L.LoraHipError):
        b.work_host_rows(rows, first, [stride] * B)
    a.close(); b.close(); m.close()
