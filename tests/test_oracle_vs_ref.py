"""CPU: pin the plain-C restatement (oracle/lora_oracle.c) against the REAL reference code
compiled in place (oracle/_ref). Skipped where oracle/_ref was never built."""
import numpy as np
import pytest

from conftest import bits


@pytest.mark.parametrize("sf", range(1, 13))
def test_detect_bit_exact(oracle, ref, sf):
    rng = np.random.default_rng(sf)
    N = 1 << sf
    for kind in range(3):
        x = (rng.standard_normal(N) + 1j * rng.standard_normal(N)).astype(np.complex64)
        if kind == 1:
            x += (4 * np.exp(2j * np.pi * (N // 3) * np.arange(N) / N)).astype(np.complex64)
        if kind == 2:
            x[:] = 0
        a, b = oracle.detect(x), ref.detect(x)
        assert a[0] == b[0]
        assert np.array_equal(bits(np.float32(a[1:4])), bits(np.float32(b[1:4])))
        assert np.array_equal(bits(a[4]), bits(b[4]))


def test_kissfft_plan_matches_survey(oracle):
    # SF7 [4,4,4,2], SF8 [4^4], SF12 [4^6]; radix-2 stage innermost
    assert [p for p, _ in oracle.stages(128)] == [4, 4, 4, 2]
    assert [p for p, _ in oracle.stages(256)] == [4, 4, 4, 4]
    assert [p for p, _ in oracle.stages(2048)] == [4, 4, 4, 4, 4, 2]
    assert oracle.stages(4096) == [(4, 1024), (4, 256), (4, 64), (4, 16), (4, 4), (4, 1)]


@pytest.mark.parametrize("args", [(128, 1, 128, 0.0, 0, 1.0, 0.0), (1024, 1, 1024, 0.3, 1, 0.7, 0.25),
                                  (4096, 1, 1024, 2.0, 0, 0.3, 6.0), (256, 4, 1024, 1.0, 1, 1.0, 0.1)])
def test_genchirp_bit_exact(oracle, ref, args):
    a, b = oracle.genchirp(*args), ref.genchirp(*args)
    assert np.array_equal(bits(a[0]), bits(b[0])) and a[1] == b[1]


def test_detector_sweep_n1024(oracle, ref):
    """TestDetector.cpp:9-35 through both implementations"""
    N = 1024
    down, _ = ref.genchirp(N, 1, N, 0.0, True, 1.0, 0.0)
    wins = np.empty((N, N), np.complex64)
    for sym in range(N):
        ch, _ = ref.genchirp(N, 1, N, np.float32(2 * np.pi * sym) / N, False, 1.0, np.float32(np.pi / 4))
        wins[sym] = down * ch
    r = ref.detect_windows(N, wins)
    o = oracle.detect_batch(10, wins, chirp_sel=2)
    assert np.array_equal(r["sym"], np.arange(N)) and (r["power"] > -10).all()
    for k in ("sym", "power", "powerAvg", "fIndex"):
        assert np.array_equal(o[k], r[k]), k


@pytest.mark.parametrize("sf,off", [(7, 0.3), (8, -0.4), (10, 0.25)])
def test_demod_block_identical(oracle, ref, sf, off):
    """whole LoRaDemod.cpp (verbatim, fake Pothos) vs the restated state machine on a stream with
    two frames, a fractional frequency offset (fine-tune recurrence active) and noise"""
    rng = np.random.default_rng(100 + sf)
    N = 1 << sf
    syms = rng.integers(0, N, 20).astype(np.uint16)
    fr = oracle.mod_frame(sf, syms, padding=3)
    st = np.concatenate([np.zeros(N // 2 + 5, np.complex64), fr, fr, np.zeros(3 * N, np.complex64)])
    st = (st * np.exp(2j * np.pi * off / N * np.arange(st.size))).astype(np.complex64)
    st += (0.05 * (rng.standard_normal(st.size) + 1j * rng.standard_normal(st.size))).astype(np.complex64)
    A, B = oracle.demod_run(sf, st, mtu=20), ref.demod_run(sf, st, mtu=20)
    assert [c["consumed"] for c in A["calls"]] == B["consumed"].tolist()
    assert [c["label"] for c in A["calls"]] == B["labels"]
    assert any(c["label"].startswith("P ") for c in A["calls"])
    for fa, fb in zip(A["fft"], B["fft"]):
        assert np.array_equal(bits(fa), bits(fb))
    for da, db in zip(A["dec"], B["dec"]):
        assert np.array_equal(bits(da[:N]), bits(db[:N]))
    assert len(A["packets"]) == len(B["packets"]) == 2
    for (ca, pa), (cb, pb) in zip(A["packets"], B["packets"]):
        assert ca == cb and np.array_equal(pa, pb) and np.array_equal(pa, syms.astype(np.int16))
    sig = [v for _, v in B["signals"]]
    assert np.allclose(np.array(A["signals"]).reshape(-1), sig, rtol=0, atol=0)


def test_demod_sync_word_and_squelch(oracle, ref):
    """non-default sync word, small MTU, frame ending by squelch (padding) -- still identical"""
    rng = np.random.default_rng(7)
    sf, N = 8, 256
    syms = rng.integers(0, N, 9).astype(np.uint16)
    fr = oracle.mod_frame(sf, syms, sync=0x34, padding=4)
    st = np.concatenate([fr, fr, np.zeros(2 * N, np.complex64)])
    st += (0.01 * (rng.standard_normal(st.size) + 1j * rng.standard_normal(st.size))).astype(np.complex64)
    for sync, mtu in ((0x34, 64), (0x34, 4), (0x12, 64)):
        A = oracle.demod_run(sf, st, sync=sync, mtu=mtu, thresh=10.0)
        B = ref.demod_run(sf, st, sync=sync, mtu=mtu, thresh=10.0)
        assert [c["consumed"] for c in A["calls"]] == B["consumed"].tolist()
        assert len(A["packets"]) == len(B["packets"])
        for (ca, pa), (cb, pb) in zip(A["packets"], B["packets"]):
            assert ca == cb and np.array_equal(pa, pb)


@pytest.mark.parametrize("sf,padding,sync,ampl", [(7, 1, 0x12, 1.0), (8, 3, 0x34, 0.5), (10, 2, 0x12, 1.0), (12, 1, 0x8e, 2.0), (7, 0, 0x12, 1.0)])
def test_mod_frame_bit_exact(oracle, ref, sf, padding, sync, ampl):
    """the restated frame layout against the verbatim LoRaMod.cpp block: 10 up-chirps, two sync chirps, 2 1/4
    down-chirps, data chirps, padding (LoRaMod.cpp:135-229), one running float phase accumulator"""
    rng = np.random.default_rng(sf * 10 + padding)
    syms = rng.integers(0, 1 << sf, 9).astype(np.uint16)
    a = oracle.mod_frame(sf, syms, sync=sync, ampl=ampl, padding=padding)
    b = ref.mod_frame(sf, syms, sync=sync, ampl=ampl, padding=padding)
    assert a.size == b.size
    assert np.array_equal(bits(a), bits(b))


def _same(a, b):
    return (a is None and b is None) or (a is not None and b is not None and a.shape == b.shape and np.array_equal(a, b))


@pytest.mark.parametrize("sf", [7, 8, 9, 10, 11, 12])
def test_decoder_matches_verbatim_block(oracle, ref, sf):
    """the restated LoRaDecoder::work() (oracle/lora_codec.c) against LoRaDecoder.cpp itself, on packets made by the
    verbatim LoRaEncoder.cpp: every coding rate, reduced symbol size, explicit / implicit header, crc on / off, header
    passthrough, error checking, clean and with symbol errors (single-bit and gross)"""
    rng = np.random.default_rng(sf)
    n_checked = n_ok = 0
    for cr in ("4/4", "4/5", "4/6", "4/7", "4/8"):
        for ppm in (0, sf - 2):
            for explicit in (True, False):
                for crc in (True, False):
                    nbytes = int(rng.integers(2, 40))
                    data = rng.integers(0, 256, nbytes).astype(np.uint8)
                    syms = ref.encode(sf, data, ppm=ppm, cr=cr, explicit=explicit, crc=crc)
                    for corrupt in (0, 1, 2, 3):
                        s = syms.copy()
                        if corrupt == 1:                         # one bit in one symbol: correctable for 4/7, 4/8
                            s[int(rng.integers(min(8, s.size - 1), s.size))] ^= np.uint16(1 << int(rng.integers(sf - (ppm or sf), sf)))
                        elif corrupt == 2:                       # the neighbouring bin (what the demod gets wrong first)
                            k = int(rng.integers(0, s.size)); s[k] = (int(s[k]) + 1) % (1 << sf)
                        elif corrupt == 3:                       # garbage in several symbols
                            for k in rng.integers(0, s.size, 4): s[int(k)] = int(rng.integers(0, 1 << sf))
                        for hdr in (False, True):
                            for ec in (False, True):
                                kw = dict(ppm=ppm, cr=cr, crcc=crc, error_check=ec, explicit=explicit, hdr=hdr, data_length=nbytes)
                                a, da = oracle.decode(sf, s, **kw)
                                b, db = ref.decode(sf, s, **kw)
                                assert _same(a, b) and bool(da) == bool(db), (sf, cr, ppm, explicit, crc, corrupt, hdr, ec)
                                n_checked += 1
                                if corrupt == 0 and a is not None and explicit and crc and not hdr:
                                    assert np.array_equal(a, data)
                                    n_ok += 1
    assert n_checked == 5 * 2 * 2 * 2 * 4 * 4 and n_ok >= 20
    # short packets, truncated packets, interleaving off
    syms = ref.encode(sf, np.arange(12, dtype=np.uint8), cr="4/6")
    for n in (0, 5, 7, 8, 9, 13, syms.size - 1):
        assert _same(oracle.decode(sf, syms[:n], cr="4/6")[0], ref.decode(sf, syms[:n], cr="4/6")[0]), n
    a, _ = oracle.decode(sf, syms, cr="4/6", interleaving=False)
    b, _ = ref.decode(sf, syms, cr="4/6", interleaving=False)
    assert _same(a, b)


def test_code_primitives_exhaustive(oracle, ref):
    """TestCodesSx.cpp checks the reference's code primitives for every 0/1/2-bit error pattern; here every input value of
    every primitive the decoder uses, restatement (parity masks, syndrome tables) against LoRaCodes.hpp itself: result,
    error flag and uncorrectable flag"""
    ref.L.loraref_code_primitive.restype = int
    oracle.L.lo_code_primitive.restype = int
    for which, span in ((0, 256), (1, 128), (2, 32), (3, 64), (4, 4096), (5, 65536)):
        a = [oracle.L.lo_code_primitive(which, b) for b in range(span)]
        b_ = [ref.L.loraref_code_primitive(which, b) for b in range(span)]
        assert a == b_, "primitive %d" % which
    # and the property TestCodesSx.cpp pins: single-bit errors of an (8,4) codeword are corrected, flagged, not "bad"
    for d in range(16):
        cw = next(c for c in range(256) if (c & 0xf) == d and oracle.L.lo_code_primitive(0, c) == d)
        for bit in range(8):
            r = oracle.L.lo_code_primitive(0, cw ^ (1 << bit))
            assert (r & 0xf) == d and (r & 0x100) and not (r & 0x200)


@pytest.mark.parametrize("sf", [7, 9])
def test_many_streams_runner_identical(oracle, ref, sf):
    """the all-channels runner bench.py's level-3 section uses (oracle.demod_run_many / ref.demod_run_many): every stream from
    the zero start state, packets, per-call consumption and label kinds equal to the one-stream entry points and equal between the
    restatement and the verbatim LoRaDemod.cpp"""
    rng = np.random.default_rng(3 + sf)
    N, S = 1 << sf, 10
    streams = []
    for _ in range(S):
        f = oracle.mod_frame(sf, rng.integers(0, N, 20).astype(np.uint16), padding=3)
        streams.append(np.concatenate([np.zeros(int(rng.integers(0, 2 * N)), np.complex64), f, np.zeros(3 * N, np.complex64), f]))
    iq = np.zeros((S, max(len(x) for x in streams) + N), np.complex64)
    for c, x in enumerate(streams):
        iq[c, :len(x)] = x
    iq += (0.1 * (rng.standard_normal(iq.shape) + 1j * rng.standard_normal(iq.shape))).astype(np.complex64)
    a = oracle.demod_run_many(sf, iq, mtu=20, nthreads=3, calls=True)
    b = ref.demod_run_many(sf, iq, mtu=20, nthreads=2, calls=True)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert (a["n_packets"] == 2).all()
    kinds = {"": 0, "SYNC": 1, "P": 2, "D": 3, "Q": 4, "S": 5}
    for c in range(S):
        one = ref.demod_run(sf, iq[c], mtu=20)
        n = int(b["n_calls"][c])
        assert n == len(one["consumed"]) and np.array_equal(b["consumed"][c, :n], one["consumed"])
        want = [kinds["SYNC" if lab == "SYNC" else lab[:1]] for lab in one["labels"]]
        assert b["cls"][c, :n].tolist() == want
        at = 0
        for j, (call, p) in enumerate(one["packets"]):
            assert b["pkt_lens"][c, j] == len(p) and b["pkt_call"][c, j] == call
            assert np.array_equal(b["pkt_syms"][c, at:at + len(p)], p)
            at += len(p)


@pytest.mark.parametrize("cr,sigma", [("4/7", 4.0), ("4/8", 4.0 / 2 ** 0.5)])
def test_loopback_kat_on_the_cpu(oracle, ref, cr, sigma):
    """TestLoopback.cpp:66-133 at its own parameters (SF10, unit signal, noise amplitude 4.0 -- both readings of the unpinned
    /comms/noise_source scale --, padding 512, MTU 512, 5 packets of 8..128 random bytes) through the verbatim blocks (encoder, mod,
    demod, decoder) and through the restated demodulator + decoder on the same samples: same packets, same bytes, the bytes sent.
    The GPU twin is tests/test_gpu_codec.py::test_loopback_at_the_reference_parameters."""
    sf, N = 10, 1024
    rng = np.random.default_rng(7)
    sent = [rng.integers(0, 256, int(rng.integers(8, 129))).astype(np.uint8) for _ in range(5)]
    frames = [ref.mod_frame(sf, ref.encode(sf, d, cr=cr), ampl=1.0, padding=512) for d in sent]
    iq = np.concatenate(frames + [np.zeros(2 * N, np.complex64)])
    iq = (iq + sigma * (rng.standard_normal(iq.size) + 1j * rng.standard_normal(iq.size))).astype(np.complex64)
    rp = [p for _c, p in ref.demod_run(sf, iq, mtu=512)["packets"]]
    op = [p for _c, p in oracle.demod_run(sf, iq, mtu=512, keep=False)["packets"]]
    assert len(rp) == len(op) == 5 and all(np.array_equal(a, b) for a, b in zip(rp, op))
    rb = [ref.decode(sf, p.astype(np.uint16), cr=cr)[0] for p in rp]
    ob = [oracle.decode(sf, p.astype(np.uint16), cr=cr)[0] for p in op]
    assert all(np.array_equal(a, b) and np.array_equal(a, s) for a, b, s in zip(rb, ob, sent))


def _same_run(A, B, N):
    assert [c["consumed"] for c in A["calls"]] == B["consumed"].tolist()
    assert [c["label"] for c in A["calls"]] == B["labels"]
    assert len(A["packets"]) == len(B["packets"])
    for (ca, pa), (cb, pb) in zip(A["packets"], B["packets"]):
        assert ca == cb and np.array_equal(pa, pb)
    for fa, fb in zip(A["fft"], B["fft"]):
        a, b = np.ascontiguousarray(fa).view(np.float32), np.ascontiguousarray(fb).view(np.float32)
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(bits(a[~np.isnan(a)]), bits(b[~np.isnan(b)]))


@pytest.mark.parametrize("sf", [7, 9, 11])
def test_idle_receiver_on_noise(oracle, ref, sf):
    """streams without a frame (the regime of tests/test_gpu_demod.py::test_record_capacity_follows_what_a_receiver_needs and of
    tools/idle_receiver.py): noise above the default threshold is not squelched, the block makes a call per N - value samples, and at
    SF7 it posts packets that were never sent -- the restatement must make the same calls and the same false packets"""
    rng = np.random.default_rng(5 + sf)
    N = 1 << sf
    x = (rng.standard_normal(150 * N + 77) + 1j * rng.standard_normal(150 * N + 77)).astype(np.complex64)
    A, B = oracle.demod_run(sf, x, mtu=16), ref.demod_run(sf, x, mtu=16)
    _same_run(A, B, N)
    assert len(A["calls"]) > 1.5 * 150
    for thresh in (3.0, float("inf"), float("-inf"), float("nan")):
        with np.errstate(all="ignore"):
            _same_run(oracle.demod_run(sf, x[:40 * N], mtu=5, thresh=thresh), ref.demod_run(sf, x[:40 * N], mtu=5, thresh=thresh), N)


@pytest.mark.parametrize("sf", [7, 10])
def test_non_finite_samples_where_the_reference_is_defined(oracle, sf):
    """NaN / Inf samples and a burst that overflows |X|^2, placed in the down-chirp, quarter-chirp and data states (in FRAMESYNC the
    reference's `_fineTuneIndex -= NaN` indexes out of bounds: every build segfaults there, which is why no test goes there).
    What the reference does with such a window depends on HOW IT WAS COMPILED: std::complex multiplication in a plain g++ -O2 / -O3
    build goes through libgcc's __mulsc3, which "recovers" infinities from NaN + iNaN products (C99 Annex G); with -fcx-limited-range
    (result-identical for finite inputs, the fastest of the three CPU builds) it is the textbook formula.
    The restatement -- and the kernels -- are the textbook formula: identical to the -fcx-limited-range build in calls, labels, bins
    (NaN pattern included) and packets, which is what tests/test_gpu_demod.py::test_streams_with_non_finite_and_extreme_samples and
    tests/test_gpu_parity.py::test_inputs_at_the_edges_of_fp32 take the oracle's word for; the plain builds differ from both on
    such windows (asserted at the end, so that the statement stays true)."""
    from oracle.oracle import Ref
    if not (Ref.available("-O3 -fcx-limited-range") and Ref.available("-O2")):
        pytest.skip("oracle/_ref not built")
    ref_lr, ref_o2 = Ref("-O3 -fcx-limited-range"), Ref("-O2")
    rng = np.random.default_rng(900 + sf)
    N = 1 << sf
    syms = [rng.integers(0, N, 9).astype(np.uint16) for _ in range(2)]
    lead = N // 2 + 5
    clean = np.concatenate([np.zeros(lead, np.complex64)] + [oracle.mod_frame(sf, s, padding=3) for s in syms] + [np.zeros(3 * N, np.complex64)])
    clean = (clean * np.exp(2j * np.pi * 0.23 / N * np.arange(clean.size))).astype(np.complex64)
    clean += (0.05 * (rng.standard_normal(clean.size) + 1j * rng.standard_normal(clean.size))).astype(np.complex64)
    base = oracle.demod_run(sf, clean, mtu=12)
    _same_run(base, ref_o2.demod_run(sf, clean, mtu=12), N)              # finite samples: every build agrees
    starts = np.concatenate([[0], np.cumsum([c["consumed"] for c in base["calls"]])])
    states = [c["state"] for c in base["calls"]]
    hit, plain_differs = 0, 0
    for state in (2, 3, 4):
        k = states.index(state)                              # first call in the down-chirp-1 / quarter-chirp / data state
        for what in ("nan", "inf", "huge", "burst"):
            st = clean.copy()
            p = int(starts[k]) + N // 8 + 3
            if what == "nan":
                st[p] = np.nan
            elif what == "inf":
                st[p] = complex(np.inf, -np.inf)
            elif what == "huge":
                st[p] = complex(3e38, -3e38)
            else:
                st[p:p + N // 2] *= np.float32(3e19)
            with np.errstate(all="ignore"):
                A, B, P = oracle.demod_run(sf, st, mtu=12), ref_lr.demod_run(sf, st, mtu=12), ref_o2.demod_run(sf, st, mtu=12)
            _same_run(A, B, N)
            hit += sum(1 for c in A["calls"] if not np.isfinite(c["power"]))
            try:
                _same_run(A, P, N)
            except AssertionError:
                plain_differs += 1
    assert hit >= 8
    assert plain_differs >= 1
    # the detector alone, 200 windows with one non-finite sample each: the same picture
    n_lr = n_o2 = 0
    for trial in range(200):
        x = (rng.standard_normal(N) + 1j * rng.standard_normal(N)).astype(np.complex64)
        x[rng.integers(0, N)] = [complex(np.inf, 0), complex(0, -np.inf), complex(np.inf, np.inf), complex(np.nan, 1)][trial % 4]
        with np.errstate(all="ignore"):
            a, b, c = oracle.detect(x), ref_lr.detect(x), ref_o2.detect(x)
        same = lambda u, v: u[0] == v[0] and all((np.isnan(s) and np.isnan(t)) or s == t for s, t in zip(u[1:4], v[1:4]))
        n_lr += not same(a, b)
        n_o2 += not same(a, c)
    assert n_lr == 0 and n_o2 > 0


@pytest.mark.parametrize("sf", [8, 11])
def test_extreme_settings_identical(oracle, ref, sf):
    """sync words 0x00 / 0xff / 0x0f / 0xf0, thresholds +-inf / NaN / +-1e30, an MTU beyond the stream, MTU 1: the settings
    tests/test_gpu_demod.py::test_extreme_settings runs through the device, restatement against the verbatim block"""
    rng = np.random.default_rng(77 + sf)
    N = 1 << sf
    inf = float("inf")
    streams = {}
    for sync in (0x00, 0xff, 0x0f, 0xf0, 0x12):
        syms = rng.integers(0, N, 7).astype(np.uint16)
        fr = oracle.mod_frame(sf, syms, sync=sync, padding=3)
        st = np.concatenate([np.zeros(N // 2 + 9, np.complex64), fr, fr, np.zeros(2 * N, np.complex64)])
        st = (st * np.exp(2j * np.pi * 0.21 / N * np.arange(st.size))).astype(np.complex64)
        st += (0.02 * (rng.standard_normal(st.size) + 1j * rng.standard_normal(st.size))).astype(np.complex64)
        streams[sync] = st
    posted = 0
    for sync, mtu, thresh in [(0x00, 64, 3.0), (0xff, 64, 3.0), (0x0f, 5, 3.0), (0xf0, 64, 3.0), (0x12, 5, -inf), (0x12, 64, inf), (0x12, 5, float("nan")),
                              (0x12, 3, 1e30), (0x12, 6, -1e30), (0x12, 100000, 3.0), (0x00, 1, inf)]:
        with np.errstate(all="ignore"):
            A, B = oracle.demod_run(sf, streams[sync], sync=sync, mtu=mtu, thresh=thresh), ref.demod_run(sf, streams[sync], sync=sync, mtu=mtu, thresh=thresh)
        _same_run(A, B, N)
        posted += len(A["packets"])
    assert posted >= 10


def test_leading_zero_windows_change_nothing(oracle, ref):
    """The premise of tests/test_gpu_demod.py::test_streams_beyond_2_pow_31_samples: an all-zero window is not squelched (snr = -inf
    - -inf = NaN), does not sync and consumes N samples (LoRaDemod.cpp:219 with value 0), so whole windows of zeros in front of a
    stream shift the calls and change nothing else -- in the verbatim block as in the restatement."""
    sf, N = 10, 1024
    rng = np.random.default_rng(41)
    syms = [rng.integers(0, N, 7).astype(np.uint16) for _ in range(2)]
    st = np.concatenate([np.zeros(N // 3, np.complex64)] + [oracle.mod_frame(sf, s, padding=3) for s in syms] + [np.zeros(3 * N, np.complex64)])
    st = (st * np.exp(2j * np.pi * 0.1 / N * np.arange(st.size))).astype(np.complex64)
    st += (0.02 * (rng.standard_normal(st.size) + 1j * rng.standard_normal(st.size))).astype(np.complex64)
    shifted = np.concatenate([np.zeros(5 * N, np.complex64), st])
    with np.errstate(all="ignore"):
        a, b = ref.demod_run(sf, st, mtu=7), ref.demod_run(sf, shifted, mtu=7)
        _same_run(oracle.demod_run(sf, shifted, mtu=7), b, N)
    assert b["consumed"].tolist() == [N] * 5 + a["consumed"].tolist()
    assert len(a["packets"]) == len(b["packets"]) >= 2
    assert all(np.array_equal(p[1], q[1]) for p, q in zip(a["packets"], b["packets"]))
