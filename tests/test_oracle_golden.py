"""CPU: the oracle restatement against the committed golden vectors (generated from the real
reference by tests/golden/make_golden.py). Runs anywhere -- no /root/reference needed."""
import numpy as np
import pytest

from conftest import bits


def test_test_detector_kat(oracle, golden):
    """TestDetector.cpp:9-35: N=1024, every symbol: index == sym, power > -10 dB"""
    g = golden("test_detector_n1024.npz")
    assert np.array_equal(g["sym"], np.arange(1024)) and (g["power"] > -10.0).all()
    N = 1024
    # rebuild all windows with the oracle's genChirp restatement
    down, _ = oracle.genchirp(N, 1, N, 0.0, True, 1.0, 0.0)
    wins = np.empty((N, N), np.complex64)
    for sym in range(N):
        ch, _ = oracle.genchirp(N, 1, N, np.float32(2 * np.pi * sym) / N, False, 1.0, np.float32(np.pi / 4))
        wins[sym] = down * ch
    assert np.array_equal(bits(wins[g["win_idx"]]), bits(g["wins"]))
    o = oracle.detect_batch(10, wins, chirp_sel=2)
    for k in ("sym", "power", "powerAvg", "fIndex"):
        assert np.array_equal(o[k], g[k]), k


@pytest.mark.parametrize("sf", range(6, 13))
def test_detector_kat(oracle, golden, sf):
    g = golden("detector_kat.npz")
    x = g["in_%d" % sf]
    o = oracle.detect_batch(sf, x, chirp_sel=2, want_fft=True)
    assert np.array_equal(o["sym"], g["sym_%d" % sf])
    assert g["sym_%d" % sf][2] == 0                       # all-zero window -> index 0
    assert g["sym_%d" % sf][3] == (1 << sf) - 1           # last bin, neighbours wrap
    assert np.array_equal(bits(o["fft"]), bits(g["fft_%d" % sf]))
    for k in ("power", "powerAvg", "fIndex"):
        assert np.array_equal(bits(o[k]), bits(g["%s_%d" % (k, sf)])), k


def test_genchirp_kat(oracle, golden):
    g = golden("genchirp_kat.npz")
    for i in range(4):
        n, ovs, nn, f0, dn, ampl, ph = g["args_%d" % i]
        s, p = oracle.genchirp(int(n), int(ovs), int(nn), np.float32(f0), int(dn), np.float32(ampl), np.float32(ph))
        assert np.array_equal(bits(s), bits(g["samps_%d" % i]))
        assert np.float32(p) == g["phase_%d" % i]


@pytest.mark.parametrize("sf", [7, 9])
def test_demod_stream(oracle, golden, sf):
    g = golden("demod_stream.npz")
    N = 1 << sf
    mtu = int(g["mtu_%d" % sf])
    r = oracle.demod_run(sf, g["iq_%d" % sf], mtu=mtu)
    assert [c["consumed"] for c in r["calls"]] == g["consumed_%d" % sf].tolist()
    assert [c["label"] for c in r["calls"]] == g["labels_%d" % sf].tolist()
    fft = np.stack(r["fft"])
    assert np.array_equal(np.abs(fft).argmax(axis=1), g["fft_peak_%d" % sf])
    assert np.array_equal(bits(fft.sum(axis=1)), bits(g["fft_sum_%d" % sf]))
    dec = np.stack(r["dec"])[:, :N][[0, 5, 12, 20]]
    assert np.array_equal(bits(dec), bits(g["dec_first_%d" % sf]))
    assert [c for c, _ in r["packets"]] == g["packet_calls_%d" % sf].tolist()
    assert np.array_equal(np.stack([p for _, p in r["packets"]]), g["packets_%d" % sf])
    assert np.array_equal(g["packets_%d" % sf][0], g["syms_%d" % sf].astype(np.int16))
    assert np.array_equal(np.array(r["signals"], np.float64).reshape(-1), g["signals_%d" % sf])


def test_mod_frame_kat(oracle, golden):
    """frames recorded from the verbatim LoRaMod.cpp block (LoRaMod.cpp:109-238)"""
    g = golden("mod_frame.npz")
    for i in range(3):
        sf, sync, pad = (int(v) for v in g["args_%d" % i])
        fr = oracle.mod_frame(sf, g["syms_%d" % i], sync=sync, ampl=float(g["ampl_%d" % i]), padding=pad)
        assert fr.size == g["frame_%d" % i].size
        assert np.array_equal(bits(fr), bits(g["frame_%d" % i]))


RDD_TO_CR = {0: "4/4", 1: "4/5", 2: "4/6", 3: "4/7", 4: "4/8"}


def test_codec_kat(oracle, golden):
    """symbol packets from the verbatim LoRaEncoder.cpp, clean and damaged, and what the verbatim LoRaDecoder.cpp posts
    for them: the restated decoder must post the same bytes / nothing / drop"""
    g = golden("codec_kat.npz")
    clean_ok = 0
    for i in range(int(g["count"])):
        sf, ppm, rdd, crcc, inter, ec, explicit, hdr, dlen = (int(v) for v in g["cfg_%d" % i])
        out, dropped = oracle.decode(sf, g["syms_%d" % i], ppm=ppm, cr=RDD_TO_CR[rdd], crcc=bool(crcc), interleaving=bool(inter),
                                     error_check=bool(ec), explicit=bool(explicit), hdr=bool(hdr), data_length=dlen)
        n, drp, damage = (int(v) for v in g["res_%d" % i])
        assert (-1 if out is None else out.size) == n and bool(dropped) == bool(drp), i
        if out is not None:
            assert np.array_equal(out, g["out_%d" % i]), i
        if damage == 0 and explicit and crcc and not hdr:
            assert np.array_equal(out, g["data_%d" % i])
            clean_ok += 1
    assert clean_ok >= 20
