"""Generate tests/golden/*.npz from the REAL reference code (oracle/_ref/libloraref.so, built
in place from /root/reference by oracle/Makefile). Run in the build container only:

    python tests/golden/make_golden.py

The fixtures pin the oracle restatement (tests/test_oracle_golden.py, CPU) and the HIP path
(tests/test_gpu_golden.py) on boxes where /root/reference does not exist.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle, Ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    assert Ref.available(), "oracle/_ref/libloraref.so missing: run `make -C oracle` where /root/reference exists"
    ref = Ref()
    orc = Oracle()   # only for the frame generator (itself pinned against genChirp below)
    rng = np.random.default_rng(20260925)

    # 1. TestDetector.cpp:9-35 -- N=1024, every symbol, chirps from the reference genChirp
    N = 1024
    down, _ = ref.genchirp(N, 1, N, 0.0, True, 1.0, 0.0)
    wins = np.empty((N, N), np.complex64)
    for sym in range(N):
        ch, _ = ref.genchirp(N, 1, N, np.float32(2 * np.pi * sym) / N, False, 1.0, np.float32(np.pi / 4))
        wins[sym] = down * ch
    r = ref.detect_windows(N, wins)
    assert np.array_equal(r["sym"], np.arange(N)), "reference fails its own test_detector?"
    assert (r["power"] > -10.0).all()
    np.savez_compressed(os.path.join(HERE, "test_detector_n1024.npz"), sym=r["sym"], power=r["power"],
                        powerAvg=r["powerAvg"], fIndex=r["fIndex"],
                        # inputs are reproducible from genChirp; keep 4 windows as a spot check
                        win_idx=np.array([0, 1, 511, 1023]), wins=wins[[0, 1, 511, 1023]])

    # 2. detector KATs per SF: chirp+noise, pure noise, all-zero, single tone, tie (two equal bins)
    kat = {}
    for sf in range(6, 13):
        n = 1 << sf
        x = np.zeros((6, n), np.complex64)
        t = np.arange(n)
        x[0] = np.exp(2j * np.pi * (sf * 5 % n) * t / n) + 0.5 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        x[1] = rng.standard_normal(n) + 1j * rng.standard_normal(n)                # noise only
        x[2] = 0                                                                   # all zero -> index 0
        x[3] = np.exp(2j * np.pi * (n - 1) * t / n)                                # last bin: wrap of neighbours
        x[4] = np.exp(2j * np.pi * 3 * t / n) + np.exp(2j * np.pi * (n // 2 + 3) * t / n)  # near-equal maxima
        x[5] = 1e-3 * np.exp(2j * np.pi * (0.37 + n // 3) * t / n)                 # off-bin tone: fIndex != 0
        x = x.astype(np.complex64)
        r = ref.detect_windows(n, x, want_fft=True)
        kat["in_%d" % sf] = x
        for k in ("sym", "power", "powerAvg", "fIndex", "fft"):
            kat["%s_%d" % (k, sf)] = r[k]
    np.savez_compressed(os.path.join(HERE, "detector_kat.npz"), **kat)

    # 3. genChirp KATs
    gc = {}
    for i, (n, ovs, nn, f0, dn, ampl, ph) in enumerate([(128, 1, 128, 0.0, 0, 1.0, 0.0), (128, 1, 32, 0.0, 1, 0.3, 1.5),
                                                         (1024, 1, 1024, 2.5, 0, 1.0, 0.785398), (4096, 2, 4096, 0.1, 1, 0.7, 3.0)]):
        s, p = ref.genchirp(n, ovs, nn, f0, dn, ampl, ph)
        gc["args_%d" % i] = np.array([n, ovs, nn, f0, dn, ampl, ph], np.float64)
        gc["samps_%d" % i] = s
        gc["phase_%d" % i] = np.float32(p)
    np.savez_compressed(os.path.join(HERE, "genchirp_kat.npz"), **gc)

    # 4. whole LoRaDemod block on a stream: 2 frames + offset + noise, SF7 and SF9 (fine-tune active)
    dm = {}
    for sf, mtu, off in ((7, 12, 0.3), (9, 8, -0.2)):
        n = 1 << sf
        syms = rng.integers(0, n, mtu).astype(np.uint16)
        fr = orc.mod_frame(sf, syms, padding=3)
        st = np.concatenate([np.zeros(n // 2 + 5, np.complex64), fr, fr, np.zeros(3 * n, np.complex64)])
        st = (st * np.exp(2j * np.pi * off / n * np.arange(st.size))).astype(np.complex64)
        st += (0.05 * (rng.standard_normal(st.size) + 1j * rng.standard_normal(st.size))).astype(np.complex64)
        r = ref.demod_run(sf, st, mtu=mtu)
        assert len(r["packets"]) == 2 and all(np.array_equal(p, syms.astype(np.int16)) for _, p in r["packets"])
        dm["iq_%d" % sf] = st
        dm["syms_%d" % sf] = syms
        dm["mtu_%d" % sf] = np.int64(mtu)
        dm["consumed_%d" % sf] = r["consumed"]
        dm["labels_%d" % sf] = np.array(r["labels"])
        dm["fft_peak_%d" % sf] = np.abs(r["fft"]).argmax(axis=1).astype(np.int32)
        dm["fft_sum_%d" % sf] = r["fft"].sum(axis=1)           # checksum of every FFT frame
        dm["dec_first_%d" % sf] = r["dec"][:, :n][[0, 5, 12, 20]]  # a few full dechirped windows (fine-tune on)
        dm["packet_calls_%d" % sf] = np.array([c for c, _ in r["packets"]], np.int64)
        dm["packets_%d" % sf] = np.stack([p for _, p in r["packets"]])
        dm["signals_%d" % sf] = np.array([v for _, v in r["signals"]], np.float64)
    np.savez_compressed(os.path.join(HERE, "demod_stream.npz"), **dm)

    # 5. the LoRaMod block (verbatim LoRaMod.cpp through the fake framework): frames for a few parameter sets
    mf = {}
    for i, (sf, nsym, sync, ampl, pad) in enumerate([(7, 11, 0x12, 1.0, 1), (8, 6, 0x34, 0.5, 3), (10, 5, 0x8e, 1.0, 0)]):
        syms = rng.integers(0, 1 << sf, nsym).astype(np.uint16)
        mf["args_%d" % i] = np.array([sf, sync, pad], np.int64)
        mf["ampl_%d" % i] = np.float32(ampl)
        mf["syms_%d" % i] = syms
        mf["frame_%d" % i] = ref.mod_frame(sf, syms, sync=sync, ampl=ampl, padding=pad)
    np.savez_compressed(os.path.join(HERE, "mod_frame.npz"), **mf)

    # 6. the codec blocks (verbatim LoRaEncoder.cpp -> symbols, verbatim LoRaDecoder.cpp -> bytes): clean and damaged
    #    packets over coding rates, symbol sizes, header modes, crc, error checking
    ck = {}
    n = 0
    rdd_of = {"4/4": 0, "4/5": 1, "4/6": 2, "4/7": 3, "4/8": 4}
    for sf in (7, 9, 10, 12):
        for cr in ("4/4", "4/5", "4/6", "4/7", "4/8"):
            for ppm, explicit, crc in ((0, True, True), (sf - 2, True, False), (0, False, True), (sf - 2, False, False)):
                nbytes = int(rng.integers(2, 48))
                data = rng.integers(0, 256, nbytes).astype(np.uint8)
                syms = ref.encode(sf, data, ppm=ppm, cr=cr, explicit=explicit, crc=crc)
                for damage in (0, 1, 2):
                    s = syms.copy()
                    if damage == 1:
                        k = int(rng.integers(0, s.size)); s[k] = (int(s[k]) + 1) % (1 << sf)
                    elif damage == 2:
                        for k in rng.integers(0, s.size, 3): s[int(k)] = int(rng.integers(0, 1 << sf))
                    for hdr, ec in ((False, False), (True, True)):
                        out, dropped = ref.decode(sf, s, ppm=ppm, cr=cr, crcc=crc, error_check=ec, explicit=explicit, hdr=hdr,
                                                  data_length=nbytes)
                        ck["cfg_%d" % n] = np.array([sf, ppm, rdd_of[cr], int(crc), 1, int(ec), int(explicit), int(hdr), nbytes], np.int32)
                        ck["syms_%d" % n] = s
                        ck["data_%d" % n] = data
                        ck["out_%d" % n] = out if out is not None else np.zeros(0, np.uint8)
                        ck["res_%d" % n] = np.array([-1 if out is None else out.size, dropped, damage], np.int32)
                        n += 1
    ck["count"] = np.int64(n)
    np.savez_compressed(os.path.join(HERE, "codec_kat.npz"), **ck)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
