"""GPU: the batched decoder (lorahip_decode_packets) against the golden vectors recorded from the
verbatim LoRaEncoder.cpp / LoRaDecoder.cpp blocks, against the CPU oracle on randomised damage, and at the end of the
whole receive chain: symbols -> modulator -> AWGN -> streaming demodulator -> decoder -> the bytes that were sent."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RDD_TO_CR = {0: "4/4", 1: "4/5", 2: "4/6", 3: "4/7", 4: "4/8"}


def configure(dec, sf, ppm, rdd, crcc, inter, ec, explicit, hdr, dlen):
    dec.setSpreadFactor(sf); dec.setSymbolSize(ppm); dec.setCodingRate(RDD_TO_CR[rdd]); dec.enableCrcc(crcc)
    dec.enableInterleaving(inter); dec.enableErrorCheck(ec); dec.enableExplicit(explicit); dec.enableHdr(hdr); dec.setDataLength(dlen)


@pytest.mark.parametrize("variant", [0, 1])
def test_golden_codec_kat(gpu, golden, variant):
    """every recorded case, grouped by decoder configuration so that each launch carries a batch of packets.
    variant 0: a group of lanes per packet, every stage data-parallel, working set in LDS (the default);
    variant 1: one lane walks one packet in the reference's statement order (the round-1 kernel, kept as the checker)"""
    import lora_sdr_amd as L
    g = golden("codec_kat.npz")
    groups = {}
    for i in range(int(g["count"])):
        groups.setdefault(tuple(int(v) for v in g["cfg_%d" % i]), []).append(i)
    dec = L.LoRaDecoder()
    dec._ctx.set_variant(variant)
    checked = 0
    for cfg, idx in groups.items():
        configure(dec, *cfg)
        res = dec.work([g["syms_%d" % i] for i in idx])
        for i, out in zip(idx, res):
            n, drp, _ = (int(v) for v in g["res_%d" % i])
            assert (-1 if out is None else out.size) == n, (cfg, i)
            if out is not None:
                assert np.array_equal(out, g["out_%d" % i]), (cfg, i)
            checked += 1
    assert checked == int(g["count"])
    # the block's "dropped" counter is the number of drop() calls
    assert dec.getDropped() == sum(int(g["res_%d" % i][1]) for i in range(int(g["count"])))


@pytest.mark.parametrize("sf", [7, 10, 12])
def test_random_damage_vs_oracle(gpu, oracle, golden, sf):
    """thousands of packets per launch: golden symbol packets with fresh random damage, every output compared with the
    CPU oracle's restated block (itself pinned to the verbatim LoRaDecoder.cpp)"""
    import lora_sdr_amd as L
    g = golden("codec_kat.npz")
    rng = np.random.default_rng(sf)
    dec = L.LoRaDecoder()
    seen = set()
    for i in range(int(g["count"])):
        cfg = tuple(int(v) for v in g["cfg_%d" % i])
        if cfg[0] != sf or cfg in seen or int(g["res_%d" % i][2]) != 0:
            continue
        seen.add(cfg)
        base = g["syms_%d" % i]
        packets = []
        for _ in range(300):
            s = base.copy()
            for k in rng.integers(0, s.size, int(rng.integers(0, 4))):
                s[int(k)] = int(rng.integers(0, 1 << sf)) if rng.random() < 0.5 else (int(s[int(k)]) ^ (1 << int(rng.integers(0, sf))))
            if rng.random() < 0.1:
                s = s[:int(rng.integers(0, s.size))]                  # truncated packet
            packets.append(s)
        configure(dec, *cfg)
        res = dec.work(packets)
        sf_, ppm, rdd, crcc, inter, ec, explicit, hdr, dlen = cfg
        for s, out in zip(packets, res):
            o, _ = oracle.decode(sf_, s, ppm=ppm, cr=RDD_TO_CR[rdd], crcc=bool(crcc), interleaving=bool(inter), error_check=bool(ec),
                                 explicit=bool(explicit), hdr=bool(hdr), data_length=dlen)
            assert (o is None) == (out is None)
            if o is not None:
                assert np.array_equal(o, out)
    assert len(seen) >= 8


def test_interleaving_off_passes_gray_symbols(gpu, oracle):
    import lora_sdr_amd as L
    rng = np.random.default_rng(1)
    dec = L.LoRaDecoder()
    dec.setSpreadFactor(9); dec.setSymbolSize(7); dec.setCodingRate("4/6"); dec.enableInterleaving(False)
    pk = [rng.integers(0, 512, n).astype(np.uint16) for n in (8, 13, 30, 7)]
    res = dec.work(pk)
    for s, out in zip(pk, res):
        o, _ = oracle.decode(9, s, ppm=7, cr="4/6", interleaving=False)
        assert (o is None) == (out is None) and (o is None or np.array_equal(o, out))


@pytest.mark.parametrize("sf,cr", [(7, "4/8"), (9, "4/7"), (10, "4/5")])
def test_receive_chain_bytes_in_bytes_out(gpu, golden, sf, cr):
    """TestLoopback.cpp's chain on the device: encoder output (golden, from the verbatim LoRaEncoder.cpp) -> batched
    modulator -> AWGN -> streaming demodulator -> batched decoder == the bytes fed to the encoder"""
    import lora_sdr_amd as L
    torch = gpu
    g = golden("codec_kat.npz")
    rdd = {v: k for k, v in RDD_TO_CR.items()}[cr]
    case = next(i for i in range(int(g["count"])) if tuple(int(v) for v in g["cfg_%d" % i][:4]) == (sf, 0, rdd, 1)
                and int(g["cfg_%d" % i][6]) == 1 and int(g["cfg_%d" % i][7]) == 0 and int(g["res_%d" % i][2]) == 0)
    syms, data = g["syms_%d" % case], g["data_%d" % case]
    N, B = 1 << sf, 200
    ctx = L.Context(sf)
    tx = torch.from_numpy(np.tile(syms.astype(np.int16), (B, 1))).cuda()
    iq = ctx.mod_frames(tx, padding=2, lead=N // 2 + 3)
    iq = torch.cat([iq, torch.zeros((B, 3 * N), dtype=torch.complex64, device="cuda")], dim=1).contiguous()
    ctx.add_awgn(iq, sigma=0.3, seed=5)
    d = L.LoRaDemod(sf, n_channels=B)
    d.setMTU(len(syms))
    d.work(iq)
    pk = d.packets()
    assert len(pk) == B
    dec = L.LoRaDecoder()
    dec.setSpreadFactor(sf); dec.setCodingRate(cr); dec.enableCrcc(True); dec.enableErrorCheck(True)
    out = dec.work([p[2] for p in pk])
    assert all(o is not None and np.array_equal(o, data) for o in out)
    assert dec.getDropped() == 0
    # the same hand-off without per-packet host work: the queued packets in the decoder's layout on the device
    d.activate(); d.work(iq)
    ps, pn, pc = d.packets_device(clear=False)
    assert ps.shape == (B, max(8, len(syms))) and sorted(pc.cpu().tolist()) == list(range(B))
    host = d.packets()
    assert all(np.array_equal(ps[i, :len(p[2])].cpu().numpy(), p[2]) and int(pn[i]) == len(p[2]) and int(pc[i]) == p[0] for i, p in enumerate(host))
    o2, l2, dr2 = dec.decode_batch(ps, pn)
    assert bool((l2 == len(data)).all()) and int(dr2.sum()) == 0
    assert np.array_equal(o2[:, :len(data)].cpu().numpy(), np.tile(data.astype(np.uint8), (B, 1)))
    # a row shorter than its packet is refused by the decoder, not read past
    short = ps[:, :16].contiguous()
    _, l3, _ = dec.decode_batch(short, pn)
    assert bool((l3 == -2).all())


def test_longest_packets_and_limits(gpu, oracle):
    """Packets of any length the demodulator can produce decode like the reference decodes them (its vectors follow the message,
    LoRaDecoder.cpp:210-213; setMTU is unchecked, LoRaDemod.cpp:134-137): 512 (TestLoopback.cpp's MTU), 600, 1024 and 2047 symbols at
    every coding rate, implicit lengths from a few bytes to the 4096 the tables reach, explicit headers on random symbols -- against
    the oracle; the C ABI refuses what it cannot do (rows beyond 16384 symbols, data_length beyond 4096) instead of truncating."""
    import ctypes as C
    import lora_sdr_amd as L
    from lora_sdr_amd import _lib
    rng = np.random.default_rng(3)
    dec = L.LoRaDecoder()
    lib = L.load()
    assert lib.lorahip_decode_max_symbols() == 16384 and lib.lorahip_decode_max_data_length() == 4096
    n_out = 0
    for sf, cr in ((12, "4/4"), (10, "4/5"), (8, "4/6"), (7, "4/7"), (9, "4/8")):
        for explicit, crcc, dlen in ((False, False, 100), (False, True, 700), (False, False, 4096), (True, True, 8), (False, False, 3)):
            dec.setSpreadFactor(sf); dec.setCodingRate(cr); dec.enableExplicit(explicit); dec.enableCrcc(crcc); dec.setDataLength(dlen)
            pk = [rng.integers(0, 1 << sf, n).astype(np.uint16) for n in (512, 511, 509, 505, 8, 600, 1024, 2047)]
            if dlen == 4096:
                pk.append(rng.integers(0, 1 << sf, 16384).astype(np.uint16))      # a full row: 4096 bytes fit it at every rate but sf 7..9 x 4/7, 4/8
            for s_, out in zip(pk, dec.work(pk)):
                o, _ = oracle.decode(sf, s_, cr=cr, explicit=explicit, crcc=crcc, data_length=dlen)
                assert (o is None) == (out is None), (sf, cr, explicit, dlen, s_.size)
                assert o is None or np.array_equal(o, out), (sf, cr, explicit, dlen, s_.size)
                n_out += o is not None
    assert n_out > 60
    # interleaving off: every Gray-coded symbol of a long packet comes out
    dec.setSpreadFactor(9); dec.setSymbolSize(7); dec.setCodingRate("4/6"); dec.enableInterleaving(False)
    pk = [rng.integers(0, 512, n).astype(np.uint16) for n in (600, 1500)]
    for s_, out in zip(pk, dec.work(pk)):
        o, _ = oracle.decode(9, s_, ppm=7, cr="4/6", interleaving=False)
        assert np.array_equal(o, out)
    cfg = _lib.DecoderCfg(C.sizeof(_lib.DecoderCfg), 10, 0, 4, 0, 1, 0, 1, 0, 8)
    ctx = L.Context(7)
    z = gpu.zeros(65536, dtype=gpu.int32, device="cuda")
    p = C.c_void_p(z.data_ptr())
    assert lib.lorahip_decode_packets(ctx._h, C.byref(cfg), p, 16385, p, 1, p, 2 * (16385 + 8), p, p) == -1   # row too long
    assert lib.lorahip_decode_packets(ctx._h, C.byref(cfg), p, 64, p, 1, p, 2 * 64, p, p) == -1               # output stride too short
    assert lib.lorahip_decode_packets(ctx._h, C.byref(cfg), p, 64, p, 0, p, 2 * 72, p, p) == 0                # empty batch
    cfg = _lib.DecoderCfg(C.sizeof(_lib.DecoderCfg), 10, 0, 4, 0, 1, 0, 0, 0, 4097)                           # implicit, data_length beyond the tables
    assert lib.lorahip_decode_packets(ctx._h, C.byref(cfg), p, 64, p, 1, p, 2 * 72, p, p) == -1
    assert b"data_length" in lib.lorahip_last_error()


def test_long_encoded_packets_equal_the_verbatim_decoder(gpu, ref):
    """Real packets longer than 512 symbols: 255 payload bytes through the verbatim LoRaEncoder.cpp at SF7 make some 550 (4/7) and 600 (4/8)
    symbols. Clean, with symbol errors, and padded by the demodulator's trailing noise symbols up to an MTU of 1024: the batched
    decoder's bytes and drop decisions equal the verbatim LoRaDecoder.cpp's (round 5 reported such packets as out_len = -2)."""
    import lora_sdr_amd as L
    rng = np.random.default_rng(21)
    dec = L.LoRaDecoder()
    seen_long = 0
    for sf, cr, nbytes in ((7, "4/8", 255), (7, "4/7", 255), (8, "4/8", 255), (7, "4/5", 255), (9, "4/8", 200)):
        data = rng.integers(0, 256, nbytes).astype(np.uint8)
        syms = ref.encode(sf, data, cr=cr)
        seen_long += syms.size > 512
        dec.setSpreadFactor(sf); dec.setCodingRate(cr); dec.enableExplicit(True); dec.enableCrcc(True); dec.enableErrorCheck(False)
        cases = [syms]
        for _ in range(6):
            s_ = syms.copy()
            for k in rng.integers(0, s_.size, int(rng.integers(1, 5))):
                s_[int(k)] ^= np.uint16(1 << int(rng.integers(0, sf)))
            cases.append(s_)
        cases.append(np.concatenate([syms, rng.integers(0, 1 << sf, 1024 - syms.size).astype(np.uint16)]))      # the demod's MTU-long packet
        for ec in (False, True):
            dec.enableErrorCheck(ec)
            before = dec.getDropped()
            res = dec.work(cases)
            drops = 0
            for s_, out in zip(cases, res):
                o, d_ = ref.decode(sf, s_, cr=cr, crcc=True, error_check=ec)
                drops += d_
                assert (o is None) == (out is None) and (o is None or np.array_equal(o, out)), (sf, cr, ec, s_.size)
            assert dec.getDropped() - before == drops
            assert np.array_equal(res[0], data) and np.array_equal(res[-1], data)
    assert seen_long >= 3


def test_decoder_edge_shapes_both_kernels_vs_oracle(gpu, oracle):
    """packet lengths around the interleaver block sizes (exactly the 8 header symbols, one symbol more, a partial last block),
    every coding rate, small symbol sizes, implicit / explicit header, crc on / off, rows of every lane-group size (8 / 16 / 32 /
    64 lanes per packet): the group kernel, the lane-per-packet checker and the CPU oracle must agree on every output"""
    import lora_sdr_amd as L
    rng = np.random.default_rng(11)
    dec = L.LoRaDecoder()
    total = 0
    for variant in (0, 1):
        dec._ctx.set_variant(variant)
        rng = np.random.default_rng(11)
        for sf, ppm in ((7, 0), (7, 5), (9, 6), (12, 0), (10, 8)):
            for rdd in range(5):
                for explicit in (True, False):
                    for crcc in (False, True):
                        configure(dec, sf, ppm, rdd, int(crcc), 1, 0, int(explicit), 0, 6)
                        for lens in ((8, 9, 11, 12, 15, 16, 17), (23, 24, 40, 41), (100, 129, 160), (161, 300, 512)):
                            pk = [rng.integers(0, 1 << sf, n).astype(np.uint16) for n in lens]
                            res = dec.work(pk)
                            for s_, out in zip(pk, res):
                                o, _ = oracle.decode(sf, s_, ppm=ppm, cr=RDD_TO_CR[rdd], crcc=crcc, explicit=explicit, data_length=6)
                                assert (o is None) == (out is None), (variant, sf, ppm, rdd, explicit, crcc, len(s_))
                                if o is not None:
                                    assert np.array_equal(o, out), (variant, sf, ppm, rdd, explicit, crcc, len(s_))
                                total += 1
    assert total == 2 * 5 * 5 * 2 * 2 * 17


def test_decoder_refuses_configurations_the_reference_cannot_decode(gpu):
    """an explicit header needs at least 5 bits per symbol (the reference whitens `PPM - 5` codewords as an unsigned short, i.e.
    ~65535, past its buffer: LoRaDecoder.cpp:235); spreading factors beyond 12 do not exist on this path. The library refuses
    both instead of corrupting device memory."""
    import lora_sdr_amd as L
    dec = L.LoRaDecoder()
    pk = [np.arange(16, dtype=np.uint16)]
    dec.setSpreadFactor(7); dec.setSymbolSize(4); dec.enableExplicit(True)
    with pytest.raises(L.LoraHipError):
        dec.work(pk)
    dec.enableExplicit(False)
    assert len(dec.work(pk)) == 1                                  # implicit header: fine
    dec.setSymbolSize(0); dec.setSpreadFactor(13)
    with pytest.raises(L.LoraHipError):
        dec.work(pk)


def loopback_case(torch, ref, cr, sigma, seed, sf=10, n_packets=5, padding=512, mtu=512):
    """TestLoopback.cpp:66-133 with the path's blocks on the device: feeder -> encoder (verbatim LoRaEncoder.cpp through oracle/_ref:
    not on the path) -> lorahip_mod_frames -> + lorahip_add_awgn -> streaming demodulator -> lorahip_decode_packets, and the
    reference's own demodulator + decoder (verbatim LoRaDemod.cpp / LoRaDecoder.cpp) on the SAME noisy samples.
    Returns (sent, hip_packets, ref_packets, hip_bytes, ref_bytes)."""
    import lora_sdr_amd as L
    rng = np.random.default_rng(seed)
    N = 1 << sf
    sent = [rng.integers(0, 256, int(rng.integers(8, 129))).astype(np.uint8) for _ in range(n_packets)]   # testPlan :103-111
    ctx = L.Context(sf)
    frames = []
    for data in sent:
        syms = ref.encode(sf, data, cr=cr)                               # encoder defaults: explicit header, crc, whitening
        tx = torch.from_numpy(syms.astype(np.int16).reshape(1, -1)).cuda()
        frames.append(ctx.mod_frames(tx, ampl=1.0, padding=padding).reshape(-1))   # mod.setAmplitude(1.0), setPadding(512) :96,100
    iq = torch.cat(frames + [torch.zeros(2 * N, dtype=torch.complex64, device="cuda")]).reshape(1, -1).contiguous()
    ctx.add_awgn(iq, sigma=sigma, seed=seed)                             # noise.setAmplitude(4.0), "NORMAL" :97-99
    d = L.LoRaDemod(sf, n_channels=1)
    d.setMTU(mtu)                                                        # demod.setMTU(512) :101
    d.work(iq)
    hip_pk = [p[2] for p in d.packets()]
    dec = L.LoRaDecoder()
    dec.setSpreadFactor(sf); dec.setCodingRate(cr)                       # :93-95; everything else at the block's defaults
    hip_bytes = [o for o in dec.work(hip_pk) if o is not None]
    host = iq.cpu().numpy().reshape(-1)
    ref_pk = [p for _c, p in ref.demod_run(sf, host, mtu=mtu)["packets"]]
    ref_bytes = [o for o in (ref.decode(sf, p.astype(np.uint16), cr=cr)[0] for p in ref_pk) if o is not None]
    d.close(); ctx.close()
    return sent, hip_pk, ref_pk, hip_bytes, ref_bytes


# /comms/noise_source is not in the reference tree (PothosComms) and its scaling is unpinned. Two readings of
# "amplitude 4.0, NORMAL" are run: 4.0 as the standard deviation of each of I and Q (complex noise power 32: -15 dB SNR), and 4.0
# as the RMS of the complex sample (sigma = 4/sqrt(2) per component: -12 dB SNR).
@pytest.mark.parametrize("sigma", [4.0, 4.0 / 2 ** 0.5], ids=["sigma4_per_component", "rms4_complex"])
@pytest.mark.parametrize("cr", ["4/7", "4/8"])
def test_loopback_at_the_reference_parameters(gpu, ref, cr, sigma):
    """the reference's integration test (TestLoopback.cpp:66-133: SF10, CR 4/7 and 4/8, unit signal, noise amplitude 4.0, padding
    512, MTU 512, 5 packets of 8..128 random bytes) through the HIP chain: the symbol packets and the decoded bytes equal what the
    verbatim LoRaDemod.cpp + LoRaDecoder.cpp deliver on the same noisy samples, and equal the bytes that were sent (verifyTestPlan)"""
    sent, hip_pk, ref_pk, hip_bytes, ref_bytes = loopback_case(gpu, ref, cr, sigma, seed=20 + int(cr[-1]))
    assert len(hip_pk) == len(ref_pk) and all(np.array_equal(a, b) for a, b in zip(hip_pk, ref_pk))
    assert len(hip_bytes) == len(ref_bytes) and all(np.array_equal(a, b) for a, b in zip(hip_bytes, ref_bytes))
    assert len(hip_bytes) == len(sent) and all(np.array_equal(a, b) for a, b in zip(hip_bytes, sent))
