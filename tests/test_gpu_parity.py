"""GPU: the HIP path (through the C ABI) against the CPU oracle on identical IQ.

Bar (BASELINE.json north_star): symbol indices bit-exact; FFT bins within 1e-4 relative --
the kernels do better: bins are compared for exact equality (same fp32 operation graph as
kissfft, no FMA). power / powerAvg / fIndex go through log10 / hypot whose device and glibc
implementations may differ in the last float ulp, and powerAvg through an fp64 sum taken in
tree order instead of sequentially: tolerances below are absolute, in dB / bins.
"""
import numpy as np
import pytest

from conftest import same_values

pytestmark = pytest.mark.gpu

TOL_DB = 2e-5       # power, powerAvg [dB]: <= ~3 float ulp at 60 dB
TOL_FIDX = 2e-6     # fIndex [bins]
# powerAvg = 20log10(sqrt(float(total - maxValue))) with total an fp64 running sum of float |X|^2.
# While the bins span < 2^29 (87 dB) every partial sum is exact in fp64 and any summation order gives
# the same bits. Beyond that (a noise-free synthetic tone: floor at -140 dB) the reference's own value is
# set by the rounding of ITS summation order; the tree order of the GPU reduction differs there by a
# fraction of a dB. Such windows are checked to TOL_DB_CLEAN.
CLEAN_SNR_DB = 80.0
TOL_DB_CLEAN = 1.0


def pavg_err_ok(got, ref_pavg, ref_power):
    got, ref_pavg, ref_power = (np.asarray(x, np.float64) for x in (got, ref_pavg, ref_power))
    fin = np.isfinite(ref_pavg)
    clean = fin & ((ref_power - ref_pavg) > CLEAN_SNR_DB)
    err = np.abs(got - ref_pavg)
    ok = np.all(err[fin & ~clean] <= TOL_DB) and np.all(err[clean] <= TOL_DB_CLEAN)
    return bool(ok), (float(err[fin & ~clean].max()) if (fin & ~clean).any() else 0.0)


def to_np(t):
    return t.detach().cpu().numpy()


def sym_np(t):
    return to_np(t).view(np.uint16)


def check(o, g, fft=True, dec=False, where=""):
    assert np.array_equal(sym_np(g["sym"]), o["sym"]), "symbol index mismatch " + where
    if fft:
        assert same_values(to_np(g["fft"]), o["fft"]), "FFT bins differ " + where
    if dec:
        assert same_values(to_np(g["dec"]), o["dec"]), "dechirped samples differ " + where
    with np.errstate(invalid="ignore"):
        ok, worst = pavg_err_ok(to_np(g["powerAvg"]), o["powerAvg"], o["power"])
        assert ok, "powerAvg differs by %g %s" % (worst, where)
        for k, tol in (("power", TOL_DB), ("fIndex", TOL_FIDX)):
            a, b = to_np(g[k]).astype(np.float64), o[k].astype(np.float64)
            fin = np.isfinite(b)
            assert np.array_equal(np.isfinite(a), fin), k + " finiteness " + where
            assert np.array_equal(a[~fin], b[~fin]) or np.all(np.isnan(a[~fin]) == np.isnan(b[~fin])), k
            err = np.abs(a[fin] - b[fin]).max() if fin.any() else 0.0
            assert err <= tol, "%s differs by %g %s" % (k, err, where)


def make_iq(rng, sf, W, snr_db=None, kind="chirp"):
    """W windows of random symbols plus AWGN.
    chirp:   down-chirp table x tone -> dechirps to exactly bin sym
    halfbin: analytic up-chirp whose dechirped tone sits at sym - 0.5: two near-equal bins
    noise:   no signal at all"""
    import lora_sdr_amd as L
    N = 1 << sf
    t = np.arange(N)
    sym = rng.integers(0, N, W)
    if kind == "chirp":
        down = L.host_tables(sf, fine=False)[1].astype(np.complex128)
        x = down[None, :] * np.exp(2j * np.pi * sym[:, None] * t[None, :] / N)
    elif kind == "halfbin":
        ph = np.pi * t * t / N - np.pi * t
        x = np.exp(1j * (ph[None, :] + 2 * np.pi * sym[:, None] * t[None, :] / N))
    else:
        x = np.zeros((W, N), np.complex128)
        snr_db = 0.0
    if snr_db is not None:
        sigma = np.sqrt(10 ** (-snr_db / 10) / 2)
        x = x + sigma * (rng.standard_normal((W, N)) + 1j * rng.standard_normal((W, N)))
    return x.astype(np.complex64), sym


@pytest.mark.parametrize("sf", range(6, 13))
def test_random_symbols_awgn(gpu, oracle, sf):
    import lora_sdr_amd as L
    rng = np.random.default_rng(sf)
    W = {6: 700, 7: 515, 8: 300, 9: 130, 10: 70, 11: 33, 12: 19}[sf]   # ragged: not a multiple of windows/block
    snr = -5.0 if sf >= 9 else 5.0       # keep the post-FFT SNR comfortably above the error floor
    iq, sent = make_iq(rng, sf, W, snr_db=snr)
    ctx = L.Context(sf)
    g = ctx.detect_batch(gpu.from_numpy(iq).cuda(), want_fft=True, want_dec=True, want_fine_idx=True)
    gpu.cuda.synchronize()
    o = oracle.detect_batch(sf, iq, want_fft=True, want_dec=True)
    check(o, g, dec=True, where="sf%d" % sf)
    assert np.array_equal(to_np(g["fineIdxOut"]), o["fineIdxOut"])
    # and the sanity of the test itself: every symbol is recovered
    assert np.array_equal(o["sym"], sent.astype(np.uint16))


@pytest.mark.parametrize("sf", [7, 9, 11])
def test_half_bin_offset_near_ties(gpu, oracle, sf):
    """tone exactly between two bins: the two candidates differ by rounding only"""
    import lora_sdr_amd as L
    rng = np.random.default_rng(30 + sf)
    W = 257
    iq, _ = make_iq(rng, sf, W, snr_db=None, kind="halfbin")
    ctx = L.Context(sf)
    g = ctx.detect_batch(gpu.from_numpy(iq).cuda(), want_fft=True)
    o = oracle.detect_batch(sf, iq, want_fft=True)
    check(o, g, where="halfbin sf%d" % sf)


@pytest.mark.parametrize("sf", [7, 10, 12])
def test_noise_only_windows_bit_exact_argmax(gpu, oracle, sf):
    """no peak at all: near-equal maxima everywhere; only an identical FFT gives identical indices"""
    import lora_sdr_amd as L
    rng = np.random.default_rng(50 + sf)
    W = {7: 4096, 10: 512, 12: 128}[sf]
    iq, _ = make_iq(rng, sf, W, kind="noise")
    ctx = L.Context(sf)
    g = ctx.detect_batch(gpu.from_numpy(iq).cuda(), want_fft=True)
    o = oracle.detect_batch(sf, iq, want_fft=True, nthreads=4)
    check(o, g, where="noise sf%d" % sf)


@pytest.mark.parametrize("sf", [7, 9, 12])
def test_chirp_tables_and_none(gpu, oracle, sf):
    """per-window chirp_sel: up / down / none (the LoRaDetector::feed seam)"""
    import lora_sdr_amd as L
    rng = np.random.default_rng(70 + sf)
    W = 96
    iq, _ = make_iq(rng, sf, W, snr_db=3.0)
    sel = rng.integers(0, 3, W).astype(np.int32)
    ctx = L.Context(sf)
    g = ctx.detect_batch(gpu.from_numpy(iq).cuda(), chirp_sel=gpu.from_numpy(sel).cuda(), want_fft=True, want_dec=True)
    o = oracle.detect_batch(sf, iq, chirp_sel=sel, want_fft=True, want_dec=True)
    check(o, g, dec=True, where="sel sf%d" % sf)
    for s in (0, 1, 2):
        g = ctx.detect_batch(gpu.from_numpy(iq).cuda(), chirp_sel_all=s, want_fft=True)
        o = oracle.detect_batch(sf, iq, chirp_sel=s, want_fft=True)
        check(o, g, where="sel_all=%d sf%d" % (s, sf))


@pytest.mark.parametrize("sf", [6, 7, 8, 9, 10, 11, 12])
def test_fine_tune_recurrence(gpu, oracle, sf):
    """LoRaDemod.cpp:160-162: the int<-float index recurrence for many (idx0, err) pairs,
    including wrap in both directions, integer and tiny steps, and large indices where
    float(idx) - d rounds differently"""
    import lora_sdr_amd as L
    rng = np.random.default_rng(90 + sf)
    N = 1 << sf
    M = 128 * N
    errs = np.array([0.0, 0.3, -0.3, 1.0, -1.0, 0.0078125, -0.0078125, 0.01, 1e-4, -1e-4, 2.5, -7.25, 0.4999, 17.3,
                     -33.7, N / 4 + 0.37, -(N / 4) - 0.12], np.float32)
    idx0 = np.array([0, 1, M - 1, M // 2, 5, M - 5, 12345 % M, M // 3], np.int32)
    E, I = np.meshgrid(errs, idx0)
    E, I = E.reshape(-1).astype(np.float32), I.reshape(-1).astype(np.int32)
    extra = 40
    E = np.concatenate([E, rng.uniform(-3, 3, extra).astype(np.float32)])
    I = np.concatenate([I, rng.integers(0, M, extra).astype(np.int32)])
    W = E.size
    iq, _ = make_iq(rng, sf, W, snr_db=10.0)
    sel = (np.arange(W) % 2).astype(np.int32)
    ctx = L.Context(sf)
    g = ctx.detect_batch(gpu.from_numpy(iq).cuda(), chirp_sel=gpu.from_numpy(sel).cuda(),
                         fine_idx0=gpu.from_numpy(I).cuda(), fine_err=gpu.from_numpy(E).cuda(),
                         want_fft=True, want_dec=True, want_fine_idx=True)
    o = oracle.detect_batch(sf, iq, chirp_sel=sel, fine_idx0=I, fine_err=E, want_fft=True, want_dec=True)
    assert np.array_equal(to_np(g["fineIdxOut"]), o["fineIdxOut"]), "index recurrence end state"
    check(o, g, dec=True, where="fine sf%d" % sf)


@pytest.mark.parametrize("sf", range(6, 13))
def test_fine_index_closed_form_special_cases(gpu, oracle, sf):
    """The tuned kernels evaluate the index recurrence in closed form and the table entry from the split tables
    (lorahip_fine.h). The cases where the form is not the plain modular one, in whole batches so that every wave / workgroup
    meets them: the index walking down to 0 and sticking there (0 < d < 1), landing on ceil(d) - 1 (the reference yields 0
    there), steps a hair above an integer (float rounding reaches the next integer: serial chain), integer steps, both
    signs -- with the split tables and with the table gather; all against the oracle's serial recurrence."""
    import lora_sdr_amd as L
    rng = np.random.default_rng(700 + sf)
    N = 1 << sf
    M = 128 * N
    cases = []
    for d in (0.3, 0.9, 0.02):                                          # sticks at 0 inside the window, before it, never
        cases += [(d, N // 3), (d, 0), (d, 5 * N)]
    for d in (38.4, 2.5, 129.7):                                        # lands on ceil(d) - 1 after a few steps
        c = int(np.ceil(d))
        cases += [(d, c - 1 + 5 * c), (d, c - 1), (d, (c - 1 + (N // 2) * c) % M)]
    for d in (38 + 2.0 ** -9, 7 + 2.0 ** -12, -(5 + 1 - 2.0 ** -10)):   # float rounding reaches the next integer in the high binades
        cases += [(d, M - 3), (d, M // 2 + 1), (d, 17)]
    for d in (3.0, -3.0, 128.0, -0.4, -77.7):
        cases += [(d, 1), (d, M - 2)]
    reps = {6: 600, 7: 600, 8: 300, 9: 150, 10: 80, 11: 40, 12: 24}[sf]
    E = np.tile(np.array([d / 128.0 for d, _ in cases], np.float32), reps)
    I = np.tile(np.array([i for _, i in cases], np.int32), reps)
    W = E.size
    iq, _ = make_iq(rng, sf, W, snr_db=8.0)
    o = oracle.detect_batch(sf, iq, fine_idx0=I, fine_err=E, nthreads=8)
    ctx = L.Context(sf)
    t = gpu.from_numpy
    for gather in (False, True):
        ctx.set_fine_gather(gather)
        assert ctx.fine_split_active() == (not gather)
        g = ctx.detect_batch(t(iq).cuda(), fine_idx0=t(I).cuda(), fine_err=t(E).cuda(), want_fine_idx=True)
        gpu.cuda.synchronize()
        assert np.array_equal(to_np(g["fineIdxOut"]), o["fineIdxOut"]), "index recurrence end state (gather=%s)" % gather
        check(o, g, fft=False, where="fine special cases sf%d gather=%s" % (sf, gather))


def test_offsets_stride_and_overlap(gpu, oracle):
    """windows at arbitrary sample offsets (the sync machine consumes N-value, N/4+err, 2N ...)"""
    import lora_sdr_amd as L
    sf, N = 8, 256
    rng = np.random.default_rng(5)
    stream = (rng.standard_normal(50 * N) + 1j * rng.standard_normal(50 * N)).astype(np.complex64)
    off = np.sort(rng.integers(0, 49 * N, 333)).astype(np.int64)
    ctx = L.Context(sf)
    d = gpu.from_numpy(stream).cuda()
    g = ctx.detect_batch(d, offsets=gpu.from_numpy(off).cuda(), want_fft=True)
    o = oracle.detect_batch(sf, stream, offsets=off, want_fft=True)
    check(o, g, where="offsets")
    g = ctx.detect_batch(d, n_windows=97, window_stride=N // 2, want_fft=True)      # 50% overlapped
    o = oracle.detect_batch(sf, stream, offsets=np.arange(97, dtype=np.int64) * (N // 2), want_fft=True)
    check(o, g, where="stride")


def test_edge_batches(gpu, oracle):
    import lora_sdr_amd as L
    sf, N = 7, 128
    ctx = L.Context(sf)
    # empty batch: no launch, no error
    g = ctx.detect_batch(gpu.zeros(0, dtype=gpu.complex64, device="cuda"))
    assert g["sym"].numel() == 0
    # one window; all-zero input -> index 0, -inf powers (log10(0)), fIndex 0 (demon == 0)
    z = np.zeros((1, N), np.complex64)
    g = ctx.detect_batch(gpu.from_numpy(z).cuda(), chirp_sel_all=L.CHIRP_NONE, want_fft=True)
    o = oracle.detect_batch(sf, z, chirp_sel=2, want_fft=True)
    check(o, g, where="zeros")
    assert sym_np(g["sym"])[0] == 0 and to_np(g["fIndex"])[0] == 0.0
    # exact tie between two bins: the lowest index wins (strict '>')
    t = np.arange(N)
    x = (np.exp(2j * np.pi * 5 * t / N) + np.exp(2j * np.pi * 69 * t / N)).astype(np.complex64)[None]
    g = ctx.detect_batch(gpu.from_numpy(x).cuda(), chirp_sel_all=L.CHIRP_NONE, want_fft=True)
    o = oracle.detect_batch(sf, x, chirp_sel=2, want_fft=True)
    check(o, g, where="tie")


def test_host_pointer_entry(gpu, oracle):
    """lorahip_detect_batch_host: numpy in, numpy out, staged through the context"""
    import lora_sdr_amd as L
    sf = 9
    rng = np.random.default_rng(11)
    iq, _ = make_iq(rng, sf, 77, snr_db=0.0)
    ctx = L.Context(sf)
    g = ctx.detect_batch(iq, want_fft=True, want_dec=True, want_fine_idx=True, fine_err=0.25)
    o = oracle.detect_batch(sf, iq, want_fft=True, want_dec=True, fine_err=0.25)
    assert np.array_equal(g["sym"], o["sym"]) and same_values(g["fft"], o["fft"]) and same_values(g["dec"], o["dec"])
    assert np.array_equal(g["fineIdxOut"], o["fineIdxOut"])
    assert np.abs(g["power"] - o["power"]).max() <= TOL_DB


def test_detector_shim_test_detector(gpu, oracle, golden):
    """TestDetector.cpp:9-35 through the LoRaDetector shim (feed/detect), a sample of symbols,
    and the full 1024-symbol sweep through the batch entry, against the reference's own values"""
    import lora_sdr_amd as L
    g = golden("test_detector_n1024.npz")
    N = 1024
    down, _ = oracle.genchirp(N, 1, N, 0.0, True, 1.0, 0.0)
    det = L.LoRaDetector(N)
    wins = np.empty((N, N), np.complex64)
    for sym in range(N):
        ch, _ = oracle.genchirp(N, 1, N, np.float32(2 * np.pi * sym) / N, False, 1.0, np.float32(np.pi / 4))
        wins[sym] = down * ch
    for sym in (0, 1, 511, 777, 1023):
        for i in range(N):
            det.feed(i, wins[sym, i])
        fft = np.empty(N, np.complex64)
        index, power, powerAvg, fIndex = det.detect(fft)
        assert index == sym and power > -10.0
        assert abs(power - g["power"][sym]) <= TOL_DB and abs(fIndex - g["fIndex"][sym]) <= TOL_FIDX
        assert same_values(fft, oracle.detect(wins[sym])[4])
    ctx = L.Context(10)
    r = ctx.detect_batch(gpu.from_numpy(wins).cuda(), chirp_sel_all=L.CHIRP_NONE)
    assert np.array_equal(sym_np(r["sym"]), np.arange(N))
    assert (to_np(r["power"]) > -10.0).all()
    assert np.abs(to_np(r["power"]) - g["power"]).max() <= TOL_DB
    assert pavg_err_ok(to_np(r["powerAvg"]), g["powerAvg"], g["power"])[0]   # -140 dB floor of a clean tone
    assert np.abs(to_np(r["fIndex"]) - g["fIndex"]).max() <= TOL_FIDX


@pytest.mark.parametrize("sf", range(6, 13))
def test_golden_detector_kat(gpu, golden, sf):
    """committed vectors from the real reference: chirp+noise, noise, zeros, last bin, twin peaks, off-bin tone"""
    import lora_sdr_amd as L
    g = golden("detector_kat.npz")
    ctx = L.Context(sf)
    r = ctx.detect_batch(gpu.from_numpy(g["in_%d" % sf]).cuda(), chirp_sel_all=L.CHIRP_NONE, want_fft=True)
    assert np.array_equal(sym_np(r["sym"]), g["sym_%d" % sf])
    assert same_values(to_np(r["fft"]), g["fft_%d" % sf])
    ref_p, ref_a, ref_f = g["power_%d" % sf], g["powerAvg_%d" % sf], g["fIndex_%d" % sf]
    fin = np.isfinite(ref_p)
    assert np.abs(to_np(r["power"])[fin] - ref_p[fin]).max() <= TOL_DB
    assert pavg_err_ok(to_np(r["powerAvg"]), ref_a, ref_p)[0]
    assert np.abs(to_np(r["fIndex"]) - ref_f).max() <= TOL_FIDX


# the kernel variants the library ships (lorahip_set_variant): 0 = tuned default, 1 = generic kernel, 10 = the one alternative per
# SF; the round-1 A/B zoo is compiled only with -DLORAHIP_ALL_VARIANTS (profiles/r01/s8_variants.txt is its record)
VARIANTS = {sf: [0, 1, 10] for sf in range(6, 13)}


@pytest.mark.parametrize("sf", range(6, 13))
def test_steady_state_kernels_all_variants(gpu, oracle, sf):
    """The launch-uniform, no-debug-output kernels (the shape bench.py times), batches large enough that every
    persistent workgroup runs several window sets and the deferred tails are flushed mid-loop and at the end;
    every variant against the oracle, and the debug-output kernel against the same."""
    import lora_sdr_amd as L
    rng = np.random.default_rng(200 + sf)
    W = {6: 70001, 7: 70001, 8: 30011, 9: 16001, 10: 9001, 11: 4099, 12: 2051}[sf]
    snr = -5.0 if sf >= 9 else 5.0
    iq, sent = make_iq(rng, sf, W, snr_db=snr)
    # sprinkle degenerate windows: all-zero, and pure noise
    iq[17] = 0
    iq[W - 1] = (rng.standard_normal(1 << sf) + 1j * rng.standard_normal(1 << sf)).astype(np.complex64)
    o = oracle.detect_batch(sf, iq, want_fft=False, nthreads=8)
    d = gpu.from_numpy(iq).cuda()
    ctx = L.Context(sf)
    for v in VARIANTS[sf]:
        ctx.set_variant(v)
        for sel in (L.CHIRP_UP,):
            g = ctx.detect_batch(d, chirp_sel_all=sel)
            gpu.cuda.synchronize()
            check(o, g, fft=False, where="steady sf%d variant %d" % (sf, v))
    ctx.set_variant(0)
    o2 = oracle.detect_batch(sf, iq[:1500], chirp_sel=1, want_fft=True, nthreads=8)
    g2 = ctx.detect_batch(d[:1500], chirp_sel_all=L.CHIRP_DOWN, want_fft=True)
    check(o2, g2, where="steady-dbg sf%d" % sf)


@pytest.mark.parametrize("sf", [7, 8, 9, 10, 11, 12])
def test_per_window_settings_at_scale(gpu, oracle, sf):
    """per-window chirp selection, fine-tune start index and error on batches large enough that every persistent wave /
    workgroup runs many window sets (the index chain's LDS scratch aliases the exchange region of the previous set)"""
    import lora_sdr_amd as L
    rng = np.random.default_rng(300 + sf)
    N = 1 << sf
    M = 128 * N
    W = {7: 60013, 8: 30011, 9: 12007, 10: 6007, 11: 3001, 12: 1801}[sf]
    iq, _ = make_iq(rng, sf, W, snr_db=0.0 if sf >= 9 else 6.0)
    sel = rng.integers(0, 3, W).astype(np.int32)
    err = np.where(rng.random(W) < 0.5, 0.0, rng.uniform(-2.5, 2.5, W)).astype(np.float32)
    err[rng.random(W) < 0.05] = np.float32(N / 8 + 0.3)                   # a few large ones: several wraps per window
    idx0 = rng.integers(0, M, W).astype(np.int32)
    ctx = L.Context(sf)
    t = gpu.from_numpy
    g = ctx.detect_batch(t(iq).cuda(), chirp_sel=t(sel).cuda(), fine_idx0=t(idx0).cuda(), fine_err=t(err).cuda(), want_fine_idx=True)
    gpu.cuda.synchronize()
    o = oracle.detect_batch(sf, iq, chirp_sel=sel, fine_idx0=idx0, fine_err=err, nthreads=8)
    assert np.array_equal(to_np(g["fineIdxOut"]), o["fineIdxOut"]), "index recurrence end state"
    check(o, g, fft=False, where="per-window at scale sf%d" % sf)


@pytest.mark.parametrize("sf", [7, 12])
def test_batches_beyond_2_pow_31_samples(gpu, oracle, sf):
    """one launch over 2.5 G samples (20 GB of IQ, what 288 GB of HBM invites): sample offsets no longer fit 32 bits. Every window
    must still decode to the symbol that was synthesised into it, and the windows past the 2^31-sample line equal the CPU oracle."""
    import lora_sdr_amd as L
    torch = gpu
    N = 1 << sf
    W = (5 << 29) // N                                            # 2.5 * 2^30 samples
    free, _ = torch.cuda.mem_get_info()
    if free < 26 * (1 << 30):
        pytest.skip("needs 26 GB of free HBM")
    ctx = L.Context(sf)
    g = torch.Generator(device="cuda"); g.manual_seed(31 + sf)
    sym = torch.randint(0, N, (W,), generator=g, device="cuda", dtype=torch.int32).to(torch.int16)
    iq = ctx.synth_symbols(sym, ampl=1.0, noise_sigma=0.3, seed=9)
    r = ctx.detect_batch(iq)
    torch.cuda.synchronize()
    got = r["sym"].view(torch.int16).to(torch.int64) & 0xffff
    assert bool((((got - (sym.to(torch.int64) & 0xffff)) % N) == 1).all())          # genChirp symbol s -> bin s+1
    first = ((1 << 31) // N) - 8                                   # 16 windows straddling sample 2^31, and the last 16
    for lo in (first, W - 16):
        o = oracle.detect_batch(sf, iq[lo * N:(lo + 16) * N].cpu().numpy(), nthreads=4)
        assert np.array_equal(o["sym"], r["sym"][lo:lo + 16].cpu().numpy().view(np.uint16))
        assert np.abs(o["power"] - r["power"][lo:lo + 16].cpu().numpy()).max() <= TOL_DB
    del iq, r
    torch.cuda.empty_cache()


def test_argument_validation_at_the_boundary(gpu):
    """host-pointer batches with a fine-tune index outside the 128*N-entry table are refused (the reference would read past its
    table); the Python mirror refuses a frame stride shorter than the frame"""
    import lora_sdr_amd as L
    ctx = L.Context(7)
    iq = np.zeros(4 * 128, np.complex64)
    for bad in (-1, 128 * 128):
        with pytest.raises(L.LoraHipError):
            ctx.detect_batch(iq, fine_idx0=np.array([0, bad, 0, 0], np.int32), fine_err=np.zeros(4, np.float32))
    ok = ctx.detect_batch(iq, fine_idx0=np.array([0, 128 * 128 - 1, 5, 0], np.int32), fine_err=np.full(4, 0.3, np.float32))
    assert ok["sym"].shape == (4,)
    syms = gpu.zeros((2, 5), dtype=gpu.int16, device="cuda")
    with pytest.raises(ValueError):
        ctx.mod_frames(syms, frame_stride=ctx.mod_frame_len(5) - 1)


@pytest.mark.parametrize("sf", [7, 10, 12])
def test_synth_symbols_is_genchirp(gpu, oracle, sf):
    """lorahip_synth_symbols -- the generator of every benchmark's input -- against genChirp (ChirpGenerator.hpp:22-47: f from
    -pi + f0, f += 2 pi / N BEFORE use, wrap at +pi, phaseAccum += f, polar(ampl, phaseAccum)), per window from phase 0:
    (1) against the definition in exact arithmetic (float64 here): only the final rounding of cos / sin separates them;
    (2) against the reference's own float recurrence (the pinned restatement oracle.genchirp): the recurrence accumulates float
        rounding in f and phaseAccum (up to ~9e-6 N: measured 4.8e-4 / 8.8e-3 / 3.5e-2 at SF7 / 10 / 12), so the tolerance is
        2e-5 N -- far below the 2 pi / N spacing of anything the demodulator resolves;
    (3) both demodulate to the same bins."""
    import lora_sdr_amd as L
    torch = gpu
    N = 1 << sf
    rng = np.random.default_rng(sf)
    sym = np.concatenate([rng.integers(0, N, 29), [0, 1, N - 1, N // 2]]).astype(np.uint16)
    ampl = 0.75
    ctx = L.Context(sf)
    got = ctx.synth_symbols(torch.from_numpy(sym.view(np.int16)).cuda(), ampl=ampl).cpu().numpy().reshape(len(sym), N)
    i = np.arange(N)
    n1 = i + 1.0
    for k, s in enumerate(sym.tolist()):
        f = -np.pi + 2 * np.pi * s / N + n1 * (2 * np.pi / N)                  # f after its pre-increment, before the wrap
        f = np.where(f > np.pi, f - 2 * np.pi, f)                               # :31 / :39
        exact = ampl * np.exp(1j * np.cumsum(f))
        assert np.abs(got[k] - exact).max() <= 1e-6 * N ** 0.5 + 2e-7, (s, np.abs(got[k] - exact).max())
        ref, _ = oracle.genchirp(N, 1, N, np.float32(2 * np.pi * s / N), False, ampl, 0.0)
        assert np.abs(got[k] - ref).max() <= 2e-5 * N, (s, np.abs(got[k] - ref).max())
    refs = np.stack([oracle.genchirp(N, 1, N, np.float32(2 * np.pi * s / N), False, ampl, 0.0)[0] for s in sym.tolist()])
    a, b = oracle.detect_batch(sf, got.reshape(-1)), oracle.detect_batch(sf, refs.reshape(-1))
    assert np.array_equal(a["sym"], b["sym"]) and np.array_equal(a["sym"], (sym.astype(np.int64) + 1) % N)   # the +1 bin of SURVEY.md section 7h
    ctx.close()


def _bits_differ(a, b):
    """floats that differ bit for bit, NaN against NaN not counted (payloads are not part of the contract)"""
    a = np.ascontiguousarray(a).view(np.float32).ravel()
    b = np.ascontiguousarray(b).view(np.float32).ravel()
    return int(np.count_nonzero((a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))))


@pytest.mark.parametrize("sf", [7, 9, 10, 12])
def test_inputs_at_the_edges_of_fp32(gpu, oracle, sf):
    """Amplitudes whose squares are subnormal (1e-19 .. 3e-39) or overflow (1e17 .. 3e37), windows that hold a NaN sample and
    windows that hold an infinite one, with and without a moving fine-tune index. The kernels keep subnormals (no flush to zero)
    and follow the reference's operation graph -- with the textbook complex product, i.e. the reference as built with
    -fcx-limited-range once infinities appear (a plain -O2 build goes through libgcc's Annex G recovery there:
    tests/test_oracle_vs_ref.py::test_non_finite_samples_where_the_reference_is_defined) -- so dechirped samples, FFT bins and the
    index are identical bit for bit, +-Inf / NaN included, and power / powerAvg / fIndex are of the same class (finite, +Inf, -Inf, NaN) and within two float ulps.
    ONE deliberate difference, stated in DESIGN.md section 2: kissfft multiplies by the twiddle (1, 0) where the kernels skip the
    multiplication, and Inf * 0 = NaN -- in a window that holds an INFINITE sample some bin components are Inf here and NaN in
    the reference. Every bin of such a window is non-finite either way (a non-finite sample leaves the two dechirp
    multiplications with a NaN component), so |X|^2 is NaN for every bin in both and everything detect() returns is the same."""
    import lora_sdr_amd as L
    rng = np.random.default_rng(50 + sf)
    N, W = 1 << sf, 36
    t = np.arange(N)
    down = L.host_tables(sf, fine=False)[1].astype(np.complex128)
    sym = rng.integers(0, N, W)
    base = down[None, :] * np.exp(2j * np.pi * sym[:, None] * t[None, :] / N)
    base = base + 0.1 * (rng.standard_normal((W, N)) + 1j * rng.standard_normal((W, N)))
    cases = []
    with np.errstate(over="ignore"):
        for scale in (1e-19, 1e-21, 1e-23, 3e-39, 1e17, 1e18, 3e19, 3e37):
            cases.append(("scale %g" % scale, (base * scale).astype(np.complex64), False))
    x = base.astype(np.complex64); x[::3, 5] = np.nan; cases.append(("NaN sample", x, False))
    x = base.astype(np.complex64); x[::2, :] = 0; x[::2, 3] = 1e-30; cases.append(("lone 1e-30 sample", x, False))
    x = base.astype(np.complex64); x[::3, N // 2] = np.inf; cases.append(("+Inf real part", x, True))
    x = base.astype(np.complex64); x[::3, 7] = complex(0.0, -np.inf); cases.append(("-Inf imaginary part", x, True))
    x = base.astype(np.complex64); x[::3, N - 1] = complex(np.inf, np.inf); cases.append(("Inf in both parts", x, True))
    ctx = L.Context(sf)
    for err in (0.0, 0.31):
        fe = np.full(W, err, np.float32)
        for name, iq, infinite in cases:
            where = "sf%d err %g %s" % (sf, err, name)
            g = ctx.detect_batch(gpu.from_numpy(iq).cuda(), fine_err=gpu.from_numpy(fe).cuda() if err else None, want_fft=True, want_dec=True)
            gpu.cuda.synchronize()
            with np.errstate(all="ignore"):
                o = oracle.detect_batch(sf, iq, fine_err=fe if err else None, want_fft=True, want_dec=True)
            assert np.array_equal(sym_np(g["sym"]), o["sym"]), where
            assert _bits_differ(to_np(g["dec"]), o["dec"]) == 0, where
            gf, of = to_np(g["fft"]).reshape(W, N), o["fft"].reshape(W, N)
            if infinite:
                hit = ~np.isfinite(iq).all(axis=1)
                assert hit.sum() == len(range(0, W, 3))
                assert _bits_differ(gf[~hit], of[~hit]) == 0, where
                for f in (gf[hit], of[hit]):                      # every bin non-finite, |X|^2 NaN: nothing can win the arg-max
                    with np.errstate(all="ignore"):
                        m = f.real.astype(np.float32) ** 2 + f.imag.astype(np.float32) ** 2
                    assert np.isnan(m).all(), where
                assert np.all(o["sym"][hit] == 0)
            else:
                assert _bits_differ(gf, of) == 0, where
            for k in ("power", "powerAvg", "fIndex"):
                a, b = to_np(g[k]), o[k]
                for cls in (np.isnan, np.isposinf, np.isneginf):
                    assert np.array_equal(cls(a), cls(b)), where + " " + k
                f = np.isfinite(b)
                if f.any():
                    tol = np.maximum(2 * np.spacing(np.abs(b[f]).astype(np.float32)), np.float32(TOL_DB if k != "fIndex" else TOL_FIDX))
                    assert np.all(np.abs(a[f].astype(np.float64) - b[f]) <= tol), where + " " + k


@pytest.mark.parametrize("sf", [6, 7, 8, 9, 10, 11, 12])
def test_contracted_variant_stays_inside_the_stated_tolerance(gpu, oracle, sf):
    """lorahip_set_variant(ctx, LORAHIP_VARIANT_FMA): the opt-in build whose complex multiplies are one multiply + one FMA. NOT bit
    exact -- that is its point (what do bit-exact bins cost: profiles/r04) -- but inside north_star's tolerance: bins within 1e-4 of
    the peak (measured: ~1e-7), the same symbol index wherever the peak's margin exceeds 1e-3, power within 1e-4 dB. With per-window
    settings and the debug ports too."""
    import lora_sdr_amd as L
    rng = np.random.default_rng(40 + sf)
    N, W = 1 << sf, 96
    t = np.arange(N)
    sym = rng.integers(0, N, W)
    down = L.host_tables(sf, fine=False)[1]
    x = down[None, :] * np.exp(2j * np.pi * sym[:, None] * t / N)
    x = (x + 0.5 * (rng.standard_normal(x.shape) + 1j * rng.standard_normal(x.shape))).astype(np.complex64)
    err = rng.uniform(-2, 2, W).astype(np.float32)
    idx0 = rng.integers(0, 128 * N, W).astype(np.int32)
    ctx = L.Context(sf)
    xd = gpu.from_numpy(x).cuda()
    for kw in (dict(), dict(fine_err=gpu.from_numpy(err).cuda(), fine_idx0=gpu.from_numpy(idx0).cuda())):
        okw = {k: v.cpu().numpy() for k, v in kw.items()}
        o = oracle.detect_batch(sf, x, want_fft=True, **okw)
        ctx.set_variant(40)
        g = ctx.detect_batch(xd, want_fft=True, want_dec=True, **kw)
        g2 = ctx.detect_batch(xd, **kw)                             # without the ports: the steady-state / per-window instances
        ctx.set_variant(0)
        e = ctx.detect_batch(xd, want_fft=True, **kw)
        gpu.cuda.synchronize()
        fft = g["fft"].cpu().numpy()
        peak = np.abs(o["fft"]).max(axis=1, keepdims=True)
        assert (np.abs(fft - o["fft"]) / peak).max() < 1e-5
        assert np.array_equal(e["fft"].cpu().numpy(), o["fft"])                     # the default stays bit-exact beside it
        assert not np.array_equal(fft, o["fft"])                                    # ... and the contracted build really is another graph
        assert np.array_equal(g["sym"].cpu().numpy().view(np.uint16), o["sym"])     # clear peaks: the same index
        assert np.array_equal(g2["sym"].cpu().numpy().view(np.uint16), o["sym"])
        assert np.abs(g["power"].cpu().numpy() - o["power"]).max() < 1e-4
    with pytest.raises(L.LoraHipError):
        ctx.set_variant(41)
    ctx.close()
    d = L.LoRaDemod(sf, n_channels=2)
    d.set_variant(10)
    with pytest.raises(L.LoraHipError):
        d.set_variant(40)                                           # level 3 runs the reference's operation graph only
    d.close()
