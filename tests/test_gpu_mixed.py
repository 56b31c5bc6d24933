"""GPU: the mixed-SF scheduler of the C ABI (lorahip_mixed_*, BASELINE configs[3]) against the CPU oracle, channel by channel.

A scattered IQ buffer (channels in shuffled order, gaps between them) holds S windows per channel, SF drawn from 6..12; the
scheduler's bucket-major result rows are mapped back through rows[] and every channel must equal the oracle's run of that
channel alone: symbol indices bit-exact, power / powerAvg / fIndex within the level-2 tolerances."""
import numpy as np
import pytest

from test_gpu_parity import make_iq, pavg_err_ok, TOL_DB, TOL_FIDX

pytestmark = pytest.mark.gpu


def build(rng, sfs, S):
    order = rng.permutation(len(sfs))
    offsets, chunks, at, iq_of = np.zeros(len(sfs), np.int64), [], 0, {}
    for c in order:
        sf = int(sfs[c])
        gap = int(rng.integers(0, 50))
        chunks.append(np.zeros(gap, np.complex64))
        at += gap
        iq, _ = make_iq(rng, sf, S, snr_db=0.0 if sf >= 9 else 8.0)
        iq_of[c] = iq
        offsets[c] = at
        chunks.append(iq.reshape(-1))
        at += iq.size
    return np.concatenate(chunks), offsets, iq_of


@pytest.mark.parametrize("variant", [0, 10])
def test_mixed_scheduler_matches_oracle_per_channel(gpu, oracle, variant):
    import lora_sdr_amd as L
    rng = np.random.default_rng(414 + variant)
    sfs = np.concatenate([7 + np.arange(30) % 6, [6, 6, 12, 7]]).astype(np.int32)
    S = 5
    buf, offsets, iq_of = build(rng, sfs, S)
    m = L.MixedDetector(sfs)
    m.set_variant(variant)
    assert [b[0] for b in m.buckets] == sorted(set(int(s) for s in sfs))
    assert sum(b[2] for b in m.buckets) == len(sfs) and sorted(m.rows) == list(range(len(sfs)))
    # bucket-major: rows ascend with (SF, channel)
    key = np.lexsort((np.arange(len(sfs)), sfs))
    assert np.array_equal(np.argsort(m.rows), key)
    m.plan(offsets, S)
    d = gpu.from_numpy(buf.view(np.float32)).cuda()
    gpu.cuda.synchronize()
    out = m.detect(d)
    again = m.detect(d)                                   # a second step into fresh arrays: same bits
    for k in out:
        assert gpu.equal(out[k], again[k]) or k != "sym"
    got = {k: v.cpu().numpy() for k, v in out.items()}
    for c in range(len(sfs)):
        o = oracle.detect_batch(int(sfs[c]), iq_of[c])
        r = int(m.rows[c])
        assert np.array_equal(got["sym"][r].view(np.uint16), o["sym"]), "channel %d (sf %d)" % (c, sfs[c])
        ok, worst = pavg_err_ok(got["powerAvg"][r], o["powerAvg"], o["power"])
        assert ok, worst
        assert np.abs(got["power"][r].astype(np.float64) - o["power"]).max() <= TOL_DB
        assert np.abs(got["fIndex"][r].astype(np.float64) - o["fIndex"]).max() <= TOL_FIDX
    m.close()


def test_mixed_scheduler_equals_single_sf_contexts(gpu):
    """the scheduler adds no arithmetic: a bucket's rows equal a plain level-2 launch over the same windows, bit for bit"""
    import lora_sdr_amd as L
    rng = np.random.default_rng(99)
    sfs = (7 + np.arange(24) % 6).astype(np.int32)
    S = 16
    buf, offsets, iq_of = build(rng, sfs, S)
    m = L.MixedDetector(sfs)
    m.plan(offsets, S)
    out = m.detect(gpu.from_numpy(buf.view(np.float32)).cuda())
    for c in range(len(sfs)):
        ctx = L.Context(int(sfs[c]))
        g = ctx.detect_batch(gpu.from_numpy(iq_of[c]).cuda())
        gpu.cuda.synchronize()
        r = int(m.rows[c])
        for k in ("sym", "power", "powerAvg", "fIndex"):
            assert gpu.equal(out[k][r].cpu(), g[k].cpu().reshape(-1)), (c, k)
        ctx.close()
    m.close()


def test_mixed_scheduler_refuses_bad_input(gpu):
    import lora_sdr_amd as L
    with pytest.raises(L.LoraHipError):
        L.MixedDetector([7, 13])
    with pytest.raises(L.LoraHipError):
        L.MixedDetector([])
    m = L.MixedDetector([7, 8])
    with pytest.raises(L.LoraHipError):                   # detect before plan
        m.detect(gpu.zeros(1 << 12, device="cuda"), out=dict(sym=gpu.zeros(2, dtype=gpu.int16, device="cuda"), power=gpu.zeros(2, device="cuda"),
                                                            powerAvg=gpu.zeros(2, device="cuda"), fIndex=gpu.zeros(2, device="cuda")))
    with pytest.raises(ValueError):
        m.plan([0], 4)
    with pytest.raises(L.LoraHipError):
        m.plan([0, -5], 4)
    m.close()


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_multi_device_scheduler_equals_single_device(gpu, devices):
    """lorahip_mixed_create_multi (one process, one host thread + streams per device) with every "device"
    being device 0 -- the one GPU a test box has: the channels are split by lorahip_shard_plan, every shard gets its OWN IQ buffer
    and result arrays (as separate devices would), the shards' launches are issued from their own host threads -- and every
    channel's results equal the single-device scheduler's over the same windows, bit for bit"""
    import lora_sdr_amd as L
    rng = np.random.default_rng(7 + len(devices))
    sfs = (7 + np.arange(41) % 6).astype(np.int32)
    S = 8
    buf, offsets, iq_of = build(rng, sfs, S)
    single = L.MixedDetector(sfs)
    single.plan(offsets, S)
    ref_out = {k: v.cpu().numpy() for k, v in single.detect(gpu.from_numpy(buf.view(np.float32)).cuda()).items()}
    multi = L.MixedDetectorMulti(sfs, devices)
    assert np.array_equal(multi.shard_of, L.shard_plan(sfs, len(devices))) and sum(multi.counts) == len(sfs)
    # per shard: its own buffer holding only its channels, back to back
    off = np.zeros(len(sfs), np.int64)
    iqs = []
    for s in range(len(devices)):
        mine = np.nonzero(multi.shard_of == s)[0]
        at, parts = 0, []
        for c in mine:
            off[c] = at
            parts.append(iq_of[c].reshape(-1))
            at += parts[-1].size
        iqs.append(gpu.from_numpy(np.concatenate(parts).view(np.float32)).cuda() if parts else None)
    multi.plan(off, S)
    outs = multi.detect(iqs)
    again = multi.detect(iqs)
    for s in range(len(devices)):
        assert all(gpu.equal(outs[s][k], again[s][k]) for k in ("sym", "power", "powerAvg", "fIndex"))
    for c in range(len(sfs)):
        s, r = int(multi.shard_of[c]), int(multi.rows[c])
        for k in ("sym", "power", "powerAvg", "fIndex"):
            assert np.array_equal(outs[s][k][r].cpu().numpy(), ref_out[k][int(single.rows[c])]), (c, k)
    # the single-device entry point refuses a multi-device object instead of reading the wrong buffer
    lib = L.load()
    z = gpu.zeros(16, device="cuda")
    assert lib.lorahip_mixed_detect(multi._h, z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr()) == -1
    assert lib.lorahip_mixed_num_devices(multi._h) == len(devices) and lib.lorahip_mixed_num_devices(single._h) == 1
    multi.close(); single.close()
