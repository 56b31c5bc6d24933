"""GPU: the host-pointer entry points behind the pinned double-buffered upload (lorahip_upload.cpp): ordinary and pinned host
buffers, sizes around the 32 MiB staging buffers, many small pieces (one per channel) -- results must equal the device-pointer
path's, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sf,W", [(7, 1), (7, 4099), (9, 16385), (10, 9000)])     # 4 KiB ... 70 MiB of IQ: below, at and above the staging size
def test_host_batches_equal_device_batches(gpu, sf, W):
    import lora_sdr_amd as L
    rng = np.random.default_rng(sf * 1000 + W)
    N = 1 << sf
    iq = (rng.standard_normal((W, N)) + 1j * rng.standard_normal((W, N))).astype(np.complex64)
    iq += np.exp(2j * np.pi * rng.integers(0, N, (W, 1)) * np.arange(N)[None, :] / N).astype(np.complex64) * L.host_tables(sf, fine=False)[1][None, :]
    ctx = L.Context(sf)
    dev = ctx.detect_batch(gpu.from_numpy(iq).cuda())
    gpu.cuda.synchronize()
    pin = L.pinned_empty(iq.shape, iq.dtype)
    pin[...] = iq
    for buf in (iq, pin, iq[: max(1, W // 3)]):
        got = ctx.detect_batch(buf)
        n = buf.shape[0]
        assert np.array_equal(got["sym"], dev["sym"].cpu().numpy().view(np.uint16)[:n])
        for k in ("power", "powerAvg", "fIndex"):
            assert np.array_equal(got[k], dev[k].cpu().numpy()[:n], equal_nan=True), k
    ctx.close()


def test_demod_streams_from_host_memory_of_both_kinds(gpu, oracle):
    """lorahip_demod_run gathers one buffer per channel: ragged lengths, an empty channel, a pinned one among ordinary ones, and a
    total beyond one staging buffer"""
    import lora_sdr_amd as L
    from test_gpu_demod import frames
    rng = np.random.default_rng(77)
    sf, B = 8, 40
    streams, want = [], []
    for c in range(B):
        st, _ = frames(oracle, rng, sf, 2 + c % 3, 6, off=rng.uniform(-0.3, 0.3), noise=0.05, lead=int(rng.integers(0, 400)))
        if c == 5:
            st = st[:0]
        if c % 7 == 3:
            st = np.concatenate([st, (0.05 * (rng.standard_normal(1 << 20) + 1j * rng.standard_normal(1 << 20))).astype(np.complex64)])   # 8 MiB more
        if c % 4 == 1:
            p = L.pinned_empty(st.shape, st.dtype)
            p[...] = st
            st = p
        streams.append(st)
        want.append([p for _k, p in oracle.demod_run(sf, np.asarray(st), mtu=6, keep=False)["packets"]] if st.size else [])
    assert sum(s.nbytes for s in streams) > (48 << 20)
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(6)
    d.work(streams)
    got = {c: [] for c in range(B)}
    for c, _r, s in d.packets():
        got[c].append(s)
    for c in range(B):
        assert len(got[c]) == len(want[c]) and all(np.array_equal(a, b) for a, b in zip(got[c], want[c])), c
    d.close()


def test_demod_takes_one_2d_host_array(gpu, oracle):
    """(n_channels, samples) in host memory: per-channel pointers computed without a Python loop -- same packets as the list form"""
    import lora_sdr_amd as L
    from test_gpu_demod import frames
    rng = np.random.default_rng(5)
    sf, B = 7, 33
    rows = [frames(oracle, rng, sf, 2, 5, off=rng.uniform(-0.3, 0.3), noise=0.05, lead=100 + c)[0] for c in range(B)]
    n = min(r.size for r in rows)
    arr = np.stack([r[:n] for r in rows])
    a = L.LoRaDemod(sf, n_channels=B); a.set_mode(1); a.setMTU(5)
    a.work([arr[c] for c in range(B)])
    want = a.packets()
    b = L.LoRaDemod(sf, n_channels=B); b.set_mode(1); b.setMTU(5)
    b.work(arr)
    got = b.packets()
    assert len(got) == len(want) > 0
    for (c0, r0, s0), (c1, r1, s1) in zip(want, got):
        assert c0 == c1 and r0 == r1 and np.array_equal(s0, s1)
    with pytest.raises(ValueError):
        b.work(arr[:-1])
    a.close(); b.close()


def test_pinned_allocation_round_trip(gpu):
    import lora_sdr_amd as L
    a = L.pinned_empty((3, 5), np.float32)
    a[...] = np.arange(15, dtype=np.float32).reshape(3, 5)
    assert a.sum() == 105.0 and a.flags["C_CONTIGUOUS"]
    del a


@pytest.mark.parametrize("pad", [0, 96])
def test_rows_of_one_pinned_array_as_one_pointer_per_channel(gpu, oracle, pad):
    """lorahip_demod_run over per-channel pointers that are the rows of ONE pinned array -- touching (pad 0: one plain copy) or at a
    constant distance (a strided copy) -- instead of a copy and a pinned-memory query per piece; the packets of the list form in
    ordinary memory"""
    import lora_sdr_amd as L
    from test_gpu_demod import frames
    rng = np.random.default_rng(9 + pad)
    sf, B = 9, 24
    rows = [frames(oracle, rng, sf, 6, 7, off=rng.uniform(-0.3, 0.3), noise=0.05, lead=100 + 3 * c)[0] for c in range(B)]
    n = min(r.size for r in rows)
    assert B * n * 8 > (2 << 20)
    pin = L.pinned_empty((B, n + pad))
    pin[...] = 0
    for c in range(B):
        pin[c, :n] = rows[c][:n]
    a = L.LoRaDemod(sf, n_channels=B); a.set_mode(1); a.setMTU(7)
    a.work([np.array(rows[c][:n]) for c in range(B)])
    want = a.packets()
    b = L.LoRaDemod(sf, n_channels=B); b.set_mode(1); b.setMTU(7)
    b.work([pin[c, :n] for c in range(B)])
    got = b.packets()
    assert len(got) == len(want) >= 6 * B - B
    assert all(x[0] == y[0] and x[1] == y[1] and np.array_equal(x[2], y[2]) for x, y in zip(want, got))
    a.close(); b.close()
