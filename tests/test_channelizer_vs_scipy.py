"""CPU: the channeliser's float64 definition (oracle/channelizer.py -- the reference has no such block)
against an independent implementation of the same published operations: frequency translation, FIR filtering by
scipy.signal.lfilter, decimation; and its low-pass design against scipy.signal.firwin with the same window. The GPU tests
(tests/test_gpu_channelizer.py) hold the fp32 kernel to oracle/channelizer.py; this file holds oracle/channelizer.py to scipy."""
import numpy as np
import pytest

scipy_signal = pytest.importorskip("scipy.signal")


@pytest.mark.parametrize("decim,n_taps", [(1, 1), (2, 17), (8, 64), (16, 129), (5, 33)])
def test_lowpass_design_is_firwin_blackmanharris(decim, n_taps):
    from oracle import channelizer as CH
    h = CH.design_lowpass(decim, n_taps).astype(np.float64)
    if n_taps == 1:
        assert h.tolist() == [1.0]
        return
    ref = scipy_signal.firwin(n_taps, cutoff=0.5 / decim, window="blackmanharris", pass_zero=True, scale=True, fs=1.0)
    assert np.abs(h - ref).max() < 2e-7 * np.abs(ref).max() + 1e-9          # the float32 rounding of the returned taps


@pytest.mark.parametrize("decim,n_taps,n", [(4, 31, 4096), (8, 64, 10000), (3, 10, 1000), (1, 5, 257)])
def test_channelize_is_mix_lfilter_decimate(decim, n_taps, n):
    from oracle import channelizer as CH
    rng = np.random.default_rng(decim * 100 + n_taps)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    taps = CH.design_lowpass(decim, n_taps)
    freqs = [0.0, 0.125, -0.2, 0.37, 1e-3]
    got = CH.channelize(x, freqs, decim, taps)
    t = np.arange(n, dtype=np.float64)
    for k, f in enumerate(freqs):
        # the oracle quantises the frequency to a 64-bit phase increment; at these lengths that moves the phase by < 1e-12 turns
        mixed = x.astype(np.complex128) * np.exp(-2j * np.pi * ((f * t) % 1.0))
        y = scipy_signal.lfilter(taps.astype(np.float64), [1.0], mixed)
        want = y[decim - 1::decim][: n // decim]
        assert got[k].shape == want.shape
        assert np.abs(got[k] - want).max() < 1e-9 * max(1.0, np.abs(want).max())


def test_phase_increment_wraps_like_a_frequency():
    from oracle import channelizer as CH
    assert CH.phase_inc(0.0) == 0 and CH.phase_inc(1.0) == 0 and CH.phase_inc(-1.0) == 0
    assert CH.phase_inc(0.25) == 1 << 62 and CH.phase_inc(-0.75) == 1 << 62 and CH.phase_inc(1.25) == 1 << 62
    assert CH.phase_inc(0.5) == 1 << 63
