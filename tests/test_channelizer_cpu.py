"""CPU: the channeliser's float64 definition (oracle/channelizer.py -- parity unpinned, the reference has no such block) behaves
like a mixer + decimating FIR, and the host-only part of the C ABI (phase increment) agrees with it. No compute calls."""
import ctypes as C
import math

import numpy as np

from oracle import channelizer as oc


def test_phase_increment_matches_library():
    import lora_sdr_amd
    lib = lora_sdr_amd.load()
    rng = np.random.default_rng(0)
    for f in [0.0, 0.1, -0.1, 0.25, -0.25, 0.5, -0.5, 1.0 / 3.0, 17.125, -3.999999, 1e-9] + list(rng.uniform(-2, 2, 200)):
        assert int(lib.lorahip_channelizer_phase_inc(float(f))) == oc.phase_inc(float(f))
    assert int(lib.lorahip_channelizer_phase_inc(float("nan"))) == 0
    # no context -> refused, never a silent CPU path
    h = C.c_void_p()
    taps = np.ones(4, np.float32); fr = np.zeros(1)
    assert lib.lorahip_channelizer_create(C.byref(h), None, 1, fr.ctypes.data, 2, taps.ctypes.data, 4) != 0
    assert not h.value
    assert lib.lorahip_channelizer_out_count(None, 100) == 0


def test_definition_is_mixer_plus_decimating_fir():
    rng = np.random.default_rng(1)
    D, L, n = 6, 30, 4000
    h = oc.design_lowpass(D, L)
    assert abs(float(h.sum()) - 1.0) < 1e-6
    f0 = 0.1875                                                    # exactly representable: phase_inc is exact
    tone = np.exp(2j * np.pi * f0 * np.arange(n))
    y = oc.channelize(tone, [f0, f0 + 0.3], D, h)
    assert y.shape == (2, n // D)
    assert np.abs(y[0, L // D + 1:] - h.astype(np.float64).sum()).max() < 1e-9   # on-channel tone -> DC with the filter's DC gain
    assert np.abs(y[1, L // D + 1:]).max() < 1e-3                  # off-channel tone -> stop band
    # linear, and causal with zero initial state: output m depends on samples <= (m+1)D-1 only
    a, b = rng.standard_normal(n) + 1j * rng.standard_normal(n), rng.standard_normal(n) + 1j * rng.standard_normal(n)
    ya, yb, yab = (oc.channelize(v, [0.05], D, h) for v in (a, b, 2 * a - 3j * b))
    assert np.abs(yab - (2 * ya - 3j * yb)).max() < 1e-9
    cut = 1000
    a2 = a.copy(); a2[cut * D:] = 0
    assert np.array_equal(oc.channelize(a2, [0.05], D, h)[:, :cut], ya[:, :cut])
    # explicit sum for a few outputs
    w = oc.phase_inc(0.05)
    for m in (0, 3, 77, 500):
        nm = (m + 1) * D - 1
        s = 0j
        for j in range(L):
            if nm - j >= 0:
                turns = ((w * (nm - j)) % (1 << 64)) / 2.0 ** 64
                s += float(h[j]) * a[nm - j] * np.exp(-2j * math.pi * turns)
        assert abs(s - ya[0, m]) < 1e-9
