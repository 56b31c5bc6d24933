"""TEST INFRASTRUCTURE ONLY -- float64 restatement of the channeliser's definition (include/lorahip.h).

PARITY UNPINNED: the reference has no channeliser ("not in reference; the step before the path" --
its example topologies put Pothos' /comms/rotate and a decimating FIR in front of each LoRaDemod, and PothosComms is not part
of /root/reference), so there are no golden vectors to pin this file against. It DEFINES what the kernel must compute, in
float64, and the GPU tests hold the fp32 kernel to it within a stated tolerance; the chunk-invariance and end-to-end
(channeliser -> demodulator recovers the sent symbols) properties do not depend on this file. What CAN be pinned is pinned:
tests/test_channelizer_vs_scipy.py holds this file to an independent implementation of the same textbook operations
(translate, scipy.signal.lfilter, decimate) and its filter design to scipy.signal.firwin with the same window.

    n_m    = (m + 1) D - 1
    y_k[m] = sum_{j<L} h[j] x[n_m - j] exp(-2 pi i frac(w_k (n_m - j) / 2^64)),   x[n<0] = 0,  w_k = floor(frac(f_k) 2^64)
"""
import math
import numpy as np


def phase_inc(freq):
    """64-bit phase increment of a frequency in cycles per input sample (same arithmetic as lorahip_channelizer_phase_inc)"""
    frac = freq - math.floor(freq)
    return 0 if frac >= 1.0 else int(math.ldexp(frac, 64))


def channelize(x, freqs, decim, taps):
    """x: complex wideband stream from sample 0; returns (K, len(x)//decim) complex128"""
    x = np.asarray(x, np.complex128)
    h = np.asarray(taps, np.float64)
    D = int(decim)
    n = np.arange(x.size, dtype=np.uint64)
    n_out = x.size // D
    out = np.empty((len(freqs), n_out), np.complex128)
    for k, f in enumerate(freqs):
        w = np.uint64(phase_inc(f))
        with np.errstate(over="ignore"):
            ph = (w * n).astype(np.int64)                        # wraps mod 2^64, read as signed: turns * 2^64 in [-0.5, 0.5)
        mixed = x * np.exp(-2j * np.pi * (ph.astype(np.float64) * 2.0 ** -64))
        y = np.convolve(mixed, h)[: x.size]                      # y[n] = sum_j h[j] mixed[n - j]
        out[k] = y[D - 1::D][:n_out]
    return out


def design_lowpass(decim, n_taps, cutoff=None):
    """windowed-sinc (Blackman-Harris) low-pass, unit DC gain; cutoff in cycles per input sample (default 0.5/decim)"""
    fc = 0.5 / decim if cutoff is None else cutoff
    t = np.arange(n_taps) - 0.5 * (n_taps - 1)
    a = 2.0 * np.pi * np.arange(n_taps) / max(n_taps - 1, 1)
    win = 0.35875 - 0.48829 * np.cos(a) + 0.14128 * np.cos(2 * a) - 0.01168 * np.cos(3 * a)
    h = 2.0 * fc * np.sinc(2.0 * fc * t) * (win if n_taps > 1 else 1.0)
    return (h / h.sum()).astype(np.float32)
