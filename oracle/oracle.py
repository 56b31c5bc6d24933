"""TEST INFRASTRUCTURE ONLY -- ctypes loaders for the two CPU checkers.

  * ``Oracle``  -> oracle/liblora_oracle.so   plain-C restatement (lora_oracle.c)
  * ``Ref``     -> oracle/_ref/libloraref.so  the real reference sources compiled in
                   place by oracle/Makefile (present only where /root/reference was)

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import
this module. The product package ``lora_sdr_amd`` never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liblora_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libloraref.so")
# timing-only builds of the same reference sources at the other flag sets BASELINE.md asks for (oracle/Makefile)
DROPIN_SO = os.path.join(HERE, "_ref", "libloradrop.so")    # the drop-in, compiled (oracle/Makefile): reference block + HIP detector, batch block
REF_VARIANTS = {"dropin": DROPIN_SO, "-O2": REF_SO, "-O3 -fcx-limited-range": os.path.join(HERE, "_ref", "libloraref_O3cx.so"),
                "-O3": os.path.join(HERE, "_ref", "libloraref_O3.so")}

_f32p = C.POINTER(C.c_float)
_i16p = C.POINTER(C.c_int16)
_u16p = C.POINTER(C.c_uint16)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)


def build(quiet=True):
    """(Re)build the checkers with oracle/Makefile (gcc/g++ only)."""
    subprocess.run(["make", "-C", HERE, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def _ptr(a, ty):
    return None if a is None else a.ctypes.data_as(ty)


def _cf(a):
    """complex64 array -> contiguous complex64"""
    return np.ascontiguousarray(a, dtype=np.complex64)


class WorkResult(C.Structure):
    _fields_ = [("consumed", C.c_int64), ("stateBefore", C.c_int32), ("value", C.c_int32),
                ("power", C.c_float), ("powerAvg", C.c_float), ("snr", C.c_float), ("fIndex", C.c_float),
                ("packetPosted", C.c_int32), ("packetLen", C.c_int32), ("signalsEmitted", C.c_int32),
                ("sigError", C.c_int32), ("sigPower", C.c_float), ("sigSnr", C.c_float),
                ("label", C.c_char * 48)]


class Oracle:
    """The plain-C restatement."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build()
        L = self.L = C.CDLL(ORACLE_SO)
        L.lo_fft_new.restype = C.c_void_p
        L.lo_fft_new.argtypes = [C.c_int]
        L.lo_fft_free.argtypes = [C.c_void_p]
        L.lo_fft_twiddles.restype = _f32p
        L.lo_fft_twiddles.argtypes = [C.c_void_p]
        L.lo_fft_stages.argtypes = [C.c_void_p, _i32p, _i32p]
        L.lo_fft_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.lo_detect.restype = C.c_size_t
        L.lo_detect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, _f32p, _f32p, _f32p]
        L.lo_demod_tables.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.lo_dechirp.restype = C.c_int
        L.lo_dechirp.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p]
        L.lo_detect_batch.argtypes = [C.c_int, C.c_void_p, C.c_size_t, _i64p, _i32p, _i32p, _f32p,
                                      _u16p, _f32p, _f32p, _f32p, _i32p, C.c_void_p, C.c_void_p, C.c_int]
        L.lo_genchirp.restype = C.c_int
        L.lo_genchirp.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, _f32p]
        L.lo_mod_frame.restype = C.c_size_t
        L.lo_mod_frame.argtypes = [C.c_int, C.c_ubyte, C.c_float, C.c_size_t, _u16p, C.c_size_t, C.c_void_p, _f32p]
        L.lo_mod_frame_len.restype = C.c_size_t
        L.lo_mod_frame_len.argtypes = [C.c_int, C.c_size_t, C.c_size_t]
        L.lo_demod_new.restype = C.c_void_p
        L.lo_demod_new.argtypes = [C.c_int]
        L.lo_demod_free.argtypes = [C.c_void_p]
        L.lo_demod_set_sync.argtypes = [C.c_void_p, C.c_ubyte]
        L.lo_demod_set_threshold.argtypes = [C.c_void_p, C.c_double]
        L.lo_demod_set_mtu.argtypes = [C.c_void_p, C.c_size_t]
        L.lo_demod_activate.argtypes = [C.c_void_p]
        L.lo_demod_work.restype = C.c_int
        L.lo_demod_work.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(WorkResult),
                                    C.c_void_p, C.c_void_p, _i16p]
        L.lo_demod_bench.restype = C.c_int64
        L.lo_demod_bench.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int]

    # -- kissfft ---------------------------------------------------------
    def twiddles(self, N):
        f = self.L.lo_fft_new(N)
        p = self.L.lo_fft_twiddles(f)
        tw = np.ctypeslib.as_array(p, shape=(2 * N,)).copy().view(np.complex64)
        self.L.lo_fft_free(f)
        return tw

    def stages(self, N):
        f = self.L.lo_fft_new(N)
        r = np.zeros(32, np.int32)
        m = np.zeros(32, np.int32)
        n = self.L.lo_fft_stages(f, _ptr(r, _i32p), _ptr(m, _i32p))
        self.L.lo_fft_free(f)
        return list(zip(r[:n].tolist(), m[:n].tolist()))

    def fft(self, x):
        x = _cf(x)
        f = self.L.lo_fft_new(x.size)
        out = np.empty_like(x)
        self.L.lo_fft_transform(f, x.ctypes.data, out.ctypes.data)
        self.L.lo_fft_free(f)
        return out

    def detect(self, x):
        """LoRaDetector feed+detect on one window -> (index, power, powerAvg, fIndex, fft)"""
        x = _cf(x)
        f = self.L.lo_fft_new(x.size)
        out = np.empty_like(x)
        p, pa, fi = C.c_float(), C.c_float(), C.c_float()
        idx = self.L.lo_detect(f, x.ctypes.data, out.ctypes.data, C.byref(p), C.byref(pa), C.byref(fi))
        self.L.lo_fft_free(f)
        return int(idx), p.value, pa.value, fi.value, out

    # -- demod -----------------------------------------------------------
    def tables(self, sf, fine=True):
        N = 1 << sf
        up = np.empty(N, np.complex64)
        down = np.empty(N, np.complex64)
        fi = np.empty(N * 128, np.complex64) if fine else None
        self.L.lo_demod_tables(sf, up.ctypes.data, down.ctypes.data, fi.ctypes.data if fine else None)
        return up, down, fi

    def dechirp(self, x, chirp, fine, idx, err):
        x = _cf(x)
        out = np.empty_like(x)
        idx1 = self.L.lo_dechirp(x.size, x.ctypes.data, _cf(chirp).ctypes.data, _cf(fine).ctypes.data,
                                 int(idx), float(err), out.ctypes.data)
        return out, idx1

    def detect_batch(self, sf, iq, n_windows=None, offsets=None, chirp_sel=None, fine_idx0=None,
                     fine_err=None, want_fft=False, want_dec=False, nthreads=1):
        N = 1 << sf
        iq = _cf(iq).reshape(-1)
        if offsets is not None:
            offsets = np.ascontiguousarray(offsets, np.int64)
            n = offsets.size
        else:
            n = iq.size // N if n_windows is None else n_windows
        cs = None if chirp_sel is None else np.ascontiguousarray(np.broadcast_to(chirp_sel, (n,)), np.int32)
        i0 = None if fine_idx0 is None else np.ascontiguousarray(np.broadcast_to(fine_idx0, (n,)), np.int32)
        fe = None if fine_err is None else np.ascontiguousarray(np.broadcast_to(fine_err, (n,)), np.float32)
        sym = np.empty(n, np.uint16)
        power = np.empty(n, np.float32)
        pavg = np.empty(n, np.float32)
        fidx = np.empty(n, np.float32)
        idx1 = np.empty(n, np.int32)
        fft = np.empty((n, N), np.complex64) if want_fft else None
        dec = np.empty((n, N), np.complex64) if want_dec else None
        self.L.lo_detect_batch(sf, iq.ctypes.data, n, _ptr(offsets, _i64p), _ptr(cs, _i32p), _ptr(i0, _i32p),
                               _ptr(fe, _f32p), _ptr(sym, _u16p), _ptr(power, _f32p), _ptr(pavg, _f32p),
                               _ptr(fidx, _f32p), _ptr(idx1, _i32p),
                               fft.ctypes.data if want_fft else None, dec.ctypes.data if want_dec else None,
                               int(nthreads))
        return dict(sym=sym, power=power, powerAvg=pavg, fIndex=fidx, fineIdxOut=idx1, fft=fft, dec=dec)

    # -- chirps / frames -------------------------------------------------
    def genchirp(self, N, ovs, NN, f0, down, ampl, phase):
        out = np.empty(NN, np.complex64)
        ph = C.c_float(phase)
        self.L.lo_genchirp(out.ctypes.data, N, ovs, NN, float(f0), int(bool(down)), float(ampl), C.byref(ph))
        return out, ph.value

    def mod_frame(self, sf, syms, sync=0x12, ampl=1.0, padding=1):
        syms = np.ascontiguousarray(syms, np.uint16)
        n = self.L.lo_mod_frame_len(sf, padding, syms.size)
        out = np.empty(n, np.complex64)
        ph = C.c_float(0)
        w = self.L.lo_mod_frame(sf, sync, float(ampl), padding, _ptr(syms, _u16p), syms.size, out.ctypes.data, C.byref(ph))
        assert w == n
        return out

    # -- the LoRaDecoder block (LoRaDecoder.cpp:196-397) ----------------
    def decode(self, sf, syms, ppm=0, cr="4/8", crcc=False, interleaving=True, error_check=False, explicit=True, hdr=False,
               data_length=8):
        """-> (bytes, or uint16 symbols when interleaving is off, or None if nothing was posted; dropped flag)"""
        syms = np.ascontiguousarray(syms, np.uint16)
        cfg = DecoderCfg(sf, ppm, CR_TO_RDD[cr], int(crcc), int(interleaving), int(error_check), int(explicit), int(hdr), data_length)
        out = np.zeros(4 * syms.size + 64, np.uint8)
        dropped = C.c_int(0)
        self.L.lo_decode.restype = C.c_long
        self.L.lo_decode.argtypes = [C.POINTER(DecoderCfg), _u16p, C.c_size_t, C.c_void_p, C.POINTER(C.c_int)]
        n = self.L.lo_decode(C.byref(cfg), _ptr(syms, _u16p), syms.size, out.ctypes.data, C.byref(dropped))
        if n < 0:
            return None, int(dropped.value)
        if not interleaving:
            return out[:2 * n].view(np.uint16).copy(), int(dropped.value)
        return out[:n].copy(), int(dropped.value)

    # -- the LoRaDemod block --------------------------------------------
    def demod_run(self, sf, iq, sync=0x12, thresh=-30.0, mtu=256, keep=True):
        """Run the restated block over a stream -> dict of per-call logs + packets"""
        N = 1 << sf
        iq = _cf(iq)
        d = self.L.lo_demod_new(sf)
        self.L.lo_demod_set_sync(d, sync)
        self.L.lo_demod_set_threshold(d, thresh)
        self.L.lo_demod_set_mtu(d, mtu)
        pos = 0
        res = WorkResult()
        dec = np.zeros(2 * N, np.complex64)
        fft = np.zeros(N, np.complex64)
        pkt = np.zeros(max(mtu, 1), np.int16)
        calls, packets, signals, ffts, decs = [], [], [], [], []
        while self.L.lo_demod_work(d, iq.ctypes.data + 8 * pos, iq.size - pos, C.byref(res),
                                   dec.ctypes.data, fft.ctypes.data, _ptr(pkt, _i16p)):
            calls.append(dict(consumed=res.consumed, state=res.stateBefore, value=res.value, power=res.power,
                              powerAvg=res.powerAvg, snr=res.snr, fIndex=res.fIndex, label=res.label.decode()))
            if keep:
                ffts.append(fft.copy())
                decs.append(dec.copy())
            if res.packetPosted:
                packets.append((len(calls) - 1, pkt[:res.packetLen].copy()))
            if res.signalsEmitted:
                signals.append((res.sigError, res.sigPower, res.sigSnr))
            pos += res.consumed
            if res.consumed == 0:
                break
        self.L.lo_demod_free(d)
        return dict(calls=calls, packets=packets, signals=signals, fft=ffts, dec=decs)

    def demod_bench(self, sf, iq, samples_per_stream, n_streams, nthreads, repeat=1):
        iq = _cf(iq)
        return int(self.L.lo_demod_bench(sf, iq.ctypes.data, samples_per_stream, n_streams, nthreads, repeat))

    def demod_run_many(self, sf, iq, sync=0x12, thresh=-30.0, mtu=256, nthreads=1, calls=False):
        """the restated block over every row of a (streams, samples) array: see run_many()"""
        return run_many(self.L.lo_demod_run_many, sf, iq, sync, thresh, mtu, nthreads, calls)


def run_many(fn, sf, iq, sync, thresh, mtu, nthreads, calls):
    """lo_demod_run_many / loraref_demod_run_many over a (streams, samples) complex64 array, every stream from the zero start
    state -> dict: n_calls (S,), n_packets (S,), pkt_lens (S, pktCap), pkt_call (S, pktCap), pkt_syms (S, symCap) with a stream's
    packets back to back, and with calls=True consumed (S, callCap) int32 and cls (S, callCap) uint8 per work() call (0 no label,
    1 SYNC, 2 P, 3 DC, 4 QC, 5 S<n>)."""
    iq = _cf(iq)
    assert iq.ndim == 2
    S, n = iq.shape
    N = 1 << sf
    # an unsquelched FRAMESYNC call consumes N - value (LoRaDemod.cpp:219), N/2 on average over noise: 4 calls per N samples is ample
    # (the C side reports an overflow instead of truncating)
    call_cap = 4 * (n // N) + 64 if calls else 1
    sym_cap = n // N + 8
    pkt_cap = n // (5 * N // 4) + 8
    fn.restype = C.c_int64
    fn.argtypes = [C.c_size_t if fn.__name__.startswith("loraref") else C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                   C.c_double, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                   C.c_void_p, C.c_void_p, C.c_size_t]
    n_calls = np.zeros(S, np.int32)
    n_packets = np.zeros(S, np.int32)
    pkt_syms = np.zeros((S, sym_cap), np.int16)
    pkt_lens = np.zeros((S, pkt_cap), np.int32)
    pkt_call = np.zeros((S, pkt_cap), np.int32)
    consumed = np.zeros((S, call_cap), np.int32) if calls else None
    cls = np.zeros((S, call_cap), np.uint8) if calls else None
    total = fn(sf, iq.ctypes.data, n, S, int(nthreads), int(sync), float(thresh), int(mtu), n_calls.ctypes.data, n_packets.ctypes.data,
               pkt_syms.ctypes.data, sym_cap, pkt_lens.ctypes.data, pkt_call.ctypes.data, pkt_cap,
               consumed.ctypes.data if calls else None, cls.ctypes.data if calls else None, call_cap)
    assert total >= 0, "run_many: a per-stream capacity was too small"
    return dict(total_calls=int(total), n_calls=n_calls, n_packets=n_packets, pkt_lens=pkt_lens, pkt_call=pkt_call, pkt_syms=pkt_syms,
                consumed=consumed, cls=cls)


CR_TO_RDD = {"4/4": 0, "4/5": 1, "4/6": 2, "4/7": 3, "4/8": 4}


class DecoderCfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("sf", "ppm", "rdd", "crcc", "interleaving", "error_check", "explicit_hdr", "hdr", "data_length")]


class Ref:
    """The real reference code (LoRaDetector.hpp, kissfft.hh, ChirpGenerator.hpp, LoRaDemod.cpp, LoRaMod.cpp)."""

    @staticmethod
    def available(flags="-O2"):
        return os.path.exists(REF_VARIANTS[flags])

    def __init__(self, flags="-O2"):
        L = self.L = C.CDLL(REF_VARIANTS[flags])
        L.loraref_detector_new.restype = C.c_void_p
        L.loraref_detector_new.argtypes = [C.c_size_t]
        L.loraref_detector_free.argtypes = [C.c_void_p]
        L.loraref_detector_detect.restype = C.c_size_t
        L.loraref_detector_detect.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, _f32p, _f32p, _f32p, C.c_void_p]
        L.loraref_detect_windows.argtypes = [C.c_size_t, C.c_void_p, C.c_size_t, _u16p, _f32p, _f32p, _f32p,
                                             C.c_void_p, C.c_int]
        L.loraref_genchirp.restype = C.c_int
        L.loraref_genchirp.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, _f32p]
        L.loraref_decode.restype = C.c_long
        L.loraref_decode.argtypes = [C.c_size_t, C.c_size_t, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t,
                                     _u16p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_ulonglong)]
        L.loraref_decode_bench.restype = C.c_double
        L.loraref_decode_bench.argtypes = [C.c_size_t, C.c_size_t, C.c_char_p, C.c_int, C.c_int, _u16p, C.c_size_t, C.c_size_t]
        L.loraref_encode.restype = C.c_long
        L.loraref_encode.argtypes = [C.c_size_t, C.c_size_t, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                     _u16p, C.c_size_t]
        L.loraref_mod_frame.restype = C.c_size_t
        L.loraref_mod_frame.argtypes = [C.c_size_t, C.c_int, C.c_float, C.c_size_t, _u16p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.loraref_demod_new.restype = C.c_void_p
        L.loraref_demod_new.argtypes = [C.c_size_t, C.c_int]
        L.loraref_demod_free.argtypes = [C.c_void_p]
        L.loraref_demod_set.restype = C.c_int
        L.loraref_demod_set.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        L.loraref_demod_run.restype = C.c_int64
        L.loraref_demod_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        for n in ("num_calls", "num_packets", "num_signals"):
            getattr(L, "loraref_demod_" + n).restype = C.c_size_t
            getattr(L, "loraref_demod_" + n).argtypes = [C.c_void_p]
        L.loraref_demod_get_calls.argtypes = [C.c_void_p, _i64p, C.c_void_p, C.c_void_p]
        L.loraref_demod_get_label.restype = C.c_size_t
        L.loraref_demod_get_label.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.loraref_demod_packet_len.restype = C.c_size_t
        L.loraref_demod_packet_len.argtypes = [C.c_void_p, C.c_size_t]
        L.loraref_demod_packet_call.restype = C.c_int64
        L.loraref_demod_packet_call.argtypes = [C.c_void_p, C.c_size_t]
        L.loraref_demod_get_packet.argtypes = [C.c_void_p, C.c_size_t, _i16p]
        L.loraref_demod_get_signal.restype = C.c_double
        L.loraref_demod_get_signal.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.loraref_demod_bench.restype = C.c_int64
        L.loraref_demod_bench.argtypes = [C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int]

    # -- the codec blocks (LoRaEncoder.cpp / LoRaDecoder.cpp, verbatim) --
    def encode(self, sf, data, ppm=0, cr="4/8", explicit=True, crc=True, whitening=True):
        data = np.ascontiguousarray(data, np.uint8)
        out = np.zeros(16 + 4 * (data.size + 8), np.uint16)
        n = self.L.loraref_encode(sf, ppm, cr.encode(), int(explicit), int(crc), int(whitening), data.ctypes.data, data.size,
                                  _ptr(out, _u16p), out.size)
        assert 0 <= n <= out.size
        return out[:n].copy()

    def decode_bench(self, sf, syms, reps, ppm=0, cr="4/8", crcc=True, error_check=False):
        """seconds the verbatim block needs for `reps` messages (one host core)"""
        syms = np.ascontiguousarray(syms, np.uint16)
        return float(self.L.loraref_decode_bench(sf, ppm, cr.encode(), int(crcc), int(error_check), _ptr(syms, _u16p), syms.size, reps))

    def decode(self, sf, syms, ppm=0, cr="4/8", crcc=False, interleaving=True, error_check=False, explicit=True, hdr=False,
               data_length=8):
        """-> (bytes or None if nothing was posted, dropped count)"""
        syms = np.ascontiguousarray(syms, np.uint16)
        out = np.zeros(4 * syms.size + 64, np.uint8)
        dropped = C.c_ulonglong(0)
        n = self.L.loraref_decode(sf, ppm, cr.encode(), int(crcc), int(interleaving), int(error_check), int(explicit), int(hdr),
                                  data_length, _ptr(syms, _u16p), syms.size, out.ctypes.data, out.size, C.byref(dropped))
        if n < 0:
            return None, int(dropped.value)
        if not interleaving:
            return out[:2 * n].view(np.uint16).copy(), int(dropped.value)
        return out[:n].copy(), int(dropped.value)

    def mod_frame(self, sf, syms, sync=0x12, ampl=1.0, padding=1):
        """the verbatim LoRaMod block (LoRaMod.cpp:109-238): one packet of symbols -> its frame"""
        syms = np.ascontiguousarray(syms, np.uint16)
        cap = (1 << sf) * (syms.size + padding + 32)
        out = np.empty(cap, np.complex64)
        n = self.L.loraref_mod_frame(sf, sync, float(ampl), padding, _ptr(syms, _u16p), syms.size, out.ctypes.data, cap)
        assert n > 0, "reference modulator produced nothing"
        return out[:n].copy()

    def detect(self, x):
        x = _cf(x)
        d = self.L.loraref_detector_new(x.size)
        out = np.empty_like(x)
        p, pa, fi = C.c_float(), C.c_float(), C.c_float()
        idx = self.L.loraref_detector_detect(d, x.size, x.ctypes.data, C.byref(p), C.byref(pa), C.byref(fi),
                                             out.ctypes.data)
        self.L.loraref_detector_free(d)
        return int(idx), p.value, pa.value, fi.value, out

    def detect_windows(self, N, iq, want_fft=False, nthreads=1):
        iq = _cf(iq).reshape(-1)
        n = iq.size // N
        sym = np.empty(n, np.uint16)
        power = np.empty(n, np.float32)
        pavg = np.empty(n, np.float32)
        fidx = np.empty(n, np.float32)
        fft = np.empty((n, N), np.complex64) if want_fft else None
        self.L.loraref_detect_windows(N, iq.ctypes.data, n, _ptr(sym, _u16p), _ptr(power, _f32p),
                                      _ptr(pavg, _f32p), _ptr(fidx, _f32p),
                                      fft.ctypes.data if want_fft else None, int(nthreads))
        return dict(sym=sym, power=power, powerAvg=pavg, fIndex=fidx, fft=fft)

    def genchirp(self, N, ovs, NN, f0, down, ampl, phase):
        out = np.empty(NN, np.complex64)
        ph = C.c_float(phase)
        self.L.loraref_genchirp(out.ctypes.data, N, ovs, NN, float(f0), int(bool(down)), float(ampl), C.byref(ph))
        return out, ph.value

    def demod_run(self, sf, iq, sync=0x12, thresh=-30.0, mtu=256):
        N = 1 << sf
        iq = _cf(iq)
        h = self.L.loraref_demod_new(sf, 1)
        assert h, "registry has no /lora/lora_demod"
        self.L.loraref_demod_set(h, b"setSync", float(sync))
        self.L.loraref_demod_set(h, b"setThreshold", float(thresh))
        self.L.loraref_demod_set(h, b"setMTU", float(mtu))
        self.L.loraref_demod_run(h, iq.ctypes.data, iq.size)
        n = self.L.loraref_demod_num_calls(h)
        consumed = np.zeros(n, np.int64)
        fft = np.zeros((n, N), np.complex64)
        dec = np.zeros((n, 2 * N), np.complex64)
        self.L.loraref_demod_get_calls(h, _ptr(consumed, _i64p), fft.ctypes.data, dec.ctypes.data)
        buf = C.create_string_buffer(64)
        labels = []
        for i in range(n):
            self.L.loraref_demod_get_label(h, i, buf, 64)
            labels.append(buf.value.decode())
        packets = []
        for i in range(self.L.loraref_demod_num_packets(h)):
            ln = self.L.loraref_demod_packet_len(h, i)
            p = np.zeros(ln, np.int16)
            self.L.loraref_demod_get_packet(h, i, _ptr(p, _i16p))
            packets.append((int(self.L.loraref_demod_packet_call(h, i)), p))
        signals = []
        for i in range(self.L.loraref_demod_num_signals(h)):
            v = self.L.loraref_demod_get_signal(h, i, buf, 64)
            signals.append((buf.value.decode(), v))
        self.L.loraref_demod_free(h)
        return dict(consumed=consumed, labels=labels, fft=fft, dec=dec, packets=packets, signals=signals)

    def demod_bench(self, sf, iq, samples_per_stream, n_streams, nthreads, repeat=1):
        iq = _cf(iq)
        return int(self.L.loraref_demod_bench(sf, iq.ctypes.data, samples_per_stream, n_streams, nthreads, repeat))

    def demod_run_many(self, sf, iq, sync=0x12, thresh=-30.0, mtu=256, nthreads=1, calls=False):
        """the verbatim LoRaDemod.cpp, one fresh block per row of a (streams, samples) array: see run_many()"""
        return run_many(self.L.loraref_demod_run_many, sf, iq, sync, thresh, mtu, nthreads, calls)


class DropInBatch:
    """lora_sdr_amd/pothos/LoRaDemodBatch.cpp (the multi-channel Pothos block of INTEGRATION.md section 2) compiled against the fake
    Pothos and linked with liblorahip.so: oracle/_ref/libloradrop.so, driven by oracle/dropin_driver.cpp."""

    @staticmethod
    def available():
        return os.path.exists(DROPIN_SO)

    def __init__(self, sf, channels, max_windows=64):
        L = self.L = C.CDLL(DROPIN_SO)
        L.loradrop_batch_new.restype = C.c_void_p
        L.loradrop_batch_new.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t]
        L.loradrop_batch_free.argtypes = [C.c_void_p]
        L.loradrop_batch_set.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        L.loradrop_batch_set_string.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.loradrop_batch_run.restype = C.c_int64
        L.loradrop_batch_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.loradrop_batch_bench.restype = C.c_int
        L.loradrop_batch_bench.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_double)]
        L.loradrop_batch_count.restype = C.c_size_t
        L.loradrop_batch_count.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p]
        L.loradrop_batch_get_stream.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_void_p]
        L.loradrop_batch_get_label.restype = C.c_size_t
        L.loradrop_batch_get_label.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.loradrop_batch_packet_len.restype = C.c_size_t
        L.loradrop_batch_packet_len.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
        L.loradrop_batch_get_packet.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, _i16p]
        L.loradrop_batch_use_input_slabs.argtypes = [C.c_void_p, C.c_int]
        L.loradrop_batch_input_slabs_active.argtypes = [C.c_void_p]
        L.loradrop_batch_num_signals.restype = C.c_size_t
        L.loradrop_batch_num_signals.argtypes = [C.c_void_p]
        L.loradrop_batch_get.restype = C.c_double
        L.loradrop_batch_get.argtypes = [C.c_void_p, C.c_char_p]
        L.loradrop_batch_get_signal.restype = C.c_double
        L.loradrop_batch_get_signal.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
        self.sf, self.B = sf, channels
        self.h = L.loradrop_batch_new(sf, channels, max_windows)
        if not self.h:
            raise RuntimeError("LoRaDemodBatch could not be created (no gfx950 device, or the block is not registered)")

    def close(self):
        if getattr(self, "h", None):
            self.L.loradrop_batch_free(self.h)
            self.h = None

    __del__ = close

    def set(self, name, v, may_fail=False):
        """a registered call with a numeric / bool argument; with may_fail the code is returned (0, -1 unknown call, -2 the block threw)"""
        rc = int(self.L.loradrop_batch_set(self.h, name.encode(), float(v)))
        if not may_fail:
            assert rc == 0, (name, rc)
        return rc

    def use_input_slabs(self, on=True):
        """before the first run: the inputs arrive through the block's own input buffer managers (getInputBufferManager: pinned slabs)
        instead of as views into the caller's array"""
        assert self.L.loradrop_batch_use_input_slabs(self.h, int(bool(on))) == 0

    def input_slabs_active(self):
        return bool(self.L.loradrop_batch_input_slabs_active(self.h))

    def get(self, name):
        """a registered getter of the block (slabRowRuns, workRuns, fftFramesDropped)"""
        v = float(self.L.loradrop_batch_get(self.h, name.encode()))
        assert v >= 0, name
        return int(v)

    def bench(self, iq, chunk):
        """the block as a receiver, timed (oracle/dropin_driver.cpp::loradrop_batch_bench): iq (channels, samples) complex64 in ordinary
        host memory arrives `chunk` samples per channel at a time. -> dict(seconds, works, packets, consumed, signals)"""
        iq = np.ascontiguousarray(iq, np.complex64)
        assert iq.shape[0] == self.B
        out = (C.c_double * 5)()
        rc = self.L.loradrop_batch_bench(self.h, iq.ctypes.data, iq.shape[1], int(chunk), out)
        if rc != 0:
            raise RuntimeError("LoRaDemodBatch::work() threw")
        return dict(seconds=out[0], works=int(out[1]), packets=int(out[2]), consumed=int(out[3]), signals=int(out[4]))

    def set_string(self, name, v):
        """a registered call with a string argument (setDevices); returns 0, -1 unknown call, -2 the block threw"""
        return int(self.L.loradrop_batch_set_string(self.h, name.encode(), str(v).encode()))

    def run(self, iq):
        """iq: (channels, samples) complex64 -> per-channel dicts: consumed, raw / dec / fft streams, labels per port as
        [(element index, id)], packets; plus the block's signal log [(name, value)]"""
        iq = np.ascontiguousarray(iq, np.complex64)
        assert iq.shape[0] == self.B
        works = self.L.loradrop_batch_run(self.h, iq.ctypes.data, iq.shape[1])
        buf = C.create_string_buffer(64)
        out = []
        for c in range(self.B):
            d = {"consumed": self.L.loradrop_batch_count(self.h, c, b"consumed")}
            for port in ("raw", "dec", "fft"):
                a = np.empty(self.L.loradrop_batch_count(self.h, c, port.encode()), np.complex64)
                self.L.loradrop_batch_get_stream(self.h, c, port.encode(), a.ctypes.data)
                d[port] = a
                labs = []
                for i in range(self.L.loradrop_batch_count(self.h, c, (port + "Labels").encode())):
                    idx = self.L.loradrop_batch_get_label(self.h, c, port.encode(), i, buf, 64)
                    labs.append((idx, buf.value.decode()))
                d[port + "_labels"] = labs
            pk = []
            for i in range(self.L.loradrop_batch_count(self.h, c, b"packets")):
                p = np.zeros(self.L.loradrop_batch_packet_len(self.h, c, i), np.int16)
                self.L.loradrop_batch_get_packet(self.h, c, i, _ptr(p, _i16p))
                pk.append(p)
            d["packets"] = pk
            out.append(d)
        sig = []
        for i in range(self.L.loradrop_batch_num_signals(self.h)):
            v = self.L.loradrop_batch_get_signal(self.h, i, buf, 64)
            sig.append((buf.value.decode(), v))
        return out, sig, works


class DropInDecoder:
    """lora_sdr_amd/pothos/LoRaDecoderBatch.cpp (/lora/lora_decoder_batch) compiled against the fake Pothos and linked with
    liblorahip.so (oracle/_ref/libloradrop.so), driven by oracle/dropin_driver.cpp: symbol packets onto its inputs, one work(),
    the byte packets and the values of the "dropped" signal back."""

    @staticmethod
    def available():
        return os.path.exists(DROPIN_SO)

    def __init__(self, channels):
        L = self.L = C.CDLL(DROPIN_SO)
        L.loradrop_decoder_new.restype = C.c_void_p
        L.loradrop_decoder_new.argtypes = [C.c_size_t]
        L.loradrop_decoder_free.argtypes = [C.c_void_p]
        L.loradrop_decoder_set.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        L.loradrop_decoder_set_string.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.loradrop_decoder_activate.argtypes = [C.c_void_p]
        L.loradrop_decoder_push.argtypes = [C.c_void_p, C.c_size_t, _u16p, C.c_size_t]
        L.loradrop_decoder_work.argtypes = [C.c_void_p]
        L.loradrop_decoder_num_out.restype = C.c_size_t
        L.loradrop_decoder_num_out.argtypes = [C.c_void_p, C.c_size_t]
        L.loradrop_decoder_out_len.restype = C.c_size_t
        L.loradrop_decoder_out_len.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
        L.loradrop_decoder_get_out.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
        L.loradrop_decoder_num_dropped_signals.restype = C.c_size_t
        L.loradrop_decoder_num_dropped_signals.argtypes = [C.c_void_p]
        L.loradrop_decoder_dropped_signal.restype = C.c_double
        L.loradrop_decoder_dropped_signal.argtypes = [C.c_void_p, C.c_size_t]
        self.B = channels
        self.h = L.loradrop_decoder_new(channels)
        if not self.h:
            raise RuntimeError("LoRaDecoderBatch could not be created (the block is not registered)")

    def close(self):
        if getattr(self, "h", None):
            self.L.loradrop_decoder_free(self.h)
            self.h = None

    __del__ = close

    def configure(self, sf, ppm=0, cr="4/8", crcc=False, interleaving=True, error_check=False, explicit=True, hdr=False, data_length=8):
        for name, v in (("setSpreadFactor", sf), ("setSymbolSize", ppm), ("enableCrcc", crcc), ("enableInterleaving", interleaving), ("enableErrorCheck", error_check),
                        ("enableExplicit", explicit), ("enableHdr", hdr), ("setDataLength", data_length)):
            assert self.L.loradrop_decoder_set(self.h, name.encode(), float(v)) == 0, name
        return self.L.loradrop_decoder_set_string(self.h, b"setCodingRate", cr.encode())

    def activate(self):
        return int(self.L.loradrop_decoder_activate(self.h))

    def push(self, channel, syms):
        syms = np.ascontiguousarray(syms, np.uint16)
        self.L.loradrop_decoder_push(self.h, channel, _ptr(syms, _u16p), syms.size)

    def work(self):
        """0, or -2 if the block threw"""
        return int(self.L.loradrop_decoder_work(self.h))

    def outputs(self, channel, interleaving=True):
        res = []
        for i in range(self.L.loradrop_decoder_num_out(self.h, channel)):
            a = np.zeros(self.L.loradrop_decoder_out_len(self.h, channel, i), np.uint8)
            self.L.loradrop_decoder_get_out(self.h, channel, i, a.ctypes.data)
            res.append(a if interleaving else a.view(np.uint16))
        return res

    def dropped_signals(self):
        return [int(self.L.loradrop_decoder_dropped_signal(self.h, i)) for i in range(self.L.loradrop_decoder_num_dropped_signals(self.h))]
