"""Host-side mirror of the reference's operator interface for the demod hot path, on top of
the C ABI (include/lorahip.h). Names and argument meaning follow the reference:

  LoRaDetector(N).feed(i, samp) / .detect()      LoRaDetector.hpp:8-72
  LoRaDemod(sf).setSync/.setThreshold/.setMTU    LoRaDemod.cpp:119-137   (factory /lora/lora_demod)
  LoRaDemod.activate() / .work(...)              LoRaDemod.cpp:139-327
  Context(sf).detect_batch(...)                  the batched form of LoRaDemod.cpp:157-172

torch is used for device memory and streams only.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Batch, DecoderCfg, WorkResult, check, load, CHIRP_UP, CHIRP_DOWN, CHIRP_NONE  # noqa: F401


# lorahip_work_result as a numpy record (include/lorahip.h)
WORK_RESULT_DTYPE = np.dtype([("consumed", np.int64), ("state_before", np.int32), ("value", np.int32), ("power", np.float32),
                              ("power_avg", np.float32), ("snr", np.float32), ("f_index", np.float32), ("worked", np.int32),
                              ("packet_len", np.int32), ("signals", np.int32), ("sig_error", np.int32), ("sig_power", np.float32),
                              ("sig_snr", np.float32), ("fine_idx_before", np.int32), ("fine_idx_after", np.int32),
                              ("fine_err_before", np.float32), ("reserved", np.int32)])


def host_tables(sf, fine=True):
    """(up, down, fine, twiddle) exactly as the reference builds them -- pure host code."""
    lib = load()
    N = 1 << sf
    up = np.empty(N, np.complex64)
    down = np.empty(N, np.complex64)
    fi = np.empty(N * _lib.FINE_STEPS, np.complex64) if fine else None
    tw = np.empty(N, np.complex64)
    check(lib.lorahip_host_tables(sf, up.ctypes.data, down.ctypes.data, fi.ctypes.data if fine else None,
                                  tw.ctypes.data), "lorahip_host_tables")
    return up, down, fi, tw


def device_count():
    return load().lorahip_device_count()


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _dptr(t, dtype=None):
    """device pointer of a torch tensor (must be contiguous, on a HIP device)"""
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError("expected a device tensor")
    if not t.is_contiguous():
        raise ValueError("expected a contiguous tensor")
    if dtype is not None and t.dtype != dtype:
        raise ValueError("expected dtype %s, got %s" % (dtype, t.dtype))
    return t.data_ptr()


class Context:
    """One (device, SF) batch context: chirp / fine-tune / twiddle tables resident in HBM."""

    def __init__(self, sf, device=0):
        self._lib = load()
        self._h = C.c_void_p()
        check(self._lib.lorahip_create(C.byref(self._h), int(device), int(sf)), "lorahip_create")
        self.sf = int(sf)
        self.N = 1 << self.sf
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.lorahip_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_stream(self, stream_handle):
        """stream_handle: integer hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); 0/None is
        HIP's null stream, which is torch's default stream"""
        check(self._lib.lorahip_set_stream(self._h, C.c_void_p(stream_handle or 0)), "lorahip_set_stream")

    def use_torch_stream(self):
        import torch
        self.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def set_variant(self, v):
        check(self._lib.lorahip_set_variant(self._h, int(v)), "lorahip_set_variant")

    def synchronize(self):
        check(self._lib.lorahip_synchronize(self._h), "lorahip_synchronize")

    def set_fine_gather(self, on):
        """A/B switch: read the fine-tune table in HBM instead of evaluating it from the split tables (include/lorahip.h)"""
        check(self._lib.lorahip_set_fine_gather(self._h, int(bool(on))), "lorahip_set_fine_gather")

    def fine_split_active(self):
        return bool(self._lib.lorahip_fine_split_active(self._h))

    def timer_start(self):
        check(self._lib.lorahip_timer_start(self._h), "lorahip_timer_start")

    def timer_stop(self):
        ms = C.c_float()
        check(self._lib.lorahip_timer_stop(self._h, C.byref(ms)), "lorahip_timer_stop")
        return ms.value

    # ------------------------------------------------------------------
    def detect_batch_raw(self, batch):
        """Launch with a prepared struct lorahip_batch of DEVICE pointers (asynchronous)."""
        check(self._lib.lorahip_detect_batch(self._h, C.byref(batch)), "lorahip_detect_batch")

    def make_batch(self, iq, n_windows, sym, power, power_avg, f_index, offsets=None, window_stride=0,
                   chirp_sel=None, chirp_sel_all=CHIRP_UP, fine_idx0=None, fine_err=None, fine_idx_out=None,
                   fft_out=None, dec_out=None):
        """Build a struct lorahip_batch from torch device tensors (kept alive by the caller)."""
        import torch
        b = Batch()
        b.struct_size = C.sizeof(Batch)
        b.iq = _dptr(iq)
        b.n_windows = int(n_windows)
        b.offsets = _dptr(offsets, torch.int64)
        b.window_stride = int(window_stride)
        b.chirp_sel = _dptr(chirp_sel, torch.int32)
        b.chirp_sel_all = int(chirp_sel_all)
        b.fine_idx0 = _dptr(fine_idx0, torch.int32)
        b.fine_err = _dptr(fine_err, torch.float32)
        b.sym = _dptr(sym)
        b.power = _dptr(power, torch.float32)
        b.power_avg = _dptr(power_avg, torch.float32)
        b.f_index = _dptr(f_index, torch.float32)
        b.fine_idx_out = _dptr(fine_idx_out, torch.int32)
        b.fft_out = _dptr(fft_out)
        b.dec_out = _dptr(dec_out)
        return b

    def detect_batch(self, iq, n_windows=None, offsets=None, window_stride=0, chirp_sel=None,
                     chirp_sel_all=CHIRP_UP, fine_idx0=None, fine_err=None, want_fft=False, want_dec=False,
                     want_fine_idx=False):
        """Demodulate a batch of independent windows.

        iq: torch complex64 device tensor (flat stream) or numpy complex64 array (host; staged
        through the context). Returns a dict of arrays of the same kind as the input:
        sym (uint16 as int16-viewed tensor for torch), power, powerAvg, fIndex [, fineIdxOut, fft, dec].
        """
        if _is_torch(iq):
            return self._detect_torch(iq, n_windows, offsets, window_stride, chirp_sel, chirp_sel_all,
                                      fine_idx0, fine_err, want_fft, want_dec, want_fine_idx)
        return self._detect_numpy(iq, n_windows, offsets, window_stride, chirp_sel, chirp_sel_all,
                                  fine_idx0, fine_err, want_fft, want_dec, want_fine_idx)

    def _count(self, n_samples, n_windows, offsets, window_stride):
        if offsets is not None:
            return int(offsets.shape[0])
        if n_windows is not None:
            return int(n_windows)
        stride = window_stride or self.N
        return 0 if n_samples < self.N else (n_samples - self.N) // stride + 1

    def _detect_torch(self, iq, n_windows, offsets, window_stride, chirp_sel, chirp_sel_all, fine_idx0, fine_err,
                      want_fft, want_dec, want_fine_idx):
        import torch
        iq = iq.reshape(-1)
        dev = iq.device
        W = self._count(iq.numel(), n_windows, offsets, window_stride)
        out = dict(sym=torch.empty(W, dtype=torch.int16, device=dev),       # uint16 payload
                   power=torch.empty(W, dtype=torch.float32, device=dev),
                   powerAvg=torch.empty(W, dtype=torch.float32, device=dev),
                   fIndex=torch.empty(W, dtype=torch.float32, device=dev))
        if want_fine_idx:
            out["fineIdxOut"] = torch.empty(W, dtype=torch.int32, device=dev)
        if want_fft:
            out["fft"] = torch.empty((W, self.N), dtype=torch.complex64, device=dev)
        if want_dec:
            out["dec"] = torch.empty((W, self.N), dtype=torch.complex64, device=dev)
        b = self.make_batch(iq, W, out["sym"], out["power"], out["powerAvg"], out["fIndex"], offsets=offsets,
                            window_stride=window_stride, chirp_sel=chirp_sel, chirp_sel_all=chirp_sel_all,
                            fine_idx0=fine_idx0, fine_err=fine_err, fine_idx_out=out.get("fineIdxOut"),
                            fft_out=out.get("fft"), dec_out=out.get("dec"))
        self.use_torch_stream()
        self.detect_batch_raw(b)
        return out

    def _detect_numpy(self, iq, n_windows, offsets, window_stride, chirp_sel, chirp_sel_all, fine_idx0, fine_err,
                      want_fft, want_dec, want_fine_idx):
        iq = np.ascontiguousarray(iq, np.complex64).reshape(-1)
        if offsets is not None:
            offsets = np.ascontiguousarray(offsets, np.int64)
        W = self._count(iq.size, n_windows, offsets, window_stride)
        cs = None if chirp_sel is None else np.ascontiguousarray(np.broadcast_to(chirp_sel, (W,)), np.int32)
        i0 = None if fine_idx0 is None else np.ascontiguousarray(np.broadcast_to(fine_idx0, (W,)), np.int32)
        fe = None if fine_err is None else np.ascontiguousarray(np.broadcast_to(fine_err, (W,)), np.float32)
        out = dict(sym=np.empty(W, np.uint16), power=np.empty(W, np.float32), powerAvg=np.empty(W, np.float32),
                   fIndex=np.empty(W, np.float32))
        if want_fine_idx:
            out["fineIdxOut"] = np.empty(W, np.int32)
        if want_fft:
            out["fft"] = np.empty((W, self.N), np.complex64)
        if want_dec:
            out["dec"] = np.empty((W, self.N), np.complex64)
        need = (int(offsets.max()) + self.N) if (offsets is not None and W) else ((W - 1) * (window_stride or self.N) + self.N if W else 0)
        if need > iq.size:
            raise ValueError("iq holds %d samples, the batch reads up to %d" % (iq.size, need))

        def p(a):
            return None if a is None else a.ctypes.data
        b = Batch()
        b.struct_size = C.sizeof(Batch)
        b.iq = p(iq)
        b.n_windows = W
        b.offsets = p(offsets)
        b.window_stride = int(window_stride)
        b.chirp_sel = p(cs)
        b.chirp_sel_all = int(chirp_sel_all)
        b.fine_idx0 = p(i0)
        b.fine_err = p(fe)
        b.sym = p(out["sym"]); b.power = p(out["power"]); b.power_avg = p(out["powerAvg"]); b.f_index = p(out["fIndex"])
        b.fine_idx_out = p(out.get("fineIdxOut"))
        b.fft_out = p(out.get("fft"))
        b.dec_out = p(out.get("dec"))
        check(self._lib.lorahip_detect_batch_host(self._h, C.byref(b)), "lorahip_detect_batch_host")
        return out

    def synth_symbols(self, sym, ampl=1.0, noise_sigma=0.0, seed=0):
        """IQ of len(sym) back-to-back up-chirp symbols generated in HBM -> complex64 device tensor."""
        import torch
        sym = sym.reshape(-1)
        if sym.dtype not in (torch.int16, torch.uint16):
            raise ValueError("sym must be a 16-bit integer device tensor")
        iq = torch.empty(sym.numel() * self.N, dtype=torch.complex64, device=sym.device)
        self.use_torch_stream()
        check(self._lib.lorahip_synth_symbols(self._h, _dptr(iq), _dptr(sym), sym.numel(), float(ampl),
                                              float(noise_sigma), int(seed) & (2 ** 64 - 1)), "lorahip_synth_symbols")
        return iq


    # ------------------------------------------------------------------
    def mod_frame_len(self, nsyms, padding=1):
        return int(self._lib.lorahip_mod_frame_len(self.sf, int(nsyms), int(padding)))

    def mod_frames(self, syms, sync=0x12, ampl=1.0, padding=1, frame_stride=None, lead=0, tail=0):
        """The LoRaMod block (LoRaMod.cpp:109-238) for a batch of packets: syms is a (n_frames, nsyms) 16-bit device
        tensor; returns a (n_frames, lead + frame_stride + tail) complex64 device tensor, zero outside the frames."""
        import torch
        if syms.dim() != 2 or syms.dtype not in (torch.int16, torch.uint16):
            raise ValueError("syms must be a (n_frames, nsyms) 16-bit integer device tensor")
        syms = syms.contiguous()
        F, S = int(syms.shape[0]), int(syms.shape[1])
        flen = self.mod_frame_len(S, padding)
        stride = int(frame_stride or flen)
        if stride < flen:
            raise ValueError("frame_stride %d is shorter than the frame (%d samples)" % (stride, flen))
        row = lead + stride + tail
        iq = torch.zeros((F, row), dtype=torch.complex64, device=syms.device)
        self.use_torch_stream()
        base = iq.data_ptr() + 8 * lead
        check(self._lib.lorahip_mod_frames(self._h, C.c_void_p(base), row, _dptr(syms), F, S, int(sync) & 0xff, float(ampl),
                                           int(padding)), "lorahip_mod_frames")
        return iq

    def add_awgn(self, iq, sigma, seed=0):
        """complex AWGN (per-component sigma) added in place to a complex64 device tensor"""
        self.use_torch_stream()
        check(self._lib.lorahip_add_awgn(self._h, _dptr(iq), iq.numel(), float(sigma), int(seed) & (2 ** 64 - 1)), "lorahip_add_awgn")
        return iq


def pinned_empty(shape, dtype=np.complex64):
    """numpy array in pinned host memory (lorahip_host_alloc): the host-pointer entry points -- Context.detect_batch_host,
    LoRaDemod.work on host streams -- hand such buffers to the DMA engine directly instead of staging them first"""
    import weakref
    lib = load()
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    ptr = lib.lorahip_host_alloc(max(n, 1))
    if not ptr:
        raise MemoryError("lorahip_host_alloc(%d)" % n)
    buf = (C.c_char * max(n, 1)).from_address(ptr)
    weakref.finalize(buf, lib.lorahip_host_free, C.c_void_p(ptr))
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


class MixedDetector:
    """Channels of different spreading factors demodulated in one call (lorahip_mixed_*: buckets by SF, one stream per bucket,
    concurrent launches, event join -- all below Python). channel_sf: SF of every channel; plan(offsets, S): channel c's S
    back-to-back windows start at sample offsets[c] of the IQ buffer. detect(iq) -> dict of (rows, S) device tensors in
    bucket-major order; rows[c] is channel c's row."""

    def __init__(self, channel_sf, device=0):
        self._lib = load()
        self._h = C.c_void_p()
        sf = np.ascontiguousarray(channel_sf, np.int32).reshape(-1)
        check(self._lib.lorahip_mixed_create(C.byref(self._h), int(device), sf.ctypes.data, sf.size), "lorahip_mixed_create")
        self.n_channels, self.device, self.S = int(sf.size), int(device), 0
        self.rows = np.empty(sf.size, np.int64)
        check(self._lib.lorahip_mixed_rows(self._h, self.rows.ctypes.data), "lorahip_mixed_rows")
        self.buckets = []
        for i in range(self._lib.lorahip_mixed_num_buckets(self._h)):
            s_, r_, n_ = C.c_int32(), C.c_size_t(), C.c_size_t()
            check(self._lib.lorahip_mixed_bucket(self._h, i, C.byref(s_), C.byref(r_), C.byref(n_)), "lorahip_mixed_bucket")
            self.buckets.append((s_.value, r_.value, n_.value))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.lorahip_mixed_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def set_variant(self, variant):
        for i in range(len(self.buckets)):
            check(self._lib.lorahip_set_variant(C.c_void_p(self._lib.lorahip_mixed_context(self._h, i)), int(variant)), "lorahip_set_variant")

    def plan(self, channel_offset, windows_per_channel):
        off = np.ascontiguousarray(channel_offset, np.int64).reshape(-1)
        if off.size != self.n_channels:
            raise ValueError("one offset per channel")
        check(self._lib.lorahip_mixed_plan(self._h, off.ctypes.data, int(windows_per_channel)), "lorahip_mixed_plan")
        self.S = int(windows_per_channel)

    def new_outputs(self):
        import torch
        dev = torch.device("cuda", self.device)
        shape = (self.n_channels, self.S)
        return dict(sym=torch.empty(shape, dtype=torch.int16, device=dev), power=torch.empty(shape, dtype=torch.float32, device=dev),
                    powerAvg=torch.empty(shape, dtype=torch.float32, device=dev), fIndex=torch.empty(shape, dtype=torch.float32, device=dev))

    def detect(self, iq, out=None, sync=True):
        """asynchronous unless sync: the buckets run on their own streams; the caller's stream must have finished producing iq"""
        import torch
        out = out or self.new_outputs()
        # the buckets launch on private non-blocking streams: whatever torch's current stream still has queued -- the producer of
        # `iq`, the previous owner of the recycled output blocks -- must be finished first (as LoRaDemod.packets_device does)
        torch.cuda.current_stream(self.device).synchronize()
        check(self._lib.lorahip_mixed_detect(self._h, _dptr(iq), _dptr(out["sym"]), _dptr(out["power"]), _dptr(out["powerAvg"]), _dptr(out["fIndex"])),
              "lorahip_mixed_detect")
        if sync:
            self.synchronize()
        return out

    def synchronize(self):
        check(self._lib.lorahip_mixed_synchronize(self._h), "lorahip_mixed_synchronize")


def shard_plan(channel_sf, n_shards):
    """lorahip_shard_plan (host-only): the shard of every channel under the library's byte-weighted rule -- the C twin of
    lora_sdr_amd.shard.shard_channels, used by lorahip_mixed_create_multi"""
    lib = load()
    sf = np.ascontiguousarray(channel_sf, np.int32).reshape(-1)
    out = np.zeros(sf.size, np.int32)
    check(lib.lorahip_shard_plan(sf.ctypes.data if sf.size else None, sf.size, int(n_shards), out.ctypes.data if sf.size else None), "lorahip_shard_plan")
    return out


class MixedDetectorMulti:
    """lorahip_mixed_create_multi: mixed-SF channels over several devices of ONE process (one scheduler, host thread and set of
    streams per device, no data-path collective). devices: list of device indices (may repeat). shard_of[c] = index into devices of
    channel c, rows[c] = its row in that device's result arrays, channels_of(s) = shard s's channels in row order per bucket."""

    def __init__(self, channel_sf, devices):
        self._lib = load()
        self._h = C.c_void_p()
        sf = np.ascontiguousarray(channel_sf, np.int32).reshape(-1)
        dv = np.ascontiguousarray(devices, np.int32).reshape(-1)
        check(self._lib.lorahip_mixed_create_multi(C.byref(self._h), dv.ctypes.data, dv.size, sf.ctypes.data, sf.size), "lorahip_mixed_create_multi")
        self.n_channels, self.devices, self.S = int(sf.size), [int(d) for d in dv], 0
        self.shard_of = np.empty(sf.size, np.int32)
        check(self._lib.lorahip_mixed_shard_of(self._h, self.shard_of.ctypes.data), "lorahip_mixed_shard_of")
        self.rows = np.empty(sf.size, np.int64)
        check(self._lib.lorahip_mixed_rows(self._h, self.rows.ctypes.data), "lorahip_mixed_rows")
        self.counts = []
        for s in range(len(self.devices)):
            d_, n_ = C.c_int32(), C.c_size_t()
            check(self._lib.lorahip_mixed_device(self._h, s, C.byref(d_), C.byref(n_)), "lorahip_mixed_device")
            self.counts.append(int(n_.value))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.lorahip_mixed_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def plan(self, channel_offset, windows_per_channel):
        """channel_offset[c]: first sample of channel c inside the IQ buffer of ITS device"""
        off = np.ascontiguousarray(channel_offset, np.int64).reshape(-1)
        if off.size != self.n_channels:
            raise ValueError("one offset per channel")
        check(self._lib.lorahip_mixed_plan(self._h, off.ctypes.data, int(windows_per_channel)), "lorahip_mixed_plan")
        self.S = int(windows_per_channel)

    def new_outputs(self):
        import torch
        outs = []
        for s, dev_i in enumerate(self.devices):
            dev = torch.device("cuda", dev_i)
            shape = (self.counts[s], self.S)
            outs.append(dict(sym=torch.empty(shape, dtype=torch.int16, device=dev), power=torch.empty(shape, dtype=torch.float32, device=dev),
                             powerAvg=torch.empty(shape, dtype=torch.float32, device=dev), fIndex=torch.empty(shape, dtype=torch.float32, device=dev)))
        return outs

    def detect(self, iqs, outs=None, sync=True):
        """iqs: one device tensor per shard (None where a shard has no channel); returns the per-shard output dicts"""
        import torch
        outs = outs or self.new_outputs()
        for dev_i in set(self.devices):
            torch.cuda.synchronize(dev_i)                           # inputs and recycled output blocks are ready (the shards run on private streams)
        n = len(self.devices)

        def arr(ts):
            return (C.c_void_p * n)(*[(t.data_ptr() if t is not None and t.numel() else None) for t in ts])
        check(self._lib.lorahip_mixed_detect_multi(self._h, arr(iqs), arr([o["sym"] for o in outs]), arr([o["power"] for o in outs]),
                                                   arr([o["powerAvg"] for o in outs]), arr([o["fIndex"] for o in outs])), "lorahip_mixed_detect_multi")
        if sync:
            check(self._lib.lorahip_mixed_synchronize(self._h), "lorahip_mixed_synchronize")
        return outs


def design_lowpass(decim, n_taps, cutoff=None):
    """Windowed-sinc (Blackman-Harris) low-pass prototype for the channeliser, unit DC gain; cutoff in cycles per input
    sample (default 0.5/decim = half the channel rate)."""
    fc = 0.5 / decim if cutoff is None else float(cutoff)
    t = np.arange(n_taps) - 0.5 * (n_taps - 1)
    a = 2.0 * np.pi * np.arange(n_taps) / max(n_taps - 1, 1)
    win = 0.35875 - 0.48829 * np.cos(a) + 0.14128 * np.cos(2 * a) - 0.01168 * np.cos(3 * a)
    h = 2.0 * fc * np.sinc(2.0 * fc * t) * (win if n_taps > 1 else 1.0)
    return (h / h.sum()).astype(np.float32)


class Channelizer:
    """K channels out of one wideband complex64 stream: mix each centre frequency (cycles per input sample) to 0, low-pass,
    keep every decim-th sample; output (K, n_out) in the layout LoRaDemod.work() takes. Stateful: consecutive run() calls
    continue one stream (filter history and mixer phase carried), reset() starts a new one. See include/lorahip.h."""

    def __init__(self, ctx, freqs, decim, taps):
        self._lib = load()
        self._ctx = ctx                                                  # borrowed: device and stream
        self._h = C.c_void_p()
        f = np.ascontiguousarray(freqs, np.float64).reshape(-1)
        t = np.ascontiguousarray(taps, np.float32).reshape(-1)
        check(self._lib.lorahip_channelizer_create(C.byref(self._h), ctx._h, f.size, f.ctypes.data, int(decim), t.ctypes.data, t.size),
              "lorahip_channelizer_create")
        self.n_channels, self.decim, self.n_taps = int(f.size), int(decim), int(t.size)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.lorahip_channelizer_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def reset(self):
        check(self._lib.lorahip_channelizer_reset(self._h), "lorahip_channelizer_reset")

    def out_count(self, n_in):
        return int(self._lib.lorahip_channelizer_out_count(self._h, int(n_in)))

    def run(self, wide, out=None):
        """wide: 1-D complex64 device tensor (the next samples of the stream); returns the (K, n_out) complex64 tensor"""
        import torch
        if wide.dim() != 1 or wide.dtype != torch.complex64:
            raise ValueError("wide must be a 1-D complex64 device tensor")
        wide = wide.contiguous()
        n_out = self.out_count(wide.numel())
        if out is None:
            out = torch.empty((self.n_channels, n_out), dtype=torch.complex64, device=wide.device)
        elif (out.dim() != 2 or out.shape[0] != self.n_channels or out.shape[1] < n_out or out.dtype != torch.complex64
              or (out.numel() and (out.stride(1) != 1 or out.stride(0) < out.shape[1]))):
            raise ValueError("out must be a (K, >= n_out) complex64 tensor with unit column stride (rows may be a slice of a wider buffer)")
        self._ctx.use_torch_stream()
        got = C.c_size_t()
        # the row stride is the tensor's own: `out` may be the columns [w, w + n) of a (K, capacity) buffer that fills chunk by chunk
        check(self._lib.lorahip_channelizer_run(self._h, C.c_void_p(wide.data_ptr()) if wide.numel() else None, wide.numel(),
                                                C.c_void_p(out.data_ptr()) if out.numel() else None,
                                                int(out.stride(0)) if out.numel() and out.shape[0] > 1 else int(out.shape[1]), C.byref(got)),
              "lorahip_channelizer_run")
        return out[:, :got.value]


    def run_captures(self, wide):
        """wide: (S, n_in) complex64 device tensor of S independent captures -> (S, K, n_in // decim) tensor, one launch; every
        capture is processed like a fresh stream, the object's own stream state is left alone"""
        import torch
        if wide.dim() != 2 or wide.dtype != torch.complex64:
            raise ValueError("wide must be a (captures, samples) complex64 device tensor")
        wide = wide.contiguous()
        S, n_in = int(wide.shape[0]), int(wide.shape[1])
        n_out = n_in // self.decim
        out = torch.empty((S, self.n_channels, n_out), dtype=torch.complex64, device=wide.device)
        self._ctx.use_torch_stream()
        got = C.c_size_t()
        check(self._lib.lorahip_channelizer_run_captures(self._h, C.c_void_p(wide.data_ptr()) if wide.numel() else None, S, n_in, n_in,
                                                         C.c_void_p(out.data_ptr()) if out.numel() else None, n_out, C.byref(got)),
              "lorahip_channelizer_run_captures")
        return out


class LoRaDetector:
    """`LoRaDetector<float>` (LoRaDetector.hpp:8-72): feed N samples, detect() -> arg-max bin."""

    def __init__(self, N, device=0):
        self._lib = load()
        self._h = C.c_void_p()
        check(self._lib.lorahip_detector_create(C.byref(self._h), int(device), int(N)), "lorahip_detector_create")
        self.N = int(N)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.lorahip_detector_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def feed(self, i, samp):
        samp = complex(samp)
        check(self._lib.lorahip_detector_feed(self._h, int(i), samp.real, samp.imag), "lorahip_detector_feed")

    def detect(self, fft_output=None):
        """returns (index, power, powerAvg, fIndex); fills fft_output (complex64[N]) if given"""
        idx = C.c_size_t()
        p, pa, fi = C.c_float(), C.c_float(), C.c_float()
        out = None
        if fft_output is not None:
            if fft_output.dtype != np.complex64 or fft_output.size != self.N or not fft_output.flags.c_contiguous:
                raise ValueError("fft_output must be a contiguous complex64[N] array")
            out = fft_output.ctypes.data
        check(self._lib.lorahip_detector_detect(self._h, C.byref(idx), C.byref(p), C.byref(pa), C.byref(fi), out),
              "lorahip_detector_detect")
        return idx.value, p.value, pa.value, fi.value


class LoRaDemod:
    """B channels of the `/lora/lora_demod` block (LoRaDemod.cpp), same parameters and defaults."""

    STATES = ("FRAMESYNC", "DOWNCHIRP0", "DOWNCHIRP1", "QUARTERCHIRP", "DATASYMBOLS")

    def __init__(self, sf=10, n_channels=1, device=0, channel_sf=None, devices=None):
        """channel_sf (one SF per channel) and devices (a list of device indices) make the mixed-SF / multi-device form
        (lorahip_demod_create_mixed): one handle, global channel numbers; sf / n_channels / device are ignored then"""
        self._lib = load()
        self._h = C.c_void_p()
        self._port_bufs = dict(fft=None, dec=None, raw=None)
        self._mtu = 256                                                 # LoRaDemod.cpp:73
        if channel_sf is None:
            check(self._lib.lorahip_demod_create(C.byref(self._h), int(device), int(sf), int(n_channels)),
                  "lorahip_demod_create")
            self.sf, self.N, self.n_channels = int(sf), 1 << int(sf), int(n_channels)
            self._device = int(device)
            self.channel_sf, self.devices = None, [int(device)]
            return
        csf = np.ascontiguousarray(channel_sf, np.int32).reshape(-1)
        dv = np.ascontiguousarray([0] if devices is None else devices, np.int32).reshape(-1)
        check(self._lib.lorahip_demod_create_mixed(C.byref(self._h), dv.ctypes.data, dv.size, csf.ctypes.data, csf.size), "lorahip_demod_create_mixed")
        self.channel_sf, self.devices, self.n_channels = csf.copy(), [int(x) for x in dv], int(csf.size)
        self.sf = int(csf[0]) if (csf == csf[0]).all() else None
        self.N = None if self.sf is None else 1 << self.sf
        self._device = int(dv[0])
        self.part_of = np.empty(csf.size, np.int32)
        self.local_of = np.empty(csf.size, np.int32)
        check(self._lib.lorahip_demod_part_of(self._h, self.part_of.ctypes.data, self.local_of.ctypes.data), "lorahip_demod_part_of")
        self.parts = []                                                 # (device, sf, n_channels, device_slot)
        for i in range(self._lib.lorahip_demod_num_parts(self._h)):
            d_, s_, n_, k_ = C.c_int32(), C.c_int32(), C.c_size_t(), C.c_int32()
            check(self._lib.lorahip_demod_part(self._h, i, C.byref(d_), C.byref(s_), C.byref(n_), C.byref(k_)), "lorahip_demod_part")
            self.parts.append((d_.value, s_.value, n_.value, k_.value))

    @staticmethod
    def make(sf):
        return LoRaDemod(sf)

    def device_slot_of(self):
        """mixed form: for every channel the index into `devices` of the GPU that holds it"""
        return np.array([self.parts[p][3] for p in self.part_of], np.int32)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.lorahip_demod_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def setSync(self, sync):
        check(self._lib.lorahip_demod_set_sync(self._h, int(sync) & 0xff), "lorahip_demod_set_sync")

    def setThreshold(self, thresh_dB):
        check(self._lib.lorahip_demod_set_threshold(self._h, float(thresh_dB)), "lorahip_demod_set_threshold")

    def setMTU(self, mtu):
        check(self._lib.lorahip_demod_set_mtu(self._h, int(mtu)), "lorahip_demod_set_mtu")
        self._mtu = int(mtu)

    def activate(self):
        check(self._lib.lorahip_demod_activate(self._h), "lorahip_demod_activate")

    def set_mode(self, mode):
        """0 auto, 1 streaming kernel (frame machine on the device), 2 host-driven lock-step rounds"""
        check(self._lib.lorahip_demod_set_mode(self._h, int(mode)), "lorahip_demod_set_mode")

    def set_variant(self, variant):
        """kernel variant of the host-driven mode's batch launches; the contracted build (40) is refused at this level"""
        check(self._lib.lorahip_demod_set_variant(self._h, int(variant)), "lorahip_demod_set_variant")

    def set_stream_grid(self, max_workgroups):
        """the streaming kernels' grid (scheduling only, same results): 0 the library's choice, < 0 one workgroup per channel set
        always, n > 0 at most n workgroups each walking several channels (SF11 / SF12)"""
        check(self._lib.lorahip_demod_set_stream_grid(self._h, int(max_workgroups)), "lorahip_demod_set_stream_grid")

    def set_stream_lanes(self, log2_lanes):
        """lanes per channel of the streaming kernels at SF7-9 (scheduling only, same results): 0 by channel count (default), < 0
        always 16 points per lane, 4 / 5 / 6 = 16 / 32 / 64 lanes per channel where the build holds that instance; 16 | l = two groups
        of 2^l lanes per channel, the second one a window ahead of the call (lorahip.h)"""
        check(self._lib.lorahip_demod_set_stream_lanes(self._h, int(log2_lanes)), "lorahip_demod_set_stream_lanes")

    def part_stream_lanes(self):
        """mixed form: log2 of the lanes per channel each part's streaming launches run on, in the order of `parts` (a part counts the
        wavefronts its sibling parts put on the same device: lorahip.h, lorahip_demod_create_mixed)"""
        out = []
        for i in range(len(self.parts)):
            h = self._lib.lorahip_demod_part_handle(self._h, i)
            v = int(self._lib.lorahip_demod_stream_lanes(C.c_void_p(h)))
            if v < 0:
                raise _lib.LoraHipError(v, "lorahip_demod_stream_lanes")
            out.append(v)
        return out

    def stream_lanes(self):
        """log2 of the lanes per channel the streaming launches of this object run on"""
        v = int(self._lib.lorahip_demod_stream_lanes(self._h))
        if v < 0:
            raise _lib.LoraHipError(v, "lorahip_demod_stream_lanes")
        return v

    def set_record_capacity(self, max_calls_per_launch):
        """bound the streaming kernels' per-launch record capacity (0: the library's own sizing): a channel that fills it is resumed by
        another launch, results unchanged -- the tests of the resume path"""
        check(self._lib.lorahip_demod_set_record_capacity(self._h, int(max_calls_per_launch)), "lorahip_demod_set_record_capacity")

    def set_trace(self, on=True):
        check(self._lib.lorahip_demod_set_trace(self._h, int(bool(on))), "lorahip_demod_set_trace")

    def work_segments(self, buf, first_sample, n_samples):
        """lorahip_demod_run_device_segments: channel c's stream is buf.view(-1)[first_sample[c] : first_sample[c] + n_samples[c]] of ONE
        torch complex64 device tensor -- the running receiver behind a channeliser, which advances first_sample[c] by consumed(c)
        and re-presents the remainder with the new samples, without copying anything. Returns the number of lock-step rounds."""
        import torch
        if not _is_torch(buf) or buf.dtype != torch.complex64 or not buf.is_contiguous():
            raise ValueError("expected a contiguous complex64 device tensor")
        first = np.ascontiguousarray(first_sample, np.int64)
        cnt = np.ascontiguousarray(n_samples, np.uint64)
        if first.shape != (self.n_channels,) or cnt.shape != (self.n_channels,):
            raise ValueError("first_sample and n_samples need one entry per channel")
        if cnt.size and int((first + cnt.astype(np.int64)).max()) > buf.numel():
            raise ValueError("a segment ends beyond the buffer")
        rounds = C.c_int64()
        if self.channel_sf is not None:
            # an object of several (device, SF) parts: each part launches on its OWN stream so that they overlap on the device; one
            # common stream would run them one after the other. What torch's stream still has queued -- the producer of `buf` -- ends first.
            torch.cuda.current_stream(buf.device).synchronize()
            check(self._lib.lorahip_demod_run_device_segments(self._h, _dptr(buf), first.ctypes.data_as(C.POINTER(C.c_int64)),
                                                              cnt.ctypes.data_as(C.POINTER(C.c_size_t)), C.byref(rounds)), "lorahip_demod_run_device_segments")
            return rounds.value
        check(self._lib.lorahip_demod_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(buf.device).cuda_stream)), "lorahip_demod_set_stream")
        try:
            check(self._lib.lorahip_demod_run_device_segments(self._h, _dptr(buf), first.ctypes.data_as(C.POINTER(C.c_int64)),
                                                              cnt.ctypes.data_as(C.POINTER(C.c_size_t)), C.byref(rounds)), "lorahip_demod_run_device_segments")
        finally:
            self._lib.lorahip_demod_reset_stream(self._h)
        return rounds.value

    def work_host_rows(self, rows, first_sample, n_samples):
        """lorahip_demod_run_host_rows: `rows` is a (n_channels, row_stride) complex64 HOST array (pinned_empty() for a straight DMA);
        channel c's stream is rows[c, first_sample[c] : first_sample[c] + n_samples[c]]. One strided copy, then the run."""
        if not isinstance(rows, np.ndarray) or rows.dtype != np.complex64 or rows.ndim != 2 or rows.shape[0] != self.n_channels or not rows.flags.c_contiguous:
            raise ValueError("expected a C-contiguous (n_channels, row_stride) complex64 host array")
        first = np.ascontiguousarray(first_sample, np.int64)
        cnt = np.ascontiguousarray(n_samples, np.uint64)
        if first.shape != (self.n_channels,) or cnt.shape != (self.n_channels,):
            raise ValueError("first_sample and n_samples need one entry per channel")
        rounds = C.c_int64()
        check(self._lib.lorahip_demod_run_host_rows(self._h, rows.ctypes.data, int(rows.shape[1]), first.ctypes.data, cnt.ctypes.data, C.byref(rounds)),
              "lorahip_demod_run_host_rows")
        return rounds.value

    def work_segments_multi(self, bufs, first_sample, n_samples):
        """lorahip_demod_run_device_segments_multi: bufs[s] is the complex64 device tensor on devices[s] that holds the segments of the
        channels of device slot s (None for a slot without channels)"""
        first = np.ascontiguousarray(first_sample, np.int64)
        cnt = np.ascontiguousarray(n_samples, np.uint64)
        if first.shape != (self.n_channels,) or cnt.shape != (self.n_channels,):
            raise ValueError("first_sample and n_samples need one entry per channel")
        import torch
        for b in bufs:
            if b is not None:
                torch.cuda.synchronize(b.device)                    # the parts launch on private streams
        ptrs = (C.c_void_p * len(bufs))(*[(b.data_ptr() if b is not None and b.numel() else None) for b in bufs])
        rounds = C.c_int64()
        check(self._lib.lorahip_demod_run_device_segments_multi(self._h, ptrs, len(bufs), first.ctypes.data_as(C.POINTER(C.c_int64)),
                                                                cnt.ctypes.data_as(C.POINTER(C.c_size_t)), C.byref(rounds)),
              "lorahip_demod_run_device_segments_multi")
        return rounds.value

    def set_signals(self, on=True):
        """keep the block's error / power / snr signals (LoRaDemod.cpp:267-269) without a per-call trace"""
        check(self._lib.lorahip_demod_set_signals(self._h, int(bool(on))), "lorahip_demod_set_signals")

    def signals(self):
        """the queued signal emissions as arrays: channel, round, error, power, snr (cleared with the packets)"""
        n = int(self._lib.lorahip_demod_num_signals(self._h))
        ch, rd, er = np.empty(n, np.int32), np.empty(n, np.int64), np.empty(n, np.int32)
        pw, sn = np.empty(n, np.float32), np.empty(n, np.float32)
        check(self._lib.lorahip_demod_get_signals(self._h, ch.ctypes.data, rd.ctypes.data, er.ctypes.data, pw.ctypes.data, sn.ctypes.data, n),
              "lorahip_demod_get_signals")
        return ch, rd, er, pw, sn

    def rewind(self):
        check(self._lib.lorahip_demod_rewind(self._h), "lorahip_demod_rewind")

    def work_append(self, buf, n_valid):
        """lorahip_demod_run_device_append: buf is a (n_channels, capacity) complex64 device tensor of which the first n_valid columns are
        valid; every channel continues at its own read position"""
        import torch
        if not _is_torch(buf) or buf.dim() != 2 or buf.shape[0] != self.n_channels or buf.dtype != torch.complex64 or not buf.is_contiguous():
            raise ValueError("expected a contiguous (n_channels, capacity) complex64 device tensor")
        rounds = C.c_int64()
        check(self._lib.lorahip_demod_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(buf.device).cuda_stream)), "lorahip_demod_set_stream")
        try:
            check(self._lib.lorahip_demod_run_device_append(self._h, _dptr(buf), int(buf.shape[1]), int(n_valid), C.byref(rounds)), "lorahip_demod_run_device_append")
        finally:
            self._lib.lorahip_demod_reset_stream(self._h)
        return rounds.value

    def receiver_rows(self, cap_packets, stride=None):
        """device tensors for receive(): (cap_packets, stride) int16 symbols, (cap_packets,) int32 lengths and channels"""
        import torch
        dev = torch.device("cuda", int(self._device))
        if stride is None:
            stride = max(8, min(self._mtu, int(self._lib.lorahip_decode_max_symbols())))      # no packet is longer than the MTU (LoRaDemod.cpp:291)
        return (torch.empty((int(cap_packets), int(stride)), dtype=torch.int16, device=dev), torch.empty(int(cap_packets), dtype=torch.int32, device=dev),
                torch.empty(int(cap_packets), dtype=torch.int32, device=dev))

    def receiver_signal_rows(self, cap, pinned_host=False):
        """lorahip_demod_receive_signal_rows: with set_signals(True), every receive() / receive_flush() delivers the block's signals
        (error / power / snr once per packet at DOWNCHIRP1, LoRaDemod.cpp:267-269) of the step whose packets it delivers into these
        rows, last_signals() of them. Returns the tensors (channel int32, error int32, power float32, snr float32): device memory, or --
        pinned_host -- numpy arrays over pinned host memory the device writes directly (read them after the stream has passed).
        cap = 0 unregisters (receiver steps drop the signals again)."""
        import torch
        r = _lib.SignalRows()
        r.struct_size = C.sizeof(_lib.SignalRows)
        cap = int(cap)
        if cap == 0:
            self._sig_rows = None
            check(self._lib.lorahip_demod_receive_signal_rows(self._h, None), "lorahip_demod_receive_signal_rows")
            return None
        if pinned_host:
            rows = (pinned_empty((cap,), np.int32), pinned_empty((cap,), np.int32), pinned_empty((cap,), np.float32), pinned_empty((cap,), np.float32))
            r.channel, r.error, r.power, r.snr = (a.ctypes.data for a in rows)
        else:
            dev = torch.device("cuda", int(self._device))
            rows = (torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.int32, device=dev),
                    torch.empty(cap, dtype=torch.float32, device=dev), torch.empty(cap, dtype=torch.float32, device=dev))
            r.channel, r.error, r.power, r.snr = (a.data_ptr() for a in rows)
        r.cap = cap
        check(self._lib.lorahip_demod_receive_signal_rows(self._h, C.byref(r)), "lorahip_demod_receive_signal_rows")
        self._sig_rows = rows                                # (kept alive while registered)
        return rows

    def register_signal_rows(self, rows):
        """register a set made by receiver_signal_rows() again: a caller that alternates two sets of packet rows (pipelined consumers, the
        resident receiver) alternates two sets of signal rows the same way, one registration before each receive()"""
        r = _lib.SignalRows()
        r.struct_size = C.sizeof(_lib.SignalRows)
        if isinstance(rows[0], np.ndarray):
            r.channel, r.error, r.power, r.snr = (a.ctypes.data for a in rows)
        else:
            r.channel, r.error, r.power, r.snr = (a.data_ptr() for a in rows)
        r.cap = int(rows[0].shape[0])
        check(self._lib.lorahip_demod_receive_signal_rows(self._h, C.byref(r)), "lorahip_demod_receive_signal_rows")
        self._sig_rows = rows

    def last_steps(self):
        """the resident steps the last receive() / receive_flush() reported, oldest first: [(packets, signals)] (a flush at depth > 1
        reports several, each in the rows of its own call)"""
        pk, sg = (C.c_size_t * 4)(), (C.c_size_t * 4)()
        n = int(self._lib.lorahip_demod_receive_steps(self._h, pk, sg, 4))
        return [(int(pk[i]), int(sg[i])) for i in range(n)]

    def resident_active(self):
        """True while the resident kernel of receive(async_=3) is on the device"""
        return bool(self._lib.lorahip_demod_resident_active(self._h))

    def last_signals(self):
        """signals the last receive() / receive_flush() delivered into the registered signal rows"""
        return int(self._lib.lorahip_demod_receive_num_signals(self._h))

    def _rows_struct(self, rows, async_, depth=0):
        syms, nsyms, chan = rows
        r = _lib.PacketRows()
        r.struct_size = C.sizeof(_lib.PacketRows)
        r.syms_dev, r.sym_stride, r.nsyms_dev, r.channel_dev = syms.data_ptr(), int(syms.shape[1]), nsyms.data_ptr(), chan.data_ptr()
        r.cap_packets, r.async_, r.reserved = int(syms.shape[0]), int(async_), int(depth)
        return r

    def receive(self, buf, n_valid, rows, async_=True, order_with_torch=True, depth=1):
        """lorahip_demod_receive: one receiver step in one call into the library -- the append run, the completed packets packed into
        `rows` (receiver_rows()) on the device, the queue cleared. Returns (n_packets, work_calls). async_: False = wait; True = the
        rows are valid in the order of the stream the object launches on; 2 = PIPELINED: the step is launched and the packets of
        the previous step are returned (receive_flush() delivers the last step's). A pipelined receiver keeps its private stream
        (the steps must stay in flight across calls), ordered against torch's current stream on the device, without a host wait:
        the launch stream follows what torch's stream holds at entry (whatever produced `buf`, whatever still reads the rows of the
        call before: lorahip_demod_stream_follow), and torch's stream waits for the packing of the rows at exit
        (lorahip_demod_stream_wait) -- work queued on it after this call sees the rows. order_with_torch=False leaves both out (two
        event records and two stream waits per step): for a caller that orders its own streams, or times the C entry itself."""
        import torch
        r = self._rows_struct(rows, async_ if async_ in (2, 3) else int(bool(async_)), depth if async_ == 3 else 0)
        n, calls = C.c_size_t(), C.c_int64()
        if async_ == 3:
            # RESIDENT: one kernel stays on the device and takes the steps as messages. The rows passed here are filled by THIS step and
            # are complete when the receive() `depth` calls later (default: the NEXT one) or receive_flush() returns (whose counts are this
            # step's): cycle depth + 1 sets of rows. The kernel reads `buf` on its
            # own: what produced the new samples must have finished (order_with_torch: torch's current stream is waited for).
            if order_with_torch:
                torch.cuda.current_stream(buf.device).synchronize()
            try:
                check(self._lib.lorahip_demod_receive(self._h, _dptr(buf), int(buf.shape[1]), int(n_valid), C.byref(r), C.byref(n), C.byref(calls)), "lorahip_demod_receive")
            except Exception as e:
                e.n_packets = n.value
                raise
            return n.value, calls.value
        if async_ == 2:
            if not order_with_torch:
                check(self._lib.lorahip_demod_receive(self._h, _dptr(buf), int(buf.shape[1]), int(n_valid), C.byref(r), C.byref(n), C.byref(calls)), "lorahip_demod_receive")
                return n.value, calls.value
            ts = C.c_void_p(torch.cuda.current_stream(buf.device).cuda_stream)
            check(self._lib.lorahip_demod_stream_follow(self._h, ts), "lorahip_demod_stream_follow")
            try:
                check(self._lib.lorahip_demod_receive(self._h, _dptr(buf), int(buf.shape[1]), int(n_valid), C.byref(r), C.byref(n), C.byref(calls)), "lorahip_demod_receive")
            except Exception as e:
                e.n_packets = n.value                       # LORAHIP_E_INVALID with rows that are too small: the rows needed (nothing is lost)
                raise
            finally:
                self._lib.lorahip_demod_stream_wait(self._h, ts)
            return n.value, calls.value
        check(self._lib.lorahip_demod_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(buf.device).cuda_stream)), "lorahip_demod_set_stream")
        try:
            check(self._lib.lorahip_demod_receive(self._h, _dptr(buf), int(buf.shape[1]), int(n_valid), C.byref(r), C.byref(n), C.byref(calls)), "lorahip_demod_receive")
        finally:
            self._lib.lorahip_demod_reset_stream(self._h)
        return n.value, calls.value

    def receive_flush(self, rows=None):
        """lorahip_demod_receive_flush: the last pipelined step's packets into `rows` (None: dropped); returns (n_packets, work_calls)
        with the launch stream drained"""
        n, calls = C.c_size_t(), C.c_int64()
        r = self._rows_struct(rows, 0) if rows is not None else None
        if rows is not None:
            import torch
            # (whatever on torch's stream still reads the rows of the call before is ordered before this call's packing)
            check(self._lib.lorahip_demod_stream_follow(self._h, C.c_void_p(torch.cuda.current_stream(rows[0].device).cuda_stream)), "lorahip_demod_stream_follow")
        try:
            check(self._lib.lorahip_demod_receive_flush(self._h, C.byref(r) if r is not None else None, C.byref(n), C.byref(calls)), "lorahip_demod_receive_flush")
        except Exception as e:
            e.n_packets = n.value
            raise
        return n.value, calls.value

    def work(self, streams):
        """Feed one complete input stream per channel and run work() until < 2N samples remain
        everywhere. streams: list of complex64 numpy arrays (one per channel, any lengths), ONE numpy array of shape
        (n_channels, samples) in host memory, or ONE torch complex64 device tensor of that shape. Returns the number of lock-step rounds."""
        rounds = C.c_int64()
        if _is_torch(streams):
            if streams.dim() != 2 or streams.shape[0] != self.n_channels:
                raise ValueError("expected a (n_channels, samples) tensor")
            import torch
            # run on torch's current stream: whatever produced `streams` there is ordered before the demodulator's kernels
            check(self._lib.lorahip_demod_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(streams.device).cuda_stream)),
                  "lorahip_demod_set_stream")
            try:
                check(self._lib.lorahip_demod_run_device(self._h, _dptr(streams), int(streams.shape[1]),
                                                         C.byref(rounds)), "lorahip_demod_run_device")
            finally:
                # torch may destroy that stream later: later calls (numpy work(), packets_device()) use the private one again
                self._lib.lorahip_demod_reset_stream(self._h)
            return rounds.value
        if isinstance(streams, np.ndarray) and streams.ndim == 2:
            # one (n_channels, samples) host array: the per-channel pointers without a Python loop
            if streams.shape[0] != self.n_channels:
                raise ValueError("expected a (n_channels, samples) array")
            a = np.ascontiguousarray(streams, np.complex64)
            ptrs = (np.uint64(a.ctypes.data) + np.arange(self.n_channels, dtype=np.uint64) * np.uint64(a.strides[0]))
            lens = np.full(self.n_channels, a.shape[1], dtype=np.uint64)
            check(self._lib.lorahip_demod_run(self._h, ptrs.ctypes.data_as(C.POINTER(C.c_void_p)), lens.ctypes.data_as(C.POINTER(C.c_size_t)),
                                              C.byref(rounds)), "lorahip_demod_run")
            return rounds.value
        if len(streams) != self.n_channels:
            raise ValueError("expected %d streams" % self.n_channels)
        keep = [np.ascontiguousarray(s, np.complex64).reshape(-1) for s in streams]
        ptrs = (C.c_void_p * self.n_channels)(*[k.ctypes.data for k in keep])
        lens = (C.c_size_t * self.n_channels)(*[k.size for k in keep])
        check(self._lib.lorahip_demod_run(self._h, ptrs, lens, C.byref(rounds)), "lorahip_demod_run")
        return rounds.value

    def packets_arrays(self, clear=True):
        """The queued packets as four flat arrays -- channel (int32), round (int64), length (int64), and all symbols back to
        back (int16) -- in posting order: what lorahip_demod_get_packets delivers, without per-packet Python objects"""
        n = self._lib.lorahip_demod_num_packets(self._h)
        ns = self._lib.lorahip_demod_num_packet_symbols(self._h)
        ch, rd, ln = np.empty(n, np.int32), np.empty(n, np.int64), np.empty(n, np.int64)
        syms = np.empty(ns, np.int16)
        check(self._lib.lorahip_demod_get_packets(self._h, ch.ctypes.data, rd.ctypes.data, ln.ctypes.data, n, syms.ctypes.data, ns),
              "lorahip_demod_get_packets")
        if clear:
            self._lib.lorahip_demod_clear_packets(self._h)
        return ch, rd, ln, syms

    def clear_packets(self):
        """drop the queued packets (what packets*(clear=True) do after reading them)"""
        self._lib.lorahip_demod_clear_packets(self._h)

    def packets(self, clear=True):
        """[(channel, round, int16 symbols)] -- the Pothos::Packet payloads of output port 0"""
        ch, rd, ln, syms = self.packets_arrays(clear)
        parts = np.split(syms, np.cumsum(ln)[:-1]) if ch.size else []       # views into one array: no per-packet copy
        return list(zip(ch.tolist(), rd.tolist(), parts))

    def packets_device(self, stride=None, clear=True):
        """The queued packets as device tensors in the decoder's input layout -- (P, stride) int16 symbols (zero padded), (P,) int32
        lengths, (P,) int32 channels -- ready for LoRaDecoder.decode_batch(); no per-packet Python work. Straight after a
        work() of the streaming mode the packets never visit the host (rows ordered by channel, then time); otherwise the
        rows follow packets()'s order. stride defaults to the MTU."""
        import torch
        n = int(self._lib.lorahip_demod_num_packets(self._h))
        dev = torch.device("cuda", int(self._device))
        if stride is None:
            stride = max(8, min(self._mtu, int(self._lib.lorahip_decode_max_symbols())))      # no packet is longer than the MTU (LoRaDemod.cpp:291)
        syms = torch.empty((n, int(stride)), dtype=torch.int16, device=dev)      # every element is written (rows are zero padded)
        nsyms = torch.empty(n, dtype=torch.int32, device=dev)
        chan = torch.empty(n, dtype=torch.int32, device=dev)
        got = C.c_size_t()
        if n:
            torch.cuda.current_stream(dev).synchronize()            # the blocks may have pending work of their previous owner
            check(self._lib.lorahip_demod_packets_to_device(self._h, C.c_void_p(syms.data_ptr()), int(stride), C.c_void_p(nsyms.data_ptr()),
                                                            C.c_void_p(chan.data_ptr()), n, C.byref(got)), "lorahip_demod_packets_to_device")
        if clear:
            self._lib.lorahip_demod_clear_packets(self._h)
        return syms, nsyms, chan

    def consumed(self, channel):
        """samples of the channel's stream consumed by the last work()"""
        return int(self._lib.lorahip_demod_consumed(self._h, int(channel)))

    def consumed_all(self):
        """consumed(c) of every channel: an int64 numpy array (one call into the library)"""
        out = np.empty(self.n_channels, np.int64)
        check(self._lib.lorahip_demod_consumed_all(self._h, out.ctypes.data_as(C.POINTER(C.c_int64))), "lorahip_demod_consumed_all")
        return out

    def work_calls(self):
        return int(self._lib.lorahip_demod_work_calls(self._h))

    def kernel_ms(self):
        """device time of the streaming kernel launches of the last work() (HIP events on the launch stream)"""
        return float(self._lib.lorahip_demod_kernel_ms(self._h))

    def last_launches(self):
        """streaming kernel launches the last work() took (1 unless its record buffers filled and the run was resumed)"""
        return int(self._lib.lorahip_demod_last_launches(self._h))

    def near_threshold(self):
        """(near_squelch, near_step): decisions since activate() that sat within float rounding of their boundary -- |snr - thresh|
        <= 4e-5 dB where the squelch is consumed, fine-tune steps within 6e-5 of an integer (include/lorahip.h). Counted, not changed."""
        a, b = C.c_int64(), C.c_int64()
        check(self._lib.lorahip_demod_near_threshold(self._h, C.byref(a), C.byref(b)), "lorahip_demod_near_threshold")
        return a.value, b.value

    def set_fine_gather(self, on):
        """A/B switch: read the fine-tune table in HBM instead of evaluating it from the split tables (include/lorahip.h)"""
        check(self._lib.lorahip_demod_set_fine_gather(self._h, int(bool(on))), "lorahip_demod_set_fine_gather")

    def labels(self, channel):
        """The stream labels the block posts at index 0 of what each work() call produces on raw / dec / fft, one per call
        ("" = none): "SYNC", "P <fIndex>", "DC", "QC", "S<n> <fIndex>" (LoRaDemod.cpp:213,221-224,245,282,302-305,314-319),
        formatted by the library from the per-call trace (set_trace(True) before work())."""
        n, nb = C.c_size_t(), C.c_size_t()
        check(self._lib.lorahip_demod_get_labels(self._h, int(channel), None, 0, C.byref(n), C.byref(nb)), "lorahip_demod_get_labels")
        buf = C.create_string_buffer(max(nb.value, 1))
        check(self._lib.lorahip_demod_get_labels(self._h, int(channel), buf, nb.value, C.byref(n), C.byref(nb)), "lorahip_demod_get_labels")
        out = buf.raw[:nb.value].split(b"\0")[:n.value]
        return [x.decode() for x in out]

    def set_ports(self, fft_frames=0, dec_samples=0, raw_samples=0):
        """Attach the block's debug ports for the next work() calls (LoRaDemod.cpp:81-83): per channel, `fft_frames` frames of N
        FFT bins (one per work() call), `dec_samples` dechirped samples, `raw_samples` consumed samples; 0 = that port off,
        all 0 = ports off. The buffers are device tensors owned by this object; read them with ports(channel)."""
        import torch
        if self.N is None:
            raise ValueError("the debug ports need one spreading factor for all channels (include/lorahip.h)")
        dev = torch.device("cuda", int(self._device))
        B = self.n_channels
        self._port_bufs = dict(
            fft=torch.zeros((B, int(fft_frames), self.N), dtype=torch.complex64, device=dev) if fft_frames else None,
            dec=torch.zeros((B, int(dec_samples)), dtype=torch.complex64, device=dev) if dec_samples else None,
            raw=torch.zeros((B, int(raw_samples)), dtype=torch.complex64, device=dev) if raw_samples else None)
        torch.cuda.synchronize(dev)
        if not (fft_frames or dec_samples or raw_samples):
            check(self._lib.lorahip_demod_set_ports(self._h, None), "lorahip_demod_set_ports")
            return
        p = _lib.DemodPorts()
        p.struct_size = C.sizeof(_lib.DemodPorts)
        b = self._port_bufs
        p.fft_dev, p.fft_cap_frames = (b["fft"].data_ptr() if fft_frames else None), int(fft_frames)
        p.dec_dev, p.dec_cap_samples = (b["dec"].data_ptr() if dec_samples else None), int(dec_samples)
        p.raw_dev, p.raw_cap_samples = (b["raw"].data_ptr() if raw_samples else None), int(raw_samples)
        check(self._lib.lorahip_demod_set_ports(self._h, C.byref(p)), "lorahip_demod_set_ports")

    def ports(self, channel):
        """what the last work() produced on the debug ports of `channel`: dict of device tensors fft (frames, N), dec (samples,),
        raw (samples,) -- cut to what was produced (or to the capacity given to set_ports)"""
        nf, nd, nr = C.c_size_t(), C.c_size_t(), C.c_size_t()
        check(self._lib.lorahip_demod_port_counts(self._h, int(channel), C.byref(nf), C.byref(nd), C.byref(nr)), "lorahip_demod_port_counts")
        b = self._port_bufs
        return dict(fft=None if b["fft"] is None else b["fft"][channel, :min(nf.value, b["fft"].shape[1])],
                    dec=None if b["dec"] is None else b["dec"][channel, :min(nd.value, b["dec"].shape[1])],
                    raw=None if b["raw"] is None else b["raw"][channel, :min(nr.value, b["raw"].shape[1])],
                    produced=dict(fft=nf.value, dec=nd.value, raw=nr.value))

    def trace_array(self, channel):
        """the per-call trace of `channel` as one numpy structured array (fields of lorahip_work_result): no per-call Python objects"""
        n = self._lib.lorahip_demod_trace_len(self._h, int(channel))
        arr = np.zeros(n, dtype=WORK_RESULT_DTYPE)
        if n:
            check(self._lib.lorahip_demod_get_trace(self._h, int(channel), C.cast(arr.ctypes.data, C.POINTER(WorkResult)), n), "lorahip_demod_get_trace")
        return arr

    def trace(self, channel):
        n = self._lib.lorahip_demod_trace_len(self._h, int(channel))
        arr = (WorkResult * n)()
        if n:
            check(self._lib.lorahip_demod_get_trace(self._h, int(channel), arr, n), "lorahip_demod_get_trace")
        return [dict((f, getattr(r, f)) for f, _ in WorkResult._fields_) for r in arr]


class LoRaDecoder:
    """The `/lora/lora_decoder` block (LoRaDecoder.cpp), same setters and defaults, for batches of symbol packets.

    work(packets): list of int16/uint16 symbol arrays (what LoRaDemod posts) -> list of bytes arrays (None where the
    block posts nothing); decode_batch(): the same on device tensors."""

    _CR = {"4/4": 0, "4/5": 1, "4/6": 2, "4/7": 3, "4/8": 4}

    def __init__(self, device=0):
        self._ctx = Context(7, device=device)             # device + stream only; the decoder has its own sf
        self._cfg = DecoderCfg(C.sizeof(DecoderCfg), 10, 0, 4, 0, 1, 0, 1, 0, 8)   # LoRaDecoder.cpp:98-110
        self._whitening = True
        self._dropped = 0

    @staticmethod
    def make():
        return LoRaDecoder()

    def setSpreadFactor(self, sf): self._cfg.sf = int(sf)
    def setSymbolSize(self, ppm): self._cfg.ppm = int(ppm)

    def setCodingRate(self, cr):
        if cr not in self._CR:
            raise ValueError("LoRaDecoder::setCodingRate(%s): unknown coding rate" % cr)   # InvalidArgumentException :150
        self._cfg.rdd = self._CR[cr]

    def enableWhitening(self, on): self._whitening = bool(on)     # stored, never read by work() -- like the reference
    def enableCrcc(self, on): self._cfg.crcc = int(bool(on))
    def enableInterleaving(self, on): self._cfg.interleaving = int(bool(on))
    def enableExplicit(self, on): self._cfg.explicit_hdr = int(bool(on))
    def enableHdr(self, on): self._cfg.hdr = int(bool(on))
    def enableErrorCheck(self, on): self._cfg.error_check = int(bool(on))
    def setDataLength(self, n): self._cfg.data_length = int(n)
    def getDropped(self): return self._dropped
    def activate(self): self._dropped = 0

    def decode_batch(self, syms, nsyms):
        """syms: (P, stride) int16/uint16 device tensor, nsyms: (P,) int32 device tensor ->
        (out uint8 (P, out_stride), out_len int32 (P,), dropped int32 (P,)) device tensors"""
        import torch
        if syms.dim() != 2 or syms.dtype not in (torch.int16, torch.uint16) or nsyms.dtype != torch.int32:
            raise ValueError("syms must be (P, stride) 16-bit, nsyms (P,) int32 device tensors")
        syms, nsyms = syms.contiguous(), nsyms.contiguous()
        P, stride = int(syms.shape[0]), int(syms.shape[1])
        out_stride = 2 * (stride + 8)
        out = torch.zeros((P, out_stride), dtype=torch.uint8, device=syms.device)
        out_len = torch.empty(P, dtype=torch.int32, device=syms.device)
        dropped = torch.empty(P, dtype=torch.int32, device=syms.device)
        self._ctx.use_torch_stream()
        check(self._ctx._lib.lorahip_decode_packets(self._ctx._h, C.byref(self._cfg), _dptr(syms), stride, _dptr(nsyms), P,
                                                    _dptr(out), out_stride, _dptr(out_len), _dptr(dropped)), "lorahip_decode_packets")
        return out, out_len, dropped

    def work(self, packets):
        import torch
        if self._cfg.ppm > self._cfg.sf:
            raise ValueError("LoRaDecoder::work(): failed check: PPM <= SF")            # Pothos::Exception :202
        P = len(packets)
        if P == 0:
            return []
        stride = max(8, max(len(p) for p in packets))
        host = np.zeros((P, stride), np.uint16)
        for i, p in enumerate(packets):
            host[i, :len(p)] = np.asarray(p).astype(np.uint16)
        n = np.array([len(p) for p in packets], np.int32)
        dev = torch.device("cuda", self._ctx.device)
        out, out_len, dropped = self.decode_batch(torch.from_numpy(host.view(np.int16)).to(dev), torch.from_numpy(n).to(dev))
        out, out_len, dropped = out.cpu().numpy(), out_len.cpu().numpy(), dropped.cpu().numpy()
        self._dropped += int(dropped.sum())
        res = []
        for i in range(P):
            if out_len[i] < 0:
                res.append(None)
            elif self._cfg.interleaving:
                res.append(out[i, :out_len[i]].copy())
            else:
                res.append(out[i, :2 * out_len[i]].view(np.uint16).copy())
        return res
