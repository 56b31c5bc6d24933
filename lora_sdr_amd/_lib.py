"""ctypes binding of liblorahip.so (include/lorahip.h). No fallbacks: if the shared library
is missing or a call fails, this raises -- there is no CPU path in the product."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# LORAHIP_LIB: another build of the same library (A/B measurements of kernel changes inside one GPU session, tools/gpu_ab.sh)
LIB_PATH = os.environ.get("LORAHIP_LIB") or os.path.join(HERE, "liblorahip.so")

OK = 0
SF_MIN, SF_MAX = 6, 12
FINE_STEPS = 128
CHIRP_UP, CHIRP_DOWN, CHIRP_NONE = 0, 1, 2

_f32p = C.POINTER(C.c_float)


class LoraHipError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        lib = load()
        msg = lib.lorahip_strerror(code).decode()
        detail = lib.lorahip_last_error().decode()
        super().__init__("%s failed: %s (%d)%s" % (where, msg, code, (": " + detail) if detail and code == -3 else ""))


class Batch(C.Structure):
    """struct lorahip_batch"""
    _fields_ = [("struct_size", C.c_size_t),
                ("iq", C.c_void_p), ("n_windows", C.c_size_t), ("offsets", C.c_void_p),
                ("window_stride", C.c_size_t), ("chirp_sel", C.c_void_p), ("chirp_sel_all", C.c_int32),
                ("fine_idx0", C.c_void_p), ("fine_err", C.c_void_p),
                ("sym", C.c_void_p), ("power", C.c_void_p), ("power_avg", C.c_void_p), ("f_index", C.c_void_p),
                ("fine_idx_out", C.c_void_p), ("fft_out", C.c_void_p), ("dec_out", C.c_void_p)]


class DecoderCfg(C.Structure):
    _fields_ = [("struct_size", C.c_size_t)] + [(n, C.c_int32) for n in ("sf", "ppm", "rdd", "crcc", "interleaving", "error_check",
                                                                          "explicit_hdr", "hdr", "data_length")]


class DemodPorts(C.Structure):
    """struct lorahip_demod_ports"""
    _fields_ = [("struct_size", C.c_size_t), ("fft_dev", C.c_void_p), ("fft_cap_frames", C.c_size_t), ("dec_dev", C.c_void_p),
                ("dec_cap_samples", C.c_size_t), ("raw_dev", C.c_void_p), ("raw_cap_samples", C.c_size_t),
                ("host_buffers", C.c_int32), ("reserved", C.c_int32)]


class PacketRows(C.Structure):
    """struct lorahip_packet_rows"""
    _fields_ = [("struct_size", C.c_size_t), ("syms_dev", C.c_void_p), ("sym_stride", C.c_size_t), ("nsyms_dev", C.c_void_p),
                ("channel_dev", C.c_void_p), ("cap_packets", C.c_size_t), ("async_", C.c_int32), ("reserved", C.c_int32)]


class SignalRows(C.Structure):
    """struct lorahip_signal_rows"""
    _fields_ = [("struct_size", C.c_size_t), ("channel", C.c_void_p), ("error", C.c_void_p), ("power", C.c_void_p), ("snr", C.c_void_p), ("cap", C.c_size_t)]


class WorkResult(C.Structure):
    """struct lorahip_work_result"""
    _fields_ = [("consumed", C.c_int64), ("state_before", C.c_int32), ("value", C.c_int32),
                ("power", C.c_float), ("power_avg", C.c_float), ("snr", C.c_float), ("f_index", C.c_float),
                ("worked", C.c_int32), ("packet_len", C.c_int32), ("signals", C.c_int32),
                ("sig_error", C.c_int32), ("sig_power", C.c_float), ("sig_snr", C.c_float),
                ("fine_idx_before", C.c_int32), ("fine_idx_after", C.c_int32), ("fine_err_before", C.c_float), ("reserved", C.c_int32)]


# every symbol include/lorahip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "lorahip_strerror": (C.c_char_p, [C.c_int]),
    "lorahip_last_error": (C.c_char_p, []),
    "lorahip_version": (C.c_int, []),
    "lorahip_selfcheck": (C.c_int, []),
    "lorahip_device_count": (C.c_int, []),
    "lorahip_host_tables": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lorahip_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "lorahip_destroy": (None, [C.c_void_p]),
    "lorahip_sf": (C.c_int, [C.c_void_p]),
    "lorahip_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lorahip_synchronize": (C.c_int, [C.c_void_p]),
    "lorahip_reset_stream": (C.c_int, [C.c_void_p]),
    "lorahip_demod_reset_stream": (C.c_int, [C.c_void_p]),
    "lorahip_set_variant": (C.c_int, [C.c_void_p, C.c_int]),
    "lorahip_set_fine_gather": (C.c_int, [C.c_void_p, C.c_int]),
    "lorahip_fine_split_active": (C.c_int, [C.c_void_p]),
    "lorahip_fine_split_selftest": (C.c_int, [C.c_int]),
    "lorahip_fine_indices_host": (C.c_int, [C.c_int, C.c_int32, C.c_float, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "lorahip_demod_set_fine_gather": (C.c_int, [C.c_void_p, C.c_int]),
    "lorahip_demod_set_variant": (C.c_int, [C.c_void_p, C.c_int]),
    "lorahip_demod_set_stream_grid": (C.c_int, [C.c_void_p, C.c_int]),
    "lorahip_demod_set_record_capacity": (C.c_int, [C.c_void_p, C.c_size_t]),
    "lorahip_demod_set_stream_lanes": (C.c_int, [C.c_void_p, C.c_int]),
    "lorahip_demod_run_host_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "lorahip_demod_stream_lanes": (C.c_int, [C.c_void_p]),
    "lorahip_demod_stream_wait": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lorahip_demod_stream_follow": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lorahip_detect_batch": (C.c_int, [C.c_void_p, C.POINTER(Batch)]),
    "lorahip_detect_batch_host": (C.c_int, [C.c_void_p, C.POINTER(Batch)]),
    "lorahip_mixed_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_size_t]),
    "lorahip_mixed_destroy": (None, [C.c_void_p]),
    "lorahip_mixed_num_buckets": (C.c_size_t, [C.c_void_p]),
    "lorahip_mixed_bucket": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "lorahip_mixed_context": (C.c_void_p, [C.c_void_p, C.c_size_t]),
    "lorahip_mixed_rows": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lorahip_mixed_plan": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "lorahip_mixed_detect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lorahip_mixed_synchronize": (C.c_int, [C.c_void_p]),
    "lorahip_shard_plan": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]),
    "lorahip_mixed_create_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "lorahip_mixed_num_devices": (C.c_size_t, [C.c_void_p]),
    "lorahip_mixed_device": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_size_t)]),
    "lorahip_mixed_shard": (C.c_void_p, [C.c_void_p, C.c_size_t]),
    "lorahip_mixed_shard_of": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lorahip_mixed_detect_multi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lorahip_host_alloc": (C.c_void_p, [C.c_size_t]),
    "lorahip_host_free": (None, [C.c_void_p]),
    "lorahip_timer_start": (C.c_int, [C.c_void_p]),
    "lorahip_timer_stop": (C.c_int, [C.c_void_p, _f32p]),
    "lorahip_detector_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_size_t]),
    "lorahip_detector_destroy": (None, [C.c_void_p]),
    "lorahip_detector_feed": (C.c_int, [C.c_void_p, C.c_size_t, C.c_float, C.c_float]),
    "lorahip_detector_detect": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), _f32p, _f32p, _f32p, C.c_void_p]),
    "lorahip_demod_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_size_t]),
    "lorahip_demod_destroy": (None, [C.c_void_p]),
    "lorahip_demod_create_mixed": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "lorahip_demod_num_channels": (C.c_size_t, [C.c_void_p]),
    "lorahip_demod_num_parts": (C.c_size_t, [C.c_void_p]),
    "lorahip_demod_part": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_size_t), C.POINTER(C.c_int32)]),
    "lorahip_demod_part_of": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "lorahip_demod_part_handle": (C.c_void_p, [C.c_void_p, C.c_size_t]),
    "lorahip_demod_run_device_segments_multi": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(C.c_int64), C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]),
    "lorahip_demod_run_device_append": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_int64)]),
    "lorahip_demod_rewind": (C.c_int, [C.c_void_p]),
    "lorahip_demod_receive": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(PacketRows), C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]),
    "lorahip_demod_receive_flush": (C.c_int, [C.c_void_p, C.POINTER(PacketRows), C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]),
    "lorahip_demod_receive_signal_rows": (C.c_int, [C.c_void_p, C.POINTER(SignalRows)]),
    "lorahip_demod_receive_num_signals": (C.c_size_t, [C.c_void_p]),
    "lorahip_demod_resident_active": (C.c_int, [C.c_void_p]),
    "lorahip_demod_receive_steps": (C.c_size_t, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_size_t]),
    "lorahip_demod_set_signals": (C.c_int, [C.c_void_p, C.c_int]),
    "lorahip_demod_num_signals": (C.c_size_t, [C.c_void_p]),
    "lorahip_demod_get_signals": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "lorahip_demod_set_sync": (C.c_int, [C.c_void_p, C.c_ubyte]),
    "lorahip_demod_set_threshold": (C.c_int, [C.c_void_p, C.c_double]),
    "lorahip_demod_set_mtu": (C.c_int, [C.c_void_p, C.c_size_t]),
    "lorahip_mod_frame_len": (C.c_size_t, [C.c_int, C.c_size_t, C.c_size_t]),
    "lorahip_mod_frames": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_ubyte, C.c_float, C.c_size_t]),
    "lorahip_add_awgn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_uint64]),
    "lorahip_decode_packets": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                        C.c_void_p, C.c_void_p]),
    "lorahip_decode_packets_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                             C.c_void_p, C.c_void_p]),
    "lorahip_decode_max_symbols": (C.c_int, []),
    "lorahip_decode_max_data_length": (C.c_int, []),
    "lorahip_channelizer_phase_inc": (C.c_uint64, [C.c_double]),
    "lorahip_channelizer_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "lorahip_channelizer_destroy": (None, [C.c_void_p]),
    "lorahip_channelizer_reset": (C.c_int, [C.c_void_p]),
    "lorahip_channelizer_out_count": (C.c_size_t, [C.c_void_p, C.c_size_t]),
    "lorahip_channelizer_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "lorahip_channelizer_run_captures": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "lorahip_demod_activate": (C.c_int, [C.c_void_p]),
    "lorahip_demod_set_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "lorahip_demod_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lorahip_demod_packets_to_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "lorahip_demod_run": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]),
    "lorahip_demod_run_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int64)]),
    "lorahip_demod_run_device_segments": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]),
    "lorahip_demod_num_packets": (C.c_size_t, [C.c_void_p]),
    "lorahip_demod_get_packet": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                           C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t]),
    "lorahip_demod_num_packet_symbols": (C.c_size_t, [C.c_void_p]),
    "lorahip_demod_get_packets": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "lorahip_demod_clear_packets": (None, [C.c_void_p]),
    "lorahip_demod_consumed": (C.c_int64, [C.c_void_p, C.c_size_t]),
    "lorahip_demod_consumed_all": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "lorahip_demod_work_calls": (C.c_int64, [C.c_void_p]),
    "lorahip_demod_kernel_ms": (C.c_double, [C.c_void_p]),
    "lorahip_demod_last_launches": (C.c_int, [C.c_void_p]),
    "lorahip_demod_near_threshold": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "lorahip_demod_set_trace": (C.c_int, [C.c_void_p, C.c_int]),
    "lorahip_demod_set_ports": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lorahip_demod_port_counts": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "lorahip_demod_get_labels": (C.c_int, [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "lorahip_demod_trace_len": (C.c_size_t, [C.c_void_p, C.c_size_t]),
    "lorahip_demod_get_trace": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(WorkResult), C.c_size_t]),
    "lorahip_membw_probe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int]),
    "lorahip_synth_symbols": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float,
                                        C.c_uint64]),
}

_lib = None


def load():
    """Load liblorahip.so (once). Raises OSError with a build hint if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("%s not found: build it with `python -m lora_sdr_amd.build` "
                      "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    try:
        # bring in torch's HIP runtime first when torch is around, so that one libamdhip64 is shared
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the C ABI itself
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)     # AttributeError here = header/library mismatch
        except AttributeError:
            if os.environ.get("LORAHIP_LIB"):
                continue                # an older build under A/B measurement lacks newer entry points: calling one still fails
            raise
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, where):
    if code != OK:
        raise LoraHipError(code, where)
