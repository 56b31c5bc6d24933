"""lora_sdr_amd -- MI355X-native LoRa demodulation hot path (dechirp -> FFT -> detect).

The compute lives in liblorahip.so (hand-written HIP for gfx950, C ABI in include/lorahip.h);
this package is the thin host-side mirror of the reference's LoRaDetector / LoRaDemod
interface. There is no CPU implementation here: without the shared library (or without a
gfx950 device) the calls raise.
"""
from ._lib import LoraHipError, load, LIB_PATH, CHIRP_UP, CHIRP_DOWN, CHIRP_NONE, SF_MIN, SF_MAX, FINE_STEPS  # noqa: F401
from .api import Context, LoRaDetector, LoRaDemod, LoRaDecoder, Channelizer, MixedDetector, MixedDetectorMulti, shard_plan, pinned_empty, design_lowpass, host_tables, device_count  # noqa: F401
from .shard import shard_channels, bytes_per_symbol  # noqa: F401

__all__ = ["Context", "LoRaDetector", "LoRaDemod", "LoRaDecoder", "Channelizer", "MixedDetector", "MixedDetectorMulti", "shard_plan", "pinned_empty", "design_lowpass", "host_tables", "device_count", "LoraHipError", "load",
           "shard_channels", "bytes_per_symbol", "CHIRP_UP", "CHIRP_DOWN", "CHIRP_NONE"]
