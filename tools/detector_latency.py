"""Latency of the level-1 seam (lorahip_detector_feed x N + lorahip_detector_detect: LoRaDetector.hpp:8-72), one window per call from host memory.
    python tools/detector_latency.py [sf ...]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lora_sdr_amd import _lib
lib = _lib.load()
for sf in [int(x) for x in sys.argv[1:]] or [7, 10, 12]:
    N = 1 << sf
    det = C.c_void_p()
    assert lib.lorahip_detector_create(C.byref(det), 0, N) == 0
    rng = np.random.default_rng(sf)
    x = (rng.normal(size=N) + 1j * rng.normal(size=N)).astype(np.complex64)
    for i in range(N):
        lib.lorahip_detector_feed(det, i, float(x[i].real), float(x[i].imag))
    idx, pw, pa, fi = C.c_size_t(), C.c_float(), C.c_float(), C.c_float()
    for _ in range(50):
        lib.lorahip_detector_detect(det, C.byref(idx), C.byref(pw), C.byref(pa), C.byref(fi), None)
    reps = 2000
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.lorahip_detector_detect(det, C.byref(idx), C.byref(pw), C.byref(pa), C.byref(fi), None)
    dt = (time.perf_counter() - t0) / reps
    print("SF%d: %.1f us per detect() (index %d, power %.3f dB)" % (sf, dt * 1e6, idx.value, pw.value))
    lib.lorahip_detector_destroy(det)
