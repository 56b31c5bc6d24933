"""lorahip_demod_run_host_rows against lorahip_demod_run on pinned and ordinary host memory: upload + streaming kernel per call.
    LORAHIP_DEMOD_TIMING=1 python tools/host_rows_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch, lora_sdr_amd as L
for sf, B, win in ((7, 16384, 130), (10, 4096, 130), (12, 1024, 130)):
    N = 1 << sf
    n = win * N
    stride = n + 2 * N
    rows = L.pinned_empty((B, stride))
    rng = np.random.default_rng(1)
    rows[...] = 0
    rows[:, :n] = (0.05 * (rng.standard_normal((64, n)) + 1j * rng.standard_normal((64, n)))).astype(np.complex64)[np.arange(B) % 64]
    first = np.zeros(B, np.int64); cnt = np.full(B, n, np.uint64)
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1)
    ordinary = np.ascontiguousarray(rows[:, :n])
    pinned_c = L.pinned_empty((B, n)); pinned_c[...] = ordinary
    for name, fn in (("run_host_rows (pinned rows, one strided DMA)", lambda: d.work_host_rows(rows, first, cnt)),
                     ("run (pinned, contiguous per channel)", lambda: d.work(pinned_c)),
                     ("run (ordinary memory)", lambda: d.work(ordinary))):
        fn(); d.clear_packets(); d.activate()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0); d.clear_packets(); d.activate()
        print("SF%d %d ch x %d windows (%.2f GB): %-48s %.1f ms = %.1f GB/s" % (sf, B, win, B * n * 8 / 1e9, name, min(ts) * 1e3, B * n * 8 / min(ts) / 1e9), flush=True)
    d.close()
