"""Where one lorahip_demod_receive() goes (the running receiver, bench.py level3[].running): the library's own stderr breakdown
(LORAHIP_DEMOD_TIMING=1) and the wall clock per call, at two chunk sizes.
    LORAHIP_DEMOD_TIMING=1 python tools/receive_breakdown.py [sf] [channels]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lora_sdr_amd as L
from lora_sdr_amd import workloads as WL

sf = int(sys.argv[1]) if len(sys.argv) > 1 else 7
B = int(sys.argv[2]) if len(sys.argv) > 2 else WL.LEVEL3_CHANNELS[sf]
ctx = L.Context(sf)
iq, data = WL.frame_streams(ctx, B, 4, 48, sigma=0.05)
d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(48)
rows = d.receiver_rows(B * 5, 48)
cap = iq.shape[1]
for cw in (128, 32, 8):
    for rep in range(3):
        d.rewind(); d.activate(); d.clear_packets()
        w, per = 0, []
        torch.cuda.synchronize()
        while w < cap:
            w = min(cap, w + (cw << sf))
            t0 = time.perf_counter()
            n, k = d.receive(iq, w, rows, async_=True)
            per.append((time.perf_counter() - t0, d.kernel_ms(), k))
        torch.cuda.synchronize()
    full = [p for p in per[:-1]] or per
    print("SF%d %d channels, chunks of %d windows: %d receive() calls, wall %.1f us each, streaming kernel %.1f us each, host + small kernels %.1f us each (full chunks)"
          % (sf, B, cw, len(per), 1e6 * sum(p[0] for p in full) / len(full), 1e3 * sum(p[1] for p in full) / len(full),
             1e6 * sum(p[0] for p in full) / len(full) - 1e3 * sum(p[1] for p in full) / len(full)), flush=True)
