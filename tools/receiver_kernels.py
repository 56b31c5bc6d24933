"""The kernels of a running receiver's steps (lorahip_demod_receive, one call per chunk), for rocprofv3 --kernel-trace --stats:
    rocprofv3 --kernel-trace --stats -d out -o rx --output-format csv -- python tools/receiver_kernels.py 7 8 [2]
(sf, windows per chunk, async mode 1 = sequential / 2 = pipelined)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lora_sdr_amd as L
from lora_sdr_amd import workloads as WL
sf, cw = int(sys.argv[1]), int(sys.argv[2])
mode = 2 if len(sys.argv) > 3 and sys.argv[3] == "2" else True
B = WL.LEVEL3_CHANNELS[sf]
ctx = L.Context(sf); iq, data = WL.frame_streams(ctx, B, 4, 48, sigma=0.05)
d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(48)
rows = d.receiver_rows(B * 5, 48); cap = iq.shape[1]
best = None
for rep in range(4):
    d.clear_packets(); d.rewind(); d.activate()
    w = steps = calls = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    while w < cap:
        w = min(cap, w + (cw << sf))
        n, k = d.receive(iq, w, rows, async_=mode, order_with_torch=False) if mode == 2 else d.receive(iq, w, rows, async_=True)
        steps += 1; calls += k
    if mode == 2:
        n, k = d.receive_flush(rows); calls += k
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
print("SF%d %s chunks of %d windows: %d steps, %.1f us per step, %.1f Msym/s" % (sf, "pipelined" if mode == 2 else "sequential", cw, steps, best / steps * 1e6, calls / best / 1e6))
