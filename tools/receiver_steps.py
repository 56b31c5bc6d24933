"""The running receiver, sequential against pipelined steps (lorahip_demod_receive, async 1 / 2), at three chunk sizes:
    python tools/receiver_steps.py [sf ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lora_sdr_amd as L
from lora_sdr_amd import workloads as WL
for sf in [int(x) for x in sys.argv[1:]] or [7, 10]:
    B = WL.LEVEL3_CHANNELS[sf]
    ctx = L.Context(sf); iq, data = WL.frame_streams(ctx, B, 4, 48, sigma=0.05)
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(48)
    rows = d.receiver_rows(B * 5, 48); cap = iq.shape[1]
    for mode in (True, 2):
        for cw in (128, 32, 8):
            best = None
            for rep in range(4):
                d.clear_packets(); d.rewind(); d.activate()
                w = npk = calls = steps = 0
                torch.cuda.synchronize(); t0 = time.perf_counter()
                while w < cap:
                    w = min(cap, w + (cw << sf)); n, k = (d.receive(iq, w, rows, async_=2, order_with_torch=False) if mode == 2 else d.receive(iq, w, rows, async_=True)); npk += n; calls += k; steps += 1
                if mode == 2:
                    n, k = d.receive_flush(rows); npk += n; calls += k
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            print("SF%d %s chunks of %3d windows: %3d steps, %.1f us per step, %.1f Msym/s, frac %.4f, %d packets" % (sf, "pipelined " if mode == 2 else "sequential", cw, steps, best / steps * 1e6, calls / best / 1e6, calls * L.bytes_per_symbol(sf) / best / 8e12, npk), flush=True)
    d.close(); ctx.close()
