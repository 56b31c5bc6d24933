"""The running receiver's rate as a function of the chunk length: B channels in one (B, capacity) device buffer, every work() sees the
unconsumed remainder of each channel plus the next `chunk` samples (lorahip_demod_run_device_segments). Packets are handed to the
device-side queue (packets_device) per chunk. Prints wall time per chunk, kernel time per chunk and the sustained rate.
    python tools/running_receiver.py --sf 7 --channels 16384 --frames 8 --chunks 2048,8192,32768"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lora_sdr_amd as L
from lora_sdr_amd import workloads as WL

ap = argparse.ArgumentParser()
ap.add_argument("--sf", type=int, default=7); ap.add_argument("--channels", type=int, default=16384)
ap.add_argument("--frames", type=int, default=8); ap.add_argument("--nsyms", type=int, default=48)
ap.add_argument("--chunks", default="2048,8192,32768")
a = ap.parse_args()
sf, N, B = a.sf, 1 << a.sf, a.channels
ctx = L.Context(sf)
iq, data = WL.frame_streams(ctx, B, a.frames, a.nsyms, sigma=0.05)
cap = iq.shape[1]
one = L.LoRaDemod(sf, n_channels=B); one.set_mode(1); one.setMTU(a.nsyms)
one.work(iq); want_calls = one.work_calls(); want_pk = len(one.packets()); k_one = one.kernel_ms(); one.close()
print("SF%d: %d channels x %d samples; one work(): %d calls, %d packets, kernel %.3f ms" % (sf, B, cap, want_calls, want_pk, k_one))
for chunk in [int(c) for c in a.chunks.split(",")]:
    # one receiver object, as it would run: the first pass over the capture is checked (and allocates), the following ones are timed
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(a.nsyms)
    row = np.arange(B, dtype=np.int64) * cap
    best = None
    for rep in range(4):
        calls0 = d.work_calls()                          # (activate() does not reset the count of a running object)
        d.activate()
        read = np.zeros(B, np.int64)
        w, npk, kms, nch = 0, 0, 0.0, 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        while w < cap:
            w = min(cap, w + chunk)
            d.work_segments(iq, row + read, w - read)
            kms += d.kernel_ms(); nch += 1
            syms, nsyms, chan = d.packets_device()
            npk += int(nsyms.numel())
            read += d.consumed_all()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        calls = d.work_calls() - calls0
        if rep == 0:
            assert calls == want_calls and npk == want_pk, (calls, want_calls, npk, want_pk)
        elif best is None or dt < best[0]:
            best = (dt, kms, nch, calls)
    d.close()
    dt, kms, nch, calls = best
    print("  chunk %6d samples (%5.1f windows): %4d work() calls of the block, %.3f ms each (kernel %.3f ms), %.1f Msym/s sustained (kernel-only %.1f); first pass: same calls and packets as one work()"
          % (chunk, chunk / N, nch, dt / nch * 1e3, kms / nch, calls / dt / 1e6, calls / kms / 1e3), flush=True)
