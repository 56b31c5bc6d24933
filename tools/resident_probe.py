"""The receiver's steps three ways -- sequential, pipelined (async = 2), resident (async = 3) -- on the level-3 workload, at a given chunk length:
    python tools/resident_probe.py --sf 7 --channels 16384 --chunk 8 [--reps 3]
prints Msym/s and us per step of each, whether the resident kernel was on the device, and the packet counts (which must agree)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lora_sdr_amd as L
from lora_sdr_amd import workloads as WL

ap = argparse.ArgumentParser()
ap.add_argument("--sf", type=int, default=7)
ap.add_argument("--channels", type=int, default=None)
ap.add_argument("--chunk", type=int, default=8)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--signals", action="store_true")
ap.add_argument("--frames", type=int, default=4, help="frames per channel: the length of the capture (4 = bench.py's level-3 workload, 37 steps of 8 windows)")
ap.add_argument("--depth", type=int, default=1, help="resident: steps in flight (depth + 1 sets of rows)")
a = ap.parse_args()
sf = a.sf
B = a.channels or WL.LEVEL3_CHANNELS[sf]
frames, nsyms = a.frames, 48
ctx = L.Context(sf)
iq, data = WL.frame_streams(ctx, B, frames, nsyms, sigma=0.05)
torch.cuda.synchronize()
cap = int(iq.shape[1])
d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(nsyms)
rows = [d.receiver_rows(cap_packets=B * (frames + 1), stride=nsyms) for _ in range(max(2, a.depth + 1))]
if a.signals:
    d.set_signals(True); d.receiver_signal_rows(B * (frames + 2))
chunk = a.chunk << sf


def one(mode):
    d.clear_packets(); d.rewind(); d.activate()
    torch.cuda.synchronize()
    w = pk = calls = k = 0
    was = False
    t0 = time.perf_counter()
    while w < cap:
        w = min(cap, w + chunk)
        r_ = rows[k % (a.depth + 1)] if mode == 3 else rows[k & 1]
        n, c = d.receive(iq, w, r_, async_=mode, order_with_torch=False, depth=a.depth) if mode in (2, 3) else d.receive(iq, w, r_, async_=True)
        pk += n; calls += c; k += 1
    if mode == 3:
        was = d.resident_active()
    if mode in (2, 3):
        n, c = d.receive_flush(rows[k % (a.depth + 1)] if mode == 3 else rows[k & 1]); pk += n; calls += c
    torch.cuda.synchronize()
    return time.perf_counter() - t0, pk, calls, k, was


for name, mode in (("sequential", 1), ("pipelined", 2), ("resident", 3)):
    try:
        one(mode)
        best = min((one(mode) for _ in range(a.reps)), key=lambda r: r[0])
        print("SF%d %d ch, %d-window steps, %-10s %8.1f Msym/s  %.4f of the roofline  %7.1f us/step  packets %d calls %d steps %d%s" % (
            sf, B, a.chunk, name, best[2] / best[0] / 1e6, best[2] * L.bytes_per_symbol(sf) / best[0] / 8e12, best[0] / best[3] * 1e6, best[1], best[2], best[3],
            "  kernel resident: %s" % best[4] if mode == 3 else ""), flush=True)
    except Exception as e:
        print("SF%d %-10s FAILED: %s" % (sf, name, str(e)[:200]), flush=True)
        try:
            d.receive_flush(None)
        except Exception:
            pass
