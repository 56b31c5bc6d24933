"""Where the host wall clock of one level-3 pass goes (LORAHIP_DEMOD_TIMING=1 prints the library's own breakdown to stderr):
    LORAHIP_DEMOD_TIMING=1 python tools/e2e_breakdown.py --sf 7"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lora_sdr_amd as L
from lora_sdr_amd import workloads as WL
ap = argparse.ArgumentParser(); ap.add_argument("--sf", type=int, default=7); a = ap.parse_args()
sf = a.sf; B = WL.LEVEL3_CHANNELS[sf]
ctx = L.Context(sf)
iq, data = WL.frame_streams(ctx, B, 4, 48, sigma=0.05)
d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(48)
d.work(iq); d.packets_device(clear=False)
for rep in range(6):
    d.clear_packets(); d.activate(); torch.cuda.synchronize()
    t0 = time.perf_counter(); d.work(iq); t1 = time.perf_counter(); d.packets_device(clear=False); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("pass %d: work() %.3f ms (kernel %.3f ms), packets_device %.3f ms" % (rep, (t1 - t0) * 1e3, d.kernel_ms(), (t2 - t1) * 1e3), flush=True)
