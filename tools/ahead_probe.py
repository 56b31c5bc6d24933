"""One launch of the streaming kernel per lanes choice on the level-3 workload, for the timing build (python tools/build_variant.py timing
-DLORAHIP_STREAM_TIMING lorahip_stream.hip lorahip_stream_lanes.hip lorahip_stream_pairs.hip; LORAHIP_LIB=lora_sdr_amd/liblorahip_timing.so):
one wavefront's s_memtime ticks per section of a call, its calls and its passes -- how many calls the AHEAD instances (16 | l) make in the
pass of the call before.   python tools/ahead_probe.py [--sf 7] [--channels 1024] [--lanes -1 19 ...] [--sigma 0.05] [--thresh -30]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lora_sdr_amd as L
from lora_sdr_amd import workloads as WL

ap = argparse.ArgumentParser()
ap.add_argument("--sf", type=int, default=7)
ap.add_argument("--channels", type=int, default=1024)
ap.add_argument("--lanes", type=int, nargs="+", default=[-1, 19, 4, 20, 5, 21])
ap.add_argument("--sigma", type=float, default=0.05)
ap.add_argument("--thresh", type=float, default=None)
a = ap.parse_args()
ctx = L.Context(a.sf)
iq, _ = WL.frame_streams(ctx, a.channels, 4, 48, sigma=a.sigma)
for lanes in a.lanes:
    d = L.LoRaDemod(a.sf, n_channels=a.channels); d.set_mode(1); d.setMTU(48); d.set_stream_lanes(lanes)
    if a.thresh is not None: d.setThreshold(a.thresh)
    for _ in range(2):
        d.clear_packets(); d.activate(); d.work(iq)
        torch.cuda.synchronize()
    print("SF%d %d channels lanes %d (runs on %d): %d calls, %d packets, kernel %.3f ms" % (a.sf, a.channels, lanes, d.stream_lanes(), d.work_calls(), len(d.packets_arrays()[0]), d.kernel_ms()), flush=True)
    d.close()
ctx.close()
