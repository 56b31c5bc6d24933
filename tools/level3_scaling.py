"""The streaming kernels against the channel count, per lanes-per-channel choice (lorahip_demod_set_stream_lanes): kernel time (HIP
events around the launch; best and median of the passes) of the level-3 workload -- whole LoRaDemod blocks, 4 frames of 48 data
symbols per channel, staggered starts.   python tools/level3_scaling.py [--sf 7 8 9] [--lanes 0 -1 4 5 6] [--counts ...]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lora_sdr_amd as L
from lora_sdr_amd import workloads as WL

ap = argparse.ArgumentParser()
ap.add_argument("--sf", type=int, nargs="+", default=[7, 8, 9])
ap.add_argument("--lanes", type=int, nargs="+", default=[-1, 4, 5, 6, 19, 20, 21, 0])   # 16 | l: the AHEAD instances (lorahip_stream_pairs.hip)
ap.add_argument("--counts", type=int, nargs="+", default=None)
ap.add_argument("--passes", type=int, default=5)
a = ap.parse_args()
DEF = {7: (1024, 2048, 4096, 8192, 12288, 16384), 8: (512, 1024, 2048, 4096, 8192), 9: (256, 512, 1024, 2048, 4096), 10: (512, 1024, 2048),
       11: (256, 512, 1024, 2048), 12: (128, 256, 512, 1024)}
AVAIL = {7: (4, 5, 19, 20, 21), 8: (5, 6, 20, 21), 9: (6, 21)}
for sf in a.sf:
    ctx = L.Context(sf)
    for B in (a.counts or DEF[sf]):
        iq, _ = WL.frame_streams(ctx, B, 4, 48, sigma=0.05)
        row = []
        for lanes in a.lanes:
            if lanes > 0 and lanes not in AVAIL.get(sf, ()):
                continue
            d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(48); d.set_stream_lanes(lanes)
            d.work(iq)
            calls, npk = d.work_calls(), len(d.packets_arrays()[0])
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.12:
                d.clear_packets(); d.activate(); d.work(iq)
            kms = []
            for _ in range(a.passes):
                d.clear_packets(); d.activate(); d.work(iq); kms.append(d.kernel_ms())
            d.close()
            kms.sort()
            frac = calls * L.bytes_per_symbol(sf) / (kms[0] / 1e3) / 8e12
            row.append("lanes %2d: %.3f ms (median %.3f) frac %.3f" % (lanes, kms[0], kms[len(kms) // 2], frac))
        print("SF%d %5d channels (%d calls, %d packets) | %s" % (sf, B, calls, npk, " | ".join(row)), flush=True)
        del iq
        torch.cuda.empty_cache()
    ctx.close()
