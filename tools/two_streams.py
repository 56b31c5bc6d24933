"""Back-to-back launches of the steady-state batch on ONE stream against the same launches alternating over TWO contexts with their
own streams (the tail of one launch overlaps the head of the next):   python tools/two_streams.py [sf ...] [--moving]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import lora_sdr_amd as L
from lora_sdr_amd import workloads as WL
class A: gpus = 1
env = bench.Env(A())
moving = "--moving" in sys.argv
for sf in [int(x) for x in sys.argv[1:] if not x.startswith("-")] or [7, 12]:
    B, S = WL.default_geometry(sf)
    sh = bench.Shape(env, L, sf, B, S, 0.05)
    fe = fi = None
    if moving: fe, fi = sh.moving_inputs()
    ctxs, batches = [], []
    for k in range(2):
        c = L.Context(sf, device=env.local)               # private stream each
        o = sh.new_out()
        ctxs.append(c); batches.append((c.make_batch(sh.iq, sh.W, o["sym"], o["power"], o["powerAvg"], o["fIndex"], chirp_sel_all=L.CHIRP_UP, fine_err=fe, fine_idx0=fi), o))
    K = 200
    def run(two):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(K):
            i = k & 1 if two else 0
            ctxs[i].detect_batch_raw(batches[i][0])
        for c in ctxs: c.synchronize()
        return time.perf_counter() - t0
    for _ in range(2): run(False); run(True)
    one = min(run(False) for _ in range(3)); two = min(run(True) for _ in range(3))
    f = lambda t: sh.W * K * L.bytes_per_symbol(sf) / t / 8e12
    same = all(torch.equal(batches[0][1][k], batches[1][1][k]) for k in ("sym", "power", "powerAvg", "fIndex"))
    print("SF%d %s: one stream %.1f us per launch (frac %.4f); two streams alternating %.1f us (frac %.4f): %+.1f %%; outputs equal: %s" %
          (sf, "moving" if moving else "steady", one / K * 1e6, f(one), two / K * 1e6, f(two), (one / two - 1) * 100, same), flush=True)
