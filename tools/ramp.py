"""Launch duration over time (clock ramp / power transients): python tools/ramp.py --sf 7 [--chunks 60 --per 50]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lora_sdr_amd as L
from bench import default_geometry
ap = argparse.ArgumentParser()
ap.add_argument("--sf", type=int, default=7); ap.add_argument("--chunks", type=int, default=60); ap.add_argument("--per", type=int, default=50)
ap.add_argument("--variant", type=int, default=0); ap.add_argument("--sleep", type=float, default=0.0)
a = ap.parse_args()
sf = a.sf; N = 1 << sf
B, S = default_geometry(sf); W = B * S
ctx = L.Context(sf); ctx.set_variant(a.variant); ctx.use_torch_stream()
sym = torch.randint(0, N, (W,), device="cuda", dtype=torch.int32).to(torch.int16)
iq = ctx.synth_symbols(sym, ampl=1.0, noise_sigma=0.5, seed=1)
out = [torch.empty(W, dtype=torch.int16, device="cuda")] + [torch.empty(W, dtype=torch.float32, device="cuda") for _ in range(3)]
b = ctx.make_batch(iq, W, *out, chirp_sel_all=L.CHIRP_UP)
torch.cuda.synchronize()
if a.sleep: time.sleep(a.sleep)
t0 = time.perf_counter(); res = []
for c in range(a.chunks):
    ctx.timer_start()
    for _ in range(a.per): ctx.detect_batch_raw(b)
    ms = ctx.timer_stop()
    res.append((time.perf_counter() - t0, ms / a.per * 1e3))
print("SF%d variant %d: t[ms] -> launch us" % (sf, a.variant))
print(" ".join("%.0f:%.0f" % (t * 1e3, us) for t, us in res))
