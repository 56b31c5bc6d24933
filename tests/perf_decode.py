"""Batched decoder throughput (packets/s) against the verbatim LoRaDecoder.cpp on one host core.
    python tests/perf_decode.py [--packets 262144]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lora_sdr_amd as L
from oracle.oracle import Ref                # CPU baseline (tests/ may use the oracle)

ap = argparse.ArgumentParser(); ap.add_argument("--packets", type=int, default=262144); a = ap.parse_args()
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "codec_kat.npz"))
for sf, rdd in ((7, 4), (10, 4), (12, 1)):
    i = next(i for i in range(int(g["count"])) if tuple(int(v) for v in g["cfg_%d" % i][:4]) == (sf, 0, rdd, 1) and int(g["cfg_%d" % i][6]) == 1
             and int(g["cfg_%d" % i][7]) == 0 and int(g["res_%d" % i][2]) == 0)
    syms, data = g["syms_%d" % i], g["data_%d" % i]
    P = a.packets
    dec = L.LoRaDecoder(); dec.setSpreadFactor(sf); dec.setCodingRate({4: "4/8", 1: "4/5"}[rdd]); dec.enableCrcc(True)
    dev_syms = torch.from_numpy(np.tile(syms.astype(np.int16), (P, 1))).cuda()
    n = torch.full((P,), len(syms), dtype=torch.int32, device="cuda")
    for _ in range(3): out, out_len, dropped = dec.decode_batch(dev_syms, n)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): out, out_len, dropped = dec.decode_batch(dev_syms, n)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    ok = bool((out_len == len(data)).all()) and np.array_equal(out[0, :len(data)].cpu().numpy(), data)
    cr = {4: "4/8", 1: "4/5"}[rdd]
    reps = 400000
    cpu = reps / Ref().decode_bench(sf, syms, reps, cr=cr, crcc=True) if Ref.available() else float("nan")
    print("SF%d CR %s, %d symbols -> %d bytes: GPU %.2f Mpackets/s (%.1f Msym/s, %d packets per launch, all correct: %s); verbatim LoRaDecoder.cpp on 1 host core %.3f Mpackets/s"
          % (sf, cr, len(syms), len(data), P / dt / 1e6, P * len(syms) / dt / 1e6, P, ok, cpu / 1e6))
