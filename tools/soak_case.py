"""One case of tools/soak_level3.py by seed, verbosely, with overrides:  python tools/soak_case.py 20004 [lanes=0] [grid=0] [how=3] [sigs=0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import lora_sdr_amd as L
from oracle.oracle import Oracle
from test_gpu_demod import frames
seed = int(sys.argv[1]); ov = dict(a.split("=") for a in sys.argv[2:])
oracle = Oracle()
rng = np.random.default_rng(seed)
sf = int(rng.integers(7, 13)); N = 1 << sf
B = int(rng.integers(1, 24 if sf < 11 else 10))
mtu = int(rng.integers(3, 40)); thresh = float(rng.uniform(-40, -5)); sync = int(rng.integers(0, 256)) if rng.random() < 0.3 else 0x12
streams = []
for c in range(B):
    s, _ = frames(oracle, rng, sf, int(rng.integers(1, 4)), int(rng.integers(2, 30)), off=float(rng.uniform(-0.45, 0.45)), noise=float(rng.uniform(0.0, 0.3)), sync=sync, lead=int(rng.integers(0, 3 * N)))
    streams.append(s)
cap = max(s.size for s in streams); cap += -cap % 16
host = np.zeros((B, cap), np.complex64)
for c, s in enumerate(streams): host[c, :s.size] = s
refs = [oracle.demod_run(sf, host[c], sync=sync, thresh=thresh, mtu=mtu) for c in range(B)]
grid = int(rng.choice([0, -1, 1, 2, 5])); lanes = int(rng.choice([0, -1, 4, 5, 6])); how = int(rng.integers(0, 4)); sigs = rng.random() < 0.5
steps = []
w = 0
while w < cap:
    w = min(cap, w + int(rng.integers(N // 2, 9 * N))); steps.append(w)
grid = int(ov.get("grid", grid)); lanes = int(ov.get("lanes", lanes)); how = int(ov.get("how", how)); sigs = bool(int(ov.get("sigs", int(sigs))))
print("seed %d: SF%d B %d mtu %d thresh %.1f grid %d lanes %d how %d sigs %s; %d steps %s" % (seed, sf, B, mtu, thresh, grid, lanes, how, sigs, len(steps), steps[:12]))
iq = torch.from_numpy(host).cuda(); torch.cuda.synchronize()
d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(mtu); d.setThreshold(thresh); d.setSync(sync); d.set_stream_grid(grid); d.set_stream_lanes(lanes)
rows = [d.receiver_rows(cap_packets=B * 40, stride=max(mtu, 8)) for _ in range(2)]
d.set_signals(sigs)
pin_ = bool(int(ov.get("pin", 0)))
srows = [d.receiver_signal_rows(B * 48, pinned_host=pin_) for _ in range(2)] if sigs else None
got = [[] for _ in range(B)]
got_sig = [[] for _ in range(B)]
log = []
def take(n, r, tag, srow=None):
    if not (how == 3 and d.resident_active()): torch.cuda.synchronize()
    sy, ns, chn = r[0][:n].cpu().numpy(), r[1][:n].cpu().numpy(), r[2][:n].cpu().numpy()
    for i in range(n):
        got[int(chn[i])].append(sy[i, :ns[i]].copy()); log.append((tag, int(chn[i]), int(ns[i])))
    if sigs:
        m = d.last_signals()
        sc, se, sp, ss = (np.asarray(t_[:m].cpu() if hasattr(t_, "cpu") else t_[:m]) for t_ in srow)
        for i in range(m): got_sig[int(sc[i])].append((int(se[i]), float(sp[i]), float(ss[i]), tag))
k = 0
for w in steps:
    if sigs: d.register_signal_rows(srows[k & 1])
    n, c_ = d.receive(iq, w, rows[k & 1], async_=(how if how in (2, 3) else True))
    res = how == 3 and d.resident_active()
    j = (k - 1) & 1 if (res and k > 0) else k & 1
    take(n, rows[j], "call %d (w %d)%s" % (k, w, " resident" if res else ""), srows[j] if sigs else None); k += 1
if how in (2, 3):
    res = how == 3 and d.resident_active()
    j = (k - 1) & 1 if res else k & 1
    if sigs and not res: d.register_signal_rows(srows[k & 1])
    n, c_ = d.receive_flush(rows[k & 1]); take(n, rows[j], "flush", srows[j] if sigs else None)
bad = 0
if sigs:
    for c, r in enumerate(refs):
        g, wv = got_sig[c], r["signals"]
        if len(g) != len(wv): print("channel %d: %d signals, reference %d" % (c, len(g), len(wv))); bad += 1; continue
        for j, (a, b) in enumerate(zip(g, wv)):
            if a[0] != int(b[0]) or abs(a[1] - b[1]) > 2e-5 or abs(a[2] - b[2]) > 2e-5:
                bad += 1; print("channel %d signal %d: got error %d power %.6f snr %.6f (%s), want %d %.6f %.6f" % (c, j, a[0], a[1], a[2], a[3], int(b[0]), b[1], b[2]))
for c, r in enumerate(refs):
    if len(got[c]) != len(r["packets"]): print("channel %d: %d packets, reference %d" % (c, len(got[c]), len(r["packets"]))); bad += 1; continue
    for j, (a, (_, b)) in enumerate(zip(got[c], r["packets"])):
        if not np.array_equal(a, b):
            bad += 1
            diff = np.nonzero(a != b)[0] if a.size == b.size else []
            print("channel %d packet %d: len %d / %d, differing positions %s\n   got  %s\n   want %s" % (c, j, a.size, b.size, list(diff)[:20], a.tolist(), b.tolist()))
            print("   delivered by:", [t for t in log if t[1] == c])
print("bad", bad)
d.close()
