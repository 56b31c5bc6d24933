"""What would ORDERING the parts of a mixed pass give? BASELINE configs[3] as receivers (bench.py's mixed_level3 workload) through the mixed
handle as shipped -- all parts at once -- against the same parts started from Python threads with delays: the long windows' parts first, the
short ones so many ms later (their chains end the pass instead of the SF12 part's). An experiment on the product's own part handles
(lorahip_demod_part_handle), not a product path.   python tools/mixed_order_probe.py [delay_ms_sf11 delay_ms_sf10 delay_ms_rest] ..."""
import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lora_sdr_amd as L
from lora_sdr_amd import workloads as WL

sys.setswitchinterval(2e-5)
n_channels, frames, nsyms = 16384, 1, 16
sfs = WL.mixed_sf_channels(n_channels)
parts, first, cnt, at = [], np.zeros(n_channels, np.int64), np.zeros(n_channels, np.uint64), 0
for sf in range(7, 13):
    local = np.nonzero(sfs == sf)[0]
    ctx = L.Context(sf, device=0)
    iq, _ = WL.frame_streams(ctx, local.size, frames, nsyms, sigma=0.05, seed=3 + sf)
    ctx.close()
    n = int(iq.shape[1])
    first[local] = at + np.arange(local.size, dtype=np.int64) * n; cnt[local] = n; at += local.size * n
    parts.append(iq.reshape(-1))
buf = torch.cat(parts); del parts
d = L.LoRaDemod(channel_sf=sfs, devices=[0]); d.setMTU(nsyms)
lib = d._lib
d.work_segments(buf, first, cnt)
want = d.work_calls(); pk = len(d.packets_arrays()[0])
ph, pf, pc = [], [], []
for i, (_dev, sf, nch, _slot) in enumerate(d.parts):
    sel = np.nonzero(d.part_of == i)[0]
    sel = sel[np.argsort(d.local_of[sel])]
    ph.append(C.c_void_p(lib.lorahip_demod_part_handle(d._h, i)))
    pf.append(np.ascontiguousarray(first[sel])); pc.append(np.ascontiguousarray(cnt[sel].astype(np.uint64)))
ptr = C.c_void_p(buf.data_ptr())

def run_part(i, t_start, delay, out):
    while True:
        left = t_start + delay - time.perf_counter()
        if left <= 0: break
        time.sleep(left if left > 2e-4 else 0)            # (sleeping releases the interpreter lock: the other threads' calls start on time)
    r = C.c_int64()
    out[i] = lib.lorahip_demod_run_device_segments(ph[i], ptr, pf[i].ctypes.data_as(C.POINTER(C.c_int64)), pc[i].ctypes.data_as(C.POINTER(C.c_size_t)), C.byref(r))

def ordered(delays):
    """delays: seconds per SF after the SF12 part's start"""
    d.clear_packets(); d.activate(); torch.cuda.synchronize()
    out = [None] * len(ph)
    t0 = time.perf_counter() + 0.002
    th = [threading.Thread(target=run_part, args=(i, t0, delays.get(d.parts[i][1], 0.0), out)) for i in range(len(ph))]
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    assert all(o == 0 for o in out), out
    return dt

def shipped():
    d.clear_packets(); d.activate(); torch.cuda.synchronize()
    t0 = time.perf_counter(); d.work_segments(buf, first, cnt); return time.perf_counter() - t0

t_r = time.perf_counter()
while time.perf_counter() - t_r < 0.3: shipped()
plans = [(0, 0, 0)] + [tuple(float(x) for x in sys.argv[k:k + 3]) for k in range(1, len(sys.argv) - 2, 3)]
for rep in range(3):
    print("shipped (the handle, all parts at once): %.3f ms" % (min(shipped() for _ in range(4)) * 1e3), flush=True)
    for a, b, c in plans:
        dl = {11: a * 1e-3, 10: b * 1e-3, 9: c * 1e-3, 8: c * 1e-3, 7: c * 1e-3}
        ts = [ordered(dl) for _ in range(4)]
        calls = d.work_calls()
        print("  threads, SF11 +%.1f ms, SF10 +%.1f ms, SF7-9 +%.1f ms: %.3f ms (best of 4; packets %d of %d)" % (a, b, c, min(ts) * 1e3, len(d.packets_arrays()[0]), pk), flush=True)
d.close()
