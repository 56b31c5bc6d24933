"""When and where the wavefronts of a batch launch (detectFast, SF6-10) ran: start, end of set-up, end, sets walked.
Needs the profiling build:   python tools/build_variant.py tl -DLORAHIP_WG_TIMELINE lorahip_fast.hip lorahip_wide.hip
    LORAHIP_LIB=lora_sdr_amd/liblorahip_tl.so python tools/wave_timeline.py <sf> [moving]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import lora_sdr_amd as L
from lora_sdr_amd import _lib, workloads as WL

sf = int(sys.argv[1]); moving = len(sys.argv) > 2 and sys.argv[2] == "moving"
class A: gpus = 1
env = bench.Env(A())
B, S = WL.default_geometry(sf)
sh = bench.Shape(env, L, sf, B, S, 0.05)
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.lorahip_debug_wave_timeline.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
sh.measure(30, 5, 0.2, moving=moving)
for p in range(3):
    el, km = sh.measure(20, 2, 0.0, moving=moving)
    print("launch %.1f us" % (km * 1e3 / 20))
    tl = np.zeros((16384, 4), dtype=np.uint64)
    assert lib.lorahip_debug_wave_timeline(tl.ctypes.data, tl.nbytes) == 0
    tl = tl[tl[:, 1] != 0]
    if len(tl) == 0:
        raise SystemExit("no records: the build lacks -DLORAHIP_WG_TIMELINE, or this SF runs lorahip_wide.hip's kernels (SF11 / SF12), which are not instrumented")
    t0, t1 = tl[:, 0].astype(np.float64) * 0.01, tl[:, 1].astype(np.float64) * 0.01
    setup = (tl[:, 3] >> np.uint64(32)).astype(np.float64) * 0.01
    sets = (tl[:, 3] & np.uint64(0xffffffff)).astype(np.int64)
    org = t0.min(); t0 -= org; t1 -= org
    slot = (tl[:, 2] & np.uint64(0xf)).astype(np.int64)
    print("pass %d SF%d %s: %d wavefronts, span %.1f us; start: median %.1f p99 %.1f max %.1f; set-up median %.1f us; end: p1 %.1f p10 %.1f median %.1f p90 %.1f max %.1f; sets per wavefront %d..%d" %
          (p, sf, "moving" if moving else "steady", len(tl), t1.max(), np.median(t0), np.percentile(t0, 99), t0.max(), np.median(setup), np.percentile(t1, 1), np.percentile(t1, 10),
           np.median(t1), np.percentile(t1, 90), t1.max(), sets.min(), sets.max()))
    for s_ in np.unique(slot):
        m = slot == s_
        print("    slot %d: n %d, end median %.1f (min %.1f max %.1f)" % (s_, m.sum(), np.median(t1[m]), t1[m].min(), t1[m].max()))
    xcc = ((tl[:, 2] >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
    hw = (tl[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
    unit = ((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 0xf)
    print("    end by XCC (median):", [round(float(np.median(t1[xcc == x])), 1) for x in np.unique(xcc)])
    um = np.array([np.median(t1[unit == u]) for u in np.unique(unit)])
    print("    end by compute unit (median of its wavefronts): min %.1f p10 %.1f median %.1f p90 %.1f max %.1f;  spread inside a unit (max - min), median %.1f" %
          (um.min(), np.percentile(um, 10), np.median(um), np.percentile(um, 90), um.max(), np.median([t1[unit == u].max() - t1[unit == u].min() for u in np.unique(unit)])))
    ts = np.linspace(0, t1.max(), 21)
    print("    resident wavefronts over the span:", [int(((t0 <= x) & (t1 > x)).sum()) for x in ts])
