"""TestLoopback.cpp:66-133 through the HIP chain for a range of noise seeds and both readings of the noise amplitude: which seeds the
REFERENCE chain itself decodes (tests/test_gpu_codec.py::test_loopback_at_the_reference_parameters pins one).
    python tools/loopback_probe.py [seeds...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle.oracle import Ref
from test_gpu_codec import loopback_case
ref = Ref()
seeds = [int(x) for x in sys.argv[1:]] or list(range(20, 30))
for cr in ("4/7", "4/8"):
    for sigma in (4.0, 4.0 / 2 ** 0.5):
        for seed in seeds:
            sent, hp, rp, hb, rb = loopback_case(torch, ref, cr, sigma, seed)
            same_pk = len(hp) == len(rp) and all(np.array_equal(a, b) for a, b in zip(hp, rp))
            same_b = len(hb) == len(rb) and all(np.array_equal(a, b) for a, b in zip(hb, rb))
            ok_ref = len(rb) == len(sent) and all(np.array_equal(a, b) for a, b in zip(rb, sent))
            print("CR %s sigma %.3f seed %d: packets hip/ref %d/%d identical %s, bytes identical %s, reference chain == sent %s (%d of %d messages)"
                  % (cr, sigma, seed, len(hp), len(rp), same_pk, same_b, ok_ref, sum(any(np.array_equal(a, b) for b in sent) for a in rb), len(sent)), flush=True)
