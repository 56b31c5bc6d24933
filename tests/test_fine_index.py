"""CPU: the two facts lorahip_fine.h rests on, checked on the host with the library's own __host__ __device__ code.

(1) the fp64 factor tables reproduce every entry of the reference's fine-tune table (LoRaDemod.cpp:108-114);
(2) the closed-form index sequence equals the reference's int <- float recurrence (LoRaDemod.cpp:160-162) wherever the
    library claims it does (path == 1), and the library falls back to the serial chain (path == 0) elsewhere.
The oracle for (2) is the recurrence written out in numpy float32 arithmetic, i.e. the statement of the reference itself."""
import ctypes as C

import numpy as np
import pytest

import lora_sdr_amd as L


def recurrence(sf, idx0, err):
    """LoRaDemod.cpp:160-162 for one window: indices used by samples 0..N-1 and the index afterwards"""
    N = 1 << sf
    M = N * 128
    d = np.float32(err) * np.float32(128)
    out = np.empty(N, np.int32)
    idx = int(idx0)
    for n in range(N):
        out[n] = idx
        idx = int(np.float32(idx) - d)          # float subtract, truncation towards zero
        if idx < 0:
            idx += M
        elif idx >= M:
            idx -= M
    return out, idx


def lib_indices(lib, sf, idx0, err):
    N = 1 << sf
    out = np.empty(N, np.int32)
    end, path = C.c_int32(), C.c_int32()
    rc = lib.lorahip_fine_indices_host(sf, int(idx0), float(err), out.ctypes.data, C.byref(end), C.byref(path))
    assert rc == 0, "closed-form routes disagree or bad arguments (rc %d)" % rc
    return out, end.value, path.value


@pytest.mark.parametrize("sf", range(6, 13))
def test_split_tables_reproduce_the_fine_tune_table(sf):
    assert L.load().lorahip_fine_split_selftest(sf) == 1


def adversarial_errs(sf, rng):
    M = (1 << sf) * 128
    e = []
    for k in (0, 1, 2, 3, 37, 127, 128, 129, 1000):
        for eps in (0.0, 2.0 ** -30, 2.0 ** -12, 2.0 ** -8, 2.0 ** -7, 2.0 ** -6, 2.0 ** -5, 0.01, 0.4, 0.5, 0.99, 1 - 2.0 ** -6, 1 - 2.0 ** -7,
                    1 - 2.0 ** -12, 1 - 2.0 ** -20):
            for sgn in (1.0, -1.0):
                e.append(sgn * (k + eps) / 128.0)
    e += list(rng.uniform(-2, 2, 60)) + list(rng.uniform(-40, 40, 30)) + list(rng.uniform(-0.01, 0.01, 20))
    e += [M / 256.0 / 128.0 * 0.999, -M / 256.0 / 128.0 * 1.001, float(M), 1e-9, -1e-9]
    e += [0.3 / 128, 0.9 / 128, 0.02 / 128, 0.51 / 128]                     # 0 < d < 1: the index walks down to 0 and stays there
    return [np.float32(x) for x in e]


@pytest.mark.parametrize("sf", [6, 7, 9, 12])
def test_closed_form_indices_equal_the_recurrence(sf):
    lib = L.load()
    rng = np.random.default_rng(sf)
    N = 1 << sf
    M = N * 128
    seen = {0: 0, 1: 0}
    errs = adversarial_errs(sf, rng)
    if sf == 12:
        errs = errs[::3]
    for err in errs:
        d = float(np.float32(err) * np.float32(128))
        starts = [0, 1, M - 1, M // 2, int(rng.integers(0, M)), int(rng.integers(0, M)), N // 3, N - 1, N, N + 1]
        c = int(np.ceil(abs(d)))
        # starts that reach the special value ceil(d)-1 (the reference yields 0 there instead of wrapping)
        if 0 < c < M:
            starts += [c - 1, (c - 1 + 5 * c) % M, (2 * c - 1) % M]
        for idx0 in starts:
            if abs(d) >= M:                      # reference UB (index leaves the table); the library must only not claim a closed form
                _, _, path = lib_indices(lib, sf, idx0, err)
                assert path == 0
                continue
            want, want_end = recurrence(sf, idx0, err)
            got, got_end, path = lib_indices(lib, sf, idx0, err)
            seen[path] += 1
            assert np.array_equal(got, want), "sf %d err %r idx0 %d path %d" % (sf, err, idx0, path)
            assert got_end == want_end, "end index: sf %d err %r idx0 %d path %d" % (sf, err, idx0, path)
    assert seen[1] > seen[0] > 0, seen         # both paths exercised (the list is adversarial: random steps take the chain ~1 % of the time)
