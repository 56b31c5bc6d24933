"""Probe (GPU box): level 2 against the oracle on inputs at the edges of fp32 -- amplitudes whose squares are subnormal or
overflow, and windows that hold a NaN or an infinite sample. Prints, per case and SF, how many symbol indices / FFT bins /
dechirped samples differ bit for bit (NaN payloads ignored) and whether power / powerAvg / fIndex agree in finiteness."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lora_sdr_amd as L
from oracle.oracle import Oracle


def bits_differ(a, b):
    a = np.ascontiguousarray(a).view(np.float32).ravel(); b = np.ascontiguousarray(b).view(np.float32).ravel()
    both_nan = np.isnan(a) & np.isnan(b)
    return int(np.count_nonzero((a.view(np.uint32) != b.view(np.uint32)) & ~both_nan))


def main():
    orc = Oracle()
    rng = np.random.default_rng(5)
    for sf in (7, 9, 10, 11, 12):
        N = 1 << sf
        W = 48
        t = np.arange(N)
        down = L.host_tables(sf, fine=False)[1].astype(np.complex128)
        sym = rng.integers(0, N, W)
        base = down[None, :] * np.exp(2j * np.pi * sym[:, None] * t[None, :] / N)
        base = base + 0.1 * (rng.standard_normal((W, N)) + 1j * rng.standard_normal((W, N)))
        cases = {}
        for name, scale in (("1e-19", 1e-19), ("1e-21", 1e-21), ("1e-23", 1e-23), ("3e-39", 3e-39), ("1e17", 1e17), ("1e18", 1e18), ("3e19", 3e19), ("3e37", 3e37)):
            with np.errstate(over="ignore"):
                cases["scale " + name] = (base * scale).astype(np.complex64)
        x = base.astype(np.complex64).copy(); x[::3, 5] = np.nan; cases["one NaN sample in every third window"] = x
        x = base.astype(np.complex64).copy(); x[::3, N // 2] = np.inf; cases["one +Inf (real) sample"] = x
        x = base.astype(np.complex64).copy(); x[::3, 7] = complex(0.0, -np.inf); cases["one -Inf (imag) sample"] = x
        x = base.astype(np.complex64).copy(); x[::2, :] = 0; x[::2, 3] = 1e-30; cases["a lone 1e-30 sample"] = x
        ctx = L.Context(sf)
        for err in (0.0, 0.31):
            for name, iq in cases.items():
                fe = np.full(W, err, np.float32)
                g = ctx.detect_batch(torch.from_numpy(iq).cuda(), fine_err=torch.from_numpy(fe).cuda() if err else None, want_fft=True, want_dec=True)
                torch.cuda.synchronize()
                with np.errstate(all="ignore"):
                    o = orc.detect_batch(sf, iq, fine_err=fe if err else None, want_fft=True, want_dec=True)
                gs = g["sym"].cpu().numpy().view(np.uint16)
                nsym = int(np.count_nonzero(gs != o["sym"]))
                nfft = bits_differ(g["fft"].cpu().numpy(), o["fft"])
                ndec = bits_differ(g["dec"].cpu().numpy(), o["dec"])
                fin = []
                for k in ("power", "powerAvg", "fIndex"):
                    a, b = g[k].cpu().numpy(), o[k]
                    same_class = np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.isposinf(a), np.isposinf(b)) and np.array_equal(np.isneginf(a), np.isneginf(b))
                    f = np.isfinite(a) & np.isfinite(b)
                    fin.append("%s %s max|d| %.2g" % (k, "class ok" if same_class else "CLASS DIFFERS", float(np.abs(a[f] - b[f]).max()) if f.any() else 0.0))
                flag = "" if (nsym == 0 and nfft == 0 and ndec == 0 and "DIFFERS" not in " ".join(fin)) else "   <<<<"
                print("SF%-2d err %.2f %-40s sym %3d  fft bins %6d  dec %6d  %s%s" % (sf, err, name, nsym, nfft, ndec, "; ".join(fin), flag), flush=True)


if __name__ == "__main__":
    main()
