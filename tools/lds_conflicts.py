"""LDS bank-conflict model for the exchange layouts of lorahip_fast.hip (MI355X_MICROARCH.md §LDS).

ds_write_b64: contiguous 16-lane groups, bank = (addr/4) % 32;  ds_read_b64: two 32-lane groups,
bank = (addr/4) % 64. Cost of one wave-instruction = sum over groups of the largest number of DISTINCT
8-byte words that map onto one bank pair. Usage: python tools/lds_conflicts.py
"""
import itertools


def rev4(x, bits):
    r = 0
    for _ in range(0, bits, 2):
        r = (r << 2) | (x & 3)
        x >>= 2
    return r


def cost(addrs, kind):
    """addrs: list of 64 byte addresses (8-byte accesses)"""
    if kind == "w":
        groups, nb = [range(g * 16, g * 16 + 16) for g in range(4)], 32
    else:
        groups, nb = [range(0, 32), range(32, 64)], 64
    tot = 0
    for g in groups:
        per = {}
        for l in g:
            a = addrs[l]
            for b in ((a // 4) % nb, (a // 4 + 1) % nb):
                per.setdefault(b, set()).add(a // 8)
        tot += max(len(v) for v in per.values())
    return tot, len(groups)


def exch0(LOG2N, LOG2T, VEC, B1, B2, rowmap, rowlen_pad=1, verbose=False):
    """exchange 0 of a config; rowmap(t,u)->row. layout [row][ws][col]"""
    N, T = 1 << LOG2N, 1 << LOG2T
    R = 1 << B1
    WPW = 64 // T
    rowlen = WPW * R + rowlen_pad
    wr = []
    for u in range(VEC):
        for e in range(R):
            addrs = [8 * (rowmap(l % T, u) * rowlen + (l // T) * R + e) for l in range(64)]
            wr.append(cost(addrs, "w"))
    rd = []
    GL = 1 << (B2 - B1)
    HB = LOG2N - B2
    P = N // T
    NG1 = P // GL
    for g in range(NG1):
        for e in range(GL):
            addrs = []
            for l in range(64):
                t, ws = l % T, l // T
                ci = t + T * g
                klow, high = ci & (R - 1), ci >> B1
                nlow = (rev4(e, B2 - B1) << HB) | rev4(high, HB)
                tt, uu = (nlow // VEC, nlow % VEC)
                addrs.append(8 * (rowmap(tt, uu) * rowlen + ws * R + klow))
            rd.append(cost(addrs, "r"))
    w = sum(c for c, _ in wr) / sum(n for _, n in wr)
    r = sum(c for c, _ in rd) / sum(n for _, n in rd)
    return w, r


if __name__ == "__main__":
    cfgs = {"sf7": (7, 3, 2, 3, 7), "sf8": (8, 3, 2, 4, 8), "sf9": (9, 5, 2, 3, 7), "sf10": (10, 5, 2, 4, 8)}
    for name, (n, t, vec, b1, b2) in cfgs.items():
        T = 1 << t
        for rm_name, rm in (("n_low", lambda tt, uu: vec * tt + uu), ("u-major", lambda tt, uu: tt + T * uu),
                            ("u-major+1", lambda tt, uu: tt + (T + 1) * uu)):
            for pad in (0, 1, 2, 3):
                w, r = exch0(n, t, vec, b1, b2, rm, pad)
                print("%-5s rows=%-10s pad=%d  write x%.2f  read x%.2f" % (name, rm_name, pad, w, r))


def exch0_general(LOG2N, LOG2T, VEC, B1, B2, off, wsstride):
    """off(n_low) -> element offset of the row start; element addr = off + ws*wsstride + col"""
    N, T = 1 << LOG2N, 1 << LOG2T
    R = 1 << B1
    wr = []
    for u in range(VEC):
        for e in range(R):
            addrs = [8 * (off(VEC * (l % T) + u) + (l // T) * wsstride + e) for l in range(64)]
            wr.append(cost(addrs, "w"))
    rd = []
    GL = 1 << (B2 - B1)
    HB = LOG2N - B2
    NG1 = (N // T) // GL
    for g in range(NG1):
        for e in range(GL):
            addrs = []
            for l in range(64):
                t, ws = l % T, l // T
                ci = t + T * g
                klow, high = ci & (R - 1), ci >> B1
                nlow = (rev4(e, B2 - B1) << HB) | rev4(high, HB)
                addrs.append(8 * (off(nlow) + ws * wsstride + klow))
            rd.append(cost(addrs, "r"))
    return (sum(c for c, _ in wr) / sum(n for _, n in wr), sum(c for c, _ in rd) / sum(n for _, n in rd))


def search(name, n, t, vec, b1, b2):
    T, R = 1 << t, 1 << b1
    WPW = 64 // T
    NL = vec * T
    best = []
    # off(n_low) = perm(n_low) * RS where perm swaps/rotates bit fields, RS = WPW*R + pad; ws stride R
    import itertools
    nb = NL.bit_length() - 1
    for rot in range(nb):
        for pad in range(0, 9):
            for extra_s in range(nb):
                for extra_d in (0, 1, 2, 4, 8, 16, 32):
                    RS = WPW * R + pad

                    def off(x, rot=rot, RS=RS, extra_s=extra_s, extra_d=extra_d):
                        p = ((x >> rot) | (x << (nb - rot))) & (NL - 1)
                        return p * RS + ((x >> extra_s) & 1) * extra_d
                    w, r = exch0_general(n, t, vec, b1, b2, off, R)
                    size = NL * RS + 64
                    best.append((w + r, w, r, rot, pad, extra_s, extra_d, size))
    best.sort()
    print(name, "best (sum, w, r, rot, pad, extra_s, extra_d, size):")
    for b in best[:5]:
        print("   ", b)


if __name__ == "__main__":
    for name, c in {"sf7": (7, 3, 2, 3, 7), "sf8": (8, 3, 2, 4, 8), "sf9": (9, 5, 2, 3, 7), "sf10": (10, 5, 2, 4, 8)}.items():
        search(name, *c)


def exch0_wide(LOG2N, VEC, off):
    """wide kernels (lorahip_wide.hip): T = N/16 lanes per window (whole wavefronts), one X region per window.
    off(n_low) -> element offset of the row start."""
    N = 1 << LOG2N
    T = N // 16
    R = 16 // VEC
    B1 = 4 if VEC == 1 else 3
    HB = 4
    wr, rd = [], []
    for wv in range(T // 64):
        for u in range(VEC):
            for e in range(R):
                wr.append(cost([8 * (off(VEC * (64 * wv + l) + u) + e) for l in range(64)], "w"))
        for e in range(16):
            addrs = []
            for l in range(64):
                t = 64 * wv + l
                klow, high = t & (R - 1), t >> B1
                nlow = (rev4(e, 4) << HB) | rev4(high, HB)
                addrs.append(8 * (off(nlow) + klow))
            rd.append(cost(addrs, "r"))
    return (sum(c for c, _ in wr) / sum(n for _, n in wr), sum(c for c, _ in rd) / sum(n for _, n in rd))


def search_wide(name, LOG2N, VEC):
    N = 1 << LOG2N
    T = N // 16
    R = 16 // VEC
    NL = VEC * T
    nb = NL.bit_length() - 1
    best = []
    for rot in range(nb):
        for pad in range(0, 9):
            for extra_s in range(nb):
                for extra_d in (0, 1, 2, 4, 8, 16, 32):
                    RS = R + pad

                    def off(x, rot=rot, RS=RS, extra_s=extra_s, extra_d=extra_d):
                        p = ((x >> rot) | (x << (nb - rot))) & (NL - 1)
                        return p * RS + ((x >> extra_s) & 1) * extra_d
                    w, r = exch0_wide(LOG2N, VEC, off)
                    best.append((w + r, w, r, rot, pad, extra_s, extra_d, NL * RS + extra_d))
    best.sort()
    print(name, "best (sum, w, r, rot, pad, extra_s, extra_d, size):")
    for b in best[:6]:
        print("   ", b)


if __name__ == "__main__":
    search_wide("sf11-wide", 11, 2)
    search_wide("sf12-wide", 12, 1)


def inplace_wide(LOG2N, VEC, off):
    """wide kernels with the in-place middle phase: exchange-0 write, exchange-0 read (= in-place write-back of
    phase 1), and the last phase's read from the SAME layout. Returns (write, read1, read2) average cost factors."""
    N = 1 << LOG2N
    T = N // 16
    R = 16 // VEC
    B1 = 4 if VEC == 1 else 3
    wr, rd1, wb, rd2 = [], [], [], []
    for wv in range(T // 64):
        for u in range(VEC):
            for e in range(R):
                wr.append(cost([8 * (off(VEC * (64 * wv + l) + u) + e) for l in range(64)], "w"))
        for e in range(16):
            addrs = []
            for l in range(64):
                t = 64 * wv + l
                klow, high = t & (R - 1), t >> B1
                addrs.append(8 * (off((rev4(e, 4) << 4) | rev4(high, 4)) + klow))
            rd1.append(cost(addrs, "r"))
            wb.append(cost(addrs, "w"))
        for e2 in range(16):
            addrs = []
            for l in range(64):
                t = 64 * wv + l
                klow, e = t & (R - 1), t >> B1
                addrs.append(8 * (off((rev4(e, 4) << 4) | rev4(e2, 4)) + klow))
            rd2.append(cost(addrs, "r"))
    f = lambda v: sum(c for c, _ in v) / sum(n for _, n in v)
    return f(wr), f(rd1), f(wb), f(rd2)


def search_inplace(name, LOG2N, VEC):
    N = 1 << LOG2N
    T = N // 16
    R = 16 // VEC
    NL = VEC * T
    nb = NL.bit_length() - 1
    best = []
    for rot in range(nb):
        for pad in range(0, 5):
            for s1 in range(nb):
                for d1 in (0, 1, 2, 4, 8, 16, 32):
                    for s2 in range(s1 + 1, nb) if d1 else [0]:
                        for d2 in ((0, 2, 4, 8, 16, 32) if d1 else (0,)):
                            RS = R + pad

                            def off(x, rot=rot, RS=RS, s1=s1, d1=d1, s2=s2, d2=d2):
                                p = ((x >> rot) | (x << (nb - rot))) & (NL - 1)
                                return p * RS + ((x >> s1) & 1) * d1 + ((x >> s2) & 1) * d2
                            w, r1, wb, r2 = inplace_wide(LOG2N, VEC, off)
                            best.append((w + r1 + wb + r2, w, r1, wb, r2, rot, pad, s1, d1, s2, d2, NL * RS + d1 + d2))
    best.sort()
    print(name, "in-place best (sum, w, r1, wb, r2, rot, pad, s1, d1, s2, d2, size):")
    for b in best[:8]:
        print("   ", b)
