"""LDS bank-conflict model (as tools/lds_conflicts.py: ds_write_b64 in 16-lane groups over 32 banks, ds_read_b64 in 32-lane groups over
64 banks) for EVERY exchange of a FastCfg as lorahip_fastcore.h::fft addresses it -- exchange 0 (x0off rows), exchange 1 and, with four
phases, the in-place middle phase and the last phase's reads of the position-indexed rows. Searches the layout parameters of the
wide-lane streaming instances (lorahip_stream_lanes.hip).   python tools/lds_conflicts_lanes.py"""
import itertools


def rev4(x, bits):
    r = 0
    for _ in range(0, bits, 2):
        r = (r << 2) | (x & 3)
        x >>= 2
    return r


def cost(addrs, kind):
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


class Cfg:
    def __init__(self, LOG2N, LOG2T, VEC, NPH, PB1, PB2, PB3=0, X0ROT=0, X0PAD=1, X0S=0, X0D=0, X1PAD=8):
        self.LOG2N, self.LOG2T, self.VEC, self.NPH = LOG2N, LOG2T, VEC, NPH
        self.N, self.T = 1 << LOG2N, 1 << LOG2T
        self.P = self.N // self.T
        self.R = self.P // VEC
        assert (1 << PB1) == self.R
        self.b = [0, PB1] + ([PB2] if NPH >= 3 else []) + ([PB3] if NPH >= 4 else []) + [LOG2N]
        self.WPW = 64 // self.T
        self.NL = VEC * self.T
        self.LOG2NL = LOG2N - PB1
        self.RS0 = self.WPW * self.R + X0PAD
        self.X0ROT, self.X0S, self.X0D = X0ROT, X0S, X0D
        self.B1 = PB1
        self.B2 = self.b[2]
        self.G1 = 1 << (self.B2 - self.B1)
        self.X1 = self.G1 * self.R + X1PAD
        self.X1ROWS = self.N // (self.G1 * self.R)
        self.BL = self.b[NPH - 1]
        self.GL = 1 << (LOG2N - self.BL)
        self.NGL = self.P // self.GL

    def x0off(self, nlow):
        rot = ((nlow >> self.X0ROT) | (nlow << (self.LOG2NL - self.X0ROT))) & (self.NL - 1)
        return rot * self.RS0 + ((nlow >> self.X0S) & 1) * self.X0D

    def lanes(self):
        for l in range(64):
            yield l, l % self.T, l // self.T

    def run(self):
        """-> (sum of cycles, sum of ideal cycles) over all exchange instructions of one window set"""
        c, tot, ideal = self, 0, 0

        def add(addrs, kind):
            nonlocal tot, ideal
            a, n = cost([8 * x for x in addrs], kind)
            tot += a; ideal += n
        T, R, VEC, B1, B2 = c.T, c.R, c.VEC, c.B1, c.B2
        # exchange 0 writes
        for u in range(VEC):
            for e in range(R):
                add([ws * R + c.x0off(VEC * t + u) + e for _, t, ws in c.lanes()], "w")
        if c.NPH == 2:
            for g in range(c.NGL):
                for e in range(c.GL):
                    add([ws * R + c.x0off(rev4(e, c.LOG2N - B1)) + (t + T * g) for _, t, ws in c.lanes()], "r")
            return tot, ideal
        HB = c.LOG2N - B2
        NG1 = c.P // c.G1
        for g in range(NG1):
            for e in range(c.G1):
                a = []
                for _, t, ws in c.lanes():
                    ci = t + T * g
                    a.append(ws * R + c.x0off((rev4(e, B2 - B1) << HB) | rev4(ci >> B1, HB)) + (ci & (R - 1)))
                add(a, "r")
        win = c.X1ROWS * c.X1
        for g in range(NG1):
            for e in range(c.G1):
                add([ws * win + ((t + T * g) >> B1) * c.X1 + ((t + T * g) & (R - 1)) + e * R for _, t, ws in c.lanes()], "w")
        if c.NPH == 4:
            B3 = c.b[3]
            G2 = 1 << (B3 - B2)
            NG2 = c.P // G2
            for kind in ("r", "w"):
                for g in range(NG2):
                    for e in range(G2):
                        add([ws * win + ((t + T * g) >> B2) * (G2 * c.X1) + ((t + T * g) & ((1 << B2) - 1)) + e * c.X1 for _, t, ws in c.lanes()], kind)
        for g in range(c.NGL):
            for e in range(c.GL):
                add([ws * win + (((t + T * g) >> B2) + (e << (c.BL - B2))) * c.X1 + ((t + T * g) & ((1 << B2) - 1)) for _, t, ws in c.lanes()], "r")
        return tot, ideal


if __name__ == "__main__":
    shapes = {"Stream7L4": (7, 4, 1, 3, 3, 5, 0), "Stream7L5": (7, 5, 2, 4, 1, 3, 5), "Stream8L5": (8, 5, 2, 4, 2, 4, 6), "Stream8L6": (8, 6, 1, 4, 2, 4, 6),
              "Stream9L6": (9, 6, 1, 4, 3, 5, 7)}
    for name, (n, t, vec, nph, b1, b2, b3) in shapes.items():
        base = Cfg(n, t, vec, nph, b1, b2, b3)
        a, i = base.run()
        best = None
        lognl = n - b1
        for rot, pad, x1pad in itertools.product(range(lognl), range(0, 10), (0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 16, 17)):
            for xs, xd in ((0, 0),) + tuple((s_, d_) for s_ in range(lognl) for d_ in (4, 8, 16)):
                c = Cfg(n, t, vec, nph, b1, b2, b3, rot, pad, xs, xd, x1pad)
                # the rows must not overlap: x0off injective over (nlow, column)
                offs = sorted(c.x0off(nl) for nl in range(c.NL))
                if any(b - a_ < c.WPW * c.R for a_, b in zip(offs, offs[1:])):
                    continue
                cy, idl = c.run()
                size = max(c.NL * c.RS0 + xd, c.WPW * c.X1ROWS * c.X1)
                if best is None or (cy, size) < (best[0], best[1]):
                    best = (cy, size, rot, pad, xs, xd, x1pad)
        print("%-10s now x%.2f of ideal (%d / %d) -> best x%.2f: X0ROT %d X0PAD %d X0S %d X0D %d X1PAD %d (%d elements per wave)"
              % (name, a / i, a, i, best[0] / i, best[2], best[3], best[4], best[5], best[6], best[1]))
