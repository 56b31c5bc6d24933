"""Probe (GPU box): the 'next' rows at sizes where 32-bit products overflow -- the decoder on 6 M packets (rows x stride beyond 2^31
bytes of output), level 3 on 131 072 channels. Inputs repeat a small set of distinct
cases, so the results must repeat with the same period; the first period is checked against the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lora_sdr_amd as L
from oracle.oracle import Oracle

orc = Oracle()
rng = np.random.default_rng(11)


def decoder():
    sf, cr = 9, "4/8"
    D = 997                                              # distinct packets
    stride = 200
    base = rng.integers(0, 1 << sf, (D, stride)).astype(np.uint16)
    lens = rng.integers(8, stride + 1, D).astype(np.int32)
    P = 6_000_000                                        # out: P x 2 (stride + 8) = 2.5e9 bytes > 2^31
    idx = torch.arange(P, device="cuda") % D
    syms = torch.from_numpy(base.view(np.int16)).cuda()[idx].contiguous()
    nsyms = torch.from_numpy(lens).cuda()[idx].contiguous()
    dec = L.LoRaDecoder(); dec.setSpreadFactor(sf); dec.setCodingRate(cr); dec.enableCrcc(False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out, out_len, dropped = dec.decode_batch(syms, nsyms)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert out.numel() > 2**31
    ok_rep = bool((out_len.view(-1)[: (P // D) * D].view(P // D, D) == out_len[:D][None, :]).all())
    # rows far beyond 2^31 bytes equal their first copies
    far = torch.tensor([P - 1, P - 2, P // 2 + 3, 5_500_000], device="cuda")
    same = bool((out[far] == out[far % D]).all())
    n_bad = 0
    for i in range(0, D, 7):
        want, _ = orc.decode(sf, base[i, :lens[i]], cr=cr, crcc=False)
        got = None if int(out_len[i]) < 0 else out[i, :int(out_len[i])].cpu().numpy()
        if (got is None) != (want is None) or (got is not None and not np.array_equal(got, want)):
            n_bad += 1
    print("decoder: %d packets x stride %d (output %.2f GB) in %.1f ms; lengths repeat %s; rows beyond 2^31 bytes equal their first copies %s; %d of %d checked packets differ from the oracle"
          % (P, stride, out.numel() / 1e9, dt * 1e3, ok_rep, same, n_bad, len(range(0, D, 7))), flush=True)


def level3():
    sf, N = 7, 128
    D, B = 64, 131072
    o = orc
    streams = []
    for c in range(D):
        syms = rng.integers(0, N, 20).astype(np.uint16)
        st = np.concatenate([np.zeros(int(rng.integers(0, N)), np.complex64), o.mod_frame(sf, syms, padding=2)])
        st = (st * np.exp(2j * np.pi * rng.uniform(-0.4, 0.4) / N * np.arange(st.size))).astype(np.complex64)
        streams.append(st)
    S = max(s.size for s in streams) + 3 * N
    x = np.zeros((D, S), np.complex64)
    for c, s in enumerate(streams):
        x[c, :s.size] = s
    x += (0.05 * (rng.standard_normal(x.shape) + 1j * rng.standard_normal(x.shape))).astype(np.complex64)
    big = torch.from_numpy(x).cuda().repeat(B // D, 1).contiguous()
    d = L.LoRaDemod(sf, n_channels=B); d.set_mode(1); d.setMTU(20)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d.work(big)
    dt = time.perf_counter() - t0
    syms, nsyms, chan = d.packets_device()
    nsyms, chan = nsyms.cpu().numpy(), chan.cpu().numpy()
    refs = [orc.demod_run(sf, x[c], mtu=20) for c in range(D)]
    want_pk = sum(len(r["packets"]) for r in refs) * (B // D)
    bad = 0
    cons = np.array([d.consumed(c) for c in list(range(D)) + [B - 1, B - D, B // 2 + 5]])
    wantc = np.array([int(sum(k["consumed"] for k in refs[c % D]["calls"])) for c in list(range(D)) + [B - 1, B - D, B // 2 + 5]])
    symsh = syms.cpu().numpy().view(np.uint16)
    pos = 0
    by_chan = {}
    for i in range(len(nsyms)):
        by_chan.setdefault(int(chan[i]), []).append(symsh[i, :nsyms[i]])
    for c in list(range(D)) + [B - 1, B - D + 3, B // 2 + 5, 100_000]:
        mine, ref = by_chan.get(c, []), [q for _, q in refs[c % D]["packets"]]
        if len(mine) != len(ref) or any(not np.array_equal(a, b.view(np.uint16)) for a, b in zip(mine, ref)):
            bad += 1
    print("level 3: %d channels SF7 x %d samples in %.1f ms (%d launches); %d packets (expected %d); consumption of the checked channels equal %s; %d of %d checked channels differ from the oracle"
          % (B, S, dt * 1e3, d.last_launches(), len(nsyms), want_pk, bool((cons == wantc).all()), bad, D + 4), flush=True)
    d.close()


if __name__ == "__main__":
    for f in (decoder, level3):
        try:
            f()
        except Exception as e:
            import traceback; traceback.print_exc()
            print("%s: EXCEPTION %r" % (f.__name__, e), flush=True)
