#!/usr/bin/env python3
"""GPU fuzz: compute="bf16r" against compute="f32" (bit for bit, NaN patterns included) on randomly STRUCTURED banks -- mixtures
of i.i.d. rows, near-duplicates of a few base rows, smooth ramps, constant bands, unlabelled rows, empty objects -- over random
sizes / channel counts / id counts / storage types.  usage: tools/fuzz_bf16r.py [cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from cvpr2020_manet_amd import ops  # noqa: E402


def make_case(g, dev):
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g).item())  # noqa: E731
    C = [4, 7, 16, 26, 30, 50, 64, 99, 100, 101, 106][ri(0, 10)]
    n_ids = ri(1, 6)
    N, M = ri(1, 2500), ri(1, 40000)
    scale = [0.05, 0.1, 0.3, 1.0][ri(0, 3)]
    gd = torch.Generator(device=dev).manual_seed(ri(0, 2 ** 31 - 1))
    rn = lambda *s: torch.randn(*s, generator=gd, device=dev)  # noqa: E731
    kind = ri(0, 4)
    if kind == 0:    # i.i.d.
        k, q = torch.relu(rn(M, C)), torch.relu(rn(N, C))
    elif kind == 1:  # near-duplicates of a few base rows
        base = torch.relu(rn(ri(1, 8), C)) + 0.05
        jit = [0.0, 1e-5, 1e-3][ri(0, 2)]
        k = base[torch.randint(0, base.shape[0], (M,), generator=gd, device=dev)] + jit * rn(M, C)
        q = base[torch.randint(0, base.shape[0], (N,), generator=gd, device=dev)] + jit * rn(N, C)
    elif kind == 2:  # smooth ramps: row i = a + (i / len) b  (neighbouring rows nearly identical)
        a, b = torch.relu(rn(1, C)), rn(1, C) * 0.5
        k = torch.relu(a + torch.linspace(0, 1, M, device=dev)[:, None] * b + 1e-4 * rn(M, C))
        q = torch.relu(a + torch.linspace(0, 1, N, device=dev)[:, None] * b + 1e-4 * rn(N, C))
    elif kind == 3:  # i.i.d. with constant bands
        k, q = torch.relu(rn(M, C)), torch.relu(rn(N, C))
        c = torch.relu(rn(C))
        k[M // 3: M // 3 + ri(1, max(1, M // 2))] = c * 1.001
        q[N // 4: N // 4 + ri(1, max(1, N // 2))] = c
    else:            # mixture: half smooth, half i.i.d.
        a, b = torch.relu(rn(1, C)), rn(1, C) * 0.3
        k = torch.cat([torch.relu(a + torch.linspace(0, 1, M - M // 2, device=dev)[:, None] * b), torch.relu(rn(M // 2, C))])
        q = torch.cat([torch.relu(a + torch.linspace(0, 1, N - N // 2, device=dev)[:, None] * b), torch.relu(rn(N // 2, C))])
    k, q = (k * scale).contiguous(), (q * scale).contiguous()
    lab_kind = ri(0, 2)
    if lab_kind == 0:
        lab = torch.randint(0, n_ids, (M,), generator=gd, device=dev, dtype=torch.int32)
    elif lab_kind == 1:  # blobs
        lab = ((torch.arange(M, device=dev) * n_ids) // max(M, 1)).to(torch.int32).clamp_(max=n_ids - 1)
    else:                # interleaved
        lab = (torch.arange(M, device=dev) % n_ids).to(torch.int32)
    if ri(0, 3) == 0:
        lab[torch.rand(M, generator=gd, device=dev) < 0.5] = -1
    if n_ids > 1 and ri(0, 4) == 0:
        lab[lab == n_ids - 1] = -1  # an empty object
    storage = torch.bfloat16 if ri(0, 3) == 0 else torch.float32
    k, q = k.to(storage), q.to(storage)
    if ri(0, 9) == 0:  # NaN rows
        k[ri(0, M - 1), ri(0, C - 1)] = float("nan")
    if ri(0, 9) == 0:
        q[ri(0, N - 1), ri(0, C - 1)] = float("nan")
    return q, k, lab, n_ids, dict(C=C, N=N, M=M, n_ids=n_ids, kind=kind, lab=lab_kind, scale=scale, storage=str(storage))


def same(a, b):
    na, nb = torch.isnan(a), torch.isnan(b)
    return bool(torch.equal(na, nb)) and bool(torch.equal(torch.where(na, torch.zeros_like(a), a), torch.where(nb, torch.zeros_like(b), b)))


def run(cases, seed, verbose=True):
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(seed)
    bad = 0
    for i in range(cases):
        q, k, lab, n_ids, info = make_case(g, dev)
        want = ops.global_match(k, q, lab, n_ids, compute="f32")
        bank = ops.PreparedBank(k, lab, n_ids, compute="bf16r")
        got = bank.match(q, adaptive=False)
        ok = same(got, want)
        st = bank.refine_stats_full()
        if verbose or not ok:
            print("%s case %d %s rows/pair %.1f rescued %d/%d" % ("ok " if ok else "BAD", i, info, st["candidate_rows_per_pair"],
                                                                 st["rescued_tiles"], st["query_tiles"]))
        bad += 0 if ok else 1
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    b = run(n, s, verbose=os.environ.get("VERBOSE") == "1")
    print("%d cases, %d mismatches" % (n, b))
    sys.exit(1 if b else 0)
