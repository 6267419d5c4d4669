"""One rank's share of the BASELINE "10M users x 1M items, 200 nnz per user, K=32, 8 GPUs" config,
generated ON THE DEVICE and run on one GPU: the per-GPU compute of the strong-scaling target without
the exchange (there is no 8-GPU box to launch on from here).  This is the regime the ML-1M bench
cannot show: the gathered factor matrices (V: 256 MB, U: 2.56 GB) live in HBM, not in L2/MALL.

    python tools/shard_bench.py [nranks=8] [rank=0] [users=10_000_000] [items=1_000_000] [per_user=200] [K=32]

Ratings: every user rates exactly `per_user` distinct items drawn from a Zipf(0.8)-like
popularity (stratified inverse CDF, made strictly increasing per user), values 1..5, all from
counter-based torch generators seeded by (seed, user chunk) so that any rank can regenerate any chunk.
The rank owns users [rank*U/n, (rank+1)*U/n) and the nnz-balanced item range number `rank`.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bpmf_amd

nranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
NU = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000
NI = int(sys.argv[4]) if len(sys.argv) > 4 else 1_000_000
PER = int(sys.argv[5]) if len(sys.argv) > 5 else 200
K = int(sys.argv[6]) if len(sys.argv) > 6 else 32
dev = torch.device("cuda", 0)
SEED = 42


def gen_chunk(c):
    """Users of chunk c (= the users rank c owns): items [n, PER] int64 (strictly increasing per row), values [n, PER]."""
    u0, u1 = c * NU // nranks, (c + 1) * NU // nranks
    n = u1 - u0
    g = torch.Generator(device=dev); g.manual_seed(SEED * 1000003 + c)
    u = (torch.arange(PER, device=dev, dtype=torch.float64)[None, :] + torch.rand((n, PER), generator=g, device=dev, dtype=torch.float64)) / PER
    a = NI ** 0.2 - 1.0
    it = ((1.0 + u * a) ** 5).floor().to(torch.int64) - 1          # Zipf(0.8)-like inverse CDF, non-decreasing along a row
    it.clamp_(0, NI - 1)
    j = torch.arange(PER, device=dev, dtype=torch.int64)[None, :]
    it = torch.cummax(it - j, dim=1).values + j                     # strictly increasing (duplicates bumped to the next item)
    over = it[:, -1:] - (NI - 1)
    it = (it - over.clamp(min=0)).clamp_(min=0)                     # keep the last ones in range
    val = torch.randint(1, 6, (n, PER), generator=g, device=dev).to(torch.float64)
    return u0, n, it, val


t0 = time.time()
# pass 1: item histogram over ALL users -> nnz-balanced item ranges
hist = torch.zeros(NI, dtype=torch.int64, device=dev)
for c in range(nranks):
    _, n, it, _ = gen_chunk(c)
    hist += torch.bincount(it.reshape(-1), minlength=NI)
    del it
cum = torch.cumsum(hist, 0)
total = int(cum[-1])
bounds = [0] + [int(torch.searchsorted(cum, torch.tensor(total * (r + 1) // nranks, device=dev))) + 1 for r in range(nranks - 1)] + [NI]
i0, i1 = bounds[rank], bounds[rank + 1]
print("generated histogram in %.1f s; item ranges (nnz-balanced): %s" % (time.time() - t0, bounds), flush=True)

# users side of this rank: CSC with one column per user (rows = items)
u0, nloc_u, it, val = gen_chunk(rank)
mean = 3.0
u_rowidx = it.reshape(-1).to(torch.int32).contiguous()
u_vals = val.reshape(-1).contiguous()
u_colptr = (np.arange(nloc_u + 1, dtype=np.int64) * PER)
del it, val

# items side of this rank: every rating of items [i0, i1), from all users, as CSC with one column per item
rows, cols, vals = [], [], []
for c in range(nranks):
    cu0, n, it, val = gen_chunk(c)
    m = (it >= i0) & (it < i1)
    usr = (torch.arange(n, device=dev, dtype=torch.int64)[:, None] + cu0).expand(-1, PER)
    rows.append(usr[m].to(torch.int32)); cols.append((it[m] - i0).to(torch.int32)); vals.append(val[m])
    del it, val, m, usr
rows = torch.cat(rows); cols = torch.cat(cols); vals = torch.cat(vals)
order = torch.sort(cols.to(torch.int64) * NU + rows.to(torch.int64)).indices      # by item, then ascending user
m_rowidx = rows[order].contiguous(); m_vals = vals[order].contiguous()
m_counts = torch.bincount(cols.to(torch.int64), minlength=i1 - i0)
m_colptr = np.concatenate([[0], np.cumsum(m_counts.cpu().numpy())]).astype(np.int64)
del rows, cols, vals, order
torch.cuda.synchronize()
print("rank %d of %d: users [%d, %d) with %d ratings; items [%d, %d) with %d ratings (max column %d); generated in %.1f s" % (
    rank, nranks, u0, u0 + nloc_u, len(u_rowidx), i0, i1, len(m_rowidx), int(m_counts.max()), time.time() - t0), flush=True)

eng = bpmf_amd.HipEngine(K)
t1 = time.time()
users = eng.side_create_dev(NU, NI, u_colptr, u_rowidx.data_ptr(), u_vals.data_ptr(), mean, col_from=u0, col_to=u0 + nloc_u, keep=(u_rowidx, u_vals))
movies = eng.side_create_dev(NI, NU, m_colptr, m_rowidx.data_ptr(), m_vals.data_ptr(), mean, col_from=i0, col_to=i1, keep=(m_rowidx, m_vals))
U = eng.items_tensor(users, dev); V = eng.items_tensor(movies, dev)
g = torch.Generator(device=dev); g.manual_seed(7)
eng.factors_view(U).copy_(0.3 * torch.randn((U.shape[0], K), generator=g, device=dev, dtype=torch.float64))      # (never the padding rows)
eng.factors_view(V).copy_(0.3 * torch.randn((V.shape[0], K), generator=g, device=dev, dtype=torch.float64))
torch.cuda.synchronize()
print("sides created (schedules built) in %.1f s" % (time.time() - t1), flush=True)

mu = np.zeros(K); LF = np.eye(K) * 2.0
B = lambda nnz, n: nnz * (4 + 8 + 8 * K) + n * (8 * K + 8)
for name, me, ot, nnz, ncol in (("users", users, movies, len(u_rowidx), nloc_u), ("items", movies, users, len(m_rowidx), i1 - i0)):
    ts = []
    for it_ in range(3):
        eng.sample_side(me, ot, it_, 2.0, mu, LF)
        ts.append(eng.last_kernel_ms(me)[0])
    t = min(ts)
    print("%s side: %d columns, %d ratings: sampler %.2f ms  =>  %.2f TB/s algorithmic (%.2f of 8 TB/s), %.1f M columns/s" % (
        name, ncol, nnz, t, B(nnz, ncol) / t / 1e9, B(nnz, ncol) / t / 1e9 / 8.0, ncol / t / 1e3), flush=True)
