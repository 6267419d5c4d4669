"""Device-generated rating matrix of BASELINE.json configs[3]: 10M users x 1M items, exactly 200
ratings per user, K = 32 -- the strong-scaling workload of the north star.

2e9 ratings are 24 GB per orientation; they are generated ON THE GPU from counter-based torch
generators seeded by (seed, chunk), so that every rank can regenerate any part of the SAME matrix
whatever the number of ranks: the users are cut into G = 8 fixed chunks and the items into G
nnz-balanced ranges; with N ranks (N divides G) rank r owns chunks / ranges [r G/N, (r+1) G/N).
N = 1 therefore holds the very matrix that N = 8 shards.

Every user rates exactly `per_user` distinct items drawn from a Zipf(0.8)-like popularity
(stratified inverse CDF, made strictly increasing per user), values 1..5.  The test set is 0.1 %
extra (user, item, value) triples from the same distributions.
"""
import numpy as np
import torch


class BigMatrix:
    def __init__(self, device, nusers=10_000_000, nitems=1_000_000, per_user=200, groups=8, seed=42):
        self.dev = torch.device(device)
        self.NU, self.NI, self.PER, self.G, self.seed = int(nusers), int(nitems), int(per_user), int(groups), int(seed)
        self._bounds = None
        self.mean_rating = 3.0                      # E[uniform 1..5]; the exact sample mean differs in the 5th digit

    # -- users of chunk c: items [n, PER] int64 (strictly increasing along a row), values [n, PER] f64
    def chunk_range(self, c):
        return c * self.NU // self.G, (c + 1) * self.NU // self.G

    def _zipf(self, u):
        a = self.NI ** 0.2 - 1.0
        return ((1.0 + u * a) ** 5).floor().to(torch.int64).clamp_(1, self.NI) - 1

    def gen_chunk(self, c):
        u0, u1 = self.chunk_range(c)
        n, PER, NI, dev = u1 - u0, self.PER, self.NI, self.dev
        g = torch.Generator(device=dev); g.manual_seed(self.seed * 1000003 + c)
        u = (torch.arange(PER, device=dev, dtype=torch.float64)[None, :] + torch.rand((n, PER), generator=g, device=dev, dtype=torch.float64)) / PER
        it = self._zipf(u)                                              # non-decreasing along a row
        del u
        j = torch.arange(PER, device=dev, dtype=torch.int64)[None, :]
        it = torch.cummax(it - j, dim=1).values + j                     # strictly increasing (duplicates bumped to the next item)
        over = it[:, -1:] - (NI - 1)
        it = (it - over.clamp(min=0)).clamp_(min=0)                     # keep the last ones in range
        val = torch.randint(1, 6, (n, PER), generator=g, device=dev).to(torch.float64)
        return u0, n, it, val

    def item_bounds(self):
        """G + 1 bounds of the nnz-balanced item ranges (histogram over ALL users)."""
        if self._bounds is None:
            hist = torch.zeros(self.NI, dtype=torch.int64, device=self.dev)
            for c in range(self.G):
                _, _, it, _ = self.gen_chunk(c)
                hist += torch.bincount(it.reshape(-1), minlength=self.NI)
                del it
            cum = torch.cumsum(hist, 0)
            total = int(cum[-1])
            b = [0]
            for r in range(self.G - 1):
                b.append(max(b[-1], int(torch.searchsorted(cum, torch.tensor(total * (r + 1) // self.G, device=self.dev))) + 1))
            b.append(self.NI)
            self._bounds = b
            self.item_counts = hist
        return self._bounds

    def users_csc(self, parts):
        """CSC with one column per user (rows = items) of the user chunks `parts` (consecutive):
        (colptr host int64, rowidx device int32, vals device f64, u0, u1)."""
        rows, vals = [], []
        for c in parts:
            _, _, it, val = self.gen_chunk(c)
            rows.append(it.reshape(-1).to(torch.int32)); vals.append(val.reshape(-1))
            del it, val
        u0, u1 = self.chunk_range(parts[0])[0], self.chunk_range(parts[-1])[1]
        rowidx = torch.cat(rows) if len(rows) > 1 else rows[0].contiguous()
        v = torch.cat(vals) if len(vals) > 1 else vals[0].contiguous()
        colptr = np.arange(u1 - u0 + 1, dtype=np.int64) * self.PER
        return colptr, rowidx, v, u0, u1

    def items_csc(self, parts):
        """CSC with one column per item (rows = users, ascending) of the item ranges `parts`
        (consecutive), ratings of ALL users: (colptr host, rowidx device, vals device, i0, i1)."""
        b = self.item_bounds()
        i0, i1 = b[parts[0]], b[parts[-1] + 1]
        out_rows, out_vals, counts = [], [], []
        for gpart in parts:
            lo, hi = b[gpart], b[gpart + 1]
            rows, cols, vals = [], [], []
            for c in range(self.G):
                cu0, n, it, val = self.gen_chunk(c)
                m = (it >= lo) & (it < hi)
                usr = (torch.arange(n, device=self.dev, dtype=torch.int64)[:, None] + cu0).expand(-1, self.PER)
                rows.append(usr[m].to(torch.int32)); cols.append((it[m] - lo).to(torch.int32)); vals.append(val[m])
                del it, val, m, usr
            rows = torch.cat(rows); cols = torch.cat(cols); vals = torch.cat(vals)
            order = torch.sort(cols.to(torch.int64) * self.NU + rows.to(torch.int64)).indices        # by item, then ascending user
            out_rows.append(rows[order]); out_vals.append(vals[order])
            counts.append(torch.bincount(cols.to(torch.int64), minlength=hi - lo))
            del rows, cols, vals, order
        rowidx = torch.cat(out_rows) if len(out_rows) > 1 else out_rows[0].contiguous()
        v = torch.cat(out_vals) if len(out_vals) > 1 else out_vals[0].contiguous()
        cnt = torch.cat(counts).cpu().numpy()
        colptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        return colptr, rowidx, v, i0, i1

    def test_csc(self, i0, i1, frac=0.001):
        """Test entries whose item lies in [i0, i1): host CSC by item (colptr rebased to the range)."""
        nt = max(1, int(self.NU * self.PER * frac))
        g = torch.Generator(device=self.dev); g.manual_seed(self.seed * 1000003 + 999)
        usr = torch.randint(0, self.NU, (nt,), generator=g, device=self.dev)
        it = self._zipf(torch.rand((nt,), generator=g, device=self.dev, dtype=torch.float64))
        val = torch.randint(1, 6, (nt,), generator=g, device=self.dev).to(torch.float64)
        m = (it >= i0) & (it < i1)
        usr, it, val = usr[m], it[m] - i0, val[m]
        # (duplicate cells are dropped: keep the value of one representative)
        order = torch.sort(it * self.NU + usr, stable=True).indices
        k = (it * self.NU + usr)[order]; v = val[order]
        keep = torch.ones_like(k, dtype=torch.bool); keep[1:] = k[1:] != k[:-1]
        k, v = k[keep], v[keep]
        cols = (k // self.NU).cpu().numpy(); rows = (k % self.NU).cpu().numpy().astype(np.int32)
        colptr = np.concatenate([[0], np.cumsum(np.bincount(cols, minlength=i1 - i0))]).astype(np.int64)
        return colptr, rows, v.cpu().numpy()
