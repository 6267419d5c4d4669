"""TEST-ONLY engine: implements the engine interface `bpmf_amd.Sys` is written against
(side_create / sample_side / predict / items / items_tensor) on numpy arrays with the CPU oracle
as the column sampler, so that the multi-rank orchestration (sharding, exchange, all-reduce) can be
exercised on CPU with the gloo backend.  It lives under tests/ on purpose: the product package has
no CPU path."""
import numpy as np

from oracle.oracle import Oracle


class _Side:
    pass


class OracleEngine:
    name = "oracle-test-engine"

    def __init__(self, K):
        self.K = K
        self.o = Oracle()

    def side_create(self, ncols, nrows, colptr, rowidx, vals, mean_rating, col_from=0, col_to=None):
        s = _Side()
        s.ncols, s.nrows, s.col_from = ncols, nrows, col_from
        s.col_to = ncols if col_to is None else col_to
        nloc = s.col_to - s.col_from
        # the oracle keys the RNG on the global column id: embed the slice in a full-width colptr
        full = np.zeros(ncols + 1, np.int64)
        full[s.col_from:s.col_to + 1] = np.asarray(colptr, np.int64)
        full[s.col_to + 1:] = colptr[nloc]
        s.csc = (full, np.ascontiguousarray(rowidx, np.int32), np.ascontiguousarray(vals, np.float64))
        s.mean = float(mean_rating)
        s.items = np.zeros((ncols, self.K))
        return s

    def items_tensor(self, side, device):
        import torch
        return torch.from_numpy(side.items)           # shares memory: collectives act on the factor in place

    def get_items(self, side):
        return side.items.copy()

    def set_items(self, side, items):
        side.items[...] = items

    def sample_side(self, side, other, it, alpha, mu, LambdaF):
        return self.o.sample_side(self.K, side.csc, side.mean, alpha, other.items, side.items, it, mu, LambdaF,
                                  from_=side.col_from, to=side.col_to)

    def test_create(self, side, tcolptr, trowidx, tvals):
        nloc = side.col_to - side.col_from
        full = np.zeros(side.ncols + 1, np.int64)
        full[side.col_from:side.col_to + 1] = np.asarray(tcolptr, np.int64)
        full[side.col_to + 1:] = tcolptr[nloc]
        tv = np.ascontiguousarray(tvals, np.float64)
        return dict(csc=(full, np.ascontiguousarray(trowidx, np.int32), tv), pavg=tv.copy(), pm2=tv.copy())

    def predict(self, test, side, other, n):
        return self.o.predict(self.K, test["csc"], side.items, other.items, side.mean, n, test["pavg"], test["pm2"],
                              from_=side.col_from, to=side.col_to)
