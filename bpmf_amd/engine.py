"""HipEngine: numpy-facing wrapper of the C ABI (device = one MI355X).

The engine interface (side_create / sample_side / predict / items get/set /
hyper_sample) is what `Sys` is written against.
"""
import ctypes as C

import numpy as np

from . import _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class _Side:
    def __init__(self, handle, K, ncols, nrows, col_from, col_to, keep):
        self.handle, self.K, self.ncols, self.nrows = handle, K, ncols, nrows
        self.col_from, self.col_to = col_from, col_to
        self._keep = keep          # arrays / tensors that must outlive the handle


class HipEngine:
    """One context = one GPU + one stream.  `stream` is an integer hipStream_t or None."""

    name = "hip"

    def __init__(self, K, device=0, stream=None, dtype="f64"):
        """dtype "f64": the reference's arithmetic, any num_latent 1 .. 128 (8, 16, 32, 64, 128 have kernels of their own, any other
        K runs on the next of those sizes with zero rows in the extra dimensions: ld()); "f32": the opt-in large-K mixed
        precision path, num_latent 65 .. 128 (fp32 factors / Gram / factorisation, fp64 everything else)."""
        self.lib = _lib.load_library()
        self.K = int(K)
        self.dtype = dtype
        code = {"f64": 0, "f32": 1}[dtype]
        h = C.c_void_p()
        _lib.check(self.lib.bpmf_hip_ctx_create_ex(int(device), self.K, code, C.c_void_p(stream) if stream else None, C.byref(h)))
        self.ctx = h
        self.device = device
        self._sides = []

    # -- lifetime -----------------------------------------------------------
    def close(self):
        if getattr(self, "ctx", None):
            for t in getattr(self, "_tests", []):
                if t[0]:
                    self.lib.bpmf_hip_test_destroy(t[0])
                    t[0] = None
            for s in self._sides:
                if s.handle:
                    self.lib.bpmf_hip_side_destroy(s.handle)
                    s.handle = None
            self.lib.bpmf_hip_ctx_destroy(self.ctx)
            self.ctx = None

    def set_no_covariance(self, on):
        """The reference's BPMF_NO_COVARIANCE build as a switch: diagonal Lambda* only."""
        _lib.check(self.lib.bpmf_hip_ctx_set_no_covariance(self.ctx, 1 if on else 0))

    def sync(self):
        _lib.check(self.lib.bpmf_hip_ctx_sync(self.ctx))

    # -- multi-GPU (RCCL inside the library) -------------------------------------
    def comm_unique_id(self):
        buf = (C.c_char * 128)()
        _lib.check(self.lib.bpmf_hip_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, nranks, rank, unique_id):
        buf = (C.c_char * 128).from_buffer_copy(unique_id)
        _lib.check(self.lib.bpmf_hip_ctx_comm_init(self.ctx, int(nranks), int(rank), buf))
        self.nranks, self.rank = int(nranks), int(rank)

    def comm_streams(self):
        """0 = no communicator, 1 = one, 2 = a second one split off for the statistics / evaluation streams (ncclCommSplit)"""
        return int(self.lib.bpmf_hip_ctx_comm_streams(self.ctx))

    def comm_nranks(self):
        """Ranks of the communicator as the communication library counts them (1 without one)."""
        return int(self.lib.bpmf_hip_ctx_comm_nranks(self.ctx))

    def side_set_ranges(self, side, bounds):
        b = np.ascontiguousarray(bounds, np.int64)
        _lib.check(self.lib.bpmf_hip_side_set_ranges(side.handle, _ptr(b)))

    def side_set_overlap(self, side, nparts):
        """Cut every rank's range into `nparts` parts: part c is exchanged while part c + 1 is sampled (collective)."""
        _lib.check(self.lib.bpmf_hip_side_set_overlap(side.handle, int(nparts)))

    def side_set_staleness(self, side, k):
        """Bounded-staleness exchange: a part of the side travels every (k + 1)-th half-iteration only (include/bpmf_hip.h)."""
        _lib.check(self.lib.bpmf_hip_side_set_staleness(side.handle, int(k)))

    def sys_set_reduce(self, a, b, on=True):
        """BPMF_REDUCE formulation for the pair of sides (preComputeMuLambda + reduce onto the owners; include/bpmf_hip.h)."""
        _lib.check(self.lib.bpmf_hip_sys_set_reduce(a.handle, b.handle, 1 if on else 0))

    def side_set_conn(self, side, send_ptr=None, send_cols=None, recv_ptr=None, recv_cols=None):
        """Connectivity-aware exchange lists (include/bpmf_hip.h); all None: back to the all-gather form."""
        if send_ptr is None:
            _lib.check(self.lib.bpmf_hip_side_set_conn(side.handle, None, None, None, None))
            return
        sp = np.ascontiguousarray(send_ptr, np.int64); sc = np.ascontiguousarray(send_cols, np.int32)
        rp = np.ascontiguousarray(recv_ptr, np.int64); rc = np.ascontiguousarray(recv_cols, np.int32)
        _lib.check(self.lib.bpmf_hip_side_set_conn(side.handle, _ptr(sp), _ptr(sc) if len(sc) else None,
                                                   _ptr(rp), _ptr(rc) if len(rc) else None))

    def side_exchange(self, side):
        _lib.check(self.lib.bpmf_hip_side_exchange(side.handle))

    # -- sides ----------------------------------------------------------------
    def side_create(self, ncols, nrows, colptr, rowidx, vals, mean_rating, col_from=0, col_to=None):
        col_to = ncols if col_to is None else col_to
        colptr = np.ascontiguousarray(colptr, np.int64)
        rowidx = np.ascontiguousarray(rowidx, np.int32)
        vals = np.ascontiguousarray(vals, np.float64)
        assert len(colptr) == col_to - col_from + 1
        h = C.c_void_p()
        _lib.check(self.lib.bpmf_hip_side_create(self.ctx, ncols, nrows, col_from, col_to, _ptr(colptr), _ptr(rowidx),
                                                 _ptr(vals), float(mean_rating), C.byref(h)))
        s = _Side(h, self.K, ncols, nrows, col_from, col_to, None)
        self._sides.append(s)
        return s

    def side_create_dev(self, ncols, nrows, colptr_host, rowidx_dev_ptr, vals_dev_ptr, mean_rating, col_from=0,
                        col_to=None, keep=None):
        col_to = ncols if col_to is None else col_to
        colptr_host = np.ascontiguousarray(colptr_host, np.int64)
        h = C.c_void_p()
        _lib.check(self.lib.bpmf_hip_side_create_dev(self.ctx, ncols, nrows, col_from, col_to, _ptr(colptr_host),
                                                     C.c_void_p(rowidx_dev_ptr), C.c_void_p(vals_dev_ptr),
                                                     float(mean_rating), C.byref(h)))
        s = _Side(h, self.K, ncols, nrows, col_from, col_to, keep)
        self._sides.append(s)
        return s

    def side_destroy(self, side):
        for t in getattr(self, "_tests", []):                      # test matrices go before the side they sit on
            if t[0] and t[2] is side:
                self.test_destroy(t)
        if side.handle:
            _lib.check(self.lib.bpmf_hip_side_destroy(side.handle))
            side.handle = None

    def items_dev_ptr(self, side):
        return self.lib.bpmf_hip_side_items_dev(side.handle)

    def ld(self):
        """Rows per column of the context's DEVICE arrays (bpmf_hip_ctx_ld): K for 8 / 16 / 32 / 64 / 128, else the next of those."""
        return int(self.lib.bpmf_hip_ctx_ld(self.ctx))

    def bind_items(self, side, dev_ptr, keep=None, ld=None, nbytes=None):
        """`ld` / `nbytes`: what the caller allocated (default: the caller's K x ncols doubles); the library refuses storage
        whose leading dimension is not the context's or that is too small (a padded num_latent needs ld() rows per column)."""
        ld = self.K if ld is None else int(ld)
        nbytes = 8 * ld * side.ncols if nbytes is None else int(nbytes)
        _lib.check(self.lib.bpmf_hip_side_bind_items(side.handle, C.c_void_p(dev_ptr), ld, nbytes))
        side._items_keep = keep

    def items_tensor(self, side, device):
        """Allocates the factor matrix as a torch tensor [ncols, ld()] on `device` (rows num_latent .. ld()-1 of every column
        stay zero), binds the side to it (bpmf_hip_side_bind_items) and returns the WHOLE tensor, so that RCCL collectives move
        full device rows in place; t[:, :K] are the factors."""
        import torch
        ld = self.ld()
        t = torch.zeros((side.ncols, ld), dtype=torch.float64, device=device)
        self.bind_items(side, t.data_ptr(), keep=t, ld=ld, nbytes=t.numel() * 8)
        return t

    def factors_view(self, t):
        """The num_latent factor rows of a tensor items_tensor() returned: t[:, :K].  WRITE through this view, never through the
        whole tensor: with a padded num_latent (20 on the K = 32 kernels) the rows K .. ld()-1 of every column must stay zero --
        the kernels read them as factor entries (`U.copy_(randn(U.shape))` filled them)."""
        return t[:, :self.K]

    def set_prop_posterior(self, side, Lambda):
        """Per-column prior precisions of the side's local columns: [ncols_local, K*K] (each row a
        column-major K x K matrix, as in *-Lambda.ddm), or None to remove them."""
        if Lambda is None:
            _lib.check(self.lib.bpmf_hip_side_set_prop_posterior(side.handle, None, None))
            return
        L = np.ascontiguousarray(Lambda, np.float64).reshape(side.col_to - side.col_from, self.K * self.K)
        _lib.check(self.lib.bpmf_hip_side_set_prop_posterior(side.handle, None, _ptr(L)))

    def get_items(self, side):
        """[ncols, K] C-order array = the K x ncols column-major factor matrix."""
        out = np.empty((side.ncols, self.K), np.float64)
        _lib.check(self.lib.bpmf_hip_side_get_items(side.handle, _ptr(out)))
        return out

    def set_items(self, side, items):
        items = np.ascontiguousarray(items, np.float64)
        assert items.shape == (side.ncols, self.K)
        _lib.check(self.lib.bpmf_hip_side_set_items(side.handle, _ptr(items)))

    # -- hot path ---------------------------------------------------------------
    def sample_side_launch(self, side, other, it, alpha, mu, LambdaF):
        mu = np.ascontiguousarray(mu, np.float64)
        LF = np.asfortranarray(LambdaF, np.float64)
        _lib.check(self.lib.bpmf_hip_sample_side_launch(side.handle, other.handle, int(it), float(alpha), _ptr(mu), _ptr(LF)))

    def sample_side_finish(self, side):
        K = self.K
        s = np.empty(K); prod = np.empty((K, K), order="F"); nrm = np.empty(1)
        _lib.check(self.lib.bpmf_hip_sample_side_finish(side.handle, _ptr(s), _ptr(prod), _ptr(nrm)))
        return s, prod, float(nrm[0])

    def sample_side(self, side, other, it, alpha, mu, LambdaF):
        """Returns (sum[K], prod[K,K], norm) over the columns [col_from,col_to) of `side`."""
        self.sample_side_launch(side, other, it, alpha, mu, LambdaF)
        return self.sample_side_finish(side)

    def sys_sample(self, side, other, alpha):
        """Sys::sample(Sys&) with the state (iter, cov, hp) kept inside the library."""
        _lib.check(self.lib.bpmf_hip_sys_sample(side.handle, other.handle, float(alpha)))

    def sys_state(self, side):
        """(iter, norm, cov, mu, LambdaF, LambdaU) of a side driven by sys_sample."""
        K = self.K
        it = C.c_int(); nrm = C.c_double()
        cov = np.empty((K, K), order="F"); mu = np.empty(K); LF = np.empty((K, K), order="F"); LU = np.empty((K, K), order="F")
        _lib.check(self.lib.bpmf_hip_sys_state(side.handle, C.byref(it), C.byref(nrm), _ptr(cov), _ptr(mu), _ptr(LF), _ptr(LU)))
        return it.value, nrm.value, cov, mu, LF, LU

    def sys_norm(self, side, it):
        """Sys::norm of half-iteration `it` of the side (one of its last 8), without draining later half-iterations."""
        nrm = C.c_double()
        _lib.check(self.lib.bpmf_hip_sys_norm(side.handle, int(it), C.byref(nrm)))
        return nrm.value

    def aggr_add(self, side):
        """aggrMu / aggrLambda += r, r r^T of the side's local columns (device)."""
        _lib.check(self.lib.bpmf_hip_side_aggr_add(side.handle))

    def aggr_finalize(self, side, nsamples):
        """Sys::finalize_mu_lambda: (mu [nloc, K], Lambda [nloc, K*K]) of the local columns."""
        nloc = side.col_to - side.col_from
        mu = np.empty((nloc, self.K)); lam = np.empty((nloc, self.K * self.K))
        _lib.check(self.lib.bpmf_hip_side_aggr_finalize(side.handle, int(nsamples), _ptr(mu), _ptr(lam)))
        return mu, lam

    def kernel_name(self, side):
        """The kernel(s) one sampler launch of the side consists of, as a profile names them."""
        buf = C.create_string_buffer(512)
        _lib.check(self.lib.bpmf_hip_side_kernel_name(side.handle, buf, 512))
        return buf.value.decode()

    def kernel_resources(self, side):
        """[{kernel, lds_bytes_per_workgroup, threads_per_workgroup, workgroups_per_cu, vgprs}] of the kernels one sampler
        launch of the side consists of (asked of the library's dispatch logic; nothing is launched)."""
        out = np.zeros(32, np.int64)
        names = C.create_string_buffer(1024)
        n = self.lib.bpmf_hip_side_kernel_resources(side.handle, _ptr(out), 8, names, 1024)
        if n < 0:
            _lib.check(n)
        spelt = names.value.decode().split(";") if n else []
        listed = [x.strip() for x in self.kernel_name(side).split(" + ")]
        res = []
        for i in range(n):
            nm = spelt[i] if i < len(spelt) else "?"
            if nm in ("kernel", "?") and len(listed) == n:            # (launched through a generic lambda: the name list says which)
                nm = listed[i]
            rec = {"kernel": nm, "lds_bytes_per_workgroup": int(out[4 * i]), "threads_per_workgroup": int(out[4 * i + 1]),
                   "workgroups_per_cu": int(out[4 * i + 2]), "vgprs": int(out[4 * i + 3])}
            # the probe asks the unfused dispatch; a fused launch (k_sample1s<64> = the same device code + the statistics rider
            # of kernels_slab.h) is reported under the name the side launches, with the probed twin named beside it
            if n == 1 and len(listed) == 1 and listed[0].split("<")[0] != nm.split("<")[0]:
                rec["kernel"], rec["probed_as"] = listed[0], nm
            res.append(rec)
        return res

    def schedule_info(self, side):
        out = np.zeros(16, np.int64)
        _lib.check(self.lib.bpmf_hip_side_schedule_info(side.handle, _ptr(out), 16))
        keys = ("mode", "work_items", "chunks", "chunked_columns", "light_columns", "other_items", "pf_le3", "pf_4to6", "pf_7to16",
                "lr_columns", "parts", "local_columns", "local_ratings", "pf_ratings", "pf_ratings_sq")
        return {k: int(v) for k, v in zip(keys, out)}

    def schedule_items(self, side):
        """The side's work items in launch order: (local column, ratings, heavy-column ordinal or -1) arrays."""
        n = C.c_int64()
        _lib.check(self.lib.bpmf_hip_side_schedule_items(side.handle, None, None, None, 0, C.byref(n)))
        col = np.zeros(n.value, np.int32); ln = np.zeros(n.value, np.int32); heavy = np.zeros(n.value, np.int32)
        if n.value:
            _lib.check(self.lib.bpmf_hip_side_schedule_items(side.handle, _ptr(col), _ptr(ln), _ptr(heavy), n.value, None))
        return col, ln, heavy

    def kernel_ms_sum(self, side):
        """(sampler ms, statistics ms, launches) summed over the half-iterations run through sys_sample."""
        a = C.c_double(); b = C.c_double(); n = C.c_int64()
        _lib.check(self.lib.bpmf_hip_side_kernel_ms_sum(side.handle, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def last_kernel_ms(self, side):
        a = C.c_float(); b = C.c_float()
        _lib.check(self.lib.bpmf_hip_side_last_kernel_ms(side.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    # -- prediction -----------------------------------------------------------
    def test_create(self, side, tcolptr, trowidx, tvals):
        tcolptr = np.ascontiguousarray(tcolptr, np.int64)
        trowidx = np.ascontiguousarray(trowidx, np.int32)
        tvals = np.ascontiguousarray(tvals, np.float64)
        h = C.c_void_p()
        _lib.check(self.lib.bpmf_hip_test_create(side.handle, _ptr(tcolptr), _ptr(trowidx), _ptr(tvals), C.byref(h)))
        t = [h, int(tcolptr[-1]), side]
        if not hasattr(self, "_tests"):
            self._tests = []
        self._tests.append(t)
        return t

    def test_set_twin(self, test, twin):
        """users.predict(movies): `twin` (a test matrix of the other side) is evaluated whenever `test` is."""
        _lib.check(self.lib.bpmf_hip_test_set_twin(test[0], twin[0] if twin is not None else None))

    def test_destroy(self, test):
        if test[0]:
            _lib.check(self.lib.bpmf_hip_test_destroy(test[0]))
            test[0] = None

    def predict(self, test, side, other, n):
        se = C.c_double(); sea = C.c_double(); cnt = C.c_int64()
        _lib.check(self.lib.bpmf_hip_predict(test[0], side.handle, other.handle, int(n), C.byref(se), C.byref(sea), C.byref(cnt)))
        return se.value, sea.value, cnt.value

    def predict_launch(self, test, side, other, n):
        _lib.check(self.lib.bpmf_hip_predict_launch(test[0], side.handle, other.handle, int(n)))

    def predict_finish(self, test):
        se = C.c_double(); sea = C.c_double(); cnt = C.c_int64()
        _lib.check(self.lib.bpmf_hip_predict_finish(test[0], C.byref(se), C.byref(sea), C.byref(cnt)))
        return se.value, sea.value, cnt.value

    def test_get(self, test):
        pavg = np.empty(test[1]); pm2 = np.empty(test[1])
        _lib.check(self.lib.bpmf_hip_test_get(test[0], _ptr(pavg), _ptr(pm2)))
        return pavg, pm2

    # -- host-side hyper parameters (also in the library, not device code) -------
    def hyper_sample(self, N, cov, counter, Um=None):
        return hyper_sample(self.K, N, cov, counter, Um)

    def randn_device(self, counter, n):
        out = np.empty(n)
        _lib.check(self.lib.bpmf_hip_randn_stream(self.ctx, int(counter) & 0xFFFFFFFF, int(n), _ptr(out)))
        return out


def hyper_sample(K, N, cov, counter, Um=None):
    """HyperParams::sample on the host (c++/bpmf.h:98-103): returns mu[K], LambdaU[K,K], LambdaF[K,K]."""
    lib = _lib.load_library()
    cov = np.asfortranarray(cov, np.float64)
    mu = np.empty(K); LU = np.empty((K, K), order="F"); LF = np.empty((K, K), order="F")
    um = np.ascontiguousarray(Um, np.float64) if Um is not None else None
    _lib.check(lib.bpmf_hyper_sample(int(K), int(N), _ptr(cov), _ptr(um), int(counter) & 0xFFFFFFFF, _ptr(mu), _ptr(LU), _ptr(LF)))
    return mu, LU, LF


def cov_from_sums(K, N, s, prod):
    lib = _lib.load_library()
    s = np.ascontiguousarray(s, np.float64); prod = np.asfortranarray(prod, np.float64)
    cov = np.empty((K, K), order="F")
    lib.bpmf_cov_from_sums(int(K), int(N), _ptr(s), _ptr(prod), _ptr(cov))
    return cov


def randn_host(counter, n):
    lib = _lib.load_library()
    out = np.empty(n)
    lib.bpmf_randn_stream(int(counter) & 0xFFFFFFFF, int(n), _ptr(out))
    return out
