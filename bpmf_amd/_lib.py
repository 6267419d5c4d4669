"""ctypes binding of include/bpmf_hip.h (the C ABI of the hot path)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_i64p = C.POINTER(C.c_int64)
c_i32p = C.POINTER(C.c_int32)
c_f64p = C.POINTER(C.c_double)

# name -> (restype, argtypes); mirrors include/bpmf_hip.h one to one
_SIGNATURES = {
    "bpmf_hip_last_error": (C.c_char_p, []),
    "bpmf_hip_abi_version": (C.c_int, []),
    "bpmf_hip_supports_k": (C.c_int, [C.c_int]),
    "bpmf_hip_supports": (C.c_int, [C.c_int, C.c_int]),
    "bpmf_hip_kernel_k": (C.c_int, [C.c_int, C.c_int]),
    "bpmf_hip_ctx_num_latent": (C.c_int, [C.c_void_p]),
    "bpmf_hip_ctx_dtype": (C.c_int, [C.c_void_p]),
    "bpmf_hip_ctx_ld": (C.c_int, [C.c_void_p]),
    "bpmf_hip_ctx_create_ex": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "bpmf_hip_ctx_create": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "bpmf_hip_ctx_set_no_covariance": (C.c_int, [C.c_void_p, C.c_int]),
    "bpmf_hip_ctx_destroy": (C.c_int, [C.c_void_p]),
    "bpmf_hip_ctx_sync": (C.c_int, [C.c_void_p]),
    "bpmf_hip_ctx_stream": (C.c_void_p, [C.c_void_p]),
    "bpmf_hip_comm_unique_id": (C.c_int, [C.c_void_p]),
    "bpmf_hip_ctx_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "bpmf_hip_ctx_comm_nranks": (C.c_int, [C.c_void_p]),
    "bpmf_hip_ctx_comm_streams": (C.c_int, [C.c_void_p]),
    "bpmf_hip_side_set_ranges": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bpmf_hip_side_set_overlap": (C.c_int, [C.c_void_p, C.c_int]),
    "bpmf_hip_side_set_staleness": (C.c_int, [C.c_void_p, C.c_int]),
    "bpmf_hip_sys_set_reduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "bpmf_hip_side_set_conn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bpmf_hip_side_exchange": (C.c_int, [C.c_void_p]),
    "bpmf_hip_side_create": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_double, C.POINTER(C.c_void_p)]),
    "bpmf_hip_side_create_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_double, C.POINTER(C.c_void_p)]),
    "bpmf_hip_side_set_prop_posterior": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "bpmf_hip_side_destroy": (C.c_int, [C.c_void_p]),
    "bpmf_hip_side_items_dev": (C.c_void_p, [C.c_void_p]),
    "bpmf_hip_side_bind_items": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t]),
    "bpmf_hip_side_get_items": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bpmf_hip_side_set_items": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bpmf_hip_sample_side": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "bpmf_hip_sample_side_launch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]),
    "bpmf_hip_sample_side_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bpmf_hip_sys_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double]),
    "bpmf_hip_sys_state": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "bpmf_hip_sys_norm": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    "bpmf_hip_failed_column": (C.c_int64, [C.c_void_p]),
    "bpmf_hip_side_aggr_add": (C.c_int, [C.c_void_p]),
    "bpmf_hip_side_aggr_finalize": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "bpmf_hip_test_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "bpmf_hip_test_destroy": (C.c_int, [C.c_void_p]),
    "bpmf_hip_test_set_twin": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bpmf_hip_predict": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bpmf_hip_predict_launch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "bpmf_hip_predict_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bpmf_hip_test_get": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "bpmf_hyper_sample": (C.c_int, [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bpmf_hyper_draws": (C.c_int, [C.c_int, C.c_int64, C.c_uint32, C.c_void_p, C.c_void_p]),
    "bpmf_hyper_finish": (C.c_int, [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bpmf_cov_from_sums": (None, [C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bpmf_randn_stream": (None, [C.c_uint32, C.c_int, C.c_void_p]),
    "bpmf_hip_randn_stream": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]),
    "bpmf_hip_side_kernel_name": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "bpmf_hip_side_kernel_resources": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_int]),
    "bpmf_hip_side_schedule_info": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "bpmf_hip_side_schedule_items": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "bpmf_hip_side_kernel_ms_sum": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "bpmf_hip_side_last_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    # include/bpmf_io.h
    "bpmf_io_last_error": (C.c_char_p, []),
    "bpmf_io_read_sparse": (C.c_int, [C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                      C.POINTER(c_i64p), C.POINTER(c_i32p), C.POINTER(c_f64p)]),
    "bpmf_io_write_sparse": (C.c_int, [C.c_char_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bpmf_io_read_dense": (C.c_int, [C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(c_f64p)]),
    "bpmf_io_write_dense": (C.c_int, [C.c_char_p, C.c_int64, C.c_int64, C.c_void_p]),
    "bpmf_io_free": (None, [C.c_void_p]),
    "bpmf_assign_greedy": (C.c_int, [C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "bpmf_assign_contiguous": (C.c_int, [C.c_int64, C.c_void_p, C.c_int, C.c_double, C.c_void_p]),
}


class BpmfHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("bpmf_hip error %d: %s" % (code, msg))
        self.code = code


def library_path():
    # BPMF_HIP_LIBRARY: another build of the same library (kernel A/B comparisons in one GPU session)
    return os.environ.get("BPMF_HIP_LIBRARY") or os.path.join(_HERE, "libbpmf_hip.so")


def build_library():
    """hipcc --offload-arch=gfx950 build of the in-tree extension (bpmf_amd/csrc/Makefile)."""
    subprocess.check_call(["make", "-s", "-j%d" % max(1, min(os.cpu_count() or 1, 10)), "-C", os.path.join(_HERE, "csrc")])


def exported_signatures():
    return dict(_SIGNATURES)


def load_library():
    """Loads libbpmf_hip.so; raises (never falls back) when it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(bpmf_amd has no CPU fallback)" % path)
    lib = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.bpmf_hip_abi_version() != 1:
        raise ImportError("libbpmf_hip.so ABI version mismatch")
    _LIB = lib
    return lib


def check(rc):
    if rc != 0:
        raise BpmfHipError(rc, load_library().bpmf_hip_last_error().decode("utf-8", "replace"))
