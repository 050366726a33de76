"""ctypes binding of ``libhipeig.so`` (the C ABI declared in ``include/hipeig.h``).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C
eigensolvers_amd/csrc``.  There is no fallback: if the shared object is missing or a
call fails, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HIPEIG_LIB", os.path.join(_HERE, "libhipeig.so"))   # override: A/B builds


class HipEigError(RuntimeError):
    pass


_P = C.c_void_p
_I64 = C.c_int64
_D = C.c_double
_DP = C.POINTER(C.c_double)
_I64P = C.POINTER(C.c_int64)
_I32P = C.POINTER(C.c_int32)
_IP = C.POINTER(C.c_int)
_PP = C.POINTER(C.c_void_p)

# name -> argument types (every function returns int status except hipeig_last_error)
SIGNATURES = {
    "hipeig_ctx_create": [C.c_int, _PP],
    "hipeig_ctx_destroy": [_P],
    "hipeig_ctx_sync": [_P],
    "hipeig_device_info": [_P, _I64P, C.c_char_p, C.c_int],
    "hipeig_comm_unique_id": [_P],
    "hipeig_comm_library": [C.c_char_p, C.c_int],
    "hipeig_comm_init": [_P, C.c_int, C.c_int, _P],
    "hipeig_comm_destroy": [_P],
    "hipeig_comm_info": [_P, _IP, _IP],
    "hipeig_comm_stats": [_P, _I64P],
    "hipeig_comm_set_partitioned": [_P, C.c_int],
    "hipeig_vec_allreduce": [_P, _P, C.c_int64],
    "hipeig_comm_init_direct": [_P, C.c_int, C.c_int],
    "hipeig_comm_set_allreduce_backend": [_P, C.c_int],
    "hipeig_direct_alloc": [_P, _I64, _P],
    "hipeig_direct_attach": [_P, _P],
    "hipeig_direct_release": [_P],
    "hipeig_comm_set_wait_limit": [_P, C.c_double],
    "hipeig_comm_set_exchange": [_P, C.c_int],
    "hipeig_comm_set_gather_backend": [_P, C.c_int],
    "hipeig_comm_gather_info": [_P, _I64P],
    "hipeig_comm_set_gather_chunks": [_P, C.c_int],
    "hipeig_phase_timing": [_P, C.c_int],
    "hipeig_phase_get": [_P, _DP],
    "hipeig_comm_bench_allreduce": [_P, C.c_int, C.c_int, _DP],
    "hipeig_loopback_group_create": [C.c_int, C.POINTER(C.c_void_p)],
    "hipeig_loopback_group_destroy": [_P],
    "hipeig_comm_init_loopback": [_P, _P, C.c_int],
    "hipeig_vec_alloc": [_P, _I64, _PP],
    "hipeig_vec_free": [_P, _P],
    "hipeig_vec_upload": [_P, _P, _P, _I64],
    "hipeig_vec_download": [_P, _P, _P, _I64],
    "hipeig_vec_copy": [_P, _P, _P, _I64],
    "hipeig_vec_fill": [_P, _P, _I64, _D],
    "hipeig_dot": [_P, _I64, _P, _P, _DP],
    "hipeig_nrm2": [_P, _I64, _P, _DP],
    "hipeig_normalize": [_P, _I64, _P, _DP],
    "hipeig_scale": [_P, _I64, _D, _P, _P],
    "hipeig_divide": [_P, _I64, _D, _P, _P],
    "hipeig_axpby": [_P, _I64, _D, _P, _D, _P],
    "hipeig_lincomb": [_P, _I64, C.c_int, _DP, _PP, _P],
    "hipeig_lincomb_block": [_P, _I64, C.c_int, C.c_int, _DP, C.c_int, _PP, _PP],
    "hipeig_multi_dot": [_P, _I64, C.c_int, _PP, _P, _DP],
    "hipeig_multi_axpy": [_P, _I64, C.c_int, _PP, _DP, _P],
    "hipeig_gram": [_P, _I64, C.c_int, _PP, C.c_int, _PP, _DP],
    "hipeig_orthonormalize": [_P, _I64, C.c_int, _PP, _P, _D, C.c_int, _DP, _IP],
    "hipeig_mgs_project": [_P, _I64, C.c_int, _PP, _P, _DP],
    "hipeig_pair_mgs_project": [_P, _I64, C.c_int, _PP, _PP, _P, _P, _DP],
    "hipeig_arnoldi_step": [_P, _I64, C.c_int, _PP, _P, _DP],
    "hipeig_pair_arnoldi_step": [_P, _I64, C.c_int, _PP, _PP, _P, _P, _DP],
    "hipeig_arnoldi_step_p": [_P, _I64, C.c_int, _PP, _P, _DP, C.c_int],
    "hipeig_pair_arnoldi_step_begin": [_P, _I64, C.c_int, _PP, _PP, _P, _P, C.c_int, C.c_int],
    "hipeig_pair_arnoldi_step_batch_begin": [_P, _I64, C.c_int, _IP, _PP, _PP, _PP, _PP],
    "hipeig_arnoldi_step_end": [_P, C.c_int, C.c_int, _DP],
    "hipeig_pair_arnoldi_step_p": [_P, _I64, C.c_int, _PP, _PP, _P, _P, _DP, C.c_int],
    "hipeig_csr_create": [_P, _I64, _I64, _I64, _I64P, _I32P, _DP, _PP],
    "hipeig_csr_generate": [_P, _I64, _I64, _I64, C.c_int, C.c_uint64, _D, C.c_uint32, _DP, C.c_int, _PP],
    "hipeig_csr_destroy": [_P, _P],
    "hipeig_csr_info": [_P, _I64P],
    "hipeig_csr_layout_info": [_P, _I64P],
    "hipeig_csr_download": [_P, _P, _I64P, _I32P, _DP],
    "hipeig_csr_set_variant": [_P, C.c_int],
    "hipeig_csr_set_reproducible": [_P, C.c_int],
    "hipeig_csr_fixed_info": [_P, _P, _DP],
    "hipeig_spmv": [_P, _P, _P, _P],
    "hipeig_spmv_shift": [_P, _P, _D, _D, _P, _P],
    "hipeig_spmv_shift_pair": [_P, _P, _D, _D, _D, _P, _P, _P, _P],
    "hipeig_csr_pair_info": [_P, _I64P],
    "hipeig_spmm": [_P, _P, C.c_int, _PP, _PP],
    "hipeig_spmm_shift_pairs": [_P, _P, C.c_int, _D, _D, _D, _PP, _PP, _PP, _PP],
    "hipeig_minres": [_P, _P, _D, _D, _P, _P, _D, C.c_int, _IP, _DP],
    "hipeig_minres_x0": [_P, _P, _D, _D, _P, _P, _P, _D, C.c_int, _IP, _DP],
    "hipeig_dense_solve_small": [_P, _P, _D, _D, _D, _P, _P, _P, _P, _IP],
    "hipeig_minres_block": [_P, _P, _D, _D, C.c_int, _PP, _PP, _D, C.c_int, _IP, _DP],
    "hipeig_csr_set_block_variant": [_P, C.c_int],
    "hipeig_csr_block_info": [_P, _I64P],
    "hipeig_timer_start": [_P],
    "hipeig_timer_stop": [_P, C.POINTER(C.c_float)],
}

_lib = None


def load():
    """Load the shared library (once) and attach prototypes.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipEigError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C eigensolvers_amd/csrc` (there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError if the ABI and the header drifted
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.hipeig_last_error.argtypes = []
    lib.hipeig_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def check(status, what=""):
    if status != 0:
        msg = load().hipeig_last_error().decode("utf-8", "replace")
        raise HipEigError(f"{what or 'libhipeig call'} failed (status {status}): {msg}")


def call(name, *args):
    check(getattr(load(), name)(*args), name)
