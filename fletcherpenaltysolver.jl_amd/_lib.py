"""ctypes binding of libfpsq.so (the C ABI of include/fpsq.h).

There is NO fallback: if the HIP library is missing or no MI355X is visible the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FPSQ_LIB_PATH: developer A/B runs against another BUILD of the same HIP library (never a CPU implementation)
LIB_PATH = os.environ.get("FPSQ_LIB_PATH") or os.path.join(_HERE, "lib", "libfpsq.so")


class Stats(C.Structure):
    _fields_ = [("solved", C.c_int32), ("inconsistent", C.c_int32), ("niter", C.c_int32),
                ("status", C.c_int32), ("rnorm", C.c_double), ("arnorm", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Options(C.Structure):
    _fields_ = [("ls_atol", C.c_double), ("ls_rtol", C.c_double), ("ls_itmax", C.c_int64),
                ("ln_atol", C.c_double), ("ln_rtol", C.c_double), ("ln_btol", C.c_double),
                ("ln_conlim", C.c_double), ("ln_itmax", C.c_int64),
                ("ne_atol", C.c_double), ("ne_rtol", C.c_double), ("ne_etol", C.c_double),
                ("ne_itmax", C.c_int64), ("ne_conlim", C.c_double),
                ("ls_axtol", C.c_double), ("ls_btol", C.c_double), ("ls_etol", C.c_double),
                ("ls_conlim", C.c_double),
                ("fuse_two_rhs", C.c_int32), ("lookahead", C.c_int32), ("device", C.c_int32),
                ("jac_format", C.c_int32), ("ln_method", C.c_int32), ("kkt_method", C.c_int32)]


class Info(C.Structure):
    _fields_ = [("n", C.c_int64), ("m", C.c_int64), ("nnz", C.c_int64),
                ("spmv_a_blocks", C.c_int64), ("spmv_at_blocks", C.c_int64),
                ("last_solve_ms", C.c_double), ("last_spmv_ms", C.c_double),
                ("last_spmv_launches", C.c_int64), ("last_kernel_launches", C.c_int64),
                ("last_prod_a", C.c_int64 * 2), ("last_prod_at", C.c_int64 * 2), ("at_sorted", C.c_int64),
                ("comm_route", C.c_int64), ("last_fused_launches", C.c_int64),
                ("fuse_fallbacks", C.c_int64), ("wait_timeouts", C.c_int64), ("p2p_timeouts", C.c_int64),
                ("last_loop_iterations", C.c_int64), ("last_loop_launches", C.c_int64),
                ("last_multi_launches", C.c_int64), ("last_multi_iterations", C.c_int64),
                ("comm_in_launch_sums", C.c_int64)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["last_prod_a"] = list(d["last_prod_a"])
        d["last_prod_at"] = list(d["last_prod_at"])
        return d


class DenseInfo(C.Structure):
    _fields_ = [("n", C.c_int64), ("m", C.c_int64), ("last_syrk_ms", C.c_double), ("last_chol_ms", C.c_double),
                ("last_solve_ms", C.c_double), ("regularized_pivots", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class BandInfo(C.Structure):
    _fields_ = [("n", C.c_int64), ("m", C.c_int64), ("nnz", C.c_int64), ("nblocks", C.c_int64),
                ("bandwidth_blocks", C.c_int64), ("factor_bytes", C.c_int64), ("last_form_ms", C.c_double),
                ("last_chol_ms", C.c_double), ("last_solve_ms", C.c_double), ("regularized_pivots", C.c_int64),
                ("reordered", C.c_int64), ("chains", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/fpsq.h declares: (name, restype, argtypes)
_VP, _DP, _I32, _I64, _D = C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_double
SYMBOLS = [
    ("fpsq_version", C.c_char_p, []),
    ("fpsq_default_options", None, [_I64, _I64, C.POINTER(Options)]),
    ("fpsq_create", C.c_int, [C.POINTER(_VP), _I64, _I64, C.POINTER(Options)]),
    ("fpsq_destroy", C.c_int, [_VP]),
    ("fpsq_last_error", C.c_char_p, [_VP]),
    ("fpsq_set_jacobian_structure_coo", C.c_int, [_VP, _I64, _DP, _DP, _I32]),
    ("fpsq_set_jacobian_structure_csr", C.c_int, [_VP, _DP, _DP]),
    ("fpsq_set_jacobian_values", C.c_int, [_VP, _DP]),
    ("fpsq_set_input_stream", C.c_int, [_VP, _I32, _VP]),
    ("fpsq_set_output_ordering", C.c_int, [_VP, _I32]),
    ("fpsq_set_delta", C.c_int, [_VP, _D]),
    ("fpsq_solve_two_mixed", C.c_int, [_VP, _DP, _DP, _DP, _DP, _DP, _DP, C.POINTER(Stats)]),
    ("fpsq_solve_two_least_squares", C.c_int, [_VP, _DP, _DP, _DP, _DP, _DP, _DP, C.POINTER(Stats)]),
    ("fpsq_solve_two_extras", C.c_int, [_VP, _DP, _DP, _DP, _DP, C.POINTER(Stats)]),
    ("fpsq_ys_gs", C.c_int, [_VP, _DP, _DP, _D, _DP, _DP, _DP, _DP, C.POINTER(Stats)]),
    ("fpsq_jac_mul", C.c_int, [_VP, _I32, _D, _DP, _D, _DP]),
    ("fpsq_qp_create", C.c_int, [_VP, _DP, _DP, _DP, C.POINTER(_VP)]),
    ("fpsq_qp_destroy", C.c_int, [_VP]),
    ("fpsq_qp_objgrad", C.c_int, [_VP, _VP, _DP, _D, _D, _D, _DP, C.POINTER(C.c_double), _DP, _DP, _DP,
                                  C.POINTER(Stats)]),
    ("fpsq_qp_hprod", C.c_int, [_VP, _VP, _DP, _D, _D, _D, _I32, _DP, C.POINTER(Stats)]),
    ("fpsq_comm_unique_id", C.c_int, [_DP]),
    ("fpsq_comm_init", C.c_int, [_VP, _I32, _I32, _DP]),
    ("fpsq_comm_set_halo", C.c_int, [_VP, _I64, _I64]),
    ("fpsq_comm_set_route", C.c_int, [_VP, _I32]),
    ("fpsq_local_group_create", C.c_int, [_I32, C.POINTER(_VP)]),
    ("fpsq_local_group_destroy", C.c_int, [_VP]),
    ("fpsq_comm_init_local", C.c_int, [_VP, _VP, _I32]),
    ("fpsq_local_group_set_p2p", C.c_int, [_VP, _I32]),
    ("fpsq_dense_create", C.c_int, [C.POINTER(_VP), _I64, _I64, _I32]),
    ("fpsq_dense_destroy", C.c_int, [_VP]),
    ("fpsq_dense_last_error", C.c_char_p, [_VP]),
    ("fpsq_dense_set_jacobian", C.c_int, [_VP, _DP]),
    ("fpsq_dense_set_structure_coo", C.c_int, [_VP, _I64, _DP, _DP, _I32]),
    ("fpsq_dense_set_jacobian_coo", C.c_int, [_VP, _DP]),
    ("fpsq_dense_factorize", C.c_int, [_VP, _D, C.POINTER(C.c_int32)]),
    ("fpsq_dense_set_regularization", C.c_int, [_VP, _D, _D]),
    ("fpsq_dense_solve_two_mixed", C.c_int, [_VP, _DP, _DP, _DP, _DP, _DP, _DP]),
    ("fpsq_dense_solve_two_least_squares", C.c_int, [_VP, _DP, _DP, _DP, _DP, _DP, _DP]),
    ("fpsq_dense_get_factor", C.c_int, [_VP, _DP]),
    ("fpsq_dense_get_info", C.c_int, [_VP, C.POINTER(DenseInfo)]),
    ("fpsq_band_create", C.c_int, [C.POINTER(_VP), _I64, _I64, _DP, _DP, _I32]),
    ("fpsq_band_create_coo", C.c_int, [C.POINTER(_VP), _I64, _I64, _I64, _DP, _DP, _I32, _I32]),
    ("fpsq_band_factorize_coo", C.c_int, [_VP, _DP, _D, C.POINTER(C.c_int32)]),
    ("fpsq_band_analyze", C.c_int, [_I64, _I64, _DP, _DP, _DP, C.POINTER(BandInfo)]),
    ("fpsq_band_destroy", C.c_int, [_VP]),
    ("fpsq_band_last_error", C.c_char_p, [_VP]),
    ("fpsq_band_set_regularization", C.c_int, [_VP, _D, _D]),
    ("fpsq_band_factorize", C.c_int, [_VP, _DP, _D, C.POINTER(C.c_int32)]),
    ("fpsq_band_solve_two_mixed", C.c_int, [_VP, _DP, _DP, _DP, _DP, _DP, _DP]),
    ("fpsq_band_solve_two_least_squares", C.c_int, [_VP, _DP, _DP, _DP, _DP, _DP, _DP]),
    ("fpsq_band_get_info", C.c_int, [_VP, C.POINTER(BandInfo)]),
    ("fpsq_get_info", C.c_int, [_VP, C.POINTER(Info)]),
    ("fpsq_set_profiling", C.c_int, [_VP, _I32]),
    ("fpsq_debug_expect_iterations", C.c_int, [_VP, _I64]),
]

_LIB = None


def load():
    """dlopen libfpsq.so and type every entry point.  Raises if the library was not built.

    A process that also uses torch must `import torch` BEFORE the first load(): torch ships its own
    libamdhip64.so.7 / libhsa-runtime64 and two HIP runtimes in one process cannot both own the GPU; with torch
    imported first the dynamic linker resolves libfpsq's libamdhip64.so.7 to the copy already loaded."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                               "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


def producer_stream(*args):
    """The HIP stream on which torch is producing the CUDA tensors among `args` (torch's current stream), or None when
    no argument is a device tensor.  Passed to fpsq_set_input_stream so the library orders its reads after them."""
    for a in args:
        if getattr(a, "is_cuda", False):
            import torch

            raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
            if raw is not None:  # (the raw handle without building a Stream object: ~0.3 us instead of ~5 per call)
                return int(raw(a.device.index))
            return int(torch.cuda.current_stream(a.device).cuda_stream)
    return None


def ptr(a):
    """Raw address of a numpy array, a torch tensor (host or device), an int address, or None."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        assert a.is_contiguous()
        return a.data_ptr()
    raise TypeError(type(a))
