"""CPU checks of the drop-in boundary: libfpsq.so loads, exports every symbol include/fpsq.h declares, and
refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

import fps_amd  # noqa: F401
from fps_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "fpsq.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fpsq_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"libfpsq.so does not export {name}"
    # and the Python binding types exactly that set
    assert sorted(n for n, _, _ in _lib.SYMBOLS) == names


def test_struct_layouts_match_header():
    assert C.sizeof(_lib.Stats) == 32
    assert C.sizeof(_lib.Options) == 8 * 17 + 24  # + ln_method, kkt_method
    assert _lib.Options.ln_method.offset == 8 * 17 + 16 and _lib.Options.kkt_method.offset == 8 * 17 + 20
    assert C.sizeof(_lib.Info) == 72 + 32


def test_default_options_follow_reference_defaults():
    lib = _lib.load()
    o = _lib.Options()
    lib.fpsq_default_options(100, 10, C.byref(o))
    se = 2.220446049250313e-16 ** 0.5
    # src/solve_two_systems_struct.jl:99-115
    assert o.ls_atol == se and o.ls_rtol == se and o.ls_itmax == 5 * 110
    assert o.ln_atol == se and o.ln_rtol == se and o.ln_btol == se and o.ln_conlim == 1 / se and o.ln_itmax == 550
    assert o.ne_atol == se and o.ne_rtol == se and o.ne_etol == se and o.ne_itmax == 0 and o.ne_conlim == 1 / se
    assert o.ln_method == 0 and o.fuse_two_rhs == 1  # CRAIG is the default least-norm workspace (struct.jl:121)
    assert o.kkt_method == 0  # LSQR + CRAIG, the reference's iterative path


def test_no_cpu_fallback_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _lib.load()
    h = C.c_void_p()
    rc = lib.fpsq_create(C.byref(h), 10, 1, None)
    assert rc < 0
    assert b"no HIP device" in lib.fpsq_last_error(None)
