"""CPU checks of the drop-in boundary: libfpsq.so loads, exports every symbol include/fpsq.h declares, and
refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
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
    # + at_sorted (round 3), comm_route, last_fused_launches (round 4), the three counters, the loop's two, comm_in_launch_sums (round 5)
    assert C.sizeof(_lib.Info) == 72 + 32 + 8 + 8 + 8 + 24 + 16 + 16 + 8
    assert _lib.Info.fuse_fallbacks.offset == 128 and _lib.Info.comm_in_launch_sums.offset == 184


def test_struct_sizes_against_the_c_compiler(tmp_path):
    """The ctypes mirrors against include/fpsq.h as gcc lays it out (sizes and the offsets of the last members)."""
    import subprocess

    src = tmp_path / "sz.c"
    src.write_text('#include <stddef.h>\n#include <stdio.h>\n#include "fpsq.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu\\n", '
                   "sizeof(fpsq_info), offsetof(fpsq_info, comm_in_launch_sums), sizeof(fpsq_options), sizeof(fpsq_stats), "
                   "sizeof(fpsq_dense_info), sizeof(fpsq_band_info)); return 0; }\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert got == [C.sizeof(_lib.Info), _lib.Info.comm_in_launch_sums.offset, C.sizeof(_lib.Options), C.sizeof(_lib.Stats),
                   C.sizeof(_lib.DenseInfo), C.sizeof(_lib.BandInfo)]


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


def test_band_symbolic_phase_on_the_host():
    """fpsq_band_analyze = the ordering decisions of fpsq_band_create without a device: a PDE-like Jacobian keeps its
    natural order (one chain when short, two chains when long and narrow: blocks alternately from the top and the bottom,
    a permutation); with its rows shuffled the natural band is full and reverse Cuthill-McKee restores a narrow one;
    malformed patterns are argument errors."""
    import numpy as np
    import scipy.sparse as sp

    from fps_amd import problems

    lib = _lib.load()

    def analyze(A):
        A = sp.csr_matrix(A)
        A.sort_indices()
        m, n = A.shape
        rp, ci = A.indptr.astype(np.int32), A.indices.astype(np.int32)
        perm = np.empty(m, dtype=np.int32)
        info = _lib.BandInfo()
        rc = lib.fpsq_band_analyze(n, m, rp.ctypes.data, ci.ctypes.data, perm.ctypes.data, C.byref(info))
        return rc, perm, info.as_dict()

    def block_bandwidth(A, perm):
        B = sp.csr_matrix(A)[perm]
        M = (abs(B) @ abs(B).T).tocoo()
        return int(np.max(np.abs(M.row // 128 - M.col // 128)))

    short = problems.pde_control_like(n=6000, m=600, per_row=24, window=512, seed=3).scipy_csr()
    rc, perm, i = analyze(short)
    assert rc == 0 and (i["chains"], i["reordered"]) == (1, 0) and np.array_equal(perm, np.arange(600))
    assert i["nblocks"] == 5 and i["bandwidth_blocks"] == block_bandwidth(short, perm)

    m = 7700
    long_ = problems.pde_control_like(n=30000, m=m, per_row=12, window=600, seed=11).scipy_csr()
    rc, perm, i = analyze(long_)
    assert rc == 0 and (i["chains"], i["reordered"]) == (2, 1)
    assert np.array_equal(np.sort(perm), np.arange(m))                      # a permutation
    assert np.array_equal(perm[:128], np.arange(128))                       # block 0: the top block
    assert np.array_equal(perm[128:256], m - 1 - np.arange(128))            # block 1: the bottom rows, descending
    assert np.array_equal(perm[256:384], 128 + np.arange(128))              # block 2: the second block from the top
    assert i["bandwidth_blocks"] == block_bandwidth(long_, perm) <= 6

    rng = np.random.default_rng(5)
    base = problems.pde_control_like(n=12000, m=2400, per_row=16, window=600, seed=8).scipy_csr()
    shuffled = sp.csr_matrix(base[rng.permutation(2400)])
    nat = block_bandwidth(shuffled, np.arange(2400))
    rc, perm, i = analyze(shuffled)
    assert rc == 0 and i["reordered"] == 1 and np.array_equal(np.sort(perm), np.arange(2400))
    assert nat >= 15 and i["bandwidth_blocks"] == block_bandwidth(shuffled, perm) <= 4

    rp = np.array([0, 2, 1], dtype=np.int32)
    ci = np.array([0, 1], dtype=np.int32)
    assert lib.fpsq_band_analyze(3, 2, rp.ctypes.data, ci.ctypes.data, None, None) == -1
    rp = np.array([0, 1, 2], dtype=np.int32)
    ci = np.array([0, 7], dtype=np.int32)
    assert lib.fpsq_band_analyze(3, 2, rp.ctypes.data, ci.ctypes.data, None, None) == -1
    assert b"column index out of range" in lib.fpsq_band_last_error(None)


def test_auto_backend_choice_follows_the_band(monkeypatch):
    """qdsolver_correspondence["ldlt"] -- the reference's key and default (parameters.jl:197, :290), fps_solve's default here
    too ("auto" is the same entry): the banded direct back-end when the symbolic phase (fpsq_band_analyze: host only)
    reports a narrow band of A A', the iterative one otherwise, and the object says which.  "iterative" is the reference's
    other key.  The constructors are stubbed: no device is needed for the decision."""
    import fps_amd  # noqa: F401
    from fps_amd import nlpmodels, problems, qdsolver

    grid = nlpmodels.EqQPModel(problems.aug2dc_like(N=30))            # grid incidence matrix: a few blocks wide
    rnd = nlpmodels.EqQPModel(problems.random_eqqp(n=20000, m=2000))  # random columns: A A' is dense
    a, r = qdsolver.band_analysis(grid), qdsolver.band_analysis(rnd)
    assert a["bandwidth_blocks"] <= qdsolver.AUTO_MAX_BAND_BLOCKS and a["nblocks"] == (grid.meta.ncon + 127) // 128
    assert r["bandwidth_blocks"] == r["nblocks"] - 1 > qdsolver.AUTO_MAX_BAND_BLOCKS
    made = []

    class Direct:
        def __init__(self, nlp, z, **kw):
            made.append(("hip_ldlt", kw))

    class Iterative:
        def __init__(self, nlp, z, **kw):
            made.append(("hip", kw))

    monkeypatch.setattr(qdsolver, "HIPBandedDirectQDSolver", Direct)
    monkeypatch.setattr(qdsolver, "HIPQDSolver", Iterative)
    reg = qdsolver.qdsolver_correspondence
    assert reg["ldlt"] is reg["auto"] and {"ldlt", "iterative", "hip", "hip_ldlt", "hip_direct"} <= set(reg)
    q = reg["ldlt"](grid, 0.0)
    assert isinstance(q, Direct) and q.qds_backend == "hip_ldlt" and q.qds_routed_from == "ldlt"
    q = reg["ldlt"](rnd, 0.0, ldlt_r2=-1e-8)
    assert isinstance(q, Iterative) and q.qds_backend == "hip"
    assert isinstance(reg["auto"](nlpmodels.HS6(), 0.0), Direct)   # every small model goes direct
    assert [k for k, _ in made] == ["hip_ldlt", "hip", "hip_ldlt"]
    assert "ldlt_r2" not in made[1][1]   # (LDLtSolver's keywords are not handed to the Krylov back-end)

    class Refuse:
        def __init__(self, nlp, z, **kw):
            raise qdsolver.FpsqError("does not fit")

    monkeypatch.setattr(qdsolver, "HIPBandedDirectQDSolver", Refuse)   # the device says no after all: iterative
    assert isinstance(reg["ldlt"](grid, 0.0), Iterative)
    from fps_amd.fps_solve import AlgoData
    assert AlgoData().qds_solver == "ldlt"   # src/parameters.jl:290


def test_ldlt_r2_defaults_to_the_reference_value():
    """`LDLtSolver(...; ldlt_r2 = -sqrt(eps))` (src/solve_two_systems_struct.jl:314) is the default of both direct back-ends;
    dropping a vanishing pivot (include/fpsq.h FPSQ_REG_DROP) is the explicit option "drop"."""
    import fps_amd  # noqa: F401
    from fps_amd import qdsolver

    se = float(np.sqrt(np.finfo(float).eps))
    assert qdsolver._ldlt_r2(None) == -se
    assert qdsolver._ldlt_r2(-1e-6) == -1e-6
    assert qdsolver._ldlt_r2("drop") == -qdsolver.REG_DROP
    with pytest.raises(ValueError):
        qdsolver._ldlt_r2("shift")
