import os
import sys

import pytest
import torch  # noqa: F401  -- before libfpsq is loaded: one HIP runtime per process (see fps_amd/_lib.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "expects_wait_timeouts: the test provokes expired bounded waits on purpose "
                                       "(fpsq_info.fuse_fallbacks / wait_timeouts / p2p_timeouts may be non-zero)")


# Every handle a test of THIS process destroys is asked for its cumulative counters first (include/fpsq.h fpsq_info:
# fuse_fallbacks, wait_timeouts, p2p_timeouts): a call that was silently repeated on two launches per iteration, or a bounded
# wait that expired, fails the test that caused it -- unless the test says it provokes them (expects_wait_timeouts).
_COUNTER_LOG = []


def _wrap_destroy():
    import ctypes as C

    import fps_amd  # noqa: F401
    from fps_amd import _lib

    lib = _lib.load()
    if getattr(lib, "_fpsq_destroy_wrapped", False):
        return
    raw = lib.fpsq_destroy

    def destroy(h):
        i = _lib.Info()
        if h and lib.fpsq_get_info(h, C.byref(i)) == 0:
            _COUNTER_LOG.append((i.fuse_fallbacks, i.wait_timeouts, i.p2p_timeouts))
        return raw(h)

    lib.fpsq_destroy = destroy
    lib._fpsq_destroy_wrapped = True


@pytest.fixture(autouse=True)
def _no_hidden_retries(request):
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    _wrap_destroy()
    del _COUNTER_LOG[:]
    yield
    import gc

    gc.collect()   # (handles closed by __del__)
    bad = [c for c in _COUNTER_LOG if any(c)]
    if request.node.get_closest_marker("expects_wait_timeouts") is None:
        assert not bad, f"(fuse_fallbacks, wait_timeouts, p2p_timeouts) of the handles this test destroyed: {bad}"


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc

    orc.build()
    return orc
