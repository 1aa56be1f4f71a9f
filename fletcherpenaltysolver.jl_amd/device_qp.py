"""Device-resident penalty evaluation on the synthetic eq-QP model: the benchmark's unit of work.

One `objgrad(x)` = one `objgrad!(::FletcherPenaltyNLP, x, gx)` at a fresh x
(src/model-Fletcherpenaltynlp.jl:403-437) executed entirely on the MI355X through `fpsq_qp_objgrad`.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .qdsolver import FpsqError


class DeviceEqQP:
    """`comm`: None (single GPU), ("rccl", nranks, rank, id_bytes) or ("local", group_ptr, shard) for a row-sharded
    model; `qp` is then the rank's row block (distributed.shard_qp).  `halo` = (overlap_left, overlap_right): the
    n-vectors are column windows (distributed.shard_qp_halo / HaloPlan.overlaps) instead of replicated.
    `comm_route` ("auto" | "rccl" | "p2p", with comm = ("rccl", ...)): how the exchanges of the halo-sharded loop travel
    (include/fpsq.h fpsq_comm_set_route; "auto" = peer to peer over hipIpc-mapped buffers when every rank can, else RCCL);
    `info()["comm_route"]` says what the handle ended up with after its first solve."""

    ROUTES = {"auto": 0, "rccl": 1, "p2p": 2}
    ROUTE_NAMES = {0: "single GPU", 1: "rccl", 2: "p2p-ipc", 3: "local", 4: "local-p2p"}

    def __init__(self, qp, sigma=1e3, rho=1.0, delta=0.0, eta=0.0, device=0, comm=None, halo=None, comm_route=None,
                 **opt_overrides):
        self._lib = _lib.load()
        self.qp, self.sigma, self.rho, self.delta, self.eta = qp, sigma, rho, delta, eta
        opts = _lib.Options()
        self._lib.fpsq_default_options(qp.n, qp.m, C.byref(opts))
        opts.device = device
        for k, v in opt_overrides.items():
            setattr(opts, k, v)
        self.opts = opts
        h = C.c_void_p()
        if self._lib.fpsq_create(C.byref(h), qp.n, qp.m, C.byref(opts)) != 0:
            raise FpsqError(self._lib.fpsq_last_error(None).decode())
        self._h = h
        rp = np.ascontiguousarray(qp.rowptr, dtype=np.int32)
        ci = np.ascontiguousarray(qp.colind, dtype=np.int32)
        self._check(self._lib.fpsq_set_jacobian_structure_csr(h, rp.ctypes.data, ci.ctypes.data))
        self._check(self._lib.fpsq_set_jacobian_values(h, np.ascontiguousarray(qp.vals).ctypes.data))
        self._check(self._lib.fpsq_set_delta(h, float(delta)))
        q = C.c_void_p()
        self._check(self._lib.fpsq_qp_create(h, qp.qdiag.ctypes.data, qp.d.ctypes.data, qp.b.ctypes.data, C.byref(q)))
        self._q = q
        self.stats = (_lib.Stats * 2)()
        self.stats4 = (_lib.Stats * 4)()   # hprod with hessian_approx = 1: + the two recurrences of solve_two_extras
        self._in_stream = -1
        try:
            self._attach_comm(comm, halo, comm_route)
        except Exception:
            self.close()
            raise

    def _attach_comm(self, comm, halo, comm_route=None):
        h = self._h
        if comm is not None:
            if comm[0] == "rccl":
                _, nranks, rank, ident = comm
                buf = (C.c_uint8 * 128).from_buffer_copy(bytes(ident))
                self._check(self._lib.fpsq_comm_init(h, nranks, rank, C.addressof(buf)))
                if comm_route is not None:
                    self._check(self._lib.fpsq_comm_set_route(h, self.ROUTES[comm_route]))
            elif comm[0] == "local":
                self._check(self._lib.fpsq_comm_init_local(h, comm[1], comm[2]))
            else:
                raise ValueError(comm[0])
            if halo is not None:
                self._check(self._lib.fpsq_comm_set_halo(h, int(halo[0]), int(halo[1])))

    def _check(self, rc):
        if rc < 0:
            raise FpsqError(self._lib.fpsq_last_error(self._h).decode())
        return rc

    def _order(self, *args):
        """Device tensors among the arguments are produced on torch's current stream: register it so the library's
        stream waits for it (fpsq_set_input_stream; include/fpsq.h "INPUT READINESS").  No host synchronisation."""
        st = _lib.producer_stream(*args)
        if st is not None and st != self._in_stream:
            self._check(self._lib.fpsq_set_input_stream(self._h, 1, st))
            # torch consumes the device-resident outputs on that same stream: let the library order them there instead of
            # blocking the host until the last kernel has ended (include/fpsq.h, "STREAM-ORDERED OUTPUTS")
            self._check(self._lib.fpsq_set_output_ordering(self._h, 1))
            self._in_stream = st

    def set_delta(self, delta):
        self.delta = delta
        self._check(self._lib.fpsq_set_delta(self._h, float(delta)))

    def set_jacobian_values(self, vals):
        """`jac_coord!` output at a new x (src/solve_linear_system.jl:223-228), in the order of the structure call: numpy array,
        torch tensor (host or device) or raw address.  Device-resident values are read in place by ONE gather launch, ordered
        on torch's current stream (no host synchronisation): include/fpsq.h fpsq_set_jacobian_values."""
        self._order(vals)
        return self._check(self._lib.fpsq_set_jacobian_values(self._h, _lib.ptr(vals)))

    def set_profiling(self, on):
        self._check(self._lib.fpsq_set_profiling(self._h, int(on)))

    def objgrad(self, x, gx=None, ys=None, gs=None, xk=None):
        """x / gx / ys / gs / xk: numpy arrays, torch tensors (host or device) or raw addresses.
        Returns (fx, rc): rc > 0 flags a Krylov solve that stopped unsolved (the reference warns)."""
        fx = C.c_double()
        self._order(x, gx, ys, gs, xk)
        rc = self._check(self._lib.fpsq_qp_objgrad(self._h, self._q, _lib.ptr(x), self.sigma, self.rho, self.eta,
                                                   _lib.ptr(xk), C.byref(fx), _lib.ptr(gx), _lib.ptr(ys),
                                                   _lib.ptr(gs), self.stats))
        return fx.value, rc

    # -- the QDSolver seam on the same handle; every argument: numpy array, torch tensor (host or device) or address
    def solve_two_mixed(self, rhs1, rhs2, p1, q1, p2, q2):
        """src/solve_linear_system.jl:107-140 on the Jacobian the model holds (outputs are caller-owned buffers)."""
        self._order(rhs1, rhs2, p1, q1, p2, q2)
        return self._check(self._lib.fpsq_solve_two_mixed(self._h, _lib.ptr(rhs1), _lib.ptr(rhs2), _lib.ptr(p1),
                                                          _lib.ptr(q1), _lib.ptr(p2), _lib.ptr(q2), self.stats))

    def solve_two_least_squares(self, rhs1, rhs2, p1, q1, p2, q2):
        """src/solve_linear_system.jl:79-105: the two solves of every hprod! (two LSQR recurrences, fused)."""
        self._order(rhs1, rhs2, p1, q1, p2, q2)
        return self._check(self._lib.fpsq_solve_two_least_squares(self._h, _lib.ptr(rhs1), _lib.ptr(rhs2),
                                                                  _lib.ptr(p1), _lib.ptr(q1), _lib.ptr(p2),
                                                                  _lib.ptr(q2), self.stats))

    def ys_gs(self, g, c, gs, ys, v, w):
        """_compute_ys_gs! after the user-model evaluations (src/model-Fletcherpenaltynlp.jl:242-248)."""
        self._order(g, c, gs, ys, v, w)
        return self._check(self._lib.fpsq_ys_gs(self._h, _lib.ptr(g), _lib.ptr(c), self.sigma, _lib.ptr(gs),
                                                _lib.ptr(ys), _lib.ptr(v), _lib.ptr(w), self.stats))

    def jac_mul(self, trans, alpha, x, beta, y):
        """y = alpha op(A) x + beta y with the model's Jacobian (fpsq_jac_mul; trans = 0: A, 1: A')."""
        self._order(x, y)
        return self._check(self._lib.fpsq_jac_mul(self._h, int(trans), float(alpha), _lib.ptr(x), float(beta), _lib.ptr(y)))

    def hprod(self, v, Hv, hessian_approx=2):
        """hprod!(::FletcherPenaltyNLP, x, v, Hv) on the device, hessian_approx = Val(2) (model-Fletcherpenaltynlp.jl:521-570)
        or Val(1) (:572-634: additionally the solve_two_extras lanes; their statistics land in self.stats4[2:4]); the model
        is quadratic with linear constraints, so the product does not depend on x.  Returns rc."""
        self._order(v, Hv)
        rc = self._check(self._lib.fpsq_qp_hprod(self._h, self._q, _lib.ptr(v), self.sigma, self.rho, self.eta,
                                                 int(hessian_approx), _lib.ptr(Hv), self.stats4))
        self.stats[0], self.stats[1] = self.stats4[0], self.stats4[1]
        return rc

    def info(self):
        i = _lib.Info()
        self._check(self._lib.fpsq_get_info(self._h, C.byref(i)))
        return i.as_dict()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fpsq_qp_destroy(self._q)
            self._lib.fpsq_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rccl_unique_id() -> bytes:
    """128-byte RCCL id (call on rank 0, broadcast, pass to every rank's DeviceEqQP(comm=("rccl", ...)))."""
    lib = _lib.load()
    buf = (C.c_uint8 * 128)()
    if lib.fpsq_comm_unique_id(C.addressof(buf)) != 0:
        raise FpsqError(lib.fpsq_last_error(None).decode())
    return bytes(buf)


class LocalGroup:
    """In-process stand-in for RCCL: `nshards` row-shard models on ONE GPU, one host thread each."""

    def __init__(self, nshards, p2p=False):
        """p2p: the shards exchange peer to peer (records written into the peers' buffers + sequence flags, no collective call
        in the Krylov loop: include/fpsq.h fpsq_local_group_set_p2p) instead of through event-ordered copy kernels."""
        self._lib = _lib.load()
        g = C.c_void_p()
        if self._lib.fpsq_local_group_create(nshards, C.byref(g)) != 0:
            raise FpsqError("local_group_create failed")
        self.ptr, self.nshards = g, nshards
        if p2p:
            self._lib.fpsq_local_group_set_p2p(g, 1)

    def run(self, fns):
        """Run one callable per shard concurrently (collectives rendezvous across the threads)."""
        import threading

        out, err = [None] * len(fns), [None] * len(fns)

        def work(i):
            try:
                out[i] = fns[i]()
            except BaseException as e:  # noqa: BLE001
                err[i] = e

        ts = [threading.Thread(target=work, args=(i,)) for i in range(len(fns))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for e in err:
            if e is not None:
                raise e
        return out

    def close(self):
        if self.ptr:
            self._lib.fpsq_local_group_destroy(self.ptr)
            self.ptr = None
