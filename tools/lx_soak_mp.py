"""Soak of the round-5 sharded path BETWEEN PROCESSES on one GPU: NRANKS processes (one rank each, as the product runs -- one
process per GPU -- except that here they share this box's device), peer-to-peer route over hipIpc-mapped buffers, sums over the
ranks inside the launches (FPSQ_LX=2; FPSQ_LX=0 in the environment: round 4's gather kernels, what ranks sharing a device run by
default), one launch per joint iteration with the halo exchange and finish inside (FPSQ_FUSE_ITER=2; 0: three launches).

    python tools/lx_soak_mp.py [evaluations=1000] [nranks=3] [delta=0] [n=200000]

Every rank evaluates the same E points; the parent then checks: return codes 0 everywhere, phi BITWISE equal on every rank, the
iteration counts equal on every rank, the overlap rows' checksums equal on the ranks sharing them, phi within 1e-9 of a single-GPU
handle on a sample of the points, and the handles' wait counters (fuse_fallbacks, wait_timeouts, p2p_timeouts) all zero.
Set-up collectives go through the loopback stand-in for librccl (tests/shim; RCCL refuses two ranks on one device).
LX_SOAK_KEEP=1: the overlap rows of gx / gs of every evaluation are kept, and for an evaluation whose ranks disagree the rows that are
off the single-GPU handle's are listed per rank.  LX_SOAK_CHUNKS=1: EVERY row of gx of every evaluation against the single-GPU
handle, 64 rows at a time.  FPSQ_FUSE_ITER=0 in the environment: three launches per iteration.
(Round 5, three ranks at n = 100000: ~1-2 evaluations in 10^3 came back with a few 128-byte lines of the overlap rows of gx / gs
wrong on ONE of the two ranks sharing them -- phi, iteration counts and return codes all fine.  Cause: see atl_product's comment on
the yin prefetch in csrc/fpsq_spmv.hip.h.  profiles/r05_soak.txt has the runs before and after.)"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def problem(n):
    from fps_amd import problems
    return problems.pde_control_like(n=n, m=n // 10, per_row=40, window=2048, seed=41)


def point(qp, e):
    rng = np.random.default_rng(1000 + e)
    if os.environ.get("LX_SOAK_POINTS") == "near":  # distances over four decades: the iteration counts move from call to call, so
        scale = 0.5 ** (e % 7) * (1.0 if e % 3 else 1e-2)  # the run-ahead's expected count is wrong again and again
        return qp.xhat + scale * rng.standard_normal(qp.n)
    return qp.xhat + (0.3 * 0.97 ** (e % 60)) * rng.standard_normal(qp.n)


def worker():
    rank, nranks, d, E, delta, n = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), float(sys.argv[6]), int(sys.argv[7])
    import torch  # noqa: F401
    import fps_amd  # noqa: F401
    from fps_amd.device_qp import DeviceEqQP, rccl_unique_id
    from fps_amd.distributed import halo_plan, row_partition, shard_qp_halo
    idf = os.path.join(d, "id.bin")
    if rank == 0:
        with open(idf + ".tmp", "wb") as f:
            f.write(rccl_unique_id())
        os.rename(idf + ".tmp", idf)
    t0 = time.time()
    while not os.path.exists(idf):
        if time.time() - t0 > 120:
            sys.exit("no unique id after 120 s")
        time.sleep(0.01)
    ident = open(idf, "rb").read()
    qp = problem(n)
    plan = halo_plan(qp.rowptr, qp.colind, qp.n, row_partition(qp.rowptr, nranks))
    loc = shard_qp_halo(qp, plan, rank)
    dev = DeviceEqQP(loc, sigma=1e3, rho=1.0, delta=delta, comm=("rccl", nranks, rank, ident), halo=plan.overlaps(rank), comm_route="p2p")
    w = plan.window(rank)
    tl, tr = plan.overlaps(rank)
    rec = np.zeros((E, 9))
    hv = np.empty(loc.n)
    fused = 0
    gx, ys, gs = np.empty(loc.n), np.empty(loc.m), np.empty(loc.n)
    keep = os.environ.get("LX_SOAK_KEEP") == "1"  # (diagnosis: the overlap rows of gx and gs of every evaluation)
    ov = np.zeros((E, 2, tl + tr)) if keep else None
    nch = (loc.n + 63) // 64
    chunks = np.zeros((E, nch)) if os.environ.get("LX_SOAK_CHUNKS") == "1" else None  # (diagnosis: sums of gx over every 64 rows of the window)
    for e in range(E):
        x = np.ascontiguousarray(point(qp, e)[w])
        try:
            f, rc = dev.objgrad(x, gx=gx, ys=ys, gs=gs if keep else None)
        except Exception as exc:
            i = dev.info()
            print(f"rank {rank} evaluation {e}: {exc}; counters {(i['fuse_fallbacks'], i['wait_timeouts'], i['p2p_timeouts'])}", flush=True)
            sys.exit(3)
        rec[e, :7] = (f, rc, dev.stats[0].niter, dev.stats[1].niter, gx[:tl].sum() if tl else 0.0, gx[loc.n - tr:].sum() if tr else 0.0, ys.sum())
        if e % 10 == 9:  # an hprod! Val(2) too: the checksums of its overlap rows
            dev.hprod(np.ascontiguousarray(point(qp, e + 7)[w] - qp.xhat[w]), hv, 2)
            rec[e, 7:] = (hv[:tl].sum() if tl else 0.0, hv[loc.n - tr:].sum() if tr else 0.0)
        fused += dev.info()["last_fused_launches"]
        if chunks is not None:
            chunks[e] = np.add.reduceat(gx, np.arange(0, loc.n, 64))
        if keep:
            ov[e, 0] = np.concatenate([gx[:tl], gx[loc.n - tr:]])
            ov[e, 1] = np.concatenate([gs[:tl], gs[loc.n - tr:]])
        if rank == 0 and e % 250 == 249:
            print(f"{e + 1} evaluations, {time.time() - t0:.0f} s", flush=True)
    i = dev.info()
    np.savez(os.path.join(d, f"soak_{rank}.npz"), rec=rec, info=np.array([i["comm_route"], i["comm_in_launch_sums"], i["fuse_fallbacks"],
                                                                           i["wait_timeouts"], i["p2p_timeouts"], fused]))
    if keep:
        np.save(os.path.join(d, f"ov_{rank}.npy"), ov)
    if chunks is not None:
        np.save(os.path.join(d, f"chunks_{rank}.npy"), chunks)
    dev.close()


def main():
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    delta = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 200_000
    import tempfile
    d = tempfile.mkdtemp(prefix="lx_soak_")
    so = os.path.join(ROOT, "tests", "shim", "libloopback_rccl.so")
    src = os.path.join(ROOT, "tests", "shim", "loopback_rccl.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "-Wno-unused-result", "-o", so, src])
    env = dict(os.environ, FPSQ_RCCL_LIB=so, FPSQ_SHIM_TIMEOUT="120", HSA_ENABLE_IPC_MODE_LEGACY="0", FPSQ_LX=os.environ.get("FPSQ_LX", "2"), FPSQ_FUSE_ITER=os.environ.get("FPSQ_FUSE_ITER", "2"))
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank", str(r), str(P), d, str(E), repr(delta), str(n)], env=env)
             for r in range(P)]
    rcs = []
    try:
        for p in procs:
            rcs.append(p.wait(timeout=900))
    finally:
        for p in procs:  # (exactly the processes started here)
            if p.poll() is None:
                p.kill()
    if any(rcs):
        print(f"ranks ended with {rcs}")
        sys.exit(1)
    res = [np.load(os.path.join(d, f"soak_{r}.npz")) for r in range(P)]
    rec = [r["rec"] for r in res]
    bad = {"rc": int(sum((q[:, 1] != 0).sum() for q in rec)),
           "phi": int(sum((q[:, 0] != rec[0][:, 0]).sum() for q in rec)),
           "iters": int(sum((q[:, 2:4] != rec[0][:, 2:4]).any(axis=1).sum() for q in rec)),
           "overlap": int(sum((rec[r][:, 5] != rec[r + 1][:, 4]).sum() for r in range(P - 1))),
           "overlap_hprod": int(sum((rec[r][:, 8] != rec[r + 1][:, 7]).sum() for r in range(P - 1)))}
    import torch  # noqa: F401
    import fps_amd  # noqa: F401
    from fps_amd.device_qp import DeviceEqQP
    from fps_amd.distributed import halo_plan, row_partition
    qp = problem(n)
    ref = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=delta)
    keep = os.environ.get("LX_SOAK_KEEP") == "1"
    plan = halo_plan(qp.rowptr, qp.colind, qp.n, row_partition(qp.rowptr, P))
    ovs = [np.load(os.path.join(d, f"ov_{r}.npy")) for r in range(P)] if keep else None
    for r in range(P - 1):
        for e in np.nonzero(rec[r][:, 5] != rec[r + 1][:, 4])[0]:
            print(f"overlap of ranks {r} | {r + 1}, evaluation {e}: checksums {rec[r][e, 5]!r} | {rec[r + 1][e, 4]!r}, iterations {rec[r][e, 2:4]}")
            if keep:
                g_ref, gs_ref, ys_ref = np.empty(qp.n), np.empty(qp.n), np.empty(qp.m)
                xe = point(qp, e)
                ref.objgrad(xe, gx=g_ref, gs=gs_ref, ys=ys_ref)
                A = qp.scipy_csr()
                bounds = row_partition(qp.rowptr, P)
                grad = qp.qdiag * xe + qp.d
                for q in (r, r + 1):
                    w = plan.window(q)
                    tl, tr = plan.overlaps(q)
                    for nm, k, full in (("gx", 0, g_ref), ("gs", 1, gs_ref)):
                        want = np.concatenate([full[w][:tl], full[w][full[w].size - tr:]])
                        dlt = np.abs(ovs[q][e, k] - want)
                        wrong = np.nonzero(dlt > 1e-8 * np.max(np.abs(want)))[0]
                        print(f"   rank {q} {nm}: overlap rows (head {tl}, tail {tr}) off the single-GPU handle's: {wrong.size}"
                              + (f", indices {wrong[:6]} .. {wrong[-3:]}, largest {dlt.max():.3e} (scale {np.max(np.abs(want)):.3e})" if wrong.size else ""))
                        if wrong.size and nm == "gs":
                            print(f"      lines of 8 rows: {sorted(set((wrong // 8).tolist()))}")
                        if wrong.size and nm == "gs":  # gs = grad f - A' ys: what was added up in the rows that are off?
                            cols = np.concatenate([np.arange(w.start, w.start + tl), np.arange(w.stop - tr, w.stop)])
                            own = (A[bounds[q]:bounds[q + 1]].T @ ys_ref[bounds[q]:bounds[q + 1]])[cols]
                            tot = (A.T @ ys_ref)[cols]
                            used = grad[cols] - ovs[q][e, 1]
                            if e > 0:   # the previous evaluation's rows?
                                gs_prev = np.empty(qp.n)
                                ref.objgrad(point(qp, e - 1), gs=gs_prev)
                                same_prev = np.abs(ovs[q][e, 1][wrong] - gs_prev[cols][wrong]) <= 1e-9 * np.max(np.abs(gs_prev))
                                print(f"      of the {wrong.size} rows, {int(same_prev.sum())} carry the PREVIOUS evaluation's gs")
                            for i in wrong[:4]:
                                print(f"      row {i}: A'ys used {used[i]:.6e}, true {tot[i]:.6e}; this rank's part {own[i]:.6e}, the neighbour's {tot[i] - own[i]:.6e}; "
                                      f"used - own = {used[i] - own[i]:.6e}")
    if os.environ.get("LX_SOAK_CHUNKS") == "1":  # every evaluation against the single-GPU handle, 64 rows at a time
        ch = [np.load(os.path.join(d, f"chunks_{r}.npy")) for r in range(P)]
        g_ref = np.empty(qp.n)
        nbad = 0
        for e in range(E):
            ref.objgrad(point(qp, e), gx=g_ref)
            scale = np.max(np.abs(g_ref))
            for r in range(P):
                w = plan.window(r)
                want = np.add.reduceat(g_ref[w], np.arange(0, w.stop - w.start, 64))
                wrong = np.nonzero(np.abs(ch[r][e] - want) > 1e-7 * scale)[0]
                if wrong.size:
                    nbad += 1
                    tl, tr = plan.overlaps(r)
                    print(f"evaluation {e} rank {r} (window of {w.stop - w.start} rows, head {tl}, tail {tr}): chunks of 64 rows off: {wrong.size}: {wrong[:12]}"
                          f"{' ..' if wrong.size > 12 else ''}; largest {np.max(np.abs(ch[r][e] - want)):.3e} (scale {scale:.3e})")
        print(f"(evaluation, rank) pairs with rows of gx off the single-GPU handle's: {nbad}")
    off, val = 0, 0
    sample = list(range(0, E, max(1, E // 100)))
    for e in sample:
        f, _ = ref.objgrad(point(qp, e))
        same = (ref.stats[0].niter, ref.stats[1].niter) == tuple(int(t) for t in rec[0][e, 2:4])
        off += not same
        val += abs(f - rec[0][e, 0]) > (1e-9 if same else 1e-6) * abs(f)
    ref.close()
    bad["value"] = int(val)
    infos = [[int(t) for t in r["info"]] for r in res]
    its, cnt = np.unique(rec[0][:, 2:4], axis=0, return_counts=True)
    print(f"iteration counts seen: {[(tuple(int(t) for t in i), int(c)) for i, c in zip(its, cnt)]}")
    print(f"{E} evaluations on {P} ranks in {P} processes (n = {n}, delta = {delta}): mismatches {bad}; of {len(sample)} sampled points "
          f"{off} stopped one iteration apart from the single-GPU handle; per rank (route, in-launch sums, fuse_fallbacks, wait_timeouts, "
          f"p2p_timeouts, fused launches) {infos}; {time.time() - t0:.0f} s")
    sys.exit(1 if any(bad.values()) or any(any(i[2:5]) for i in infos) else 0)


if __name__ == "__main__":
    worker() if len(sys.argv) > 1 and sys.argv[1] == "--rank" else main()
