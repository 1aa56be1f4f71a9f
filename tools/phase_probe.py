"""Developer probe (GPU box): per-workgroup phase timestamps of the A' product (s_memrealtime, 100 MHz) from a library built
with -DFPSQ_PHASE_PROBE (tools/experiments/phase_probe.patch): 0 entry, 1 stream landed, 2 gathers landed, 3 products in
LDS (barrier), 4 row sums + stores issued, 5 end (stores acknowledged).  The buffer keeps the LAST A' launch of the call."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FPSQ_LIB_PATH", os.path.join(ROOT, "tools/ab/libfpsq_probe.so"))
from fps_amd import problems, _lib
from fps_amd.device_qp import DeviceEqQP

qp = problems.pde_control_like(n=1_000_000, m=100_000)
dev = DeviceEqQP(qp, sigma=1e3, rho=1.0, delta=0.0, device=0)
lib = ctypes.CDLL(os.environ["FPSQ_LIB_PATH"])
nblk = dev.info()["spmv_at_blocks"]
d = torch.device("cuda", 0)
buf = torch.zeros(nblk * 8, dtype=torch.int64, device=d)
x = torch.from_numpy(qp.point(1)).to(d)
gx = torch.empty(qp.n, dtype=torch.float64, device=d)
for _ in range(3):
    dev.objgrad(x, gx=gx)
assert lib.fpsq_debug_set_probe(ctypes.c_void_p(buf.data_ptr())) == 0
dev.objgrad(x, gx=gx)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(nblk, 8)[:, :6].astype(np.float64) * 0.01  # us
t0 = t[:, 0].min()
print("sorted layout:", dev.info()["at_sorted"], " blocks:", nblk)
print("launch span (first entry -> last end): %.2f us" % (t[:, 5].max() - t0))
names = ["entry -> stream landed", "stream -> gathers landed", "gathers -> products in LDS (barrier)", "phase 2 (row sums, stores issued)",
         "partial reduce + stores acknowledged"]
for i, nm in enumerate(names):
    dlt = t[:, i + 1] - t[:, i]
    print("  %-44s mean %6.2f  median %6.2f  p90 %6.2f us" % (nm, dlt.mean(), np.median(dlt), np.percentile(dlt, 90)))
life = t[:, 5] - t[:, 0]
print("  %-44s mean %6.2f  median %6.2f  p90 %6.2f us" % ("workgroup life", life.mean(), np.median(life), np.percentile(life, 90)))
# resident workgroups over time, and how many of them are waiting for their stream
grid = np.linspace(t0, t[:, 5].max(), 200)
res = [(np.sum((t[:, 0] <= g) & (t[:, 5] > g)), np.sum((t[:, 0] <= g) & (t[:, 1] > g))) for g in grid]
res = np.array(res)
print("resident workgroups: mean %.0f (max %d); of them in the stream phase: mean %.0f" % (res[:, 0].mean(), res[:, 0].max(), res[:, 1].mean()))
start = np.sort(t[:, 0] - t0)
print("entry times: 25%% %.2f  50%% %.2f  75%% %.2f  100%% %.2f us" % tuple(np.percentile(start, [25, 50, 75, 100])))
