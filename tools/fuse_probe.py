"""Developer tool: where the time of ONE fused-iteration launch (k_iter_fused) goes, from the per-workgroup time stamps the library
leaves when FPSQ_FUSE_PROBE=<file> is set (100 MHz counter; stamps 0 = entry, 1 = dependences met / record taken, 2 = tiles done /
partials summed, 3 = exit).  usage (on the GPU box):
  FPSQ_FUSE_PROBE=gpurun_out/fuse_probe.txt python3 bench.py --steps 10 --warmup 2 --cpu-evals 0 --no-roofline-pass
  python3 tools/fuse_probe.py gpurun_out/fuse_probe.txt"""
import sys
import numpy as np

lines = open(sys.argv[1]).read().split("\n")
head = [int(v) for v in lines[0].split()]
grid, layout = head[0], head[1:]
st = np.array([[int(v) for v in l.split()] for l in lines[1:1 + grid]], dtype=np.float64)
names = ["head leaders", "A' workgroups", "mid leaders", "row groups of A", "updates with A'", "updates with A"]
t0 = st[:, 0][st[:, 0] > 0].min()
us = lambda v: (v - t0) / 100.0
print(f"launch: {grid} workgroups; first entry -> last exit {us(st[:, 3].max()):.1f} us")
o = 0
for nm, cnt in zip(names, layout):
    s = st[o:o + cnt]
    o += cnt
    live = s[(s[:, 0] > 0) & (s[:, 3] > 0)]
    if live.size == 0:
        print(f"{nm:18s} {cnt:5d} workgroups, none ran")
        continue
    e, x = us(live[:, 0]), us(live[:, 3])
    line = (f"{nm:18s} {len(live):5d} ran: entry {e.min():6.1f} .. {np.median(e):6.1f} .. {e.max():6.1f} us, exit {x.min():6.1f} .. "
            f"{np.median(x):6.1f} .. {x.max():6.1f} us, life median {np.median(x - e):5.1f} (max {np.max(x - e):5.1f})")
    print(line)
    for k, what in ((1, "stamp 1"), (2, "stamp 2")):
        ok = live[live[:, k] > 0]
        if ok.size:
            d = us(ok[:, k])
            print(f"{'':18s}   {what}: {d.min():6.1f} .. {np.median(d):6.1f} .. {d.max():6.1f} us; after entry median {np.median(d - us(ok[:, 0])):5.1f} max {np.max(d - us(ok[:, 0])):5.1f}")
# machine occupancy over time: workgroups resident per microsecond
T = int(us(st[:, 3].max())) + 1
occ = np.zeros(T + 1)
kinds = np.zeros((len(layout), T + 1))
o = 0
for i, cnt in enumerate(layout):
    for r in st[o:o + cnt]:
        if r[0] > 0 and r[3] > 0:
            a, b = int(us(r[0])), int(us(r[3]))
            kinds[i, a:b + 1] += 1
    o += cnt
print("resident workgroups by kind, every 4 us:")
print("  t(us) " + " ".join(f"{n.split()[0][:6]:>7s}" for n in names))
for t in range(0, T + 1, 4):
    print(f"  {t:5d} " + " ".join(f"{int(kinds[i, t]):7d}" for i in range(len(layout))))
