"""Developer tool: where the time of one evaluation goes, from a rocprofv3 --kernel-trace CSV.
usage (on the GPU box):
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python3 bench.py --steps 10 --warmup 2 --cpu-evals 0 --no-roofline-pass
  python3 tools/timeline.py gpurun_out/tl/t_kernel_trace.csv
Evaluations are delimited by k_qp_grad launches; prints, for the median evaluation, kernel time and the idle gaps
between consecutive kernels by (previous -> next) kind, and the gap between evaluations (host turn-around)."""
import csv
import sys
from collections import defaultdict

import numpy as np


def short(name):
    if "k_iter_fused" in name:
        return "k_iter_fused"
    if "k_spmv_atl" in name:
        return "k_spmv_atl"
    if "k_spmv_rgcs" in name and name.split("(")[0].rstrip(">").endswith("true"):
        return "k_spmv_rgcs(lead)"
    for k in ("k_spmv_rgcs", "k_spmv", "k_step", "k_qp_grad", "k_qp_penalty_grad", "k_startup", "k_ys", "k_updates",
              "k_minres_ew", "k_qp_hsv", "k_qp_hprod_fin", "k_gather", "k_persist", "k_axpby", "k_presum", "k_qp_fx",
              "k_halo_finish", "k_p2p_gather", "k_p2p_halo"):
        if "fpsq::" + k + "<" in name or "fpsq::" + k + "(" in name or "::" + k + "<" in name or "::" + k + "(" in name:
            return k
    return name.split("(")[0][-30:]


rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
marker = sys.argv[2] if len(sys.argv) > 2 else "k_qp_grad"
starts = [i for i, r in enumerate(rows) if r[2] == marker]
evals = []
for a, b in zip(starts[:-1], starts[1:]):
    ks = rows[a:b]
    kern = defaultdict(float)
    cnt = defaultdict(int)
    gaps = defaultdict(float)
    gcnt = defaultdict(int)
    for i, (s, e, k) in enumerate(ks):
        kern[k] += (e - s) / 1e3
        cnt[k] += 1
        if i + 1 < len(ks):
            g = (ks[i + 1][0] - e) / 1e3
            key = f"{k} -> {ks[i + 1][2]}"
            gaps[key] += g
            gcnt[key] += 1
    span = (ks[-1][1] - ks[0][0]) / 1e3
    turn = (rows[b][0] - ks[-1][1]) / 1e3
    evals.append((span + turn, span, turn, kern, cnt, gaps, gcnt))
evals = evals[len(evals) // 4:]  # drop warm-up
tot = np.array([e[0] for e in evals])
mid = evals[int(np.argsort(tot)[len(tot) // 2])]
print(f"{len(evals)} evaluations; start-to-start median {np.median(tot):.1f} us (min {tot.min():.1f}, max {tot.max():.1f})")
print(f"median evaluation: first kernel start -> last kernel end {mid[1]:.1f} us, then {mid[2]:.1f} us until the next evaluation starts")
print("kernel time:")
for k, v in sorted(mid[3].items(), key=lambda kv: -kv[1]):
    print(f"  {k:22s} {mid[4][k]:4d} x {v / mid[4][k]:7.2f} us = {v:8.1f} us")
print(f"  total {sum(mid[3].values()):.1f} us")
print("idle gaps between consecutive kernels:")
for k, v in sorted(mid[5].items(), key=lambda kv: -kv[1]):
    print(f"  {k:44s} {mid[6][k]:4d} x {v / mid[6][k]:6.2f} us = {v:7.1f} us")
print(f"  total {sum(mid[5].values()):.1f} us")
