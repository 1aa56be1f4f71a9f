"""Post-process the two rocprofv3 PMC passes into profiles/rNN_pmc_traffic.json.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 2 --warmup 1 --cpu-evals 0
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 2 --warmup 1 --cpu-evals 0
  python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv J > profiles/r02_pmc_traffic.json

J = joint Krylov iterations per evaluation (max of the LSQR / CRAIG counts bench.py prints).  Corrections per
MI355X_MICROARCH.md (HBM section): FETCH_SIZE is in KB and reports half of the bytes of coalesced streaming reads on
gfx950 (hbm_read = 2 * 1024 * FETCH_SIZE); WRITE_SIZE is exact (hbm_write = 1024 * WRITE_SIZE)."""
import collections, csv, hashlib, json, os, subprocess, sys

fetch_csv, write_csv, J = sys.argv[1], sys.argv[2], int(sys.argv[3])
WORKLOAD = sys.argv[4] if len(sys.argv) > 4 else "pde-control-hashed n=1e6 m=1e5 nnz=1e7"  # (bench.py's default)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_sources_sha16():
    """sha256 over the kernel sources the profile was taken on (csrc/*.hip, *.hip.h, sorted by name): bench.py computes the same
    and REFUSES a profile whose kernels are no longer the ones it runs (roofline.traffic = null, traffic_source says why)."""
    d = os.path.join(ROOT, "fletcherpenaltysolver.jl_amd", "csrc")
    hh = hashlib.sha256()
    for fn in sorted(os.listdir(d)):
        if fn.endswith(".hip") or fn.endswith(".hip.h"):
            hh.update(fn.encode())
            hh.update(open(os.path.join(d, fn), "rb").read())
    return hh.hexdigest()[:16]
n, m, nnz = 1_000_000, 100_000, 10_000_000


def load(path, floor):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        v = float(r["Counter_Value"])
        if v > floor:  # launches past convergence move nothing
            d[r["Kernel_Name"]].append(v)
    return d


f, w = load(fetch_csv, 1000.0), load(write_csv, 100.0)


def pb(nrhs):
    a = 12 * nnz + 4 * (m + 1) + 8 * nrhs * (n + 2 * m)
    at = 12 * nnz + 4 * (n + 1) + 8 * nrhs * (m + 2 * n)
    return a, at


a1, at1 = pb(1)
a2, at2 = pb(2)
upd_at, upd_a = 8 * 5 * m, 8 * (3 * n + 5 * m)
# One evaluation (fast start + paired epilogue product, bench.py default): J + 1 two-RHS A products (the start-up one
# without riding updates), J two-RHS A' products with the LSQR update riding + ONE raw two-RHS A' product (no yin read).
fused = any("k_iter_fused" in k for k in f)
if fused:
    # one launch per joint iteration (k_iter_fused: A' product + LSQR update + A product + CRAIG updates), the start-up A product
    # and the raw two-RHS A' product of the epilogue on their own
    alg = {"k_iter_fused": (J * (at2 + a2 + upd_a) + (J - 1) * upd_at) / J, "k_spmv_rgcs<2": float(a2),
           "k_spmv<2, 1": float(at2 - 8 * 2 * n)}
    mix = {"k_iter_fused": J, "k_spmv_rgcs<2": 1, "k_spmv<2, 1": 1}
    family = {"k_iter_fused": ("k_iter_fused",), "k_spmv_rgcs<2": ("k_spmv_rgcs<2",), "k_spmv<2, 1": ("k_spmv<2, 1", "k_spmv_atl<")}
else:
    alg = {  # average algorithmic bytes of a productive launch (products + the vector updates riding in them)
        "k_spmv_rgcs<2": (J * (a2 + upd_a) + a2) / (J + 1),
        "k_spmv<2, 1": (J * at2 + upd_at * (J - 1) + (at2 - 8 * 2 * n)) / (J + 1),
    }
    mix = {"k_spmv_rgcs<2": J + 1, "k_spmv<2, 1": J + 1}
    # (the loop's products are the variants with riding leaders -- k_spmv_rgcs<.., LEAD>, k_spmv_atl -- the start-up and
    # epilogue products the plain ones: one kernel family per matrix)
    family = {"k_spmv_rgcs<2": ("k_spmv_rgcs<2",), "k_spmv<2, 1": ("k_spmv<2, 1", "k_spmv_atl<")}
try:
    head = os.environ.get("FPSQ_GIT_HEAD") or subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except OSError:
    head = None
out = {"_how": __doc__.strip().split("\n\n")[-1].replace("\n", " "), "workload": WORKLOAD,
       "kernel_sources_sha16": kernel_sources_sha16(), "git_head_when_post_processed": head,
       "joint_iterations": J, "fused_iterations": fused, "kernels": {}}
tb = ab = 0.0
for key in alg:
    kn = [k for k in f if any(pat in k for pat in family[key])]
    assert len(kn) >= 1, (key, list(f))
    fv = [v for k in kn for v in f[k]]
    wv = [v for k in kn for v in w[k]]
    hbm = 2 * 1024 * sum(fv) / len(fv) + 1024 * sum(wv) / len(wv)
    out["kernels"][key] = {"productive_launches": len(fv), "FETCH_SIZE_KB": round(sum(fv) / len(fv), 1),
                           "WRITE_SIZE_KB": round(sum(wv) / len(wv), 1), "hbm_bytes": round(hbm),
                           "algorithmic_bytes": round(alg[key]), "hbm_over_algorithmic": round(hbm / alg[key], 3)}
    tb += mix[key] * hbm
    ab += mix[key] * alg[key]
tot = sum(mix.values())
out["per_evaluation_mix"] = mix
out["traffic_bytes_per_productive_launch"] = round(tb / tot)
out["algorithmic_bytes_per_productive_launch"] = round(ab / tot)
out["traffic_over_algorithmic"] = round(tb / ab, 3)
out["note"] = ("Round 4: the blocks of A' hold no values of their own -- both products read ONE copy (the row-group array of A: 8 B "
               "value + 4 B packed index per entry); A' adds 3 B per entry of index planes and 16 B of segment descriptors per 64 "
               "entries.  The counters sit between the L2s and the fabric: reads served by the 256 MB Infinity Cache count here like "
               "reads from HBM, so a value A' re-reads right after the A product still shows (the algorithmic count prices 12 B/nnz "
               "for each product).")
print(json.dumps(out, indent=1))
