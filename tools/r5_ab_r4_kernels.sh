# developer script (GPU box): per-kernel average durations (rocprofv3 --kernel-trace --stats), this tree's library against round 4's
export TMPDIR=/tmp
O=gpurun_out/r5/abk
rm -rf $O; mkdir -p $O
W="pde-control-like n=1e6 m=1e5 nnz=1e7"
for k in old new old2 new2; do
  if [ "${k:0:3}" = "old" ]; then export FPSQ_LIB_PATH=$PWD/gpurun_ab_libfpsq_r4.so; else unset FPSQ_LIB_PATH; fi
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$k -o k -- python3 bench.py --steps 20 --warmup 3 --cpu-evals 0 --repeats 2 --no-roofline-pass --workload "$W" > $O/$k.log 2>&1; echo "$k rc=$?"
  python3 - "$O/$k/k_kernel_stats.csv" "$k" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    for key in ("k_iter_fused", "k_spmv_rgcs", "k_spmv<", "k_startup", "k_qp_penalty_grad", "k_ys", "k_step"):
        if key in n:
            print(sys.argv[2], key.ljust(18), "calls", r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 2))
PY
  find $O/$k -name "*kernel_trace.csv" -delete
done
