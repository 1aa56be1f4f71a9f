mkdir -p gpurun_out/r5
timeout -k 10 600 python -m pytest "tests/test_gpu_parity.py::test_several_iterations_per_launch_are_bitwise_one_iteration_per_launch" -q -m gpu -x > gpurun_out/r5/multi.log 2>&1; rc=$?; echo "multi rc=$rc"; tail -5 gpurun_out/r5/multi.log | cut -c1-300
if [ $rc -eq 0 ]; then
B="timeout -k 10 200 python bench.py --cpu-evals 0"
FPSQ_MULTI_ITER=1 $B > gpurun_out/r5/n_k1.json 2> gpurun_out/r5/n_k1.err; echo "k1 rc=$?"
FPSQ_MULTI_ITER=8 $B > gpurun_out/r5/n_k8.json 2> gpurun_out/r5/n_k8.err; echo "k8 rc=$?"
FPSQ_MULTI_ITER=8 FPSQ_MULTI_UPD=32,96 $B > gpurun_out/r5/n_k8s.json 2> gpurun_out/r5/n_k8s.err; echo "k8s rc=$?"
FPSQ_MULTI_ITER=8 FPSQ_MULTI_UPD=128,384 $B > gpurun_out/r5/n_k8m.json 2> gpurun_out/r5/n_k8m.err; echo "k8m rc=$?"
FPSQ_MULTI_ITER=8 FPSQ_MULTI_UPD=1024,2048 $B > gpurun_out/r5/n_k8l.json 2> gpurun_out/r5/n_k8l.err; echo "k8l rc=$?"
FPSQ_MULTI_ITER=4 $B > gpurun_out/r5/n_k4.json 2> gpurun_out/r5/n_k4.err; echo "k4 rc=$?"
FPSQ_MULTI_ITER=1 $B > gpurun_out/r5/n_k1b.json 2> gpurun_out/r5/n_k1b.err; echo "k1b rc=$?"
python - <<'PY'
import json
for f in ("n_k1","n_k8","n_k8s","n_k8m","n_k8l","n_k4","n_k1b"):
    try:
        d=json.load(open(f"gpurun_out/r5/{f}.json")); c=d["config"]; r=d["roofline"]
        print(f, d["value"], d["ms_per_step"], c["iters_lsqr_craig_median"], c["loop_launches_per_iteration"], c["fuse_fallbacks"], c["wait_timeouts"], r["frac"], r["launches_per_eval"], r["avg_launch_us"])
    except Exception as e:
        print(f, "ERR", e)
PY
fi
