# developer script (GPU box): several iterations per launch against one -- tests, then an interleaved A/B of the headline bench
mkdir -p gpurun_out/r5
timeout -k 10 600 python -m pytest "tests/test_gpu_parity.py::test_several_iterations_per_launch_are_bitwise_one_iteration_per_launch" -q -m gpu -x > gpurun_out/r5/multi.log 2>&1; rc=$?; echo "multi rc=$rc"; tail -3 gpurun_out/r5/multi.log | cut -c1-200
B="timeout -k 10 200 python bench.py --cpu-evals 0"
for r in 1 2; do for k in 1 8 4 2; do FPSQ_MULTI_ITER=$k $B > gpurun_out/r5/ab_k${k}_$r.json 2> gpurun_out/r5/ab_k${k}_$r.err; echo "k=$k run $r rc=$?"; done; done
python - <<'PY'
import json
print("FPSQ_MULTI_ITER  evals/s  ms/eval  launches/eval  avg product-launch us  frac")
for r in (1, 2):
    for k in (1, 8, 4, 2):
        d=json.load(open(f"gpurun_out/r5/ab_k{k}_{r}.json")); ro=d["roofline"]
        print(f"{k:>3} (run {r})  {d['value']:8.1f}  {d['ms_per_step']:.4f}  {ro['launches_per_eval']:5.1f}  {ro['avg_launch_us']:8.2f}  {ro['frac']:.4f}")
PY
