mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "headline_full_size_matches_oracle" > gpurun_out/r4_c7.log 2>&1; echo rc=$?; tail -5 gpurun_out/r4_c7.log | cut -c1-200
timeout -k 10 300 python bench.py --workload "pde-control-hashed n=1e6 m=1e5 nnz=1e7" --cpu-evals 2 > gpurun_out/r4_c7_hashed.json 2> gpurun_out/r4_c7_hashed.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r4_c7_hashed.json'))
print(d['value'], d['ms_per_step'], d['config']['iters_lsqr_craig_median'], d['roofline']['frac'], d['roofline']['avg_productive_launch_us'], d.get('cpu_baseline',{}).get('value'))
PY
