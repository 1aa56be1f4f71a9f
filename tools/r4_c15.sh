mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_dense.py tests/test_gpu_fps_solve.py -q -m gpu -x > gpurun_out/r4_c15.log 2>&1; rc=$?; echo "test rc=$rc"; tail -5 gpurun_out/r4_c15.log | cut -c1-200
timeout -k 10 200 python bench.py --workload "dense-block n=4096 m=2048" --steps 10 --warmup 2 --cpu-evals 0 > gpurun_out/r4_c15_dense.json 2>/dev/null; python3 -c "import json;d=json.load(open('gpurun_out/r4_c15_dense.json'));print(d['value'],d['ms_per_step'],d['roofline']['device_ms'],d['roofline']['frac'])"
