mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests/test_gpu_fps_solve.py tests/test_gpu_dense.py tests/test_gpu_p2p_ipc.py tests/test_gpu_bench.py -q -m gpu -x > gpurun_out/r5/check1.log 2>&1; rc=$?; echo "check1 rc=$rc"; tail -8 gpurun_out/r5/check1.log | cut -c1-300
timeout -k 10 200 python bench.py --cpu-evals 0 > gpurun_out/r5/bench_head0.json 2> gpurun_out/r5/bench_head0.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r5/bench_head0.json
