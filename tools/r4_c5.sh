mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "recurrences_agree" > gpurun_out/r4_c5.log 2>&1; echo rc=$?; tail -40 gpurun_out/r4_c5.log | cut -c1-220
