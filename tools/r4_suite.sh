mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/r4_suite.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -6 gpurun_out/r4_suite.log | cut -c1-200
