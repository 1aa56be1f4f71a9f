mkdir -p gpurun_out/r5
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/r5/suite.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -8 gpurun_out/r5/suite.log | cut -c1-300
