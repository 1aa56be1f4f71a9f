#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_dense.py -m gpu -x -v -s -o faulthandler_timeout=100 > $O/r2_t5.log 2>&1; rc=$?; echo "dense tests rc=$rc"; tail -25 $O/r2_t5.log
[ $rc -ne 0 ] && exit 1
for S in 1 8 12 15 16; do
FPSQ_DENSE_SPLITK=$S timeout -k 10 120 python bench.py --workload "dense-block n=4096 m=2048" --steps 10 --warmup 2 --cpu-evals 0 --repeats 3 > $O/r2_b5_dense_$S.json 2> $O/r2_b5.err; echo "dense S=$S rc=$?"; python3 -c "import json;d=json.load(open('$O/r2_b5_dense_$S.json'));print(d['value'], d['roofline']['achieved'], d['roofline']['device_ms'])"
done
