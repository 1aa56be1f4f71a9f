#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_dense.py -m gpu -x -q -o faulthandler_timeout=100 > $O/r2_t7.log 2>&1; rc=$?; echo "dense tests rc=$rc"; tail -5 $O/r2_t7.log
[ $rc -ne 0 ] && exit 1
for G in 4 3; do
FPSQ_DENSE_SPLITK=1 FPSQ_DENSE_POTRF=$G timeout -k 10 120 python bench.py --workload "dense-block n=4096 m=2048" --steps 10 --warmup 2 --cpu-evals 0 --repeats 3 > $O/r2_b7_dense_$G.json 2> $O/r2_b7.err; echo "dense gen=$G rc=$?"; python3 -c "import json;d=json.load(open('$O/r2_b7_dense_$G.json'));print(d['value'], d['roofline']['achieved'], d['roofline']['device_ms'])"
done
rm -rf $O/dprof; FPSQ_DENSE_SPLITK=1 timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dprof -o d -- python3 bench.py --workload "dense-block n=4096 m=2048" --steps 5 --warmup 1 --cpu-evals 0 --repeats 1 > $O/dprof.log 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open("$O/dprof/d_kernel_stats.csv")):
    print(r["Name"][:70].ljust(70), r["Calls"], round(float(r["AverageNs"])/1e3,2), "us avg", r["Percentage"])
PY
