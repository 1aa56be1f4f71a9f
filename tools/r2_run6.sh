#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
for S in 1 15; do
rm -rf $O/dprof_$S; FPSQ_DENSE_SPLITK=$S timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dprof_$S -o d -- python3 bench.py --workload "dense-block n=4096 m=2048" --steps 5 --warmup 1 --cpu-evals 0 --repeats 1 > $O/dprof_$S.log 2>&1; echo "S=$S rc=$?"
python3 - <<PY
import csv
for r in csv.DictReader(open("$O/dprof_$S/d_kernel_stats.csv")):
    print(r["Name"][:70].ljust(70), r["Calls"], round(float(r["AverageNs"])/1e3,2), "us avg", r["Percentage"])
PY
done
