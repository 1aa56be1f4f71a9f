#!/bin/bash
# developer tool: average duration of the kernels matching $1 under rocprofv3, for every library build given after it
# usage: tools/kernel_avg.sh k_step tools/ab/a.so tools/ab/b.so
export TMPDIR=/tmp
pat=$1; shift
for lib in "$@"; do
  d=gpurun_out/kavg_$(basename $lib .so)
  FPSQ_LIB_PATH=$PWD/$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o k -- python3 bench.py --steps 10 --warmup 2 --cpu-evals 0 > $d.log 2>&1
  python3 - "$pat" "$d/k_kernel_stats.csv" "$lib" <<'PY'
import csv, sys
pat, path, lib = sys.argv[1:4]
for r in csv.DictReader(open(path)):
    if pat in r["Name"]:
        print(f"{lib}: {r['Name'][:40]} calls {r['Calls']} avg {float(r['AverageNs'])/1e3:.2f} us")
PY
done
