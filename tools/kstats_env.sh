#!/bin/bash
# developer script (GPU box): per-kernel average durations of the headline bench under each value of one environment
# variable.  usage: tools/kstats_env.sh NAME v1 v2 ...   (FPSQ_LIB_PATH selects another build of the library)
export TMPDIR=/tmp
name=$1; shift
for w in "$@"; do
  d=gpurun_out/ks_${name}_$w
  rm -rf $d
  export $name=$w
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o k -- python3 bench.py --steps 20 --warmup 3 --cpu-evals 0 --repeats 1 --no-roofline-pass > $d.log 2>&1 || exit 1
  rm -f $d/k_kernel_trace.csv
  python3 - <<PY
import csv
print("$name=$w")
for r in csv.reader(open("$d/k_kernel_stats.csv")):
    if 'k_spmv' in r[0] or 'k_step' in r[0]: print("  %-62s calls %5s avg %9.1f ns min %s max %s" % (r[0][:62], r[1], float(r[3]), r[5], r[6]))
PY
done
