#!/bin/bash
# developer script (GPU box): tests + bench + kernel-trace timeline
export TMPDIR=/tmp
O=gpurun_out
T=${1:-full}
if [ "$T" = full ]; then
  timeout -k 10 300 python -m pytest tests -m gpu -x -v -o faulthandler_timeout=120 > $O/r2_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/r2_tests.log
  [ $rc -ne 0 ] && exit 1
fi
timeout -k 10 200 python bench.py > $O/r2_b2.json 2> $O/r2_b2.err; echo "default rc=$?"; cat $O/r2_b2.json
timeout -k 10 200 python bench.py --op hprod --cpu-evals 0 > $O/r2_b2_hprod.json 2> $O/r2_b2_hprod.err; echo "hprod rc=$?"; cat $O/r2_b2_hprod.json
rm -rf $O/tl; timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/tl -o t -- python3 bench.py --steps 10 --warmup 2 --cpu-evals 0 --no-roofline-pass --repeats 1 > $O/tl.log 2>&1; echo "trace rc=$?"
python3 tools/timeline.py $O/tl/t_kernel_trace.csv > $O/r2_timeline.txt 2>&1; cat $O/r2_timeline.txt
