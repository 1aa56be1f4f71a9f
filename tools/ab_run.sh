#!/bin/bash
# developer A/B: bench with every library variant under tools/ab/  (usage: tools/ab_run.sh [bench args])
for f in tools/ab/*.so fletcherpenaltysolver.jl_amd/lib/libfpsq.so; do
  r=$(FPSQ_LIB_PATH=$PWD/$f timeout -k 10 200 python bench.py --steps 20 --warmup 3 --cpu-evals 0 "$@" 2>&1 | tail -1)
  echo "$f $(echo "$r" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['config']['iters_lsqr_craig_median'], d['config'].get('ride_fallbacks'))")"
done
