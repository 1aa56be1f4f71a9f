#!/bin/bash
# developer A/B over FPSQ_RIDE_STEPS settings of the in-tree library
for v in 0 1 2 0 1 2; do
  r=$(FPSQ_RIDE_STEPS=$v timeout -k 10 200 python bench.py --steps 20 --warmup 3 --cpu-evals 0 "$@" 2>&1 | tail -1)
  echo "ride=$v $(echo "$r" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['config']['iters_lsqr_craig_median'], d['config'].get('ride_fallbacks'))")"
done
