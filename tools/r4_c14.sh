mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "speculative or run_ahead or repeat or riding" > gpurun_out/r4_c14.log 2>&1; rc=$?; echo "test rc=$rc"; tail -4 gpurun_out/r4_c14.log | cut -c1-200
for a in "" "--delta 1.4901161193847656e-08" "--alternate-delta"; do
timeout -k 10 300 python bench.py --cpu-evals 0 --no-roofline-pass $a > gpurun_out/r4_c14.json 2> /dev/null; echo "[$a]: $(python3 -c "import json;d=json.load(open('gpurun_out/r4_c14.json'));print(d['value'],d['ms_per_step'],d['config']['iters_lsqr_craig_median'],d['config'].get('iters_lsqr_craig_seen'))")"
done
