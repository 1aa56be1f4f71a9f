export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "spmv or column_sorted or riding or one_pass or late_leader or headline_full_size_matches or speculative" > gpurun_out/r4_c10.log 2>&1; rc=$?; echo "test rc=$rc"; tail -12 gpurun_out/r4_c10.log | cut -c1-200
[ $rc -ne 0 ] && exit 1
for sh in 1 0; do
FPSQ_AT_SHARED=$sh timeout -k 10 300 python bench.py --cpu-evals 0 > gpurun_out/r4_c10_sh$sh.json 2> /dev/null; echo "shared=$sh: $(cut -c1-140 gpurun_out/r4_c10_sh$sh.json)"
done
FPSQ_AT_SHARED=1 timeout -k 10 300 python bench.py --cpu-evals 0 --pointers device+jac > gpurun_out/r4_c10_jac.json 2> /dev/null; echo "shared device+jac: $(cut -c1-140 gpurun_out/r4_c10_jac.json)"
rm -rf gpurun_out/r4_c10_ks; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_c10_ks -o k -- python3 bench.py --pointers device+jac --steps 10 --warmup 2 --cpu-evals 0 --repeats 2 --no-roofline-pass > gpurun_out/r4_c10_ks.log 2>&1
find gpurun_out/r4_c10_ks -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r4_c10_ks/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage'])>0.5: print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,2))
PY
