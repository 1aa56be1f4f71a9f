export TMPDIR=/tmp
mkdir -p gpurun_out
for sh in 1 0; do
rm -rf gpurun_out/r4_c12_ks$sh; FPSQ_AT_SHARED=$sh timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_c12_ks$sh -o k -- python3 bench.py --steps 10 --warmup 2 --cpu-evals 0 --repeats 2 --no-roofline-pass > gpurun_out/r4_c12_ks$sh.log 2>&1
find gpurun_out/r4_c12_ks$sh -name "*kernel_trace.csv" -delete
echo "== shared=$sh"
python3 - $sh <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/r4_c12_ks%s/**/*kernel_stats.csv'%sys.argv[1],recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage'])>0.5: print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,2))
PY
done
