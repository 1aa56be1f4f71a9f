export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/r4_c9_ks; FPSQ_REFRESH_SPLIT=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4_c9_ks -o k -- python3 bench.py --pointers device+jac --steps 4 --warmup 1 --cpu-evals 0 --repeats 1 --no-roofline-pass > gpurun_out/r4_c9_ks.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r4_c9_ks/**/*kernel_trace.csv',recursive=True)[0]
d=[ (int(r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X',0)), (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in csv.DictReader(open(f)) if 'k_refresh' in r['Kernel_Name']]
from collections import defaultdict
g=defaultdict(list)
for a,b in d: g[a].append(b)
for a,b in g.items(): print(a, len(b), sum(b)/len(b))
PY
find gpurun_out/r4_c9_ks -name "*kernel_trace.csv" -delete
