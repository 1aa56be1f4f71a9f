# developer script (GPU box): phi by the FIRST workgroup of the tail's product (this tree) against the previous build (phi by the last: gpurun_ab_lib_philast.so)
mkdir -p gpurun_out/r5
timeout -k 10 400 python tools/tail_check.py 16 > gpurun_out/r5/tail_check2.txt 2>&1; echo "check rc=$?"; grep -v amdgpu gpurun_out/r5/tail_check2.txt | tail -5
B="timeout -k 10 200 python bench.py --cpu-evals 0"
for r in 1 2 3; do
  FPSQ_LIB_PATH=$PWD/gpurun_ab_lib_philast.so $B > gpurun_out/r5/phi_last_$r.json 2> gpurun_out/r5/phi_last_$r.err; echo "phi last ($r) rc=$?"
  $B > gpurun_out/r5/phi_first_$r.json 2> gpurun_out/r5/phi_first_$r.err; echo "phi first ($r) rc=$?"
done
python - <<'PY'
import json
for r in (1, 2, 3):
    for c in ("last", "first"):
        try:
            d = json.load(open(f"gpurun_out/r5/phi_{c}_{r}.json")); ro = d["roofline"]
            print("phi", c, r, d["value"], d["ms_per_step"], ro["avg_launch_us"], ro["frac"])
        except Exception as e:
            print(c, r, "ERR", e)
PY
