# developer script (GPU box): this tree's library against round 4's final one (946e069, built into gpurun_ab_libfpsq_r4.so),
# interleaved on one box
mkdir -p gpurun_out/r5
B="timeout -k 10 200 python bench.py --cpu-evals 0 --workload"
W="pde-control-like n=1e6 m=1e5 nnz=1e7"
for r in 1 2 3; do
  FPSQ_LIB_PATH=$PWD/gpurun_ab_libfpsq_r4.so $B "$W" > gpurun_out/r5/abr4_old_$r.json 2> gpurun_out/r5/abr4_old_$r.err; echo "old $r rc=$?"
  $B "$W" > gpurun_out/r5/abr4_new_$r.json 2> gpurun_out/r5/abr4_new_$r.err; echo "new $r rc=$?"
done
python - <<'PY'
import json
for r in (1, 2, 3):
    for k in ("old", "new"):
        try:
            d = json.load(open(f"gpurun_out/r5/abr4_{k}_{r}.json")); ro = d["roofline"]
            print(k, r, d["value"], d["ms_per_step"], ro["avg_launch_us"], ro["frac"])
        except Exception as e:
            print(k, r, "ERR", e)
PY
