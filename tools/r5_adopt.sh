# developer script (GPU box): the library enqueueing on the caller's stream (FPSQ_ADOPT_STREAM=1) against its own stream + events (=0)
mkdir -p gpurun_out/r5
B="timeout -k 10 200 python bench.py --cpu-evals 0"
for r in 1 2 3; do
  for c in 0 1; do
    FPSQ_ADOPT_STREAM=$c $B > gpurun_out/r5/adopt_${c}_$r.json 2> gpurun_out/r5/adopt_${c}_$r.err; echo "adopt $c ($r) rc=$?"
  done
done
python - <<'PY'
import json
for r in (1, 2, 3):
    for c in (0, 1):
        try:
            d = json.load(open(f"gpurun_out/r5/adopt_{c}_{r}.json")); ro = d["roofline"]
            print("adopt", c, r, d["value"], d["ms_per_step"], ro["avg_launch_us"], ro["frac"])
        except Exception as e:
            print(c, r, "ERR", e, open(f"gpurun_out/r5/adopt_{c}_{r}.err").read()[-400:])
PY
