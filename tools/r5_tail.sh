# developer script (GPU box): the one-launch tail (FPSQ_FUSE_TAIL=1) against the two launches (=0): bitwise check, then interleaved A/B
mkdir -p gpurun_out/r5
timeout -k 10 400 python tools/tail_check.py 24 > gpurun_out/r5/tail_check.txt 2>&1; echo "check rc=$?"; grep -v amdgpu gpurun_out/r5/tail_check.txt | tail -7
B="timeout -k 10 200 python bench.py --cpu-evals 0"
for r in 1 2 3; do
  for c in 0 1; do
    FPSQ_FUSE_TAIL=$c $B > gpurun_out/r5/tail_${c}_$r.json 2> gpurun_out/r5/tail_${c}_$r.err; echo "fuse tail $c ($r) rc=$?"
  done
done
python - <<'PY'
import json
for r in (1, 2, 3):
    for c in (0, 1):
        try:
            d = json.load(open(f"gpurun_out/r5/tail_{c}_{r}.json")); ro = d["roofline"]
            print("fuse tail", c, r, d["value"], d["ms_per_step"], ro["avg_launch_us"], ro["frac"])
        except Exception as e:
            print(c, r, "ERR", e)
PY
