# developer script (GPU box): cfg2 (n = 1e5, m = 1e4, random columns) -- two launches per iteration (default) / one / several iterations per launch
mkdir -p gpurun_out/r5
B="timeout -k 10 200 python bench.py --cpu-evals 0 --workload"
W="random-eqqp n=1e5 m=1e4 nnz=1e6"
for r in 1 2; do
  $B "$W" > gpurun_out/r5/c2_def_$r.json 2> gpurun_out/r5/c2_def_$r.err; echo "default $r rc=$?"
  FPSQ_FUSE_ITER=2 $B "$W" > gpurun_out/r5/c2_f_$r.json 2> gpurun_out/r5/c2_f_$r.err; echo "fused $r rc=$?"
  FPSQ_FUSE_ITER=2 FPSQ_MULTI_ITER=8 $B "$W" > gpurun_out/r5/c2_m8_$r.json 2> gpurun_out/r5/c2_m8_$r.err; echo "multi 8 $r rc=$?"
  FPSQ_FUSE_ITER=2 FPSQ_MULTI_ITER=16 $B "$W" > gpurun_out/r5/c2_m16_$r.json 2> gpurun_out/r5/c2_m16_$r.err; echo "multi 16 $r rc=$?"
done
python - <<'PY'
import json
for r in (1, 2):
    for c in ("def", "f", "m8", "m16"):
        try:
            d = json.load(open(f"gpurun_out/r5/c2_{c}_{r}.json")); cf = d["config"]
            print(c, r, d["value"], d["ms_per_step"], cf.get("loop_launches_per_iteration"), cf.get("fuse_fallbacks"), cf.get("wait_timeouts"))
        except Exception as e:
            print(c, r, "ERR", e)
PY
