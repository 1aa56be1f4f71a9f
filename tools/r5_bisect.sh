# developer script (GPU box): where did k_iter_fused get slower?  in-run product-launch averages of libraries built at this round's commits
mkdir -p gpurun_out/r5
B="timeout -k 10 200 python bench.py --cpu-evals 0 --repeats 20 --workload"
W="pde-control-like n=1e6 m=1e5 nnz=1e7"
for r in 1 2; do
  for c in r4 cur; do
    if [ "$c" = "cur" ]; then unset FPSQ_LIB_PATH; elif [ "$c" = "r4" ]; then export FPSQ_LIB_PATH=$PWD/gpurun_ab_libfpsq_r4.so; else export FPSQ_LIB_PATH=$PWD/gpurun_ab_lib_$c.so; fi
    $B "$W" > gpurun_out/r5/bis_${c}_$r.json 2> gpurun_out/r5/bis_${c}_$r.err; echo "$c $r rc=$?"
  done
done
python - <<'PY'
import json
for c in ("r4", "cur"):
    out = []
    for r in (1, 2):
        try:
            d = json.load(open(f"gpurun_out/r5/bis_{c}_{r}.json")); out.append((d["value"], d["roofline"]["avg_launch_us"]))
        except Exception as e:
            out.append(("ERR", str(e)[:40]))
    print(c, out)
PY
