# developer script (GPU box): hprod! Val(2) with the rows of A'(A v) going straight into Hv (FPSQ_FUSE_TAIL=1) against product + k_qp_hprod_fin (=0)
mkdir -p gpurun_out/r5
timeout -k 10 400 python tools/tail_check.py 16 > gpurun_out/r5/tail_check3.txt 2>&1; echo "check rc=$?"; grep -v amdgpu gpurun_out/r5/tail_check3.txt | tail -5
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_launch_tail or hprod" 2>&1 | tail -2
B="timeout -k 10 200 python bench.py --cpu-evals 0 --op hprod"
for r in 1 2 3; do
  for c in 0 1; do
    FPSQ_FUSE_TAIL=$c $B > gpurun_out/r5/htail_${c}_$r.json 2> gpurun_out/r5/htail_${c}_$r.err; echo "hprod fuse tail $c ($r) rc=$?"
  done
done
python - <<'PY'
import json
for r in (1, 2, 3):
    for c in (0, 1):
        try:
            d = json.load(open(f"gpurun_out/r5/htail_{c}_{r}.json"))
            print("hprod fuse tail", c, r, d["value"], d["ms_per_step"])
        except Exception as e:
            print(c, r, "ERR", e)
PY
