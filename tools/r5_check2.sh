mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests/test_gpu_p2p_ipc.py tests/test_gpu_bench.py "tests/test_gpu_parity.py::test_halo_sharded_objgrad_hprod_match_single_gpu" "tests/test_gpu_parity.py::test_rccl_single_rank_communicator" tests/test_gpu_fps_solve.py -q -m gpu -x > gpurun_out/r5/check2.log 2>&1; rc=$?; echo "check2 rc=$rc"; tail -8 gpurun_out/r5/check2.log | cut -c1-300
B="timeout -k 10 200 python bench.py --cpu-evals 0"
$B > gpurun_out/r5/b_single.json 2> gpurun_out/r5/b_single.err; echo "single rc=$?"
$B --force-shard > gpurun_out/r5/b_fs_p2p.json 2> gpurun_out/r5/b_fs_p2p.err; echo "fs p2p rc=$?"
$B --force-shard --comm-route rccl > gpurun_out/r5/b_fs_rccl.json 2> gpurun_out/r5/b_fs_rccl.err; echo "fs rccl rc=$?"
$B > gpurun_out/r5/b_single2.json 2> gpurun_out/r5/b_single2.err; echo "single2 rc=$?"
python - <<'PY'
import json
for f in ("b_single","b_fs_p2p","b_fs_rccl","b_single2"):
    try:
        d=json.load(open(f"gpurun_out/r5/{f}.json"))
        c=d["config"]
        print(f, d["value"], d["ms_per_step"], c["iters_lsqr_craig_median"], c.get("loop_launches_per_iteration"), c.get("comm_route"), c.get("comm_in_launch_sums"), c["fuse_fallbacks"], c["wait_timeouts"], d["roofline"] and d["roofline"]["frac"])
    except Exception as e:
        print(f, "ERR", e)
PY
