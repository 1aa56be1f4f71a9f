export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_p2p_ipc.py tests/test_gpu_parity.py -q -m gpu -x -k "p2p or halo or peer or missing" > gpurun_out/r4_c4_p2p.log 2>&1; rc=$?; echo "p2p rc=$rc"; tail -5 gpurun_out/r4_c4_p2p.log
[ $rc -ne 0 ] && exit 1
for r in p2p rccl; do timeout -k 10 200 python bench.py --force-shard --comm-route $r --cpu-evals 0 > gpurun_out/r4_c4_shard1_$r.json 2> gpurun_out/r4_c4_shard1_$r.err; echo "shard1 $r rc=$?"; cut -c1-200 gpurun_out/r4_c4_shard1_$r.json; done
rm -rf gpurun_out/r4_c4_ks; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_c4_ks -o k -- python3 bench.py --force-shard --comm-route p2p --steps 10 --warmup 2 --cpu-evals 0 --repeats 2 --no-roofline-pass > gpurun_out/r4_c4_ks.log 2>&1; echo "ks rc=$?"
rm -f gpurun_out/r4_c4_ks/*/k_kernel_trace.csv gpurun_out/r4_c4_ks/k_kernel_trace.csv
find gpurun_out/r4_c4_ks -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {} | cut -c1-160'
