mkdir -p gpurun_out
timeout -k 10 300 python tools/sumorder_probe.py > gpurun_out/r4_c3_sumorder.log 2>&1; echo rc=$?; cat gpurun_out/r4_c3_sumorder.log | tail -20
for r in p2p rccl; do timeout -k 10 200 python bench.py --force-shard --comm-route $r --cpu-evals 0 > gpurun_out/r4_c3_shard1_$r.json 2> gpurun_out/r4_c3_shard1_$r.err; echo "shard1 $r rc=$?"; cut -c1-200 gpurun_out/r4_c3_shard1_$r.json; done
