#!/bin/bash
# developer script (GPU box): bench variants + kernel-trace timeline
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out
python bench.py > $O/r2_b1.json 2> $O/r2_b1.err; echo "default rc=$?"; cat $O/r2_b1.json
python bench.py --force-shard --cpu-evals 0 > $O/r2_b_shard1.json 2> $O/r2_b_shard1.err; echo "force-shard rc=$?"; cat $O/r2_b_shard1.json
FPSQ_BENCH_REHEARSE=1 python bench.py --gpus 2 --parallel replicas --steps 10 --cpu-evals 0 > $O/r2_b_reh2.json 2> $O/r2_b_reh2.err; echo "rehearse rc=$?"; cat $O/r2_b_reh2.json
python bench.py --workload "dense-block n=4096 m=2048" --steps 5 --warmup 1 > $O/r2_b_dense.json 2> $O/r2_b_dense.err; echo "dense rc=$?"; cat $O/r2_b_dense.json
python bench.py --pointers host --cpu-evals 0 > $O/r2_b_host.json 2> $O/r2_b_host.err; echo "host rc=$?"; cat $O/r2_b_host.json
python bench.py --pointers host+jac --cpu-evals 0 > $O/r2_b_hostjac.json 2> $O/r2_b_hostjac.err; echo "host+jac rc=$?"; cat $O/r2_b_hostjac.json
python bench.py --workload "random-eqqp n=1e5 m=1e4 nnz=1e6" --cpu-evals 0 > $O/r2_b_cfg2.json 2> $O/r2_b_cfg2.err; echo "cfg2 rc=$?"; cat $O/r2_b_cfg2.json
rm -rf $O/tl; rocprofv3 --kernel-trace --output-format csv -d $O/tl -o t -- python3 bench.py --steps 10 --warmup 2 --cpu-evals 0 --no-roofline-pass --repeats 1 > $O/tl.log 2>&1; echo "trace rc=$?"
python3 tools/timeline.py $O/tl/t_kernel_trace.csv > $O/r2_timeline.txt 2>&1; cat $O/r2_timeline.txt
