#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 200 python bench.py --force-shard --cpu-evals 0 --steps 10 > $O/r2_b8_shard1.json 2> $O/r2_b8_shard1.err; echo "force-shard rc=$?"; cut -c1-400 $O/r2_b8_shard1.json
FPSQ_BENCH_WATCHDOG=60 FPSQ_BENCH_REHEARSE=1 timeout -k 10 200 python bench.py --gpus 2 --steps 5 --cpu-evals 0 > $O/r2_b8_reh_shard.json 2> $O/r2_b8_reh_shard.err; echo "rehearse shard (expect fallback) rc=$?"; cat $O/r2_b8_reh_shard.json; tail -3 $O/r2_b8_reh_shard.err
FPSQ_BENCH_REHEARSE=1 timeout -k 10 200 python bench.py --gpus 2 --parallel replicas --steps 5 --cpu-evals 0 > $O/r2_b8_reh_rep.json 2> $O/r2_b8_reh_rep.err; echo "rehearse replicas rc=$?"; cut -c1-300 $O/r2_b8_reh_rep.json
