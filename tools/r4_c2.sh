set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_p2p_ipc.py -q -m gpu -x > gpurun_out/r4_c2_p2p.log 2>&1; rc=$?; echo "p2p rc=$rc"
tail -30 gpurun_out/r4_c2_p2p.log
[ $rc -eq 0 ] && { timeout -k 10 900 python -m pytest tests/test_gpu_bench.py tests/test_gpu_parity.py -q -m gpu -x -k "shard or halo or rccl or bench" > gpurun_out/r4_c2_shard.log 2>&1; echo "shard rc=$?"; tail -8 gpurun_out/r4_c2_shard.log; }
