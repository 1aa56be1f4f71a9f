# developer script (GPU box): soaks of the round-5 sharded path (sums over the ranks inside the launches + one launch per iteration with the halo inside)
mkdir -p gpurun_out/r5
timeout -k 10 500 python tools/lx_soak_mp.py 2000 2 0 200000 > gpurun_out/r5/lx_soak_mp2.txt 2>&1; echo "mp soak 2 ranks rc=$?"; tail -1 gpurun_out/r5/lx_soak_mp2.txt | cut -c1-400
for k in 1 2 3 4; do
  LX_SOAK_KEEP=1 FPSQ_P2P_POLLS=3000000 timeout -k 10 300 python tools/lx_soak_mp.py 2000 3 0 100000 > gpurun_out/r5/lx_soak_mp3_$k.txt 2>&1; echo "mp soak 3 ranks n=100000 ($k) rc=$?"; tail -1 gpurun_out/r5/lx_soak_mp3_$k.txt | cut -c1-400
done
LX_SOAK_CHUNKS=1 FPSQ_P2P_POLLS=3000000 timeout -k 10 300 python tools/lx_soak_mp.py 1000 3 1.4901161193847656e-08 100000 > gpurun_out/r5/lx_soak_mp3_d.txt 2>&1; echo "mp soak 3 ranks delta, every row rc=$?"; tail -2 gpurun_out/r5/lx_soak_mp3_d.txt | cut -c1-400
FPSQ_DEBUG_XCH_DELAY=2 FPSQ_DEBUG_XCH_LONG_DELAY_MS=300 timeout -k 10 500 python tools/lx_soak_mp.py 200 2 0 200000 > gpurun_out/r5/lx_soak_mp2_late.txt 2>&1; echo "mp soak 2 ranks, rank 1 300 ms late every 128th exchange rc=$?"; tail -1 gpurun_out/r5/lx_soak_mp2_late.txt | cut -c1-400
timeout -k 10 300 python tools/lx_soak.py 2000 2 0 > gpurun_out/r5/lx_soak_2.txt 2>&1; echo "in-process soak 2 shards rc=$?"; tail -1 gpurun_out/r5/lx_soak_2.txt | cut -c1-400
