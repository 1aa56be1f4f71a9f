# developer script (GPU box): soaks of the round-5 sharded path (in-launch sums + one launch per iteration with the halo inside)
mkdir -p gpurun_out/r5
timeout -k 10 500 python tools/lx_soak_mp.py 1000 2 0 200000 > gpurun_out/r5/lx_soak_mp2.txt 2>&1; echo "mp soak 2 ranks rc=$?"; tail -3 gpurun_out/r5/lx_soak_mp2.txt
FPSQ_DEBUG_XCH_DELAY=2 FPSQ_DEBUG_XCH_LONG_DELAY_MS=300 timeout -k 10 500 python tools/lx_soak_mp.py 200 2 0 200000 > gpurun_out/r5/lx_soak_mp2_late.txt 2>&1; echo "mp soak 2 ranks, rank 1 300 ms late every 512th exchange rc=$?"; tail -3 gpurun_out/r5/lx_soak_mp2_late.txt
FPSQ_DEBUG_XCH_DELAY=1 FPSQ_DEBUG_XCH_LONG_DELAY_MS=300 timeout -k 10 500 python tools/lx_soak_mp.py 200 3 0 24000 > gpurun_out/r5/lx_soak_mp3_late.txt 2>&1; echo "mp soak 3 ranks small, rank 0 late rc=$?"; tail -3 gpurun_out/r5/lx_soak_mp3_late.txt
timeout -k 10 300 python tools/lx_soak.py 2000 2 0 > gpurun_out/r5/lx_soak_2.txt 2>&1; echo "soak 2 shards rc=$?"; tail -2 gpurun_out/r5/lx_soak_2.txt
bash tools/r5_ab_r4.sh
