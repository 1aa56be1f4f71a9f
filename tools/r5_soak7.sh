# developer script (GPU box): the single-GPU soaks of rounds 3-4 on round 5's final code
mkdir -p gpurun_out/r5
timeout -k 10 400 python tools/fuse_soak_two.py 300 > gpurun_out/r5/s7_two.txt 2>&1; echo "two handles at once rc=$?"; tail -1 gpurun_out/r5/s7_two.txt | cut -c1-400
timeout -k 10 300 python tools/ride_soak.py 600 > gpurun_out/r5/s7_ride.txt 2>&1; echo "riding steps rc=$?"; tail -1 gpurun_out/r5/s7_ride.txt | cut -c1-400
timeout -k 10 300 python tools/chain_soak.py 400 > gpurun_out/r5/s7_chain.txt 2>&1; echo "chained sweeps rc=$?"; tail -1 gpurun_out/r5/s7_chain.txt | cut -c1-400
timeout -k 10 300 python tools/fuse_soak.py 600 1000000 100000 1.4901161193847656e-08 stratified 3 > gpurun_out/r5/s7_fuse_rot.txt 2>&1; echo "one launch, every hand-over across XCDs, delta rc=$?"; tail -1 gpurun_out/r5/s7_fuse_rot.txt | cut -c1-400
