# developer script (GPU box): long soaks of the final code
mkdir -p gpurun_out/r5
FPSQ_P2P_POLLS=3000000 timeout -k 10 500 python tools/lx_soak_mp.py 20000 3 0 100000 > gpurun_out/r5/long_mp3.txt 2>&1; echo "3 ranks 20000 rc=$?"; tail -1 gpurun_out/r5/long_mp3.txt | cut -c1-420
LX_SOAK_POINTS=near FPSQ_P2P_POLLS=3000000 timeout -k 10 500 python tools/lx_soak_mp.py 10000 4 1.4901161193847656e-08 100000 > gpurun_out/r5/long_mp4.txt 2>&1; echo "4 ranks 10000 near delta rc=$?"; tail -1 gpurun_out/r5/long_mp4.txt | cut -c1-420
LX_SOAK_POINTS=near FPSQ_P2P_POLLS=3000000 timeout -k 10 500 python tools/lx_soak_mp.py 10000 2 0 200000 > gpurun_out/r5/long_mp2.txt 2>&1; echo "2 ranks 10000 near rc=$?"; tail -1 gpurun_out/r5/long_mp2.txt | cut -c1-420
