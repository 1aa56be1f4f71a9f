# developer script (GPU box): sharded soaks at points whose iteration counts move (the run-ahead mispredicts)
mkdir -p gpurun_out/r5
LX_SOAK_POINTS=near FPSQ_P2P_POLLS=3000000 timeout -k 10 400 python tools/lx_soak_mp.py 2000 3 0 100000 > gpurun_out/r5/lx_soak_mp3_near.txt 2>&1; echo "3 ranks one launch, moving counts rc=$?"; tail -2 gpurun_out/r5/lx_soak_mp3_near.txt | cut -c1-420
LX_SOAK_POINTS=near FPSQ_FUSE_ITER=0 FPSQ_P2P_POLLS=3000000 timeout -k 10 400 python tools/lx_soak_mp.py 2000 3 1.4901161193847656e-08 100000 > gpurun_out/r5/lx_soak_mp3_near_unf.txt 2>&1; echo "3 ranks three launches delta, moving counts rc=$?"; tail -2 gpurun_out/r5/lx_soak_mp3_near_unf.txt | cut -c1-420
LX_SOAK_POINTS=near FPSQ_LX=0 FPSQ_P2P_POLLS=3000000 timeout -k 10 400 python tools/lx_soak_mp.py 2000 2 0 100000 > gpurun_out/r5/lx_soak_mp2_near_gather.txt 2>&1; echo "2 ranks gather kernels, moving counts rc=$?"; tail -2 gpurun_out/r5/lx_soak_mp2_near_gather.txt | cut -c1-420
