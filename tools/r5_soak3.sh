# developer script (GPU box): the other routes of the sharded path between processes, soaked
mkdir -p gpurun_out/r5
FPSQ_LX=0 FPSQ_P2P_POLLS=3000000 timeout -k 10 400 python tools/lx_soak_mp.py 2000 3 0 100000 > gpurun_out/r5/lx_soak_mp3_gather.txt 2>&1; echo "3 ranks, gather kernels (FPSQ_LX=0) rc=$?"; tail -1 gpurun_out/r5/lx_soak_mp3_gather.txt | cut -c1-420
FPSQ_LX=0 FPSQ_HALO_FUSE=0 FPSQ_P2P_POLLS=3000000 timeout -k 10 400 python tools/lx_soak_mp.py 2000 3 1.4901161193847656e-08 100000 > gpurun_out/r5/lx_soak_mp3_gather2.txt 2>&1; echo "3 ranks, gather kernels, exchange and finish apart, delta rc=$?"; tail -1 gpurun_out/r5/lx_soak_mp3_gather2.txt | cut -c1-420
FPSQ_FUSE_ITER=0 FPSQ_P2P_POLLS=3000000 timeout -k 10 400 python tools/lx_soak_mp.py 2000 3 0 100000 > gpurun_out/r5/lx_soak_mp3_unfused.txt 2>&1; echo "3 ranks, in-launch sums, three launches rc=$?"; tail -1 gpurun_out/r5/lx_soak_mp3_unfused.txt | cut -c1-420
FPSQ_P2P_POLLS=3000000 timeout -k 10 400 python tools/lx_soak_mp.py 3000 3 0 100000 > gpurun_out/r5/lx_soak_mp3_h.txt 2>&1; echo "3 ranks, one launch, with hprod rc=$?"; tail -1 gpurun_out/r5/lx_soak_mp3_h.txt | cut -c1-420
FPSQ_P2P_POLLS=3000000 timeout -k 10 400 python tools/lx_soak_mp.py 2000 4 0 100000 > gpurun_out/r5/lx_soak_mp4.txt 2>&1; echo "4 ranks, one launch rc=$?"; tail -2 gpurun_out/r5/lx_soak_mp4.txt | cut -c1-420
