# developer script (GPU box): full-size rehearsals of the DEFAULT choices between processes sharing one device (FPSQ_LX, FPSQ_FUSE_ITER unforced:
# sums over the ranks by the gather kernels, because the ranks share a device)
mkdir -p gpurun_out/r5
FPSQ_LX=1 FPSQ_FUSE_ITER=1 timeout -k 10 500 python tools/lx_soak_mp.py 400 2 0 1000000 > gpurun_out/r5/lx_soak_mp2_full.txt 2>&1; echo "2 ranks n=1e6 defaults rc=$?"; tail -1 gpurun_out/r5/lx_soak_mp2_full.txt | cut -c1-420
FPSQ_LX=1 FPSQ_FUSE_ITER=1 timeout -k 10 500 python tools/lx_soak_mp.py 400 4 1.4901161193847656e-08 1000000 > gpurun_out/r5/lx_soak_mp4_full.txt 2>&1; echo "4 ranks n=1e6 defaults delta rc=$?"; tail -1 gpurun_out/r5/lx_soak_mp4_full.txt | cut -c1-420
