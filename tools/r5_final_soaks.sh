# developer script (GPU box): every soak of the round once more on the final tree
mkdir -p gpurun_out/r5
bash tools/r5_soak.sh 2>&1 | grep "rc=" 
bash tools/r5_soak3.sh 2>&1 | grep "rc="
bash tools/r5_soak5.sh 2>&1 | grep "rc="
bash tools/r5_soak7.sh 2>&1 | grep "rc="
timeout -k 10 600 python tools/adopt_soak.py 600 > gpurun_out/r5/adopt_soak.txt 2>&1; echo "adopt soak rc=$?"
timeout -k 10 600 python tools/direct_soak.py 60 40 8 > gpurun_out/r5/direct_soak2.txt 2>&1; echo "direct soak rc=$?"
timeout -k 10 600 python tools/minres_soak.py 300 > gpurun_out/r5/minres_soak2.txt 2>&1; echo "minres soak rc=$?"
