export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "spmv or column_sorted or riding or one_pass" > gpurun_out/r4_c13.log 2>&1; rc=$?; echo "test rc=$rc"; tail -4 gpurun_out/r4_c13.log | cut -c1-200
[ $rc -ne 0 ] && exit 1
for sh in 1 0 1 0; do
FPSQ_AT_SHARED=$sh timeout -k 10 300 python bench.py --cpu-evals 0 --no-roofline-pass > gpurun_out/r4_c13_sh$sh.json 2> /dev/null; echo "shared=$sh: $(cut -c1-140 gpurun_out/r4_c13_sh$sh.json)"
done
bash tools/r4_c12.sh
