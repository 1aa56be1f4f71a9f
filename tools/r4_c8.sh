export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "one_pass or duplicates or coo" > gpurun_out/r4_c8.log 2>&1; rc=$?; echo "test rc=$rc"; tail -15 gpurun_out/r4_c8.log | cut -c1-200
[ $rc -ne 0 ] && exit 1
for p in device device+jac host+jac; do timeout -k 10 300 python bench.py --pointers $p --cpu-evals 0 > gpurun_out/r4_c8_$p.json 2> gpurun_out/r4_c8_$p.err; echo "$p rc=$? $(cut -c1-130 gpurun_out/r4_c8_$p.json)"; done
FPSQ_JAC_REFRESH=3 timeout -k 10 300 python bench.py --pointers device+jac --cpu-evals 0 > gpurun_out/r4_c8_3pass.json 2> /dev/null; echo "3pass device+jac: $(cut -c1-130 gpurun_out/r4_c8_3pass.json)"
rm -rf gpurun_out/r4_c8_ks; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_c8_ks -o k -- python3 bench.py --pointers device+jac --steps 10 --warmup 2 --cpu-evals 0 --repeats 2 --no-roofline-pass > gpurun_out/r4_c8_ks.log 2>&1; echo "ks rc=$?"
find gpurun_out/r4_c8_ks -name "*kernel_trace.csv" -delete
grep -h "k_refresh\|k_gather" $(find gpurun_out/r4_c8_ks -name "*kernel_stats.csv") | cut -c1-200
