mkdir -p gpurun_out
timeout -k 10 900 python tools/report_configs.py > gpurun_out/r04_configs.md 2> gpurun_out/r04_configs.err; echo rc=$?; tail -3 gpurun_out/r04_configs.err; cat gpurun_out/r04_configs.md | cut -c1-260
timeout -k 10 600 python tools/shape_check.py > gpurun_out/r04_shape_check.txt 2>&1; echo rc=$?; cat gpurun_out/r04_shape_check.txt | grep -v amdgpu.ids
