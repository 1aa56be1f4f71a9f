mkdir -p gpurun_out
AB_LIBS=tools/ab/libfpsq_v1.so,fletcherpenaltysolver.jl_amd/lib/libfpsq.so AB_ENV="FPSQ_AT_SHARED=0,1,0,1" timeout -k 10 500 python tools/ab_modes.py 8 20 > gpurun_out/r4_c16.log 2>&1; echo rc=$?; tail -12 gpurun_out/r4_c16.log
