#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -v -o faulthandler_timeout=100 -k "extras or halo" > $O/r2_t3.log 2>&1; rc=$?; echo "extras tests rc=$rc"; tail -15 $O/r2_t3.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python -m pytest tests -m gpu -x -q -o faulthandler_timeout=100 > $O/r2_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/r2_tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 200 python bench.py --op extras --cpu-evals 2 > $O/r2_b3_extras.json 2> $O/r2_b3_extras.err; echo "extras rc=$?"; cat $O/r2_b3_extras.json
timeout -k 10 200 python bench.py --op extras --fuse 0 --cpu-evals 0 > $O/r2_b3_extras_unfused.json 2> $O/r2_b3_extras.err; echo "extras unfused rc=$?"; cat $O/r2_b3_extras_unfused.json
timeout -k 10 200 python bench.py --op hprod-solves --cpu-evals 0 > $O/r2_b3_hps.json 2> $O/r2_b3_hps.err; echo "hprod-solves rc=$?"; cat $O/r2_b3_hps.json
