#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -v -o faulthandler_timeout=100 -k "lnlq" > $O/r2_t4.log 2>&1; rc=$?; echo "lnlq tests rc=$rc"; tail -25 $O/r2_t4.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python -m pytest tests -m gpu -x -q -o faulthandler_timeout=100 > $O/r2_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/r2_tests.log
