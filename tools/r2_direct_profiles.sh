#!/bin/bash
# developer script (GPU box): the direct back-ends' part of profiles/ (run after tools/r2_profiles.sh)
export TMPDIR=/tmp
O=gpurun_out/r2prof
mkdir -p $O
timeout -k 10 250 python bench.py --workload "dense-block n=4096 m=2048" --steps 10 --warmup 2 > $O/bench_dense_block.json 2>> $O/err.log; echo "dense rc=$?"
rm -rf $O/dks; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dks -o d -- python3 bench.py --workload "dense-block n=4096 m=2048" --steps 5 --warmup 1 --cpu-evals 0 --repeats 1 > $O/dks.log 2>&1; echo "dense stats rc=$?"
rm -rf $O/bks; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bks -o b -- python3 tools/band_headline.py > $O/band_headline_prof.txt 2>&1; echo "band stats rc=$?"
timeout -k 10 200 python tools/band_headline.py > $O/band_headline.txt 2>&1; echo "band rc=$?"
timeout -k 10 60 ./tools/potrf_probe 5 > $O/potrf_phase_probe.txt 2>&1; echo "potrf probe rc=$?"
timeout -k 10 60 ./tools/mfma_probe > $O/mfma_probe.txt 2>&1; echo "mfma probe rc=$?"
rm -rf $O/gp; timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/gp -o g -- python3 bench.py --workload "dense-block n=4096 m=2048" --steps 2 --warmup 1 --cpu-evals 0 --repeats 1 > $O/gp.log 2>&1; echo "dense pmc rc=$?"
python3 tools/pmc_kernel_summary.py $O/gp/g_counter_collection.csv > $O/dense_pmc_sq.txt 2>&1
rm -f $O/dks/d_kernel_trace.csv $O/bks/b_kernel_trace.csv $O/gp/g_kernel_trace.csv
ls $O
