export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp && rocprofv3 -L 2>/dev/null | grep -oE "^\s*(Name|name)\s*:\s*\S+|^\S+_\S+" | head -0
cd $GRAFT_REPO_ROOT
rocprofv3 -L > gpurun_out/r4_pmc_list.txt 2>&1; grep -c "" gpurun_out/r4_pmc_list.txt
O=gpurun_out/r4pmc
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/sq -o g -- python3 bench.py --steps 2 --warmup 1 --cpu-evals 0 --repeats 1 --no-roofline-pass > $O/sq.log 2>&1; echo "sq rc=$?"
python3 tools/pmc_kernel_summary.py $O/sq/g_counter_collection.csv 2>&1 | grep "k_spmv" | cut -c1-400
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $O/in -o g -- python3 bench.py --steps 2 --warmup 1 --cpu-evals 0 --repeats 1 --no-roofline-pass > $O/in.log 2>&1; echo "insts rc=$?"
python3 tools/pmc_kernel_summary.py $O/in/g_counter_collection.csv 2>&1 | grep "k_spmv" | cut -c1-400
timeout -k 10 300 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum --kernel-trace --output-format csv -d $O/tc -o g -- python3 bench.py --steps 2 --warmup 1 --cpu-evals 0 --repeats 1 --no-roofline-pass > $O/tc.log 2>&1; echo "tc rc=$?"; tail -3 $O/tc.log
python3 tools/pmc_kernel_summary.py $O/tc/g_counter_collection.csv 2>&1 | grep "k_spmv" | cut -c1-400
find $O -name "*kernel_trace.csv" -delete
