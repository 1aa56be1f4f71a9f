#!/bin/bash
# developer script (GPU box): everything the round's profiles/ directory records (round 5)
export TMPDIR=/tmp
O=gpurun_out/r5prof
mkdir -p $O
B="timeout -k 10 200 python bench.py"
# the traffic passes FIRST: the bench lines below then find a profile of THESE kernel sources (bench.py refuses any other) -- the
# box has no .git, so the caller passes the commit: FPSQ_GIT_HEAD=$(git rev-parse --short HEAD) in front of the gpurun command
rm -rf $O/pf; timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -o f -- python3 bench.py --steps 2 --warmup 1 --cpu-evals 0 --repeats 1 --no-roofline-pass > $O/pf.log 2>&1; echo "pmc fetch rc=$?"
rm -rf $O/pw; timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -o w -- python3 bench.py --steps 2 --warmup 1 --cpu-evals 0 --repeats 1 --no-roofline-pass > $O/pw.log 2>&1; echo "pmc write rc=$?"
python3 tools/pmc_traffic.py $O/pf/f_counter_collection.csv $O/pw/w_counter_collection.csv 15 > $O/pmc_traffic.json 2> $O/pmc_traffic.err; echo "pmc post rc=$?"
cp $O/pmc_traffic.json profiles/r05_pmc_traffic.json
$B > $O/bench_headline.json 2> $O/err.log; echo "headline rc=$?"
$B --delta 1.4901161193847656e-08 --cpu-evals 0 > $O/bench_headline_delta_sqrteps.json 2>> $O/err.log; echo "delta rc=$?"
$B --workload "pde-control-like n=1e6 m=1e5 nnz=1e7" --cpu-evals 0 > $O/bench_headline_stratified.json 2>> $O/err.log; echo "stratified rc=$?"
$B --force-shard --cpu-evals 0 > $O/bench_force_shard_world1.json 2>> $O/err.log; echo "force-shard rc=$?"
$B --force-shard --comm-route rccl --cpu-evals 0 > $O/bench_force_shard_world1_rccl.json 2>> $O/err.log; echo "force-shard rccl rc=$?"
$B --cpu-evals 0 > $O/bench_headline_again.json 2>> $O/err.log; echo "headline again rc=$?"
$B --workload "random-eqqp n=1e5 m=1e4 nnz=1e6" --cpu-evals 2 > $O/bench_cfg2.json 2>> $O/err.log; echo "cfg2 rc=$?"
$B --workload "aug2dc-like N=100" --delta 1.4901161193847656e-08 --cpu-evals 1 --steps 5 --repeats 3 > $O/bench_cfg4.json 2>> $O/err.log; echo "cfg4 rc=$?"
$B --workload "dense-block n=4096 m=2048" --steps 10 --warmup 2 > $O/bench_dense_block.json 2>> $O/err.log; echo "dense rc=$?"
$B --op hprod --cpu-evals 0 > $O/bench_hprod.json 2>> $O/err.log; echo "hprod rc=$?"
$B --op hprod --hessian-approx 1 --cpu-evals 0 > $O/bench_hprod_val1.json 2>> $O/err.log; echo "hprod val1 rc=$?"
$B --op extras --cpu-evals 0 > $O/bench_extras.json 2>> $O/err.log; echo "extras rc=$?"
$B --pointers device+jac --cpu-evals 0 > $O/bench_device_pointers_jac.json 2>> $O/err.log; echo "device+jac rc=$?"
rm -rf $O/ks; timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o k -- python3 bench.py --steps 20 --warmup 3 --cpu-evals 0 --repeats 2 > $O/ks.log 2>&1; echo "kernel stats rc=$?"
python3 tools/timeline.py $O/ks/k_kernel_trace.csv k_startup > $O/timeline.txt 2>&1
rm -rf $O/ksf; timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ksf -o k -- python3 bench.py --steps 20 --warmup 3 --cpu-evals 0 --repeats 2 --force-shard > $O/ksf.log 2>&1; echo "kernel stats force-shard rc=$?"
python3 tools/timeline.py $O/ksf/k_kernel_trace.csv k_startup > $O/timeline_force_shard.txt 2>&1
# SQ / TCP counters of the FINAL loop kernel (k_iter_fused): separate --pmc passes, --kernel-trace only
rm -f $O/pmc_sq.txt
n=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum"; do
  n=$((n+1))
  rm -rf $O/sq_$n; timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/sq_$n -o g -- python3 bench.py --steps 2 --warmup 1 --cpu-evals 0 --repeats 1 --no-roofline-pass > $O/sq_$n.log 2>&1; echo "pmc group $n rc=$?"
  python3 tools/pmc_kernel_summary.py $O/sq_$n/g_counter_collection.csv k_ >> $O/pmc_sq.txt 2>&1
done
rm -rf $O/dks; timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dks -o d -- python3 bench.py --workload "dense-block n=4096 m=2048" --steps 5 --warmup 1 --cpu-evals 0 --repeats 1 > $O/dks.log 2>&1; echo "dense stats rc=$?"
find $O -name "*kernel_trace.csv" -delete
ls $O
