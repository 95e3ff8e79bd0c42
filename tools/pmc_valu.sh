#!/bin/bash
# VALU-side counters of the bench workload (separate --pmc passes, kernel trace only): tools/pmc_valu.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/pmcv_*
for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  n=$(echo $P | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P -d gpurun_out/pmcv_$n -o pmc --output-format csv -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 --seconds 20 --sample-every 0 > /dev/null 2>gpurun_out/pmcv_$n.err || echo "pmc $n returned non-zero"
done
python tools/pmc_summary.py gpurun_out/pmcv_* > gpurun_out/pmcv_summary.txt
find gpurun_out -name "*kernel_trace.csv" -delete
cat gpurun_out/pmcv_summary.txt
