#!/bin/bash
# rocprofv3 kernel stats of one BASELINE config: tools/profile_config.sh <config> <tag>  -> gpurun_out/<tag>_<config>_kernel_stats.csv
set -e
C=$1; T=${2:-x}
export TMPDIR=/tmp
O=$PWD/gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/${T}_prof_$C -o run --output-format csv -- python3 bench.py --config $C --no-cpu-baseline --steps 3 --warmup 1 > $O/${T}_${C}_bench_under_rocprof.json 2> $O/${T}_${C}_prof.err
cp $(ls $O/${T}_prof_$C/*/run_kernel_stats.csv $O/${T}_prof_$C/run_kernel_stats.csv 2>/dev/null | head -1) $O/${T}_${C}_kernel_stats.csv
rm -rf $O/${T}_prof_$C
head -7 $O/${T}_${C}_kernel_stats.csv | cut -c1-140
