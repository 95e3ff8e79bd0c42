#!/bin/bash
# One round's measurement set on the GPU box (run through gpurun):  tools/profile_round.sh <tag>
#   gpurun_out/<tag>_bench.json                 default bench.py line (with cpu_baseline)
#   gpurun_out/<tag>_bench_under_rocprof.json   the same command under rocprofv3 --kernel-trace --stats
#   gpurun_out/<tag>_kernel_stats.csv           its per-kernel summary
#   gpurun_out/<tag>_pmc/                       FETCH_SIZE and WRITE_SIZE passes (separate --pmc runs)
#   gpurun_out/<tag>_traffic.json               HBM bytes per launch from those passes
#   gpurun_out/<tag>_valu.json                  vector instructions per slice (SQ_INSTS_VALU pass)
set -e
T=${1:-vX}
export TMPDIR=/tmp
O=$PWD/gpurun_out
timeout -k 10 400 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/${T}_prof -o run --output-format csv -- python3 bench.py --no-cpu-baseline > $O/${T}_bench_under_rocprof.json 2> $O/${T}_prof.err
cp $(ls $O/${T}_prof/*/run_kernel_stats.csv $O/${T}_prof/run_kernel_stats.csv 2>/dev/null | head -1) $O/${T}_kernel_stats.csv
rm -rf $O/${T}_prof
echo "stats done"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C -d $O/${T}_pmc/$C -o pmc --output-format csv -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 --seconds 20 --sample-every 0 > /dev/null 2> $O/${T}_pmc_$C.err || echo "pmc pass $C returned non-zero"
done
python tools/make_traffic.py $O/${T}_pmc $O/${T}_traffic.json > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d $O/${T}_pmc/VALU -o pmc --output-format csv -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 --seconds 20 --sample-every 0 > /dev/null 2> $O/${T}_pmc_VALU.err || echo "pmc pass VALU returned non-zero"
SPL=$(python -c "import json; print(json.load(open('$O/${T}_bench.json'))['roofline']['slices_per_launch'])")
python tools/make_valu.py $O/${T}_pmc/VALU 131072 $O/${T}_valu.json > /dev/null   # (the 20 s pass's median launch is a full 512-slice chunk of 256 rows)
python tools/pmc_summary.py $O/${T}_pmc/VALU > $O/${T}_pmc_valu_summary.txt
find $O/${T}_pmc -name "*kernel_trace.csv" -delete
echo "pmc done (bench slices per launch $SPL)"
python - <<PY
import json
d = json.load(open("$O/${T}_bench.json"))
print(d["value"], d["unit"], d["ms_per_step"], "ms/step", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("copy_ceiling_GBps"))
print({k: v["avg_ms"] for k, v in d["roofline"]["per_kernel"].items()})
print(d.get("cpu_baseline"))
PY
