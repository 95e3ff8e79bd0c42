set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1; tail -2 gpurun_out/t.log
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/b_fastload.json 2>gpurun_out/b.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/b_fastload.json"))
print(d["value"], d["ms_per_step"], {k:v["avg_ms"] for k,v in d["roofline"]["per_kernel"].items()})
PY
export TMPDIR=/tmp
for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  n=$(echo $P | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P -d gpurun_out/pmc_$n -o pmc --output-format csv -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 --sample-every 0 > /dev/null 2>gpurun_out/pmc_$n.err || echo "pmc $n failed"
done
python tools/pmc_summary.py gpurun_out/pmc_* > gpurun_out/pmc_summary.txt
cat gpurun_out/pmc_summary.txt
