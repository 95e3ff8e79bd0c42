mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q > gpurun_out/r03/t3.txt 2>&1; echo "suite rc $?"; tail -5 gpurun_out/r03/t3.txt
for v in 1 0; do
AUDIOMOD_PV_SEQ_RING=$v timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03/c_bench_ring$v.json 2> gpurun_out/r03/c_bench_ring$v.err; echo rc $?
python - <<PY
import json
l=json.loads(open("gpurun_out/r03/c_bench_ring$v.json").read().strip().splitlines()[-1])
print("ring=$v", l["value"], l["ms_per_step"], l["verified"]["max_rms_vs_oracle"], l["verified"]["ok"], l["verified"]["batch_checksum_sha256"][:12])
print({k:v["avg_ms"] for k,v in l["roofline"]["per_kernel"].items()})
PY
done
