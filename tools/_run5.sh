mkdir -p gpurun_out/r03
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/sincos_hw_probe.hip -o /tmp/sincos_probe && /tmp/sincos_probe > gpurun_out/r03/sincos_hw_probe.txt 2>&1; cat gpurun_out/r03/sincos_hw_probe.txt
python -m pytest tests -m gpu -x -q > gpurun_out/r03/t4.txt 2>&1; echo "suite rc $?"; tail -15 gpurun_out/r03/t4.txt
run() { # tag env...
tag=$1; shift
env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03/g_$tag.json 2> gpurun_out/r03/g_$tag.err; echo rc $?
python - <<PY
import json
l=json.loads(open("gpurun_out/r03/g_$tag.json").read().strip().splitlines()[-1])
print("$tag", l["value"], l["ms_per_step"], l["verified"]["ok"], l["verified"]["max_rms_vs_oracle"], l["verified"]["batch_checksum_sha256"][:12], {k[3:-7]:v["avg_ms"] for k,v in l["roofline"]["per_kernel"].items()})
PY
}
run fast
run fast_np AUDIOMOD_PV_PIPELINE=0
run exact AUDIOMOD_PV_EXACT=1
