mkdir -p gpurun_out/r03
echo skip suite
run() { # tag env...
tag=$1; shift
env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03/h_$tag.json 2> gpurun_out/r03/h_$tag.err; echo rc $?
python - <<PY
import json
l=json.loads(open("gpurun_out/r03/h_$tag.json").read().strip().splitlines()[-1])
print("$tag", l["value"], l["ms_per_step"], l["verified"]["ok"], l["verified"]["max_rms_vs_oracle"], l["verified"]["batch_checksum_sha256"][:12], {k[3:-7]:v["avg_ms"] for k,v in l["roofline"]["per_kernel"].items()})
PY
}
run ahead1
run ahead0 AUDIOMOD_PV_AHEAD=0
run ahead1_ring0 AUDIOMOD_PV_SEQ_RING=0
run ahead1_b
