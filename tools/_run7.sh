mkdir -p gpurun_out/r03
run() { # tag env...
tag=$1; shift
env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03/i_$tag.json 2> gpurun_out/r03/i_$tag.err; echo rc $?
python - <<PY
import json
l=json.loads(open("gpurun_out/r03/i_$tag.json").read().strip().splitlines()[-1])
print("$tag", l["value"], l["ms_per_step"], l["verified"]["ok"], l["verified"]["max_rms_vs_oracle"], l["verified"]["batch_checksum_sha256"][:12], l["config"]["launches_per_step"], {k[3:-7]:v["avg_ms"] for k,v in l["roofline"]["per_kernel"].items()})
PY
}
run base
run c768 AUDIOMOD_PV_CHUNK_SLICES=768
run c1024 AUDIOMOD_PV_CHUNK_SLICES=1024
run c384 AUDIOMOD_PV_CHUNK_SLICES=384
run resstream AUDIOMOD_PV_RES_STREAM=1
run base2
