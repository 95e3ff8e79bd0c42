mkdir -p gpurun_out/r03
run() { # tag env...
tag=$1; shift
env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03/d_$tag.json 2> gpurun_out/r03/d_$tag.err; echo rc $?
python - <<PY
import json
l=json.loads(open("gpurun_out/r03/d_$tag.json").read().strip().splitlines()[-1])
print("$tag", l["value"], l["ms_per_step"], l["verified"]["ok"], l["verified"]["batch_checksum_sha256"][:12], {k[3:-7]:v["avg_ms"] for k,v in l["roofline"]["per_kernel"].items()})
PY
}
run ring0 AUDIOMOD_PV_SEQ_RING=0
run ring8 AUDIOMOD_PV_SEQ_DEPTH=8
run ring4 AUDIOMOD_PV_SEQ_DEPTH=4
run np_ring0 AUDIOMOD_PV_SEQ_RING=0 AUDIOMOD_PV_PIPELINE=0
run np_ring8 AUDIOMOD_PV_SEQ_DEPTH=8 AUDIOMOD_PV_PIPELINE=0
run np_ring4 AUDIOMOD_PV_SEQ_DEPTH=4 AUDIOMOD_PV_PIPELINE=0
