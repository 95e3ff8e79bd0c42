mkdir -p gpurun_out/r03
run() { # tag env...
tag=$1; shift
env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03/j_$tag.json 2> gpurun_out/r03/j_$tag.err; echo rc $?
python - <<PY
import json
l=json.loads(open("gpurun_out/r03/j_$tag.json").read().strip().splitlines()[-1])
print("$tag", l["value"], l["ms_per_step"], l["verified"]["ok"], l["verified"]["batch_checksum_sha256"][:12], {k[3:-7]:v["avg_ms"] for k,v in l["roofline"]["per_kernel"].items()})
PY
}
run base
run prio0 AUDIOMOD_PV_SEQ_PRIO=0
run narrow AUDIOMOD_PV_SEQ_NARROW=1
run narrow_prio0 AUDIOMOD_PV_SEQ_NARROW=1 AUDIOMOD_PV_SEQ_PRIO=0
run ring8 AUDIOMOD_PV_SEQ_DEPTH=8
run base2
