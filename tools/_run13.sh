mkdir -p gpurun_out/r03
run() { # tag cfg env...
tag=$1; cfg=$2; shift; shift
env "$@" timeout -k 10 300 python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03/p_$tag.json 2> gpurun_out/r03/p_$tag.err; echo rc $?
python - <<PY
import json
l=json.loads(open("gpurun_out/r03/p_$tag.json").read().strip().splitlines()[-1])
print("$tag", l["value"], l["ms_per_step"], l["roofline"]["pipeline_frac"], l["verified"]["ok"], l["verified"]["batch_checksum_sha256"][:12], {k[3:-7]:v["avg_ms"] for k,v in l["roofline"]["per_kernel"].items()})
PY
}
run cfg2_single cfg2 AUDIOMOD_PV_RES_MULTI=0
run cfg2_ng4 cfg2 AUDIOMOD_PV_RES_MULTI=1
run cfg2_ng2 cfg2 AUDIOMOD_PV_RES_MULTI=2
run cfg2_ng1 cfg2 AUDIOMOD_PV_RES_MULTI=3
run m7_ng2 cfg4_formant-7 AUDIOMOD_PV_RES_MULTI=2
run m7_ng1 cfg4_formant-7 AUDIOMOD_PV_RES_MULTI=3
