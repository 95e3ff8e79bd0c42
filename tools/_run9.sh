mkdir -p gpurun_out/r03
run() { # tag cfg env...
tag=$1; cfg=$2; shift; shift
env "$@" timeout -k 10 300 python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03/l_$tag.json 2> gpurun_out/r03/l_$tag.err; echo rc $?
python - <<PY
import json
l=json.loads(open("gpurun_out/r03/l_$tag.json").read().strip().splitlines()[-1])
print("$tag", l["value"], l["ms_per_step"], l["roofline"]["pipeline_frac"], l["verified"]["ok"], l["config"]["launches_per_step"], {k[3:-7]:v["avg_ms"] for k,v in l["roofline"]["per_kernel"].items()})
PY
}
for c in 128 256 384; do
run m7_c$c cfg4_formant-7 AUDIOMOD_PV_CHUNK_SLICES=$c
run cfg3_c$c cfg3 AUDIOMOD_PV_CHUNK_SLICES=$c
done
run cfg2_c256 cfg2 AUDIOMOD_PV_CHUNK_SLICES=256
