mkdir -p gpurun_out/r03
run() { # tag streams env...
tag=$1; st=$2; shift; shift
env "$@" timeout -k 10 300 python bench.py --streams $st --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03/u_$tag.json 2> gpurun_out/r03/u_$tag.err; echo rc $?
python - <<PY
import json
l=json.loads(open("gpurun_out/r03/u_$tag.json").read().strip().splitlines()[-1])
print("$tag", l["value"], l["ms_per_step"], l["verified"]["ok"], l["config"]["launches_per_step"], {k[3:-7]:v["avg_ms"] for k,v in l["roofline"]["per_kernel"].items()})
PY
}
for s in 4 8 16 32; do
run s${s}_tile $s AUDIOMOD_PV_CHUNK_SLICES=512
run s${s}_fused $s AUDIOMOD_PV_FUSED=2 AUDIOMOD_PV_CHUNK_SLICES=512
done
run s64_fused_c512 64 AUDIOMOD_PV_FUSED=2 AUDIOMOD_PV_CHUNK_SLICES=512
run s96_tile 95
run s96_fused 95 AUDIOMOD_PV_FUSED=2
