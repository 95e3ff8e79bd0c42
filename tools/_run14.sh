mkdir -p gpurun_out/r03
run() { # tag streams env...
tag=$1; st=$2; shift; shift
env "$@" timeout -k 10 300 python bench.py --streams $st --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03/s_$tag.json 2> gpurun_out/r03/s_$tag.err; echo rc $?
python - <<PY
import json
l=json.loads(open("gpurun_out/r03/s_$tag.json").read().strip().splitlines()[-1])
print("$tag", l["value"], l["ms_per_step"], l["verified"]["ok"], l["config"]["launches_per_step"], {k[3:-7]:v["avg_ms"] for k,v in l["roofline"]["per_kernel"].items()})
PY
}
run s20_base 20
run s20_c512 20 AUDIOMOD_PV_CHUNK_SLICES=512
run s20_c1024 20 AUDIOMOD_PV_CHUNK_SLICES=1024
run s20_c512_d8 20 AUDIOMOD_PV_CHUNK_SLICES=512 AUDIOMOD_PV_SEQ_DEPTH=8
run s20_c512_ring0 20 AUDIOMOD_PV_CHUNK_SLICES=512 AUDIOMOD_PV_SEQ_RING=0
run s64_c512 64 AUDIOMOD_PV_CHUNK_SLICES=512
run s64_fused 64 AUDIOMOD_PV_FUSED=2
run s20_fused 20 AUDIOMOD_PV_FUSED=2 AUDIOMOD_PV_CHUNK_SLICES=512
