mkdir -p gpurun_out/r03
run() { # tag env...
tag=$1; shift
env "$@" timeout -k 10 300 python bench.py --no-verify --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03/n_$tag.json 2> gpurun_out/r03/n_$tag.err; echo rc $?
python - <<PY
import json
l=json.loads(open("gpurun_out/r03/n_$tag.json").read().strip().splitlines()[-1])
print("$tag", l["ms_per_step"], {k[3:-7]:v["avg_ms"] for k,v in l["roofline"]["per_kernel"].items()})
PY
}
L=$PWD/audiomod_amd/lib/diag
run np_base AUDIOMOD_PV_PIPELINE=0
run np_nox AUDIOMOD_PV_PIPELINE=0 AUDIOMOD_PV_LIB=$L/res1/libaudiomod_pv.so
run np_nocoef AUDIOMOD_PV_PIPELINE=0 AUDIOMOD_PV_LIB=$L/res2/libaudiomod_pv.so
run np_neither AUDIOMOD_PV_PIPELINE=0 AUDIOMOD_PV_LIB=$L/res3/libaudiomod_pv.so
