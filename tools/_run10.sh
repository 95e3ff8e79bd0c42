mkdir -p gpurun_out/r03
run() { # tag cfg env...
tag=$1; cfg=$2; shift; shift
env "$@" timeout -k 10 300 python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03/m_$tag.json 2> gpurun_out/r03/m_$tag.err; echo rc $?
python - <<PY
import json
l=json.loads(open("gpurun_out/r03/m_$tag.json").read().strip().splitlines()[-1])
print("$tag", l["value"], l["ms_per_step"], l["roofline"]["pipeline_frac"], l["verified"]["ok"], l["verified"]["max_rms_vs_oracle"], {k[3:-7]:v["avg_ms"] for k,v in l["roofline"]["per_kernel"].items()})
PY
}
run cfg2_r8 cfg2 AUDIOMOD_PV_RES_ROWS=8
run cfg2_r16 cfg2 AUDIOMOD_PV_RES_ROWS=16
run m7_r8 cfg4_formant-7 AUDIOMOD_PV_RES_ROWS=8
run m7_r16 cfg4_formant-7 AUDIOMOD_PV_RES_ROWS=16
run cfg2_r8_np cfg2 AUDIOMOD_PV_RES_ROWS=8 AUDIOMOD_PV_PIPELINE=0
run cfg2_r16_np cfg2 AUDIOMOD_PV_RES_ROWS=16 AUDIOMOD_PV_PIPELINE=0
