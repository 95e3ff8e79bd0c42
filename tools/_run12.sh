mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fast_arith or golden_batch or bench" > gpurun_out/r03/t6.txt 2>&1; echo "subset rc $?"; tail -4 gpurun_out/r03/t6.txt
run() { # tag cfg env...
tag=$1; cfg=$2; shift; shift
env "$@" timeout -k 10 300 python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03/o_$tag.json 2> gpurun_out/r03/o_$tag.err; echo rc $?
python - <<PY
import json
l=json.loads(open("gpurun_out/r03/o_$tag.json").read().strip().splitlines()[-1])
print("$tag", l["value"], l["ms_per_step"], l["roofline"]["pipeline_frac"], l["verified"]["ok"], l["verified"]["max_rms_vs_oracle"], l["verified"]["batch_checksum_sha256"][:12], {k[3:-7]:v["avg_ms"] for k,v in l["roofline"]["per_kernel"].items()})
PY
}
run cfg2_multi cfg2
run cfg2_single cfg2 AUDIOMOD_PV_RES_MULTI=0
run m7_multi cfg4_formant-7
run m7_single cfg4_formant-7 AUDIOMOD_PV_RES_MULTI=0
run cfg2_multi_np cfg2 AUDIOMOD_PV_PIPELINE=0
