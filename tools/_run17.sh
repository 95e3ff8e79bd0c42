mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q > gpurun_out/r03/t8.txt 2>&1; echo "suite rc $?"; tail -12 gpurun_out/r03/t8.txt
run() { # tag cfg env...
tag=$1; cfg=$2; shift; shift
env "$@" timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline > gpurun_out/r03/v_$tag.json 2> gpurun_out/r03/v_$tag.err; echo rc $?
python - <<PY
import json
l=json.loads(open("gpurun_out/r03/v_$tag.json").read().strip().splitlines()[-1])
print("$tag", l["value"], l["ms_per_step"], l["roofline"]["pipeline_frac"], l["verified"]["ok"], l["verified"]["max_rms_vs_oracle"], l["verified"]["batch_checksum_sha256"][:12], {k[3:-7]:v["avg_ms"] for k,v in l["roofline"]["per_kernel"].items()})
PY
}
run cfg2_xc cfg2
run cfg2_polar cfg2 AUDIOMOD_PV_XC=0
run cfg3_xc cfg3
run cfg3_polar cfg3 AUDIOMOD_PV_XC=0
run cfg2_xc_np cfg2 AUDIOMOD_PV_PIPELINE=0
