mkdir -p gpurun_out/r03
run() { # tag env...
tag=$1; shift
env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --steps 4 --warmup 1 > gpurun_out/r03/f_$tag.json 2> gpurun_out/r03/f_$tag.err; echo rc $?
python - <<PY
import json
l=json.loads(open("gpurun_out/r03/f_$tag.json").read().strip().splitlines()[-1])
print("$tag", l["value"], l["ms_per_step"], l["verified"]["ok"], l["verified"]["batch_checksum_sha256"][:12], {k[3:-7]:v["avg_ms"] for k,v in l["roofline"]["per_kernel"].items()})
PY
}
L=$PWD/audiomod_amd/lib/diag
for v in cur nopin oldall diagonly; do
  if [ $v = cur ]; then lib=$PWD/audiomod_amd/lib/libaudiomod_pv.so; else lib=$L/$v/libaudiomod_pv.so; fi
  run np_$v AUDIOMOD_PV_LIB=$lib AUDIOMOD_PV_PIPELINE=0 AUDIOMOD_PV_EXACT=1
done
run np_cur_fast AUDIOMOD_PV_PIPELINE=0
