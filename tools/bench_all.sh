#!/bin/bash
# every BASELINE config through bench.py (128 stereo streams x 60 s), one summary line each.  usage: tools/bench_all.sh <out.txt> [env...]
out=${1:-gpurun_out/bench_all.txt}
shift
: > $out
for cfg in "cfg2 --coremode 1" "cfg2 --coremode 0" "cfg2 --coremode 2" "cfg3 --coremode 1" "cfg4_formant+7 --coremode 1" "cfg4_formant-7 --coremode 1" "cfg4_gender+7 --coremode 1" "cfg4_gender-7 --coremode 1"; do
  set -- $cfg
  line=$(env "${EXTRA_ENV[@]}" timeout -k 10 400 python bench.py --config $1 $2 $3 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1)
  python - "$cfg" "$line" >> $out <<'PY'
import json, sys
try:
    l = json.loads(sys.argv[2])
    pk = {k.replace("pv_", "").replace("_kernel", ""): v["avg_ms"] for k, v in l["roofline"]["per_kernel"].items()}
    print(sys.argv[1].ljust(28), f'{l["value"]:9.1f} Msamples/s {l["x_realtime_per_gpu"]:9.0f} xRT {l["ms_per_step"]:7.2f} ms/step  pipeline {l["roofline"]["pipeline_GBps"]:7.1f} GB/s ({l["roofline"]["pipeline_frac"]:.3f})  verified {l.get("verified", {}).get("ok")} rms {l.get("verified", {}).get("max_rms_vs_oracle"):.2e}  {pk}')
except Exception as e:
    print(sys.argv[1], "FAILED", e, sys.argv[2][:200])
PY
done
cat $out
