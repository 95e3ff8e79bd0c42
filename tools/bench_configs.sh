#!/bin/bash
# all BASELINE configs through bench.py (numbers for DESIGN.md); run on the GPU box
for C in cfg2 cfg3 cfg4_formant+7 cfg4_formant-7 cfg4_gender+7 cfg4_gender-7; do
  timeout -k 10 300 python bench.py --config $C --no-cpu-baseline 2>/dev/null > /tmp/b.json
  python - <<PY
import json
d=json.load(open("/tmp/b.json"))
print("$C", d["value"], "Msamples/s", d["ms_per_step"], "ms/step", d["x_realtime_per_gpu"], "xRT", d["roofline"]["stage"], d["roofline"]["achieved"], "GB/s", {k:v["avg_ms"] for k,v in d["roofline"]["per_kernel"].items()})
PY
done
for CM in 0 2; do
  timeout -k 10 300 python bench.py --coremode $CM --no-cpu-baseline 2>/dev/null > /tmp/b.json
  python - <<PY
import json
d=json.load(open("/tmp/b.json"))
print("cfg2 coremode $CM", d["value"], "Msamples/s", d["ms_per_step"], "ms/step", d["x_realtime_per_gpu"], "xRT", {k:v["avg_ms"] for k,v in d["roofline"]["per_kernel"].items()})
PY
done
