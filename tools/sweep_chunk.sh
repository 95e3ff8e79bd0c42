#!/bin/bash
# usage: tools/sweep_chunk.sh "32 64 128" [extra bench args]
for T in $1; do
  AUDIOMOD_PV_CHUNK_SLICES=$T timeout -k 10 200 python bench.py --no-cpu-baseline ${@:2} 2>/dev/null > /tmp/b_$T.json
  python - <<PY
import json
d=json.load(open("/tmp/b_$T.json"))
print("Tc", $T, d["value"], d["ms_per_step"], {k:v["avg_ms"] for k,v in d["roofline"]["per_kernel"].items()})
PY
done
