#!/bin/bash
# bench every diagnostic build under audiomod_amd/lib/diag/ and print per-kernel times
for d in audiomod_amd/lib/diag/*/; do
  n=$(basename $d)
  timeout -k 10 120 python tools/diag_run.py $n --no-cpu-baseline --steps 3 --warmup 1 > /tmp/d.json 2>/tmp/d.err || { echo "$n failed"; tail -3 /tmp/d.err; continue; }
  python - "$n" <<'PY'
import json, sys
d = json.load(open("/tmp/d.json"))
print(f"{sys.argv[1]:14s} {d['ms_per_step']:8.3f} ms/step ", {k.replace('pv_','').replace('_kernel',''): v["avg_ms"] for k, v in d["roofline"]["per_kernel"].items()})
PY
done
