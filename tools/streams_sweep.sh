#!/bin/bash
# the bench workload at other batch sizes (every line verified against the oracle).  usage: tools/streams_sweep.sh <out.txt> [streams...]
out=${1:-gpurun_out/streams_sweep.txt}; shift
S=${@:-20 64 97 100 128 130 160 192}
: > $out
for s in $S; do
  line=$(timeout -k 10 300 python bench.py --streams $s --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1)
  python - "$s" "$line" >> $out <<'PY'
import json, sys
try:
    l = json.loads(sys.argv[2])
    pk = {k.replace("pv_", "").replace("_kernel", ""): v["avg_ms"] for k, v in l["roofline"]["per_kernel"].items()}
    print(f'{int(sys.argv[1]):4d} streams {l["value"]:9.1f} Msamples/s {l["ms_per_step"]:7.2f} ms/step  launches {l["config"]["launches_per_step"]:3d}  verified {l["verified"]["ok"]} rms {l["verified"]["max_rms_vs_oracle"]:.2e}  {pk}')
except Exception as e:
    print(sys.argv[1], "FAILED", e, sys.argv[2][:200])
PY
done
cat $out
