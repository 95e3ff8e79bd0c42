#!/bin/bash
# bench.py under several environment settings, one summary line each: tools/ab_env.sh <tag> "VAR=.. VAR2=.." "..." (use "-" for none)
T=$1; shift
mkdir -p gpurun_out/r02
O=gpurun_out/r02/ab_$T.txt
: > $O
for E in "$@"; do
  if [ "$E" = "-" ]; then EV=""; else EV="$E"; fi
  env $EV timeout -k 10 200 python bench.py --no-cpu-baseline > /tmp/ab.json 2>/tmp/ab.err
  python - "$E" >> $O <<'PY'
import json,sys
try:
    d=json.load(open("/tmp/ab.json"))
    print(sys.argv[1].ljust(40), d["value"], d["ms_per_step"], {k.replace("pv_","").replace("_kernel",""):v["avg_ms"] for k,v in d["roofline"]["per_kernel"].items()}, d["verified"]["ok"], d["verified"]["max_rms_vs_oracle"], d["verified"]["batch_checksum_sha256"][:12])
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("/tmp/ab.err").read()[-400:])
PY
done
cat $O
