#!/bin/bash
# bench the default library and diagnostic builds one after the other on one box: tools/variants_run.sh <tag> <name>...
T=$1; shift
mkdir -p gpurun_out/r02
O=gpurun_out/r02/var_$T.txt
: > $O
show() { python - "$1" "$2" <<'PY' >> "$O"
import json,sys
try:
    d=json.load(open(sys.argv[2]))
    print(sys.argv[1], d["value"], d["ms_per_step"], {k:v["avg_ms"] for k,v in d["roofline"]["per_kernel"].items()}, d["verified"]["ok"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
export O
timeout -k 10 200 python bench.py --no-cpu-baseline > /tmp/b_default.json 2>/tmp/b_default.err; show default /tmp/b_default.json
for N in "$@"; do
  timeout -k 10 200 python tools/diag_run.py $N --no-cpu-baseline > /tmp/b_$N.json 2>/tmp/b_$N.err; show $N /tmp/b_$N.json
done
cat $O
