#!/bin/bash
# bench.py with diagnostic builds and environment settings: tools/ab_lib.sh <tag> "<lib|-> [VAR=..]..." ...
T=$1; shift
mkdir -p gpurun_out/r02
O=gpurun_out/r02/abl_$T.txt
: > $O
for E in "$@"; do
  set -- $E
  L=$1; shift
  EV="$*"
  if [ "$L" != "-" ]; then EV="$EV AUDIOMOD_PV_LIB=$PWD/audiomod_amd/lib/diag/$L/libaudiomod_pv.so"; fi
  env $EV timeout -k 10 200 python bench.py --no-cpu-baseline > /tmp/ab.json 2>/tmp/ab.err
  python - "$E" >> $O <<'PY'
import json,sys
try:
    d=json.load(open("/tmp/ab.json"))
    print(sys.argv[1].ljust(40), d["value"], d["ms_per_step"], {k.replace("pv_","").replace("_kernel",""):v["avg_ms"] for k,v in d["roofline"]["per_kernel"].items()}, d["verified"]["ok"], d["verified"]["batch_checksum_sha256"][:12])
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("/tmp/ab.err").read()[-400:])
PY
done
cat $O
