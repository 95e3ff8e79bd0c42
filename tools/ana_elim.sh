#!/bin/bash
# analysis-kernel elimination builds (tools/build_variant.sh anaN -DPV_EXP_ANA=N; results invalid, timings not), one stream
# (AUDIOMOD_PV_PIPELINE=0) so that the kernel runs alone.  usage: tools/ana_elim.sh <out.txt>
out=${1:-gpurun_out/ana_elim.txt}
: > $out
L=$PWD/audiomod_amd/lib/diag
for v in base ana1 ana2 ana3 ana4 ana8 ana16 ana28 ana31; do
  if [ $v = base ]; then lib=$PWD/audiomod_amd/lib/libaudiomod_pv.so; else lib=$L/$v/libaudiomod_pv.so; fi
  line=$(AUDIOMOD_PV_LIB=$lib AUDIOMOD_PV_PIPELINE=0 timeout -k 10 200 python bench.py --no-verify --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1)
  python - "$v" "$line" >> $out <<'PY'
import json, sys
try:
    l = json.loads(sys.argv[2]); print(sys.argv[1].ljust(8), l["ms_per_step"], {k[3:-7]: v["avg_ms"] for k, v in l["roofline"]["per_kernel"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
cat $out
