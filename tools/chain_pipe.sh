#!/bin/bash
# pipelined-schedule timings of the fused path for a few waves-per-workgroup settings. usage: tools/chain_pipe.sh <outdir> [extra bench args]
out=${1:-gpurun_out/r02/pipe}
shift
mkdir -p $out
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --sample-every 4 $EXTRA > $out/$tag.json 2> $out/$tag.err || echo "$tag failed"; }
EXTRA="$*"
run w16
run w14 AUDIOMOD_PV_CHAIN_WAVES=14
run w12 AUDIOMOD_PV_CHAIN_WAVES=12
run w11 AUDIOMOD_PV_CHAIN_WAVES=11
run w12nopipe AUDIOMOD_PV_CHAIN_WAVES=12 AUDIOMOD_PV_PIPELINE=0
run w16nopipe AUDIOMOD_PV_PIPELINE=0
run tiles AUDIOMOD_PV_FUSED=0
python - <<'PY' $out
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        l = json.loads(open(f).read().strip().splitlines()[-1])
        pk = {k: v["avg_ms"] for k, v in l["roofline"]["per_kernel"].items()}
        print(os.path.basename(f)[:-5].ljust(14), l["ms_per_step"], pk)
    except Exception as e:
        print(f, "ERR", e)
PY
