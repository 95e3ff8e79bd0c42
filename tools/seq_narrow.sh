#!/bin/bash
# the narrow rotation-chain kernel (three peaks per lane) beside the fused kernel: schedules and wave counts
out=${1:-gpurun_out/r02/narrow}
mkdir -p $out
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --sample-every 4 > $out/$tag.json 2> $out/$tag.err || echo "$tag failed"; }
run base
run narrow3 AUDIOMOD_PV_SEQ_NARROW=1
run narrow2_w14 AUDIOMOD_PV_SEQ_NARROW=1 AUDIOMOD_PV_THREE_STAGE=0 AUDIOMOD_PV_CHAIN_WAVES=14
run narrow2_w15 AUDIOMOD_PV_SEQ_NARROW=1 AUDIOMOD_PV_THREE_STAGE=0 AUDIOMOD_PV_CHAIN_WAVES=15
run narrow2_w12 AUDIOMOD_PV_SEQ_NARROW=1 AUDIOMOD_PV_THREE_STAGE=0 AUDIOMOD_PV_CHAIN_WAVES=12
run narrow2_w16 AUDIOMOD_PV_SEQ_NARROW=1 AUDIOMOD_PV_THREE_STAGE=0 AUDIOMOD_PV_CHAIN_WAVES=16
python - <<'PY' $out
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        l = json.loads(open(f).read().strip().splitlines()[-1])
        pk = {k: v["avg_ms"] for k, v in l["roofline"]["per_kernel"].items()}
        print(os.path.basename(f)[:-5].ljust(14), l["ms_per_step"], l["verified"]["ok"], pk)
    except Exception as e:
        print(f, "ERR", e)
PY
