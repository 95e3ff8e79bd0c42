#!/bin/bash
# Elimination passes over the fused synthesis + overlap-add kernel (results are numerically meaningless when a
# diag bit is set; timings are not).  usage: tools/chain_diag.sh <outdir>
out=${1:-gpurun_out/r02/diag}
mkdir -p $out
run() { # tag, env...
    tag=$1; shift
    env "$@" AUDIOMOD_PV_PIPELINE=0 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --sample-every 4 > $out/$tag.json 2> $out/$tag.err || echo "$tag failed"
}
run base
run nores AUDIOMOD_PV_CHAIN_DIAG=1
run noadd AUDIOMOD_PV_CHAIN_DIAG=2
run noturn AUDIOMOD_PV_CHAIN_DIAG=4
run nosynth AUDIOMOD_PV_CHAIN_DIAG=8
run nosynth_nores AUDIOMOD_PV_CHAIN_DIAG=9
run onlysynth AUDIOMOD_PV_CHAIN_DIAG=7
run w8 AUDIOMOD_PV_CHAIN_WAVES=8
run w10 AUDIOMOD_PV_CHAIN_WAVES=10
run w12 AUDIOMOD_PV_CHAIN_WAVES=12
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
