#!/bin/bash
# rotation-chain elimination builds (tools/build_variant.sh seqN -DPV_EXP_SEQ=N; results invalid, timings not):
# 1 = no princarg_small, 2 = no princarg_f, 4 = no global store, 8 = no barrier
mkdir -p gpurun_out/r02
for L in ${SEQ_LIBS:-- seq16 seq20 seq31}; do
  EV=""; if [ "$L" != "-" ]; then EV="AUDIOMOD_PV_LIB=$PWD/audiomod_amd/lib/diag/$L/libaudiomod_pv.so"; fi
  env $EV timeout -k 10 200 python bench.py --no-cpu-baseline --no-verify --streams 20 --seconds 30 --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$L', d['ms_per_step'], d['roofline']['per_kernel']['pv_seq_kernel']['avg_ms'])" | tee -a gpurun_out/r02/seq_elim.txt
done
