#!/bin/bash
# Second gpurun call, after profiles/traffic.json and valu.json have been refreshed: the bench line in the driver's form (now
# quoting the counters), the parity suite on the -DPV_POISON build (tools/build_variant.sh poison -DPV_POISON first), rocprofv3
# kernel stats of cfg3 and cfg4, the FFT-size sweep.
mkdir -p gpurun_out/r03
timeout -k 10 300 python bench.py --gpus 1 --steps 10 --warmup 3 > gpurun_out/r03/zz_bench_driver_form.json 2> gpurun_out/r03/zz_bench.err; python - <<'PY'
import json
l=json.loads(open('gpurun_out/r03/zz_bench_driver_form.json').read().strip().splitlines()[-1])
r=l['roofline']; print(l['value'], l['ms_per_step'], {k:r[k] for k in ('bound','kernel','achieved','frac','traffic','hbm_real_GBps','pipeline_frac')})
PY
AUDIOMOD_PV_LIB=$PWD/audiomod_amd/lib/diag/poison/libaudiomod_pv.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r03/z_poison_build_tests.txt 2>&1; echo "poison rc $?"; tail -2 gpurun_out/r03/z_poison_build_tests.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile_config.sh cfg3 r03/z && bash tools/profile_config.sh cfg4_formant+7 r03/z && bash tools/profile_config.sh cfg4_formant-7 r03/z
timeout -k 10 300 python tools/fft_size_bench.py 256 512 1024 2048 4096 8192 > gpurun_out/r03/z_fft_sizes.txt 2>gpurun_out/r03/z_fft_sizes.err; cat gpurun_out/r03/z_fft_sizes.txt
