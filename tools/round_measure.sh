#!/bin/bash
# The round's measurement set, first gpurun call (~6 min of box time): GPU test suite, tools/profile_round.sh (bench line,
# rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU passes -> traffic.json, valu.json), every BASELINE config,
# the batch-size sweep, the drop-in single-stream bench, the host-memory bench, PV_ARITH_EXACT.  Outputs: gpurun_out/r03/z_*;
# copy z_traffic.json / z_valu.json to profiles/ (their source hash must match the committed kernels) and the rest to profiles/r03/.
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q > gpurun_out/r03/z_gpu_tests.txt 2>&1; echo "suite rc $?"; tail -3 gpurun_out/r03/z_gpu_tests.txt
bash tools/profile_round.sh r03/z
bash tools/bench_all.sh gpurun_out/r03/z_bench_all_configs.txt > /dev/null; cat gpurun_out/r03/z_bench_all_configs.txt | cut -c1-170
bash tools/streams_sweep.sh gpurun_out/r03/z_streams_sweep.txt 4 8 20 32 64 97 128 160 192 > /dev/null; cut -c1-110 gpurun_out/r03/z_streams_sweep.txt
./audiomod_amd/lib/stream_bench 60 480 2 > gpurun_out/r03/z_dropin_stream_bench.json; ./audiomod_amd/lib/stream_bench 60 4800 2 >> gpurun_out/r03/z_dropin_stream_bench.json; cat gpurun_out/r03/z_dropin_stream_bench.json
timeout -k 10 300 python bench.py --host-io --no-cpu-baseline > gpurun_out/r03/z_bench_hostio.json 2> gpurun_out/r03/z_bench_hostio.err; python -c "
import json; l=json.loads(open('gpurun_out/r03/z_bench_hostio.json').read().strip().splitlines()[-1]); print(l['host_io'])"
AUDIOMOD_PV_EXACT=1 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03/z_bench_exact.json 2> gpurun_out/r03/z_bench_exact.err; python -c "
import json; l=json.loads(open('gpurun_out/r03/z_bench_exact.json').read().strip().splitlines()[-1]); print('exact', l['value'], l['ms_per_step'], l['verified']['max_rms_vs_oracle'], l['verified']['batch_checksum_sha256'][:12])"
