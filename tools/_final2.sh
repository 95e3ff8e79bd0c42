mkdir -p gpurun_out/r03
bash tools/profile_round.sh r03/z
python -m pytest tests -m gpu -x -q > gpurun_out/r03/z_gpu_tests.txt 2>&1; echo "suite rc $?"; tail -3 gpurun_out/r03/z_gpu_tests.txt
