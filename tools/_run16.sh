mkdir -p gpurun_out/r03
for seed in 101 102 103 104; do
  timeout -k 10 250 python tests/sweeps/fuzz_parity.py 400 $seed 200 > gpurun_out/r03/fuzz_$seed.txt 2>&1; echo "seed $seed rc $?"; tail -2 gpurun_out/r03/fuzz_$seed.txt
done
for seed in 105 106; do
  AUDIOMOD_PV_EXACT=1 timeout -k 10 250 python tests/sweeps/fuzz_parity.py 400 $seed 200 > gpurun_out/r03/fuzz_exact_$seed.txt 2>&1; echo "exact seed $seed rc $?"; tail -2 gpurun_out/r03/fuzz_exact_$seed.txt
done
