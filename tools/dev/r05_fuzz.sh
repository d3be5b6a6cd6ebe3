#!/bin/bash
# round 5: the three parity campaigns of rounds 3-4 re-run on the build that carries its slacks (5160 draws: seeds 0-459, 1000-2199, 3000-6499),
# in chunks that fit one GPU call each:  gpurun -- bash tools/dev/r05_fuzz.sh <a|b|c|d>
# (i), (j): last campaigns of the round on the final tree, all kinds, seeds no build has seen
# (h): horizons 160 .. 600 on the random vehicles (the block-parallel tail; batches <= 65)
# (g): the two other warm starts of the fuzzer - an arbitrary, dynamically inconsistent linearisation trajectory; each side's own result - on new seeds
# (e), (f): late in the round, two more campaigns on seeds no build has seen (10000-13599)
# plus (d) a campaign on seeds no build has seen (7000-7999) and the short fuzzers of the other kinds
mkdir -p gpurun_out
case "$1" in
  a) timeout -k 10 500 python tools/dev/fuzz_parity.py 460 0 > gpurun_out/r05_fuzz_parity_draws_0_459.txt 2>&1; tail -1 gpurun_out/r05_fuzz_parity_draws_0_459.txt
     timeout -k 10 560 python tools/dev/fuzz_parity.py 1200 1000 > gpurun_out/r05_fuzz_parity_draws_1000_2199.txt 2>&1; tail -1 gpurun_out/r05_fuzz_parity_draws_1000_2199.txt ;;
  b) timeout -k 10 1060 python tools/dev/fuzz_parity.py 1750 3000 > gpurun_out/r05_fuzz_parity_draws_3000_4749.txt 2>&1; tail -1 gpurun_out/r05_fuzz_parity_draws_3000_4749.txt ;;
  c) timeout -k 10 1060 python tools/dev/fuzz_parity.py 1750 4750 > gpurun_out/r05_fuzz_parity_draws_4750_6499.txt 2>&1; tail -1 gpurun_out/r05_fuzz_parity_draws_4750_6499.txt ;;
  d) timeout -k 10 500 python tools/dev/fuzz_parity.py 1000 7000 > gpurun_out/r05_fuzz_parity_draws_7000_7999.txt 2>&1; tail -1 gpurun_out/r05_fuzz_parity_draws_7000_7999.txt
     timeout -k 10 200 python tools/dev/fuzz_perm.py 200 800 > gpurun_out/r05_fuzz_permutation_draws_800_999.txt 2>&1; tail -1 gpurun_out/r05_fuzz_permutation_draws_800_999.txt
     timeout -k 10 150 python tools/dev/fuzz_f32io.py 150 500 > gpurun_out/r05_fuzz_f32io_draws_500_649.txt 2>&1; tail -1 gpurun_out/r05_fuzz_f32io_draws_500_649.txt
     timeout -k 10 120 python tools/dev/fuzz_nan.py 150 500 > gpurun_out/r05_fuzz_nan_isolation_draws_500_649.txt 2>&1; tail -1 gpurun_out/r05_fuzz_nan_isolation_draws_500_649.txt
     timeout -k 10 100 python tools/dev/fuzz_parity.py 120 8000 --lane > gpurun_out/r05_fuzz_lane_kernel_draws_8000_8119.txt 2>&1; tail -1 gpurun_out/r05_fuzz_lane_kernel_draws_8000_8119.txt
     timeout -k 10 100 python tools/dev/fuzz_parity.py 60 8200 --cond > gpurun_out/r05_fuzz_condensed_kernel_draws_8200_8259.txt 2>&1; tail -1 gpurun_out/r05_fuzz_condensed_kernel_draws_8200_8259.txt ;;
  e) timeout -k 10 1080 python tools/dev/fuzz_parity.py 1800 10000 > gpurun_out/r05_fuzz_parity_draws_10000_11799.txt 2>&1; tail -1 gpurun_out/r05_fuzz_parity_draws_10000_11799.txt ;;
  f) timeout -k 10 1080 python tools/dev/fuzz_parity.py 1800 11800 > gpurun_out/r05_fuzz_parity_draws_11800_13599.txt 2>&1; tail -1 gpurun_out/r05_fuzz_parity_draws_11800_13599.txt ;;
  g) timeout -k 10 520 python tools/dev/fuzz_parity.py 700 20000 --random-init > gpurun_out/r05_fuzz_parity_random_init_draws_20000_20699.txt 2>&1; tail -1 gpurun_out/r05_fuzz_parity_random_init_draws_20000_20699.txt
     timeout -k 10 520 python tools/dev/fuzz_parity.py 700 21000 --chained > gpurun_out/r05_fuzz_parity_chained_draws_21000_21699.txt 2>&1; tail -1 gpurun_out/r05_fuzz_parity_chained_draws_21000_21699.txt ;;
  h) timeout -k 10 1080 python tools/dev/fuzz_parity.py 180 30000 --long > gpurun_out/r05_fuzz_parity_long_horizon_draws_30000_30179.txt 2>&1; tail -1 gpurun_out/r05_fuzz_parity_long_horizon_draws_30000_30179.txt ;;
  i) timeout -k 10 1080 python tools/dev/fuzz_parity.py 2500 40000 > gpurun_out/r05_fuzz_parity_draws_40000_42499.txt 2>&1; tail -1 gpurun_out/r05_fuzz_parity_draws_40000_42499.txt ;;
  j) timeout -k 10 300 python tools/dev/fuzz_parity.py 200 31000 --long > gpurun_out/r05_fuzz_parity_long_horizon_draws_31000_31199.txt 2>&1; tail -1 gpurun_out/r05_fuzz_parity_long_horizon_draws_31000_31199.txt
     timeout -k 10 200 python tools/dev/fuzz_perm.py 200 2000 > gpurun_out/r05_fuzz_permutation_draws_2000_2199.txt 2>&1; tail -1 gpurun_out/r05_fuzz_permutation_draws_2000_2199.txt
     timeout -k 10 150 python tools/dev/fuzz_f32io.py 150 2000 > gpurun_out/r05_fuzz_f32io_draws_2000_2149.txt 2>&1; tail -1 gpurun_out/r05_fuzz_f32io_draws_2000_2149.txt
     timeout -k 10 120 python tools/dev/fuzz_nan.py 150 2000 > gpurun_out/r05_fuzz_nan_isolation_draws_2000_2149.txt 2>&1; tail -1 gpurun_out/r05_fuzz_nan_isolation_draws_2000_2149.txt
     timeout -k 10 100 python tools/dev/fuzz_parity.py 120 9000 --lane > gpurun_out/r05_fuzz_lane_kernel_draws_9000_9119.txt 2>&1; tail -1 gpurun_out/r05_fuzz_lane_kernel_draws_9000_9119.txt
     timeout -k 10 100 python tools/dev/fuzz_parity.py 60 9200 --cond > gpurun_out/r05_fuzz_condensed_kernel_draws_9200_9259.txt 2>&1; tail -1 gpurun_out/r05_fuzz_condensed_kernel_draws_9200_9259.txt ;;
esac
