#!/bin/bash
# wider fuzz campaigns on the final build of round 4 (new seed ranges; the in-place continuation is the default schedule in all of them)
mkdir -p gpurun_out
timeout -k 10 900 python tools/dev/fuzz_parity.py 1200 1000 > gpurun_out/r04u_fuzz_parity_draws_1000_2199.txt 2>&1; tail -1 gpurun_out/r04u_fuzz_parity_draws_1000_2199.txt
timeout -k 10 600 python tools/dev/fuzz_perm.py 300 500 > gpurun_out/r04u_fuzz_permutation_draws_500_799.txt 2>&1; tail -1 gpurun_out/r04u_fuzz_permutation_draws_500_799.txt
timeout -k 10 600 python tools/dev/fuzz_f32io.py 200 300 > gpurun_out/r04u_fuzz_f32io_draws_300_499.txt 2>&1; tail -1 gpurun_out/r04u_fuzz_f32io_draws_300_499.txt
timeout -k 10 600 python tools/dev/fuzz_rollout.py 60 100 > gpurun_out/r04u_fuzz_rollout_draws_100_159.txt 2>&1; tail -1 gpurun_out/r04u_fuzz_rollout_draws_100_159.txt
timeout -k 10 600 python tools/dev/fuzz_nan.py 200 300 > gpurun_out/r04u_fuzz_nan_isolation_draws_300_499.txt 2>&1; tail -1 gpurun_out/r04u_fuzz_nan_isolation_draws_300_499.txt
timeout -k 10 600 python tools/dev/fuzz_facade.py 100 200 > gpurun_out/r04u_fuzz_facade_draws_200_299.txt 2>&1; tail -1 gpurun_out/r04u_fuzz_facade_draws_200_299.txt
