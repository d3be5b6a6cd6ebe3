#!/bin/bash
# The large-sample parity evidence of a round in one GPU call (outputs under gpurun_out/; copy what is to be judged into profiles/):
#   stress sweep against the oracle (default path, plain interior point), 460-draw fuzz over what `reconfigure` can change,
#   permutation fuzz (results must not depend on wave-mates) with the block-parallel tail forced on at short horizons.
TAG=${1:-r05}
mkdir -p gpurun_out
python tools/stress_parity.py 8192 > gpurun_out/${TAG}_stress_parity.txt 2>&1; tail -2 gpurun_out/${TAG}_stress_parity.txt
python tools/stress_parity.py 4096 --no-polish > gpurun_out/${TAG}_stress_parity_plain_ipm.txt 2>&1; tail -2 gpurun_out/${TAG}_stress_parity_plain_ipm.txt
python tools/dev/fuzz_parity.py 460 0 > gpurun_out/${TAG}_fuzz_parity_draws_0_459.txt 2>&1; tail -1 gpurun_out/${TAG}_fuzz_parity_draws_0_459.txt
NMPC_BLOCK_TAIL=1 NMPC_BLOCK_J=4 python tools/dev/fuzz_perm.py 100 0 > gpurun_out/${TAG}_fuzz_permutation_tail_forced.txt 2>&1; tail -1 gpurun_out/${TAG}_fuzz_permutation_tail_forced.txt
