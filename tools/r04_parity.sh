#!/bin/bash
# Round 4 parity evidence on the final sources: stress sweeps, 460-draw fuzz, permutation fuzz with the tail forced on, tail sweep, kernel check under canaries.
mkdir -p gpurun_out
bash tools/evidence_parity.sh r04
NMPC_GUARD=64 timeout -k 10 600 python tools/dev/qp_kernel_check.py > gpurun_out/r04_qp_kernel_check_guarded.txt 2>&1; echo "qp_kernel_check rc $?"; tail -2 gpurun_out/r04_qp_kernel_check_guarded.txt
timeout -k 10 900 python tools/dev/tail_sweep_check.py > gpurun_out/r04_tail_sweep_check.txt 2>&1; echo "tail_sweep rc $?"; tail -2 gpurun_out/r04_tail_sweep_check.txt
NMPC_BLOCK_TAIL=1 NMPC_BLOCK_J=4 timeout -k 10 600 python tools/dev/fuzz_parity.py 180 0 > gpurun_out/r04_fuzz_tail_forced.txt 2>&1; tail -1 gpurun_out/r04_fuzz_tail_forced.txt
