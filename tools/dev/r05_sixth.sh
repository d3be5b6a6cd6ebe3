#!/bin/bash
# round 5, sixth dev call: the pivot range rule (PIVOT_MAX) - GPU suite, then the fuzz campaigns again on the new kernels, newest first
set -o pipefail
mkdir -p gpurun_out
python -c "from rotors_mpc_controller_amd import _lib; print(_lib.load().nmpc_version().decode())" > gpurun_out/r05j_gpu_tests.log 2>&1
timeout -k 10 600 python -m pytest tests -m gpu -q >> gpurun_out/r05j_gpu_tests.log 2>&1
tail -12 gpurun_out/r05j_gpu_tests.log
for part in "$@"; do bash tools/dev/r05_fuzz.sh $part || true; done
