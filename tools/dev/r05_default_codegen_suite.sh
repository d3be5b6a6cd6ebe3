#!/bin/bash
# The GPU suite on a `make HIPCC_EXPECT=9.9` library (codegen gate fails on the hipcc version: the default-codegen twins become what runs by
# default), built on the box in its scratch copy of the tree - the shipped library stays the flag build.
set -o pipefail
mkdir -p gpurun_out
( cd rotors_mpc_controller_amd/csrc && make -j8 HIPCC_EXPECT=9.9 > /tmp/make_default.log 2>&1 ) || { tail -20 /tmp/make_default.log; exit 1; }
python -c "from rotors_mpc_controller_amd import _lib; print(_lib.load().nmpc_version().decode())" > gpurun_out/r05z_gpu_tests_default_codegen.log 2>&1
timeout -k 10 600 python -m pytest tests -m gpu -q >> gpurun_out/r05z_gpu_tests_default_codegen.log 2>&1
head -1 gpurun_out/r05z_gpu_tests_default_codegen.log; tail -3 gpurun_out/r05z_gpu_tests_default_codegen.log
