#!/bin/bash
# round 5, fourth dev call: (1) config 5 as G concurrent groups on G streams; (2) the GPU suite on a `make HIPCC_EXPECT=9.9` library
# (default-codegen twins as the default), built on the box in a scratch copy of csrc so the shipped library stays untouched
set -o pipefail
mkdir -p gpurun_out
python tools/dev/split_streams.py --groups 1 2 4 8 > gpurun_out/r05f_split_streams.txt 2>&1 && \
python tools/dev/split_streams.py --groups 1 2 4 --interleave >> gpurun_out/r05f_split_streams.txt 2>&1 && \
python tools/dev/split_streams.py --groups 1 2 4 --horizon 250 >> gpurun_out/r05f_split_streams.txt 2>&1
cat gpurun_out/r05f_split_streams.txt
cp rotors_mpc_controller_amd/librotors_nmpc_hip.so /tmp/flag.so
( cd rotors_mpc_controller_amd/csrc && make -j8 HIPCC_EXPECT=9.9 > /tmp/make_default.log 2>&1 ) || { tail -20 /tmp/make_default.log; exit 1; }
python -c "from rotors_mpc_controller_amd import _lib; print(_lib.load().nmpc_version().decode())" > gpurun_out/r05f_gpu_tests_default_codegen.log 2>&1
timeout -k 10 600 python -m pytest tests -m gpu -q >> gpurun_out/r05f_gpu_tests_default_codegen.log 2>&1
tail -5 gpurun_out/r05f_gpu_tests_default_codegen.log
