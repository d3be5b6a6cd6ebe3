#!/bin/bash
# Captures the rocprofv3 evidence that bench.py's roofline object cites.  Run on the GPU box:
#   gpurun -- bash tools/rocprof_capture.sh <tag>
# Three separate passes (kernel trace; FETCH_SIZE; WRITE_SIZE) as MI355X_MICROARCH.md prescribes:
# the TCC counters do not fit one pass and --pmc must not be combined with other trace domains.
set -e
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/rocprof_$TAG
rm -rf $OUT      # (a merged-back copy of an earlier capture must not be summarised with this one)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-secondary ${BENCH_ARGS}"
# the kernel trace runs bench.py's DEFAULT step counts (a 10-step run ends while the GPU's clocks are still ramping: its kernel averages
# are ~10 % above the steady state the default bench line reports); the counter passes need few dispatches
ARGS_TRACE="--no-cpu-baseline --no-secondary ${BENCH_ARGS}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS_TRACE > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $ARGS > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $ARGS > $OUT/bench_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py $ARGS > $OUT/bench_sq.log 2>&1 || true
find $OUT -name "*.csv" | head -20
tail -1 $OUT/bench_trace.log | cut -c1-200
