#!/bin/bash
# Deeper SQ counter passes for the team kernel (one small group per pass; a group that the device
# rejects is skipped).  usage on the GPU box: bash tools/pmc_deep.sh <tag>
TAG=${1:-deep}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/rocprof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary ${BENCH_ARGS}"
run() { rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -- python3 $REPO/bench.py $ARGS > $OUT/$1.log 2>&1 || echo "pass $1 failed"; }
run pmc_sq  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
run pmc_sq2 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"
run pmc_sq3 "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL"
run pmc_sq4 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT"
run pmc_sq5 "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_INSTS_BRANCH SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_ADD_F64"
python3 $REPO/tools/summarize_pmc.py $OUT $OUT/summary.json
