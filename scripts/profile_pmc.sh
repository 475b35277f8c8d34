#!/bin/bash
# Runs on the GPU box (through gpurun): separate rocprofv3 --pmc passes over one short bench run.
# Usage: scripts/profile_pmc.sh <tag> [bench args...]
# Output: gpurun_out/pmc_<tag>/<pass>/..._counter_collection.csv (+ kernel trace)
set -u
TAG=${1:-x}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-cpu-baseline $*"
run() { # name, counters
  timeout 180 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/$1 -- python3 $ROOT/bench.py $ARGS > $OUT/$1.log 2>&1 || echo "pass $1 failed"
}
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run l2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
run sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
run grbm "GRBM_GUI_ACTIVE"
python3 $ROOT/scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
