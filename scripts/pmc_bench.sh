#!/bin/bash
# Runs on the GPU box (through gpurun): separate rocprofv3 --pmc passes over bench.py's DEFAULT workload (the command
# whose JSON line the driver records; the CPU baseline leg is skipped, it launches no kernel), plus the FETCH_SIZE
# calibration microbenchmark.  Usage: scripts/pmc_bench.sh <tag> [extra bench args]
# Output: gpurun_out/pmc_<tag>/{<pass>/...csv, summary.txt, pmc_bench.json, fetch_calib.txt}
set -u
TAG=${1:-x}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-other-workloads $*"
run() { # name, counters
  timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/$1 -- python3 $ROOT/bench.py $ARGS > $OUT/$1.log 2>&1 || echo "pass $1 failed"
}
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run l2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
run sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
run grbm "GRBM_GUI_ACTIVE"
# the hardware's VALU class counters: the DYNAMIC instruction mix of the same launches (class membership of every opcode:
# scripts/valu_class_calib.sh -> profiles/r05_valu_class_calib.txt; issue cost inside a class: lib/valu_mix.json)
run cls "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
# kernel-trace only (no counters): the un-serialised durations rocprofv3 sees for the same command
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1 || echo "trace pass failed"
# FETCH_SIZE calibration on the traversal's access pattern
hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib $ROOT/scripts/microbench/fetch_calib.hip > $OUT/fetch_calib_build.log 2>&1
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calib -- /tmp/fetch_calib > $OUT/fetch_calib.txt 2>&1 || echo "calib failed"
timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/calibw -- /tmp/fetch_calib > $OUT/fetch_calibw.txt 2>&1 || echo "calibw failed"
python3 $ROOT/scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
# the raw per-dispatch CSVs are tens of MB (gpurun brings back at most 64 MiB): keep the summary, the JSON and the
# kernel-stats table
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
tail -n 60 $OUT/summary.txt
