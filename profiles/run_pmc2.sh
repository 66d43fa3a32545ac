#!/bin/bash
# Second set of SQ counters (issue-slot occupancy by instruction class, LDS stalls) for the bench workload.
# Usage (on the GPU box, from the repo root): bash profiles/run_pmc2.sh <outdir-under-gpurun_out>
set -u
OUT=${1:-pmc2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-micro --steady-steps 0"
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $ROOT/gpurun_out/$OUT/$name -- $BENCH > $ROOT/gpurun_out/$OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $ROOT/gpurun_out/$OUT/$name.log; return 1; }
  echo "pass $name ok"
}
mkdir -p $ROOT/gpurun_out/$OUT
pass sqa SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM &&
pass sqb SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE &&
pass sqc SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_LDS_UNALIGNED_STALL SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INSTS_SENDMSG
