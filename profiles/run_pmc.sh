#!/bin/bash
# Collects rocprofv3 PMC counters for the bench workload, one pass per run.
# Usage (on the GPU box, from the repo root): bash profiles/run_pmc.sh <outdir-under-gpurun_out>
set -u
OUT=${1:-pmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-micro --steady-steps 0"
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $ROOT/gpurun_out/$OUT/$name -- $BENCH > $ROOT/gpurun_out/$OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $ROOT/gpurun_out/$OUT/$name.log; return 1; }
  echo "pass $name ok"
}
mkdir -p $ROOT/gpurun_out/$OUT
pass sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY &&
pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY &&
pass fetch FETCH_SIZE TCC_HIT_sum &&
pass write WRITE_SIZE TCC_MISS_sum
