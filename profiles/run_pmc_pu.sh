#!/bin/bash
# rocprofv3 counter passes for the full-resolution PU search benchmark (benchmarks/pu_me_bench.py).
# Usage (on the GPU box, from the repo root): bash profiles/run_pmc_pu.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pu_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/benchmarks/pu_me_bench.py --cpu-pus 0 --reps 1 > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; return 1; }
  echo "pass $name ok"
}
pass sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY &&
pass ta TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE &&
python3 $ROOT/profiles/summarize_pmc.py $OUT > $ROOT/gpurun_out/pu_pmc_summary.json && echo "summary ok"
