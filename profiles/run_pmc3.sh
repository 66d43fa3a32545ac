#!/bin/bash
# Extra rocprofv3 counter passes for the search kernel (what a wave waits for): LDS, scalar, branch and instruction-
# fetch activity.  Usage (on the GPU box, from the repo root): bash profiles/run_pmc3.sh [bench.py arguments]
# Each pass is a run of its own; leaves gpurun_out/pmc3/<pass>/ and one log per pass, then the summary pmc3.json.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--cpu-frames 0 --no-micro --steady-steps 0 --in-flight 1 $*"
pass() {
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc/$name -- python3 $ROOT/bench.py --steps 1 --warmup 1 $ARGS > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; return 1; }
  echo "pass $name ok"
}
pass lds SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_BRANCH
pass act SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_IFETCH SQ_INSTS_VALU
python3 $ROOT/profiles/summarize_pmc.py $OUT/pmc > $OUT/pmc3.json && echo "summary ok"
