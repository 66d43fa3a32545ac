#!/bin/bash
# LDS counter pass of the bench's segment workload (one instance, no child processes), beside collect_config.sh's four.
# Usage (GPU box, repo root): bash profiles/collect_lds.sh <key> [bench.py arguments]; leaves gpurun_out/<key>/pmc_lds.json
set -u
KEY=${1:?key}
shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$KEY
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--cpu-frames 0 --no-micro --no-other-configs --steady-steps 0 --in-flight 1 --reps 1 $*"
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_]*LDS[A-Z_]*" | sort -u > $OUT/lds_counters_available.txt
cat $OUT/lds_counters_available.txt | tr '\n' ' '; echo
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_lds/lds -- python3 $ROOT/bench.py --steps 1 --warmup 1 $ARGS > $OUT/lds.log 2>&1 || { echo "pass lds failed"; tail -5 $OUT/lds.log; exit 1; }
python3 $ROOT/profiles/summarize_pmc.py $OUT/pmc_lds > $OUT/pmc_lds.json && echo "summary ok"
