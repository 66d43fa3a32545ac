#!/bin/bash
# One configuration's evidence set: rocprofv3 kernel stats and four counter passes of the bench's segment workload.
# Usage (on the GPU box, from the repo root): bash profiles/collect_config.sh <key> [bench.py arguments]
#   e.g. bash profiles/collect_config.sh 3840x2160_10bit_hme --width 3840 --height 2160 --depth 10 --hme
# Leaves gpurun_out/<key>/kernel_stats.csv, gpurun_out/<key>/pmc.json and one log per pass; copy the two files to
# profiles/r<round>_kernel_stats_<key>.csv and profiles/pmc_<key>.json (the name bench.py looks for).
# One lookahead instance (--in-flight 1), so that no kernel of another instance runs beside the one being measured.
# The bench runs without its child processes (microbenchmark, CPU baseline pool) and without the steady workload:
# a program under rocprofv3 must not start other programs, and the steady loop launches the same kernels with
# other job counts, which would blur the per-launch averages.
set -u
KEY=${1:?key}
shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$KEY
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--cpu-frames 0 --no-micro --no-other-configs --steady-steps 0 --in-flight 1 --reps 1 $*"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 5 --warmup 2 $ARGS > $OUT/bench_traced.json 2> $OUT/trace.log || { echo "kernel trace failed"; tail -5 $OUT/trace.log; exit 1; }
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
echo "trace ok"
pass() {
  name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc/$name -- python3 $ROOT/bench.py --steps 1 --warmup 1 $ARGS > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; return 1; }
  echo "pass $name ok"
}
pass sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY &&
pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY &&
pass fetch FETCH_SIZE TCC_HIT_sum &&
pass write WRITE_SIZE TCC_MISS_sum &&
python3 $ROOT/profiles/summarize_pmc.py $OUT/pmc > $OUT/pmc.json && echo "summary ok"
