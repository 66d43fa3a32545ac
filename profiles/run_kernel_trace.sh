#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command; writes the per-kernel summary CSV.
# Usage (on the GPU box, from the repo root): bash profiles/run_kernel_trace.sh <tag>
set -u
TAG=${1:-trace}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p $ROOT/gpurun_out/$TAG
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/$TAG -- python3 $ROOT/bench.py --steps 5 --warmup 2 --cpu-frames 0 --no-micro --steady-steps 0 > $ROOT/gpurun_out/$TAG/bench.json 2> $ROOT/gpurun_out/$TAG/bench.err || { tail -5 $ROOT/gpurun_out/$TAG/bench.err; exit 1; }
find $ROOT/gpurun_out/$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $ROOT/gpurun_out/$TAG/kernel_stats.csv
cat $ROOT/gpurun_out/$TAG/kernel_stats.csv
