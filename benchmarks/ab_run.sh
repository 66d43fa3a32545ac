#!/bin/bash
# A/B of two builds of the library (ab/libA.so, ab/libB.so: copies of x265_amd/libx265gpu.so made by hand, not tracked) on one box: bash benchmarks/ab_run.sh [bench args]
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
for v in ${VARIANTS:-A B}; do
  cp ab/lib$v.so x265_amd/libx265gpu.so
  [ -f ab/host$v.so ] && cp ab/host$v.so x265_amd/libx265gpu_host.so
  timeout -k 10 150 python bench.py --steps 40 --warmup 3 --cpu-frames 0 --no-micro --steady-steps 0 --in-flight ${INFLIGHT:-1} "$@" > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { echo "$v failed"; tail -3 gpurun_out/ab_$v.err; exit 1; }
  python -c "import json;d=json.load(open('gpurun_out/ab_$v.json'));print('$v', round(d['value']), round(d['ms_per_step'],3), round(d['kernels']['search']['ms_per_step'],3), round(d['kernels']['cost']['ms_per_step'],3), round(d['kernels']['intra']['ms_per_step'],3), d.get('steady_state',{}).get('ms_per_slicetypeDecide'))"
done
done
