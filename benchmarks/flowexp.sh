# compares the search launch modes on the bench's segment and steady workloads (GPU box, from the repo root)
for mode in flow diag; do
  export X265GPU_SEARCH_MODE=$mode
  timeout -k 10 200 python bench.py --steps 8 --warmup 2 --cpu-frames 0 --no-micro --steady-steps 2 2>/dev/null > gpurun_out/exp_$mode.json
  python profiles/bench_line.py < gpurun_out/exp_$mode.json | sed "s/^/$mode /"
  python -c "import json;d=json.load(open('gpurun_out/exp_$mode.json'));print('  steady', d['steady_state']['ms_per_segment'], d['steady_state']['ms_per_slicetypeDecide'])"
done
