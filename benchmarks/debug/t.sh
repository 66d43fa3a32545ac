timeout -k 10 400 python -u -m pytest tests/test_gpu_lookahead.py tests/test_gpu_lds_poison.py -x -q --timeout 200 --timeout-method thread > gpurun_out/t_tab.log 2>&1; rc=$?; tail -n 2 gpurun_out/t_tab.log; [ $rc -eq 0 ] || exit $rc
X265GPU_FLOW_STAGED_JOBS=100000 timeout -k 10 400 python -u -m pytest tests/test_gpu_lookahead.py tests/test_gpu_lds_poison.py -x -q --timeout 200 --timeout-method thread > gpurun_out/t_tab2.log 2>&1; rc=$?; tail -n 2 gpurun_out/t_tab2.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --cpu-frames 0 --no-micro --no-other-configs > gpurun_out/b_tab.json 2>/dev/null || exit 1
python3 -c "
import json
d=json.load(open('gpurun_out/b_tab.json'))
print(d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d['kernels'].items()}, d['steady_state']['ms_per_slicetypeDecide'])"
