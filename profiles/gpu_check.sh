#!/bin/bash
# GPU box: parity tests, then bench, then (optionally) PMC counters.  Usage: bash profiles/gpu_check.sh <tag> [pmc]
TAG=${1:-run}
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/test_$TAG.log 2>&1
rc=$?; tail -4 gpurun_out/test_$TAG.log
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" gpurun_out/test_$TAG.log | head -20; exit $rc; }
timeout -k 10 300 python -u bench.py --steps 5 --warmup 2 --cpu-frames 0 --steady-steps 0 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || { tail -5 gpurun_out/bench_$TAG.err; exit 1; }
python3 - <<PY
import json
d=json.load(open("gpurun_out/bench_$TAG.json"))
print("fps %.1f  ms/step %.3f" % (d["value"], d["ms_per_step"]))
print({k:(round(v["ms_per_step"],3), v["launches_per_step"]) for k,v in d["kernels"].items()})
PY
if [ "${2:-}" = "pmc" ]; then
  bash profiles/run_pmc.sh pmc_$TAG > /dev/null && python3 profiles/summarize_pmc.py gpurun_out/pmc_$TAG > gpurun_out/pmc_$TAG/summary.json
  python3 - <<PY
import json
d=json.load(open("gpurun_out/pmc_$TAG/summary.json"))
for k,s in d.items():
    w=s.get("SQ_WAVES",0)
    if w < 1000: continue
    print(k.split("::")[-1][:28], "waves/disp %.0f VALU %.0f SALU %.0f VMEM %.1f L1acc %.0f wavecyc %.0f" % (w/s["dispatches"], s["SQ_INSTS_VALU"]/w, s["SQ_INSTS_SALU"]/w, s["SQ_INSTS_VMEM_RD"]/w, s["TCP_TOTAL_CACHE_ACCESSES_sum"]/w, 4*s["SQ_WAVE_CYCLES"]/w))
PY
fi
