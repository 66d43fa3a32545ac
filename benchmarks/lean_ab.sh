#!/bin/bash
# steady-state decision loop (bench.py's steady workload) with the lean flow kernel taken for launches of at most N searches
run() {
  tag=$1; shift
  env "$@" timeout -k 10 280 python bench.py --steps 10 --warmup 2 --reps 1 --cpu-frames 0 --no-micro --no-other-configs --steady-steps 5 > gpurun_out/lean_$tag.json 2> gpurun_out/lean_$tag.err || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/lean_$tag.json"))
s=d.get("steady_state")
print("$tag", round(d["value"]), round(d["ms_per_step"],3), s["ms_per_slicetypeDecide"], json.dumps(s["ms_per_slicetypeDecide_by_phase"]))
PY
}
run off X265GPU_SEARCH_LEAN_JOBS=0
run le64 X265GPU_SEARCH_LEAN_JOBS=64
run off2 X265GPU_SEARCH_LEAN_JOBS=0
run le64b X265GPU_SEARCH_LEAN_JOBS=64
