"""Prints fps and per-phase ms from a bench.py JSON line on stdin (helper for GPU-box scripts)."""
import json
import sys

d = json.loads(sys.stdin.read())
print("%.1f fps  %.3f ms/step " % (d["value"], d["ms_per_step"]),
      {k: (round(v["ms_per_step"], 3), v["launches_per_step"]) for k, v in d["kernels"].items()})
