"""GPU: bench.py end to end at a small size - the one-rank RCCL rehearsal of the multi-GPU path and the report's shape."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--width", "640", "--height", "368", "--frames", "12", "--steps", "2", "--warmup", "1", "--reps", "2",
         "--steady-steps", "0", "--no-micro", "--no-other-configs"]


def _bench(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + extra, capture_output=True, timeout=280, env=env)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


@pytest.mark.timeout(300)
def test_force_dist_gathers_this_ranks_records():
    """`--force-dist`: a process group of one over RCCL; the records rank 0 finds at its place in the gathered set are
    the ones its own host layer collected for the step (and they are not empty)."""
    rec = _bench(["--force-dist", "--cpu-frames", "0"])
    assert rec["gathered_records_match_local"] is True
    assert rec["n_gpus"] == 1 and rec["value"] > 0
    assert "RCCL" in rec["config"]["sharding"]


@pytest.mark.timeout(300)
def test_report_carries_repetitions_host_cpu_and_algorithmic_counts():
    rec = _bench(["--cpu-frames", "12", "--cpu-cores", "1"])
    reps = rec["repetitions"]
    assert reps["count"] == 2 and len(reps["ms_per_step"]) == 2
    assert reps["ms_per_step_min"] <= rec["ms_per_step"] <= reps["ms_per_step_max"]
    assert rec["host_cpu"]["host_cpu_seconds_per_step"] > 0
    assert rec["cpu_baseline"]["value"] > 0 and rec["cpu_baseline"]["kind"] == "port"
    search = rec["kernels"]["search"]
    # hexagon search: a couple of dozen SADs and about ten SATDs per block (oracle's executed compares)
    assert 10 < search["sads_per_block_search"] < 60 and 5 < search["satds_per_block_search"] < 20
    assert 0 < search["algorithmic_frac"] < 1
