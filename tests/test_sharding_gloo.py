"""CPU, world_size 2 over gloo: frame-segment sharding and the cost gather.

Each rank runs the request pattern on its own frame segment (with the oracle
standing in for the device, since this box has no GPU) and the per-request
scalars are gathered; rank 0 must see exactly what a single process computes
for both segments.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, TOTAL, BFRAMES = 192, 128, 13, 3


def _segment_costs(start, end):
    sys.path.insert(0, ROOT)
    import bench
    from oracle import orc
    from x265_amd import synth
    clip = synth.luma_clip(W, H, TOTAL, 8, seed=7)[start:end]
    param = orc.make_param(W, H, bframes=BFRAMES)
    ref = orc.Lookahead(8, param)
    for f in clip:
        ref.add_frame(f)
    b1, b2 = bench.batch_requests(end - start, BFRAMES)
    rows = []
    for (p0, p1, b) in b1 + b2:
        ref.frame_cost(p0, p1, b)
        fr = ref.frames[b]
        rows.append((fr.i64("costEst", b - p0, p1 - b), fr.i64("costEstAq", b - p0, p1 - b),
                     fr.i64("intraMbs", b - p0)))
    ref.close()
    return np.array(rows, np.int64).reshape(-1, 3)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from x265_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    start, end = sharding.segment_bounds(TOTAL, world, rank)
    local = _segment_costs(start, end)
    parts = sharding.gather_costs(local)
    if rank == 0:
        q.put([p.tolist() for p in parts])
    dist.barrier()
    dist.destroy_process_group()


def _window_costs(requests, lo, hi):
    """The oracle on frames [lo, hi) of the clip, running `requests` (indices relative to lo)."""
    sys.path.insert(0, ROOT)
    from oracle import orc
    from x265_amd import synth
    clip = synth.luma_clip(W, H, TOTAL, 8, seed=7)[lo:hi]
    ref = orc.Lookahead(8, orc.make_param(W, H, bframes=BFRAMES))
    for f in clip:
        ref.add_frame(f)
    rows = []
    for (p0, p1, b) in requests:
        ref.frame_cost(p0, p1, b)
        fr = ref.frames[b]
        rows.append((fr.i64("costEst", b - p0, p1 - b), fr.i64("costEstAq", b - p0, p1 - b), fr.i64("intraMbs", b - p0)))
    ref.close()
    return np.array(rows, np.int64).reshape(-1, 3)


def _halo_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import bench
    import torch.distributed as dist
    from x265_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b1, b2 = bench.batch_requests(TOTAL, BFRAMES)
    start, end = sharding.segment_bounds(TOTAL, world, rank)
    lo, hi = sharding.halo_bounds(TOTAL, world, rank, BFRAMES + 1)
    pos, mine = sharding.shard_requests(b1 + b2, start, end, origin=lo)
    parts = sharding.gather_costs(_window_costs(mine, lo, hi))
    if rank == 0:
        q.put([p.tolist() for p in parts])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_one_window_sharded_with_halo_matches_single_process(oracle_built):
    """One lookahead window split by frame b across two ranks, each holding its segment plus a halo of
    bframes + 1 frames: the gathered per-request scalars, put back in request order, are those of one process."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_halo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    parts = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    import bench
    from x265_amd import sharding
    b1, b2 = bench.batch_requests(TOTAL, BFRAMES)
    want = _window_costs(b1 + b2, 0, TOTAL)
    merged = np.full_like(want, -1)
    for r in range(2):
        s, e = sharding.segment_bounds(TOTAL, 2, r)
        pos, _ = sharding.shard_requests(b1 + b2, s, e)
        merged[pos] = np.array(parts[r], np.int64).reshape(-1, 3)
    np.testing.assert_array_equal(merged, want)


def test_segment_bounds_cover_everything():
    sys.path.insert(0, ROOT)
    from x265_amd import sharding
    for total in (1, 7, 8, 13, 60):
        for world in (1, 2, 4, 8):
            edges = [sharding.segment_bounds(total, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            for a, b in zip(edges, edges[1:]):
                assert a[1] == b[0]
            sizes = [e - s for s, e in edges]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_gather_matches_single_process(oracle_built):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    parts = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from x265_amd import sharding
    assert len(parts) == 2
    for r in range(2):
        s, e = sharding.segment_bounds(TOTAL, 2, r)
        np.testing.assert_array_equal(np.array(parts[r], np.int64).reshape(-1, 3), _segment_costs(s, e))


def _launch(argv, gpus, timeout):
    """bench.launch_ranks in a process of its own (it passes rank 0's output on to its standard output)."""
    import subprocess
    code = ("import argparse, sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.launch_ranks(argparse.Namespace(gpus=%d, rank_timeout=%d), %r, script=%r))"
            % (ROOT, gpus, timeout, argv, os.path.join(ROOT, "tests", "standin_rank.py")))
    return subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=240,
                          env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})


@pytest.mark.timeout(300)
def test_bench_launcher_starts_two_ranks(oracle_built):
    """The launcher behind `bench.py --gpus 2` (outside torchrun) starts two rank processes itself and passes on
    rank 0's one JSON line: here with gloo and the oracle standing in for the device (tests/standin_rank.py)."""
    import json
    out = _launch(["--width", str(W), "--height", str(H), "--frames", "8", "--bframes", str(BFRAMES)], 2, 200)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    import bench
    b1, b2 = bench.batch_requests(8, BFRAMES)
    assert rec["gathered_requests"] == [len(b1) + len(b2)] * 2


@pytest.mark.timeout(120)
def test_bench_launcher_ends_a_stuck_rank():
    """A rank that never finishes: after the launcher's time limit every remaining rank is ended and the exit status
    is non-zero, instead of the caller waiting for its own limit."""
    import time
    t0 = time.monotonic()
    out = _launch(["--hang-rank", "1", "--width", str(W), "--height", str(H), "--frames", "4"], 2, 5)
    assert out.returncode != 0
    assert time.monotonic() - t0 < 90
    assert b"ending the remaining ranks" in out.stderr
