"""CPU rehearsal of one rank of `bench.py --gpus N`: gloo, and the oracle in the device's place.

TEST SCAFFOLDING, not a measurement and not part of the product path: started by tests/test_sharding_gloo.py through
bench.launch_ranks (the launcher the GPU run uses) so that the N > 1 plumbing - rank environment, the gather of the
per-request cost scalars, the one JSON line on rank 0 - runs on a machine without a GPU.  `--hang-rank R` makes
rank R sleep instead (the launcher's timeout test).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=192)
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--bframes", type=int, default=3)
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--hang-rank", type=int, default=-1)
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if rank == args.hang_rank:
        time.sleep(3600)
    # libraries under the process group print banners on stdout; fd 1 is kept for the one JSON line
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    import bench
    from oracle import orc
    from x265_amd import sharding
    dist.init_process_group(args.backend, rank=rank, world_size=world)
    clip, _ = bench.make_clip(args.width, args.height, args.frames, args.depth, 1234 + rank)
    look = orc.Lookahead(args.depth, orc.make_param(args.width, args.height, bframes=args.bframes))
    for f in clip:
        look.add_frame(f)
    b1, b2 = bench.batch_requests(args.frames, args.bframes)
    t0 = time.perf_counter()
    rows = []
    for (p0, p1, b) in b1 + b2:
        look.frame_cost(p0, p1, b)
        fr = look.frames[b]
        rows.append((fr.i64("costEst", b - p0, p1 - b), fr.i64("costEstAq", b - p0, p1 - b), fr.i64("intraMbs", b - p0)))
    parts = sharding.gather_costs(np.array(rows, np.int64).reshape(-1, 3))
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    look.close()
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(json_fd, 1)
    os.close(json_fd)
    if rank == 0:
        print(json.dumps({"n_gpus": world, "scaling": "weak", "value": args.frames * world / float(t.item()),
                          "workload": "CPU stand-in rehearsal of the multi-rank path (oracle + gloo); NOT a measurement",
                          "gathered_requests": [int(p.shape[0]) for p in parts]}), flush=True)


if __name__ == "__main__":
    main()
