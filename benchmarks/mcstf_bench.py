"""Throughput of the --mcstf motion search (x265gpu_mcstf_motion_search).

Workload: one picture against four references (mcstfFrameRange 2), all four pyramid levels, the
result arrays copied back to the host.  Prints one JSON line; `--cpu` also times the CPU oracle on
the same pictures (one reference) and checks it equal.

    python benchmarks/mcstf_bench.py [--width 1920 --height 1080 --depth 8 --reps 5 --cpu]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x265_amd import api, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cpu", action="store_true")
    a = ap.parse_args()
    W, H = a.width, a.height
    clip = synth.luma_clip(W, H, 5, a.depth, seed=1234)
    ctx = api.Context(W, H, depth=a.depth, num_slots=5, mcstf=True, keep_full_res=True)
    for i, f in enumerate(clip):
        ctx.lowres_init(i, f, i)
    refs = [0, 1, 3, 4]
    ctx.mcstf_motion_search(2, refs)          # warm-up
    times = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        got = ctx.mcstf_motion_search(2, refs)
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    out = {"metric": "mcstf motion searches (picture x reference, 4 levels) per second", "value": len(refs) / dt,
           "unit": "searches/s", "seconds_per_picture": dt, "references": len(refs),
           "config": {"workload": "%dx%d %d-bit, 4 references" % (W, H, a.depth)},
           "note": "wall time of the call, result download included"}
    if a.cpu:
        from oracle import orc
        param = orc.make_param(W, H, mcstf=True)
        L = orc.lib(a.depth)
        pics = []
        for i in (2, 1):
            lr = orc.Lowres(L, a.depth, param)
            lr.init(clip[i], i)
            pics.append(orc.McstfPicture(lr, clip[i]))
        t0 = time.perf_counter()
        want = orc.mcstf_motion_search(pics[0], [pics[1]])
        cdt = time.perf_counter() - t0
        for g, w in zip(got, want):
            assert np.array_equal(g[1], w[0]), "GPU result differs from the oracle"
        out["cpu_baseline"] = {"value": 1 / cdt, "unit": "searches/s", "cores": 1, "kind": "port",
                               "sample": "one reference, checked equal to the GPU result"}
    ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
