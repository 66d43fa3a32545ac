"""Throughput of the full-resolution PU motion estimation (x265gpu_pu_motion_estimate).

Workload: every square PU (64, 32, 16, 8) of every CTU of one 3840x2160 frame against the previous
frame, star search, merange 57, subme 3 (BASELINE config 4), zero predictors, search window clipped to
the padded picture.  Prints one JSON line; `--cpu-pus N` times the CPU oracle on the first N PUs.

    python benchmarks/pu_me_bench.py [--width 3840 --height 2160 --depth 8 --reps 3 --cpu-pus 400]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x265_amd import api, synth  # noqa: E402


def make_jobs(fw, fh, W, H, mx, my, merange, sizes):
    jobs = []
    for s in sizes:
        for y in range(0, H - s + 1, s):
            for x in range(0, W - s + 1, s):
                lo = (max(-merange, -x - mx + 24), max(-merange, -y - my + 24))
                hi = (min(merange, fw - x - s + mx - 24), min(merange, fh - y - s + my - 24))
                jobs.append((x, y, s, s, (0, 0), lo, hi, []))
    return jobs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--merange", type=int, default=57)
    ap.add_argument("--subme", type=int, default=3)
    ap.add_argument("--cpu-pus", type=int, default=400)
    ap.add_argument("--pageable", action="store_true")
    ap.add_argument("--method", default="star", choices=["dia", "hex", "umh", "star"])
    a = ap.parse_args()
    W, H, ctu = a.width, a.height, 64
    method = {"dia": api.DIA_SEARCH, "hex": api.HEX_SEARCH, "umh": api.UMH_SEARCH, "star": api.STAR_SEARCH}[a.method]
    clip = synth.luma_clip(W, H, 2, a.depth, seed=1234)
    mx, my = ctu + 32, ctu + 16
    fw, fh = -(-W // ctu) * ctu, -(-H // ctu) * ctu
    jobs = make_jobs(fw, fh, W, H, mx, my, a.merange, (64, 32, 16, 8))

    from oracle import orc          # the MV cost table is a host input; the oracle builds it like BitCost does
    table, centre = orc.bitcost_table(a.depth)
    ctx = api.Context(W, H, depth=a.depth, num_slots=2, ctu=ctu, keep_full_res=True)
    ctx.lowres_init(0, clip[0])
    ctx.lowres_init(1, clip[1])
    ctx.me_set_mvcost(table, centre)
    ctx.pu_motion_estimate(1, 0, method, a.merange, a.subme, jobs[:1024])      # warm-up
    # the job array is packed once, outside the timed region: a C++ caller fills x265gpu_pu_job structs directly
    import ctypes as C
    # ... into page-locked arrays from x265gpu_host_alloc (--pageable: plain host memory instead)
    arr = (api.PuJob * len(jobs))() if a.pageable else ctx.host_alloc(api.PuJob, len(jobs))
    for k, (x, y, w, h, mvp, mvmin, mvmax, mvc) in enumerate(jobs):
        j = arr[k]
        j.x, j.y, j.w, j.h = x, y, w, h
        j.mvpX, j.mvpY = mvp
        j.mvminX, j.mvminY = mvmin
        j.mvmaxX, j.mvmaxY = mvmax
        j.numCandidates = 0
    out_arr = (api.PuResult * len(jobs))() if a.pageable else ctx.host_alloc(api.PuResult, len(jobs))
    times = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        ctx._chk(ctx.L.x265gpu_pu_motion_estimate(ctx.h, 1, 0, method, a.merange, a.subme, len(jobs), arr, out_arr))
        times.append(time.perf_counter() - t0)
    res = [((o.mvX, o.mvY), o.cost) for o in out_arr]
    dt = sorted(times)[len(times) // 2]
    out = {"metric": "full-res PU motion estimations per second (%s, merange %d, subme %d)" % (a.method, a.merange, a.subme),
           "value": len(jobs) / dt, "unit": "PU/s", "pus": len(jobs), "seconds": dt,
           "config": {"workload": "%dx%d %d-bit, square PUs 64/32/16/8 of one frame" % (W, H, a.depth)},
           "note": "wall time of x265gpu_pu_motion_estimate: job upload, kernels and result download; job and "
                   "result arrays in %s host memory" % ("pageable" if a.pageable else "page-locked")}
    if a.cpu_pus:
        pads = [np.pad(f, ((my, my + fh - H), (mx, mx + fw - W)), mode="edge") for f in clip]
        ref = orc.FullResME(a.depth, table, centre, pads[1], pads[0], mx, my)
        step = max(1, len(jobs) // a.cpu_pus)
        sample = jobs[::step]
        t0 = time.perf_counter()
        want = [ref.estimate(j[0], j[1], j[2], j[3], j[4], j[5], j[6], method, a.merange, a.subme, j[7]) for j in sample]
        cdt = time.perf_counter() - t0
        assert want == res[::step], "GPU result differs from the oracle"
        out["cpu_baseline"] = {"value": len(sample) / cdt, "unit": "PU/s", "cores": 1, "kind": "port",
                               "sample": "every %dth PU (%d PUs), checked equal to the GPU result" % (step, len(sample))}
    if not a.pageable:
        ctx.host_free(arr)
        ctx.host_free(out_arr)
    ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
