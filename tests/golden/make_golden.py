#!/usr/bin/env python3
"""Regenerates tests/golden/lookahead_*.npz from the CPU oracle.

The vectors pin the oracle against regressions and give the GPU tests a target
that does not need the oracle at run time.  They are oracle outputs, not
reference outputs: the reference cannot be built in this container, so parity
with it stays unpinned.  Inputs are the seeded clips of x265_amd.synth.

    python tests/golden/make_golden.py
"""
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

CASES = {
    # name: (width, height, frames, depth, bframes, seed, with_aq, search settings)
    "lookahead_8bit": (320, 192, 7, 8, 4, 1234, True, {}),
    "lookahead_10bit": (328, 184, 6, 10, 3, 4321, False, {}),
    "lookahead_umh_8bit": (352, 208, 6, 8, 4, 2468, False, {"search": 2}),
    # --hme with x265's defaults: hex at quarter resolution, umh at half resolution, ranges 16 / 32
    "lookahead_hme_10bit": (640, 368, 6, 10, 4, 1357, False,
                            {"hme": True, "hme_methods": [1, 2, 2], "hme_range": [16, 32, 48]}),
}
PU_CASES = {
    # name: (width, height, depth, seed, search method, merange, subme, jobs)
    "pu_me_star_8bit": (416, 240, 8, 11, 3, 57, 3, 64),
    "pu_me_hex_10bit": (416, 240, 10, 12, 1, 16, 5, 64),
}
PU_SHAPES = [(8, 8), (8, 4), (4, 8), (16, 16), (16, 8), (8, 16), (16, 4), (16, 12), (4, 16), (12, 16),
             (32, 32), (32, 16), (16, 32), (32, 8), (32, 24), (8, 32), (24, 32),
             (64, 64), (64, 32), (32, 64), (64, 16), (64, 48), (16, 64), (48, 64)]
REQUESTS = [(0, 1, 1), (0, 2, 1), (0, 2, 2), (1, 3, 2), (0, 3, 3), (2, 3, 3), (2, 4, 3), (3, 5, 4), (1, 5, 5), (1, 5, 3)]


def aq_factors(n, seed):
    return np.random.default_rng(seed).integers(128, 513, size=n).astype(np.int32)


def generate(name):
    from oracle import orc
    from x265_amd import synth
    W, H, N, depth, bframes, seed, with_aq, search = CASES[name]
    clip = synth.luma_clip(W, H, N, depth, seed=seed)
    param = orc.make_param(W, H, bframes=bframes, **search)
    ref = orc.Lookahead(depth, param, with_aq=with_aq)
    out = {"meta": np.array([W, H, N, depth, bframes, seed, int(with_aq)], np.int64),
           "search": np.array(json.dumps(search)),
           "requests": np.array(REQUESTS, np.int32)}
    for i, f in enumerate(clip):
        n = ((W // 2 + 7) // 8) * ((H // 2 + 7) // 8)
        lr = ref.add_frame(f, aq_factors(n, seed + i) if with_aq else None)
        out["plane_crc_%d" % i] = np.array([zlib.crc32(lr.buffer(p).tobytes()) for p in range(4)], np.uint32)
        out["intra_cost_%d" % i] = lr.intra_cost().copy()
        out["intra_mode_%d" % i] = lr.intra_mode().copy()
        out["intra_rows_%d" % i] = lr.row_satds(0, 0).copy()
        out["intra_sums_%d" % i] = np.array([lr.i64("costEst", 0, 0), lr.i64("costEstAq", 0, 0)], np.int64)
    for k, (p0, p1, b) in enumerate(REQUESTS):
        if max(p0, p1, b) >= N or b - p0 > bframes + 1 or p1 - b > bframes + 1:
            continue
        score = ref.frame_cost(p0, p1, b)
        fr = ref.frames[b]
        d0, d1 = b - p0, p1 - b
        out["score_%d" % k] = np.array([score, fr.i64("costEstAq", d0, d1), fr.i64("intraMbs", d0)], np.int64)
        out["mv0_%d" % k] = fr.mvs(0, d0).copy()
        out["mvcost0_%d" % k] = fr.mv_costs(0, d0).copy()
        if d1:
            out["mv1_%d" % k] = fr.mvs(1, d1).copy()
            out["mvcost1_%d" % k] = fr.mv_costs(1, d1).copy()
        if search.get("hme"):
            out["hme_mv0_%d" % k] = fr.lower_mvs(0, d0).copy()
            out["hme_mvcost0_%d" % k] = fr.lower_mv_costs(0, d0).copy()
        out["lowres_costs_%d" % k] = fr.lowres_costs(d0, d1).copy()
        out["row_satds_%d" % k] = fr.row_satds(d0, d1).copy()
    ref.close()
    return out


def pu_pictures(W, H, depth, seed, ctu=64):
    """Two frames of the seeded clip, padded the way PicYuv + extendPicBorder pad a picture."""
    from x265_amd import synth
    clip = synth.luma_clip(W, H, 2, depth, seed=seed)
    mx, my = ctu + 32, ctu + 16
    fw, fh = -(-W // ctu) * ctu, -(-H // ctu) * ctu
    pads = [np.ascontiguousarray(np.pad(f, ((my, my + fh - H), (mx, mx + fw - W)), mode="edge")) for f in clip]
    return clip, pads, mx, my, fw, fh


def pu_jobs(rng, fw, fh, mx, my, merange, count):
    """Random PUs of every shape with a predictor, a search window kept inside the padded picture, and candidates."""
    jobs = []
    while len(jobs) < count:
        w, h = PU_SHAPES[rng.integers(len(PU_SHAPES))]
        x = int(rng.integers(0, (fw - w) // 4 + 1)) * 4
        y = int(rng.integers(0, (fh - h) // 4 + 1)) * 4
        mvp = (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))
        lo = (max((mvp[0] >> 2) - merange, -x - mx + 24), max((mvp[1] >> 2) - merange, -y - my + 24))
        hi = (min((mvp[0] >> 2) + merange, fw - x - w + mx - 24), min((mvp[1] >> 2) + merange, fh - y - h + my - 24))
        if lo[0] > hi[0] or lo[1] > hi[1]:
            continue
        mvc = [(int(rng.integers(-60, 61)), int(rng.integers(-60, 61))) for _ in range(int(rng.integers(0, 4)))]
        jobs.append((x, y, w, h, mvp, lo, hi, mvc))
    return jobs


def generate_pu(name):
    from oracle import orc
    W, H, depth, seed, method, merange, subme, count = PU_CASES[name]
    clip, pads, mx, my, fw, fh = pu_pictures(W, H, depth, seed)
    table, centre = orc.bitcost_table(depth)
    ref = orc.FullResME(depth, table, centre, pads[1], pads[0], mx, my)
    jobs = pu_jobs(np.random.default_rng(seed), fw, fh, mx, my, merange, count)
    res = [ref.estimate(j[0], j[1], j[2], j[3], j[4], j[5], j[6], method, merange, subme, j[7]) for j in jobs]
    ref.close()
    return {"meta": np.array([W, H, depth, seed, method, merange, subme, count], np.int64),
            "results": np.array([[mv[0], mv[1], cost] for mv, cost in res], np.int32)}


if __name__ == "__main__":
    for name in CASES:
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **generate(name))
        print("wrote", name)
    for name in PU_CASES:
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **generate_pu(name))
        print("wrote", name)
