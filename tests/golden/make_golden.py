#!/usr/bin/env python3
"""Regenerates tests/golden/lookahead_*.npz from the CPU oracle.

The vectors pin the oracle against regressions and give the GPU tests a target
that does not need the oracle at run time.  They are oracle outputs, not
reference outputs: the reference cannot be built in this container, so parity
with it stays unpinned.  Inputs are the seeded clips of x265_amd.synth.

    python tests/golden/make_golden.py
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

CASES = {
    # name: (width, height, frames, depth, bframes, seed, with_aq)
    "lookahead_8bit": (320, 192, 7, 8, 4, 1234, True),
    "lookahead_10bit": (328, 184, 6, 10, 3, 4321, False),
}
REQUESTS = [(0, 1, 1), (0, 2, 1), (0, 2, 2), (1, 3, 2), (0, 3, 3), (2, 3, 3), (2, 4, 3), (3, 5, 4), (1, 5, 5), (1, 5, 3)]


def aq_factors(n, seed):
    return np.random.default_rng(seed).integers(128, 513, size=n).astype(np.int32)


def generate(name):
    from oracle import orc
    from x265_amd import synth
    W, H, N, depth, bframes, seed, with_aq = CASES[name]
    clip = synth.luma_clip(W, H, N, depth, seed=seed)
    param = orc.make_param(W, H, bframes=bframes)
    ref = orc.Lookahead(depth, param, with_aq=with_aq)
    out = {"meta": np.array([W, H, N, depth, bframes, seed, int(with_aq)], np.int64),
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
        out["lowres_costs_%d" % k] = fr.lowres_costs(d0, d1).copy()
        out["row_satds_%d" % k] = fr.row_satds(d0, d1).copy()
    ref.close()
    return out


if __name__ == "__main__":
    for name in CASES:
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **generate(name))
        print("wrote", name)
