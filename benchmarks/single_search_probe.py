"""Latency of ONE cost request with motion searches (what CostEstimateGroup::singleCost costs outside a batch),
by search mode and cooperative slices.  Diagnostic; run on the GPU box."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from x265_amd import api, synth
    W, H, D, F = 1920, 1080, 8, 8
    slices = int(os.environ.get("PROBE_SLICES", "0"))
    clip = synth.luma_clip(W, H, F, D)
    ctx = api.Context(W, H, depth=D, num_slots=F, lookahead_slices=slices)
    ctx.set_mvcost(*api.bitcost_table(D))
    for i in range(F):
        ctx.lowres_init(i, clip[i], i)
    ctx.intra_estimate(list(range(F)))
    ctx.sync()
    for label, req in (("P request, 1 search", (0, 1, 1, 0, 1, 1, 1, 0, 0, 1)), ("B request, 2 searches", (0, 1, 2, 0, 1, 2, 1, 1, 0, 1)),
                       ("cost-only request", (0, 1, 2, 0, 1, 2, 0, 0, 0, 1))):
        ts = []
        for rep in range(6):
            t0 = time.perf_counter()
            ctx.estimate_costs([req])
            ts.append(time.perf_counter() - t0)
        print("%s slices=%d mode=%s: %.3f ms" % (label, slices, os.environ.get("X265GPU_SEARCH_MODE", "diag"), 1000 * min(ts)))
    ctx.close()


if __name__ == "__main__":
    main()
