"""Probe: do two independent lookahead pipelines (two contexts, two host threads) overlap on one GPU?

Prints frames/s for 1 and 2 pipelines on the default 1080p workload.  Not part of bench.py's contract."""
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from bench import plan_step            # noqa: E402
from oracle import orc                 # noqa: E402
from x265_amd import api, synth       # noqa: E402


def main():
    W, H, D, F, steps = 1920, 1080, 8, 60, 8
    clip = torch.from_numpy(synth.luma_clip(W, H, F, D, seed=1234)).cuda()
    ptrs = [clip.data_ptr() + i * W * H for i in range(F)]
    slots = list(range(F))
    table, centre = orc.bitcost_table(D)
    for pipes in (1, 2, 3):
        ctxs = []
        for _ in range(pipes):
            st = torch.cuda.Stream()
            c = api.Context(W, H, depth=D, bframes=4, num_slots=F, device=0, stream=st.cuda_stream)
            c.set_mvcost(table, centre)
            r1, r2 = plan_step(c, F, 4)
            a1, a2 = c._pack(r1), c._pack(r2)
            ctxs.append((c, st, a1, a2, (api.FrameCost * len(a1))(), (api.FrameCost * len(a2))()))

        def run(k, n):
            c, st, a1, a2, o1, o2 = ctxs[k]
            for _ in range(n):
                c.lowres_init_device(slots, ptrs, [W] * F)
                c.intra_estimate(slots)
                c._chk(c.L.x265gpu_estimate_costs(c.h, len(a1), a1, o1))
                c._chk(c.L.x265gpu_estimate_costs(c.h, len(a2), a2, o2))

        for k in range(pipes):
            run(k, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=(k, steps)) for k in range(pipes)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("pipelines=%d  %.0f frames/s  (%.2f ms per 60-frame segment)" % (pipes, pipes * steps * F / dt,
                                                                             1000 * dt / (pipes * steps)), flush=True)
        for c in ctxs:
            c[0].close()


if __name__ == "__main__":
    main()
