"""Time of one search launch against the number of searches in it, for the two ways the single-launch search takes a
block's neighbour vectors (X265GPU_FLOW_STAGED_JOBS = 0: one wait; large: as the dependent chain produces them, the
search run ahead on assumed ones).
Diagnostic; run on the GPU box.  The crossover is where the library's default for that limit comes from.

  python benchmarks/launch_size_sweep.py [--cut]          (--cut: a clip with a scene cut, whose vector fields are rough)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from x265_amd import api, synth
    W, H, F, D = 1920, 1080, 60, 8
    clip = synth.luma_clip(W, H, F, D, scene_cut="--cut" in sys.argv)
    ctx = api.Context(W, H, depth=D, num_slots=F, lib_path=os.environ.get("PROBE_LIB"))
    ctx.set_mvcost(*api.bitcost_table(D))
    for i in range(F):
        ctx.lowres_init(i, clip[i], i)
    ctx.intra_estimate(list(range(F)))
    ctx.sync()
    pairs = [(b - d, b) for d in range(1, 6) for b in range(d, F)]
    for n in (1, 4, 10, 20, 30, 40, 60, 80, 120, 160, len(pairs)):
        reqs = [(p0, b, b, p0, b, b, 1, 0) for p0, b in pairs[:n]]
        row = []
        for limit in ("0", "100000"):
            os.environ["X265GPU_FLOW_STAGED_JOBS"] = limit
            best = None
            for rep in range(5):
                t0 = time.perf_counter()
                ctx.estimate(reqs)
                ctx.sync()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            row.append(best * 1e3)
        print("%4d searches: one wait %.3f ms, as they come %.3f ms (%+.1f %%)" % (n, row[0], row[1], 100 * (row[1] / row[0] - 1)), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
