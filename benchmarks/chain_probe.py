"""Where a flow search's dependency chain spends its time.  Diagnostic; run on the GPU box against the timing build
(make -C x265_amd/csrc stamp): every block of the launch's first job leaves the 100 MHz clock at the ends of its
stages, and this script walks the chain backwards from the last block to finish, always to the neighbour whose vector
arrived last, and adds up the stages along it.

  PROBE_JOBS=10 python benchmarks/chain_probe.py        (searches in the launch; the stamps are the first one's)
  X265GPU_SEARCH_LEAN_JOBS=64 ...                       (launches of at most that many searches take the lean kernel)"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from x265_amd import api, synth
    W, H, D = 1920, 1080, 8
    jobs = int(os.environ.get("PROBE_JOBS", "10"))
    F = min(jobs + 1, 24) + 1
    clip = synth.luma_clip(W, H, F, D)
    lib = os.path.join(ROOT, "x265_amd", os.environ.get("PROBE_LIB", "libx265gpu_stamp.so"))
    ctx = api.Context(W, H, depth=D, num_slots=F, lib_path=lib)
    ctx.set_mvcost(*api.bitcost_table(D))
    for i in range(F):
        ctx.lowres_init(i, clip[i], i)
    ctx.intra_estimate(list(range(F)))
    ctx.sync()
    pairs = [(b - d, b) for d in range(1, 5) for b in range(d, F)]
    assert jobs <= len(pairs)
    reqs = [(p0, b, b, p0, b, b, 1, 0) for p0, b in pairs[:jobs]]
    best = None
    for rep in range(4):
        t0 = time.perf_counter()
        ctx.estimate(reqs)
        ctx.sync()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    g = ctx.geom
    W8, H8 = g.widthInCU, g.heightInCU
    n = W8 * H8
    st2 = np.zeros((1 << 16, 8), np.uint64)
    st = st2[:n]
    ctx.L.x265gpu_debug_read_stamps.argtypes = [C.c_void_p, C.c_size_t]
    rc = ctx.L.x265gpu_debug_read_stamps(st2.ctypes.data, st2.nbytes)
    assert rc == 0, rc
    ctx.close()
    st = st.astype(np.int64) * 10                          # nanoseconds
    store = st[:, 6]

    def neighbours(x, y):
        out = []
        if x < W8 - 1:
            out.append((x + 1, y))
        if y < H8 - 1:
            out.append((x, y + 1))
            if x > 0:
                out.append((x - 1, y + 1))
            if x < W8 - 1:
                out.append((x + 1, y + 1))
        return out

    # stamps: 0 start, 1 early stage done (candidates of the early vectors, windows, their SATDs), 2 late vectors seen,
    # 3 predictor picked, 4 integer search done, 5 sub-pel window staged, 6 result stored, 7 sub-pel window re-staged
    names = ("handoff", "late-candidates+pick", "integer", "subpel-stage", "subpel", "early-stage-not-done")
    tot = dict.fromkeys(names, 0)
    x = y = 0                                               # the last block of the wavefront
    steps = 0
    staged = 0
    not_done = 0
    early_lag = []                                          # early vectors stored -> early stage done, along the chain
    while True:
        i = x + y * W8
        s = st[i]
        nb = neighbours(x, y)
        dep = max((store[a + b * W8] for a, b in nb), default=None)
        if dep is not None:
            early = [store[a + b * W8] for a, b in ((x, y + 1), (x + 1, y + 1)) if a < W8 and b < H8]
            if early:
                early_lag.append(int(s[1] - max(early)))
            if s[1] > dep:                                  # the inputs were there before the early stage was through
                not_done += 1
                tot["early-stage-not-done"] += int(s[2] - dep)
            else:
                tot["handoff"] += int(s[2] - dep)
        tot["late-candidates+pick"] += int(s[3] - s[2])
        tot["integer"] += int(s[4] - s[3])
        tot["subpel-stage"] += int(s[5] - s[4])
        tot["subpel"] += int(s[6] - s[5])
        staged += int(s[7] != s[0])
        steps += 1
        if not nb:
            break
        x, y = max(nb, key=lambda p: store[p[0] + p[1] * W8])
    span = int(store[0] - st[:, 0].min())
    print(json.dumps({"jobs": jobs, "lean_jobs": os.environ.get("X265GPU_SEARCH_LEAN_JOBS", "0"), "call_ms": round(best * 1e3, 3), "first_job_span_us": span / 1e3, "chain_steps": steps,
                      "subpel_windows_staged_on_chain": staged, "steps_early_stage_not_done": not_done,
                      "early_vectors_to_early_stage_done_ns": {"mean": float(np.mean(early_lag)), "median": float(np.median(early_lag)),
                                                               "p10": float(np.percentile(early_lag, 10)), "p90": float(np.percentile(early_lag, 90))},
                      "chain_us": {k: round(v / 1e3, 1) for k, v in tot.items()},
                      "per_step_ns": {k: round(v / steps) for k, v in tot.items()},
                      "all_blocks_mean_ns": {"wave_life": float((st[:, 6] - st[:, 0]).mean()),
                                             "early-stage": float((st[:, 1] - st[:, 0]).mean()),
                                             "late-wait": float((st[:, 2] - st[:, 1]).mean()),
                                             "late-candidates+pick": float((st[:, 3] - st[:, 2]).mean()),
                                             "integer": float((st[:, 4] - st[:, 3]).mean()),
                                             "subpel-stage": float((st[:, 5] - st[:, 4]).mean()),
                                             "subpel": float((st[:, 6] - st[:, 5]).mean())}}))


if __name__ == "__main__":
    main()
