"""Where a flow search's dependency chain spends its time.  Diagnostic; run on the GPU box against the timing build
(make -C x265_amd/csrc stamp): every block of the launch's first job leaves the 100 MHz clock at the ends of its
stages, and this script walks the chain backwards from the last block to finish, always to the neighbour whose vector
arrived last, and adds up the steps along it: those whose search had run ahead on assumed vectors that agreed (a
hand-over and a compare) and those that searched after the last vector came.

  PROBE_JOBS=10 python benchmarks/chain_probe.py        (searches in the launch; the stamps are the first one's)
  X265GPU_FLOW_STAGED_JOBS=0 ...                        (the one-wait form: every step searches after)"""
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

    # stamps: 0 start, 1 first stage done (what hangs on the vectors at hand at the first look), 2 last neighbour
    # vector seen, 3 predictor picked, 4 integer search done, 5 sub-pel window staged, 6 result stored.  A block whose
    # search ran ahead on assumed vectors that then agreed carries the moment of that agreement in 2..5.
    names = ("ran-ahead", "searched-after")
    tot = {k: 0 for k in names}
    count = {k: 0 for k in names}
    rows = {}
    x = y = 0                                               # the last block of the wavefront
    steps = 0
    while True:
        i = x + y * W8
        s = st[i]
        nb = neighbours(x, y)
        if not nb:
            break
        a, b = max(nb, key=lambda p: store[p[0] + p[1] * W8])
        kind = "ran-ahead" if s[4] - s[3] < 50 else "searched-after"
        tot[kind] += int(s[6] - store[a + b * W8])
        count[kind] += 1
        rows[y] = rows.get(y, 0) + 1
        steps += 1
        x, y = a, b
    span = int(store[0] - st[:, 0].min())
    print(json.dumps({"jobs": jobs, "staged_jobs": os.environ.get("X265GPU_FLOW_STAGED_JOBS", "default"), "call_ms": round(best * 1e3, 3),
                      "first_job_span_us": span / 1e3, "chain_steps": steps,
                      "chain_steps_by_kind": count, "chain_us_by_kind": {k: round(v / 1e3, 1) for k, v in tot.items()},
                      "mean_step_ns_by_kind": {k: round(tot[k] / count[k]) if count[k] else None for k in names},
                      "chain_steps_in_bottom_row": rows.get(H8 - 1, 0),
                      "all_blocks_mean_ns": {"wave_life": float((st[:, 6] - st[:, 0]).mean()),
                                             "first-stage": float((st[:, 1] - st[:, 0]).mean())}}))


if __name__ == "__main__":
    main()
