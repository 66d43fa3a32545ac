"""Where a lookahead step's wall time goes on the host side: times each host-layer call of the segment workload
and counts the requests of the steady workload by kind.  Diagnostic; run on the GPU box."""
import ctypes as C
import sys
import time
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    from x265_amd import api
    W, H, D, F, BF = 1920, 1080, 8, 60, 4
    clip, chroma = bench.make_clip(W, H, F, D, 1234)
    dev_y = torch.from_numpy(clip).cuda()
    dev_u = torch.from_numpy(np.stack([c[0] for c in chroma])).cuda()
    dev_v = torch.from_numpy(np.stack([c[1] for c in chroma])).cuda()
    torch.cuda.synchronize()
    yb, cb = W * H, (W // 2) * (H // 2)
    y_ptrs = [dev_y.data_ptr() + i * yb for i in range(F)]
    u_ptrs = [dev_u.data_ptr() + i * cb for i in range(F)]
    v_ptrs = [dev_v.data_ptr() + i * cb for i in range(F)]
    ctx = api.Context(W, H, depth=D, bframes=BF, num_slots=F, stream=torch.cuda.current_stream().cuda_stream)
    ctx.set_mvcost(*api.bitcost_table(D))
    n = ctx.geom.cuCount
    Hlib = api.load_host_library()
    win = api.HostWindow(ctx, F, weightp=True)
    offs = [(np.zeros(n), np.zeros(n)) for _ in range(F)]
    invs = [np.zeros(n, np.int32) for _ in range(F)]
    for i in range(F):
        win.set_qp_offsets(i, *offs[i])
    for k, t in enumerate(bench.segment_slice_types(F, BF)):
        win.set_slice_type(k, t)
    b1, b2 = bench.batch_requests(F, BF)
    a1 = np.ascontiguousarray(np.array(b1, np.int32))
    a2 = np.ascontiguousarray(np.array(b2, np.int32))
    vpn = C.c_void_p * F
    fr = (C.c_int32 * F)(*range(F))
    ys, us, vs = vpn(*y_ptrs), vpn(*u_ptrs), vpn(*v_ptrs)
    sy, sc = (C.c_ssize_t * F)(*[W] * F), (C.c_ssize_t * F)(*[W // 2] * F)
    aqp = api.AqParams(2, 1, 0, 0, 1.0, 32, 0, None, None, None, 0, 0, 0.0, None)
    iq = vpn(*[a.ctypes.data for a in invs])
    acc = {}

    def timed(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0

    reps = 5
    for rep in range(reps + 1):
        if rep == 1:
            acc.clear()
        timed("pre_lookahead_aq", lambda: win._chk(Hlib.xgh_pre_lookahead_aq(win.h, F, fr, ys, us, vs, sy, sc, C.byref(aqp), iq, api.MEM_DEVICE)))
        timed("batch1_add", lambda: Hlib.xgh_add_many(win.h, len(a1), a1.ctypes.data_as(C.c_void_p)))
        timed("batch1_finish", win.finish_batch)
        timed("batch2_add", lambda: Hlib.xgh_add_many(win.h, len(a2), a2.ctypes.data_as(C.c_void_p)))
        timed("batch2_finish", win.finish_batch)
        timed("cu_tree", lambda: win.cu_tree(F - 1, False, b_pyramid=True))
    print("segment step, ms per call:", {k: round(1000 * v / reps, 3) for k, v in acc.items()})
    # pieces of the pre-lookahead
    acc.clear()
    for rep in range(3):
        timed("lowres_init x60", lambda: ctx.lowres_init_device(list(range(F)), y_ptrs, [W] * F))
        timed("aq_frame x60", lambda: [ctx.aq_frame_device(i, y_ptrs[i], u_ptrs[i], v_ptrs[i], W, W // 2) for i in range(F)] if hasattr(ctx, "aq_frame_device") else None)
        timed("intra x60", lambda: ctx.intra_estimate(list(range(F))))
        timed("get_costs x60", lambda: [ctx.get_frame_cost(i, 0, 0) for i in range(F)])
    print("pre-lookahead pieces, ms:", {k: round(1000 * v / 3, 3) for k, v in acc.items()})

    p = api.LookaheadParam(**dict(bench.MEDIUM, bitDepth=D))
    la = api.HostLookahead(ctx, p)
    t0 = time.perf_counter()
    for i in range(F):
        la.add_device_picture(y_ptrs[i], u_ptrs[i], v_ptrs[i], W, W // 2)
        la.get_decided()
    la.flush()
    while la.get_decided() is not None:
        pass
    dt = time.perf_counter() - t0
    tr = la.trace()
    batch = tr[:, 5] == 1
    srch = (tr[:, 3] + tr[:, 4]) > 0
    print("steady: %.1f ms, %d decide calls, %d requests: %d batched (%d with searches), %d single (%d with searches)"
          % (1000 * dt, la.decide_calls(), len(tr), batch.sum(), (batch & srch).sum(), (~batch).sum(), (~batch & srch).sum()))
    print("steady phases, ms:", {k: round(1000 * v, 2) for k, v in la.seconds().items()}, "presearch hits", la.presearch_hits(), "precost hits", la.precost_hits())
    la.close()
    win.close()
    ctx.close()


if __name__ == "__main__":
    main()
