"""GPU parity through the C++ host layer (x265_amd/host): PreLookaheadGroup, CostEstimateGroup, LookaheadTLD and
CuTree written in x265's own language above the C-ABI, against the CPU oracle."""
import numpy as np
import pytest

from oracle import orc
from x265_amd import api, synth

pytestmark = pytest.mark.gpu

I, P, B, R = 2, 3, 5, 4          # X265_TYPE_I / P / B / BREF


class Slots:
    """What _check_request of the Python-mirror tests expects of a frame: its slot."""

    def __init__(self, slot):
        self.slot = slot


def _check(ref, ctx, p0, p1, b):
    from test_gpu_lookahead import _check_request
    _check_request(ref, ctx, [Slots(i) for i in range(len(ref.frames))], p0, p1, b)


@pytest.mark.parametrize("depth,slices", [(8, 0), (10, 0), (8, 3)])
def test_single_costs_and_batches(depth, slices):
    W, H, N = (448, 528, 6) if slices else (384, 208, 8)
    clip = synth.luma_clip(W, H, N, depth, seed=71)
    rng = np.random.default_rng(72)
    ref = orc.Lookahead(depth, orc.make_param(W, H, lookahead_slices=slices), with_aq=True)
    ctx = api.Context(W, H, depth=depth, num_slots=N, lookahead_slices=slices)
    win = api.HostWindow(ctx, N)
    try:
        n = ctx.geom.cuCount
        invs = [rng.integers(128, 513, size=n).astype(np.int32) for _ in range(N)]
        for f, q in zip(clip, invs):
            ref.add_frame(f, q)
        ctx.set_mvcost(ref.table, ref.centre)
        win.pre_lookahead(list(clip), invs)
        for i in range(N):
            assert win.cost_est(i, 0, 0) == ref.frames[i].i64("costEst", 0, 0)
        seq = [(0, 1, 1), (0, 2, 2), (0, 2, 1), (1, 2, 2), (0, 3, 3), (0, 3, 1), (1, 3, 2), (3, 4, 4), (3, 5, 4), (0, 2, 1)]
        for (p0, p1, b) in seq:
            for pen in (False, True):
                assert win.single_cost(p0, p1, b, pen) == ref.frame_cost(p0, p1, b, pen, single=True), (p0, p1, b, pen)
            _check(ref, ctx, p0, p1, b)
            assert win.intra_mbs(b, b - p0) == ref.frames[b].i64("intraMbs", b - p0)
        batch = [(2, 5, 4), (1, 5, 3), (2, 5, 5), (3, 5, 5), (4, 5, 5)]
        for r in batch:
            win.add(*r)
        win.finish_batch()
        for (p0, p1, b) in batch:
            assert win.cost_est(b, b - p0, p1 - b) == ref.frame_cost(p0, p1, b), (p0, p1, b)
            _check(ref, ctx, p0, p1, b)
    finally:
        win.close()
        ctx.close()
        ref.close()


def test_weights_analyse():
    W, H, N, depth = 384, 208, 6, 8
    base = synth.luma_clip(W, H, N, depth, seed=9)[0]
    rng = np.random.default_rng(3)
    clip = [np.clip(np.roll(base, i, axis=1).astype(np.int32) * (64 - 6 * i) // 64 + 2 * i + rng.integers(-1, 2, base.shape),
                    0, 255).astype(np.uint8) for i in range(N)]
    ref = orc.Lookahead(depth, orc.make_param(W, H, weightp=True))
    ctx = api.Context(W, H, depth=depth, num_slots=N)
    win = api.HostWindow(ctx, N, weightp=True)
    try:
        for i, f in enumerate(clip):
            lr = ref.add_frame(f)
            v = f.astype(np.int64)
            ssum = int(v.sum())
            ssd = int((v * v).sum()) - ssum * ssum // v.size
            wsum, wssd = lr.wp()
            wsum[0], wssd[0] = ssum, ssd
        ctx.set_mvcost(ref.table, ref.centre)
        win.pre_lookahead(clip)
        for i in range(N):                       # after pre_lookahead: Lowres::init resets the host state
            wsum, wssd = ref.frames[i].wp()
            win.set_wp(i, int(wsum[0]), int(wssd[0]))
        nweighted = 0
        for (p0, p1, b) in [(0, 1, 1), (0, 2, 2), (0, 2, 1), (1, 3, 3), (1, 3, 2), (0, 5, 5), (2, 5, 3), (3, 4, 4), (2, 4, 4)]:
            assert win.single_cost(p0, p1, b) == ref.frame_cost(p0, p1, b), (p0, p1, b)
            _check(ref, ctx, p0, p1, b)
            want = bool(ref.frames[b].i64("isWeighted", b - p0))
            assert win.is_weighted(b, b - p0) == want, (p0, p1, b)
            nweighted += want
        assert nweighted >= 5
    finally:
        win.close()
        ctx.close()
        ref.close()


@pytest.mark.parametrize("types,kw,b_intra", [
    ([I, B, B, P, B, B, B, P, B, P], {}, False),
    ([I, B, B, P, B, R, B, P, B, P], {"b_pyramid": True, "weighted_bipred": True}, False),
    ([I, B, P, B, B, P, P, B], {"vbv_buffer_size": 1000, "fps_num": 24000, "fps_denom": 1001}, True),
    ([I, P], {"lookahead_depth": 0}, False),
])
def test_cu_tree(types, kw, b_intra):
    W, H, depth = 384, 208, 8
    N = len(types)
    clip = synth.luma_clip(W, H, N, depth, seed=11)
    rng = np.random.default_rng(11)
    ref = orc.Lookahead(depth, orc.make_param(W, H), with_aq=True)
    ctx = api.Context(W, H, depth=depth, num_slots=N)
    win = api.HostWindow(ctx, N)
    try:
        ctx.set_mvcost(ref.table, ref.centre)
        n = ctx.geom.cuCount
        invs, cutree = [], []
        for k, f in enumerate(clip):
            inv = rng.integers(128, 513, size=n).astype(np.int32)
            inv[rng.integers(0, n, 8)] = 0
            invs.append(inv)
            lr = ref.add_frame(f, inv)
            lr.set_slice_type(types[k])
            qa = rng.normal(0, 2, n)
            lr.qp_aq_offset()[:] = qa
            lr.qp_cutree_offset()[:] = qa
            cutree.append((qa.copy(), qa.copy()))
        win.pre_lookahead(list(clip), invs)
        for k in range(N):
            win.set_slice_type(k, types[k])
            win.set_qp_offsets(k, *cutree[k])
        cp = orc.CuTreeParam(kw.get("fps_num", 30), kw.get("fps_denom", 1), int(kw.get("b_pyramid", False)),
                             int(kw.get("weighted_bipred", False)), kw.get("lookahead_depth", 20),
                             kw.get("vbv_buffer_size", 0), kw.get("q_compress", 0.6))
        ref.cu_tree(cp, N - 1, b_intra)
        win.cu_tree(N - 1, b_intra, **kw)
        touched = 0
        for k in range(N):
            want = ref.frames[k].propagate_cost()
            np.testing.assert_array_equal(ctx.cutree_get_propagate_cost(k), want, err_msg="propagateCost of frame %d" % k)
            np.testing.assert_array_equal(cutree[k][1], ref.frames[k].qp_cutree_offset(), err_msg="qpCuTreeOffset %d" % k)
            touched += int(np.any(want))
        assert touched >= 1
    finally:
        win.close()
        ctx.close()
        ref.close()


@pytest.mark.parametrize("depth", [8, 10])
def test_pre_lookahead_with_adaptive_quant(depth):
    """PreLookaheadGroup::processTasks in full: Lowres::init, calcAdaptiveQuantFrame and the intra estimate in
    one host-layer call; scale factors, offsets, wp sums and the AQ frame costs against the oracle, then a
    weighted-prediction analysis that starts from those wp sums."""
    W, H, N = 384, 208, 3
    clip = synth.luma_clip(W, H, N, depth, seed=81)
    rng = np.random.default_rng(82)
    dt = clip[0].dtype
    top = (1 << depth) - 1
    chroma = [(rng.integers(0, top + 1, (H // 2, W // 2)).astype(dt), rng.integers(0, top + 1, (H // 2, W // 2)).astype(dt))
              for _ in range(N)]
    ref = orc.Lookahead(depth, orc.make_param(W, H), with_aq=True)
    ctx = api.Context(W, H, depth=depth, num_slots=N)
    win = api.HostWindow(ctx, N)
    try:
        n = ctx.geom.cuCount
        want = [orc.calc_adaptive_quant_frame(depth, clip[i], chroma[i][0], chroma[i][1], 2, 1.0, weightp=True) for i in range(N)]
        for i in range(N):
            ref.add_frame(clip[i], want[i][2])
        ctx.set_mvcost(ref.table, ref.centre)
        offs = [(np.full(n, np.nan), np.full(n, np.nan)) for _ in range(N)]
        for i in range(N):
            win.set_qp_offsets(i, *offs[i])
        invs = win.pre_lookahead_aq([(clip[i],) + chroma[i] for i in range(N)], 2, 1.0, weightp=True)
        for i in range(N):
            np.testing.assert_array_equal(invs[i], want[i][2])
            np.testing.assert_array_equal(offs[i][0], want[i][0])
            np.testing.assert_array_equal(offs[i][1], want[i][1])
            assert win.get_wp(i) == (int(want[i][3][0]), int(want[i][4][0]))
            assert win.cost_est(i, 0, 0) == ref.frames[i].i64("costEst", 0, 0)
        assert win.single_cost(0, 1, 1) == ref.frame_cost(0, 1, 1, single=True)
        _check(ref, ctx, 0, 1, 1)
    finally:
        win.close()
        ctx.close()
        ref.close()


@pytest.mark.parametrize("depth", [8, 10])
def test_window_on_device_pictures_weights_follow_intra(depth):
    """Device-resident pictures take the batched pre-lookahead: the AQ mapping runs on host threads and is only joined
    by the first call that reads its results.  A weighted batch right behind it, on a context whose arrays have never
    held an intra estimate: weightCostLuma reads the per-block intra costs, so the estimate must have run by then
    although its AQ-weighted sums cannot have.  Everything against the oracle: weights, costs, vectors, and the
    intra totals once the scale factors are in."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")        # the runtime the library under test already runs on
    held = []

    def to_device(a):
        a = np.ascontiguousarray(a)
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(a.nbytes)) == 0
        held.append(p)
        assert hip.hipMemcpy(p, C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), 1) == 0     # hipMemcpyHostToDevice
        return p.value
    W, H, N = 384, 208, 6
    top = (1 << depth) - 1
    base = synth.luma_clip(W, H, N, depth, seed=19)[0]
    rng = np.random.default_rng(5)
    dt = base.dtype
    clip = [np.clip(np.roll(base, i, axis=1).astype(np.int32) * (64 - 6 * i) // 64 + (2 * i << (depth - 8)) +
                    rng.integers(-1, 2, base.shape), 0, top).astype(dt) for i in range(N)]
    chroma = [(rng.integers(0, top + 1, (H // 2, W // 2)).astype(dt), rng.integers(0, top + 1, (H // 2, W // 2)).astype(dt))
              for _ in range(N)]
    ref = orc.Lookahead(depth, orc.make_param(W, H, weightp=True), with_aq=True)
    ctx = api.Context(W, H, depth=depth, num_slots=N)
    win = api.HostWindow(ctx, N, weightp=True)
    try:
        n = ctx.geom.cuCount
        want = [orc.calc_adaptive_quant_frame(depth, clip[i], chroma[i][0], chroma[i][1], 2, 1.0, weightp=True) for i in range(N)]
        for i in range(N):
            lr = ref.add_frame(clip[i], want[i][2])
            s, d = lr.wp()
            s[:], d[:] = want[i][3], want[i][4]
        ctx.set_mvcost(ref.table, ref.centre)
        offs = [(np.full(n, np.nan), np.full(n, np.nan)) for _ in range(N)]
        for i in range(N):
            win.set_qp_offsets(i, *offs[i])
        ptrs = [(to_device(clip[i]), to_device(chroma[i][0]), to_device(chroma[i][1]), W, W // 2) for i in range(N)]
        invs = win.pre_lookahead_aq(None, 2, 1.0, weightp=True, device_ptrs=ptrs, resolve=False)
        batch = [(0, 1, 1), (0, 2, 2), (0, 2, 1), (1, 3, 3), (1, 3, 2), (0, 5, 5), (2, 5, 3), (3, 4, 4), (2, 4, 4)]
        for r in batch:
            win.add(*r)
        win.finish_batch()
        nweighted = 0
        for (p0, p1, b) in batch:
            assert win.cost_est(b, b - p0, p1 - b) == ref.frame_cost(p0, p1, b), (p0, p1, b)
            _check(ref, ctx, p0, p1, b)
            if (p0, p1, b) == (0, 2, 1):
                continue        # finds the list-0 search of (0, 1, 1) done: the flag it leaves depends on request order
            weighted = bool(ref.frames[b].i64("isWeighted", b - p0))
            assert win.is_weighted(b, b - p0) == weighted, (p0, p1, b)
            nweighted += weighted
        assert nweighted >= 3
        for i in range(N):
            np.testing.assert_array_equal(invs[i], want[i][2])
            assert win.cost_est(i, 0, 0) == ref.frames[i].i64("costEst", 0, 0)
            np.testing.assert_array_equal(ctx.get_intra(i)[0], ref.frames[i].intra_cost())
            lc, rs, fc = ctx.get_costs(i, 0, 0)
            assert fc[:2] == (ref.frames[i].i64("costEst", 0, 0), ref.frames[i].i64("costEstAq", 0, 0))
            np.testing.assert_array_equal(rs, ref.frames[i].row_satds(0, 0))
    finally:
        win.close()
        ctx.close()
        ref.close()
        for p in held:
            hip.hipFree(p)
