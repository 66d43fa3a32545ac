"""GPU parity of the AQ block energies and weighted-prediction frame sums against the oracle, bit-exact."""
import numpy as np
import pytest

from oracle import orc
from x265_amd import api

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth", [8, 10, 12])
@pytest.mark.parametrize("size", [(416, 240), (1920, 1080), (200, 120), (178, 94)])
def test_aq_energy(depth, size):
    W, H = size
    rng = np.random.default_rng(W + depth)
    dt = np.uint8 if depth == 8 else np.uint16
    top = (1 << depth) - 1
    y = rng.integers(0, top + 1, (H, W)).astype(dt)
    y[: H // 3] = (np.arange(W) * top // W).astype(dt)            # a smooth area next to noise
    u = rng.integers(top // 3, top // 2, (H // 2, W // 2)).astype(dt)
    v = rng.integers(0, top + 1, (H // 2, W // 2)).astype(dt)
    blocks = ((W + 15) // 16) * ((H + 15) // 16)
    want_e = np.zeros(blocks, np.uint32)
    want_wp = np.zeros(6, np.uint64)
    orc.lib(depth).orc_aq_energy(orc.ptr(y), W, orc.ptr(u), orc.ptr(v), W // 2, W, H, orc.ptr(want_e), orc.ptr(want_wp))
    with api.Context(W, H, depth=depth, num_slots=1) as ctx:
        e, wsum, wssd = ctx.aq_energy(y, u, v)
    np.testing.assert_array_equal(e, want_e)
    np.testing.assert_array_equal(wsum, want_wp[:3])
    np.testing.assert_array_equal(wssd, want_wp[3:])


def _picture(W, H, depth, seed):
    rng = np.random.default_rng(seed)
    dt = np.uint8 if depth == 8 else np.uint16
    top = (1 << depth) - 1
    y = rng.integers(0, top + 1, (H, W)).astype(dt)
    y[: H // 3] = (np.arange(W) * top // W).astype(dt)
    y[H // 3: H // 2] = top // 5                                    # flat: energy 0 -> the log2(max(e, 1)) branch
    u = rng.integers(top // 3, top // 2, (H // 2, W // 2)).astype(dt)
    v = rng.integers(0, top + 1, (H // 2, W // 2)).astype(dt)
    return y, u, v, rng


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("mode,strength", [(1, 1.0), (2, 1.0), (3, 0.8), (2, 0.0), (0, 1.0)])
@pytest.mark.parametrize("weightp", [False, True])
def test_aq_frame(depth, mode, strength, weightp):
    """calcAdaptiveQuantFrame end to end: every double offset and every scale factor equal to the oracle's."""
    W, H = 416, 240
    y, u, v, rng = _picture(W, H, depth, 99 + depth + mode)
    blocks = ((W + 15) // 16) * ((H + 15) // 16)
    for qo in (None, rng.normal(0, 1.5, blocks).astype(np.float32)):
        for hdr in (False, True):
            want = orc.calc_adaptive_quant_frame(depth, y, u, v, mode, strength, weightp=weightp, hdr10=hdr,
                                                 quant_offsets=qo)
            with api.Context(W, H, depth=depth, num_slots=1) as ctx:
                got = ctx.aq_frame(y, u, v, mode, strength, weightp=weightp, hdr10=hdr, quant_offsets=qo, slot=0)
            for g, w_, name in zip(got, want, ("qpAqOffset", "qpCuTreeOffset", "invQscaleFactor", "wp_sum", "wp_ssd")):
                np.testing.assert_array_equal(g, w_, err_msg="%s mode %d hdr %s" % (name, mode, hdr))


def test_aq_frame_feeds_intra_costs():
    """The scale factors installed by aq_frame(slot=...) drive costEstAq exactly as the oracle's do."""
    from x265_amd import synth
    W, H, depth = 416, 240, 8
    clip = synth.luma_clip(W, H, 1, depth, seed=5)
    y, u, v, _ = _picture(W, H, depth, 7)
    y = clip[0]
    _, _, inv, _, _ = orc.calc_adaptive_quant_frame(depth, y, u, v, 2, 1.0)
    param = orc.make_param(W, H)
    ref = orc.Lookahead(depth, param, with_aq=True)
    ref.add_frame(y, inv)
    with api.Context(W, H, depth=depth, num_slots=1) as ctx:
        ctx.set_mvcost(ref.table, ref.centre)
        ctx.lowres_init(0, y)
        got = ctx.aq_frame(y, u, v, 2, 1.0, slot=0)
        np.testing.assert_array_equal(got[2], inv)
        ctx.intra_estimate([0])
        _, _, (ce, ceaq, _) = ctx.get_costs(0, 0, 0)
    assert ce == ref.frames[0].i64("costEst", 0, 0)
    assert ceaq == ref.frames[0].i64("costEstAq", 0, 0)
    ref.close()


def test_aq_frames_batches_of_shrinking_size_keep_the_earlier_factors():
    """A batch of four pictures followed at once by a batch of one (another layout of the shared page-locked
    arrays: HDR10 adds an array per picture): the factors of the first batch's last picture must reach its slot
    untouched although nothing waited for their upload in between."""
    import ctypes as C
    from x265_amd import synth
    hip = C.CDLL("libamdhip64.so")        # the runtime the library under test already runs on
    held = []

    def to_device(a):
        a = np.ascontiguousarray(a)
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(a.nbytes)) == 0
        held.append(p)
        assert hip.hipMemcpy(p, C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), 1) == 0     # hipMemcpyHostToDevice
        return p.value
    W, H, depth, N = 416, 240, 8, 4
    clip = synth.luma_clip(W, H, N, depth, seed=11)
    chroma = [_picture(W, H, depth, 20 + i)[1:3] for i in range(N)]
    param = orc.make_param(W, H)
    ref = orc.Lookahead(depth, param, with_aq=True)
    invs = []
    for i in range(N):
        _, _, inv, _, _ = orc.calc_adaptive_quant_frame(depth, clip[i], chroma[i][0], chroma[i][1], 2, 1.0)
        invs.append(inv)
        ref.add_frame(clip[i], inv)
    ptrs = [(to_device(clip[i]), to_device(chroma[i][0]), to_device(chroma[i][1])) for i in range(N)]
    with api.Context(W, H, depth=depth, num_slots=N) as ctx:
        ctx.set_mvcost(ref.table, ref.centre)
        for rep in range(3):
            ctx.lowres_init_device(list(range(N)), [p[0] for p in ptrs], [W] * N)
            ctx.intra_estimate(list(range(N)))                 # keeps the side stream busy ahead of the installing copies
            first = ctx.aq_frames_device(list(range(N)), ptrs, W, W // 2, 2, 1.0)
            keep = ctx._aq_keep
            second = ctx.aq_frames_device([0], ptrs[:1], W, W // 2, 2, 1.0, hdr10=bool(rep & 1), finish=False)
            for i in range(N):
                np.testing.assert_array_equal(first[i][2], invs[i])
            for i in range(N):
                _, _, (ce, ceaq, _) = ctx.get_costs(i, 0, 0)
                assert ce == ref.frames[i].i64("costEst", 0, 0)
                if i == 0 and (rep & 1):
                    continue                                   # re-mapped with the HDR10 luma term by the second batch
                assert ceaq == ref.frames[i].i64("costEstAq", 0, 0), "picture %d, round %d" % (i, rep)
            del keep, second
    ref.close()
    for p in held:
        hip.hipFree(p)


def test_aq_motion_frame_against_its_formula():
    """x265gpu_aq_motion_frame (calcMotionAdaptiveQuantFrame, slicetype.cpp:3609-3663) on a bidirectional request:
    recomputed here from the vectors and list decisions the device holds, with the host's libm (math.pow / math.sqrt)
    and sums in block order."""
    import math
    from x265_amd import synth
    from x265_amd.lookahead import CostEstimateGroup, LowresState, PreLookaheadGroup
    W, H, depth, N = 416, 240, 8, 3
    clip = synth.luma_clip(W, H, N, depth, seed=3, scene_cut=False)
    table, centre = api.bitcost_table(depth)
    with api.Context(W, H, depth=depth, num_slots=N) as ctx:
        ctx.set_mvcost(table, centre)
        frames = [LowresState(i, i, ctx.D) for i in range(N)]
        PreLookaheadGroup(ctx).process_tasks(frames, clip)
        CostEstimateGroup(ctx, frames).single_cost(0, 2, 1)
        n = ctx.geom.cuCount
        rng = np.random.default_rng(4)
        qa, qc = rng.normal(0, 1, n), rng.normal(0, 1, n)
        inv = rng.integers(100, 400, n).astype(np.int32)
        want_qa, want_qc, want_inv = qa.copy(), qc.copy(), inv.copy()
        got_motion = ctx.aq_motion_frame(1, 1, 1, qa, qc, inv)
        lc = ctx.get_costs(1, 1, 1)[0]
        mv0, mv1 = ctx.get_mvs(1, 0, 1)[0], ctx.get_mvs(1, 1, 1)[0]
        adj = np.zeros(n)
        for i in range(n):
            lists = int(lc[i]) >> 14
            d = 0.0
            if lists & 1:
                d += math.sqrt(float(abs(int(mv0[i, 0])) ** 2) + float(abs(int(mv0[i, 1])) ** 2))
            if lists & 2:
                d += math.sqrt(float(abs(int(mv1[i, 0])) ** 2) + float(abs(int(mv1[i, 1])) ** 2))
            if lists == 3:
                d = d / 2
            adj[i] = math.pow(d, 0.1)
        avg = avg2 = 0.0
        for v in adj.tolist():
            avg += v
            avg2 += v * v
        avg /= n
        avg2 /= n
        sd = math.sqrt(avg2 - avg * avg)
        assert sd > 0
        raised = 0
        for i in range(n):
            q = (adj[i] - avg) / sd
            if q > 1:
                want_qa[i] += q
                want_qc[i] += q
                want_inv[i] += orc.lib(depth).orc_exp2fix8(orc.C.c_double(q))
                raised += 1
        assert 0 < raised < n
        np.testing.assert_array_equal(got_motion, adj)
        np.testing.assert_array_equal(qa, want_qa)
        np.testing.assert_array_equal(qc, want_qc)
        np.testing.assert_array_equal(inv, want_inv)


def test_aq_frame_rejects_uncovered_modes():
    W, H = 64, 64
    y = np.zeros((H, W), np.uint8)
    u = v = np.zeros((H // 2, W // 2), np.uint8)
    with api.Context(W, H, depth=8, num_slots=1) as ctx:
        with pytest.raises(api.X265GpuError):
            ctx.aq_frame(y, u, v, 5, 1.0)                           # no such mode


def _edgy_picture(W, H, depth, seed):
    """Edges at many angles: bars, both diagonals at several slopes, discs, a ramp, noise and a flat band."""
    rng = np.random.default_rng(seed)
    top = (1 << depth) - 1
    yy, xx = np.mgrid[0:H, 0:W]
    y = np.zeros((H, W), np.int64)
    y += ((xx // 23) % 2) * (top // 2) * (yy < H // 4)                                 # vertical bars
    y += ((yy // 11) % 2) * (top // 3) * ((yy >= H // 4) & (yy < H // 2) & (xx < W // 3))   # horizontal bars
    for k, slope in enumerate((1, -1, 2, -3, 0.5)):
        band = (xx >= W // 3 + k * (W // 8)) & (xx < W // 3 + (k + 1) * (W // 8)) & (yy >= H // 4) & (yy < 3 * H // 4)
        y += ((((xx + slope * yy) // 9).astype(np.int64)) % 2) * (top * 2 // 3) * band
    y += (((xx - W // 6) ** 2 + (yy - 3 * H // 4) ** 2) < (H // 6) ** 2) * (top // 2)  # a disc: every angle
    y += (xx * top // (4 * W)) * (yy >= 3 * H // 4)                                    # ramp
    y += rng.integers(0, max(2, top // 40), (H, W)) * (xx > 3 * W // 4)                # mild noise on the right
    y = np.clip(y, 0, top).astype(np.uint8 if depth == 8 else np.uint16)
    dt = y.dtype
    u = rng.integers(top // 3, top // 2, (H // 2, W // 2)).astype(dt)
    v = rng.integers(0, top + 1, (H // 2, W // 2)).astype(dt)
    return y, u, v


@pytest.mark.parametrize("depth", [8, 10, 12])
@pytest.mark.parametrize("size,qg", [((416, 240), 32), ((200, 120), 32), ((178, 94), 32), ((424, 248), 8), ((1920, 1080), 32)])
def test_aq_edge_mode(depth, size, qg):
    """X265_AQ_EDGE: blurred picture, Sobel gradients, edge density and mean angle per block, the offsets mapped from
    them, and wp_sum / wp_ssd with the edge picture's sums added in, all equal to the oracle's."""
    W, H = size
    y, u, v = _edgy_picture(W, H, depth, 40 + depth)
    names = ("qpAqOffset", "qpCuTreeOffset", "invQscaleFactor", "wp_sum", "wp_ssd", "invQscaleFactor8x8")
    for weightp, strength in ((False, 1.0), (True, 0.7)):
        want = orc.calc_adaptive_quant_frame(depth, y, u, v, 4, strength, weightp=weightp, qg_size=qg)
        with api.Context(W, H, depth=depth, num_slots=1) as ctx:
            got = ctx.aq_frame(y, u, v, 4, strength, weightp=weightp, slot=0, qg_size=qg)
        assert len(got) == len(want)
        for g, w_, name in zip(got, want, names):
            np.testing.assert_array_equal(g, w_, err_msg="%s weightp %s" % (name, weightp))
    # the edge term changes the outcome: same picture, plain auto-variance
    plain = orc.calc_adaptive_quant_frame(depth, y, u, v, 2, 0.7, weightp=True, qg_size=qg)
    assert (plain[0] != want[0]).mean() > 0.2


@pytest.mark.parametrize("qg", [8, 32])
def test_aq_edge_mode_full_swing_12bit(qg):
    """12-bit pictures with black-to-white steps: Sobel gradients of the blurred picture reach 16 * 4095, more than
    16 bits hold; the pixels whose angle the host decides carry them whole."""
    W, H, depth = 554, 140, 12
    rng = np.random.default_rng(615366048)
    y = rng.integers(0, 4096, (H, W)).astype(np.uint16)
    blocks = np.repeat(np.repeat(rng.integers(0, 2, (H // 4 + 1, W // 4 + 1)), 4, axis=0), 4, axis=1)[:H, :W]
    y[H // 2:] = (blocks[H // 2:] * 4095).astype(np.uint16)             # 4x4 checks of 0 / 4095: steep diagonals
    u = rng.integers(0, 4096, (H // 2, W // 2)).astype(np.uint16)
    v = rng.integers(1365, 2048, (H // 2, W // 2)).astype(np.uint16)
    want = orc.calc_adaptive_quant_frame(depth, y, u, v, 4, 0.4, hdr10=True, qg_size=qg)
    with api.Context(W, H, depth=depth, num_slots=1) as ctx:
        got = ctx.aq_frame(y, u, v, 4, 0.4, hdr10=True, slot=0, qg_size=qg)
    for k, (g, w_) in enumerate(zip(got, want)):
        np.testing.assert_array_equal(g, w_, err_msg="output %d" % k)


def test_aq_edge_mode_in_bands(monkeypatch):
    """The picture pass in bands of three tile rows (as a picture too large for one list of host-side angles runs)."""
    monkeypatch.setenv("X265GPU_AQ_EDGE_BAND_ROWS", "3")
    W, H, depth = 416, 240, 8
    y, u, v = _edgy_picture(W, H, depth, 77)
    want = orc.calc_adaptive_quant_frame(depth, y, u, v, 4, 1.0, weightp=True)
    with api.Context(W, H, depth=depth, num_slots=1) as ctx:
        got = ctx.aq_frame(y, u, v, 4, 1.0, weightp=True, slot=0)
    for g, w_ in zip(got, want):
        np.testing.assert_array_equal(g, w_)


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("mode,strength", [(1, 1.0), (2, 1.2), (3, 0.8), (2, 0.0)])
@pytest.mark.parametrize("size", [(416, 240), (424, 248), (200, 120)])
def test_aq_frame_qg_size_8(depth, mode, strength, size):
    """qgSize 8: 8x8 + 4x4 blocks, its own constants, the running index over a grid that can be narrower than
    the arrays (424 = 53 blocks against rows of 54), and invQscaleFactor8x8."""
    W, H = size
    y, u, v, rng = _picture(W, H, depth, 5 + depth + mode)
    w8, h8 = ((W // 2) + 7) >> 3, ((H // 2) + 7) >> 3
    names = ("qpAqOffset", "qpCuTreeOffset", "invQscaleFactor", "wp_sum", "wp_ssd", "invQscaleFactor8x8")
    for qo in (None, rng.normal(0, 1.5, 4 * w8 * h8).astype(np.float32)):
        for weightp, hdr in ((False, False), (True, True)):
            want = orc.calc_adaptive_quant_frame(depth, y, u, v, mode, strength, weightp=weightp, hdr10=hdr,
                                                 quant_offsets=qo, qg_size=8)
            with api.Context(W, H, depth=depth, num_slots=1) as ctx:
                got = ctx.aq_frame(y, u, v, mode, strength, weightp=weightp, hdr10=hdr, quant_offsets=qo, slot=0,
                                   qg_size=8)
            assert len(got) == len(want) == 6
            for g, w_, name in zip(got, want, names):
                np.testing.assert_array_equal(g, w_, err_msg="%s mode %d hdr %s" % (name, mode, hdr))
            assert np.all(got[5] >= 0)


@pytest.mark.parametrize("qg", [8, 32])
@pytest.mark.parametrize("weightp", [False, True])
@pytest.mark.parametrize("size", [(416, 240), (424, 248), (410, 230), (1920, 1080)])
def test_aq_block_variance_tail(qg, weightp, size):
    """bDynamicRefine / bEnableFades: blockVariance in loop order over the (possibly rounded) picture, the
    running-row frameVariance, and wp_sum / wp_ssd after the second energy round."""
    W, H = size
    y, u, v, _ = _picture(W, H, 8, 17)
    for mode in (0, 2):
        want = orc.calc_adaptive_quant_frame(8, y, u, v, mode, 1.0, weightp=weightp, qg_size=qg, block_variance=True)
        with api.Context(W, H, depth=8, num_slots=1) as ctx:
            got = ctx.aq_frame(y, u, v, mode, 1.0, weightp=weightp, qg_size=qg, block_variance=True)
        np.testing.assert_array_equal(got[-2], want[-2], err_msg="blockVariance")
        assert got[-1] == want[-1]
        np.testing.assert_array_equal(got[3], want[3], err_msg="wp_sum")
        np.testing.assert_array_equal(got[4], want[4], err_msg="wp_ssd")
        if mode:
            np.testing.assert_array_equal(got[2], want[2], err_msg="invQscaleFactor")
        assert np.any(want[-2])


def test_qg_size_8_end_to_end_intra_and_cutree_finish():
    """invQscaleFactor8x8 from aq_frame drives costEstAq, and cuTreeFinish's qgSize 8 loop spreads each lowres
    block's ratio over its four full-resolution entries."""
    from x265_amd import synth
    W, H, depth = 416, 240, 8
    clip = synth.luma_clip(W, H, 2, depth, seed=5)
    _, u, v, rng = _picture(W, H, depth, 7)
    param = orc.make_param(W, H)
    ref = orc.Lookahead(depth, param, with_aq=True)
    L = orc.lib(depth)
    with api.Context(W, H, depth=depth, num_slots=2) as ctx:
        ctx.set_mvcost(ref.table, ref.centre)
        w8, h8 = ctx.geom.widthInCU, ctx.geom.heightInCU
        offs = []
        for k in range(2):
            want = orc.calc_adaptive_quant_frame(depth, clip[k], u, v, 2, 1.0, qg_size=8)
            ref.add_frame(clip[k], want[5])
            ctx.lowres_init(k, clip[k])
            got = ctx.aq_frame(clip[k], u, v, 2, 1.0, slot=k, qg_size=8)
            np.testing.assert_array_equal(got[5], want[5])
            offs.append(got[0])
        ctx.intra_estimate([0, 1])
        _, _, (ce, ceaq, _) = ctx.get_costs(0, 0, 0)
        assert ceaq == ref.frames[0].i64("costEstAq", 0, 0) and ce != ceaq
        # P frame 1 from frame 0, then propagate into frame 0 and finish it
        ref.frame_cost(0, 1, 1)
        ctx.estimate([(0, 1, 1, 0, 1, 1, 1, 0, 0, 0)])
        ctx.cutree_clear(0)
        ctx.cutree_clear(1)
        ctx.cutree_propagate(0, 1, 1, 1, 0, 1, 1.0, 32)
        prop = ctx.cutree_get_propagate_cost(0)
        assert np.any(prop)
        intra = ctx.get_intra(0)[0]
        inv8 = orc.calc_adaptive_quant_frame(depth, clip[0], u, v, 2, 1.0, qg_size=8)[5]
        qa = offs[0].copy()
        want_qc = qa.copy()
        got_qc = qa.copy()
        L.orc_cutree_finish_qg8(w8, h8, orc.ptr(np.ascontiguousarray(intra, np.int32)), orc.ptr(inv8),
                                orc.ptr(np.ascontiguousarray(prop, np.uint16)), 256, 0.25, 2.0, orc.ptr(qa), orc.ptr(want_qc))
        ctx.cutree_finish(0, 256, 0.25, 2.0, qa, got_qc, qg8=True)
        np.testing.assert_array_equal(got_qc, want_qc)
        assert np.any(got_qc != qa)
    ref.close()


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("size,qg", [((416, 240), 32), ((424, 250), 16), ((416, 240), 8), ((1920, 1080), 64)])
def test_hevc_aq_layers(depth, size, qg):
    """rc.hevcAq: per-layer quadrant activities, dAvgActivity and QP offsets, invQscaleFactor from the finest
    layer, and the wp sums of the closing acEnergyCu round - against the oracle's xPreanalyze."""
    W, H = size
    y, u, v, _ = _picture(W, H, depth, 31 + depth)
    y[H // 2:, : W // 2] = (np.arange(W // 2) % 7 + 5).astype(y.dtype)       # low-variance quadrants
    w8, h8 = ((W // 2) + 7) >> 3, ((H // 2) + 7) >> 3
    # CTU 64: the depths aqLayerDepth enables for this qgSize (lowres.h:120-126); the finest one has the
    # 2 * lowres grid, the others ceil(picture / part) parts (lowres.cpp:139-150)
    depths = {64: (0, 2), 32: (0, 1, 2), 16: (0, 1, 2), 8: (0, 1, 2, 3)}[qg]
    parts = []
    for d in depths:
        s = 64 >> d
        if d == 3:
            parts.append((s, 2 * w8, 2 * h8))
        else:
            parts.append((s, (W + s - 1) // s, (H + s - 1) // s))
    L = orc.lib(depth)
    want_layers = []
    for (s, nw, nh) in parts:
        arrs = [np.full(nw * nh, np.nan) for _ in range(3)]
        avg = L.orc_hevc_aq_layer(orc.ptr(y), W, W, H, s, s, nw, nh, 1.5, *(orc.ptr(a) for a in arrs))
        want_layers.append(tuple(arrs) + (avg,))
    with api.Context(W, H, depth=depth, num_slots=1) as ctx:
        got = ctx.aq_frame(y, u, v, 2, 1.0, weightp=True, qg_size=qg, hevc_aq_parts=parts, qp_adaptation_range=1.5)
    for k, (g, w_) in enumerate(zip(got[-1], want_layers)):
        for a, b, name in zip(g, w_, ("dActivity", "dQpOffset", "dCuTreeOffset", "dAvgActivity")):
            np.testing.assert_array_equal(a, b, err_msg="layer %d %s" % (k, name))
    B = 8 if qg == 8 else 16
    visited = ((W + B - 1) // B) * ((H + B - 1) // B)
    want_inv = np.array([L.orc_exp2fix8(float(q)) for q in want_layers[-1][1][:visited]], np.int32)
    np.testing.assert_array_equal(got[2][:visited], want_inv)
    assert len(set(want_inv.tolist())) > 3
    # the energy round feeds wp_sum / wp_ssd exactly as in the other modes
    ref = orc.calc_adaptive_quant_frame(depth, y, u, v, 2, 1.0, weightp=True, qg_size=qg)
    np.testing.assert_array_equal(got[3], ref[3])
    np.testing.assert_array_equal(got[4], ref[4])
    if qg == 8:
        grid = got[2].reshape(2 * h8, 2 * w8)
        np.testing.assert_array_equal(got[5].reshape(h8, w8),
                                      (grid[0::2, 0::2] + grid[0::2, 1::2] + grid[1::2, 0::2] + grid[1::2, 1::2]) // 4)
