"""GPU parity of cuTree (propagateCost of every frame, qpCuTreeOffset) against the oracle, bit-exact."""
import numpy as np
import pytest

from oracle import orc
from x265_amd import api, synth
from x265_amd.lookahead import (CuTree, LowresState, PreLookaheadGroup, TYPE_B, TYPE_BREF, TYPE_I, TYPE_P)

pytestmark = pytest.mark.gpu

I, P, B, R = TYPE_I, TYPE_P, TYPE_B, TYPE_BREF


def _run(depth, types, cp_kwargs, b_intra, size=(384, 208), seed=11):
    W, H = size
    N = len(types)
    clip = synth.luma_clip(W, H, N, depth, seed=seed)
    param = orc.make_param(W, H, bframes=4)
    ref = orc.Lookahead(depth, param, with_aq=True)
    rng = np.random.default_rng(seed)
    ctx = api.Context(W, H, depth=depth, bframes=4, num_slots=N)
    try:
        ctx.set_mvcost(ref.table, ref.centre)
        frames = [LowresState(i, i, ctx.D) for i in range(N)]
        invs = []
        for k, f in enumerate(clip):
            n = ctx.geom.widthInCU * ctx.geom.heightInCU
            inv = rng.integers(128, 513, size=n).astype(np.int32)
            inv[rng.integers(0, n, 8)] = 0                         # a zero scaled intra cost: offset left alone
            invs.append(inv)
            lr = ref.add_frame(f, inv)
            lr.set_slice_type(types[k])
            qa = rng.normal(0, 2, n)
            lr.qp_aq_offset()[:] = qa
            lr.qp_cutree_offset()[:] = qa
            frames[k].slice_type = types[k]
            frames[k].qp_aq_offset = qa.copy()
            frames[k].qp_cutree_offset = qa.copy()
        PreLookaheadGroup(ctx).process_tasks(frames, clip, invs)
        cp = orc.CuTreeParam(cp_kwargs.get("fps_num", 30), cp_kwargs.get("fps_denom", 1),
                             int(cp_kwargs.get("b_pyramid", False)), int(cp_kwargs.get("weighted_bipred", False)),
                             cp_kwargs.get("lookahead_depth", 20), cp_kwargs.get("vbv_buffer_size", 0),
                             cp_kwargs.get("q_compress", 0.6))
        ref.cu_tree(cp, N - 1, b_intra)
        CuTree(ctx, **cp_kwargs).cu_tree(frames, N - 1, b_intra)
        touched = 0
        for k in range(N):
            want = ref.frames[k].propagate_cost()
            got = ctx.cutree_get_propagate_cost(frames[k].slot)
            np.testing.assert_array_equal(got, want, err_msg="propagateCost of frame %d" % k)
            np.testing.assert_array_equal(frames[k].qp_cutree_offset, ref.frames[k].qp_cutree_offset(),
                                          err_msg="qpCuTreeOffset of frame %d" % k)
            touched += int(np.any(want))
        return touched
    finally:
        ctx.close()
        ref.close()


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("b_intra", [False, True])
def test_cutree_plain_gops(depth, b_intra):
    assert _run(depth, [I, B, B, P, B, B, B, P, B, P], {}, b_intra) >= 3


def test_cutree_b_pyramid_and_weighted_bipred():
    types = [I, B, B, P, B, R, B, P, B, P]
    assert _run(8, types, {"b_pyramid": True, "weighted_bipred": True}, False) >= 4


def test_cutree_vbv_finishes_every_referenced_frame():
    assert _run(8, [I, B, P, B, B, P, P, B], {"vbv_buffer_size": 1000, "fps_num": 24000, "fps_denom": 1001}, False) >= 2


def test_cutree_without_lookahead_swaps_buffers():
    assert _run(8, [I, P], {"lookahead_depth": 0}, False) >= 1
    _run(8, [I, P], {"lookahead_depth": 0}, True)


def test_cutree_trailing_b_frames():
    """Trailing B frames are skipped; with a single mini-GOP only the intra call reaches frame 0."""
    types = [I, B, P, B, B]
    assert _run(8, types, {}, False) == 0
    assert _run(8, types, {}, True) >= 1


def test_cutree_saturates_like_uint16():
    """Scale factors at the top of their range push sums past 65535: the clamp must match CLIP_ADD."""
    W, H, depth, N = 256, 144, 8, 6
    clip = synth.luma_clip(W, H, N, depth, seed=3)
    clip[:] = clip[0]                                               # a still scene: every vector zero, maximal pile-up
    types = [I, P, P, P, P, P]
    param = orc.make_param(W, H, bframes=4)
    ref = orc.Lookahead(depth, param, with_aq=True)
    with api.Context(W, H, depth=depth, bframes=4, num_slots=N) as ctx:
        ctx.set_mvcost(ref.table, ref.centre)
        frames = [LowresState(i, i, ctx.D) for i in range(N)]
        n = ctx.geom.widthInCU * ctx.geom.heightInCU
        inv = np.full(n, 40000, np.int32)
        for k, f in enumerate(clip):
            # a moving bright bar makes inter cost < intra cost somewhere without moving the vectors much
            f = f.copy()
            f[:, 16 * k:16 * k + 8] = 255
            clip[k] = f
        for k, f in enumerate(clip):
            lr = ref.add_frame(f, inv)
            lr.set_slice_type(types[k])
            frames[k].slice_type = types[k]
            frames[k].qp_aq_offset = np.zeros(n)
            frames[k].qp_cutree_offset = np.zeros(n)
        PreLookaheadGroup(ctx).process_tasks(frames, clip, [inv] * N)
        ref.cu_tree(orc.CuTreeParam(30, 1, 0, 0, 20, 0, 0.6), N - 1, False)
        CuTree(ctx).cu_tree(frames, N - 1, False)
        saturated = 0
        for k in range(N):
            want = ref.frames[k].propagate_cost()
            np.testing.assert_array_equal(ctx.cutree_get_propagate_cost(k), want)
            np.testing.assert_array_equal(frames[k].qp_cutree_offset, ref.frames[k].qp_cutree_offset())
            saturated += int(np.sum(want == 65535))
        assert saturated > 0
    ref.close()


@pytest.mark.parametrize("qg", [8, 32])
def test_cutree_finish_with_hevc_aq_layers(qg):
    """cuTreeFinish with rc.hevcAq (computeCUTreeQpOffset): per-layer dCuTreeOffset from the device's propagate
    totals, intra costs and scale factors, against the oracle's statement of the layer loop."""
    W, H, depth = 416, 240, 8
    clip = synth.luma_clip(W, H, 2, depth, seed=9)
    rng = np.random.default_rng(10)
    param = orc.make_param(W, H)
    ref = orc.Lookahead(depth, param, with_aq=True)
    L = orc.lib(depth)
    with api.Context(W, H, depth=depth, num_slots=2) as ctx:
        ctx.set_mvcost(ref.table, ref.centre)
        w8, h8 = ctx.geom.widthInCU, ctx.geom.heightInCU
        n = w8 * h8
        invs = []
        for k in range(2):
            inv = rng.integers(128, 513, size=n).astype(np.int32)
            inv[rng.integers(0, n, 6)] = 0
            invs.append(inv)
            ref.add_frame(clip[k], inv)
            ctx.lowres_init(k, clip[k])
            ctx.set_aq(k, inv)
        ctx.intra_estimate([0, 1])
        ref.frame_cost(0, 1, 1)
        ctx.estimate([(0, 1, 1, 0, 1, 1, 1, 0)])
        ctx.cutree_clear(0)
        ctx.cutree_clear(1)
        ctx.cutree_propagate(0, 1, 1, 1, 0, 1, 1.0, 32)
        prop = np.ascontiguousarray(ctx.cutree_get_propagate_cost(0), np.uint16)
        intra = np.ascontiguousarray(ctx.get_intra(0)[0], np.int32)
        assert np.any(prop)
        depths = (0, 1, 2, 3) if qg == 8 else (0, 1, 2)
        layers = []
        for d in depths:
            s = 64 >> d
            nw, nh = (2 * w8, 2 * h8) if d == 3 else ((W + s - 1) // s, (H + s - 1) // s)
            layers.append((s, nw, nh, rng.normal(0, 2, nw * nh)))
        qg8 = qg == 8
        fps, wd, strength = 300, 0.125, 2.4
        ratio_want = rng.normal(0, 1, 4 * n) if qg8 else None
        ratio_got = None if ratio_want is None else ratio_want.copy()
        if qg8:
            L.orc_cutree_ratio8x8(w8, h8, orc.ptr(intra), orc.ptr(invs[0]), orc.ptr(prop), fps, wd, orc.ptr(ratio_want))
        got = ctx.cutree_finish_hevc(0, fps, wd, strength, qg, layers, ratio_got)
        for (s, nw, nh, qp), g in zip(layers, got):
            want = np.full(nw * nh, np.nan)
            qp = np.ascontiguousarray(qp)
            L.orc_cutree_hevc_layer(int(qg8), 2 * w8 if qg8 else w8, W, H, s, s, nw, nh, orc.ptr(intra), orc.ptr(invs[0]),
                                    orc.ptr(prop), fps, wd, strength, orc.ptr(ratio_want) if qg8 else None, orc.ptr(qp),
                                    orc.ptr(want))
            np.testing.assert_array_equal(g, want, err_msg="layer of %d" % s)
        if qg8:
            np.testing.assert_array_equal(ratio_got, ratio_want)
    ref.close()
