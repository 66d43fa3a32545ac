"""The C oracle's motion search and block cost estimate against a second derivation written from the x265 sources
in plain Python (tests/independent_lookahead.py), array for array, on pictures small enough for Python loops."""
import numpy as np
import pytest

from oracle import orc
from tests.independent_lookahead import IndependentLookahead
from x265_amd import synth


def _fast_clip(W, H, n, depth, seed):
    """textured frames that move by up to ~11 pixels a frame, so that searches leave the zero vector and reach
    the picture edge"""
    rng = np.random.default_rng(seed)
    big = rng.integers(0, 1 << depth, (H + 80, W + 80)).astype(np.float64)
    k = np.ones(5) / 5.0
    big = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, big)
    big = np.apply_along_axis(lambda c: np.convolve(c, k, mode="same"), 0, big)
    out = []
    for f in range(n):
        ox, oy = 20 + 11 * f, 20 + 7 * f
        img = big[oy:oy + H, ox:ox + W] + rng.integers(-3, 4, (H, W))
        out.append(np.clip(np.rint(img), 0, (1 << depth) - 1).astype(np.uint8 if depth == 8 else np.uint16))
    return out


def _compare(clip, W, H, depth, requests, with_aq, seed):
    param = orc.make_param(W, H, bframes=4)
    look = orc.Lookahead(depth, param, with_aq=with_aq)
    rng = np.random.default_rng(seed)
    aq = []
    for f in clip:
        q = None
        if with_aq:
            tmp = orc.Lowres(look.L, depth, param)
            q = rng.integers(128, 513, size=tmp.ncu()).astype(np.int32)
            tmp.close()
        aq.append(q)
        look.add_frame(f, q)
    g = look.frames[0].geometry
    stride = g["stride"]
    oy, ox = divmod(g["padoffset"], stride)
    planes = [[lr.buffer(i).reshape(-1, stride).copy() for i in range(4)] for lr in look.frames]
    ind = IndependentLookahead(planes, (oy, ox), g["widthInCU"], g["heightInCU"],
                               [lr.intra_cost().copy() for lr in look.frames], aq, look.table, look.centre)
    moved = 0
    for p0, p1, b in requests:
        want = look.frame_cost(p0, p1, b)
        got = ind.estimate_frame_cost(p0, p1, b)
        fo, fi = look.frames[b], ind.frames[b]
        d0, d1 = b - p0, p1 - b
        what = " of request %s" % ((p0, p1, b),)
        np.testing.assert_array_equal(fi.mvs[0][d0], fo.mvs(0, d0), err_msg="L0 vectors" + what)
        np.testing.assert_array_equal(fi.mv_costs[0][d0], fo.mv_costs(0, d0), err_msg="L0 costs" + what)
        if d1:
            np.testing.assert_array_equal(fi.mvs[1][d1], fo.mvs(1, d1), err_msg="L1 vectors" + what)
            np.testing.assert_array_equal(fi.mv_costs[1][d1], fo.mv_costs(1, d1), err_msg="L1 costs" + what)
        np.testing.assert_array_equal(fi.lowres_costs[d0][d1], fo.lowres_costs(d0, d1), err_msg="lowresCosts" + what)
        np.testing.assert_array_equal(fi.row_satds[d0][d1], fo.row_satds(d0, d1), err_msg="rowSatds" + what)
        assert got == want and fi.cost_est[d0][d1] == fo.i64("costEst", d0, d1)
        assert fi.cost_est_aq[d0][d1] == fo.i64("costEstAq", d0, d1)
        if not d1:
            assert fi.intra_mbs[d0] == fo.i64("intraMbs", d0)
        moved += int(np.abs(fi.mvs[0][d0]).max() >= 16)
    look.close()
    return moved


@pytest.mark.parametrize("depth", [8, 10])
def test_block_costs_and_vectors_match_second_derivation(depth):
    W, H = 256, 160
    clip = list(synth.luma_clip(W, H, 4, depth, seed=21, scene_cut=False))
    # P requests at distances 1..3, then B requests that reuse their vectors and search list 1
    requests = [(0, 1, 1), (0, 2, 2), (0, 3, 3), (0, 2, 1), (0, 3, 1), (0, 3, 2), (1, 3, 2)]
    _compare(clip, W, H, depth, requests, with_aq=False, seed=1)


def test_fast_motion_and_aq_weights():
    W, H = 192, 128
    clip = _fast_clip(W, H, 3, 8, seed=5)
    moved = _compare(clip, W, H, 8, [(0, 1, 1), (0, 2, 1), (0, 2, 2), (1, 2, 2)], with_aq=True, seed=2)
    assert moved >= 2                      # vectors of several pixels: the hexagon walks, not only the refinement


def test_bidirectional_first():
    """A B request before any P request searches both lists itself (skip shortcut included)."""
    W, H = 96, 64
    clip = list(synth.luma_clip(W, H, 3, 8, seed=8, scene_cut=False))
    _compare(clip, W, H, 8, [(0, 2, 1), (0, 1, 1), (1, 2, 2)], with_aq=False, seed=3)
