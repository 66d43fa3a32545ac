"""GPU: the HIP path against the committed golden vectors (tests/golden/*.npz), and
size-independent properties at the benchmark's full frame size."""
import importlib.util
import json
import os
import zlib

import numpy as np
import pytest

from x265_amd import api, synth
from x265_amd.lookahead import CostEstimateGroup, LowresState, PreLookaheadGroup

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _mg():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _table(depth):
    from oracle import orc            # the MV-cost table is a host-built INPUT of the device path
    return orc.bitcost_table(depth)


@pytest.mark.parametrize("name", ["lookahead_8bit", "lookahead_10bit", "lookahead_umh_8bit", "lookahead_hme_10bit"])
def test_against_golden(name):
    mg = _mg()
    want = np.load(os.path.join(HERE, "golden", name + ".npz"))
    W, H, N, depth, bframes, seed, with_aq = [int(v) for v in want["meta"]]
    search = json.loads(str(want["search"]))
    clip = synth.luma_clip(W, H, N, depth, seed=seed)
    with api.Context(W, H, depth=depth, bframes=bframes, num_slots=N, **search) as ctx:
        ctx.set_mvcost(*_table(depth))
        frames = [LowresState(i, i, ctx.D) for i in range(N)]
        aq = [mg.aq_factors(ctx.geom.cuCount, seed + i) for i in range(N)] if with_aq else None
        PreLookaheadGroup(ctx).process_tasks(frames, clip, aq)
        for i in range(N):
            crc = np.array([zlib.crc32(ctx.get_plane(i, p).tobytes()) for p in range(4)], np.uint32)
            np.testing.assert_array_equal(crc, want["plane_crc_%d" % i])
            cost, mode = ctx.get_intra(i)
            np.testing.assert_array_equal(cost, want["intra_cost_%d" % i])
            np.testing.assert_array_equal(mode, want["intra_mode_%d" % i])
            _, rs, (ce, ceaq, _) = ctx.get_costs(i, 0, 0)
            np.testing.assert_array_equal(rs, want["intra_rows_%d" % i])
            np.testing.assert_array_equal([ce, ceaq], want["intra_sums_%d" % i])
        est = CostEstimateGroup(ctx, frames)
        for k, (p0, p1, b) in enumerate(want["requests"]):
            if "score_%d" % k not in want.files:
                continue
            p0, p1, b = int(p0), int(p1), int(b)
            d0, d1 = b - p0, p1 - b
            score = est.single_cost(p0, p1, b)
            assert [score, frames[b].cost_est_aq[(d0, d1)], frames[b].intra_mbs[d0]] == list(want["score_%d" % k])
            mv, c = ctx.get_mvs(b, 0, d0)
            np.testing.assert_array_equal(mv, want["mv0_%d" % k])
            np.testing.assert_array_equal(c, want["mvcost0_%d" % k])
            if d1:
                mv, c = ctx.get_mvs(b, 1, d1)
                np.testing.assert_array_equal(mv, want["mv1_%d" % k])
                np.testing.assert_array_equal(c, want["mvcost1_%d" % k])
            if search.get("hme"):
                mv, c = ctx.get_hme_mvs(b, 0, d0)
                np.testing.assert_array_equal(mv, want["hme_mv0_%d" % k])
                np.testing.assert_array_equal(c, want["hme_mvcost0_%d" % k])
            lc, rs, _ = ctx.get_costs(b, d0, d1)
            np.testing.assert_array_equal(lc, want["lowres_costs_%d" % k])
            np.testing.assert_array_equal(rs, want["row_satds_%d" % k])


@pytest.mark.parametrize("name", ["pu_me_star_8bit", "pu_me_hex_10bit"])
def test_pu_me_against_golden(name):
    mg = _mg()
    want = np.load(os.path.join(HERE, "golden", name + ".npz"))
    W, H, depth, seed, method, merange, subme, count = [int(v) for v in want["meta"]]
    clip, pads, mx, my, fw, fh = mg.pu_pictures(W, H, depth, seed)
    jobs = mg.pu_jobs(np.random.default_rng(seed), fw, fh, mx, my, merange, count)
    with api.Context(W, H, depth=depth, num_slots=2, keep_full_res=True) as ctx:
        ctx.lowres_init(0, clip[0])
        ctx.lowres_init(1, clip[1])
        ctx.me_set_mvcost(*_table(depth))
        got = ctx.pu_motion_estimate(1, 0, method, merange, subme, jobs)
    np.testing.assert_array_equal(np.array([[mv[0], mv[1], c] for mv, c in got], np.int32), want["results"])


@pytest.mark.parametrize("depth", [8, 10])
def test_full_size_properties(depth):
    """1920x1080: things that must hold whatever the content, checked without the oracle."""
    W, H, N = 1920, 1080, 4
    clip = synth.luma_clip(W, H, N, depth, seed=99, scene_cut=False)
    clip[2] = clip[1]                                         # a repeated frame
    with api.Context(W, H, depth=depth, bframes=4, num_slots=N) as ctx:
        ctx.set_mvcost(*_table(depth))
        g = ctx.geom
        assert (g.width, g.lines, g.stride, g.widthInCU, g.heightInCU) == (960, 544, 1152, 120, 68)
        frames = [LowresState(i, i, ctx.D) for i in range(N)]
        PreLookaheadGroup(ctx).process_tasks(frames, clip)
        rows = g.lines + 2 * g.marginY
        for p in range(4):
            buf = ctx.get_plane(0, p).reshape(rows, g.stride)
            inner = buf[g.marginY:g.marginY + g.lines, g.marginX:g.marginX + g.width]
            np.testing.assert_array_equal(buf, np.pad(inner, ((g.marginY,) * 2, (g.marginX,) * 2), mode="edge"))
        # plane 0 is the rounding 2x2 mean of the source (nested averages, vertical first)
        s = clip[0].astype(np.int64)
        v = (s[0::2] + s[1::2] + 1) >> 1
        want = (v[:, 0::2] + v[:, 1::2] + 1) >> 1
        buf = ctx.get_plane(0, 0).reshape(rows, g.stride)
        np.testing.assert_array_equal(buf[g.marginY:g.marginY + 540, g.marginX:g.marginX + 960], want)

        est = CostEstimateGroup(ctx, frames)
        # identical frames: zero vectors, P cost = blocks * (mvcost(0,0) + penalty) capped by intra
        p_same = est.single_cost(1, 2, 2)
        mv, c = ctx.get_mvs(2, 0, 1)
        assert not mv.any()
        lc, rs, (ce, ceaq, imb) = ctx.get_costs(2, 1, 0)
        assert ce == p_same and ce == ceaq and imb == 0
        assert int(rs.sum()) == int((lc & 0x3FFF).astype(np.int64).sum())     # rows sum the per-block costs
        interior = (lc & 0x3FFF).reshape(g.heightInCU, g.widthInCU)[1:-1, 1:-1]
        assert ce == int(interior.astype(np.int64).sum())                     # frame cost excludes edge blocks
        # vectors stay inside the search bounds (block position +- frame edge + 8, plus refinement)
        est.single_cost(0, 3, 1)
        for lst, dist in ((0, 1), (1, 2)):
            mv, _ = ctx.get_mvs(1, lst, dist)
            mv = mv.reshape(g.heightInCU, g.widthInCU, 2)
            xs = np.arange(g.widthInCU)[None, :] * 8
            ys = np.arange(g.heightInCU)[:, None] * 8
            assert ((mv[..., 0] >= (-xs - 8 - 4) * 4) & (mv[..., 0] <= (g.width - 8 - xs + 8 + 4) * 4)).all()
            assert ((mv[..., 1] >= (-ys - 8) * 4) & (mv[..., 1] <= (g.lines - 8 - ys + 8) * 4)).all()
        # a B request never costs more than its cheaper list plus the penalty, per block
        lc, _, _ = ctx.get_costs(1, 1, 2)
        _, c0 = ctx.get_mvs(1, 0, 1)
        _, c1 = ctx.get_mvs(1, 1, 2)
        assert ((lc & 0x3FFF) <= np.minimum(np.minimum(c0, c1) + 4, 0x3FFF)).all()
        # determinism: the same request on a fresh context gives identical vectors
        mv_a, c_a = ctx.get_mvs(1, 0, 1)
    with api.Context(W, H, depth=depth, bframes=4, num_slots=N) as ctx2:
        ctx2.set_mvcost(*_table(depth))
        frames = [LowresState(i, i, ctx2.D) for i in range(N)]
        PreLookaheadGroup(ctx2).process_tasks(frames, clip)
        CostEstimateGroup(ctx2, frames).single_cost(0, 3, 1)
        mv_b, c_b = ctx2.get_mvs(1, 0, 1)
        np.testing.assert_array_equal(mv_a, mv_b)
        np.testing.assert_array_equal(c_a, c_b)
