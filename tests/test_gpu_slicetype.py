"""GPU: x265's slice-type decision run by the C++ host layer over device-side cost estimation, against the oracle's
restatement of the same logic over the CPU cost estimation (oracle/orc_slicetype.c).

What has to be identical: the decided types, scene-cut and keyframe flags in coded order; the list of cost requests
in the order the decision logic issued them (lazy singleCost order without a pool, the two batches of
slicetypeAnalyse with one), with their bDoSearch flags; every array of every such request; the cuTree offsets;
and with a VBV lookahead the planned costs.  Reference: source/encoder/slicetype.cpp:1982-2751 (slicetypeDecide),
:2846-3184 (slicetypeAnalyse), :3186-3321 (scenecut), :3484-3579 (slicetypePath), :3665-3766 (cuTree)."""
import numpy as np
import pytest

from oracle import orc
from x265_amd import api, synth

import decision_util as du

pytestmark = pytest.mark.gpu


def _compare(clip, chroma, depth, p, forced=None, ctx_kw=None, param_kw=None):
    N, H, W = clip.shape
    forced = forced or {}
    slices = p.get("lookaheadSlices", 0)
    look, dec = du.oracle_lookahead(clip, chroma, depth, p, lookahead_slices=slices, **(param_kw or {}))
    ctx = api.Context(W, H, depth=depth, bframes=p["bframes"], num_slots=N, bframe_bias=p["bFrameBias"], lookahead_slices=slices,
                      **(ctx_kw or {}))
    la = api.HostLookahead(ctx, api.LookaheadParam(**dict(p, bitDepth=depth)))
    try:
        ctx.set_mvcost(look.table, look.centre)
        want_order = du.drive(lambda i: dec.add_picture(i, 0, forced.get(i, 0)), dec.get_decided, dec.flush, N)
        got = du.drive(lambda i: la.add_picture(clip[i], chroma[i][0], chroma[i][1], 0, forced.get(i, 0)),
                       la.get_decided, la.flush, N)
        # decisions, in coded order
        assert [d.frame for d in got] == want_order
        for d in got:
            fr = look.frames[d.frame]
            have = (d.sliceType, d.bScenecut, d.bKeyframe, d.bLastMiniGopBFrame, d.leadingBframes)
            want = tuple(int(fr.i64(k)) for k in ("sliceType", "bScenecut", "bKeyframe", "bLastMiniGopBFrame", "leadingBframes"))
            assert have == want, "frame %d: %s, oracle %s" % (d.frame, have, want)
            assert bool(d.bIsFadeEnd) == orc.is_fade_end(fr), "fade end of frame %d" % d.frame
            have = (d.gopOffset, d.gopId, d.tempLayer)
            want = tuple(int(fr.i64(k)) for k in ("gopOffset", "gopId", "tempLayer"))
            assert have == want, "frame %d: place in its mini-GOP %s, oracle %s" % (d.frame, have, want)
        # the requests, in issue order
        t_want, t_got = dec.trace(), la.trace()
        assert t_got.shape == t_want.shape, "host issued %d requests, oracle %d" % (len(t_got), len(t_want))
        bad = np.nonzero((t_got != t_want).any(axis=1))[0]
        assert bad.size == 0, "first differing request #%d: host %s oracle %s" % (bad[0], t_got[bad[0]], t_want[bad[0]])
        # pre-analysis and every request's arrays
        for i in range(N):
            fr = look.frames[i]
            assert la.cost_est(i, 0, 0) == fr.i64("costEst", 0, 0)
            if p["bAdaptiveQuant"]:
                np.testing.assert_array_equal(la.inv_qscale(i), fr.inv_qscale(), err_msg="invQscaleFactor %d" % i)
                np.testing.assert_array_equal(la.qp_offsets(i), fr.qp_aq_offset(), err_msg="qpAqOffset %d" % i)
                if p["cuTree"]:
                    np.testing.assert_array_equal(la.qp_offsets(i, True), fr.qp_cutree_offset(), err_msg="qpCuTreeOffset %d" % i)
            if p["cuTree"]:
                np.testing.assert_array_equal(ctx.cutree_get_propagate_cost(i), fr.propagate_cost(), err_msg="propagateCost %d" % i)
            if p["bAdaptiveQuant"] and p.get("bHevcAq"):
                for which in (0, 1):
                    have = la.aq_layers(i, which)
                    assert len(have) == len(du.hevc_parts(W, H, p["qgSize"]))
                    for k, a in enumerate(have):
                        np.testing.assert_array_equal(a, orc.hevc_layer(fr, k, which), err_msg="layer %d of frame %d (%d)" % (k, i, which))
        if p["bAdaptiveQuant"] and p.get("bHevcAq") and p["cuTree"]:
            # computeCUTreeQpOffset did run: some frame's tree offsets left its AQ offsets
            assert any((la.aq_layers(i, 1)[-1] != la.aq_layers(i, 0)[-1]).any() for i in range(N))
        seen_mv = set()
        for (p0, p1, b, do0, do1, _batch) in t_want.tolist():
            d0, d1 = b - p0, p1 - b
            fr = look.frames[b]
            assert la.cost_est(b, d0, d1) == fr.i64("costEst", d0, d1), (p0, p1, b)
            assert la.cost_est_aq(b, d0, d1) == fr.i64("costEstAq", d0, d1), (p0, p1, b)
            lc, rs, (ce, ceaq, _) = ctx.get_costs(b, d0, d1)
            np.testing.assert_array_equal(lc, fr.lowres_costs(d0, d1), err_msg="lowresCosts %s" % ((p0, p1, b),))
            np.testing.assert_array_equal(rs, fr.row_satds(d0, d1), err_msg="rowSatds %s" % ((p0, p1, b),))
            for lst, dist, did in ((0, d0, do0), (1, d1, do1)):
                if did and (b, lst, dist) not in seen_mv:
                    seen_mv.add((b, lst, dist))
                    mv, mc = ctx.get_mvs(b, lst, dist)
                    np.testing.assert_array_equal(mv, fr.mvs(lst, dist), err_msg="L%d mvs %s" % (lst, (p0, p1, b)))
                    np.testing.assert_array_equal(mc, fr.mv_costs(lst, dist), err_msg="L%d mv costs %s" % (lst, (p0, p1, b)))
        for i in range(N):
            for d in range(1, p["bframes"] + 2):
                assert la.intra_mbs(i, d) == look.frames[i].i64("intraMbs", d), (i, d)
        if p["vbvBufferSize"] and p["lookaheadDepth"]:
            planned = 0
            for i in range(N):
                fr = look.frames[i]
                satd, types, ind_b = la.planned(i)
                wt = fr._view("plannedType", 0, 0, 251, np.int32)
                ws = fr._view("plannedSatd", 0, 0, 251, np.int64)
                n = int(np.argmax(wt == 0)) if (wt == 0).any() else 251
                assert types == wt[:n].tolist() and satd == ws[:n].tolist(), "planned costs of frame %d" % i
                assert ind_b == fr.i64("indB")
                planned += n
                rr = la.recalc_row_satds(i)
                wr = fr._view("recalcRowSatds", 0, 0, ctx.geom.heightInCU, np.int32)
                assert (rr is None) == (wr is None)
                if rr is not None:
                    np.testing.assert_array_equal(rr, wr, err_msg="frameCostRecalculate rows of frame %d" % i)
            assert planned > 0
        if p["poolWorkers"] and p["bFrameAdaptive"] == 2 and p.get("bPresearch", 1) and N > 8:
            assert la.presearch_hits() > 0           # single requests did find searches that ran ahead with a batch
        if not p.get("bPresearch", 1):
            assert la.presearch_hits() == 0
        _compare.fade_ends = [d.frame for d in got if d.bIsFadeEnd]
        return [d.sliceType for d in sorted(got, key=lambda d: d.frame)], [d.bScenecut for d in sorted(got, key=lambda d: d.frame)], t_want
    finally:
        la.close()
        ctx.close()
        dec.close()
        look.close()


CASES = [
    {},                                                       # preset medium, no pool: lazy request order
    {"poolWorkers": 16},                                      # pool > 12 workers: both batches, every time
    {"poolWorkers": 16, "bPresearch": 0},                     # the same without the searches queued ahead of single requests
    {"poolWorkers": 8},                                       # 4..12 workers: frame-cost batch only the first time
    {"poolWorkers": 2},                                       # < 4 workers: batches only the first time
    {"bFrameAdaptive": 1},
    {"bFrameAdaptive": 0, "bBPyramid": 0},
    {"cuTree": 0, "bEnableWeightedPred": 0, "bAdaptiveQuant": 0},
    {"bOpenGOP": 0, "keyframeMax": 12, "keyframeMin": 1},
    {"vbvBufferSize": 1000, "vbvMaxBitrate": 1000},
    {"vbvBufferSize": 1000, "vbvMaxBitrate": 1000, "bBPyramid": 0, "poolWorkers": 16},
    {"lookaheadDepth": 10, "bframes": 3, "bFrameBias": 20},
    {"bEnableTemporalSubLayers": 2},
    {"gopLookahead": 4, "keyframeMax": 14, "keyframeMin": 2},
    {"radl": 2, "bOpenGOP": 0, "keyframeMax": 10, "keyframeMin": 10},
    {"bframes": 0, "bFrameAdaptive": 0, "bBPyramid": 0},
    {"lookaheadDepth": 0, "bframes": 0, "bFrameAdaptive": 0, "bBPyramid": 0, "cuTree": 0},
    {"scenecutThreshold": 0},
    {"bEnableWeightedBiPred": 1, "fpsNum": 24000, "fpsDenom": 1001, "qCompress": 0.7},
    {"bAQMotion": 1},                                         # --aq-motion, lazy order
    {"bAQMotion": 1, "poolWorkers": 16},                      # ... with both batches (costs measured ahead must not go stale)
    {"bAQMotion": 1, "poolWorkers": 16, "vbvBufferSize": 1000, "vbvMaxBitrate": 1000, "bBPyramid": 0},
    {"bAQMotion": 1, "cuTree": 0},
    {"qgSize": 8},                                            # --qg-size 8: invQscaleFactor8x8, offsets per 8x8 source block
    {"qgSize": 8, "poolWorkers": 16, "vbvBufferSize": 1000, "vbvMaxBitrate": 1000},   # ... and frameCostRecalculate's mean of four
    {"qgSize": 8, "bFrameAdaptive": 1, "bEnableWeightedPred": 0},
    {"qgSize": 16, "bEnableFades": 1},
    {"bHevcAq": 1, "qpAdaptationRange": 1.5},                 # --hevc-aq: layers of offsets, computeCUTreeQpOffset
    {"bHevcAq": 1, "qgSize": 8, "poolWorkers": 16, "vbvBufferSize": 1000, "vbvMaxBitrate": 1000},
    {"bHevcAq": 1, "qgSize": 16, "vbvBufferSize": 1000, "vbvMaxBitrate": 1000, "bBPyramid": 0},
    {"bHevcAq": 1, "qgSize": 64, "bFrameAdaptive": 1},
    # --temporal-layers 3..5: hierarchical mini-GOPs of 4 / 8 / 16 pictures; the scene cut and the end of the clip
    # leave partial ones, which split into shorter orders and a plain remainder
    {"bEnableTemporalSubLayers": 3, "bframes": 3, "bFrameAdaptive": 0},
    {"bEnableTemporalSubLayers": 4, "bframes": 7, "bFrameAdaptive": 0, "poolWorkers": 16},
    {"bEnableTemporalSubLayers": 4, "bframes": 7, "bFrameAdaptive": 0, "bBPyramid": 0, "vbvBufferSize": 1000, "vbvMaxBitrate": 1000},
    {"bEnableTemporalSubLayers": 5, "bframes": 15, "bFrameAdaptive": 0},
    {"bEnableTemporalSubLayers": 5, "bframes": 15, "bFrameAdaptive": 0, "bOpenGOP": 0, "keyframeMax": 13, "keyframeMin": 1},
]


@pytest.mark.parametrize("kw", CASES, ids=lambda kw: ",".join("%s=%s" % kv for kv in kw.items()) or "medium")
def test_decisions_and_request_order(kw):
    p = dict(api.LookaheadParam.MEDIUM, **kw)
    N, W, H, depth = 34, 256, 160, 8
    clip = du.two_scene_clip(W, H, N, depth, seed=11)
    types, cuts, trace = _compare(clip, du.chroma_for(clip, depth, 12), depth, p)
    assert types[0] == 1
    if p["scenecutThreshold"] and p["lookaheadDepth"]:
        assert types[N // 2] in (1, 2) and cuts[N // 2] == 1
    assert len(trace) >= N - 1


@pytest.mark.parametrize("kw", [{}, {"poolWorkers": 16, "bframes": 3}, {"bFrameAdaptive": 0, "bBPyramid": 0, "cuTree": 0}],
                         ids=["lazy", "pool16-bframes3", "no-badapt"])
def test_histogram_scene_cut_decisions(kw):
    """--hist-scenecut: collectPictureStatistics on the device (x265gpu_hist_stats) feeding histBasedScenecut /
    detectHistBasedSceneChange (slicetype.cpp:3323-3482) in the host layer, against the oracle's statistics and its
    restatement of the same decision.  A darker scene with other chroma follows a brighter one; a one-picture flash
    and a slow brightening inside the scenes exercise the flash / fade / intensity-change branches."""
    p = dict(api.LookaheadParam.MEDIUM, bHistBasedSceneCut=1, **kw)
    N, W, H, depth = 30, 640, 368, 8
    a = synth.luma_clip(W, H, N, depth, seed=61, scene_cut=False)
    clip = np.empty_like(a)
    chroma = du.chroma_for(a, depth, 62)
    for i in range(N):
        f = a[i].astype(np.int32)
        if i >= 14:
            f = f[::-1, ::-1] // 2 + 10                       # the second scene: mirrored, darker, flatter
            chroma[i] = (np.clip(chroma[i][0].astype(np.int32) + 50, 0, 255).astype(np.uint8), chroma[i][1][::-1].copy())
        if i == 6:
            f = f + 90                                         # a flash
        if 20 <= i < 26:
            f = f + 3 * (i - 19)                               # a slow brightening
        clip[i] = np.clip(f, 0, 255).astype(np.uint8)
    types, cuts, trace = _compare(clip, chroma, depth, p)
    assert cuts[14] == 1 and types[14] in (1, 2)


def test_fades_weighted_prediction_in_decision_order():
    """A fade makes weightsAnalyse pick weights inside the requests the decision logic issues."""
    p = dict(api.LookaheadParam.MEDIUM)
    N, W, H, depth = 24, 256, 160, 8
    base = synth.luma_clip(W, H, N, depth, seed=21, scene_cut=False)
    clip = np.stack([np.clip(f.astype(np.int32) * (64 - 2 * i) // 64 + i, 0, 255).astype(np.uint8) for i, f in enumerate(base)])
    _compare(clip, du.chroma_for(clip, depth, 22), depth, p)


def test_forced_types_and_ten_bit():
    p = dict(api.LookaheadParam.MEDIUM)
    N, W, H, depth = 20, 256, 160, 10
    clip = du.two_scene_clip(W, H, N, depth, seed=31)
    types, _, _ = _compare(clip, du.chroma_for(clip, depth, 32), depth, p, forced={5: 2, 8: 3, 13: 1})
    assert types[5] == 2 and types[8] == 3 and types[13] == 1


def test_720p_pool_with_cooperative_slices():
    """x265's default with a pool: --lookahead-slices 8 (preset medium), so single requests run in cooperative slices
    while batched ones do not (slicetype.cpp:4312); searches queued ahead of single requests must follow suit."""
    p = dict(api.LookaheadParam.MEDIUM, poolWorkers=16, lookaheadSlices=8)
    N, W, H, depth = 30, 1280, 720, 8
    clip = du.two_scene_clip(W, H, N, depth, seed=51)
    _compare(clip, du.chroma_for(clip, depth, 52), depth, p)


@pytest.mark.parametrize("pool", [0, 16], ids=["lazy-no-pool", "batched-pool16"])
def test_1080p_60_frames_medium(pool):
    """The reference configuration: 60 frames of 1920x1080 8-bit, preset medium (bframes 4, b-adapt 2,
    rc-lookahead 20, B-pyramid, weightp, AQ mode 2, cuTree), whole-frame lookahead (--lookahead-slices 0), scene
    change at the clip midpoint; without a pool (the lazy request order) and with one of 16 workers (batches)."""
    p = dict(api.LookaheadParam.MEDIUM, poolWorkers=pool)
    N, W, H, depth = 60, 1920, 1080, 8
    clip = synth.luma_clip(W, H, N, depth)
    chroma = du.chroma_for(clip, depth, 7)
    types, cuts, trace = _compare(clip, chroma, depth, p)
    assert types[0] == 1 and types[-1] == 3
    assert any(t == 5 for t in types)


def test_4k_10bit_hme_decisions():
    """3840x2160 10-bit with --hme (quarter-res hexagon, half-res multi-hexagon): the decision loop over a short window."""
    p = dict(api.LookaheadParam.MEDIUM, lookaheadDepth=8, bframes=3)
    N, W, H, depth = 12, 3840, 2160, 10
    clip = synth.luma_clip(W, H, N, depth, seed=41)
    chroma = du.chroma_for(clip, depth, 42)
    hme = dict(hme=True, hme_methods=(orc.HEX, orc.UMH, orc.UMH), hme_range=(16, 32, 48))
    ctx_hme = dict(hme=True, hme_methods=(api.HEX_SEARCH, api.UMH_SEARCH, api.UMH_SEARCH), hme_range=(16, 32, 48))
    _compare(clip, chroma, depth, p, ctx_kw=ctx_hme, param_kw=hme)


@pytest.mark.timeout(900)
def test_4k_10bit_hme_rc_lookahead_40():
    """3840x2160 10-bit, --hme, rc-lookahead 40, bframes 4 over 48 frames: the window depth of the 4K configuration as
    it is quoted (slot count, request mix of the two batch loops, cuTree over 40 frames), not a short window."""
    p = dict(api.LookaheadParam.MEDIUM, lookaheadDepth=40, bframes=4)
    N, W, H, depth = 48, 3840, 2160, 10
    clip = synth.luma_clip(W, H, N, depth, seed=43)
    chroma = du.chroma_for(clip, depth, 44)
    hme = dict(hme=True, hme_methods=(orc.HEX, orc.UMH, orc.UMH), hme_range=(16, 32, 48))
    ctx_hme = dict(hme=True, hme_methods=(api.HEX_SEARCH, api.UMH_SEARCH, api.UMH_SEARCH), hme_range=(16, 32, 48))
    types, cuts, trace = _compare(clip, chroma, depth, p, ctx_kw=ctx_hme, param_kw=hme)
    assert len(types) == N and any(t == 5 for t in types)


def test_driver_refuses_what_x265_refuses():
    """rc-lookahead must exceed bframes (x265_check_params, param.cpp:1811): the decision reads frames[bframes + 1]."""
    with api.Context(256, 160, num_slots=4) as ctx:
        with pytest.raises(api.X265GpuError):
            api.HostLookahead(ctx, api.LookaheadParam(bframes=6, lookaheadDepth=5))
        api.HostLookahead(ctx, api.LookaheadParam(bframes=4, lookaheadDepth=5)).close()
        # three to five temporal sub-layers fix the B-frame count and turn adaptive placement off
        # (Encoder::configure, encoder.cpp:4303-4332); one layer and more than five do not reach the lookahead
        for kw in (dict(bEnableTemporalSubLayers=4, bframes=4, bFrameAdaptive=0), dict(bEnableTemporalSubLayers=3, bframes=3, bFrameAdaptive=2),
                   dict(bEnableTemporalSubLayers=1), dict(bEnableTemporalSubLayers=6, bframes=15, bFrameAdaptive=0)):
            with pytest.raises(api.X265GpuError):
                api.HostLookahead(ctx, api.LookaheadParam(**kw))
        api.HostLookahead(ctx, api.LookaheadParam(bEnableTemporalSubLayers=3, bframes=3, bFrameAdaptive=0)).close()


@pytest.mark.parametrize("qg", [8, 32])
def test_hevc_aq_layers_that_overhang_the_picture(qg):
    """368x180: the finest layer's doubled lowres grid has a row of parts below the picture (and the coarser layers'
    last rows are cut); computeCUTreeQpOffset divides 0 by 0 there, here as in the oracle."""
    p = dict(api.LookaheadParam.MEDIUM, bHevcAq=1, qgSize=qg, qpAdaptationRange=3.0, aqMode=3, poolWorkers=16,
             vbvBufferSize=1000, vbvMaxBitrate=1000)
    N, W, H, depth = 21, 368, 180, 10
    clip = du.two_scene_clip(W, H, N, depth, seed=9681)
    _compare(clip, du.chroma_for(clip, depth, 681), depth, p)


@pytest.mark.parametrize("kw", [{}, {"poolWorkers": 16, "bOpenGOP": 0}], ids=["lazy", "pool16-closed-gop"])
def test_fade_in_ends_on_a_keyframe(kw):
    """--fades: the picture variances of calcAdaptiveQuantFrame's tail come from the device; a fade-in of a second or
    more (here 16 pictures at 10 fps with rising contrast) ends on a picture that the fix-ups make a
    keyframe (slicetype.cpp:2042-2088, 2153-2155)."""
    p = dict(api.LookaheadParam.MEDIUM, bEnableFades=1, fpsNum=10, scenecutThreshold=0, **kw)
    N, W, H, depth = 40, 256, 160, 8
    base = synth.luma_clip(W, H, N, depth, seed=21, scene_cut=False).astype(np.float64)
    gain = np.ones(N)
    gain[:4] = np.linspace(0.3, 0.2, 4)
    gain[4:20] = np.linspace(0.2, 1.0, 16)
    gain[20:] = np.linspace(1.0, 0.4, N - 20)              # and a fade-out afterwards, which ends the run
    clip = 0.8 * base.mean(axis=0, keepdims=True) + 0.2 * base      # one scene that moves a little
    mean = clip.mean(axis=(1, 2), keepdims=True)
    clip = np.clip(mean + (clip - mean) * gain[:, None, None] * 2.0, 0, 255).astype(np.uint8)
    types, _, _ = _compare(clip, du.chroma_for(clip, depth, 22), depth, p)
    ends = _compare.fade_ends
    assert ends == [20], ends
    for f in ends:
        assert types[f] in (1, 2)
