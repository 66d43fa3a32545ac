"""GPU parity: the HIP path through the C-ABI against the CPU oracle, bit-exact."""
import ctypes as C

import numpy as np
import pytest

from oracle import orc
from x265_amd import api, synth
from x265_amd.lookahead import CostEstimateGroup, LowresState, PreLookaheadGroup

pytestmark = pytest.mark.gpu


def _setup(width, height, nframes, depth, with_aq=False, seed=1234, bframes=4, clip=None, **search):
    """`search` (search=, hme=, hme_methods=, hme_range=) goes to the oracle's param and the context alike."""
    if clip is None:
        clip = synth.luma_clip(width, height, nframes, depth, seed=seed)
    # an odd size reaches x265's lookahead padded to the minimum CU size with the edge samples repeated
    # (Encoder::configure, PicYuv): that picture is what the oracle gets, the library gets the raw one
    ow, oh, oclip = width, height, clip
    if (width | height) & 1:
        ow, oh = (width + 7) // 8 * 8, (height + 7) // 8 * 8
        oclip = np.pad(np.asarray(clip), ((0, 0), (0, oh - height), (0, ow - width)), mode="edge")
    param = orc.make_param(ow, oh, bframes=bframes, **search)
    ref = orc.Lookahead(depth, param, with_aq=with_aq)
    aq = None
    if with_aq:
        rng = np.random.default_rng(seed + 1)
    for f in oclip:
        q = None
        if with_aq:
            tmp = orc.Lowres(ref.L, depth, param)
            n = tmp.ncu()
            tmp.close()
            q = rng.integers(128, 513, size=n).astype(np.int32)
            aq = (aq or []) + [q]
        ref.add_frame(f, q)
    ctx = api.Context(width, height, depth=depth, bframes=bframes, num_slots=nframes, **search)
    ctx.set_mvcost(ref.table, ref.centre)
    frames = [LowresState(i, i, ctx.D) for i in range(nframes)]
    PreLookaheadGroup(ctx).process_tasks(frames, clip, aq)
    return clip, ref, ctx, frames


def _check_request(ref, ctx, frames, p0, p1, b):
    d0, d1 = b - p0, p1 - b
    fr = ref.frames[b]
    lc, rs, (ce, ceaq, _) = ctx.get_costs(frames[b].slot, d0, d1)
    mv0, c0 = ctx.get_mvs(frames[b].slot, 0, d0)
    np.testing.assert_array_equal(mv0, fr.mvs(0, d0), err_msg="L0 mvs %s" % ((p0, p1, b),))
    np.testing.assert_array_equal(c0, fr.mv_costs(0, d0), err_msg="L0 mv costs %s" % ((p0, p1, b),))
    if d1:
        mv1, c1 = ctx.get_mvs(frames[b].slot, 1, d1)
        np.testing.assert_array_equal(mv1, fr.mvs(1, d1), err_msg="L1 mvs %s" % ((p0, p1, b),))
        np.testing.assert_array_equal(c1, fr.mv_costs(1, d1), err_msg="L1 mv costs %s" % ((p0, p1, b),))
    if ctx.cfg.bEnableHME:
        for lst, dist in ((0, d0), (1, d1)) if d1 else ((0, d0),):
            mv4, c4 = ctx.get_hme_mvs(frames[b].slot, lst, dist)
            np.testing.assert_array_equal(mv4, fr.lower_mvs(lst, dist), err_msg="HME L%d mvs %s" % (lst, (p0, p1, b)))
            np.testing.assert_array_equal(c4, fr.lower_mv_costs(lst, dist),
                                          err_msg="HME L%d costs %s" % (lst, (p0, p1, b)))
    np.testing.assert_array_equal(lc, fr.lowres_costs(d0, d1), err_msg="lowresCosts %s" % ((p0, p1, b),))
    np.testing.assert_array_equal(rs, fr.row_satds(d0, d1), err_msg="rowSatds %s" % ((p0, p1, b),))
    assert ce == fr.i64("costEst", d0, d1)
    assert ceaq == fr.i64("costEstAq", d0, d1)


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("size", [(320, 192), (424, 242)])
def test_lowres_planes_and_intra(depth, size):
    W, H = size
    clip, ref, ctx, frames = _setup(W, H, 3, depth)
    try:
        for i in range(3):
            for p in range(4):
                np.testing.assert_array_equal(ctx.get_plane(i, p), ref.frames[i].buffer(p),
                                              err_msg="frame %d plane %d" % (i, p))
            cost, mode = ctx.get_intra(i)
            np.testing.assert_array_equal(cost, ref.frames[i].intra_cost())
            np.testing.assert_array_equal(mode, ref.frames[i].intra_mode())
            lc, rs, (ce, ceaq, _) = ctx.get_costs(i, 0, 0)
            np.testing.assert_array_equal(lc, ref.frames[i].lowres_costs(0, 0))
            np.testing.assert_array_equal(rs, ref.frames[i].row_satds(0, 0))
            assert ce == ref.frames[i].i64("costEst", 0, 0)
            assert ceaq == ref.frames[i].i64("costEstAq", 0, 0)
    finally:
        ctx.close()
        ref.close()


@pytest.mark.parametrize("size", [(321, 193), (337, 192), (320, 209), (33, 65)])
@pytest.mark.parametrize("hme", [False, True])
def test_odd_picture_sizes_are_padded_to_the_minimum_cu_size(size, hme):
    """A picture one sample wider (taller) than a multiple of 16 gains a column (row) of 8x8 lowres blocks when x265
    pads it to the minimum CU size of 8 (Encoder::configure; PicYuv repeats the last column / row into the padding).
    The library gets the raw planes and the raw size; the oracle the picture x265's lookahead would see: padded, with
    the edge samples repeated.  Planes, intra estimate and searches must agree."""
    W, H = size
    if hme and (W < 64 or H < 64):
        pytest.skip("quarter-res level needs a larger picture")
    PW, PH = (W + 7) // 8 * 8, (H + 7) // 8 * 8
    N, depth = 3, 8
    raw = synth.luma_clip(W, H, N, depth, seed=71)
    padded = np.pad(raw, ((0, 0), (0, PH - H), (0, PW - W)), mode="edge")
    kw = dict(hme=True) if hme else {}
    ref = orc.Lookahead(depth, orc.make_param(PW, PH, **kw))
    for f in padded:
        ref.add_frame(f)
    ctx = api.Context(W, H, depth=depth, num_slots=N, **kw)
    try:
        assert (ctx.geom.widthInCU, ctx.geom.heightInCU) == ((PW // 2 + 7) >> 3, (PH // 2 + 7) >> 3)
        ctx.set_mvcost(ref.table, ref.centre)
        frames = [LowresState(i, i, ctx.D) for i in range(N)]
        PreLookaheadGroup(ctx).process_tasks(frames, raw)
        for i in range(N):
            for p in range(8 if hme else 4):
                want = ref.frames[i].buffer(p) if p < 4 else ref.frames[i].lower_buffer(p - 4)
                np.testing.assert_array_equal(ctx.get_plane(i, p), want, err_msg="frame %d plane %d" % (i, p))
            cost, mode = ctx.get_intra(i)
            np.testing.assert_array_equal(cost, ref.frames[i].intra_cost())
            np.testing.assert_array_equal(mode, ref.frames[i].intra_mode())
        est = CostEstimateGroup(ctx, frames)
        for (p0, p1, b) in [(0, 1, 1), (0, 2, 1), (1, 2, 2)]:
            assert est.single_cost(p0, p1, b) == ref.frame_cost(p0, p1, b), (p0, p1, b)
            _check_request(ref, ctx, frames, p0, p1, b)
        with pytest.raises(api.X265GpuError):
            ctx.aq_energy(raw[0], raw[0][::2, ::2], raw[0][::2, ::2])       # 4:2:0 has no odd sizes
    finally:
        ctx.close()
        ref.close()


@pytest.mark.parametrize("depth", [8, 10])
def test_scale_factors_after_the_intra_estimate(depth):
    """The intra estimate's per-block costs do not depend on AQ; its AQ-weighted row and frame sums do.  Factors
    installed after the estimate was launched (also a second set, and their removal) must show in those sums."""
    W, H = 392, 216
    clip = synth.luma_clip(W, H, 1, depth, seed=23)
    rng = np.random.default_rng(24)
    with api.Context(W, H, depth=depth, num_slots=1) as ctx:
        n = ctx.geom.cuCount
        ctx.lowres_init(0, clip[0])
        ctx.intra_estimate([0])
        for inv in (rng.integers(128, 513, size=n).astype(np.int32), rng.integers(64, 300, size=n).astype(np.int32), None):
            ref = orc.Lookahead(depth, orc.make_param(W, H), with_aq=inv is not None)
            try:
                fr = ref.add_frame(clip[0], inv) if inv is not None else ref.add_frame(clip[0])
                ctx.set_aq(0, inv)
                lc, rs, (ce, ceaq, _) = ctx.get_costs(0, 0, 0)
                np.testing.assert_array_equal(ctx.get_intra(0)[0], fr.intra_cost())
                np.testing.assert_array_equal(lc, fr.lowres_costs(0, 0))
                np.testing.assert_array_equal(rs, fr.row_satds(0, 0))
                assert (ce, ceaq) == (fr.i64("costEst", 0, 0), fr.i64("costEstAq", 0, 0))
            finally:
                ref.close()


def test_two_host_threads_on_one_context():
    """Calls on one context take turns under the context's lock: two host threads, each preparing and reading back
    its own pictures (shared staging buffers, job lists and streams underneath), both get the oracle's numbers."""
    import threading
    W, H, N = 392, 216, 8
    clip = synth.luma_clip(W, H, N, 8, seed=31)
    ref = orc.Lookahead(8, orc.make_param(W, H))
    for f in clip:
        ref.add_frame(f)
    got, errors = {}, []
    with api.Context(W, H, depth=8, num_slots=N) as ctx:
        def worker(slots):
            try:
                for rep in range(3):
                    for i in slots:
                        ctx.lowres_init(i, clip[i], poc=i)
                        ctx.intra_estimate([i])
                        got[i] = ([ctx.get_plane(i, p) for p in range(4)], ctx.get_intra(i), ctx.get_costs(i, 0, 0))
            except Exception as e:      # noqa: BLE001 - reported by the parent thread
                errors.append(e)
        threads = [threading.Thread(target=worker, args=(range(k, N, 2),)) for k in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    try:
        assert not errors, errors
        for i in range(N):
            planes, (cost, mode), (lc, rs, (ce, ceaq, _)) = got[i]
            fr = ref.frames[i]
            for p in range(4):
                np.testing.assert_array_equal(planes[p], fr.buffer(p), err_msg="frame %d plane %d" % (i, p))
            np.testing.assert_array_equal(cost, fr.intra_cost())
            np.testing.assert_array_equal(mode, fr.intra_mode())
            np.testing.assert_array_equal(lc, fr.lowres_costs(0, 0))
            np.testing.assert_array_equal(rs, fr.row_satds(0, 0))
            assert (ce, ceaq) == (fr.i64("costEst", 0, 0), fr.i64("costEstAq", 0, 0))
    finally:
        ref.close()


@pytest.mark.parametrize("depth", [8, 10])
def test_mcstf_downscale_filter(depth):
    """--mcstf: rounded 2x2 mean instead of cascaded averages, and quarter-res planes without HME."""
    W, H = 424, 242
    clip, ref, ctx, frames = _setup(W, H, 2, depth, mcstf=True)
    try:
        for i in range(2):
            for p in range(4):
                np.testing.assert_array_equal(ctx.get_plane(i, p), ref.frames[i].buffer(p), err_msg="plane %d" % p)
                np.testing.assert_array_equal(ctx.get_plane(i, 4 + p), ref.frames[i].lower_buffer(p),
                                              err_msg="quarter-res plane %d" % p)
            cost, mode = ctx.get_intra(i)
            np.testing.assert_array_equal(cost, ref.frames[i].intra_cost())
        est = CostEstimateGroup(ctx, frames)
        assert est.single_cost(0, 1, 1) == ref.frame_cost(0, 1, 1, False)
        _check_request(ref, ctx, frames, 0, 1, 1)
    finally:
        ctx.close()
        ref.close()


@pytest.mark.parametrize("depth,with_aq", [(8, False), (8, True), (10, True)])
def test_single_cost_sequence(depth, with_aq):
    """Lazy request order (no pool): every singleCost is its own batch."""
    W, H, N = 384, 208, 8
    clip, ref, ctx, frames = _setup(W, H, N, depth, with_aq=with_aq)
    try:
        est = CostEstimateGroup(ctx, frames)
        seq = [(0, 1, 1), (0, 2, 2), (0, 2, 1), (1, 2, 2), (0, 3, 3), (0, 3, 1), (0, 3, 2), (1, 3, 2),
               (3, 4, 4), (3, 5, 4), (3, 7, 5), (4, 7, 7), (2, 7, 7), (3, 4, 4), (0, 2, 1)]
        for (p0, p1, b) in seq:
            for pen in (False, True):
                assert est.single_cost(p0, p1, b, pen) == ref.frame_cost(p0, p1, b, pen), (p0, p1, b, pen)
            _check_request(ref, ctx, frames, p0, p1, b)
            assert frames[b].intra_mbs[b - p0] == ref.frames[b].i64("intraMbs", b - p0)
    finally:
        ctx.close()
        ref.close()


def test_batch_order_matches_sequential():
    """slicetypeAnalyse batch 1 + batch 2 (slicetype.cpp:2925-2988) issued as two device batches."""
    W, H, N = 320, 192, 8
    clip, ref, ctx, frames = _setup(W, H, N, 8)
    try:
        est = CostEstimateGroup(ctx, frames)
        bframes = 4
        batch1 = []
        for b in range(2, N):
            for i in range(1, bframes + 2):
                p0 = b - i
                if p0 < 0:
                    continue
                p1 = b + i
                if p1 >= N:
                    p1 = b
                batch1.append((p0, p1, b))
        for r in batch1:
            est.add(*r)
        est.finish_batch()
        for (p0, p1, b) in batch1:          # the oracle runs them one after another in the same order
            ref.frame_cost(p0, p1, b)
        batch2 = []
        for b in range(2, N):
            for i in range(1, bframes + 2):
                if b < i:
                    continue
                for j in range(0, bframes + 1):
                    if b + j >= N:
                        break
                    batch2.append((b - i, b + j, b))
        for r in batch2:
            est.add(*r)
        est.finish_batch()
        for (p0, p1, b) in batch2:
            want = ref.frame_cost(p0, p1, b)
            assert frames[b].cost_est[(b - p0, p1 - b)] == want, (p0, p1, b)
            _check_request(ref, ctx, frames, p0, p1, b)
    finally:
        ctx.close()
        ref.close()


SEQ = [(0, 1, 1), (0, 2, 2), (0, 2, 1), (1, 2, 2), (0, 3, 3), (0, 3, 1), (1, 3, 2), (3, 4, 4), (3, 5, 4), (2, 5, 5)]


def _run_sequence(ref, ctx, frames, seq=SEQ):
    est = CostEstimateGroup(ctx, frames)
    for (p0, p1, b) in seq:
        assert est.single_cost(p0, p1, b) == ref.frame_cost(p0, p1, b), (p0, p1, b)
        _check_request(ref, ctx, frames, p0, p1, b)


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("search", [orc.DIA, orc.UMH])
def test_lowres_search_methods(depth, search):
    """searchMethod other than the hard-wired HEX (x265gpu_config.searchMethod): DIA and UMH walks."""
    clip, ref, ctx, frames = _setup(384, 208, 6, depth, seed=77, search=search)
    try:
        _run_sequence(ref, ctx, frames)
    finally:
        ctx.close()
        ref.close()


@pytest.mark.parametrize("depth,kw", [(8, {}), (10, {}), (8, dict(search=orc.UMH)),
                                      (8, dict(hme=True, hme_methods=(orc.HEX, orc.UMH, orc.UMH), hme_range=(16, 32, 48)))],
                         ids=["hex8", "hex10", "umh8", "hme8"])
@pytest.mark.parametrize("staged_jobs", ["0", "100000"], ids=["one-wait", "as-they-come"])
def test_flow_search_both_ways_of_taking_neighbour_vectors(monkeypatch, staged_jobs, depth, kw):
    """The single-launch search takes a block's neighbour vectors in one wait (launches of many searches) or as the
    dependent chain produces them (launches of few, X265GPU_FLOW_STAGED_JOBS sets the limit): the same results
    either way, whatever the launch size.  The clip has a scene cut, across which neighbouring vectors differ and
    late ones are fetched and measured in a pass of their own."""
    monkeypatch.setenv("X265GPU_FLOW_STAGED_JOBS", staged_jobs)
    clip, ref, ctx, frames = _setup(400, 240, 8, depth, seed=4242, **kw)
    try:
        _run_sequence(ref, ctx, frames)
    finally:
        ctx.close()
        ref.close()


def test_multi_hexagon_search_moves_its_centre_three_times():
    """A multi-hexagon search with a short range prices vectors up to three ranges from its start (the crosses move the
    centre, the hexagon grid moves it again, the closing hexagon walk once more): a quarter-res level with range 8 whose
    predictor lies ~30 pixels out reaches past the MV-cost entries the kernel keeps in LDS.  (Found by the randomised
    sweep: one block's cost in 14 000 rounds.)"""
    clip, ref, ctx, frames = _setup(616, 287, 4, 12, seed=995066814, bframes=3, hme=True,
                                    hme_methods=(orc.UMH, orc.HEX, orc.HEX), hme_range=(8, 16, 48))
    try:
        _run_sequence(ref, ctx, frames, [(0, 3, 3), (0, 1, 1), (1, 3, 2), (0, 2, 2)])
    finally:
        ctx.close()
        ref.close()


def test_mvcost_table_must_reach_1024_entries_each_way():
    """The search kernels keep the table's entries for MVD -199..200 in LDS and read further ones in place: the
    library takes no table shorter than +-1024 entries (x265's spans +-2^17)."""
    table, centre = api.bitcost_table(8)
    with api.Context(256, 160, num_slots=2) as ctx:
        with pytest.raises(api.X265GpuError):
            ctx.set_mvcost(table[centre - 600:centre + 601], 600)
        ctx.set_mvcost(table[centre - 1024:centre + 1025], 1024)


@pytest.mark.parametrize("depth,methods,ranges", [
    (8, (orc.HEX, orc.UMH, orc.UMH), (16, 32, 48)),        # x265's defaults for --hme
    (10, (orc.HEX, orc.UMH, orc.UMH), (16, 32, 48)),
    (8, (orc.UMH, orc.HEX, orc.UMH), (24, 16, 48)),
    (8, (orc.DIA, orc.DIA, orc.UMH), (16, 16, 48)),
])
def test_hme_two_level(depth, methods, ranges):
    """--hme: quarter-res search first, its vector doubled as a fifth predictor of the half-res search
    (slicetype.cpp:4351-4360, 4456-4462)."""
    clip, ref, ctx, frames = _setup(640, 368, 6, depth, seed=5, hme=True, hme_methods=methods, hme_range=ranges)
    try:
        for i in range(2):
            for p in range(4):
                np.testing.assert_array_equal(ctx.get_plane(i, 4 + p), ref.frames[i].lower_buffer(p))
        _run_sequence(ref, ctx, frames)
    finally:
        ctx.close()
        ref.close()


def test_weighted_prediction():
    """weightp: the host guess loop over x265gpu_weight_cost / x265gpu_weight_planes, then searches against
    the weighted planes (slicetype.cpp:870-971, 4302-4304, 4391).  The clip fades, so weights are found."""
    W, H, N, depth = 384, 208, 6, 8
    base = synth.luma_clip(W, H, N, depth, seed=9)[0]
    rng = np.random.default_rng(3)
    clip = [np.clip(np.roll(base, i, axis=1).astype(np.int32) * (64 - 6 * i) // 64 + 2 * i + rng.integers(-1, 2, base.shape),
                    0, 255).astype(np.uint8) for i in range(N)]        # one picture, slow pan, fading
    param = orc.make_param(W, H, weightp=True)
    ref = orc.Lookahead(depth, param)
    ctx = api.Context(W, H, depth=depth, num_slots=N)
    try:
        frames = [LowresState(i, i, ctx.D) for i in range(N)]
        for i, f in enumerate(clip):
            lr = ref.add_frame(f)
            v = f.astype(np.int64)
            ssum = int(v.sum())
            ssd = int((v * v).sum()) - ssum * ssum // v.size
            wsum, wssd = lr.wp()
            wsum[0], wssd[0] = ssum, ssd
            frames[i].wp_sum, frames[i].wp_ssd = ssum, ssd
        ctx.set_mvcost(ref.table, ref.centre)
        PreLookaheadGroup(ctx).process_tasks(frames, clip)
        est = CostEstimateGroup(ctx, frames, weightp=True)
        nweighted = 0
        for (p0, p1, b) in [(0, 1, 1), (0, 2, 2), (0, 2, 1), (1, 3, 3), (1, 3, 2), (0, 5, 5), (2, 5, 3), (3, 4, 4), (2, 4, 4)]:
            assert est.single_cost(p0, p1, b) == ref.frame_cost(p0, p1, b), (p0, p1, b)
            _check_request(ref, ctx, frames, p0, p1, b)
            want = bool(ref.frames[b].i64("isWeighted", b - p0))
            assert frames[b].weighted.get(b - p0, False) == want, (p0, p1, b)
            nweighted += want
        assert nweighted >= 5          # the fade must actually exercise the weighted path
    finally:
        ctx.close()
        ref.close()


@pytest.mark.parametrize("mode", ["rows", "diag", "flow"])
def test_search_launch_modes(monkeypatch, mode):
    """X265GPU_SEARCH_MODE: "flow" (the default: one launch, blocks in diagonal order, every vector its own ready
    flag), "diag" (one launch per anti-diagonal) and "rows" (one launch, rows handing their progress to the row
    above) give the same results."""
    monkeypatch.setenv("X265GPU_SEARCH_MODE", mode)
    clip, ref, ctx, frames = _setup(448, 256, 6, 8, seed=21)
    try:
        _run_sequence(ref, ctx, frames)
    finally:
        ctx.close()
        ref.close()
    clip, ref, ctx, frames = _setup(640, 368, 5, 10, seed=22, hme=True)
    try:
        _run_sequence(ref, ctx, frames, SEQ[:7])
    finally:
        ctx.close()
        ref.close()


@pytest.mark.parametrize("after", [0, 2])
def test_flow_search_wait_that_runs_out_is_run_again_per_diagonal(monkeypatch, after):
    """The flow search's bounded wait: with the error word raised ahead of a launch (X265GPU_TEST_FLOW_FAULT, a test
    hook) every wave of that launch that has to wait gives up and searches with made-up predictors.  The call that
    next waits for the stream must notice, run the estimate calls made since the last check again with a launch per
    anti-diagonal, and deliver the same results as ever; later batches take that mode by themselves."""
    monkeypatch.setenv("X265GPU_SEARCH_MODE", "flow")
    monkeypatch.setenv("X265GPU_TEST_FLOW_FAULT", str(after))
    clip, ref, ctx, frames = _setup(448, 256, 6, 8, seed=23)
    try:
        _run_sequence(ref, ctx, frames)
        assert b"run again" in ctx.L.x265gpu_last_error(ctx.h)
    finally:
        ctx.close()
        ref.close()


@pytest.mark.parametrize("depth,slices", [(8, 3), (10, 2), (8, 8)])
def test_cooperative_slices(depth, slices):
    """--lookahead-slices: a singleCost outside batch mode runs each slice as its own bottom-up walk
    (slicetype.cpp:4248-4276, 4312-4343); requests of a batch stay whole-frame."""
    W, H, N = 448, 528, 5                        # 33 block rows: 3 x 11, 2 x 16 (+1), or 8 -> 10 rows x 3 (+3)
    clip, ref, ctx, frames = _setup(W, H, N, depth, seed=31, lookahead_slices=slices)
    whole = orc.Lookahead(depth, orc.make_param(W, H))
    try:
        rows = orc.C.c_int()
        n = ref.L.orc_coop_slices(orc.C.byref(ref.param), ctx.geom.heightInCU, orc.C.byref(rows))
        assert (ctx.geom.numCoopSlices, ctx.geom.numRowsPerSlice) == (n, rows.value)
        assert n == {3: 3, 2: 2, 8: 3}[slices]
        for f in clip:
            whole.add_frame(f)
        est = CostEstimateGroup(ctx, frames)
        differs = False
        for (p0, p1, b) in [(0, 1, 1), (0, 2, 1), (1, 2, 2), (0, 3, 3), (1, 3, 2), (3, 4, 4)]:
            assert est.single_cost(p0, p1, b) == ref.frame_cost(p0, p1, b, single=True), (p0, p1, b)
            _check_request(ref, ctx, frames, p0, p1, b)
            whole.frame_cost(p0, p1, b)
            differs |= not np.array_equal(ref.frames[b].mvs(0, b - p0), whole.frames[b].mvs(0, b - p0))
        assert differs                           # the slice boundaries must actually change some predictors
        # the same context in batch mode: whole frames
        batch = CostEstimateGroup(ctx, frames)
        for r in [(0, 4, 2), (2, 4, 3), (0, 4, 4)]:
            batch.add(*r)
        batch.finish_batch()
        for (p0, p1, b) in [(0, 4, 2), (2, 4, 3), (0, 4, 4)]:
            assert frames[b].cost_est[(b - p0, p1 - b)] == ref.frame_cost(p0, p1, b), (p0, p1, b)
            _check_request(ref, ctx, frames, p0, p1, b)
    finally:
        ctx.close()
        ref.close()
        whole.close()


def test_cooperative_slices_single_launch_mode(monkeypatch):
    monkeypatch.setenv("X265GPU_SEARCH_MODE", "rows")
    clip, ref, ctx, frames = _setup(448, 528, 4, 8, seed=32, lookahead_slices=3)
    try:
        est = CostEstimateGroup(ctx, frames)
        for (p0, p1, b) in [(0, 1, 1), (0, 2, 1), (1, 3, 2)]:
            assert est.single_cost(p0, p1, b) == ref.frame_cost(p0, p1, b, single=True), (p0, p1, b)
            _check_request(ref, ctx, frames, p0, p1, b)
    finally:
        ctx.close()
        ref.close()


def test_cooperative_slices_with_hme_refused():
    """x265's slice tasks read quarter-res vectors across slice boundaries while other tasks write them."""
    clip, ref, ctx, frames = _setup(640, 528, 2, 8, hme=True, lookahead_slices=3)
    try:
        with pytest.raises(api.X265GpuError):
            CostEstimateGroup(ctx, frames).single_cost(0, 1, 1)
        assert ref.frame_cost(0, 1, 1, single=True) == -1
    finally:
        ctx.close()
        ref.close()


def test_encoder_slices_do_not_change_the_lowres_search():
    """param->maxSlices > 1: the oracle applies motionEstimate's row clamp (motion.cpp:1654-1659); the device
    path has none because no lowres search can leave the row range.  Multi-hexagon, small frame: most blocks
    sit near a border."""
    clip, ref, ctx, frames = _setup(256, 160, 5, 8, seed=41, search=orc.UMH, max_slices=4)
    try:
        _run_sequence(ref, ctx, frames, SEQ[:7])
    finally:
        ctx.close()
        ref.close()


def test_intra_on_the_main_stream(monkeypatch):
    """X265GPU_INTRA_STREAM=0: the intra estimate queued on the context's own stream instead of beside the searches."""
    monkeypatch.setenv("X265GPU_INTRA_STREAM", "0")
    clip, ref, ctx, frames = _setup(384, 208, 5, 8, with_aq=True, seed=51)
    try:
        for i in range(5):
            cost, mode = ctx.get_intra(i)
            np.testing.assert_array_equal(cost, ref.frames[i].intra_cost())
            np.testing.assert_array_equal(mode, ref.frames[i].intra_mode())
        _run_sequence(ref, ctx, frames, SEQ[:6])
    finally:
        ctx.close()
        ref.close()


def _hard_clip(width, height, nframes, depth, kind, seed):
    """Content at the edges of the arithmetic: full-range noise (largest SAD / SATD, long walks), flat extremes
    (zero residuals, early exits), and a fast pan with a hard luminance step (vectors at the range limits)."""
    rng = np.random.default_rng(seed)
    top = (1 << depth) - 1
    dt = np.uint8 if depth == 8 else np.uint16
    if kind == "noise":
        return rng.integers(0, top + 1, (nframes, height, width)).astype(dt)
    if kind == "flat":
        vals = [0, top, top, 0, top // 2, 0, top, top]
        out = np.stack([np.full((height, width), vals[i % len(vals)], dt) for i in range(nframes)])
        out[:, height // 3:height // 2, width // 4:width // 2] = top     # one patch so that something moves
        return out
    base = rng.integers(0, top + 1, (height + 64, width + 40 * nframes + 64)).astype(dt)
    base[:, ::37] = top
    base[::29, :] = 0
    return np.stack([base[(7 * i) % 32:(7 * i) % 32 + height, 23 * i:23 * i + width] for i in range(nframes)])


@pytest.mark.parametrize("depth", [8, 10, 12])
@pytest.mark.parametrize("kind", ["noise", "flat", "pan"])
def test_extreme_content(depth, kind):
    W, H, N = 320, 192, 6
    clip = _hard_clip(W, H, N, depth, kind, seed=depth * 7 + len(kind))
    _, ref, ctx, frames = _setup(W, H, N, depth, with_aq=True, seed=3, clip=clip)
    try:
        for i in range(N):
            cost, mode = ctx.get_intra(i)
            np.testing.assert_array_equal(cost, ref.frames[i].intra_cost())
            np.testing.assert_array_equal(mode, ref.frames[i].intra_mode())
        _run_sequence(ref, ctx, frames)
    finally:
        ctx.close()
        ref.close()


@pytest.mark.parametrize("kind", ["noise", "pan"])
def test_extreme_content_hme_umh_12bit(kind):
    W, H, N = 640, 368, 5
    clip = _hard_clip(W, H, N, 12, kind, seed=19)
    _, ref, ctx, frames = _setup(W, H, N, 12, seed=3, clip=clip, hme=True)
    try:
        _run_sequence(ref, ctx, frames, SEQ[:7])
    finally:
        ctx.close()
        ref.close()


@pytest.mark.parametrize("width,height,depth,hme", [
    (1920, 1080, 8, False),      # the benchmark's configuration
    (3840, 2160, 10, False),
    (3840, 2160, 10, True),      # 4K Main10 with --hme: hexagon on the quarter-res, multi-hexagon on the half-res planes
    (7680, 4320, 10, False),
    (7680, 4320, 10, True),      # 8K Main10 with --hme
])
def test_full_size_against_oracle(width, height, depth, hme):
    """The benchmark sizes, three frames, against the oracle directly (a request takes it under a second)."""
    clip, ref, ctx, frames = _setup(width, height, 3, depth, seed=5, hme=hme)
    try:
        for p in range(4):
            np.testing.assert_array_equal(ctx.get_plane(1, p), ref.frames[1].buffer(p), err_msg="plane %d" % p)
        cost, mode = ctx.get_intra(2)
        np.testing.assert_array_equal(cost, ref.frames[2].intra_cost())
        np.testing.assert_array_equal(mode, ref.frames[2].intra_mode())
        _run_sequence(ref, ctx, frames, [(0, 1, 1), (0, 2, 1), (1, 2, 2), (0, 2, 2)])
    finally:
        ctx.close()
        ref.close()


def test_hme_odd_block_grid_wild_predictor():
    """39 half-res block columns: the index arithmetic of slicetype.cpp:4397 ((cuY / 2) * widthInCU / 2 against a
    quarter-res grid 20 wide) hands a block the doubled vector of a block elsewhere in the frame, which can point
    far past the margin.  The reference measures it unclipped, as a flat offset into its four-plane buffer, and
    so must the device (found by tests/fuzz_gpu_parity.py, round 314 of seed 1)."""
    W, H, N, depth = 624, 284, 4, 10
    clip = _hard_clip(W, H, N, depth, "pan", 997691383)
    _, ref, ctx, frames = _setup(W, H, N, depth, seed=3, bframes=2, clip=clip, hme=True,
                                 hme_methods=(orc.UMH, orc.HEX, orc.UMH), hme_range=(24, 16, 48))
    try:
        _run_sequence(ref, ctx, frames, [(0, 1, 1), (0, 2, 2), (0, 3, 1), (1, 3, 2), (0, 3, 2)])
    finally:
        ctx.close()
        ref.close()


def test_duplicate_search_in_one_batch_is_refused():
    """Two requests of one call that both search (frame 2, list 0, distance 2) would race on one vector array."""
    clip, ref, ctx, frames = _setup(320, 192, 4, 8, seed=5)
    try:
        with pytest.raises(api.X265GpuError):
            ctx.estimate([(0, 2, 2, 0, 2, 2, 1, 0), (0, 2, 3, 0, 2, 3, 1, 1)])
        # the same pair is fine once only one of them searches list 0
        ctx.estimate([(0, 2, 2, 0, 2, 2, 1, 0), (0, 2, 3, 0, 2, 3, 0, 1)])
        ref.frame_cost(0, 2, 2)
        ref.frame_cost(0, 3, 2)
        np.testing.assert_array_equal(ctx.get_mvs(2, 0, 2)[0], ref.frames[2].mvs(0, 2))
        np.testing.assert_array_equal(ctx.get_mvs(2, 1, 1)[0], ref.frames[2].mvs(1, 1))
    finally:
        ctx.close()
        ref.close()


def test_set_aq_copies_the_callers_array():
    """x265gpu_lowres_set_aq must have taken its copy when it returns, also from page-locked memory where the
    upload is truly asynchronous: the caller overwrites the array right after the call."""
    clip, ref, ctx, frames = _setup(320, 192, 2, 8, seed=6, with_aq=True)
    try:
        n = ctx.geom.cuCount
        pinned = ctx.host_alloc(C.c_int32, n)
        want = np.asarray(ref.frames[1].inv_qscale()).copy()
        view = np.frombuffer(pinned, np.int32, n)
        for _ in range(4):                       # keep the stream busy so that the upload is queued behind work
            ctx.intra_estimate([0, 1])
        view[:] = want
        ctx.set_aq(1, view)
        view[:] = 1                              # the caller reuses its array for the next frame
        ctx.estimate([(0, 1, 1, 0, 1, 1, 1, 0)])
        ref.frame_cost(0, 1, 1)
        _, _, (ce, ceaq, _) = ctx.get_costs(1, 1, 0)
        assert ceaq == ref.frames[1].i64("costEstAq", 1, 0) and ce == ref.frames[1].i64("costEst", 1, 0)
        ctx.host_free(pinned)
    finally:
        ctx.close()
        ref.close()
