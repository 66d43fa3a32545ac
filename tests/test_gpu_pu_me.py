"""GPU parity of the full-resolution PU motion estimation against the CPU oracle, bit-exact."""
import numpy as np
import pytest

from oracle import orc
from x265_amd import api, synth

pytestmark = pytest.mark.gpu

# every luma PU shape of HEVC inter prediction: 2Nx2N, 2NxN, Nx2N and the AMP splits (source/common/common.h:266-279)
PU_SHAPES = [(8, 8), (8, 4), (4, 8), (16, 16), (16, 8), (8, 16), (16, 4), (16, 12), (4, 16), (12, 16),
             (32, 32), (32, 16), (16, 32), (32, 8), (32, 24), (8, 32), (24, 32),
             (64, 64), (64, 32), (32, 64), (64, 16), (64, 48), (16, 64), (48, 64)]


def _pictures(W, H, depth, ctu, seed):
    clip = synth.luma_clip(W, H, 2, depth, seed=seed)
    mx, my = ctu + 32, ctu + 16
    fw, fh = -(-W // ctu) * ctu, -(-H // ctu) * ctu
    pads = [np.pad(f, ((my, my + fh - H), (mx, mx + fw - W)), mode="edge") for f in clip]   # clamp = pad + extend
    return clip, pads, mx, my, fw, fh


def _jobs(rng, fw, fh, mx, my, merange, count, narrow=False):
    """narrow: a third of the jobs get a range one to three positions wide in x or y (what a picture edge leaves)"""
    jobs = []
    while len(jobs) < count:
        w, h = PU_SHAPES[rng.integers(len(PU_SHAPES))]
        x = int(rng.integers(0, (fw - w) // 4 + 1)) * 4
        y = int(rng.integers(0, (fh - h) // 4 + 1)) * 4
        mvp = (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))
        # Search::setSearchRange: a window of merange around the predictor, kept inside the padded picture
        lo = (max((mvp[0] >> 2) - merange, -x - mx + 24), max((mvp[1] >> 2) - merange, -y - my + 24))
        hi = (min((mvp[0] >> 2) + merange, fw - x - w + mx - 24), min((mvp[1] >> 2) + merange, fh - y - h + my - 24))
        if lo[0] > hi[0] or lo[1] > hi[1]:
            continue
        if narrow and rng.integers(0, 3) == 0:
            axis, width = int(rng.integers(0, 2)), int(rng.integers(0, 3))
            at = int(rng.integers(lo[axis], hi[axis] + 1))
            lo = tuple(at if k == axis else v for k, v in enumerate(lo))
            hi = tuple(min(at + width, hi[axis]) if k == axis else v for k, v in enumerate(hi))
        nc = int(rng.integers(0, 4))
        mvc = [(int(rng.integers(-60, 61)), int(rng.integers(-60, 61))) for _ in range(nc)]
        jobs.append((x, y, w, h, mvp, lo, hi, mvc))
    return jobs


def _run(depth, method, merange, subpel, count, seed, size=(416, 240), lam=None, content="synth", narrow=False):
    """content: "synth", "flat" (both pictures one value: no residual anywhere) or "half" (the right half flat)"""
    W, H = size
    ctu = 64
    clip, pads, mx, my, fw, fh = _pictures(W, H, depth, ctu, seed)
    if content != "synth":
        clip = clip.copy()
        clip[:, :, (0 if content == "flat" else W // 2):] = clip[0, H // 2, W // 2]
        pads = [np.pad(f, ((my, my + fh - H), (mx, mx + fw - W)), mode="edge") for f in clip]
    table, centre = orc.bitcost_table(depth, lam)
    ref = orc.FullResME(depth, table, centre, pads[1], pads[0], mx, my)
    ctx = api.Context(W, H, depth=depth, num_slots=2, ctu=ctu, keep_full_res=True)
    try:
        ctx.lowres_init(0, clip[0])
        ctx.lowres_init(1, clip[1])
        for k in range(2):
            plane, ox, oy = ctx.get_full_plane(k)
            assert (ox, oy) == (mx, my)
            np.testing.assert_array_equal(plane, pads[k], err_msg="padded full-resolution plane %d" % k)
        ctx.me_set_mvcost(table, centre)
        jobs = _jobs(np.random.default_rng(seed), fw, fh, mx, my, merange, count, narrow)
        got = ctx.pu_motion_estimate(1, 0, method, merange, subpel, jobs)
        for j, g in zip(jobs, got):
            want = ref.estimate(j[0], j[1], j[2], j[3], j[4], j[5], j[6], method, merange, subpel, j[7])
            assert g == want, (j, g, want)
    finally:
        ctx.close()
        ref.close()


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("method,merange,subpel", [
    (orc.STAR, 57, 3),        # --me star --merange 57 --subme 3
    (orc.HEX, 16, 2),
    (orc.DIA, 16, 1),
    (orc.STAR, 24, 7),
    (orc.HEX, 32, 5),
    (orc.UMH, 16, 2),         # --me umh, with the adaptive range from the candidate list
    (orc.UMH, 44, 3),
])
def test_pu_motion_estimate(depth, method, merange, subpel):
    _run(depth, method, merange, subpel, 96, seed=100 + depth + merange)


SEA_SHAPES = [s for s in PU_SHAPES if s not in ((8, 4), (4, 8), (32, 8), (8, 32))]


@pytest.mark.parametrize("depth,merange,subpel,seed", [(8, 16, 2, 1), (8, 57, 3, 2), (10, 24, 3, 3), (12, 9, 1, 4), (8, 100, 0, 5)])
def test_pu_successive_elimination(depth, merange, subpel, seed):
    """--me sea: the block sums are made on the device from the reference slot; the oracle builds the reference's
    twelve integral planes.  Windows stay 36 pixels inside the padding so that every sum the reference reads is one
    it has computed."""
    W, H, ctu = 416, 240, 64
    clip, pads, mx, my, fw, fh = _pictures(W, H, depth, ctu, 300 + seed)
    table, centre = orc.bitcost_table(depth)
    ref = orc.FullResME(depth, table, centre, pads[1], pads[0], mx, my)
    ref.build_integral()
    rng = np.random.default_rng(seed)
    jobs = []
    while len(jobs) < 80:
        w, h = SEA_SHAPES[rng.integers(len(SEA_SHAPES))]
        x = int(rng.integers(0, (fw - w) // 4 + 1)) * 4
        y = int(rng.integers(0, (fh - h) // 4 + 1)) * 4
        mvp = (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))
        lo = (max((mvp[0] >> 2) - merange, -x - mx + 36), max((mvp[1] >> 2) - merange, -y - my + 36))
        hi = (min((mvp[0] >> 2) + merange, fw - x - w + mx - 36), min((mvp[1] >> 2) + merange, fh - y - h + my - 36))
        if lo[0] > hi[0] or lo[1] > hi[1]:
            continue
        mvc = [(int(rng.integers(-60, 61)), int(rng.integers(-60, 61))) for _ in range(int(rng.integers(0, 3)))]
        jobs.append((x, y, w, h, mvp, lo, hi, mvc))
    with api.Context(W, H, depth=depth, num_slots=2, ctu=ctu, keep_full_res=True) as ctx:
        ctx.lowres_init(0, clip[0])
        ctx.lowres_init(1, clip[1])
        ctx.me_set_mvcost(table, centre)
        got = ctx.pu_motion_estimate(1, 0, api.SEA_SEARCH, merange, subpel, jobs)
        # the planes are kept per reference: a second call, and one after the reference was uploaded again
        again = ctx.pu_motion_estimate(1, 0, api.SEA_SEARCH, merange, subpel, jobs[:16])
        assert again == got[:16]
        ctx.lowres_init(0, clip[1])
        same = ctx.pu_motion_estimate(1, 0, api.SEA_SEARCH, merange, subpel, jobs[:16])
    for j, g in zip(jobs, got):
        want = ref.estimate(j[0], j[1], j[2], j[3], j[4], j[5], j[6], orc.SEA, merange, subpel, j[7])
        assert g == want, (j, g, want)
    ref.close()
    ref = orc.FullResME(depth, table, centre, pads[1], pads[1], mx, my)
    ref.build_integral()
    for j, g in zip(jobs[:16], same):
        want = ref.estimate(j[0], j[1], j[2], j[3], j[4], j[5], j[6], orc.SEA, merange, subpel, j[7])
        assert g == want, (j, g, want)
    ref.close()


@pytest.mark.parametrize("method", [orc.DIA, orc.HEX, orc.STAR])
def test_pu_start_point_without_residual_returns_at_once(method):
    """A start point with SAD 0 ends motionEstimate before the search (motion.cpp:1007-1011), and before the closing
    zero-vector check: in flat pictures a PU whose horizontal range is one position wide (as the picture edge leaves
    it for a PU in the padding) keeps its clipped predictor although (0,0) costs the same."""
    W, H, depth, ctu = 416, 336, 8, 64
    clip, pads, mx, my, fw, fh = _pictures(W, H, depth, ctu, 9)
    clip = np.full_like(clip, 100)
    pads = [np.full_like(p, 100) for p in pads]
    table, centre = orc.bitcost_table(depth, 1.0)
    ref = orc.FullResME(depth, table, centre, pads[1], pads[0], mx, my)
    with api.Context(W, H, depth=depth, num_slots=2, ctu=ctu, keep_full_res=True) as ctx:
        ctx.lowres_init(0, clip[0])
        ctx.lowres_init(1, clip[1])
        ctx.me_set_mvcost(table, centre)
        jobs = [(432, 336, 16, 8, (7, 4), (-9, -7), (-9, 7), [(7, 4), (-36, 4), (12, 4)]),
                (432, 336, 16, 16, (24, 0), (-9, -8), (-9, 7), []),
                (432, 352, 16, 16, (-36, 4), (-9, -7), (-9, 7), [])]
        got = ctx.pu_motion_estimate(1, 0, method, 8, 2, jobs)
        for j, g in zip(jobs, got):
            want = ref.estimate(j[0], j[1], j[2], j[3], j[4], j[5], j[6], method, 8, 2, j[7])
            assert g == want, (j, g, want)
        assert got[0][0] == (-36, 4) and got[1][0] == (-36, 0)
    ref.close()


def test_pu_motion_estimate_12bit():
    _run(12, orc.STAR, 57, 3, 64, seed=1212)
    _run(12, orc.UMH, 24, 5, 48, seed=1213)


def test_pu_exhaustive_search_and_lambda():
    """--me full on a small range, with a steeper MV cost (higher QP lambda)."""
    _run(8, orc.FULL, 6, 4, 24, seed=7, lam=9.5)


@pytest.mark.parametrize("method,merange", [(orc.STAR, 57), (orc.HEX, 24), (orc.UMH, 24), (orc.DIA, 16)])
def test_pu_slice_bounded_vertical_range(method, merange):
    """param->maxSlices > 1: Search::setSearchRange keeps vectors inside the slice, so the vertical range is a
    few rows and often excludes 0.  Every candidate row is range-checked, which leaves the closing clamp of
    motionEstimate (motion.cpp:1654-1659) nothing to do: the oracle run with maxSlices = 4 gives what the
    library gives with the caller's bounds alone."""
    depth, W, H, ctu = 8, 416, 256, 64
    clip, pads, mx, my, fw, fh = _pictures(W, H, depth, ctu, 77)
    table, centre = orc.bitcost_table(depth)
    ref = orc.FullResME(depth, table, centre, pads[1], pads[0], mx, my)
    rng = np.random.default_rng(merange)
    jobs = []
    while len(jobs) < 64:
        w, h = PU_SHAPES[rng.integers(len(PU_SHAPES))]
        x = int(rng.integers(0, (fw - w) // 4 + 1)) * 4
        y = int(rng.integers(0, (fh - h) // 4 + 1)) * 4
        top, bottom = (y // 64) * 64, (y // 64) * 64 + 64                   # slices of one CTU row
        mvp = (int(rng.integers(-40, 41)), int(rng.integers(-120, 121)))
        lo = (max((mvp[0] >> 2) - merange, -x - mx + 24), max((mvp[1] >> 2) - merange, top - y))
        hi = (min((mvp[0] >> 2) + merange, fw - x - w + mx - 24), min((mvp[1] >> 2) + merange, bottom - y - h))
        if lo[0] > hi[0] or lo[1] > hi[1]:
            continue
        mvc = [(int(rng.integers(-60, 61)), int(rng.integers(-200, 201))) for _ in range(int(rng.integers(0, 4)))]
        jobs.append((x, y, w, h, mvp, lo, hi, mvc))
    assert sum(1 for j in jobs if j[5][1] > 0 or j[6][1] < 0) > 5           # ranges that exclude the zero row
    with api.Context(W, H, depth=depth, num_slots=2, ctu=ctu, keep_full_res=True) as ctx:
        ctx.lowres_init(0, clip[0])
        ctx.lowres_init(1, clip[1])
        ctx.me_set_mvcost(table, centre)
        got = ctx.pu_motion_estimate(1, 0, method, merange, 3, jobs)
    for j, g in zip(jobs, got):
        want = ref.estimate(j[0], j[1], j[2], j[3], j[4], j[5], j[6], method, merange, 3, j[7], max_slices=4)
        assert g == want, (j, g, want)
        # (the fork's closing zero-vector override, motion.cpp:1762-1768, is the one step that ignores the range)
        assert g[0] == (0, 0) or j[5][1] * 4 <= g[0][1] <= j[6][1] * 4
    ref.close()


def test_pu_rejects_bad_jobs():
    ctx = api.Context(416, 240, num_slots=2, keep_full_res=True)
    try:
        table, centre = orc.bitcost_table(8)
        with pytest.raises(api.X265GpuError):                    # no table yet
            ctx.pu_motion_estimate(0, 1, api.STAR_SEARCH, 57, 3, [(0, 0, 8, 8, (0, 0), (-4, -4), (4, 4), [])])
        ctx.me_set_mvcost(table, centre)
        for bad in [(0, 0, 6, 8, (0, 0), (-4, -4), (4, 4), []),              # width not a multiple of 4
                    (448, 0, 8, 8, (0, 0), (-4, -4), (4, 4), []),            # outside the picture
                    (0, 0, 8, 8, (0, 0), (-400, -4), (4, 4), [])]:           # vectors leave the padded plane
            with pytest.raises(api.X265GpuError):
                ctx.pu_motion_estimate(0, 1, api.STAR_SEARCH, 57, 3, [bad])
        ctx.lowres_init(0, np.zeros((240, 416), np.uint8))
        ctx.lowres_init(1, np.zeros((240, 416), np.uint8))
        ctx.pu_motion_estimate(0, 1, api.SEA_SEARCH, 16, 2, [(64, 64, 8, 8, (0, 0), (-4, -4), (4, 4), [])])
        for w, h in [(8, 4), (4, 8), (32, 8), (8, 32)]:          # the elimination test of these reads outside the PU
            with pytest.raises(api.X265GpuError):
                ctx.pu_motion_estimate(0, 1, api.SEA_SEARCH, 16, 2, [(64, 64, w, h, (0, 0), (-4, -4), (4, 4), [])])
        with pytest.raises(api.X265GpuError):                    # window wider than the survivor masks
            ctx.pu_motion_estimate(0, 1, api.SEA_SEARCH, 200, 2, [(64, 64, 8, 8, (0, 0), (-4, -4), (4, 4), [])])
        with pytest.raises(api.X265GpuError):                    # block sums would be read past the padded picture
            ctx.pu_motion_estimate(0, 1, api.SEA_SEARCH, 16, 2, [(384, 64, 32, 24, (0, 0), (-4, -4), (110, 4), [])])
    finally:
        ctx.close()


@pytest.mark.parametrize("depth", [8, 10])
def test_area_diamond_search(depth):
    """MotionEstimate::diamondSearch (the isMVP search of puMotionEstimation): zero predictor, range 32,
    full-pel result.  The reference's relative COST_MV_X4 doubles the origin inside the range; the
    vectors below move, so that branch is exercised with a non-zero origin."""
    W, H, ctu = 416, 240, 64
    clip, pads, mx, my, fw, fh = _pictures(W, H, depth, ctu, seed=9)
    table, centre = orc.bitcost_table(depth)
    ref = orc.FullResME(depth, table, centre, pads[1], pads[0], mx, my)
    with api.Context(W, H, depth=depth, num_slots=2, ctu=ctu, keep_full_res=True) as ctx:
        ctx.lowres_init(0, clip[0])
        ctx.lowres_init(1, clip[1])
        ctx.me_set_mvcost(table, centre)
        rng = np.random.default_rng(31)
        jobs = []
        for _ in range(128):
            w, h = PU_SHAPES[rng.integers(len(PU_SHAPES))]
            x = int(rng.integers(0, (fw - w) // 4 + 1)) * 4
            y = int(rng.integers(0, (fh - h) // 4 + 1)) * 4
            mvp = (0, 0) if rng.integers(3) else (int(rng.integers(-12, 13)), int(rng.integers(-12, 13)))
            r = 32
            lo = (max(-r, -x - 8), max(-r, -y - 8))                     # setSearchRange also clips near the picture edge
            hi = (min(r, fw - x - w + 8), min(r, fh - y - h + 8))
            jobs.append((x, y, w, h, mvp, lo, hi, []))
        got = ctx.pu_motion_estimate(1, 0, api.AREA_DIAMOND_SEARCH, 32, 0, jobs)
        moved = 0
        for j, g in zip(jobs, got):
            want = ref.diamond_search(j[0], j[1], j[2], j[3], j[4], j[5], j[6])
            assert g == want, (j, g, want)
            moved += int(max(abs(g[0][0]), abs(g[0][1])) >= 8)
        assert moved > 0
    ref.close()
