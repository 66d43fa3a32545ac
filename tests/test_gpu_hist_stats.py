"""GPU parity of the histogram scene-cut picture statistics against the oracle, bit-exact."""
import numpy as np
import pytest

from oracle import orc
from x265_amd import api, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("size", [(416, 240), (1920, 1080), (362, 202), (640, 360)])
def test_collect_picture_statistics(depth, size):
    W, H = size
    rng = np.random.default_rng(W * 3 + depth)
    dt = np.uint8 if depth == 8 else np.uint16
    top = (1 << depth) - 1
    y = synth.luma_clip(W, H, 1, depth, seed=W)[0]
    u = rng.integers(top // 4, top // 2, (H // 2, W // 2)).astype(dt)
    v = rng.integers(0, top + 1, (H // 2, W // 2)).astype(dt)
    v[: H // 6] = top // 3                                             # a flat band: zero variance blocks
    param = orc.make_param(W, H)
    lr = orc.Lowres(orc.lib(depth), depth, param)
    lr.init(y, 0)
    want = api.hist_stats_dict(orc.collect_picture_statistics(lr, y, u, v))
    with api.Context(W, H, depth=depth, num_slots=1) as ctx:
        ctx.lowres_init(0, y)
        got = ctx.hist_stats(0, y, u, v)
    lr.close()
    for k in ("picHistogram", "averageIntensityPerSegment", "averageIntensity"):
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    assert got["picAvgVariance"] == want["picAvgVariance"]
    # every sample lands in exactly one bin: counts are (n + 1) << 4 per bin
    n_luma = (W // 4) * (H // 4)
    assert int((got["picHistogram"][:, :, 0, :] >> 4).sum()) - 16 * 256 == n_luma
