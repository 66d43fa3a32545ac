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


def test_aq_frame_rejects_uncovered_modes():
    W, H = 64, 64
    y = np.zeros((H, W), np.uint8)
    u = v = np.zeros((H // 2, W // 2), np.uint8)
    with api.Context(W, H, depth=8, num_slots=1) as ctx:
        with pytest.raises(api.X265GpuError):
            ctx.aq_frame(y, u, v, 4, 1.0)                           # X265_AQ_EDGE
