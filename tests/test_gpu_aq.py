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
