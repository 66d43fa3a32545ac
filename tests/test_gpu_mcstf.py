"""GPU parity: the --mcstf motion search through the C-ABI against the CPU oracle, bit-exact."""
import numpy as np
import pytest

from oracle import orc
from x265_amd import api, synth

pytestmark = pytest.mark.gpu


def _clip(width, height, n, depth, seed):
    """A textured picture that pans by a fractional amount per frame plus noise: the search has real sub-pel work."""
    rng = np.random.default_rng(seed)
    base = synth.luma_clip(width + 64, height + 64, 1, depth, seed=seed)[0].astype(np.float64)
    frames = []
    for i in range(n):
        sx, sy = 1.37 * i, -0.81 * i
        x0, y0 = int(np.floor(sx)), int(np.floor(sy))
        fx, fy = sx - x0, sy - y0
        ys, xs = 32 + y0, 32 + x0
        a = base[ys:ys + height + 1, xs:xs + width + 1]
        f = (a[:-1, :-1] * (1 - fx) * (1 - fy) + a[:-1, 1:] * fx * (1 - fy) + a[1:, :-1] * (1 - fx) * fy + a[1:, 1:] * fx * fy)
        f = f + rng.normal(0, 1.5 * (1 << (depth - 8)), f.shape)
        frames.append(np.clip(np.rint(f), 0, (1 << depth) - 1).astype(orc.pixel_dtype(depth)))
    return frames


@pytest.mark.parametrize("depth,size", [(8, (384, 208)), (10, (320, 184)), (8, (424, 250)), (12, (256, 144))])
def test_mcstf_motion_search(depth, size):
    W, H = size
    N = 5
    clip = _clip(W, H, N, depth, seed=7 + depth)
    param = orc.make_param(W, H, mcstf=True)
    L = orc.lib(depth)
    pics = []
    for i, f in enumerate(clip):
        lr = orc.Lowres(L, depth, param)
        lr.init(f, i)
        pics.append(orc.McstfPicture(lr, f))
    ctx = api.Context(W, H, depth=depth, num_slots=N, mcstf=True, keep_full_res=True)
    try:
        for i, f in enumerate(clip):
            ctx.lowres_init(i, f, i)
        cur, refs = 2, [0, 1, 3, 4]                    # mcstfFrameRange 2: two pictures either side
        want = orc.mcstf_motion_search(pics[cur], [pics[r] for r in refs])
        got = ctx.mcstf_motion_search(cur, refs)
        for name, g, w in zip(("mvs", "error", "coarse"), got, want):
            np.testing.assert_array_equal(g, w, err_msg=name)
        mvs = want[0][:, :H // 8, :W // 8]
        assert np.any(mvs & 15), "the clip must exercise fractional vectors"
        assert len(np.unique(mvs.reshape(-1, 2), axis=0)) > 8
    finally:
        ctx.close()
        for p in pics:
            p.lowres.close()


def test_mcstf_motion_search_needs_the_planes():
    ctx = api.Context(320, 192, num_slots=2)
    try:
        with pytest.raises(api.X265GpuError):
            ctx.mcstf_motion_search(0, [1])
    finally:
        ctx.close()
