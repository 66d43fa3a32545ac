"""CPU: the oracle's successive-elimination search (X265_SEA) against a second derivation in numpy.

The numpy side is written from source/encoder/motion.cpp:1438-1591 and source/common/pixel.cpp:121-165 on its own:
block sums come from a summed-area table instead of the twelve running planes of
source/encoder/framefilter.cpp:731-831, and what the partition-dependent code of the reference works out at run time
(which sub-blocks give the DC terms, which plane, where each sum is read) is a table worked out by hand per partition."""
import numpy as np
import pytest

from oracle import orc
from x265_amd import synth

# partition -> (plane block size, [(encode sub-block x, y, w, h)], [(offset of the compared sum x, y)])
# The offsets of the lower two terms of the asymmetric horizontal partitions are horizontal: the reference scales
# its row delta by the stride for a fixed list of partitions only (motion.cpp:1545-1550), and those are not in it.
SEA_TABLE = {
    (64, 64): ((32, 32), [(0, 0, 32, 32), (32, 0, 32, 32), (0, 32, 32, 32), (32, 32, 32, 32)], [(0, 0), (32, 0), (0, 32), (32, 32)]),
    (32, 32): ((16, 16), [(0, 0, 16, 16), (16, 0, 16, 16), (0, 16, 16, 16), (16, 16, 16, 16)], [(0, 0), (16, 0), (0, 16), (16, 16)]),
    (16, 16): ((8, 8), [(0, 0, 8, 8), (8, 0, 8, 8), (0, 8, 8, 8), (8, 8, 8, 8)], [(0, 0), (8, 0), (0, 8), (8, 8)]),
    (8, 8): ((8, 8), [(0, 0, 8, 8)], [(0, 0)]),
    (32, 64): ((16, 16), [(0, 0, 32, 32), (0, 32, 32, 32)], [(0, 0), (0, 32)]),
    (16, 32): ((8, 8), [(0, 0, 16, 16), (0, 16, 16, 16)], [(0, 0), (0, 16)]),
    (8, 16): ((8, 8), [(0, 0, 8, 8), (0, 8, 8, 8)], [(0, 0), (0, 8)]),
    (64, 32): ((32, 32), [(0, 0, 32, 32), (32, 0, 32, 32)], [(0, 0), (32, 0)]),
    (32, 16): ((16, 16), [(0, 0, 16, 16), (16, 0, 16, 16)], [(0, 0), (16, 0)]),
    (16, 8): ((8, 8), [(0, 0, 8, 8), (8, 0, 8, 8)], [(0, 0), (8, 0)]),
    (64, 48): ((32, 24), [(0, 0, 32, 24), (32, 0, 32, 24), (0, 24, 32, 24), (32, 24, 32, 24)], [(0, 0), (32, 0), (24, 0), (56, 0)]),
    (64, 16): ((32, 8), [(0, 0, 32, 8), (32, 0, 32, 8), (0, 8, 32, 8), (32, 8, 32, 8)], [(0, 0), (32, 0), (8, 0), (40, 0)]),
    (48, 64): ((24, 32), [(0, 0, 24, 32), (24, 0, 24, 32), (0, 32, 24, 32), (24, 32, 24, 32)], [(0, 0), (24, 0), (0, 32), (24, 32)]),
    (16, 64): ((8, 32), [(0, 0, 8, 32), (8, 0, 8, 32), (0, 32, 8, 32), (8, 32, 8, 32)], [(0, 0), (8, 0), (0, 32), (8, 32)]),
    (32, 24): ((16, 12), [(0, 0, 16, 12), (16, 0, 16, 12), (0, 12, 16, 12), (16, 12, 16, 12)], [(0, 0), (16, 0), (12, 0), (28, 0)]),
    (24, 32): ((12, 16), [(0, 0, 12, 16), (12, 0, 12, 16), (0, 16, 12, 16), (12, 16, 12, 16)], [(0, 0), (12, 0), (0, 16), (12, 16)]),
    (16, 12): ((8, 8), [(0, 0, 16, 12)], [(0, 0)]),
    (12, 16): ((4, 4), [(0, 0, 12, 16)], [(0, 0)]),
    (16, 4): ((8, 8), [(0, 0, 16, 4)], [(0, 0)]),
    (4, 16): ((4, 4), [(0, 0, 4, 16)], [(0, 0)]),
}
PLANE_ORDER = [(32, 32), (32, 24), (32, 8), (24, 32), (16, 16), (16, 12), (16, 4), (12, 16), (8, 32), (8, 8), (4, 16), (4, 4)]


def block_sums(plane, bw, bh):
    """S[y, x] = sum of plane[y:y+bh, x:x+bw] wherever the block fits (0 elsewhere)"""
    sat = np.zeros((plane.shape[0] + 1, plane.shape[1] + 1), np.int64)
    sat[1:, 1:] = plane.astype(np.int64).cumsum(0).cumsum(1)
    out = np.zeros(plane.shape, np.int64)
    h, w = plane.shape
    out[:h - bh + 1, :w - bw + 1] = (sat[bh:, bw:] - sat[:-bh, bw:] - sat[bh:, :-bw] + sat[:-bh, :-bw])
    return out


def sea_numpy(fenc, ref, ox, oy, x, y, w, h, mvp, mvmin, mvmax, merange, table, centre, start_mv, start_cost):
    (bw, bh), enc_blocks, offs = SEA_TABLE[(w, h)]
    S = block_sums(ref, bw, bh)
    src = fenc[oy + y:oy + y + h, ox + x:ox + x + w].astype(np.int64)
    dc = [int(src[ey:ey + eh, ex:ex + ew].sum()) for (ex, ey, ew, eh) in enc_blocks]
    cost = lambda q: int(table[centre + q])                 # noqa: E731
    bmv, bcost = start_mv, start_cost
    min_x, min_y = max(start_mv[0] - merange, mvmin[0]), max(start_mv[1] - merange, mvmin[1])
    max_x, max_y = min(start_mv[0] + merange, mvmax[0]), min(start_mv[1] + merange, mvmax[1])
    width = (max_x - min_x + 3) & ~3

    def sad(mx, my):
        blk = ref[oy + y + my:oy + y + my + h, ox + x + mx:ox + x + mx + w].astype(np.int64)
        return int(np.abs(src - blk).sum())

    for ty in range(min_y, max_y + 1):
        ycost = cost(ty - mvp[1]) << 2
        if bcost <= ycost:
            continue
        thresh = bcost - ycost
        keep = []
        for i in range(width):
            mx = min_x + i
            ads = cost(4 * mx - mvp[0])
            for d, (sx, sy) in zip(dc, offs):
                ads += abs(d - int(S[oy + y + ty + sy, ox + x + mx + sx]))
            if ads < thresh:
                keep.append(mx)
        triples = len(keep) - len(keep) % 3 if len(keep) >= 3 else 0
        reduced = thresh
        for mx in keep[:triples]:
            c = sad(mx, ty) + cost(4 * mx - mvp[0])
            if c < reduced:
                reduced, bmv = c, (mx, ty)
        bcost = reduced + ycost
        for mx in keep[triples:]:
            c = sad(mx, ty) + ((cost(4 * mx - mvp[0]) + cost(4 * ty - mvp[1])) & 0xffff)
            if c < bcost:
                bcost, bmv = c, (mx, ty)
    return bmv, bcost


def _pictures(W, H, depth, seed):
    clip = synth.luma_clip(W, H, 2, depth, seed=seed)
    mx, my = 96, 80
    pads = [np.pad(f, ((my, my), (mx, mx)), mode="edge") for f in clip]
    return pads, mx, my


@pytest.mark.parametrize("depth", [8, 10])
def test_integral_planes_are_block_sums(oracle_built, depth):
    pads, mx, my = _pictures(192, 128, depth, seed=5)
    table, centre = orc.bitcost_table(depth)
    me = orc.FullResME(depth, table, centre, pads[1], pads[0], mx, my)
    me.build_integral()
    rows, stride = pads[0].shape
    for k, (bw, bh) in enumerate(PLANE_ORDER):
        want = block_sums(pads[0], bw, bh)
        got = me.integral_plane(k).astype(np.int64)
        # the first padded row stays zero (the running sums start there) and rows whose block reaches the last
        # padded row are left unfinished by the reference (framefilter.cpp:760-770)
        assert not got[0].any()
        np.testing.assert_array_equal(got[1:rows - bh, :stride - bw], want[1:rows - bh, :stride - bw], err_msg=str((bw, bh)))
    me.close()


@pytest.mark.parametrize("depth,merange", [(8, 12), (10, 9), (8, 25)])
def test_sea_search_against_numpy(oracle_built, depth, merange):
    W, H = 192, 128
    pads, ox, oy = _pictures(W, H, depth, seed=11 + merange)
    table, centre = orc.bitcost_table(depth)
    me = orc.FullResME(depth, table, centre, pads[1], pads[0], ox, oy)
    me.build_integral()
    rng = np.random.default_rng(merange)
    shapes = sorted(SEA_TABLE)
    moved = 0
    for n in range(3 * len(shapes)):
        w, h = shapes[n % len(shapes)]
        x = int(rng.integers(0, (W - w) // 4 + 1)) * 4
        y = int(rng.integers(0, (H - h) // 4 + 1)) * 4
        mvp = (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))
        lo = (max((mvp[0] >> 2) - merange, -x - ox + 36), max((mvp[1] >> 2) - merange, -y - oy + 36))
        hi = (min((mvp[0] >> 2) + merange, W - x - w + ox - 36), min((mvp[1] >> 2) + merange, H - y - h + oy - 36))
        start = (int(rng.integers(lo[0], hi[0] + 1)), int(rng.integers(lo[1], hi[1] + 1)))
        blk = pads[0][oy + y + start[1]:oy + y + start[1] + h, ox + x + start[0]:ox + x + start[0] + w].astype(np.int64)
        src = pads[1][oy + y:oy + y + h, ox + x:ox + x + w].astype(np.int64)
        start_cost = int(np.abs(src - blk).sum()) + ((int(table[centre + 4 * start[0] - mvp[0]]) + int(table[centre + 4 * start[1] - mvp[1]])) & 0xffff)
        want = sea_numpy(pads[1], pads[0], ox, oy, x, y, w, h, mvp, lo, hi, merange, table, centre, start, start_cost)
        got = me.sea_search_only(x, y, w, h, mvp, lo, hi, merange, start, start_cost)
        assert got == want, ((w, h), (x, y), mvp, start, got, want)
        moved += got[0] != start
    assert moved > len(shapes)
    me.close()
