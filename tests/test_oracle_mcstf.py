"""Pins the oracle's --mcstf motion search (oracle/orc_mcstf.c) against a plain numpy formulation:
whole error sums for every candidate, then the first minimum in the reference's candidate order
(source/common/temporalfilter.cpp:421-478, 642-897)."""
import numpy as np
import pytest

from oracle import orc

# taps 1..6 of the 1/16-pixel interpolation filter (source/common/temporalfilter.h:36-54)
FILT = np.array([
    [0, 0, 64, 0, 0, 0], [1, -3, 64, 4, -2, 0], [1, -6, 62, 9, -3, 1], [2, -8, 60, 14, -5, 1],
    [2, -9, 57, 19, -7, 2], [3, -10, 53, 24, -8, 2], [3, -11, 50, 29, -9, 2], [3, -11, 44, 35, -10, 3],
    [1, -7, 38, 38, -7, 1], [3, -10, 35, 44, -11, 3], [2, -9, 29, 50, -11, 3], [2, -8, 24, 53, -10, 3],
    [2, -7, 19, 57, -9, 2], [1, -5, 14, 60, -8, 2], [1, -3, 9, 62, -6, 1], [0, -2, 4, 64, -3, 1]], np.int64)


NP_PAD = 128          # numpy-side pictures carry this much border on every side


def np_error(src, buf, x, y, dx, dy, bs, depth, P=NP_PAD):
    """src / buf: 2-D int64 arrays indexed [row + P, col + P]"""
    o = src[P + y:P + y + bs, P + x:P + x + bs]
    if (dx | dy) & 15 == 0:
        b = buf[P + y + dy // 16:P + y + dy // 16 + bs, P + x + dx // 16:P + x + dx // 16 + bs]
        return int(((o - b) ** 2).sum())
    fx, fy = FILT[dx & 15], FILT[dy & 15]
    y0, x0 = P + y + (dy >> 4) - 2, P + x + (dx >> 4) - 2
    win = buf[y0:y0 + bs + 5, x0:x0 + bs + 5]
    h = sum(fx[k] * win[:, k:k + bs] for k in range(6))
    v = sum(fy[k] * h[k:k + bs, :] for k in range(6))
    v = np.clip((v + 2048) >> 12, 0, (1 << depth) - 1)
    return int(((v - o) ** 2).sum())


def np_first_min(cands, err, best, least):
    for mv, e in zip(cands, err):
        if e < least:
            best, least = mv, e
    return best, least


def np_level(src, buf, width, height, bs, task_rows, prev, factor, full_range, frac, depth, num_tasks):
    bw = width // bs
    mvs, out_err = {}, {}
    for task in range(num_tasks):
        row_start = task * 16
        if row_start > height:
            continue
        row_end = min(row_start + 16, height)
        by = row_start
        while by + bs <= row_end:
            for bx in range(0, bw * bs, bs):
                E = lambda mv: np_error(src, buf, bx, by, mv[0], mv[1], bs, depth)
                best, least = (0, 0), 2 ** 31 - 1
                if prev is not None:
                    cands = []
                    for py in (-1, 0, 1):
                        for px in (-1, 0, 1):
                            tx, ty = bx // (2 * bs) + px, by // (2 * bs) + py
                            if 0 <= tx < width // (2 * bs) and 0 <= ty < height // (2 * bs):
                                cands.append((prev[ty, tx, 0] * factor, prev[ty, tx, 1] * factor))
                    cands.append((0, 0))
                    best, least = np_first_min(cands, [E(c) for c in cands], best, least)
                cx, cy = int(best[0] / 16), int(best[1] / 16)       # C division truncates
                cands = [((cx + i) * 16, (cy + j) * 16) for j in range(-full_range, full_range + 1)
                         for i in range(-full_range, full_range + 1)]
                best, least = np_first_min(cands, [E(c) for c in cands], best, least)
                if frac:
                    cands = [(best[0] + i, best[1] + j) for j in range(-3, 4) for i in range(-3, 4)]
                    best, least = np_first_min(cands, [E(c) for c in cands], best, least)
                    if by != row_start:
                        best, least = np_first_min([mvs[(by - bs, bx)]], [E(mvs[(by - bs, bx)])], best, least)
                if bx > 0:
                    best, least = np_first_min([mvs[(by, bx - bs)]], [E(mvs[(by, bx - bs)])], best, least)
                mvs[(by, bx)] = (int(best[0]), int(best[1]))
                if frac:
                    o = src[NP_PAD + by:NP_PAD + by + bs, NP_PAD + bx:NP_PAD + bx + bs]
                    var = float((o * o).sum()) - float(o.sum()) * float(o.sum()) / (bs * bs)
                    out_err[(by, bx)] = int(20 * ((least + 5.0) / (var + 5.0)) + (least // (bs * bs)) // 50)
            by += bs
    return mvs, out_err


@pytest.mark.parametrize("depth", [8, 10])
def test_motion_error_matches_numpy(depth):
    rng = np.random.default_rng(depth)
    L = orc.lib(depth)
    P, dt = orc.MCSTF_PAD, orc.pixel_dtype(depth)
    for kind in ("random", "max", "min"):
        shape = (64 + 2 * P, 80 + 2 * P)
        mk = {"random": lambda: rng.integers(0, 1 << depth, shape), "max": lambda: np.full(shape, (1 << depth) - 1),
              "min": lambda: np.zeros(shape, np.int64)}[kind]
        src, buf = mk().astype(dt), (mk() if kind == "random" else rng.integers(0, 1 << depth, shape)).astype(dt)
        org = (P * shape[1] + P) * src.itemsize
        for bs in (8, 16):
            for _ in range(40):
                x, y = int(rng.integers(0, 80 - bs)), int(rng.integers(0, 64 - bs))
                dx, dy = (int(v) for v in rng.integers(-200, 200, 2))
                if rng.integers(0, 3) == 0:
                    dx, dy = dx & ~15, dy & ~15
                got = L.orc_mcstf_motion_error(src.ctypes.data + org, shape[1], buf.ctypes.data + org, x, y, dx, dy, bs,
                                               2 ** 31 - 1)
                assert got == np_error(src.astype(np.int64), buf.astype(np.int64), x, y, dx, dy, bs, depth, P), (kind, bs, dx, dy)


@pytest.mark.parametrize("depth,size", [(8, (96, 72)), (10, (112, 64))])
def test_pyramid_search_matches_numpy(depth, size):
    """All four levels on a small pair of pictures; 72 lines leave a half row task at the bottom of the last level."""
    W, H = size
    rng = np.random.default_rng(3 + depth)
    base = rng.integers(0, 1 << depth, (H + 40, W + 40)).astype(np.float64)
    k = np.ones(5) / 5                                                    # blur: gives the search a gradient to follow
    base = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 1, base)
    base = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 0, base)
    dt = orc.pixel_dtype(depth)
    cur = np.rint(base[20:20 + H, 20:20 + W]).astype(dt)
    ref = np.rint(0.5 * (base[18:18 + H, 23:23 + W] + base[18:18 + H, 24:24 + W])
                  + rng.normal(0, 1.0, (H, W))).clip(0, (1 << depth) - 1).astype(dt)
    param = orc.make_param(W, H, mcstf=True)
    L = orc.lib(depth)
    pics = []
    for i, f in enumerate((cur, ref)):
        lr = orc.Lowres(L, depth, param)
        lr.init(f, i)
        pics.append(orc.McstfPicture(lr, f))
    try:
        mvs, err, coarse = orc.mcstf_motion_search(pics[0], [pics[1]])
        geo = pics[0].lowres.geometry
        P = NP_PAD

        def padded(lr, kind):
            """a Lowres plane re-wrapped so that index [row + P, col + P] works like the padded picture's"""
            st, pad = geo["stride"], geo["padoffset"]
            if kind == "half":
                buf, lines, my, mx = lr.buffer(0), geo["lines"], geo["marginY"], geo["marginX"]
            else:
                # the quarter-res origin sits padoffset / 2 into its buffer: marginY rows of half the stride down
                # (source/common/lowres.cpp:173-185), although only marginY / 2 of them hold extended pixels
                buf, lines, my, mx, st = lr.lower_buffer(0), geo["lines"] // 2, geo["marginY"], geo["marginX"] // 2, st // 2
            a = buf.reshape(-1, st).astype(np.int64)
            out = np.zeros((a.shape[0] + 2 * P, st + 2 * P), np.int64)
            out[P - my:P - my + a.shape[0], P - mx:P - mx + st] = a
            return out

        levels = [
            (padded(pics[0].lowres, "quarter"), padded(pics[1].lowres, "quarter"), geo["width"] // 2, geo["lines"] // 2, H >> 2),
            (padded(pics[0].lowres, "half"), padded(pics[1].lowres, "half"), geo["width"], geo["lines"], H >> 1),
            (np.pad(cur, P, mode="edge").astype(np.int64), np.pad(ref, P, mode="edge").astype(np.int64), W, H, H),
        ]
        prev = None
        for lvl, (s, b, w, h, level_h) in enumerate(levels):
            want, _ = np_level(s, b, w, h, 16, 1, prev, 2, 3, False, depth, (level_h + 15) // 16)
            got = coarse[0, lvl]
            for (by, bx), mv in want.items():
                assert tuple(got[by // 16, bx // 16]) == mv, (lvl, by, bx)
            assert len(want) > 0
            prev = got
        s, b = levels[2][0], levels[2][1]
        want, want_err = np_level(s, b, W, H, 8, 2, coarse[0, 2], 1, 0, True, depth, (H + 15) // 16)
        for (by, bx), mv in want.items():
            assert tuple(mvs[0, by // 8, bx // 8]) == mv, (by, bx)
            assert err[0, by // 8, bx // 8] == want_err[(by, bx)], (by, bx)
        assert len(want) == (W // 8) * (H // 8)
        assert np.any(mvs[0] & 15)
    finally:
        for p in pics:
            p.lowres.close()
