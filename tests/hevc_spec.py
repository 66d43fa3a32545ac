"""Independent numpy formulations used to cross-check the C oracle.

Nothing here is derived from the oracle's code: the intra predictors follow the
equations of the HEVC specification (ITU-T H.265, clauses 8.4.4.2.3-8.4.4.2.6,
p[x][y] indexed column-first), the interpolation filter follows clause 8.5.3.3.3,
and the transforms are plain Hadamard matrix products.
"""
import numpy as np
from scipy.linalg import hadamard

H4 = hadamard(4).astype(np.int64)
H8 = hadamard(8).astype(np.int64)

INTRA_ANGLE = {2: 32, 3: 26, 4: 21, 5: 17, 6: 13, 7: 9, 8: 5, 9: 2, 10: 0, 11: -2, 12: -5, 13: -9, 14: -13,
               15: -17, 16: -21, 17: -26, 18: -32, 19: -26, 20: -21, 21: -17, 22: -13, 23: -9, 24: -5, 25: -2,
               26: 0, 27: 2, 28: 5, 29: 9, 30: 13, 31: 17, 32: 21, 33: 26, 34: 32}
INV_ANGLE = {11: -4096, 12: -1638, 13: -910, 14: -630, 15: -482, 16: -390, 17: -315, 18: -256, 19: -315,
             20: -390, 21: -482, 22: -630, 23: -910, 24: -1638, 25: -4096}
LUMA_TAPS = np.array([[0, 0, 0, 64, 0, 0, 0, 0], [-1, 4, -10, 58, 17, -5, 1, 0],
                      [-1, 4, -11, 40, 40, -11, 4, -1], [0, 1, -5, 17, 58, -10, 4, -1]], np.int64)


def sad(a, b):
    return int(np.abs(a.astype(np.int64) - b.astype(np.int64)).sum())


def satd(a, b):
    """Sum over 4x4 tiles of |H4 D H4|, halved per 4x4 (4- and 12-wide) or per 8x4 pair."""
    d = a.astype(np.int64) - b.astype(np.int64)
    h, w = d.shape
    tot = 0
    pair = (w % 8) == 0
    for y in range(0, h, 4):
        x = 0
        while x < w:
            s = int(np.abs(H4 @ d[y:y + 4, x:x + 4] @ H4).sum())
            if pair:
                s += int(np.abs(H4 @ d[y:y + 4, x + 4:x + 8] @ H4).sum())
                x += 8
            else:
                x += 4
            tot += s >> 1
    return tot


def sa8d(a, b):
    d = a.astype(np.int64) - b.astype(np.int64)
    n = d.shape[0]
    if n == 4:
        return int(np.abs(H4 @ d @ H4).sum()) >> 1
    if n == 8:
        return (int(np.abs(H8 @ d @ H8).sum()) + 2) >> 2
    tot = 0
    for y in range(0, n, 16):
        for x in range(0, n, 16):
            s = 0
            for yy in (0, 8):
                for xx in (0, 8):
                    s += int(np.abs(H8 @ d[y + yy:y + yy + 8, x + xx:x + xx + 8] @ H8).sum())
            tot += (s + 2) >> 2
    return tot


def lowres_planes(src, width, height):
    """Four half-res planes from a source that is valid up to row 2*height and column 2*width."""
    s = src.astype(np.int64)

    def ravg(a, b):
        return (a + b + 1) >> 1

    v01 = ravg(s[0:2 * height:2], s[1:2 * height + 1:2])
    v12 = ravg(s[1:2 * height + 1:2], s[2:2 * height + 2:2])
    cols = lambda v, o: v[:, o:o + 2 * width:2]   # noqa: E731
    p0 = ravg(cols(v01, 0), cols(v01, 1))
    ph = ravg(cols(v01, 1), cols(v01, 2))
    pv = ravg(cols(v12, 0), cols(v12, 1))
    pc = ravg(cols(v12, 1), cols(v12, 2))
    return p0, ph, pv, pc


def _neighbours(nb, n):
    """Splits the oracle's neighbour array into spec samples: corner, top p[x][-1], left p[-1][y]."""
    nb = np.asarray(nb, np.int64)
    return int(nb[0]), nb[1:2 * n + 1], nb[2 * n + 1:4 * n + 1]


def filter_neighbours(nb, n):
    """Clause 8.4.4.2.3 without strong filtering: [1 2 1]/4 along the L-shaped array, ends kept."""
    corner, top, left = _neighbours(nb, n)
    line = np.concatenate([left[::-1], [corner], top])        # bottom-left ... corner ... top-right
    out = line.copy()
    out[1:-1] = (line[:-2] + 2 * line[1:-1] + line[2:] + 2) >> 2
    m = 2 * n
    return np.concatenate([[out[m]], out[m + 1:], out[:m][::-1]])


def intra_pred(nb, n, mode, depth, edge_filter=True):
    """predSamples as an (n, n) array indexed [y][x]."""
    corner, top, left = _neighbours(nb, n)
    maxv = (1 << depth) - 1
    pred = np.zeros((n, n), np.int64)
    if mode == 1:   # DC, 8.4.4.2.5
        k = n.bit_length() - 1
        dc = (int(top[:n].sum()) + int(left[:n].sum()) + n) >> (k + 1)
        pred[:] = dc
        if edge_filter and n < 32:
            pred[0, 0] = (left[0] + 2 * dc + top[0] + 2) >> 2
            pred[0, 1:] = (top[1:n] + 3 * dc + 2) >> 2
            pred[1:, 0] = (left[1:n] + 3 * dc + 2) >> 2
        return pred
    if mode == 0:   # planar, 8.4.4.2.4
        k = n.bit_length() - 1
        for y in range(n):
            for x in range(n):
                pred[y, x] = ((n - 1 - x) * left[y] + (x + 1) * top[n] + (n - 1 - y) * top[x] +
                              (y + 1) * left[n] + n) >> (k + 1)
        return pred
    ang = INTRA_ANGLE[mode]
    vertical = mode >= 18
    main, side = (top, left) if vertical else (left, top)
    ref = {}
    for x in range(0, n + 1):
        ref[x] = corner if x == 0 else int(main[x - 1])
    last = (n * ang) >> 5
    if ang < 0 and last < -1:
        inv = INV_ANGLE[mode]
        for x in range(-1, last - 1, -1):
            idx = -1 + ((x * inv + 128) >> 8)
            ref[x] = corner if idx == -1 else int(side[idx])
    else:
        for x in range(n + 1, 2 * n + 1):
            ref[x] = int(main[x - 1])
    for a in range(n):          # a: coordinate along the prediction direction (y if vertical)
        idx = ((a + 1) * ang) >> 5
        fact = ((a + 1) * ang) & 31
        for b in range(n):
            if fact:
                v = ((32 - fact) * ref[b + idx + 1] + fact * ref[b + idx + 2] + 16) >> 5
            else:
                v = ref[b + idx + 1]
            if vertical:
                pred[a, b] = v
            else:
                pred[b, a] = v
    if edge_filter and n < 32 and mode in (10, 26):
        for k in range(n):
            v = int(main[0]) + ((int(side[k]) - corner) >> 1)
            v = min(max(v, 0), maxv)
            if vertical:
                pred[k, 0] = v
            else:
                pred[0, k] = v
    return pred


def luma_interp(plane, y0, x0, h, w, xfrac, yfrac, depth):
    """Clause 8.5.3.3.3.1 for a w x h block whose integer origin is (x0, y0); plane indexed [y][x]."""
    p = plane.astype(np.int64)
    shift1 = min(4, depth - 8)
    maxv = (1 << depth) - 1
    if xfrac == 0 and yfrac == 0:
        return p[y0:y0 + h, x0:x0 + w]
    if yfrac == 0:
        acc = sum(LUMA_TAPS[xfrac][k] * p[y0:y0 + h, x0 - 3 + k:x0 - 3 + k + w] for k in range(8))
        return np.clip((acc + 32) >> 6, 0, maxv)
    if xfrac == 0:
        acc = sum(LUMA_TAPS[yfrac][k] * p[y0 - 3 + k:y0 - 3 + k + h, x0:x0 + w] for k in range(8))
        return np.clip((acc + 32) >> 6, 0, maxv)
    inter = sum(LUMA_TAPS[xfrac][k] * p[y0 - 3:y0 + h + 4, x0 - 3 + k:x0 - 3 + k + w] for k in range(8)) >> shift1
    acc = sum(LUMA_TAPS[yfrac][k] * inter[k:k + h] for k in range(8)) >> 6
    shift = 14 - depth
    return np.clip((acc + (1 << (shift - 1))) >> shift, 0, maxv)
