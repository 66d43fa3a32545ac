"""GPU parity of the pixel primitives through the C-ABI against the oracle's C restatement of
x265's primitives (pu[].sad, pu[].satd, cu[].sa8d, luma_hpp / luma_vpp / luma_hvpp), bit-exact."""
import ctypes as C

import numpy as np
import pytest

from oracle import orc
from x265_amd import api, synth

pytestmark = pytest.mark.gpu

# the 25 luma partitions, source/common/primitives.h:54-65
PARTITIONS = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (8, 4), (4, 8), (16, 8), (8, 16), (32, 16), (16, 32),
              (64, 32), (32, 64), (16, 12), (12, 16), (16, 4), (4, 16), (32, 24), (24, 32), (32, 8), (8, 32),
              (64, 48), (48, 64), (64, 16), (16, 64)]


def _setup(depth, seed, kind="clip"):
    W, H, ctu = 320, 192, 64
    if kind == "clip":
        clip = synth.luma_clip(W, H, 2, depth, seed=seed)
    else:        # the buffer classes of the reference's harness: random, and the two extremes against each other
        rng = np.random.default_rng(seed)
        top = (1 << depth) - 1
        dt = np.uint8 if depth == 8 else np.uint16
        clip = [rng.integers(0, top + 1, (H, W)).astype(dt), rng.integers(0, top + 1, (H, W)).astype(dt)] \
            if kind == "random" else [np.full((H, W), top, dt), np.zeros((H, W), dt)]
    mx, my = ctu + 32, ctu + 16
    fw, fh = -(-W // ctu) * ctu, -(-H // ctu) * ctu
    pads = [np.ascontiguousarray(np.pad(f, ((my, my + fh - H), (mx, mx + fw - W)), mode="edge")) for f in clip]
    ctx = api.Context(W, H, depth=depth, num_slots=2, ctu=ctu, keep_full_res=True)
    ctx.lowres_init(0, clip[0])
    ctx.lowres_init(1, clip[1])
    return ctx, pads, mx, my, fw, fh


def _ptr(a, y, x):
    return C.c_void_p(a.ctypes.data + (y * a.shape[1] + x) * a.itemsize)


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("kind", ["clip", "random", "extremes"])
def test_sad_satd_all_partitions(depth, kind):
    ctx, pads, mx, my, fw, fh = _setup(depth, 31 + depth, kind)
    L = orc.lib(depth)
    try:
        rng = np.random.default_rng(5)
        jobs, want = [], []
        st = pads[0].shape[1]
        for (w, h) in PARTITIONS:
            for _ in range(6):
                x = int(rng.integers(0, (fw - w) // 4 + 1)) * 4
                y = int(rng.integers(0, (fh - h) // 4 + 1)) * 4
                dx, dy = int(rng.integers(-20, 21)), int(rng.integers(-20, 21))
                for op, fn in ((api.OP_SAD, L.orc_sad), (api.OP_SATD, L.orc_satd)):
                    jobs.append((x, y, w, h, 4 * dx, 4 * dy, op))
                    want.append(fn(w, h, _ptr(pads[0], my + y, mx + x), st, _ptr(pads[1], my + y + dy, mx + x + dx), st))
        got = ctx.block_compare(0, 1, jobs)
        np.testing.assert_array_equal(got, np.array(want, np.int32))
    finally:
        ctx.close()


@pytest.mark.parametrize("depth", [8, 10, 12])
def test_bipred_satd_all_partitions(depth):
    """SATD against the average of two predictions (the bidirectional candidate of puMotionEstimation): every
    partition, full-pel and fractional vectors in any mix, three pictures."""
    W, H, ctu = 320, 192, 64
    clip = synth.luma_clip(W, H, 3, depth, seed=70 + depth)
    mx, my = ctu + 32, ctu + 16
    pads = [np.ascontiguousarray(np.pad(f, ((my, my), (mx, mx)), mode="edge")) for f in clip]
    table, centre = orc.bitcost_table(depth)
    me0 = orc.FullResME(depth, table, centre, pads[1], pads[0], mx, my)
    me1 = orc.FullResME(depth, table, centre, pads[1], pads[2], mx, my)
    rng = np.random.default_rng(depth)
    jobs, want = [], []
    for (w, h) in PARTITIONS[1:]:
        for k in range(6):
            x = int(rng.integers(0, (W - w) // 4 + 1)) * 4
            y = int(rng.integers(0, (H - h) // 4 + 1)) * 4
            q0 = (int(rng.integers(-80, 81)), int(rng.integers(-80, 81)))
            q1 = (int(rng.integers(-80, 81)), int(rng.integers(-80, 81)))
            if k == 0:
                q0 = q1 = (0, 0)
            elif k == 1:
                q0 = (q0[0] & ~3, q0[1] & ~3)
            jobs.append((x, y, w, h, q0, q1))
            want.append(me0.bipred_satd(x, y, w, h, q0, me1, q1))
    with api.Context(W, H, depth=depth, num_slots=3, ctu=ctu, keep_full_res=True) as ctx:
        for k in range(3):
            ctx.lowres_init(k, clip[k])
        got = ctx.bipred_satd(1, 0, 2, jobs)
        np.testing.assert_array_equal(got, np.array(want, np.int32))
        same = ctx.bipred_satd(1, 0, 0, [(j[0], j[1], j[2], j[3], j[4], j[4]) for j in jobs])
        plain = ctx.block_compare(1, 0, [(j[0], j[1], j[2], j[3], j[4][0], j[4][1], api.OP_SATD) for j in jobs])
        np.testing.assert_array_equal(same, plain)           # averaging a prediction with itself changes nothing
        with pytest.raises(api.X265GpuError):
            ctx.bipred_satd(1, 0, 2, [(0, 0, 8, 8, (0, 0), (-800, 0))])
    me0.close()
    me1.close()


@pytest.mark.parametrize("depth", [8, 10])
def test_sa8d(depth):
    ctx, pads, mx, my, fw, fh = _setup(depth, 77, "random")
    L = orc.lib(depth)
    try:
        rng = np.random.default_rng(6)
        jobs, want = [], []
        st = pads[0].shape[1]
        for size in (8, 16, 32, 64):
            for _ in range(8):
                x = int(rng.integers(0, (fw - size) // 4 + 1)) * 4
                y = int(rng.integers(0, (fh - size) // 4 + 1)) * 4
                dx, dy = int(rng.integers(-12, 13)), int(rng.integers(-12, 13))
                jobs.append((x, y, size, size, 4 * dx, 4 * dy, api.OP_SA8D))
                want.append(L.orc_sa8d(size, _ptr(pads[0], my + y, mx + x), st, _ptr(pads[1], my + y + dy, mx + x + dx), st))
        np.testing.assert_array_equal(ctx.block_compare(0, 1, jobs), np.array(want, np.int32))
    finally:
        ctx.close()


@pytest.mark.parametrize("depth", [8, 10])
@pytest.mark.parametrize("kind", ["clip", "random", "extremes"])
def test_luma_interpolation(depth, kind):
    """luma_hpp / luma_vpp / luma_hvpp for every fraction pair, and the compares on interpolated blocks."""
    ctx, pads, mx, my, fw, fh = _setup(depth, 90 + depth, kind)
    if kind == "extremes":          # give the filter an edge to ring on
        ctx.close()
        W, H = 320, 192
        top = (1 << depth) - 1
        dt = np.uint8 if depth == 8 else np.uint16
        f = np.zeros((H, W), dt)
        f[:, ::3] = top
        f[::5, :] = top
        pads = [np.ascontiguousarray(np.pad(f, ((my, my + fh - H), (mx, mx + fw - W)), mode="edge"))] * 2
        ctx = api.Context(W, H, depth=depth, num_slots=2, ctu=64, keep_full_res=True)
        ctx.lowres_init(0, f)
        ctx.lowres_init(1, f)
    L = orc.lib(depth)
    try:
        st = pads[1].shape[1]
        rng = np.random.default_rng(8)
        jobs, want = [], []
        for (w, h) in [(8, 8), (16, 16), (64, 64), (12, 16), (4, 8), (32, 24), (16, 4)]:
            x = int(rng.integers(0, (fw - w) // 4 + 1)) * 4
            y = int(rng.integers(0, (fh - h) // 4 + 1)) * 4
            for xf in range(4):
                for yf in range(4):
                    if not (xf | yf):
                        continue
                    buf = np.zeros((h, w), pads[1].dtype)
                    src = _ptr(pads[1], my + y, mx + x)
                    if not yf:
                        L.orc_luma_hpp(w, h, src, st, orc.ptr(buf), w, xf)
                    elif not xf:
                        L.orc_luma_vpp(w, h, src, st, orc.ptr(buf), w, yf)
                    else:
                        L.orc_luma_hvpp(w, h, src, st, orc.ptr(buf), w, xf, yf)
                    np.testing.assert_array_equal(ctx.interp_luma(1, x, y, w, h, xf, yf), buf,
                                                  err_msg="%dx%d frac (%d,%d)" % (w, h, xf, yf))
                    for op, fn in ((api.OP_SAD, L.orc_sad), (api.OP_SATD, L.orc_satd)):
                        jobs.append((x, y, w, h, xf, yf, op))
                        want.append(fn(w, h, _ptr(pads[0], my + y, mx + x), st, orc.ptr(buf), w))
        np.testing.assert_array_equal(ctx.block_compare(0, 1, jobs), np.array(want, np.int32))
    finally:
        ctx.close()
