"""Seeded synthetic luma clips for tests and the benchmark.

A smooth two-dimensional sinusoid field panned by (3, 2) pixels per frame, a
+-12 uniform texture panned with it, a flat 128x128 square moving (11, 5)
pixels per frame, +-2 per-pixel noise, and a hard scene change (mirrored,
inverted field) at the clip midpoint.  8-bit values are 0..255; deeper clips
are the same field scaled up with LSB noise.  Only luma is produced: the
lookahead path reads chroma only for AQ variance, which stays on the host.
"""
import numpy as np


def luma_clip(width, height, frames, depth=8, seed=1234, scene_cut=True):
    """Returns an array (frames, height, width) of uint8 / uint16."""
    rng = np.random.default_rng(seed)
    pad = 8 * frames + 64
    tex_h, tex_w = height + 2 * pad, width + 2 * pad
    yy, xx = np.mgrid[0:tex_h, 0:tex_w].astype(np.float32)
    field = 128.0 + 60.0 * np.sin(xx / 37.0) * np.cos(yy / 23.0) + 30.0 * np.sin((xx + yy) / 91.0)
    field += rng.integers(-12, 13, size=field.shape).astype(np.float32)
    out = np.empty((frames, height, width), np.uint8 if depth == 8 else np.uint16)
    scale = 1 << (depth - 8)
    sq = min(128, height // 3, width // 3)
    for f in range(frames):
        ox, oy = pad + 3 * f, pad + 2 * f
        img = field[oy:oy + height, ox:ox + width].copy()
        if scene_cut and f >= frames // 2:
            img = 255.0 - img[:, ::-1]
        sx = (17 + 11 * f) % max(1, width - sq)
        sy = (9 + 5 * f) % max(1, height - sq)
        img[sy:sy + sq, sx:sx + sq] = 200.0
        img += rng.integers(-2, 3, size=img.shape).astype(np.float32)
        img = np.clip(np.rint(img), 0, 255)
        if depth > 8:
            img = img * scale + rng.integers(0, scale, size=img.shape)
            img = np.clip(img, 0, (1 << depth) - 1)
        out[f] = img.astype(out.dtype)
    return out
