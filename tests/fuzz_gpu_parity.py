"""Randomised GPU-vs-oracle sweep of the lookahead path, run by hand (pytest does not collect it):

    python tests/fuzz_gpu_parity.py --seconds 300 [--seed 1]

Every round draws a size, a bit depth, a search configuration, a content class and a request sequence, runs
them through the C-ABI and through the oracle and compares every array.  Prints a progress line per round and
stops at the first difference with the configuration that produced it.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import orc  # noqa: E402
from x265_amd import api, synth  # noqa: E402
from x265_amd.lookahead import CostEstimateGroup  # noqa: E402
import test_gpu_lookahead as T  # noqa: E402


def draw(rng):
    depth = int(rng.choice([8, 8, 10, 10, 12]))
    mode = str(rng.choice(["hex", "hex", "hex", "dia", "umh", "hme", "slices"]))
    if mode == "hme":
        W, H = int(rng.integers(60, 100)) * 8, int(rng.integers(34, 60)) * 8 + int(rng.integers(0, 8))
    elif mode == "slices":
        W, H = int(rng.integers(20, 60)) * 8 + int(rng.integers(0, 16)), int(rng.integers(40, 70)) * 8
    else:
        W, H = int(rng.integers(16, 80)) * 4 + int(rng.integers(0, 4)) * 2, int(rng.integers(16, 60)) * 4 + int(rng.integers(0, 4)) * 2
    kind = str(rng.choice(["synth", "synth", "noise", "flat", "pan"]))
    n = int(rng.integers(3, 8))
    bframes = int(rng.integers(1, 6))
    return dict(depth=depth, mode=mode, W=W, H=H, kind=kind, n=n, bframes=bframes, aq=bool(rng.integers(0, 2)),
                seed=int(rng.integers(1, 1 << 30)))


def requests(rng, n, bframes, count):
    out = []
    while len(out) < count:
        b = int(rng.integers(1, n))
        d0 = int(rng.integers(1, min(b, bframes + 1) + 1))
        d1 = int(rng.integers(0, min(n - 1 - b, bframes + 1) + 1))
        out.append((b - d0, b + d1, b))
    return out


def one_round(cfg):
    rng = np.random.default_rng(cfg["seed"])
    kw = {}
    if cfg["mode"] == "dia":
        kw["search"] = orc.DIA
    elif cfg["mode"] == "umh":
        kw["search"] = orc.UMH
    elif cfg["mode"] == "hme":
        kw.update(hme=True, hme_methods=tuple(int(v) for v in rng.choice([orc.DIA, orc.HEX, orc.UMH], 3)),
                  hme_range=(int(rng.choice([8, 16, 24])), int(rng.choice([16, 32])), 48))
    elif cfg["mode"] == "slices":
        kw["lookahead_slices"] = int(rng.integers(2, 6))
    clip = None if cfg["kind"] == "synth" else T._hard_clip(cfg["W"], cfg["H"], cfg["n"], cfg["depth"], cfg["kind"], cfg["seed"])
    _, ref, ctx, frames = T._setup(cfg["W"], cfg["H"], cfg["n"], cfg["depth"], with_aq=cfg["aq"], seed=cfg["seed"],
                                   bframes=cfg["bframes"], clip=clip, **kw)
    try:
        for i in range(cfg["n"]):
            for p in range(4):
                np.testing.assert_array_equal(ctx.get_plane(i, p), ref.frames[i].buffer(p))
            cost, mode = ctx.get_intra(i)
            np.testing.assert_array_equal(cost, ref.frames[i].intra_cost())
            np.testing.assert_array_equal(mode, ref.frames[i].intra_mode())
        est = CostEstimateGroup(ctx, frames)
        single = cfg["mode"] == "slices"
        for (p0, p1, b) in requests(rng, cfg["n"], cfg["bframes"], 8):
            pen = bool(rng.integers(0, 2))
            got, want = est.single_cost(p0, p1, b, pen), ref.frame_cost(p0, p1, b, pen, single=single)
            assert got == want, ("frame cost", (p0, p1, b), got, want)
            T._check_request(ref, ctx, frames, p0, p1, b)
    finally:
        ctx.close()
        ref.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    t0, rounds = time.time(), 0
    while time.time() - t0 < a.seconds:
        cfg = draw(rng)
        try:
            one_round(cfg)
        except Exception:
            print("MISMATCH in round %d: %r" % (rounds, cfg), flush=True)
            raise
        rounds += 1
        if rounds % 10 == 0:
            print("%d rounds ok, %.0f s" % (rounds, time.time() - t0), flush=True)
    print("fuzz ok: %d rounds in %.0f s" % (rounds, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
