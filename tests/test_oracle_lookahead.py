"""CPU: the oracle's lookahead (Lowres::init, intra estimate, motion search, frame
costs) against the committed golden vectors and domain properties."""
import importlib.util
import os
import zlib

import numpy as np
import pytest

from oracle import orc
from x265_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def _golden_module():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("name", ["lookahead_8bit", "lookahead_10bit", "lookahead_umh_8bit", "lookahead_hme_10bit",
                                  "pu_me_star_8bit", "pu_me_hex_10bit"])
def test_oracle_matches_golden(oracle_built, name):
    mg = _golden_module()
    want = np.load(os.path.join(HERE, "golden", name + ".npz"))
    got = mg.generate_pu(name) if name.startswith("pu_me") else mg.generate(name)
    assert sorted(got.keys()) == sorted(want.files)
    for k in want.files:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)


def test_lowres_planes_are_edge_extended(oracle_built):
    W, H = 200, 120   # lowres 100x60 -> rounded to 104x64: rows/cols beyond the picture are replicas
    clip = synth.luma_clip(W, H, 1, 8, seed=3)
    param = orc.make_param(W, H)
    la = orc.Lookahead(8, param)
    lr = la.add_frame(clip[0])
    g = lr.geometry
    assert (g["width"], g["lines"], g["widthInCU"], g["heightInCU"]) == (104, 64, 13, 8)
    assert g["stride"] % 32 == 0 and g["marginX"] == 96 and g["marginY"] == 80
    rows = g["lines"] + 2 * g["marginY"]
    for p in range(4):
        buf = lr.buffer(p).reshape(rows, g["stride"])
        inner = buf[g["marginY"]:g["marginY"] + g["lines"], g["marginX"]:g["marginX"] + g["width"]]
        full = np.pad(inner, ((g["marginY"],) * 2, (g["marginX"],) * 2), mode="edge")
        np.testing.assert_array_equal(buf[:, :g["width"] + 2 * g["marginX"]], full)
    la.close()


def test_static_clip_gives_zero_vectors_and_flat_costs(oracle_built):
    W, H = 192, 128
    frame = synth.luma_clip(W, H, 1, 8, seed=5, scene_cut=False)[0]
    param = orc.make_param(W, H)
    la = orc.Lookahead(8, param)
    for _ in range(3):
        la.add_frame(frame)
    la.frame_cost(0, 2, 1)
    mv0, c0 = la.frames[1].mvs(0, 1), la.frames[1].mv_costs(0, 1)
    assert not mv0.any()
    # identical reference, zero residual.  The first visited block (bottom-right) has no
    # predictor candidates and returns the MV cost of (0,0); every later block has a zero-cost
    # zero-vector candidate, so the bidir skip shortcut stores cost 0 (slicetype.cpp:4485-4489)
    t, c = la.table, la.centre
    assert c0[-1] == 2 * int(t[c]) and not c0[:-1].any()
    # the bidirectional average has SATD 0 everywhere: every block costs lowresPenalty, lists = 3 or 1
    assert (la.frames[1].lowres_costs(1, 1) & 0x3FFF == 4).all()
    # as a P request the search has no skip shortcut: MV cost of (0,0) everywhere
    la.frame_cost(1, 2, 2)
    assert (la.frames[2].mv_costs(0, 1) == 2 * int(t[c])).all()
    la.close()


def test_global_translation_is_recovered(oracle_built):
    """A pure pan by (+6,+4) half-res quarter-pels shows up as the dominant vector."""
    W, H = 320, 192
    clip = synth.luma_clip(W, H, 4, 8, seed=11, scene_cut=False)
    param = orc.make_param(W, H)
    la = orc.Lookahead(8, param)
    for f in clip:
        la.add_frame(f)
    la.frame_cost(0, 1, 1)
    mv = la.frames[1].mvs(0, 1)
    vals, counts = np.unique(mv, axis=0, return_counts=True)
    assert tuple(vals[counts.argmax()]) == (6, 4)
    assert counts.max() > 0.6 * len(mv)
    la.close()


def test_cost_cache_and_bias(oracle_built):
    W, H = 192, 128
    clip = synth.luma_clip(W, H, 4, 8, seed=2)
    la = orc.Lookahead(8, orc.make_param(W, H, bframe_bias=0))
    lb = orc.Lookahead(8, orc.make_param(W, H, bframe_bias=20))
    for f in clip:
        la.add_frame(f)
        lb.add_frame(f)
    a = la.frame_cost(0, 2, 1)
    assert la.frame_cost(0, 2, 1) == a                       # cached
    b = lb.frame_cost(0, 2, 1)
    # both are the same raw sum scaled by 100/(130 + bias), truncating
    lo = a * 130 // 100
    raw = [r for r in range(lo, lo + 3) if r * 100 // 130 == a]
    assert any(r * 100 // 150 == b for r in raw)
    assert la.frame_cost(0, 1, 1) == la.frames[1].i64("costEst", 1, 0)   # P requests are not scaled
    la.close()
    lb.close()
