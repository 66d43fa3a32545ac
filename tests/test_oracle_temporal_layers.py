"""CPU: the oracle's mini-GOP orders under --temporal-layers 3..5 against the reference's own table data
(tests/golden/gop_ra.json, read out of source/x265.h by tests/golden/make_gop_tables.py): every complete mini-GOP must
leave the decision in the table's order with the table's layers, and the B-frame count the layers imply is the one
Encoder::configure sets (x265_temporal_layer_bframes).  The GPU driver is compared with the oracle in
tests/test_gpu_slicetype.py; this pins the oracle's side of that comparison to data the reference holds."""
import json
import os

import numpy as np
import pytest

from x265_amd import api, synth

import decision_util as du

GOLDEN = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gop_ra.json")))


@pytest.mark.parametrize("layers", [3, 4, 5])
def test_complete_mini_gops_follow_the_reference_table(layers):
    gop = GOLDEN["gop_ra"][layers - 3]
    length = GOLDEN["gop_ra_length"][layers - 3]
    bframes = GOLDEN["temporal_layer_bframes"][layers - 1]
    assert bframes == length - 1 and len(gop["poc_offset"]) == length
    p = dict(api.LookaheadParam.MEDIUM, bEnableTemporalSubLayers=layers, bframes=bframes, bFrameAdaptive=0,
             scenecutThreshold=0, lookaheadDepth=max(20, bframes + 2), keyframeMax=250)
    N, W, H, depth = 2 * length + 3, 128, 96, 8
    clip = synth.luma_clip(W, H, N, depth, seed=5, scene_cut=False)
    look, dec = du.oracle_lookahead(clip, du.chroma_for(clip, depth, 6), depth, p)
    try:
        order = du.drive(lambda i: dec.add_picture(i, 0, 0), dec.get_decided, dec.flush, N)
        assert order[0] == 0
        # frames 1..length and length+1..2*length are complete mini-GOPs
        for g in range(2):
            base = g * length
            coded = order[1 + g * length:1 + (g + 1) * length]
            assert [f - base for f in coded] == gop["poc_offset"], (layers, g, coded)
            for k, f in enumerate(coded):
                fr = look.frames[f]
                assert int(fr.i64("tempLayer")) == gop["layer"][k], (layers, f)
                assert int(fr.i64("gopOffset")) == k and int(fr.i64("gopId")) == layers - 3
            # the closing picture is a P, the middles down to the last-but-one layer are referenced Bs
            types = [int(look.frames[f].i64("sliceType")) for f in coded]
            assert types[0] == 3                      # X265_TYPE_P
            top = max(gop["layer"])
            for k in range(1, length):
                assert types[k] == (5 if gop["layer"][k] == top else 4), (layers, k, types)   # B / BREF
    finally:
        dec.close()
        look.close()
