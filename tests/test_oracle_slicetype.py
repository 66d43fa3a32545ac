"""CPU: the oracle's slice-type decision (oracle/orc_slicetype.c) behaves like the reference's on cases whose
outcome the reference's logic fixes by construction (source/encoder/slicetype.cpp:1982-3321, 3484-3579):
first frame IDR, the hard cut of the synthetic clip becomes an I frame flagged bScenecut, coded order puts the
non-B frame of a mini-GOP first, and no request is ever computed twice (the cost cache of :4290)."""
import numpy as np
import pytest

from oracle import orc
from x265_amd import api, synth

import decision_util as du

IDR, I, P, BREF, B = 1, 2, 3, 4, 5


def _run(W, H, N, depth=8, seed=3, **kw):
    p = dict(api.LookaheadParam.MEDIUM, **kw)
    clip = du.two_scene_clip(W, H, N, depth, seed)
    chroma = du.chroma_for(clip, depth, seed + 1)
    look, dec = du.oracle_lookahead(clip, chroma, depth, p)
    try:
        order = du.drive(lambda i: dec.add_picture(i), dec.get_decided, dec.flush, N)
        types = [int(look.frames[i].i64("sliceType")) for i in range(N)]
        cuts = [int(look.frames[i].i64("bScenecut")) for i in range(N)]
        keys = [int(look.frames[i].i64("bKeyframe")) for i in range(N)]
        return order, types, cuts, keys, dec.trace()
    finally:
        dec.close()
        look.close()


@pytest.mark.parametrize("kw", [{}, {"poolWorkers": 16}, {"bFrameAdaptive": 1}, {"bFrameAdaptive": 0},
                                {"bBPyramid": 0, "cuTree": 0}, {"bOpenGOP": 0, "keyframeMax": 12, "keyframeMin": 1},
                                {"vbvBufferSize": 1000, "vbvMaxBitrate": 1000}, {"lookaheadDepth": 10, "bframes": 4},
                                {"bEnableTemporalSubLayers": 2}, {"gopLookahead": 4, "keyframeMax": 14, "keyframeMin": 2}])
def test_decisions_are_well_formed(kw):
    N = 30
    order, types, cuts, keys, trace = _run(192, 128, N, **kw)
    assert sorted(order) == list(range(N))                      # every frame comes out exactly once
    assert types[0] == IDR and keys[0] == 1
    assert all(t in (IDR, I, P, BREF, B) for t in types)
    # the second half is an unrelated scene: an I frame with the scenecut flag
    assert types[N // 2] in (I, IDR) and cuts[N // 2] == 1, (types, cuts)
    assert sum(cuts) == 1
    # B frames never run longer than --bframes, and the clip ends on a non-B frame
    run = 0
    for t in types:
        run = run + 1 if t in (B, BREF) else 0
        assert run <= 4
    assert types[-1] not in (B, BREF)
    # coded order: a mini-GOP's closing non-B frame first, then its B-refs, then its B frames
    pos = {f: k for k, f in enumerate(order)}
    for f in range(N):
        if types[f] in (B, BREF):
            nxt = next(g for g in range(f + 1, N) if types[g] not in (B, BREF))
            assert pos[nxt] < pos[f]
            if types[f] == B:
                assert all(pos[g] < pos[f] for g in range(f + 1, nxt) if types[g] == BREF)
    # the cost cache: a (p0, p1, b) is computed once, and each (b, list, distance) is searched once
    reqs = [tuple(r[:3]) for r in trace]
    assert len(reqs) == len(set(reqs)) and len(reqs) > N
    searched = [(r[2], 0, r[2] - r[0]) for r in trace if r[3]] + [(r[2], 1, r[1] - r[2]) for r in trace if r[4]]
    assert len(searched) == len(set(searched))
    if kw.get("poolWorkers"):
        assert trace[:, 5].any()                                # the two batches were issued as batches
    if kw.get("keyframeMax") == 12 and not kw.get("gopLookahead"):
        assert all(types[f] == IDR for f in range(N) if keys[f])    # closed GOP: keyframes are IDR
        assert max(b - a for a, b in zip([f for f in range(N) if keys[f]], [f for f in range(N) if keys[f]][1:])) <= 12


def test_batch_order_changes_requests_not_wellformedness():
    """Lazy (no pool) and batched (pool) request orders issue different request lists, which is why a parity
    test has to follow the order of the configuration it pins."""
    _, types_lazy, _, _, trace_lazy = _run(192, 128, 24)
    _, types_batch, _, _, trace_batch = _run(192, 128, 24, poolWorkers=16)
    assert len(trace_batch) > len(trace_lazy)
    assert types_lazy[0] == types_batch[0] == IDR
    assert types_lazy[12] in (I, IDR) and types_batch[12] in (I, IDR)


def test_forced_slice_types_are_kept():
    p = dict(api.LookaheadParam.MEDIUM)
    N = 12
    clip = synth.luma_clip(192, 128, N, 8, seed=5, scene_cut=False)
    look, dec = du.oracle_lookahead(clip, du.chroma_for(clip, 8, 6), 8, p)
    try:
        forced = {4: I, 7: P}
        du.drive(lambda i: dec.add_picture(i, 0, forced.get(i, 0)), dec.get_decided, dec.flush, N)
        types = [int(look.frames[i].i64("sliceType")) for i in range(N)]
        assert types[4] == I and types[7] == P, types
    finally:
        dec.close()
        look.close()
