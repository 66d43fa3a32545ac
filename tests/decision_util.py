"""Shared by the slice-type decision tests: an oracle lookahead whose frames went through the reference's whole
pre-analysis (Lowres::init, calcAdaptiveQuantFrame, intra estimate), and the Encoder::encode-style driver loop."""
import numpy as np

from oracle import orc

TYPE_NAMES = {0: "AUTO", 1: "IDR", 2: "I", 3: "P", 4: "BREF", 5: "B"}


def two_scene_clip(width, height, frames, depth, seed):
    """Two unrelated scenes of frames // 2 pictures each (different textures and motion): the second scene's first
    picture costs more to predict from the first scene than to code as intra, whatever the picture size."""
    from x265_amd import synth
    a = synth.luma_clip(width, height, frames, depth, seed=seed, scene_cut=False)[:frames // 2]
    b = synth.luma_clip(width, height, frames, depth, seed=seed + 977, scene_cut=False)[frames - (frames - frames // 2):]
    rng = np.random.default_rng(seed + 5)
    tex = rng.integers(-40, 41, size=(height + 64, width + 64))
    top = (1 << depth) - 1
    scale = 1 << (depth - 8)
    out = [f for f in a]
    for k, f in enumerate(b):
        t = tex[2 * k:2 * k + height, k:k + width] * scale
        out.append(np.clip(top - f[::-1].astype(np.int64) + t, 0, top).astype(a.dtype))
    return np.stack(out)


def chroma_for(clip, depth, seed):
    """Chroma planes for a luma clip: AQ reads them (block variances), nothing else on the path does."""
    rng = np.random.default_rng(seed)
    n, h, w = clip.shape
    top = (1 << depth) - 1
    base_u = rng.integers(top // 3, 2 * top // 3, (h // 2, w // 2))
    out = []
    for i in range(n):
        u = np.clip(np.roll(base_u, 2 * i, axis=1) + rng.integers(-3, 4, base_u.shape), 0, top)
        v = np.clip(np.roll(base_u[::-1], i, axis=0) + rng.integers(-3, 4, base_u.shape), 0, top)
        out.append((u.astype(clip.dtype), v.astype(clip.dtype)))
    return out


def oracle_lookahead(clip, chroma, depth, p, **param_kw):
    """-> (orc.Lookahead with every frame pre-analysed, orc.SliceDecider).  p: api.LookaheadParam.MEDIUM-style dict."""
    h, w = clip.shape[1:]
    aq_on = bool(p["bAdaptiveQuant"])
    qg = p.get("qgSize", 32) if aq_on else 32
    param = orc.make_param(w, h, bframes=p["bframes"], weightp=bool(p["bEnableWeightedPred"]),
                           bframe_bias=p["bFrameBias"], qg_size=qg, **param_kw)
    look = orc.Lookahead(depth, param, with_aq=True)
    for i, y in enumerate(clip):
        if aq_on:
            res = orc.calc_adaptive_quant_frame(
                depth, y, chroma[i][0], chroma[i][1], p["aqMode"], p["aqStrength"],
                weightp=bool(p["bEnableWeightedPred"] or p["bEnableWeightedBiPred"]),
                block_variance=bool(p.get("bEnableFades")), qg_size=qg)
            qa, qc, inv, wsum, wssd = res[:5]
            # with qgSize 8 the estimates read invQscaleFactor8x8; the offset arrays are per 8x8 source block
            lr = look.add_frame(y, res[5] if qg == 8 else inv)
            if p.get("bEnableFades"):
                orc.set_frame_variance(lr, res[-1])
            lr.qp_aq_offset()[:] = qa
            lr.qp_cutree_offset()[:] = qc
            s, d = lr.wp()
            s[:], d[:] = wsum, wssd
        else:
            lr = look.add_frame(y, np.full(look_ncu(param), 256, np.int32))
        if p.get("bHistBasedSceneCut"):
            orc.set_hist_stats(lr, orc.collect_picture_statistics(lr, y, chroma[i][0], chroma[i][1]))
    return look, orc.SliceDecider(look, p)


def look_ncu(param):
    return (((param.sourceWidth // 2) + 7) >> 3) * (((param.sourceHeight // 2) + 7) >> 3)


def drive(add_picture, get_decided, flush, nframes):
    """The order in which Encoder::encode feeds and drains the lookahead (source/encoder/encoder.cpp:1977, 2259):
    one picture in, one decided picture out if there is one; after the last input, flush and drain."""
    out = []
    for i in range(nframes):
        add_picture(i)
        d = get_decided()
        if d is not None:
            out.append(d)
    flush()
    while True:
        d = get_decided()
        if d is None:
            break
        out.append(d)
    return out
