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
import test_gpu_mcstf as TM  # noqa: E402
import test_gpu_pu_me as TP  # noqa: E402


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
        if not single and rng.integers(0, 2):
            # one device batch (CostEstimateGroup::add / finishBatch); the oracle runs the same requests in order
            reqs = list(dict.fromkeys(requests(rng, cfg["n"], cfg["bframes"], int(rng.integers(2, 40)))))
            for r in reqs:
                est.add(*r)
            est.finish_batch()
            for (p0, p1, b) in reqs:
                want = ref.frame_cost(p0, p1, b)
                assert frames[b].cost_est[(b - p0, p1 - b)] == want, ("batch frame cost", (p0, p1, b))
            for (p0, p1, b) in reqs:
                T._check_request(ref, ctx, frames, p0, p1, b)
        for (p0, p1, b) in requests(rng, cfg["n"], cfg["bframes"], 8):
            pen = bool(rng.integers(0, 2))
            got, want = est.single_cost(p0, p1, b, pen), ref.frame_cost(p0, p1, b, pen, single=single)
            assert got == want, ("frame cost", (p0, p1, b), got, want)
            T._check_request(ref, ctx, frames, p0, p1, b)
    finally:
        ctx.close()
        ref.close()


def draw_pu(rng):
    return dict(depth=int(rng.choice([8, 10, 12])), method=int(rng.choice([orc.DIA, orc.HEX, orc.UMH, orc.STAR, orc.FULL, orc.SEA])),
                merange=int(rng.choice([4, 8, 16, 24, 57])), subpel=int(rng.integers(0, 8)),
                size=(int(rng.integers(20, 70)) * 8 + int(rng.integers(0, 4)) * 2, int(rng.integers(12, 40)) * 8 + int(rng.integers(0, 4)) * 2),
                lam=float(rng.choice([0, 0, 2.5, 9.5, 40.0])), seed=int(rng.integers(1, 1 << 30)),
                content=str(rng.choice(["synth", "synth", "flat", "half"])), narrow=bool(rng.integers(0, 2)))


def pu_round(cfg):
    if cfg["method"] == orc.SEA:
        TP.test_pu_successive_elimination(cfg["depth"], cfg["merange"], cfg["subpel"], cfg["seed"] % 100000)
        return
    merange = min(cfg["merange"], 6) if cfg["method"] == orc.FULL else cfg["merange"]
    TP._run(cfg["depth"], cfg["method"], merange, cfg["subpel"], 24, cfg["seed"], size=cfg["size"], lam=cfg["lam"] or None,
            content=cfg["content"], narrow=cfg["narrow"])


def draw_mcstf(rng):
    return dict(depth=int(rng.choice([8, 10, 12])), W=int(rng.integers(8, 40)) * 16 + int(rng.integers(0, 8)) * 2,
                H=int(rng.integers(6, 24)) * 16 + int(rng.integers(0, 8)) * 2, refs=int(rng.integers(1, 5)),
                seed=int(rng.integers(1, 1 << 30)))


def mcstf_round(cfg):
    depth, W, H = cfg["depth"], cfg["W"], cfg["H"]
    clip = TM._clip(W, H, 5, depth, seed=cfg["seed"] % 100000)
    param = orc.make_param(W, H, mcstf=True)
    L = orc.lib(depth)
    pics = []
    for i, f in enumerate(clip):
        lr = orc.Lowres(L, depth, param)
        lr.init(f, i)
        pics.append(orc.McstfPicture(lr, f))
    ctx = api.Context(W, H, depth=depth, num_slots=5, mcstf=True, keep_full_res=True)
    try:
        for i, f in enumerate(clip):
            ctx.lowres_init(i, f, i)
        refs = [0, 1, 3, 4][:cfg["refs"]]
        want = orc.mcstf_motion_search(pics[2], [pics[r] for r in refs])
        got = ctx.mcstf_motion_search(2, refs)
        for name, g, w in zip(("mvs", "error", "coarse"), got, want):
            np.testing.assert_array_equal(g, w, err_msg=name)
    finally:
        ctx.close()
        for p in pics:
            p.lowres.close()


def draw_aq(rng):
    return dict(depth=int(rng.choice([8, 10, 12])), W=int(rng.integers(16, 140)) * 4 + int(rng.integers(0, 2)) * 2,
                H=int(rng.integers(16, 90)) * 4 + int(rng.integers(0, 2)) * 2, mode=int(rng.integers(0, 5)),
                strength=float(rng.choice([0.0, 0.4, 1.0, 1.7])), qg=int(rng.choice([8, 16, 32, 64])),
                weightp=bool(rng.integers(0, 2)), hdr=bool(rng.integers(0, 2)), qo=bool(rng.integers(0, 2)),
                var=bool(rng.integers(0, 2)), hevc=bool(rng.integers(0, 3) == 0), skip=bool(rng.integers(0, 8) == 0),
                kind=str(rng.choice(["noise", "mixed", "flat", "edges"])), seed=int(rng.integers(1, 1 << 30)))


def aq_round(cfg):
    """calcAdaptiveQuantFrame in every covered mode: all arrays, the wp sums, the variance tail, the HEVC AQ layers."""
    depth, W, H = cfg["depth"], cfg["W"], cfg["H"]
    rng = np.random.default_rng(cfg["seed"])
    dt = np.uint8 if depth == 8 else np.uint16
    top = (1 << depth) - 1
    if cfg["kind"] == "edges":
        import test_gpu_aq as TA
        y = TA._edgy_picture(W, H, depth, cfg["seed"] % 1000)[0]
    elif cfg["kind"] == "flat":
        y = np.full((H, W), int(rng.integers(0, top + 1)), dt)
        y[H // 2:, W // 3:] = int(rng.integers(0, top + 1))
    else:
        y = rng.integers(0, top + 1, (H, W)).astype(dt)
        if cfg["kind"] == "mixed":
            y[: H // 3] = (np.arange(W) * top // W).astype(dt)
            y[H // 3: H // 2] = top // 5
    u = rng.integers(0, top + 1, (H // 2, W // 2)).astype(dt)
    v = rng.integers(top // 3, top // 2 + 1, (H // 2, W // 2)).astype(dt)
    w8, h8 = ((W // 2) + 7) >> 3, ((H // 2) + 7) >> 3
    qg8 = cfg["qg"] == 8
    nblocks = 4 * w8 * h8 if qg8 else w8 * h8
    qo = rng.normal(0, 1.5, nblocks).astype(np.float32) if cfg["qo"] else None
    hevc = cfg["hevc"] and cfg["mode"] not in (0, 4) and cfg["strength"] != 0 and not cfg["skip"]
    kw = dict(weightp=cfg["weightp"], hdr10=cfg["hdr"], skip_offsets=cfg["skip"], quant_offsets=qo, qg_size=cfg["qg"],
              block_variance=cfg["var"])
    with api.Context(W, H, depth=depth, num_slots=1) as ctx:
        if hevc:
            depths = {64: (0, 2), 32: (0, 1, 2), 16: (0, 1, 2), 8: (0, 1, 2, 3)}[cfg["qg"]]
            parts = [((64 >> d), 2 * w8, 2 * h8) if d == 3 else ((64 >> d), (W + (64 >> d) - 1) // (64 >> d),
                                                                  (H + (64 >> d) - 1) // (64 >> d)) for d in depths]
            got = ctx.aq_frame(y, u, v, cfg["mode"], cfg["strength"], hevc_aq_parts=parts, qp_adaptation_range=1.25, **kw)
            L = orc.lib(depth)
            for (s_, nw, nh), g in zip(parts, got[-1]):
                arrs = [np.full(nw * nh, np.nan) for _ in range(3)]
                avg = L.orc_hevc_aq_layer(orc.ptr(y), W, W, H, s_, s_, nw, nh, 1.25, *(orc.ptr(a) for a in arrs))
                for a, b in zip(g, tuple(arrs) + (avg,)):
                    np.testing.assert_array_equal(a, b)
            B = 8 if qg8 else 16
            visited = ((W + B - 1) // B) * ((H + B - 1) // B)
            want_inv = np.array([L.orc_exp2fix8(float(q)) for q in got[-1][-1][1][:visited]], np.int32)
            np.testing.assert_array_equal(got[2][:visited], want_inv)
            want = orc.calc_adaptive_quant_frame(depth, y, u, v, cfg["mode"], cfg["strength"], **kw)
            np.testing.assert_array_equal(got[3], want[3])
            np.testing.assert_array_equal(got[4], want[4])
            if cfg["var"]:
                np.testing.assert_array_equal(got[-3], want[-2])
                assert got[-2] == want[-1]
        else:
            got = ctx.aq_frame(y, u, v, cfg["mode"], cfg["strength"], **kw)
            want = orc.calc_adaptive_quant_frame(depth, y, u, v, cfg["mode"], cfg["strength"], **kw)
            assert len(got) == len(want)
            for k, (g, w_) in enumerate(zip(got, want)):
                np.testing.assert_array_equal(g, w_, err_msg="output %d" % k)


def draw_decide(rng):
    """A slice-type decision run: parameter combinations of the decision driver over a short two-scene clip."""
    p = {}
    if rng.integers(0, 2):
        p["poolWorkers"] = int(rng.choice([2, 8, 16]))
    p["bFrameAdaptive"] = int(rng.choice([0, 1, 2, 2]))
    p["bframes"] = int(rng.choice([0, 1, 3, 4, 4, 6]))
    p["bBPyramid"] = int(rng.integers(0, 2)) if p["bframes"] > 1 else 0
    p["lookaheadDepth"] = int(rng.choice([5, 10, 20, 20]))
    p["bOpenGOP"] = int(rng.integers(0, 2))
    if rng.integers(0, 3) == 0:
        p["keyframeMax"], p["keyframeMin"] = int(rng.integers(8, 30)), int(rng.integers(1, 8))
    if rng.integers(0, 3) == 0:
        p["vbvBufferSize"] = p["vbvMaxBitrate"] = 1000
    p["cuTree"] = int(rng.integers(0, 4) != 0)
    p["bEnableWeightedPred"] = int(rng.integers(0, 2))
    p["bEnableWeightedBiPred"] = int(rng.integers(0, 4) == 0)
    p["scenecutThreshold"] = int(rng.choice([0, 40, 40]))
    kind = int(rng.integers(0, 6))
    if kind == 0:
        p["bAQMotion"] = 1
    elif kind == 1:
        p["qgSize"] = 8
    elif kind == 2:
        p.update(bHevcAq=1, qgSize=int(rng.choice([8, 16, 32, 64])), qpAdaptationRange=float(rng.choice([1.0, 1.5, 3.0])))
    elif kind == 3:
        p.update(bAdaptiveQuant=0, cuTree=0, bEnableWeightedPred=0, bEnableWeightedBiPred=0)
    if p.get("bAdaptiveQuant", 1):
        p["aqMode"] = int(rng.choice([1, 2, 2, 3]))
        p["aqStrength"] = float(rng.choice([0.6, 1.0, 1.4]))
        if rng.integers(0, 4) == 0:
            p["bEnableFades"], p["fpsNum"] = 1, int(rng.choice([5, 10]))
    if rng.integers(0, 5) == 0:
        p["bHistBasedSceneCut"] = 1
    if "poolWorkers" not in p and rng.integers(0, 4) == 0:
        p["lookaheadSlices"] = int(rng.integers(2, 5))         # single requests run in cooperative slices
    p["lookaheadDepth"] = max(p["lookaheadDepth"], p["bframes"] + 1)      # x265_check_params (param.cpp:1811)
    if p.get("bEnableFades"):
        # the fade detection reads frameVariance of the first bframes + 2 queued frames, of which x265 has pre-analysed
        # only the first lookaheadDepth: beyond them it reads what Lowres::create left
        p["lookaheadDepth"] = max(p["lookaheadDepth"], p["bframes"] + 2)
    if "poolWorkers" in p:                     # the harness expects searches run ahead of single requests to be found
        p["bframes"], p["lookaheadDepth"] = max(p["bframes"], 3), max(p["lookaheadDepth"], 10)
    layers = int(rng.choice([0, 0, 0, 0, 2, 3, 4, 5]))
    if layers:
        p["bEnableTemporalSubLayers"] = layers
    if layers > 2:                             # Encoder::configure's consequences (encoder.cpp:4322-4332)
        p["bframes"], p["bFrameAdaptive"], p["bBPyramid"] = (4 << (layers - 3)) - 1, 0, int(rng.integers(0, 4) != 0)
        p["lookaheadDepth"] = max(p["lookaheadDepth"], p["bframes"] + (2 if p.get("bEnableFades") else 1))
    return dict(p=p, n=int(rng.integers(12, 30)), size=(int(rng.integers(96, 192)) * 2, int(rng.integers(64, 112)) * 2),
                depth=int(rng.choice([8, 8, 10, 12])), seed=int(rng.integers(1, 1 << 30)))


def decide_round(cfg):
    import decision_util as du
    import test_gpu_slicetype as TS
    p = dict(api.LookaheadParam.MEDIUM, **cfg["p"])
    W, H = cfg["size"]
    clip = du.two_scene_clip(W, H, cfg["n"], cfg["depth"], seed=cfg["seed"] % 100000)
    TS._compare(clip, du.chroma_for(clip, cfg["depth"], cfg["seed"] % 1000), cfg["depth"], p)


def draw_tme(rng):
    """--threaded-me: CTU rows of a small picture through the level-ordered generator against the serial restatement."""
    ctu = int(rng.choice([64, 64, 32]))
    kw = dict(method=int(rng.choice([api.STAR_SEARCH, api.HEX_SEARCH, api.DIA_SEARCH, api.UMH_SEARCH])),
              merange=int(rng.choice([8, 16, 57])), subpel=int(rng.integers(0, 5)), amp=bool(rng.integers(0, 2)),
              rect=bool(rng.integers(0, 4) != 0), min_cu=int(rng.choice([8, 8, 16])), ctu=ctu,
              num_refs=int(rng.integers(1, 3)), num_l1=int(rng.integers(0, 2)), lam=float(rng.choice([1.0, 4.0, 9.5])),
              lookahead=bool(rng.integers(0, 2)), median=bool(rng.integers(0, 3) == 0), temporal=bool(rng.integers(0, 2)),
              seed=int(rng.integers(1, 100000)))
    if rng.integers(0, 2):
        lo = int(rng.integers(18, 34))
        kw["cu_qp"] = (lo, lo + int(rng.integers(0, 5)))
        kw["area_qp"] = bool(rng.integers(0, 2))
    W, H = int(rng.integers(5, 11)) * 64 + int(rng.integers(0, 4)) * 16, int(rng.integers(3, 6)) * 64 + int(rng.integers(0, 4)) * 16
    rows = -(-H // ctu)
    r0 = int(rng.integers(0, rows))
    return dict(W=W, H=H, depth=int(rng.choice([8, 10])), rows=(r0, int(rng.integers(r0 + 1, rows + 1))), kw=kw)


def tme_round(cfg):
    import test_gpu_threaded_me as TT
    TT._run(cfg["W"], cfg["H"], cfg["depth"], cfg["rows"], **cfg["kw"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--skip", type=int, default=0, help="draw this many configurations without running them first")
    ap.add_argument("--log", default=None, help="file that receives every round's configuration before it runs")
    ap.add_argument("--what", default="lookahead", choices=["lookahead", "pu", "mcstf", "aq", "decide", "tme"])
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    log = open(a.log, "w") if a.log else None
    drawer, runner = {"lookahead": (draw, one_round), "pu": (draw_pu, pu_round), "mcstf": (draw_mcstf, mcstf_round),
                      "aq": (draw_aq, aq_round), "decide": (draw_decide, decide_round), "tme": (draw_tme, tme_round)}[a.what]
    t0, rounds, refused = time.time(), 0, 0
    for _ in range(a.skip):
        drawer(rng)
    while time.time() - t0 < a.seconds:
        cfg = drawer(rng)
        # the single-launch search: one wait for the neighbour vectors, or as they come, in turn (by round, so that
        # the draws stay what they are for a seed)
        staged = ("0", "100000")[(a.skip + rounds + refused) & 1]
        os.environ["X265GPU_FLOW_STAGED_JOBS"] = staged
        if log:
            log.write("staged_jobs=%s " % staged + repr(cfg) + "\n")
            log.flush()
        try:
            runner(cfg)
        except api.X265GpuError as e:
            if "x265gpu_open failed with status -5" not in str(e) and "unsupported parameters" not in str(e):
                print("FAILED in round %d: %r" % (rounds, cfg), flush=True)
                raise
            refused += 1            # a size the library declines (lowres grid of the padded picture one block larger)
            continue
        except Exception:
            print("MISMATCH in round %d: %r" % (rounds, cfg), flush=True)
            raise
        rounds += 1
        if rounds % 10 == 0:
            print("%d rounds ok, %.0f s" % (rounds, time.time() - t0), flush=True)
    print("fuzz ok: %d rounds in %.0f s (%d draws with a refused size)" % (rounds, time.time() - t0, refused), flush=True)


if __name__ == "__main__":
    main()
