"""Host-side mirror of the lookahead's cost-estimation interface.

Mirrors, name for name, the pieces of x265's host code that sit directly above
the offload boundary, so that parity tests read like calls into the reference:

  PreLookaheadGroup.process_tasks   <- PreLookaheadGroup::processTasks (slicetype.cpp:1836-1861)
  CostEstimateGroup.single_cost     <- CostEstimateGroup::singleCost   (slicetype.cpp:4148-4152)
  CostEstimateGroup.add / finish_batch <- ::add / ::finishBatch        (slicetype.cpp:4154-4196)

  LookaheadTLD.weights_analyse      <- LookaheadTLD::weightsAnalyse    (slicetype.cpp:870-971)

The cache check, the bDoSearch decision from the 0x7FFF sentinel, the B-frame
bias and the intra penalty of estimateFrameCost (slicetype.cpp:4284-4383) are
host logic and live here; pixels never do.  Each frame of the window owns one
device slot.
"""
import numpy as np

_f32 = np.float32


class LowresState:
    """Host bookkeeping of one frame: what x265 keeps in Lowres besides pixels."""

    def __init__(self, slot, poc, D):
        self.slot = slot
        self.poc = poc
        self.searched = [[False] * D for _ in range(2)]        # lowresMvs[l][d][0].x != 0x7FFF
        self.cost_est = {}                                       # (d0, d1) -> costEst (after bias)
        self.cost_est_aq = {}
        self.intra_mbs = [0] * D
        self.pending = set()                                     # (d0, d1) submitted, not fetched
        # luma sum and sum of squared deviations of the source frame, as calcAdaptiveQuantFrame
        # leaves them in Lowres::wp_sum[0] / wp_ssd[0] (slicetype.cpp:600-611); host inputs
        self.wp_sum = 0
        self.wp_ssd = 0
        self.weighted = {}                                       # d0 -> weightedRef[d0].isWeighted
        self.weighted_cost_delta = {}


class LookaheadTLD:
    """weightsAnalyse: the float guess logic stays on the host, in single precision as in the
    reference; the two pixel passes (weightCostLuma, weight_pp) run on the device."""

    def __init__(self, ctx):
        self.ctx = ctx

    def weights_analyse(self, fenc, ref):
        ctx = self.ctx
        g = ctx.geom
        delta = fenc.poc - ref.poc
        fenc.weighted[delta] = False
        epsilon = _f32(1.0) / _f32(128.0)
        shift = _f32(1 << (ctx.depth - 8))
        if fenc.wp_ssd and ref.wp_ssd:
            guess_scale = np.sqrt(_f32(fenc.wp_ssd) / _f32(ref.wp_ssd))
        else:
            guess_scale = _f32(1.0)
        area = _f32(g.lines * g.width)
        fenc_mean = _f32(fenc.wp_sum) / area / shift
        ref_mean = _f32(ref.wp_sum) / area / shift
        if abs(ref_mean - fenc_mean) < _f32(0.5) and abs(_f32(1.0) - guess_scale) < epsilon:
            return
        # WeightParam::setFromWeightAndOffset with normalisation (common/slice.h)
        w, denom = int(guess_scale * _f32(128) + _f32(0.5)), 7
        while denom > 0 and w > 127:
            denom -= 1
            w >>= 1
        w = min(w, 127)
        mindenom, minscale, minoff = denom, w, 0
        origscore = minscore = ctx.weight_cost(fenc.slot, ref.slot)
        if not minscore:
            return
        cur_scale = minscale
        cur_offset = int(fenc_mean - ref_mean * _f32(cur_scale) / _f32(1 << mindenom) + _f32(0.5))
        if cur_offset < -128 or cur_offset > 127:
            cur_offset = max(-128, min(127, cur_offset))
            cur_scale = int(_f32(1 << mindenom) * (fenc_mean - _f32(cur_offset)) / ref_mean + _f32(0.5))
            cur_scale = max(0, min(127, cur_scale))
        s = ctx.weight_cost(fenc.slot, ref.slot, 1, cur_scale, mindenom, cur_offset)
        found = False
        if s < minscore:
            minscore, minscale, minoff, found = s, cur_scale, cur_offset, True
        if mindenom > 0 and minscale and not (minscale & 1):
            idx = (minscale & -minscale).bit_length() - 1
            sh = min(idx, mindenom)
            mindenom -= sh
            minscale >>= sh
        if not found or (minscale == (1 << mindenom) and minoff == 0) or \
                _f32(minscore) / _f32(origscore) > _f32(0.998):
            return
        fenc.weighted_cost_delta[delta] = minscore // origscore       # integer division, slicetype.cpp:955
        ctx.weight_planes(fenc.slot, delta, ref.slot, minscale, mindenom, minoff)
        fenc.weighted[delta] = True


class PreLookaheadGroup:
    def __init__(self, ctx):
        self.ctx = ctx

    def process_tasks(self, frames, lumas, inv_qscales=None):
        """Lowres::init + lowresIntraEstimate for each (LowresState, luma) pair."""
        for k, (fr, luma) in enumerate(zip(frames, lumas)):
            self.ctx.lowres_init(fr.slot, luma, fr.poc)
            self.ctx.set_aq(fr.slot, None if inv_qscales is None else inv_qscales[k])
            D = self.ctx.D
            fr.searched = [[False] * D for _ in range(2)]
            fr.cost_est.clear()
            fr.cost_est_aq.clear()
            fr.intra_mbs = [0] * D
        self.ctx.intra_estimate([fr.slot for fr in frames])
        for fr in frames:
            ce, ceaq, _ = self.ctx.get_frame_cost(fr.slot, 0, 0)
            fr.cost_est[(0, 0)] = ce
            fr.cost_est_aq[(0, 0)] = ceaq


class CostEstimateGroup:
    def __init__(self, ctx, frames, weightp=False):
        self.ctx = ctx
        self.tld = LookaheadTLD(ctx) if weightp else None
        self.frames = frames            # list of LowresState indexed like x265's frames[]
        g = ctx.geom
        W, H = g.widthInCU, g.heightInCU
        # LookaheadTLD::ncu = m_8x8Blocks (slicetype.cpp:1065-1066)
        self.ncu = (W * H + 4 - 2 * (W + H)) if (W > 2 and H > 2) else W * H
        self.batch = []

    def _request(self, p0, p1, b):
        """Builds the device request for (p0, p1, b) or None when the cost is cached."""
        fenc = self.frames[b]
        d0, d1 = b - p0, p1 - b
        if (d0, d1) in fenc.cost_est or (d0, d1) in fenc.pending:
            return None
        do0 = not fenc.searched[0][d0]
        do1 = p1 > b and not fenc.searched[1][d1]
        if do0:
            fenc.searched[0][d0] = True
        if do1:
            fenc.searched[1][d1] = True
        fenc.pending.add((d0, d1))
        weighted = False
        if self.tld is not None:               # slicetype.cpp:4300-4304: reset on every request
            fenc.weighted[d0] = False
            if do0:
                self.tld.weights_analyse(fenc, self.frames[p0])
                weighted = fenc.weighted[d0]
        return (self.frames[p0].slot, fenc.slot, self.frames[p1].slot, p0, b, p1, do0, do1, int(weighted))

    def _collect(self, b, d0, d1, res=None):
        fenc = self.frames[b]
        ce, ceaq, imb = res if res is not None else self.ctx.get_frame_cost(fenc.slot, d0, d1)
        fenc.cost_est[(d0, d1)] = ce
        fenc.cost_est_aq[(d0, d1)] = ceaq
        if d1 == 0:
            fenc.intra_mbs[d0] += imb
        fenc.pending.discard((d0, d1))

    def single_cost(self, p0, p1, b, intra_penalty=False):
        r = self._request(p0, p1, b)
        fenc = self.frames[b]
        d0, d1 = b - p0, p1 - b
        if r is not None:
            self._collect(b, d0, d1, self.ctx.estimate_costs([r])[0])
        score = fenc.cost_est[(d0, d1)]
        if intra_penalty:
            # C integer division truncates toward zero; operands are non-negative here
            score += score * fenc.intra_mbs[d0] // (self.ncu * 8)
        return score

    def add(self, p0, p1, b):
        r = self._request(p0, p1, b)
        if r is not None:
            self.batch.append((r, b, b - p0, p1 - b))

    def finish_batch(self):
        if not self.batch:
            return
        res = self.ctx.estimate_costs([r for r, _, _, _ in self.batch])
        for (_, b, d0, d1), rc in zip(self.batch, res):
            self._collect(b, d0, d1, rc)
        self.batch = []
