"""Host-side mirror of the lookahead's cost-estimation interface.

Mirrors, name for name, the pieces of x265's host code that sit directly above
the offload boundary, so that parity tests read like calls into the reference:

  PreLookaheadGroup.process_tasks   <- PreLookaheadGroup::processTasks (slicetype.cpp:1836-1861)
  CostEstimateGroup.single_cost     <- CostEstimateGroup::singleCost   (slicetype.cpp:4148-4152)
  CostEstimateGroup.add / finish_batch <- ::add / ::finishBatch        (slicetype.cpp:4154-4196)

  LookaheadTLD.weights_analyse      <- LookaheadTLD::weightsAnalyse    (slicetype.cpp:870-971)
  CuTree.cu_tree / estimate_cu_propagate / cu_tree_finish
                                    <- Lookahead::cuTree, ::estimateCUPropagate, ::cuTreeFinish
                                       (slicetype.cpp:3665-3874, 4016-4064)

The cache check, the bDoSearch decision from the 0x7FFF sentinel, the B-frame
bias and the intra penalty of estimateFrameCost (slicetype.cpp:4284-4383) are
host logic and live here; pixels never do.  Each frame of the window owns one
device slot.
"""
import numpy as np

_f32 = np.float32

# X265_TYPE_* (source/x265.h)
TYPE_AUTO, TYPE_IDR, TYPE_I, TYPE_P, TYPE_BREF, TYPE_B = range(6)


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
        self.slice_type = TYPE_AUTO                              # Lowres::sliceType
        self.qp_aq_offset = None                                 # float64[n], from calcAdaptiveQuantFrame
        self.qp_cutree_offset = None


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
        self.batch_mode = False         # m_batchMode: set by the first add(), as in the reference

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
        # outside batch mode a request is split into the lookahead's cooperative slices (slicetype.cpp:4312)
        coop = int(not self.batch_mode)
        return (self.frames[p0].slot, fenc.slot, self.frames[p1].slot, p0, b, p1, do0, do1, int(weighted), coop)

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
        self.batch_mode = True
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


def _clip_duration(f):
    """CLIP_DURATION (source/encoder/ratecontrol.h:44-49)"""
    return min(max(f, 0.01), 1.00)


class CuTree:
    """Lookahead::cuTree: the walk over the mini-GOPs is host logic; the per-block passes run on the device.

    Parameters follow x265_param: fps_num / fps_denom, b_pyramid (bBPyramid), weighted_bipred
    (bEnableWeightedBiPred), lookahead_depth, vbv_buffer_size (rc.vbvBufferSize), q_compress (rc.qCompress)."""

    def __init__(self, ctx, fps_num=30, fps_denom=1, b_pyramid=False, weighted_bipred=False, lookahead_depth=20,
                 vbv_buffer_size=0, q_compress=0.6):
        self.ctx = ctx
        self.frame_duration = float(fps_denom) / fps_num
        self.b_pyramid = b_pyramid
        self.weighted_bipred = weighted_bipred
        self.lookahead_depth = lookahead_depth
        self.vbv = vbv_buffer_size
        self.strength = 5.0 * (1.0 - q_compress)                 # m_cuTreeStrength, slicetype.cpp:1075

    def cu_tree_finish(self, frame, average_duration, ref0_distance):
        fps_factor = int(_clip_duration(average_duration) / _clip_duration(self.frame_duration) * 256)
        delta = 0.0
        if ref0_distance and frame.weighted_cost_delta.get(ref0_distance - 1, 0) > 0:
            delta = 1.0 - frame.weighted_cost_delta[ref0_distance - 1]
        self.ctx.cutree_finish(frame.slot, fps_factor, delta, self.strength, frame.qp_aq_offset, frame.qp_cutree_offset)

    def estimate_cu_propagate(self, frames, average_duration, p0, p1, b, referenced):
        dist_scale = (((b - p0) << 8) + ((p1 - p0) >> 1)) // (p1 - p0)
        bipred_weight = 64 - (dist_scale >> 2) if self.weighted_bipred else 32
        fps_factor = _clip_duration(self.frame_duration) / _clip_duration(average_duration)
        self.ctx.cutree_propagate(frames[p0].slot, frames[p1].slot, frames[b].slot, b - p0, p1 - b, referenced,
                                  fps_factor, bipred_weight)
        if self.vbv and self.lookahead_depth and referenced:
            self.cu_tree_finish(frames[b], average_duration, b - p0 if b == p1 else 0)

    def cu_tree(self, frames, numframes, b_intra):
        ctx = self.ctx
        idx = 0 if b_intra else 1
        bframes = 0
        total = 0.0
        for _ in range(numframes + 1):
            total += self.frame_duration
        average_duration = total / (numframes + 1)

        i = numframes
        while i > 0 and frames[i].slice_type == TYPE_B:
            i -= 1
        lastnonb = i

        if not self.lookahead_depth:
            if b_intra:
                ctx.cutree_clear(frames[0].slot)
                frames[0].qp_cutree_offset[:] = frames[0].qp_aq_offset
                return
            ctx.cutree_swap(frames[lastnonb].slot, frames[0].slot)
            ctx.cutree_clear(frames[0].slot)
        else:
            if lastnonb < idx:
                return
            ctx.cutree_clear(frames[lastnonb].slot)

        est = CostEstimateGroup(ctx, frames)
        while i > idx:
            i -= 1
            curnonb = i
            while frames[curnonb].slice_type == TYPE_B and curnonb > 0:
                curnonb -= 1
            if curnonb < idx:
                break
            est.single_cost(curnonb, lastnonb, lastnonb)
            ctx.cutree_clear(frames[curnonb].slot)
            bframes = lastnonb - curnonb - 1
            if self.b_pyramid and bframes > 1:
                middle = (bframes + 1) // 2 + curnonb
                est.single_cost(curnonb, lastnonb, middle)
                ctx.cutree_clear(frames[middle].slot)
                while i > curnonb:
                    p0 = middle if i > middle else curnonb
                    p1 = middle if i < middle else lastnonb
                    if i != middle:
                        est.single_cost(p0, p1, i)
                        self.estimate_cu_propagate(frames, average_duration, p0, p1, i, 0)
                    i -= 1
                self.estimate_cu_propagate(frames, average_duration, curnonb, lastnonb, middle, 1)
            else:
                while i > curnonb:
                    est.single_cost(curnonb, lastnonb, i)
                    self.estimate_cu_propagate(frames, average_duration, curnonb, lastnonb, i, 0)
                    i -= 1
            self.estimate_cu_propagate(frames, average_duration, curnonb, lastnonb, lastnonb, 1)
            lastnonb = curnonb

        if not self.lookahead_depth:
            est.single_cost(0, lastnonb, lastnonb)
            self.estimate_cu_propagate(frames, average_duration, 0, lastnonb, lastnonb, 1)
            ctx.cutree_swap(frames[lastnonb].slot, frames[0].slot)

        self.cu_tree_finish(frames[lastnonb], average_duration, lastnonb)
        if self.b_pyramid and bframes > 1 and not self.vbv:
            self.cu_tree_finish(frames[lastnonb + (bframes + 1) // 2], average_duration, 0)
