"""Host-side mirror of the lookahead's cost-estimation interface.

Mirrors, name for name, the pieces of x265's host code that sit directly above
the offload boundary, so that parity tests read like calls into the reference:

  PreLookaheadGroup.process_tasks   <- PreLookaheadGroup::processTasks (slicetype.cpp:1836-1861)
  CostEstimateGroup.single_cost     <- CostEstimateGroup::singleCost   (slicetype.cpp:4148-4152)
  CostEstimateGroup.add / finish_batch <- ::add / ::finishBatch        (slicetype.cpp:4154-4196)

The cache check, the bDoSearch decision from the 0x7FFF sentinel, the B-frame
bias and the intra penalty of estimateFrameCost (slicetype.cpp:4284-4383) are
host logic and live here; pixels never do.  Each frame of the window owns one
device slot.
"""


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
    def __init__(self, ctx, frames, ncu_interior=None):
        self.ctx = ctx
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
        return (self.frames[p0].slot, fenc.slot, self.frames[p1].slot, p0, b, p1, do0, do1, 0)

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
