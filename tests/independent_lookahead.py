"""Test infrastructure: a second, independent derivation of the lookahead's per-block cost estimate in plain
Python, written from the x265 sources alone and structured differently from the C oracle (arrays and slices
instead of pointers, one block at a time, no shared helpers).  It exists to catch a misreading that the oracle
and the kernels could share.  Parity unpinned: the reference tree holds no output fixture for this path.

Restated here, for 8x8 lowres blocks without HME, weights or cooperative slices:
  CostEstimateGroup::estimateFrameCost   source/encoder/slicetype.cpp:4284-4383
  CostEstimateGroup::estimateCUCost      source/encoder/slicetype.cpp:4385-4557
  MotionEstimate::motionEstimate         source/encoder/motion.cpp:923-1012 (start), 1041-1140 (hexagon),
                                         1644-1699 and 1759-1772 (lowres sub-pel refinement, zero-vector override)
  ReferencePlanes::lowresMC / lowresQPelCost   source/common/lowres.h:72-117
  sad, satd_8x4 / satd8, pixelavg_pp     source/common/pixel.cpp:238-289, 538-549
Inputs taken as given: the four half-pel planes of every frame, the intra costs, the AQ scale factors and the
lambda-scaled MV cost table."""
import numpy as np

COST_MAX = 1 << 28
LOWRES_COST_MASK, LOWRES_COST_SHIFT = (1 << 14) - 1, 14
MERANGE = 16                                   # Lookahead::s_merange (slicetype.h:337)
HEX2 = [(-1, -2), (-2, 0), (-1, 2), (1, 2), (2, 0), (1, -2), (-1, -2), (-2, 0)]
MOD6M1 = [5, 0, 1, 2, 3, 4, 5, 0]
SQUARE1 = [(0, 0), (0, -1), (0, 1), (-1, 0), (1, 0), (-1, -1), (-1, 1), (1, -1), (1, 1)]
HPEL_DIRS, QPEL_DIRS = 4, 4                    # workload[1]: the lookahead searches with subpelRefine 1
H4 = np.array([[1, 1, 1, 1], [1, 1, -1, -1], [1, -1, -1, 1], [1, -1, 1, -1]], np.int64)


def sad(a, b):
    return int(np.abs(a.astype(np.int64) - b.astype(np.int64)).sum())


def satd8x8(a, b):
    """satd8<8,8>: per 8x4 half, two 4x4 Hadamard transforms, the absolute sum halved"""
    d = a.astype(np.int64) - b.astype(np.int64)
    total = 0
    for row in (0, 4):
        s = 0
        for col in (0, 4):
            s += int(np.abs(H4 @ d[row:row + 4, col:col + 4] @ H4.T).sum())
        total += s >> 1
    return total


def avg(a, b):
    return (a.astype(np.int64) + b.astype(np.int64) + 1) >> 1


class Frame:
    def __init__(self, planes, oy, ox, intra_cost, inv_qscale, ncu, rows, nd):
        self.planes, self.oy, self.ox = planes, oy, ox
        self.intra_cost, self.inv_qscale = intra_cost, inv_qscale
        self.mvs = [[np.full((ncu, 2), 0x7FFF, np.int64) for _ in range(nd)] for _ in range(2)]
        self.mv_costs = [[np.zeros(ncu, np.int64) for _ in range(nd)] for _ in range(2)]
        self.lowres_costs = [[np.zeros(ncu, np.int64) for _ in range(nd)] for _ in range(nd)]
        self.row_satds = [[np.full(rows, -1, np.int64) for _ in range(nd)] for _ in range(nd)]
        self.cost_est = [[-1] * nd for _ in range(nd)]
        self.cost_est_aq = [[-1] * nd for _ in range(nd)]
        self.intra_mbs = [0] * nd

    def block(self, plane, x, y):
        """8x8 pixels of a half-pel plane at picture position (x, y)"""
        y0, x0 = self.oy + y, self.ox + x
        assert y0 >= 0 and x0 >= 0, "read outside the padded plane"
        out = self.planes[plane][y0:y0 + 8, x0:x0 + 8]
        assert out.shape == (8, 8), "read outside the padded plane"
        return out

    def mc(self, x, y, qmv):
        """ReferencePlanes::lowresMC: the prediction at a quarter-pel vector"""
        qx, qy = qmv
        if (qx | qy) & 1:
            a = self.block((qy & 2) | ((qx & 2) >> 1), x + (qx >> 2), y + (qy >> 2))
            rx, ry = qx + (qx & 1), qy + (qy & 1)
            b = self.block((ry & 2) | ((rx & 2) >> 1), x + (rx >> 2), y + (ry >> 2))
            return avg(a, b)
        return self.block((qy & 2) | ((qx & 2) >> 1), x + (qx >> 2), y + (qy >> 2))


class IndependentLookahead:
    def __init__(self, frames_planes, origin, width_in_cu, height_in_cu, intra_costs, inv_qscales, cost_table, centre,
                 bframes=4, bframe_bias=0):
        self.w, self.h = width_in_cu, height_in_cu
        nd = bframes + 2
        oy, ox = origin
        self.frames = [Frame(p, oy, ox, ic, iq, self.w * self.h, self.h, nd)
                       for p, ic, iq in zip(frames_planes, intra_costs, inv_qscales)]
        self.tab, self.c = cost_table, centre
        self.bias = bframe_bias

    def mvcost(self, q, mvp):
        return (int(self.tab[self.c + q[0] - mvp[0]]) + int(self.tab[self.c + q[1] - mvp[1]])) & 0xFFFF

    # ---- MotionEstimate::motionEstimate on a lowres reference, hexagon search ----
    def motion_estimate(self, fenc, ref, x, y, mvmin, mvmax, qmvp):
        def fpel_sad(mx, my):
            return sad(fenc, ref.block(0, x + mx, y + my))

        def cost_at(mx, my):
            return fpel_sad(mx, my) + self.mvcost((mx << 2, my << 2), qmvp)

        def y_ok(v):
            return mvmin[1] <= v <= mvmax[1]

        def in_range(mx, my):
            return mvmin[0] <= mx <= mvmax[0] and mvmin[1] <= my <= mvmax[1]

        qmin, qmax = (mvmin[0] << 2, mvmin[1] << 2), (mvmax[0] << 2, mvmax[1] << 2)
        pmv = tuple(max(min(v, hi), lo) for v, lo, hi in zip(qmvp, qmin, qmax))
        bestpre = pmv
        bprecost = sad(fenc, ref.mc(x, y, pmv))
        bx, by = (pmv[0] + 2) >> 2, (pmv[1] + 2) >> 2
        bcost = bprecost
        if (pmv[0] | pmv[1]) & 3:
            bcost = cost_at(bx, by)
        if pmv != (0, 0):
            c = fpel_sad(0, 0) + self.mvcost((0, 0), qmvp)
            if c < bcost:
                bcost, bx, by = c, 0, max(min(0, mvmax[1]), mvmin[1])
        if bcost == 0:
            return (bx << 2, by << 2), self.mvcost((bx << 2, by << 2), qmvp)

        # hexagon, radius 2
        c = [cost_at(bx - 2, by), cost_at(bx - 1, by + 2), cost_at(bx + 1, by + 2)]
        bcost <<= 3
        if y_ok(by):
            bcost = min(bcost, (c[0] << 3) + 2)
        if y_ok(by + 2):
            bcost = min(bcost, (c[1] << 3) + 3)
            bcost = min(bcost, (c[2] << 3) + 4)
        c = [cost_at(bx + 2, by), cost_at(bx + 1, by - 2), cost_at(bx - 1, by - 2)]
        if y_ok(by):
            bcost = min(bcost, (c[0] << 3) + 5)
        if y_ok(by - 2):
            bcost = min(bcost, (c[1] << 3) + 6)
            bcost = min(bcost, (c[2] << 3) + 7)
        if bcost & 7:
            d = (bcost & 7) - 2
            if y_ok(by + HEX2[d + 1][1]):
                bx, by = bx + HEX2[d + 1][0], by + HEX2[d + 1][1]
                i = (MERANGE >> 1) - 1
                while i > 0 and in_range(bx, by):
                    c = [cost_at(bx + HEX2[d + k][0], by + HEX2[d + k][1]) for k in range(3)]
                    bcost &= ~7
                    for k in range(3):
                        if y_ok(by + HEX2[d + k][1]):
                            bcost = min(bcost, (c[k] << 3) + k + 1)
                    if not bcost & 7:
                        break
                    d += (bcost & 7) - 2
                    d = MOD6M1[d + 1]
                    bx, by = bx + HEX2[d + 1][0], by + HEX2[d + 1][1]
                    i -= 1
        bcost >>= 3

        # square refine
        best = 0
        c = [cost_at(bx, by - 1), cost_at(bx, by + 1), cost_at(bx - 1, by), cost_at(bx + 1, by)]
        if y_ok(by - 1) and c[0] < bcost:
            bcost, best = c[0], 1
        if y_ok(by + 1) and c[1] < bcost:
            bcost, best = c[1], 2
        if c[2] < bcost:
            bcost, best = c[2], 3
        if c[3] < bcost:
            bcost, best = c[3], 4
        c = [cost_at(bx - 1, by - 1), cost_at(bx - 1, by + 1), cost_at(bx + 1, by - 1), cost_at(bx + 1, by + 1)]
        for k, dy in enumerate((-1, 1, -1, 1)):
            if y_ok(by + dy) and c[k] < bcost:
                bcost, best = c[k], 5 + k
        bx, by = bx + SQUARE1[best][0], by + SQUARE1[best][1]

        # the better of the measured predictor and the integer search, then sub-pel refinement
        if bprecost < bcost:
            bmv, bcost = bestpre, bprecost
        else:
            bmv = (bx << 2, by << 2)
        if not bcost:
            bcost = self.mvcost(bmv, qmvp)
        else:
            best = 0
            for i in range(1, HPEL_DIRS + 1):
                q = (bmv[0] + SQUARE1[i][0] * 2, bmv[1] + SQUARE1[i][1] * 2)
                if q[1] < qmin[1] or q[1] > qmax[1]:
                    continue
                c1 = sad(fenc, ref.mc(x, y, q)) + self.mvcost(q, qmvp)
                if c1 < bcost:
                    bcost, best = c1, i
            bmv = (bmv[0] + SQUARE1[best][0] * 2, bmv[1] + SQUARE1[best][1] * 2)
            bcost = satd8x8(fenc, ref.mc(x, y, bmv)) + self.mvcost(bmv, qmvp)
            best = 0
            for i in range(1, QPEL_DIRS + 1):
                q = (bmv[0] + SQUARE1[i][0], bmv[1] + SQUARE1[i][1])
                if q[1] < qmin[1] or q[1] > qmax[1]:
                    continue
                c1 = satd8x8(fenc, ref.mc(x, y, q)) + self.mvcost(q, qmvp)
                if c1 < bcost:
                    bcost, best = c1, i
            bmv = (bmv[0] + SQUARE1[best][0], bmv[1] + SQUARE1[best][1])
        if bmv != (0, 0):
            c0 = satd8x8(fenc, ref.block(0, x, y)) + self.mvcost((0, 0), qmvp)
            if c0 <= bcost:
                bmv = (0, 0)                     # the vector changes, the returned cost does not
        return bmv, bcost

    # ---- CostEstimateGroup::estimateCUCost ----
    def estimate_cu_cost(self, cu_x, cu_y, p0, p1, b, do_search, last_row):
        fr0, fr1, fe = self.frames[p0], self.frames[p1], self.frames[b]
        w, h = self.w, self.h
        bidir = b < p1
        xy = cu_x + cu_y * w
        x, y = 8 * cu_x, 8 * cu_y
        fenc = fe.block(0, x, y)
        dist = (b - p0, p1 - b)
        mvmin = (-cu_x * 8 - 8, -cu_y * 8 - 8)
        mvmax = ((w - cu_x - 1) * 8 + 8, (h - cu_y - 1) * 8 + 8)
        bcost, listused = COST_MAX, 0
        for i in range(1 + bidir):
            mvs, costs = fe.mvs[i][dist[i]], fe.mv_costs[i][dist[i]]
            if not do_search[i]:
                if costs[xy] < bcost:
                    bcost, listused = int(costs[xy]), i + 1
                continue
            ref = fr1 if i else fr0
            skip_cost = None
            cand = []
            if cu_x < w - 1:
                cand.append(tuple(int(v) for v in mvs[xy + 1]))
            if not last_row:
                cand.append(tuple(int(v) for v in mvs[xy + w]))
                if cu_x > 0:
                    cand.append(tuple(int(v) for v in mvs[xy + w - 1]))
                if cu_x < w - 1:
                    cand.append(tuple(int(v) for v in mvs[xy + w + 1]))
            mvp = (0, 0)
            mvpcost = COST_MAX
            for m in cand:
                c = satd8x8(fenc, ref.mc(x, y, m))
                if c < mvpcost:
                    mvpcost, mvp = c, m
                if mvp == (0, 0) and bidir:
                    skip_cost = c
            mv, cost = self.motion_estimate(fenc, ref, x, y, mvmin, mvmax, mvp)
            if skip_cost is not None and skip_cost < 64 and skip_cost < cost and bidir:
                cost, mv = skip_cost, (0, 0)
            mvs[xy] = mv
            costs[xy] = cost
            if cost < bcost:
                bcost, listused = cost, i + 1
        if bidir:
            m0 = tuple(int(v) for v in fe.mvs[0][dist[0]][xy])
            m1 = tuple(int(v) for v in fe.mvs[1][dist[1]][xy])
            bi = satd8x8(fenc, avg(fr0.mc(x, y, m0), fr1.mc(x, y, m1)))
            if bi < bcost:
                bcost, listused = bi, 3
            bi = satd8x8(fenc, avg(fr0.block(0, x, y), fr1.block(0, x, y)))
            if bi < bcost:
                bcost, listused = bi, 3
            bcost += 4
        else:
            bcost += 4
            if fe.intra_cost[xy] < bcost:
                bcost, listused = int(fe.intra_cost[xy]), 0
        scored = (0 < cu_x < w - 1 and 0 < cu_y < h - 1) or w <= 2 or h <= 2
        bcost_aq = bcost
        if scored and fe.inv_qscale is not None:
            bcost_aq = (bcost * int(fe.inv_qscale[xy]) + 128) >> 8
        d0, d1 = dist
        if scored:
            fe.cost_est[d0][d1] += bcost
            fe.cost_est_aq[d0][d1] += bcost_aq
            if not listused and not bidir:
                fe.intra_mbs[d0] += 1
        fe.row_satds[d0][d1][cu_y] += bcost_aq
        fe.lowres_costs[d0][d1][xy] = min(bcost, LOWRES_COST_MASK) | (listused << LOWRES_COST_SHIFT)

    # ---- CostEstimateGroup::estimateFrameCost ----
    def estimate_frame_cost(self, p0, p1, b):
        fe = self.frames[b]
        d0, d1 = b - p0, p1 - b
        if fe.cost_est[d0][d1] >= 0 and fe.row_satds[d0][d1][0] != -1:
            return fe.cost_est[d0][d1]
        do_search = (fe.mvs[0][d0][0][0] == 0x7FFF, p1 > b and fe.mvs[1][d1][0][0] == 0x7FFF)
        fe.cost_est[d0][d1] = 0
        fe.cost_est_aq[d0][d1] = 0
        last_row = True
        for cu_y in range(self.h - 1, -1, -1):
            fe.row_satds[d0][d1][cu_y] = 0
            for cu_x in range(self.w - 1, -1, -1):
                self.estimate_cu_cost(cu_x, cu_y, p0, p1, b, do_search, last_row)
            last_row = False
        score = fe.cost_est[d0][d1]
        if b != p1:
            score = score * 100 // (130 + self.bias)
        fe.cost_est[d0][d1] = score
        return score
