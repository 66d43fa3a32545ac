"""Test infrastructure: --threaded-me's per-CTU motion estimation, restated serially in x265's own call order
over the CPU oracle's searches (parity unpinned: the reference tree holds no output fixture for this path).

Follows Analysis::deriveMVsForCTU / computeMVForPUs (source/encoder/analysis.cpp:161-306) and
Search::puMotionEstimation (source/encoder/search.cpp:226-560) statement by statement for a P or B slice: one CTU
after another, children before parents, one search at a time, each reading the CTU's result table as the calls
before it left it.  The host layer under test reorders the same calls into dependency levels over many CTUs."""
import numpy as np

from x265_amd.api import ME_DATA, MAX_NUM_PUS_PER_CTU

SIZE_2Nx2N, SIZE_2NxN, SIZE_Nx2N, SIZE_NxN, SIZE_2NxnU, SIZE_2NxnD, SIZE_nLx2N, SIZE_nRx2N = range(8)
MD_LEFT, MD_ABOVE, MD_ABOVE_RIGHT, MD_BELOW_LEFT, MD_ABOVE_LEFT = range(5)
MAX_UINT = 0xFFFFFFFF
U32 = 0xFFFFFFFF

# g_puLookup (source/encoder/threadedme.h:67-92): width, height, partition, AMP
PU_LOOKUP = [
    (8, 4, SIZE_2NxN, 0), (4, 8, SIZE_Nx2N, 0), (8, 8, SIZE_2Nx2N, 0),
    (16, 4, SIZE_2NxnU, 1), (16, 12, SIZE_2NxnD, 1), (4, 16, SIZE_nLx2N, 1), (12, 16, SIZE_nRx2N, 1),
    (16, 8, SIZE_2NxN, 0), (8, 16, SIZE_Nx2N, 0), (16, 16, SIZE_2Nx2N, 0),
    (32, 8, SIZE_2NxnU, 1), (32, 24, SIZE_2NxnD, 1), (8, 32, SIZE_nLx2N, 1), (24, 32, SIZE_nRx2N, 1),
    (32, 16, SIZE_2NxN, 0), (16, 32, SIZE_Nx2N, 0), (32, 32, SIZE_2Nx2N, 0),
    (64, 16, SIZE_2NxnU, 1), (64, 48, SIZE_2NxnD, 1), (16, 64, SIZE_nLx2N, 1), (48, 64, SIZE_nRx2N, 1),
    (64, 32, SIZE_2NxN, 0), (32, 64, SIZE_Nx2N, 0), (64, 64, SIZE_2Nx2N, 0),
]


def part_rect(part, pu_idx, s):
    """CUData::getPartIndexAndSize: (x, y, w, h) of a part inside a CU of size s"""
    q, h = s // 4, s // 2
    return {
        SIZE_2Nx2N: [(0, 0, s, s)],
        SIZE_2NxN: [(0, 0, s, h), (0, h, s, h)],
        SIZE_Nx2N: [(0, 0, h, s), (h, 0, h, s)],
        SIZE_2NxnU: [(0, 0, s, q), (0, q, s, s - q)],
        SIZE_2NxnD: [(0, 0, s, s - q), (0, s - q, s, q)],
        SIZE_nLx2N: [(0, 0, q, s), (q, 0, s - q, s)],
        SIZE_nRx2N: [(0, 0, s - q, s), (s - q, 0, q, s)],
    }[part][pu_idx]


def clip3(lo, hi, v):
    return lo if v < lo else hi if v > hi else v


def scale_mv_by_poc_dist(mv, cur_poc, cur_ref_poc, col_poc, col_ref_poc):
    """CUData::scaleMvByPOCDist + scaleMv (source/common/cudata.cpp:104-110, 2229-2244)"""
    d, b = col_poc - col_ref_poc, cur_poc - cur_ref_poc
    if d == b:
        return mv
    tdb, tdd = clip3(-128, 127, b), clip3(-128, 127, d)
    x = int((0x4000 + abs(int(tdd / 2))) / tdd)           # C division truncates toward zero
    scale = clip3(-4096, 4095, (tdb * x + 32) >> 6)
    return tuple(clip3(-32768, 32767, (scale * c + 127 + (scale * c < 0)) >> 8) for c in mv)


class SerialTME:
    def __init__(self, me_per_ref, ref_pocs, poc, pic_w, pic_h, bitsizes, cost_table, lam, method=3, merange=57,
                 subpel=3, amp=False, rect=True, ctu=64, min_cu=8, bframes=4, lowres_mvs=None, ref_is_intra=None,
                 ref_tables=None, median_mv=None, ref_lag_pixels=None, num_l0=None, cu_qp=None, area_qp=None,
                 qp_sets=None, temporal=None):
        """me_per_ref: one oracle FullResME (source PU in the current picture, searches in that reference) per
        reference, list 0's first (num_l0 of them; all of them for a P slice), then list 1's; bitsizes / cost_table:
        (array, centre) pairs; lowres_mvs[r]: (blocks, 2) or None.
        Delta QP: cu_qp[addr] = the QPs of the CTU's CUs (depth 0, then depth 1 in raster order, ...), area_qp[addr] =
        the five QPs of the area searches (default: the depth-0 and depth-1 CUs'), qp_sets[qp] = (me_per_ref,
        cost_table, lam) under that QP's lambda.  temporal: (CTUs, positions, 2) records of the temporal candidate."""
        self.me, self.ref_pocs, self.poc = me_per_ref, list(ref_pocs), poc
        self.cu_qp, self.area_qp, self.qp_sets, self.temporal = cu_qp, area_qp, qp_sets, temporal
        self.num_l0 = len(self.ref_pocs) if num_l0 is None else num_l0
        self.is_p = self.num_l0 == len(self.ref_pocs)
        self.list_pocs = [self.ref_pocs[:self.num_l0], self.ref_pocs[self.num_l0:]]
        self.pic_w, self.pic_h = pic_w, pic_h
        self.bs, self.bs0 = bitsizes
        self.ct, self.ct0 = cost_table
        self.lambda_fix8 = int(np.floor(256.0 * lam))
        self.method, self.merange, self.subpel = method, merange, subpel
        self.amp, self.rect, self.ctu, self.min_cu, self.bframes = amp, rect, ctu, min_cu, bframes
        n = len(self.ref_pocs)
        self.lowres_mvs = lowres_mvs or [None] * n
        self.ref_is_intra = ref_is_intra or [True] * n
        self.ref_tables = ref_tables or [None] * n
        self.median_mv = median_mv
        self.ref_lag = pic_h if ref_lag_pixels is None else ref_lag_pixels
        self.cols, self.rows = -(-pic_w // ctu), -(-pic_h // ctu)
        self.table = np.zeros((self.cols * self.rows, MAX_NUM_PUS_PER_CTU), ME_DATA)
        self.table["ref"] = -1
        self.lowres_in_row = ((pic_w // 2) + 7) >> 3
        self.searches = 0
        # ThreadedME::initPuStartIdx (source/encoder/threadedme.cpp:84-106)
        self.start_idx, at = {}, 0
        for w, h, part, is_amp in PU_LOOKUP:
            if w > ctu or h > ctu:
                continue
            iw = max(w, h) if is_amp else w
            ih = iw if is_amp else h
            self.start_idx[(w + h, part)] = at
            at += (2 if is_amp else 1) * (ctu // iw) * (ctu // ih)

    # ---- BitCost / RDCost ----
    def bitcost(self, mv, mvp):
        return int(self.bs[self.bs0 + mv[0] - mvp[0]] + self.bs[self.bs0 + mv[1] - mvp[1]] + np.float32(0.5))

    def mvcost(self, mv, mvp):
        return (int(self.ct[self.ct0 + mv[0] - mvp[0]]) + int(self.ct[self.ct0 + mv[1] - mvp[1]])) & 0xFFFF

    def get_cost(self, bits):
        return ((bits * self.lambda_fix8 + 128) >> 8) & U32

    # ---- CUData helpers ----
    def clip_mv(self, cu_x, cu_y, mv):
        xmax, xmin = (self.pic_w + 8 - cu_x - 1) << 2, -((self.ctu + 8 + cu_x - 1) << 2)
        ymax, ymin = (self.pic_h + 8 - cu_y - 1) << 2, -((self.ctu + 8 + cu_y - 1) << 2)
        return (min(xmax, max(xmin, mv[0])), min(ymax, max(ymin, mv[1])))

    def set_search_range(self, cu_x, cu_y, mvp, merange):
        """Search::setSearchRange (search.cpp:4969-5022): one slice, no intra refresh, no current-picture reference"""
        d = merange << 2
        mn = self.clip_mv(cu_x, cu_y, (mvp[0] - d, mvp[1] - d))
        mx = self.clip_mv(cu_x, cu_y, (mvp[0] + d, mvp[1] + d))
        lim = (1 << 15) - 1
        mn = (max(mn[0], -lim) >> 2, max(mn[1], -lim) >> 2)
        mx = (min(mx[0], lim) >> 2, min(mx[1], lim) >> 2)
        mn = (mn[0], min(mn[1], self.ref_lag))
        mx = (mx[0], max(min(mx[1], self.ref_lag), mn[1]))
        return mn, mx

    def set_lambda_from_qp(self, qp):
        """Search::setLambdaFromQP (search.cpp:214-224): the MV-cost table and the fixed-point lambda of that QP"""
        self.me, (self.ct, self.ct0), lam = self.qp_sets[int(qp)]
        self.lambda_fix8 = int(np.floor(256.0 * lam))

    def cu_index(self, rel_x, rel_y, cu_size):
        at, size = 0, self.ctu
        while size > cu_size:
            at += (self.ctu // size) ** 2
            size //= 2
        return at + (rel_y // cu_size) * (self.ctu // cu_size) + rel_x // cu_size

    def get_pmv(self, neighbours, pic_list, ref_idx, temporal=None):
        """CUData::getPMV (cudata.cpp:1845-1905, 1959-1983), with getDirectPMV /
        getIndirectPMV (:2110-2156): the asked list of the neighbour first, then the other one;
        neighbours[dir] = ((mv of list 0, mv of list 1), (refIdx of list 0, refIdx of list 1)) or None"""
        cur_ref_poc = self.list_pocs[pic_list][ref_idx]
        direct, indirect = {}, {}
        for d in range(MD_ABOVE_LEFT + 1):
            nb = neighbours[d]
            if nb is None:
                continue
            for pl in (pic_list, 1 - pic_list):
                if nb[1][pl] >= 0 and self.list_pocs[pl][nb[1][pl]] == cur_ref_poc:
                    direct[d] = nb[0][pl]
                    break
            for pl in (pic_list, 1 - pic_list):
                if nb[1][pl] >= 0:
                    indirect[d] = scale_mv_by_poc_dist(nb[0][pl], self.poc, cur_ref_poc, self.poc, self.list_pocs[pl][nb[1][pl]])
                    break
        amvp = []
        for src, d in ((direct, MD_BELOW_LEFT), (direct, MD_LEFT), (indirect, MD_BELOW_LEFT), (indirect, MD_LEFT)):
            if d in src:
                amvp.append(src[d])
                break
        added_left = len(amvp) > 0
        for d in (MD_ABOVE_RIGHT, MD_ABOVE, MD_ABOVE_LEFT):
            if d in direct:
                amvp.append(direct[d])
                break
        if not added_left:
            for d in (MD_ABOVE_RIGHT, MD_ABOVE, MD_ABOVE_LEFT):
                if d in indirect:
                    amvp.append(indirect[d])
                    break
        mvc = []
        for d in range(MD_ABOVE_LEFT + 1):
            if d in direct and direct[d] != (0, 0):
                mvc.append(direct[d])
            if d in indirect and indirect[d] != (0, 0):
                mvc.append(indirect[d])
        if len(amvp) == 2 and amvp[0] == amvp[1]:
            amvp.pop()
        if temporal is not None and temporal["valid"] and len(amvp) < 2:
            t = scale_mv_by_poc_dist((int(temporal["mv"][0]), int(temporal["mv"][1])), self.poc, cur_ref_poc,
                                     int(temporal["colPoc"]), int(temporal["colRefPoc"]))
            amvp.append(t)
            mvc.append(t)
        while len(amvp) < 2:
            amvp.append((0, 0))
        return amvp, mvc

    def get_blk_bits(self, part, pu_idx, last_mode):
        """Search::getBlkBits (search.cpp:4894-4940)"""
        if part in (SIZE_2Nx2N, SIZE_NxN):
            return [1 if self.is_p else 3, 3, 5]
        if self.is_p:
            return [3, 0, 0]
        if part in (SIZE_2NxN, SIZE_2NxnU, SIZE_2NxnD):
            return [[[0, 0, 3], [0, 0, 0], [0, 0, 0]], [[5, 7, 7], [7, 5, 7], [6, 6, 6]]][pu_idx][last_mode]
        return [[[0, 2, 3], [0, 0, 0], [0, 0, 0]], [[5, 7, 7], [5, 5, 7], [6, 6, 6]]][pu_idx][last_mode]

    def get_lowres_mv(self, bx, by, ref):
        """Search::getLowresMV (search.cpp:2323-2343)"""
        if abs(self.poc - self.ref_pocs[ref]) > self.bframes + 1 or self.lowres_mvs[ref] is None:
            return (0, 0)
        v = self.lowres_mvs[ref][(by >> 4) * self.lowres_in_row + (bx >> 4)]
        return (int(v[0]) << 1, int(v[1]) << 1)

    # ---- Search::puMotionEstimation ----
    def pu_motion_estimation(self, addr, cu, part, area_idx, final_idx, pu_offset, is_mvp, neighbor_idx=None):
        cu_x, cu_y, cu_size = cu
        search_range = 32 if is_mvp else self.merange
        num_refs = len(self.ref_pocs)
        mvp = (0, 0)
        best = [dict(cost=MAX_UINT), dict(cost=MAX_UINT)]  # bestME[2], declared outside the loop over the parts
        last_mode = 0
        entries = self.table[addr]
        for pu_idx in range(1 if part == SIZE_2Nx2N else 2):
            rx, ry, w, h = part_rect(part, pu_idx, cu_size)
            x, y = cu_x + rx, cu_y + ry
            pos = final_idx + pu_idx * pu_offset
            for ref in range(num_refs):                    # list 0's references, then list 1's
                pic_list = 0 if ref < self.num_l0 else 1
                ref_idx = ref - pic_list * self.num_l0
                num_idx = len(self.list_pocs[pic_list])
                me = self.me[ref]
                sel_bits = self.get_blk_bits(part, pu_idx, last_mode)
                bits = sel_bits[pic_list] + 1 + ref_idx + (ref_idx < num_idx - 1)
                if not is_mvp:
                    mvp = self.area_best[area_idx][ref]
                amvp, mvp_idx, mvc = [(0, 0), (0, 0)], 0, []
                if not is_mvp:
                    neighbours = []
                    for d in range(MD_ABOVE_LEFT + 1):
                        n = neighbor_idx[d]
                        e = entries[n] if n >= 0 else None
                        if e is not None and (e["ref"][0] >= 0 or e["ref"][1] >= 0):
                            neighbours.append((((int(e["mv"][0][0]), int(e["mv"][0][1])), (int(e["mv"][1][0]), int(e["mv"][1][1]))),
                                               (int(e["ref"][0]), int(e["ref"][1]))))
                        else:
                            neighbours.append(None)
                    cand, mvc = self.get_pmv(neighbours, pic_list, ref_idx,
                                             None if self.temporal is None else self.temporal[addr][pos][pic_list])
                    if mvc:
                        amvp = cand
                        if amvp[0] != amvp[1]:              # Search::selectMVP (search.cpp:2347-2383)
                            costs = [me.pred_sad(x, y, w, h, self.clip_mv(cu_x, cu_y, c)) for c in amvp]
                            mvp_idx = 0 if costs[0] <= costs[1] else 1
                        mvp = amvp[mvp_idx]
                    elif not self.ref_is_intra[ref] and self.ref_tables[ref] is not None:
                        md = self.ref_tables[ref][addr][pos]
                        r0, r1 = int(md["ref"][0]), int(md["ref"][1])
                        if r0 >= 0 and r1 >= 0:
                            mvp = (int(md["mv"][pic_list][0]), int(md["mv"][pic_list][1]))
                        elif r0 >= 0:
                            mvp = (int(md["mv"][0][0]), int(md["mv"][0][1]))
                        elif r1 >= 0:
                            mvp = (int(md["mv"][1][0]), int(md["mv"][1][1]))
                b_lowres, mvp_lowres = False, (0, 0)
                bx, by = x + (w >> 1), y + (h >> 1)
                if bx < self.pic_w and by < self.pic_h:
                    lmv = self.get_lowres_mv(bx, by, ref)
                    if lmv != (0, 0):
                        mvc = mvc + [lmv]
                        b_lowres = True
                    mvp_lowres = lmv
                mvmin, mvmax = self.set_search_range(cu_x, cu_y, mvp, search_range)
                if is_mvp:
                    outmv, _ = me.diamond_search(x, y, w, h, mvp, mvmin, mvmax)
                    self.searches += 1
                    self.area_best[area_idx][ref] = outmv
                    me_mvp = mvp
                    satd = 0
                else:
                    outmv, satd = me.estimate(x, y, w, h, mvp, mvmin, mvmax, self.method, self.merange, self.subpel, mvc)
                    self.searches += 1
                    me_mvp = mvp
                    if b_lowres and mvp_lowres != (0, 0) and mvp_lowres != mvp:
                        b_lowres = False
                        mn2, mx2 = self.set_search_range(cu_x, cu_y, mvp_lowres, self.merange)
                        out2, cost2 = me.estimate(x, y, w, h, mvp_lowres, mn2, mx2, self.method, self.merange, self.subpel, mvc)
                        self.searches += 1
                        me_mvp = mvp_lowres                  # motionEstimate leaves its predictor in MotionEstimate
                        if cost2 < satd:
                            outmv, satd, b_lowres = out2, cost2, True
                if is_mvp:
                    continue
                bits += self.bitcost(outmv, me_mvp)
                mv_cost = self.mvcost(outmv, me_mvp)
                cost = (((satd - mv_cost) & U32) + self.get_cost(bits)) & U32
                if b_lowres:                                 # Search::updateMVP (search.cpp:4961-4967)
                    diff = self.bitcost(outmv, mvp) - self.bitcost(outmv, mvp_lowres)
                    orig = bits
                    bits = (orig + diff) & U32
                    cost = (cost - self.get_cost(orig) + self.get_cost(bits)) & U32
                diff = self.bitcost(outmv, amvp[1 - mvp_idx]) - self.bitcost(outmv, amvp[mvp_idx])
                if diff < 0:                                 # Search::checkBestMVP (search.cpp:4947-4958)
                    mvp_idx = 1 - mvp_idx
                    orig = bits
                    bits = (orig + diff) & U32
                    cost = (cost - self.get_cost(orig) + self.get_cost(bits)) & U32
                mvp = amvp[mvp_idx]
                if cost < best[pic_list]["cost"]:
                    best[pic_list] = dict(mv=outmv, mvp=mvp, cost=cost, bits=bits, mvCost=mv_cost, ref=ref_idx)
            if is_mvp:
                return
            # the bidirectional candidate (search.cpp:415-511)
            sel_bits = self.get_blk_bits(part, pu_idx, last_mode)
            bidir_cost, bidir_bits, bidir_mv = MAX_UINT, 0, None
            restricted = cu_size == 8 and part != SIZE_2Nx2N            # CUData::isBipredRestriction
            if (not self.is_p and not restricted and part != SIZE_2Nx2N and best[0]["cost"] != MAX_UINT
                    and best[1]["cost"] != MAX_UINT):
                me0, me1 = self.me[best[0]["ref"]], self.me[self.num_l0 + best[1]["ref"]]
                bidir_mv = [best[0]["mv"], best[1]["mv"]]
                satd = me0.bipred_satd(x, y, w, h, best[0]["mv"], me1, best[1]["mv"])
                bidir_bits = best[0]["bits"] + best[1]["bits"] + sel_bits[2] - (sel_bits[0] + sel_bits[1])
                bidir_cost = (satd + self.get_cost(bidir_bits & U32)) & U32
                try_zero = best[0]["mv"] != (0, 0) or best[1]["mv"] != (0, 0)
                if try_zero:
                    mn, mx = self.set_search_range(cu_x, cu_y, (0, 0), max(self.pic_w, self.pic_h))
                    mx = (mx[0], mx[1] + 2)
                    for l in (0, 1):
                        p = best[l]["mvp"]
                        try_zero = try_zero and mn[0] * 4 <= p[0] <= mx[0] * 4 and mn[1] * 4 <= p[1] <= mx[1] * 4
                if try_zero:
                    satd = me0.bipred_satd(x, y, w, h, (0, 0), me1, (0, 0))
                    bits0 = (best[0]["bits"] - self.bitcost(best[0]["mv"], best[0]["mvp"]) + self.bitcost((0, 0), best[0]["mvp"])) & U32
                    bits1 = (best[1]["bits"] - self.bitcost(best[1]["mv"], best[1]["mvp"]) + self.bitcost((0, 0), best[1]["mvp"])) & U32
                    cost = (satd + self.get_cost(bits0) + self.get_cost(bits1)) & U32
                    if cost < bidir_cost:
                        bidir_mv = [(0, 0), (0, 0)]
                        bidir_cost = cost
                        bidir_bits = bits0 + bits1 + sel_bits[2] - (sel_bits[0] + sel_bits[1])
            o = entries[pos]
            o["ref"] = (-1, -1)
            if bidir_cost < best[0]["cost"] and bidir_cost < best[1]["cost"]:
                last_mode = 2
                for l in (0, 1):
                    o["mv"][l] = bidir_mv[l]
                    o["mvp"][l] = best[l]["mvp"]
                    o["mvCost"][l] = best[l]["mvCost"]
                    o["ref"][l] = best[l]["ref"]
                o["bits"] = ((bidir_bits + (1 << 31)) & U32) - (1 << 31)     # int bidirBits: unsigned sums, kept as int
                o["cost"] = bidir_cost
            else:
                l = 0 if best[0]["cost"] <= best[1]["cost"] else 1
                last_mode = l
                o["mv"][l] = best[l]["mv"]
                o["mvp"][l] = best[l]["mvp"]
                o["mvCost"][l] = best[l]["mvCost"]
                o["cost"] = best[l]["cost"]
                o["bits"] = ((best[l]["bits"] + (1 << 31)) & U32) - (1 << 31)     # uint32_t bits (checkBestMVP may wrap it), read as int
                o["ref"][l] = best[l]["ref"]

    # ---- Analysis::computeMVForPUs ----
    def compute_mv_for_pus(self, addr, ctu_x, ctu_y, rel_x, rel_y, cu_size):
        ctu = self.ctu
        if cu_size > self.min_cu:
            hs = cu_size // 2
            for sub in range(4):
                self.compute_mv_for_pus(addr, ctu_x, ctu_y, rel_x + (sub & 1) * hs, rel_y + (sub >> 1) * hs, hs)
        area_id = 0 if cu_size == ctu else (rel_x >= ctu // 2) + 2 * (rel_y >= ctu // 2) + 1
        if self.cu_qp is not None:
            self.set_lambda_from_qp(self.cu_qp[addr][self.cu_index(rel_x, rel_y, cu_size)])
        for w, h, part, is_amp in PU_LOOKUP:
            if w > cu_size or h > cu_size or (w != cu_size and h != cu_size):
                continue
            if is_amp and not self.amp:
                continue
            if not self.rect and w != h and not is_amp:
                continue
            start = self.start_idx[(w + h, part)]
            bw = max(w, h) if is_amp else w
            bh = bw if is_amp else h
            ncols, nrows = ctu // bw, ctu // bh
            pu_offset = nrows * ncols if is_amp else ncols if part == SIZE_2NxN else 1 if part == SIZE_Nx2N else 0
            col, row = rel_x // bw, rel_y // bh
            final_idx = start + row * ncols + col
            left = start + row * ncols + col - 1 if col > 0 else -1
            above = start + (row - 1) * ncols + col if row > 0 else -1
            above_left = start + (row - 1) * ncols + col - 1 if row > 0 and col > 0 else -1
            above_right = start + (row - 1) * ncols + col + 1 if row > 0 and col < ncols - 1 else -1
            self.pu_motion_estimation(addr, (ctu_x + rel_x, ctu_y + rel_y, cu_size), part, area_id, final_idx, pu_offset,
                                      False, [left, above, above_right, -1, above_left])

    # ---- Analysis::deriveMVsForCTU ----
    def derive_mvs_for_ctu(self, addr):
        ctu = self.ctu
        ctu_x, ctu_y = (addr % self.cols) * ctu, (addr // self.cols) * ctu
        n = len(self.ref_pocs)
        self.area_best = [[(0, 0)] * n for _ in range(5)]
        area_qp = None
        if self.cu_qp is not None:
            area_qp = self.area_qp[addr] if self.area_qp is not None else self.cu_qp[addr][:5]
            self.set_lambda_from_qp(area_qp[0])
        self.pu_motion_estimation(addr, (ctu_x, ctu_y, ctu), SIZE_2Nx2N, 0, 0, 0, True)
        if ctu != self.min_cu:
            for sub in range(4):
                hs = ctu // 2
                if area_qp is not None:
                    self.set_lambda_from_qp(area_qp[sub + 1])
                self.pu_motion_estimation(addr, (ctu_x + (sub & 1) * hs, ctu_y + (sub >> 1) * hs, hs), SIZE_2Nx2N,
                                          sub + 1, 0, 0, True)
        if self.median_mv is not None:
            for ref in range(n):
                valid, mx, my = (int(v) for v in self.median_mv[addr][ref])
                if valid:
                    for a in range(5):
                        self.area_best[a][ref] = (mx, my)
        self.compute_mv_for_pus(addr, ctu_x, ctu_y, 0, 0, ctu)

    def run_rows(self, row_begin, row_end):
        for addr in range(row_begin * self.cols, row_end * self.cols):
            self.derive_mvs_for_ctu(addr)
        return self.table
