"""Python face of the C++ host layer for the per-request parity tests.

The host side of the lookahead above the C-ABI exists once, in C++ (x265_amd/host/lookahead_host.{h,cpp}:
PreLookaheadGroup, CostEstimateGroup, LookaheadTLD, CuTree, Lookahead).  The classes below carry the same names
and are thin adapters over its plain-C face (xgh_*), so that a test reads like calls into the reference while every
decision - the cost cache, bDoSearch from the sentinel, request order, cooperative slices outside batch mode, the
weight analysis, the B-frame bias and the intra penalty, the cuTree walk - is made by that one implementation:

  PreLookaheadGroup.process_tasks      -> PreLookaheadGroup::processTasks   (slicetype.cpp:1836-1861)
  CostEstimateGroup.single_cost        -> CostEstimateGroup::singleCost     (slicetype.cpp:4148-4152)
  CostEstimateGroup.add / finish_batch -> ::add / ::finishBatch             (slicetype.cpp:4154-4196)
  CuTree.cu_tree                       -> Lookahead::cuTree                 (slicetype.cpp:3665-3766)

A LowresState is the test's view of one frame: what it sets (wp sums, slice type, the AQ offset arrays) goes down
to the window before a call, what the call produced (cost cache, intraMbs, weighted flags) comes back after it.
Frame i of the list uses slot i of the context.
"""
import ctypes as C

import numpy as np

from . import api

# X265_TYPE_* (source/x265.h)
TYPE_AUTO, TYPE_IDR, TYPE_I, TYPE_P, TYPE_BREF, TYPE_B = range(6)


class LowresState:
    """One frame of the window as the tests see it."""

    def __init__(self, slot, poc, D):
        self.slot = slot
        self.poc = poc
        self.D = D
        self.cost_est = {}                  # (d0, d1) -> costEst (after the B-frame bias)
        self.cost_est_aq = {}
        self.intra_mbs = [0] * D
        self.weighted = {}                  # d0 -> weightedRef[d0].isWeighted
        # Lowres::wp_sum[0] / wp_ssd[0] as calcAdaptiveQuantFrame leaves them (slicetype.cpp:600-611)
        self.wp_sum = 0
        self.wp_ssd = 0
        self.slice_type = TYPE_AUTO         # Lowres::sliceType
        self.qp_aq_offset = None            # float64[n], from calcAdaptiveQuantFrame
        self.qp_cutree_offset = None        # float64[n], updated in place by cu_tree()


class _Session:
    """The C++ window behind a list of LowresState objects of one context."""

    def __init__(self, ctx, frames):
        for i, fr in enumerate(frames):
            if fr.slot != i:
                raise ValueError("frame %d of the window must use slot %d" % (i, i))
        self.ctx = ctx
        self.frames = frames
        self.win = api.HostWindow(ctx, len(frames))
        self._offsets = {}

    def close(self):
        self.win.close()

    def down(self):
        """what the test set on its frames -> the window"""
        for i, fr in enumerate(self.frames):
            self.win.set_wp(i, fr.wp_sum, fr.wp_ssd)
            self.win.set_slice_type(i, fr.slice_type)
            if fr.qp_aq_offset is not None and fr.qp_cutree_offset is not None:
                key = (fr.qp_aq_offset.ctypes.data, fr.qp_cutree_offset.ctypes.data)
                if self._offsets.get(i) != key:
                    self.win.set_qp_offsets(i, fr.qp_aq_offset, fr.qp_cutree_offset)
                    self._offsets[i] = key

    def up(self):
        """the window's cost cache and flags -> the frames"""
        H, w = self.win.H, self.win.h
        D = self.frames[0].D
        trip = np.array([(b - d0, b + d1, b) for b in range(len(self.frames)) for d0 in range(D) for d1 in range(D)], np.int32)
        out = np.zeros((len(trip), 3), np.int64)
        H.xgh_collect_costs(w, len(trip), trip.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        k = 0
        for b, fr in enumerate(self.frames):
            fr.cost_est, fr.cost_est_aq = {}, {}
            for d0 in range(D):
                fr.intra_mbs[d0] = self.win.intra_mbs(b, d0)
                fr.weighted[d0] = self.win.is_weighted(b, d0)
                for d1 in range(D):
                    if out[k, 0] >= 0:
                        fr.cost_est[(d0, d1)] = int(out[k, 0])
                        fr.cost_est_aq[(d0, d1)] = int(out[k, 1])
                    k += 1


def _session(ctx, frames, create=False):
    s = getattr(ctx, "_lookahead_session", None)
    if s is not None and (create or s.frames is not frames):
        s.close()
        s = None
    if s is None:
        s = _Session(ctx, frames)
        ctx._lookahead_session = s
        close = ctx.close

        def close_with_session():
            cur = getattr(ctx, "_lookahead_session", None)
            if cur is not None:
                cur.close()
                ctx._lookahead_session = None
            close()
        if not getattr(ctx, "_lookahead_close_hooked", False):
            ctx.close = close_with_session
            ctx._lookahead_close_hooked = True
    return s


class PreLookaheadGroup:
    def __init__(self, ctx):
        self.ctx = ctx

    def process_tasks(self, frames, lumas, inv_qscales=None):
        """Lowres::init + lowresIntraEstimate for every frame of the window (with the caller's AQ scale factors)."""
        s = _session(self.ctx, frames, create=True)
        s.win.pre_lookahead(list(lumas), None if inv_qscales is None else list(inv_qscales))
        s.up()


class CostEstimateGroup:
    def __init__(self, ctx, frames, weightp=False):
        self.ctx = ctx
        self.frames = frames
        self.s = _session(ctx, frames)
        self.s.win.new_group(weightp)

    def single_cost(self, p0, p1, b, intra_penalty=False):
        self.s.down()
        v = self.s.win.single_cost(p0, p1, b, intra_penalty)
        self.s.up()
        return v

    def add(self, p0, p1, b):
        self.s.down()
        self.s.win.add(p0, p1, b)

    def finish_batch(self):
        self.s.win.finish_batch()
        self.s.up()


class CuTree:
    """Lookahead::cuTree over the window.  Parameters follow x265_param: fps_num / fps_denom, b_pyramid
    (bBPyramid), weighted_bipred (bEnableWeightedBiPred), lookahead_depth, vbv_buffer_size (rc.vbvBufferSize),
    q_compress (rc.qCompress)."""

    def __init__(self, ctx, fps_num=30, fps_denom=1, b_pyramid=False, weighted_bipred=False, lookahead_depth=20,
                 vbv_buffer_size=0, q_compress=0.6):
        self.ctx = ctx
        self.kw = dict(fps_num=fps_num, fps_denom=fps_denom, b_pyramid=b_pyramid, weighted_bipred=weighted_bipred,
                       lookahead_depth=lookahead_depth, vbv_buffer_size=vbv_buffer_size, q_compress=q_compress)

    def cu_tree(self, frames, numframes, b_intra):
        s = _session(self.ctx, frames)
        s.win.new_group(False)
        s.down()
        s.win.cu_tree(numframes, b_intra, **self.kw)
        s.up()
