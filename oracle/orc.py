"""ctypes binding of the CPU oracle (oracle/_build/liborc{8,10,12}.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (x265_amd) never imports it.
Parity of the oracle itself is unpinned (see oracle/orc.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

BC_MAX_MV = 1 << 15
BFRAME_MAX = 16
DIA, HEX, UMH, STAR, SEA, FULL = range(6)


class MV(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32)]


class Param(C.Structure):
    _fields_ = [
        ("sourceWidth", C.c_int), ("sourceHeight", C.c_int), ("maxCUSize", C.c_int), ("bframes", C.c_int),
        ("bEnableHME", C.c_int), ("hmeSearchMethod", C.c_int * 3), ("hmeRange", C.c_int * 3),
        ("qgSize", C.c_int), ("bFrameBias", C.c_int), ("bEnableWeightedPred", C.c_int), ("maxSlices", C.c_int),
        ("lookaheadSearchMethod", C.c_int), ("bEnableTemporalFilter", C.c_int),
        ("lookaheadSlices", C.c_int),
    ]


class CuTreeParam(C.Structure):
    _fields_ = [("fpsNum", C.c_int), ("fpsDenom", C.c_int), ("bBPyramid", C.c_int), ("bEnableWeightedBiPred", C.c_int),
                ("lookaheadDepth", C.c_int), ("vbvBufferSize", C.c_int), ("qCompress", C.c_double)]


class LaParam(C.Structure):
    """orc_la_param (defaults: preset medium as Encoder::configure leaves it)."""
    _fields_ = [("lookaheadDepth", C.c_int), ("bFrameAdaptive", C.c_int), ("bBPyramid", C.c_int),
                ("scenecutThreshold", C.c_int), ("scenecutBias", C.c_double), ("keyframeMin", C.c_int),
                ("keyframeMax", C.c_int), ("bOpenGOP", C.c_int), ("bIntraRefresh", C.c_int), ("radl", C.c_int),
                ("gopLookahead", C.c_int), ("cuTree", C.c_int), ("aqMode", C.c_int), ("vbvBufferSize", C.c_int),
                ("vbvMaxBitrate", C.c_int), ("bCqp", C.c_int), ("bEnableTemporalSubLayers", C.c_int),
                ("maxNumReferences", C.c_int), ("poolWorkers", C.c_int), ("bAQMotion", C.c_int),
                ("bHistBasedSceneCut", C.c_int), ("bEnableFades", C.c_int)]


class HistStats(C.Structure):
    _fields_ = [("picHistogram", C.c_uint32 * 256 * 3 * 4 * 4), ("averageIntensityPerSegment", C.c_uint64 * 3 * 4 * 4),
                ("averageIntensity", C.c_uint8 * 3), ("pad", C.c_uint8 * 5), ("picAvgVariance", C.c_uint16),
                ("picAvgVarianceCb", C.c_uint16), ("picAvgVarianceCr", C.c_uint16), ("pad2", C.c_uint16)]


class McstfPlanes(C.Structure):
    _fields_ = [("lowerRes", C.c_void_p), ("lowres", C.c_void_p), ("lumaStride", C.c_ssize_t), ("width", C.c_int),
                ("lines", C.c_int), ("pic", C.c_void_p), ("picStride", C.c_ssize_t), ("picWidth", C.c_int),
                ("picHeight", C.c_int)]


class McstfMvs(C.Structure):
    _fields_ = [("mvs0", C.c_void_p), ("mvs1", C.c_void_p), ("mvs2", C.c_void_p), ("mvs", C.c_void_p),
                ("mvsStride0", C.c_uint32), ("mvsStride1", C.c_uint32), ("mvsStride2", C.c_uint32),
                ("mvsStride", C.c_uint32), ("error", C.c_void_p)]


class RefPlanes(C.Structure):
    _fields_ = [("plane", C.c_void_p * 4), ("lowerPlane", C.c_void_p * 4), ("stride", C.c_ssize_t),
                ("isLowres", C.c_int), ("isHMELowres", C.c_int)]


def build(force=False):
    """Compile the oracle for all depths (gcc, a few seconds)."""
    if force or not all(os.path.exists(os.path.join(_BUILD, "liborc%d.so" % d)) for d in (8, 10, 12)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-j4"])


_libs = {}


def lib(depth=8):
    if depth in _libs:
        return _libs[depth]
    path = os.path.join(_BUILD, "liborc%d.so" % depth)
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    vp, i, ip = C.c_void_p, C.c_int, C.c_ssize_t
    L.orc_depth.restype = i
    L.orc_sad.argtypes = [i, i, vp, ip, vp, ip]
    L.orc_sad_x3.argtypes = [i, i, vp, vp, vp, vp, ip, vp]
    L.orc_sad_x4.argtypes = [i, i, vp, vp, vp, vp, vp, ip, vp]
    L.orc_satd.argtypes = [i, i, vp, ip, vp, ip]
    L.orc_sa8d.argtypes = [i, vp, ip, vp, ip]
    L.orc_pixelavg_pp.argtypes = [i, i, vp, ip, vp, ip, vp, ip]
    L.orc_frame_init_lowres_core.argtypes = [vp, vp, vp, vp, vp, ip, ip, i, i]
    L.orc_frame_init_lowres_core_mcstf.argtypes = [vp, vp, vp, vp, vp, ip, ip, i, i]
    L.orc_extend_pic_border.argtypes = [vp, ip, i, i, i, i]
    L.orc_weight_pp.argtypes = [vp, vp, ip, i, i, i, i, i, i]
    L.orc_pixel_var.argtypes = [i, vp, ip]
    L.orc_pixel_var.restype = C.c_uint64
    L.orc_intra_filter.argtypes = [i, vp, vp]
    L.orc_intra_pred.argtypes = [i, i, vp, ip, vp, i]
    L.orc_intra_filter_flag.argtypes = [i, i]
    for f in (L.orc_luma_hpp, L.orc_luma_vpp):
        f.argtypes = [i, i, vp, ip, vp, ip, i]
    L.orc_luma_hvpp.argtypes = [i, i, vp, ip, vp, ip, i, i]
    L.orc_bitcost_table.argtypes = [C.c_double, vp]
    L.orc_lookahead_lambda.restype = C.c_double
    L.orc_lowres_create.argtypes = [C.POINTER(Param), i]
    L.orc_lowres_create.restype = vp
    L.orc_lowres_destroy.argtypes = [vp]
    L.orc_lowres_init.argtypes = [vp, C.POINTER(Param), vp, ip, i]
    L.orc_picyuv_pad.argtypes = [C.POINTER(Param), vp, C.POINTER(vp), C.POINTER(ip)]
    L.orc_picyuv_pad.restype = vp
    L.orc_tld_create.argtypes = [C.POINTER(Param), vp]
    L.orc_tld_create.restype = vp
    L.orc_tld_destroy.argtypes = [vp]
    L.orc_intra_estimate.argtypes = [vp, vp, C.POINTER(Param)]
    L.orc_estimate_frame_cost.argtypes = [vp, C.POINTER(Param), C.POINTER(vp), i, i, i, i]
    L.orc_estimate_frame_cost.restype = C.c_int64
    L.orc_estimate_frame_cost_single.argtypes = [vp, C.POINTER(Param), C.POINTER(vp), i, i, i, i]
    L.orc_estimate_frame_cost_single.restype = C.c_int64
    L.orc_coop_slices.argtypes = [C.POINTER(Param), i, C.POINTER(i)]
    L.orc_weight_cost_luma.argtypes = [vp, vp, vp, i, i, i, i]
    L.orc_weight_cost_luma.restype = C.c_uint32
    L.orc_weights_analyse.argtypes = [vp, vp, vp]
    L.orc_lowres_ptr.argtypes = [vp, C.c_char_p, i, i]
    L.orc_lowres_ptr.restype = vp
    L.orc_lowres_i64.argtypes = [vp, C.c_char_p, i, i]
    L.orc_lowres_i64.restype = C.c_int64
    L.orc_aq_energy.argtypes = [vp, ip, vp, vp, ip, i, i, vp, vp]
    L.orc_exp2fix8.argtypes = [C.c_double]
    L.orc_hevc_aq_layer.argtypes = [vp, ip, i, i, i, i, i, i, C.c_double, vp, vp, vp]
    L.orc_hevc_aq_layer.restype = C.c_double
    L.orc_exp2fix8.argtypes = [C.c_double]
    L.orc_calc_adaptive_quant_frame.argtypes = [vp, ip, vp, vp, ip, i, i, i, C.c_double, i, i, i, vp, vp, vp, vp, vp,
                                                i, i, i, vp, i, vp, vp]
    L.orc_cutree.argtypes = [vp, C.POINTER(Param), C.POINTER(CuTreeParam), C.POINTER(vp), i, i]
    L.orc_cutree_hevc_layer.argtypes = [i, i, i, i, i, i, i, i, vp, vp, vp, i, C.c_double, C.c_double, vp, vp, vp]
    L.orc_cutree_ratio8x8.argtypes = [i, i, vp, vp, vp, i, C.c_double, vp]
    L.orc_cutree_finish_qg8.argtypes = [i, i, vp, vp, vp, i, C.c_double, C.c_double, vp, vp]
    L.orc_lowres_set_slice_type.argtypes = [vp, i]
    L.orc_propagate_cost.argtypes = [vp, vp, vp, vp, vp, vp, i]
    L.orc_collect_picture_statistics.argtypes = [vp, ip, vp, ip, vp, vp, ip, i, i, C.POINTER(HistStats)]
    L.orc_me_init.argtypes = [vp, vp]
    L.orc_me_create.argtypes = [vp]
    L.orc_me_create.restype = vp
    L.orc_me_destroy.argtypes = [vp]
    L.orc_me_set_source.argtypes = [vp, vp, ip, ip, i, i, i, i, i, i]
    L.orc_motion_estimate.argtypes = [vp, C.POINTER(RefPlanes), MV, MV, MV, i, vp, i, C.POINTER(MV), C.c_uint32, i]
    L.orc_motion_estimate.restype = i
    L.orc_diamond_search.argtypes = [vp, C.POINTER(RefPlanes), MV, MV, MV, C.POINTER(MV)]
    L.orc_diamond_search.restype = i
    L.orc_me_pred_sad.argtypes = [vp, C.POINTER(RefPlanes), MV]
    L.orc_me_pred_sad.restype = i
    L.orc_me_bipred_satd.argtypes = [vp, C.POINTER(RefPlanes), MV, C.POINTER(RefPlanes), MV]
    L.orc_me_bipred_satd.restype = i
    L.orc_me_set_vert_restriction.argtypes = [vp, i]
    L.orc_sea_integral.argtypes = [vp, ip, i, i, i, vp]
    L.orc_me_set_integral.argtypes = [vp, vp]
    L.orc_sea_search_only.argtypes = [vp, C.POINTER(RefPlanes), MV, MV, MV, i, C.POINTER(MV), i]
    L.orc_sea_search_only.restype = i
    L.orc_bitsizes.argtypes = [vp]
    L.orc_mcstf_motion_error.argtypes = [vp, i, vp, i, i, i, i, i, i]
    L.orc_mcstf_level.argtypes = [i, C.POINTER(McstfPlanes), C.POINTER(McstfPlanes), C.POINTER(McstfMvs)]
    L.orc_la_create.argtypes = [C.POINTER(Param), C.POINTER(LaParam), C.POINTER(CuTreeParam), vp]
    L.orc_la_create.restype = vp
    L.orc_la_destroy.argtypes = [vp]
    L.orc_la_add_picture.argtypes = [vp, vp, i, i]
    L.orc_la_flush.argtypes = [vp]
    L.orc_la_get_decided_picture.argtypes = [vp]
    L.orc_la_get_decided_picture.restype = vp
    L.orc_la_trace.argtypes = [vp, C.POINTER(C.POINTER(C.c_int32))]
    L.orc_frame_cost_recalculate.argtypes = [C.POINTER(vp), i, i, i]
    L.orc_frame_cost_recalculate.restype = C.c_int64
    L.libc_free = C.CDLL(None).free
    L.libc_free.argtypes = [vp]
    _libs[depth] = L
    return L


def pixel_dtype(depth):
    return np.uint8 if depth == 8 else np.uint16


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def make_param(width, height, bframes=4, hme=False, weightp=False, ctu=64, search=HEX,
               hme_methods=(HEX, UMH, UMH), hme_range=(16, 32, 48), bframe_bias=0, mcstf=False,
               lookahead_slices=0, max_slices=1, qg_size=32):
    p = Param()
    p.sourceWidth, p.sourceHeight, p.maxCUSize, p.bframes = width, height, ctu, bframes
    p.bEnableHME = int(hme)
    for k in range(3):
        p.hmeSearchMethod[k] = hme_methods[k]
        p.hmeRange[k] = hme_range[k]
    p.qgSize = qg_size
    p.bFrameBias = bframe_bias
    p.bEnableWeightedPred = int(weightp)
    p.maxSlices = max_slices
    p.lookaheadSlices = lookahead_slices
    p.lookaheadSearchMethod = search
    p.bEnableTemporalFilter = int(mcstf)
    return p


def bitcost_table(depth, lam=None):
    """(table, centre index): uint16[4*BC_MAX_MV+1] for the lookahead lambda."""
    L = lib(depth)
    t = np.zeros(4 * BC_MAX_MV + 1, np.uint16)
    L.orc_bitcost_table(L.orc_lookahead_lambda() if lam is None else lam, ptr(t))
    return t, 2 * BC_MAX_MV


def bitsizes(depth=8):
    """(float32 array, centre index): BitCost::s_bitsizes."""
    t = np.zeros(4 * BC_MAX_MV + 1, np.float32)
    lib(depth).orc_bitsizes(ptr(t))
    return t, 2 * BC_MAX_MV


class Lowres:
    """One reference-side Lowres (oracle) with numpy views on its arrays."""

    def __init__(self, L, depth, param, with_aq=False):
        self.L, self.depth, self.param = L, depth, param
        self.h = L.orc_lowres_create(C.byref(param), int(with_aq))
        self.with_aq = with_aq

    def close(self):
        if self.h:
            self.L.orc_lowres_destroy(self.h)
            self.h = None

    def i64(self, field, i=0, j=0):
        return self.L.orc_lowres_i64(self.h, field.encode(), i, j)

    def _view(self, field, i, j, count, dtype):
        p = self.L.orc_lowres_ptr(self.h, field.encode(), i, j)
        if not p:
            return None
        buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=dtype, count=count)

    @property
    def geometry(self):
        return {k: self.i64(k) for k in ("width", "lines", "stride", "marginX", "marginY", "widthInCU",
                                          "heightInCU", "planesize", "padoffset", "widthInCU4", "heightInCU4")}

    def init(self, tight_luma, poc):
        """tight_luma: (H, W) array; padded the way PicYuv::copyFromPicture does."""
        org, st = C.c_void_p(), C.c_ssize_t()
        tight = np.ascontiguousarray(tight_luma, dtype=pixel_dtype(self.depth))
        buf = self.L.orc_picyuv_pad(C.byref(self.param), ptr(tight), C.byref(org), C.byref(st))
        self.L.orc_lowres_init(self.h, C.byref(self.param), org, st, poc)
        self.L.libc_free(buf)

    def buffer(self, i):
        return self._view("buffer", i, 0, self.i64("planesize"), pixel_dtype(self.depth))

    def lower_buffer(self, i):
        return self._view("lowerBuffer", i, 0, self.i64("planesize") // 2, pixel_dtype(self.depth))

    def ncu(self):
        return self.i64("widthInCU") * self.i64("heightInCU")

    def intra_cost(self):
        return self._view("intraCost", 0, 0, self.ncu(), np.int32)

    def intra_mode(self):
        return self._view("intraMode", 0, 0, self.ncu(), np.uint8)

    def inv_qscale(self):
        return self._view("invQscaleFactor", 0, 0, self.ncu(), np.int32)

    def mvs(self, lst, dist):
        return self._view("lowresMvs", lst, dist, 2 * self.ncu(), np.int32).reshape(-1, 2)

    def mv_costs(self, lst, dist):
        return self._view("lowresMvCosts", lst, dist, self.ncu(), np.int32)

    def ncu4(self):
        return self.i64("widthInCU4") * self.i64("heightInCU4")

    def lower_mvs(self, lst, dist):
        return self._view("lowerResMvs", lst, dist, 2 * self.ncu4(), np.int32).reshape(-1, 2)

    def lower_mv_costs(self, lst, dist):
        return self._view("lowerResMvCosts", lst, dist, self.ncu4(), np.int32)

    def lowres_costs(self, d0, d1):
        return self._view("lowresCosts", d0, d1, self.ncu(), np.uint16)

    def row_satds(self, d0, d1):
        return self._view("rowSatds", d0, d1, self.i64("heightInCU"), np.int32)

    def wp(self):
        return self._view("wp_sum", 0, 0, 3, np.uint64), self._view("wp_ssd", 0, 0, 3, np.uint64)

    def propagate_cost(self):
        return self._view("propagateCost", 0, 0, self.ncu(), np.uint16)

    def qp_aq_offset(self):
        return self._view("qpAqOffset", 0, 0, self.ncu() * (4 if self.i64("qg8") else 1), np.float64)

    def qp_cutree_offset(self):
        return self._view("qpCuTreeOffset", 0, 0, self.ncu() * (4 if self.i64("qg8") else 1), np.float64)

    def set_slice_type(self, t):
        self.L.orc_lowres_set_slice_type(self.h, int(t))


class Lookahead:
    """Drives the oracle's estimateFrameCost over a list of frames."""

    def __init__(self, depth, param, with_aq=False):
        self.depth, self.param = depth, param
        self.L = lib(depth)
        self.table, self.centre = bitcost_table(depth)
        self.tld = self.L.orc_tld_create(C.byref(param), C.c_void_p(self.table.ctypes.data + 2 * self.centre))
        self.frames = []
        self.with_aq = with_aq

    def add_frame(self, tight_luma, inv_qscale=None):
        lr = Lowres(self.L, self.depth, self.param, self.with_aq)
        lr.init(tight_luma, len(self.frames))
        if inv_qscale is not None:
            lr.inv_qscale()[:] = inv_qscale
        self.L.orc_intra_estimate(self.tld, lr.h, C.byref(self.param))
        self.frames.append(lr)
        return lr

    def frame_cost(self, p0, p1, b, intra_penalty=False, single=False):
        """estimateFrameCost; single: outside batch mode (cooperative slices apply)."""
        arr = (C.c_void_p * len(self.frames))(*[f.h for f in self.frames])
        fn = self.L.orc_estimate_frame_cost_single if single else self.L.orc_estimate_frame_cost
        return fn(self.tld, C.byref(self.param), arr, p0, p1, b, int(intra_penalty))

    def cu_tree(self, cp, numframes, b_intra, first=0):
        """Lookahead::cuTree over frames[first .. first + numframes]."""
        sub = self.frames[first:first + numframes + 1]
        arr = (C.c_void_p * len(sub))(*[f.h for f in sub])
        self.L.orc_cutree(self.tld, C.byref(self.param), C.byref(cp), arr, numframes, int(b_intra))

    def close(self):
        for f in self.frames:
            f.close()
        self.frames = []
        if self.tld:
            self.L.orc_tld_destroy(self.tld)
            self.tld = None


class SliceDecider:
    """The oracle's Lookahead::slicetypeDecide driver over an orc.Lookahead's frames (oracle/orc_slicetype.c).

    p: dict with the fields of api.LookaheadParam (the x265_param names); frames are those of `look`
    (already pre-analysed: Lowres::init, scale factors, intra estimate), added in order with add_picture()."""

    def __init__(self, look, p):
        self.look = look
        L = self.L = look.L
        lp = LaParam(p["lookaheadDepth"], p["bFrameAdaptive"], p["bBPyramid"], p["scenecutThreshold"], p["scenecutBias"],
                     p["keyframeMin"], p["keyframeMax"], p["bOpenGOP"], p["bIntraRefresh"], p["radl"], p["gopLookahead"],
                     p["cuTree"], p["aqMode"] if p["bAdaptiveQuant"] else 0, p["vbvBufferSize"], p["vbvMaxBitrate"],
                     p["bCqp"], p["bEnableTemporalSubLayers"], p["maxNumReferences"], p["poolWorkers"],
                     p.get("bAQMotion", 0), p.get("bHistBasedSceneCut", 0), p.get("bEnableFades", 0))
        cp = CuTreeParam(p["fpsNum"], p["fpsDenom"], p["bBPyramid"], p["bEnableWeightedBiPred"], p["lookaheadDepth"],
                         p["vbvBufferSize"], p["qCompress"])
        assert look.param.bframes == p["bframes"] and look.param.bEnableWeightedPred == p["bEnableWeightedPred"]
        assert look.param.bFrameBias == p["bFrameBias"] and look.param.lookaheadSlices == p.get("lookaheadSlices", 0)
        self.h = L.orc_la_create(C.byref(look.param), C.byref(lp), C.byref(cp), look.tld)
        self.by_handle = {}

    def close(self):
        if self.h:
            self.L.orc_la_destroy(self.h)
            self.h = None

    def add_picture(self, index, slice_type=0, slice_type_req=0):
        lr = self.look.frames[index]
        self.by_handle[lr.h] = index
        self.L.orc_la_add_picture(self.h, lr.h, slice_type, slice_type_req)

    def flush(self):
        self.L.orc_la_flush(self.h)

    def get_decided(self):
        """-> frame index or None"""
        h = self.L.orc_la_get_decided_picture(self.h)
        return None if not h else self.by_handle[h]

    def trace(self):
        p = C.POINTER(C.c_int32)()
        n = self.L.orc_la_trace(self.h, C.byref(p))
        if not n:
            return np.zeros((0, 6), np.int32)
        return np.ctypeslib.as_array(p, shape=(n, 6)).copy()


class FullResME:
    """MotionEstimate on full-resolution pictures: setSourcePU + motionEstimate per PU (oracle side).

    fenc / ref are padded planes (2-D arrays) with the picture origin at (originY, originX)."""

    def __init__(self, depth, table, centre, fenc, ref, origin_x, origin_y):
        self.L = lib(depth)
        self.table = table
        self.me = self.L.orc_me_create(C.c_void_p(table.ctypes.data + 2 * centre))
        self.fenc, self.ref = np.ascontiguousarray(fenc), np.ascontiguousarray(ref)
        assert self.fenc.shape == self.ref.shape and self.fenc.dtype == pixel_dtype(depth)
        self.stride = self.fenc.shape[1]
        self.org = origin_y * self.stride + origin_x
        item = self.fenc.itemsize
        self.rp = RefPlanes()
        self.rp.plane[0] = self.ref.ctypes.data + self.org * item
        self.rp.stride = self.stride
        self.rp.isLowres = 0
        self.rp.isHMELowres = 0
        self.fenc_org = self.fenc.ctypes.data + self.org * item

    def close(self):
        if self.me:
            self.L.orc_me_destroy(self.me)
            self.me = None

    def build_integral(self):
        """The reference picture's twelve block-sum planes (FrameFilter::computeMEIntegral), needed by method 4 (SEA).
        The padded plane is taken to have the same margin on both sides."""
        rows, stride = self.ref.shape
        pad_x, pad_y = self.org % stride, self.org // stride
        self.integral = [np.zeros((rows + 1) * stride, np.uint32) for _ in range(12)]
        arr = (C.c_void_p * 12)(*[a.ctypes.data + 4 * self.org for a in self.integral])
        self.L.orc_sea_integral(C.c_void_p(self.rp.plane[0]), stride, pad_x, pad_y, rows - 2 * pad_y, arr)
        self.L.orc_me_set_integral(self.me, arr)

    def integral_plane(self, k):
        """plane k as a 2-D array shaped like the padded reference"""
        rows, stride = self.ref.shape
        return self.integral[k][:rows * stride].reshape(rows, stride)

    def sea_search_only(self, x, y, w, h, mvp, mvmin, mvmax, merange, start_mv, start_cost):
        """the SEA stage of motionEstimate alone, from (start_mv, start_cost) -> ((full-pel mvx, mvy), cost)"""
        L = self.L
        L.orc_me_set_source(self.me, C.c_void_p(self.fenc_org), self.stride, y * self.stride + x, w, h, 4, 4, 4, 3)
        bmv = MV(*start_mv)
        cost = L.orc_sea_search_only(self.me, C.byref(self.rp), MV(*mvmin), MV(*mvmax), MV(*mvp), merange, C.byref(bmv),
                                     start_cost)
        return (bmv.x, bmv.y), cost

    def bipred_satd(self, x, y, w, h, qmv0, other, qmv1):
        """SATD of the PU against the average of its prediction from this reference at qmv0 and from `other`'s
        reference at qmv1 (the bidirectional candidate of Search::puMotionEstimation)"""
        L = self.L
        L.orc_me_set_source(self.me, C.c_void_p(self.fenc_org), self.stride, y * self.stride + x, w, h, 3, 3, 3, 3)
        return L.orc_me_bipred_satd(self.me, C.byref(self.rp), MV(*qmv0), C.byref(other.rp), MV(*qmv1))

    def pred_sad(self, x, y, w, h, qmv):
        """SAD of the PU against the prediction at quarter-pel vector qmv (one candidate of Search::selectMVP)"""
        L = self.L
        L.orc_me_set_source(self.me, C.c_void_p(self.fenc_org), self.stride, y * self.stride + x, w, h, 3, 3, 3, 3)
        return L.orc_me_pred_sad(self.me, C.byref(self.rp), MV(*qmv))

    def estimate(self, x, y, w, h, mvp, mvmin, mvmax, method, merange, subpel, mvc=(), max_slices=1,
                 vert_restriction=False):
        """-> ((mvx, mvy), cost)"""
        L = self.L
        L.orc_me_set_vert_restriction(self.me, int(vert_restriction))
        L.orc_me_set_source(self.me, C.c_void_p(self.fenc_org), self.stride, y * self.stride + x, w, h,
                            method, method, method, subpel)
        arr = (MV * max(1, len(mvc)))(*[MV(int(a), int(b)) for a, b in mvc])
        out = MV()
        cost = L.orc_motion_estimate(self.me, C.byref(self.rp), MV(*mvmin), MV(*mvmax), MV(*mvp), len(mvc), arr,
                                     merange, C.byref(out), max_slices, 0)
        return (out.x, out.y), cost


def _frme_diamond(self, x, y, w, h, mvp, mvmin, mvmax):
    """MotionEstimate::diamondSearch -> ((full-pel mvx, mvy), cost)"""
    L = self.L
    L.orc_me_set_source(self.me, C.c_void_p(self.fenc_org), self.stride, y * self.stride + x, w, h, 3, 3, 3, 3)
    out = MV()
    cost = L.orc_diamond_search(self.me, C.byref(self.rp), MV(*mvmin), MV(*mvmax), MV(*mvp), C.byref(out))
    return (out.x, out.y), cost


FullResME.diamond_search = _frme_diamond


def calc_adaptive_quant_frame(depth, y, u, v, aq_mode, strength, weightp=False, hdr10=False, skip_offsets=False,
                              quant_offsets=None, qg_size=32, block_variance=False):
    """calcAdaptiveQuantFrame on tight 4:2:0 planes -> (qpAqOffset, qpCuTreeOffset, invQscaleFactor, wp_sum, wp_ssd)
    [+ (invQscaleFactor8x8,) with qg_size 8] [+ (blockVariance, frameVariance) with block_variance]."""
    L = lib(depth)
    h, w = y.shape
    w8, h8 = ((w // 2) + 7) >> 3, ((h // 2) + 7) >> 3
    qg8 = qg_size == 8
    if qg8:
        n = 4 * w8 * h8                     # allocated zeroed, as Lowres::create does
        qa, qc, inv = np.zeros(n), np.zeros(n), np.zeros(n, np.int32)
    else:
        n = w8 * h8
        qa = np.full(n, np.nan)
        qc = np.full(n, np.nan)
        inv = np.full(n, -1, np.int32)
    inv8 = np.full(w8 * h8, -1, np.int32)
    var = np.zeros(n, np.uint32)
    fvar = np.full(1, np.nan)
    wp = np.zeros(6, np.uint64)
    qo = None if quant_offsets is None else np.ascontiguousarray(quant_offsets, np.float32)
    L.orc_calc_adaptive_quant_frame(ptr(y), y.strides[0] // y.itemsize, ptr(u), ptr(v), u.strides[0] // u.itemsize,
                                    w, h, aq_mode, float(strength), int(weightp), int(hdr10), int(skip_offsets),
                                    None if qo is None else ptr(qo), ptr(qa), ptr(qc), ptr(inv), ptr(wp),
                                    int(qg_size), w8, h8, ptr(inv8) if qg8 else None, int(block_variance), ptr(var),
                                    ptr(fvar))
    out = (qa, qc, inv, wp[:3].copy(), wp[3:].copy())
    if qg8:
        out += (inv8,)
    if block_variance:
        out += (var, float(fvar[0]))
    return out


def set_frame_variance(lowres, v):
    """Lowres::frameVariance, read by the fade detection of slicetypeDecide (--fades)."""
    lowres.L.orc_lowres_set_frame_variance.argtypes = [C.c_void_p, C.c_double]
    lowres.L.orc_lowres_set_frame_variance(C.c_void_p(lowres.h), float(v))


def is_fade_end(lowres):
    return bool(lowres.L.orc_lowres_is_fade_end(C.c_void_p(lowres.h)))


def set_hevc_layer(lowres, k, part, num_w, num_h, qp_offset, cutree_offset):
    """rc.hevcAq: hands a frame layer k (increasing depth, the one of minAQDepth last) as xPreanalyze left it."""
    L = lowres.L
    L.orc_lowres_set_hevc_layer.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    qp = np.ascontiguousarray(qp_offset, np.float64)
    tree = np.ascontiguousarray(cutree_offset, np.float64)
    assert qp.size == num_w * num_h == tree.size
    L.orc_lowres_set_hevc_layer(C.c_void_p(lowres.h), k, part, part, num_w, num_h, ptr(qp), ptr(tree))


def hevc_layer(lowres, k, which):
    """-> layer k's dQpOffset (which 0) or dCuTreeOffset (1) as the oracle holds it now"""
    L = lowres.L
    L.orc_lowres_hevc_layer.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.orc_lowres_hevc_layer.restype = C.POINTER(C.c_double)
    n = C.c_int()
    p = L.orc_lowres_hevc_layer(C.c_void_p(lowres.h), k, which, C.byref(n))
    return np.ctypeslib.as_array(p, shape=(n.value,)).copy()


def set_hist_stats(lowres, st):
    """Hands a frame its collectPictureStatistics results for the --hist-scenecut decision (the oracle keeps a copy)."""
    lowres.L.orc_lowres_set_hist_stats(C.c_void_p(lowres.h), C.byref(st))


def collect_picture_statistics(lowres, y, u, v):
    """collectPictureStatistics for the frame behind `lowres` (an initialised Lowres) -> HistStats."""
    L = lowres.L
    h, w = y.shape
    item = y.itemsize
    plane0 = lowres.L.orc_lowres_ptr(lowres.h, b"buffer", 0, 0) + lowres.i64("padoffset") * item
    st = HistStats()
    L.orc_collect_picture_statistics(C.c_void_p(plane0), lowres.i64("stride"), ptr(y), y.strides[0] // item, ptr(u), ptr(v),
                                     u.strides[0] // item, w, h, C.byref(st))
    return st


MCSTF_PAD = 64


class McstfPicture:
    """What the --mcstf motion search reads of one frame: the Lowres planes (made with the param's mcstf
    filter) and the border-extended source picture."""

    def __init__(self, lowres, tight_luma):
        self.lowres = lowres
        self.pic = np.ascontiguousarray(np.pad(tight_luma, MCSTF_PAD, mode="edge"))
        item = self.pic.itemsize
        h, w = tight_luma.shape
        p = McstfPlanes()
        p.lowerRes = lowres.L.orc_lowres_ptr(lowres.h, b"lowerPlane", 0, 0)
        p.lowres = lowres.L.orc_lowres_ptr(lowres.h, b"plane", 0, 0)
        p.lumaStride = lowres.i64("stride")
        p.width, p.lines = lowres.i64("width"), lowres.i64("lines")
        p.picStride = self.pic.shape[1]
        p.pic = self.pic.ctypes.data + (MCSTF_PAD * self.pic.shape[1] + MCSTF_PAD) * item
        p.picWidth, p.picHeight = w, h
        self.planes = p


def mcstf_motion_search(cur, refs):
    """All four levels for each reference -> (mvs[r, H/4, W/4, 2], error[r, H/4, W/4], coarse[r, 3, H/16, W/16, 2])."""
    L = cur.lowres.L
    W, H = cur.planes.picWidth, cur.planes.picHeight
    n = len(refs)
    mvs = np.zeros((n, H // 4, W // 4, 2), np.int32)
    err = np.zeros((n, H // 4, W // 4), np.int32)
    coarse = np.zeros((n, 3, H // 16, W // 16, 2), np.int32)
    for r, ref in enumerate(refs):
        m = McstfMvs()
        m.mvs0, m.mvs1, m.mvs2 = (coarse[r, k].ctypes.data for k in range(3))
        m.mvs, m.error = mvs[r].ctypes.data, err[r].ctypes.data
        m.mvsStride0 = m.mvsStride1 = m.mvsStride2 = W // 16
        m.mvsStride = W // 4
        for level in range(4):
            L.orc_mcstf_level(level, C.byref(cur.planes), C.byref(ref.planes), C.byref(m))
    return mvs, err, coarse


OP_PHASES = ("intra", "search", "cost", "weights", "other")


def op_counters(depth, reset=False):
    """Executed block compares of this depth's library since load (or the last reset), per phase:
    {phase: (SADs, SATDs, intra predictions)} in 8x8 units (orc_op_counters, orc_pixel.c)."""
    arr = (C.c_longlong * 15).in_dll(lib(depth), "orc_op_counters")
    out = {name: (int(arr[3 * i]), int(arr[3 * i + 1]), int(arr[3 * i + 2])) for i, name in enumerate(OP_PHASES)}
    if reset:
        for i in range(15):
            arr[i] = 0
    return out
