/* CPU oracle for the x265 lowres-lookahead / motion-search hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * algorithm, used by tests/, __graft_entry__.smoke() and the cpu_baseline leg of
 * bench.py as the checker.  Nothing in the product path may call into it.
 *
 * PARITY UNPINNED: the reference tree cannot be compiled in this container from
 * its own sources alone (every translation unit includes the cmake-generated
 * x265_config.h), and its own tests hold no golden vectors for this path
 * (source/test/pixelharness.cpp compares C against SIMD on time-seeded random
 * buffers).  The restatement is therefore checked only against independent
 * numpy formulations of the same arithmetic (tests/test_oracle_*.py) and the
 * constants recorded from the reference's headers.
 *
 * Compiled once per bit depth (ORC_DEPTH = 8, 10 or 12), mirroring the way the
 * reference fixes `pixel` at compile time (source/common/common.h:127-141).
 */
#ifndef ORC_H
#define ORC_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef ORC_DEPTH
#define ORC_DEPTH 8
#endif

#if ORC_DEPTH > 8
typedef uint16_t pixel;
#else
typedef uint8_t pixel;
#endif

#define ORC_PIXEL_MAX ((1 << ORC_DEPTH) - 1)
#define ORC_FENC_STRIDE 64           /* source/common/common.h:71 */
#define ORC_CU 8                     /* X265_LOWRES_CU_SIZE, source/common/common.h:225 */
#define ORC_BFRAME_MAX 16            /* X265_BFRAME_MAX, source/x265.h:599 */
#define ORC_LOOKAHEAD_MAX 250        /* X265_LOOKAHEAD_MAX, source/x265.h:103 */
#define ORC_COST_MAX (1 << 28)       /* MotionEstimate::COST_MAX, source/encoder/motion.h:65 */
#define ORC_LOWRES_COST_MASK ((1 << 14) - 1)  /* source/encoder/slicetype.h:42 */
#define ORC_LOWRES_COST_SHIFT 14
#define ORC_BC_MAX_MV (1 << 15)      /* source/encoder/bitcost.h:92 */

/* search methods, numbering of source/x265.h X265_*_SEARCH */
enum { ORC_DIA = 0, ORC_HEX = 1, ORC_UMH = 2, ORC_STAR = 3, ORC_SEA = 4, ORC_FULL = 5 };

typedef struct { int32_t x, y; } orc_mv;   /* source/common/mv.h:37-45: x then y, 8 bytes */

/* ---- primitives (orc_pixel.c, orc_intra.c, orc_ipfilter.c) ---- */
int  orc_depth(void);
int  orc_sad(int w, int h, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb);
void orc_sad_x3(int w, int h, const pixel* fenc, const pixel* r0, const pixel* r1, const pixel* r2,
                intptr_t rstride, int32_t* res);
void orc_sad_x4(int w, int h, const pixel* fenc, const pixel* r0, const pixel* r1, const pixel* r2,
                const pixel* r3, intptr_t rstride, int32_t* res);
int  orc_satd(int w, int h, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb);
/* executed block compares per phase (orc_pixel.c): [phase][SAD, SATD, intra predictions] in 8x8 units; phases:
 * 0 intra estimate, 1 motion search, 2 cost decision, 3 weight analysis, 4 anything else */
extern long long orc_op_counters[5][3];
extern int orc_op_phase;
int  orc_sa8d(int size, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb);
void orc_pixelavg_pp(int w, int h, pixel* dst, intptr_t ds, const pixel* s0, intptr_t ss0,
                     const pixel* s1, intptr_t ss1);
void orc_frame_init_lowres_core_mcstf(const pixel* src0, pixel* dst0, pixel* dsth, pixel* dstv, pixel* dstc,
                                      intptr_t src_stride, intptr_t dst_stride, int width, int height);
void orc_frame_init_lowres_core(const pixel* src0, pixel* dst0, pixel* dsth, pixel* dstv, pixel* dstc,
                                intptr_t src_stride, intptr_t dst_stride, int width, int height);
void orc_extend_pic_border(pixel* pic, intptr_t stride, int width, int height, int marginX, int marginY);
void orc_weight_pp(const pixel* src, pixel* dst, intptr_t stride, int width, int height,
                   int w0, int round, int shift, int offset);
uint64_t orc_pixel_var(int size, const pixel* pix, intptr_t stride);
void orc_aq_energy(const pixel* y, intptr_t sy, const pixel* u, const pixel* v, intptr_t sc, int w, int h,
                   uint32_t* energy, uint64_t* wp);
int  orc_exp2fix8(double x);
/* edgeFilter: edgePic / thetaPic are tight w x h planes */
void orc_edge_filter(const pixel* src, intptr_t stride, int w, int h, pixel* edgePic, pixel* thetaPic);
double orc_hevc_aq_layer(const pixel* src, intptr_t stride, int width, int height, int aqPartWidth, int aqPartHeight,
                         int numW, int numH, double qpAdaptationRange, double* dActivity, double* dQpOffset,
                         double* dCuTreeOffset);
void orc_calc_adaptive_quant_frame(const pixel* y, intptr_t sy, const pixel* u, const pixel* v, intptr_t sc,
                                   int w, int h, int aqMode, double aqStrength, int weightedPred, int hdr10,
                                   int skipOffsets, const float* quantOffsets, double* qpAqOffset,
                                   double* qpCuTreeOffset, int* invQscaleFactor, uint64_t* wp,
                                   int qgSize, int w8, int h8, int* invQscaleFactor8x8,
                                   int blockVarianceOn, uint32_t* blockVariance, double* frameVarianceOut);

void orc_intra_filter(int tuSize, const pixel* samples, pixel* filtered);
void orc_intra_pred(int size, int mode, pixel* dst, intptr_t dstStride, const pixel* srcPix, int bFilter);
int  orc_intra_filter_flag(int mode, int size);

void orc_luma_hpp(int w, int h, const pixel* src, intptr_t ss, pixel* dst, intptr_t ds, int coeffIdx);
void orc_luma_vpp(int w, int h, const pixel* src, intptr_t ss, pixel* dst, intptr_t ds, int coeffIdx);
void orc_luma_hvpp(int w, int h, const pixel* src, intptr_t ss, pixel* dst, intptr_t ds, int idxX, int idxY);

/* ---- MV cost table (orc_bitcost.c) ---- */
/* out has 4*ORC_BC_MAX_MV+1 entries; element [2*ORC_BC_MAX_MV] is MVD 0 */
void   orc_bitcost_table(double lambda, uint16_t* out);
void   orc_bitsizes(float* out);     /* BitCost::s_bitsizes, centred at out[2 * ORC_BC_MAX_MV] */
double orc_lookahead_lambda(void);   /* x265_lambda_tab[X265_LOOKAHEAD_QP] for this depth */

/* ---- reference planes + motion search (orc_motion.c) ---- */
typedef struct orc_refplanes {
    pixel*   plane[4];        /* lowresPlane[0..3] (origin pointers); plane[0] doubles as fpelPlane[0] */
    pixel*   lowerPlane[4];   /* lowerResPlane[0..3] (HME level 0), or NULL */
    intptr_t stride;          /* lumaStride */
    int      isLowres;
    int      isHMELowres;
} orc_refplanes;

typedef struct orc_me {
    pixel     fenc[ORC_FENC_STRIDE * 64];
    int       bw, bh;
    intptr_t  blockOffset;
    const uint16_t* cost;     /* centred MV cost table */
    orc_mv    mvp;
    int       searchMethod, searchMethodL0, searchMethodL1, subpelRefine;
    /* executed-compare counters (for the algorithmic-op denominators) */
    uint64_t  nSad, nSatd;
    int       vertRestriction;   /* motionEstimate's m_vertRestriction argument (orc_me_set_vert_restriction) */
    const uint32_t* integral[12];/* MotionEstimate::integral before the PU offset is added: element (0,0) of the
                                  * reference's twelve block-sum planes (orc_me_set_integral), for ORC_SEA */
} orc_me;

/* FrameFilter::computeMEIntegral for a whole padded picture (source/encoder/framefilter.cpp:731-831): planes[k]
 * points at element (0,0) of a zeroed buffer shaped like the picture's (same stride, padX columns and padY rows
 * before the origin, `height` + padY rows from it).  Plane order as FrameData::m_meIntegral:
 * 32x32 32x24 32x8 24x32 16x16 16x12 16x4 12x16 8x32 8x8 4x16 4x4. */
void orc_sea_integral(const pixel* picOrg, intptr_t stride, int padX, int padY, int height, uint32_t* const planes[12]);
void orc_me_set_integral(orc_me* me, uint32_t* const planes[12]);

void orc_me_init(orc_me* me, const uint16_t* centredCost);
orc_me* orc_me_create(const uint16_t* centredCost);
void orc_me_destroy(orc_me* me);
void orc_me_set_source(orc_me* me, const pixel* fencY, intptr_t stride, intptr_t offset, int w, int h,
                       int method, int searchL0, int searchL1, int refine);
int  orc_me_buf_satd(orc_me* me, const pixel* ref, intptr_t stride);
void orc_me_set_vert_restriction(orc_me* me, int on);
/* SAD of the source PU against the prediction at a quarter-pel vector: predInterLumaPixel + bufSAD as
 * Search::selectMVP measures a candidate (source/encoder/search.cpp:2372-2379) */
int  orc_me_pred_sad(orc_me* me, const orc_refplanes* ref, orc_mv qmv);
int  orc_me_bipred_satd(orc_me* me, const orc_refplanes* ref0, orc_mv qmv0, const orc_refplanes* ref1, orc_mv qmv1);
/* the X265_SEA stage of motionEstimate alone, from the start (*bmv, bcost); returns the cost, *bmv the vector */
int  orc_sea_search_only(orc_me* me, const orc_refplanes* ref, orc_mv mvmin, orc_mv mvmax, orc_mv qmvp, int merange,
                         orc_mv* bmv, int bcost);
/* hmePlane: 0 = search the half-res plane set, 1 = search the quarter-res (HME level-0) plane set */
int  orc_motion_estimate(orc_me* me, const orc_refplanes* ref, orc_mv mvmin, orc_mv mvmax, orc_mv qmvp,
                         int numCandidates, const orc_mv* mvc, int merange, orc_mv* outQMv,
                         uint32_t maxSlices, int hmePlane);
/* MotionEstimate::diamondSearch: full-pel vector out, returns its SAD + MV cost */
int  orc_diamond_search(orc_me* me, const orc_refplanes* ref, orc_mv mvmin, orc_mv mvmax, orc_mv qmvp, orc_mv* outMV);
const pixel* orc_lowres_mc(const orc_refplanes* ref, intptr_t blockOffset, orc_mv qmv, pixel* buf,
                           intptr_t* outstride, int hme);
int  orc_lowres_qpel_cost(orc_me* me, const orc_refplanes* ref, intptr_t blockOffset, orc_mv qmv,
                          int useSatd, int hme);

/* ---- lookahead frame state and cost estimation (orc_lookahead.c) ---- */
typedef struct orc_param {
    int sourceWidth, sourceHeight;
    int maxCUSize;            /* 64 */
    int bframes;
    int bEnableHME;
    int hmeSearchMethod[3];
    int hmeRange[3];
    int qgSize;               /* 32 */
    int bFrameBias;
    int bEnableWeightedPred;
    int maxSlices;
    int lookaheadSearchMethod;/* x265 hard-wires ORC_HEX (source/encoder/slicetype.cpp:4402-4404) */
    int bEnableTemporalFilter;/* --mcstf: mean downscale filter + quarter-res planes (source/common/lowres.cpp:377-400) */
    int lookaheadSlices;      /* as Lookahead::Lookahead reads it (source/encoder/slicetype.cpp:1122); 0/1: whole frames */
} orc_param;

typedef struct orc_lowres {
    orc_refplanes rp;         /* must be first */
    pixel*   buffer[4];
    pixel*   lowerResBuffer[4];
    int      frameNum;
    int      width, lines;    /* rounded up to multiples of 8 */
    int      marginX, marginY;
    int      bframes;
    int      widthInCU, heightInCU, cuCount;
    size_t   planesize, padoffset;
    int64_t  costEst[ORC_BFRAME_MAX + 2][ORC_BFRAME_MAX + 2];
    int64_t  costEstAq[ORC_BFRAME_MAX + 2][ORC_BFRAME_MAX + 2];
    int32_t* rowSatds[ORC_BFRAME_MAX + 2][ORC_BFRAME_MAX + 2];
    int      intraMbs[ORC_BFRAME_MAX + 2];
    int32_t* intraCost;
    uint8_t* intraMode;
    uint16_t* lowresCosts[ORC_BFRAME_MAX + 2][ORC_BFRAME_MAX + 2];
    int32_t* lowresMvCosts[2][ORC_BFRAME_MAX + 2];
    orc_mv*  lowresMvs[2][ORC_BFRAME_MAX + 2];
    int32_t* lowerResMvCosts[2][ORC_BFRAME_MAX + 2];
    orc_mv*  lowerResMvs[2][ORC_BFRAME_MAX + 2];
    int      widthInCU4, heightInCU4;
    int*     invQscaleFactor; /* NULL when AQ/weightp are all off */
    uint64_t wp_ssd[3], wp_sum[3];
    orc_refplanes weightedRef[ORC_BFRAME_MAX + 2];
    int      weightedRefIsWeighted[ORC_BFRAME_MAX + 2];
    double   weightedCostDelta[ORC_BFRAME_MAX + 2];
    /* cuTree state (allocated with AQ): source/common/lowres.cpp:121-140 */
    uint16_t* propagateCost;
    double*  qpAqOffset;
    double*  qpCuTreeOffset;
    double*  qpAqMotionOffset; /* --aq-motion (source/common/lowres.cpp:114-115), allocated on first use */
    struct orc_hist_stats* hist; /* --hist-scenecut: collectPictureStatistics' results (orc_lowres_set_hist_stats) */
    int      bHistScenecutAnalyzed;
    int      qg8;                /* rc.qgSize == 8: invQscaleFactor holds Lowres::invQscaleFactor8x8 (what the estimates,
                                  * the intra estimate and cuTree read then: slicetype.cpp:798, 3791, 4532) and the two
                                  * offset arrays hold (2 widthInCU) x (2 heightInCU) doubles (lowres.cpp:92-95) */
    /* rc.hevcAq: the PicQPAdaptationLayers of the enabled depths (source/common/lowres.cpp:128-155), the one of
     * minAQDepth last; set with orc_lowres_set_hevc_layer from what xPreanalyze (orc_hevc_aq_layer) produced */
    int      numAqLayers;
    struct { int partW, partH, numW, numH; double* dQpOffset; double* dCuTreeOffset; } aqLayer[4];
    double*  cuTreeOffset8x8;    /* qgSize 8: PicQPAdaptationLayer::dCuTreeOffset8x8 of that last layer */
    int      sourceWidth, sourceHeight;
    double   frameVariance;      /* calcAdaptiveQuantFrame's tail (slicetype.cpp:688-703), read by the fade detection */
    int      bIsFadeEnd;
    int      sliceType;       /* X265_TYPE_* (source/x265.h) */
    /* what the slice-type decision keeps per frame (source/common/lowres.h:166-213) */
    int      sliceTypeReq;
    int      bScenecut, bKeyframe, bLastMiniGopBFrame, leadingBframes;
    int      gopOffset, gopId, tempLayer;   /* Frame::m_gopOffset, m_gopId, m_tempLayer (source/common/frame.h:96, 167-168) */
    int      indB;
    int64_t  plannedSatd[ORC_LOOKAHEAD_MAX + 1];
    int      plannedType[ORC_LOOKAHEAD_MAX + 1];
    int32_t* recalcRowSatds;  /* not in the reference: frameCostRecalculate's row sums are kept apart from the
                                 request's rowSatds (which the reference overwrites in place) so that tests can
                                 compare both */
} orc_lowres;

typedef struct orc_tld {
    orc_me   me;
    pixel*   wbuffer[4];
    int      paddedLines;
    int      widthInCU, heightInCU, ncu;
} orc_tld;

orc_lowres* orc_lowres_create(const orc_param* p, int withAq);
void orc_lowres_destroy(orc_lowres* lr);
/* luma: pointer to pixel (0,0) of a source plane whose row/col padding follows
 * PicYuv::copyFromPicture (source/common/picyuv.cpp:291-300,572-583) */
void orc_lowres_init(orc_lowres* lr, const orc_param* p, const pixel* luma, intptr_t stride, int poc);
/* pads a tight w x h plane the way PicYuv does; returns malloc'd buffer, *origin = pixel (0,0) */
pixel* orc_picyuv_pad(const orc_param* p, const pixel* tight, pixel** origin, intptr_t* stride);

orc_tld* orc_tld_create(const orc_param* p, const uint16_t* centredCost);
void orc_tld_destroy(orc_tld* t);
void orc_intra_estimate(orc_tld* t, orc_lowres* fenc, const orc_param* p);
int64_t orc_estimate_frame_cost(orc_tld* t, const orc_param* p, orc_lowres** frames,
                                int p0, int p1, int b, int bIntraPenalty);
int64_t orc_estimate_frame_cost_single(orc_tld* t, const orc_param* p, orc_lowres** frames,
                                       int p0, int p1, int b, int bIntraPenalty);
int orc_coop_slices(const orc_param* p, int heightInCU, int* numRowsPerSlice);
uint32_t orc_weight_cost_luma(orc_tld* t, orc_lowres* fenc, orc_lowres* ref,
                              int wtPresent, int scale, int denom, int offset);
void orc_weights_analyse(orc_tld* t, orc_lowres* fenc, orc_lowres* ref);

/* ---- cuTree (orc_cutree.c) ---- */
#define ORC_TYPE_B 5
typedef struct orc_cutree_param {
    int fpsNum, fpsDenom;
    int bBPyramid;
    int bEnableWeightedBiPred;
    int lookaheadDepth;
    int vbvBufferSize;
    double qCompress;
} orc_cutree_param;
void orc_propagate_cost(int* dst, const uint16_t* propagateIn, const int32_t* intraCosts, const uint16_t* interCosts,
                        const int32_t* invQscales, const double* fpsFactor, int len);
void orc_estimate_cu_propagate(const orc_cutree_param* cp, orc_lowres** frames, int* scratch, double averageDuration,
                               int p0, int p1, int b, int referenced);
void orc_cutree_finish(const orc_cutree_param* cp, orc_lowres* frame, double averageDuration, int ref0Distance);
void orc_cutree_hevc_layer(int qg8, int maxCols, int widthFullRes, int heightFullRes, int aqPartWidth, int aqPartHeight,
                           int numAQPartInWidth, int numAQPartInHeight, const int32_t* intraCost,
                           const int32_t* invQscaleFactor, const uint16_t* propagateCostArr, int fpsFactor,
                           double weightdelta, double strength, const double* pcCuTree8x8, const double* pcQP,
                           double* pcCuTree);
void orc_cutree_ratio8x8(int W, int H, const int32_t* intraCost, const int32_t* invQscale8x8, const uint16_t* propagateCostArr,
                         int fpsFactor, double weightdelta, double* pcCuTree8x8);
void orc_cutree_finish_qg8(int W, int H, const int32_t* intraCost, const int32_t* invQscale8x8,
                           const uint16_t* propagateCost, int fpsFactor, double weightdelta, double strength,
                           const double* qpAqOffset, double* qpCuTreeOffset);
void orc_cutree(orc_tld* t, const orc_param* p, const orc_cutree_param* cp, orc_lowres** frames, int numframes, int bIntra);
void orc_lowres_set_slice_type(orc_lowres* lr, int sliceType);

/* ---- slice-type decision (orc_slicetype.c): Lookahead::slicetypeDecide and what it calls ---- */
enum { ORC_TYPE_AUTO = 0, ORC_TYPE_IDR = 1, ORC_TYPE_I = 2, ORC_TYPE_P = 3, ORC_TYPE_BREF = 4 };
typedef struct orc_la_param {
    int lookaheadDepth;           /* the x265_param fields of the same names */
    int bFrameAdaptive;           /* 0 none, 1 fast, 2 trellis */
    int bBPyramid;
    int scenecutThreshold;
    double scenecutBias;          /* as Encoder::configure leaves it: the command-line percentage / 100 */
    int keyframeMin, keyframeMax; /* keyframeMin after Encoder::configure (never 0) */
    int bOpenGOP;
    int bIntraRefresh;
    int radl;
    int gopLookahead;
    int cuTree;                   /* rc.cuTree */
    int aqMode;                   /* rc.aqMode */
    int vbvBufferSize, vbvMaxBitrate;
    int bCqp;                     /* rc.rateControlMode == X265_RC_CQP */
    int bEnableTemporalSubLayers; /* 0..2 */
    int maxNumReferences;
    int poolWorkers;              /* ThreadPool::m_numWorkers of the lookahead's pool, 0: no pool */
    int bAQMotion;                /* --aq-motion */
    int bHistBasedSceneCut;       /* --hist-scenecut: frames carry orc_hist_stats (orc_lowres_set_hist_stats) */
    int bEnableFades;             /* --fades: frames carry frameVariance (orc_lowres_set_frame_variance) */
} orc_la_param;
typedef struct orc_la orc_la;
orc_la* orc_la_create(const orc_param* p, const orc_la_param* lp, const orc_cutree_param* cp, orc_tld* tld);
void orc_la_destroy(orc_la* la);
void orc_la_add_picture(orc_la* la, orc_lowres* frame, int sliceType, int sliceTypeReq);
void orc_la_flush(orc_la* la);
orc_lowres* orc_la_get_decided_picture(orc_la* la);
/* requests in the order estimateFrameCost computed them (cached ones are not listed): 6 ints each,
 * { frameNum p0, frameNum p1, frameNum b, bDoSearch[0], bDoSearch[1], batch mode } */
int orc_la_trace(const orc_la* la, const int32_t** out);
int64_t orc_frame_cost_recalculate(orc_lowres** frames, int p0, int p1, int b);
typedef void (*orc_request_hook)(void* user, orc_lowres** frames, int p0, int p1, int b, const int* bDoSearch, int batchMode);
void orc_set_request_hook(orc_request_hook hook, void* user);

/* ---- histogram scene-cut statistics (orc_scenecut.c) ---- */
typedef struct orc_hist_stats {
    uint32_t picHistogram[4][4][3][256];              /* [segment x][segment y][plane][bin] */
    uint64_t averageIntensityPerSegment[4][4][3];
    uint8_t  averageIntensity[3];
    uint8_t  pad[5];
    uint16_t picAvgVariance, picAvgVarianceCb, picAvgVarianceCr, pad2;
} orc_hist_stats;
void orc_lowres_set_hist_stats(orc_lowres* lr, const orc_hist_stats* st);   /* keeps a copy */
void orc_lowres_set_frame_variance(orc_lowres* lr, double v);
void orc_lowres_set_hevc_layer(orc_lowres* lr, int k, int partW, int partH, int numW, int numH, const double* dQpOffset,
                               const double* dCuTreeOffset);      /* layers in increasing depth, k = 0.. */
double* orc_lowres_hevc_layer(orc_lowres* lr, int k, int which /* 0 dQpOffset, 1 dCuTreeOffset */, int* count);
int  orc_lowres_is_fade_end(const orc_lowres* lr);
void orc_collect_picture_statistics(const pixel* lowresPlane0, intptr_t lowresStride, const pixel* y, intptr_t sy,
                                    const pixel* u, const pixel* v, intptr_t sc, int w, int h, orc_hist_stats* out);

/* ---- MCSTF motion search (orc_mcstf.c) ---- */
typedef struct orc_mcstf_planes {
    const pixel* lowerRes;    /* Lowres::lowerResPlane[0] (origin) */
    const pixel* lowres;      /* Lowres::lowresPlane[0] */
    intptr_t lumaStride;      /* Lowres::lumaStride; the quarter-res planes use half of it */
    int width, lines;         /* Lowres::width / lines */
    const pixel* pic;         /* PicYuv::m_picOrg[0] of the border-extended source picture */
    intptr_t picStride;
    int picWidth, picHeight;
} orc_mcstf_planes;
typedef struct orc_mcstf_mvs {  /* the arrays of one TemporalFilterRefPicInfo (source/common/temporalfilter.h:155-172) */
    orc_mv *mvs0, *mvs1, *mvs2, *mvs;
    uint32_t mvsStride0, mvsStride1, mvsStride2, mvsStride;
    int* error;
} orc_mcstf_mvs;
int  orc_mcstf_motion_error(const pixel* src, int stride, const pixel* buf, int x, int y, int dx, int dy, int bs,
                            int besterror);
void orc_mcstf_me_luma(orc_mv* mvs, uint32_t mvStride, const pixel* src, int stride, int height, int width,
                       const pixel* buf, int row, int rowSize, const orc_mv* previous, uint32_t prevMvStride, int factor);
void orc_mcstf_me_luma_double_res(orc_mv* mvs, uint32_t mvStride, const pixel* orig, int stride, int origWidth,
                                  int origHeight, const pixel* buffer, const orc_mv* previous, uint32_t prevMvStride,
                                  int factor, int* minError, int row, int rowSize);
void orc_mcstf_level(int level, const orc_mcstf_planes* cur, const orc_mcstf_planes* ref, orc_mcstf_mvs* out);

/* field accessors for ctypes */
void* orc_lowres_ptr(orc_lowres* lr, const char* field, int i, int j);
int64_t orc_lowres_i64(orc_lowres* lr, const char* field, int i, int j);

#ifdef __cplusplus
}
#endif
#endif
