/* x265gpu: C-ABI over hand-written gfx950 (MI355X) HIP kernels for x265's
 * lowres lookahead: half-res downscale with half-pel planes, the 8x8 intra
 * estimate, and per-8x8-block motion search / bidir cost estimation.
 *
 * The boundary sits one level above x265's per-block EncoderPrimitives slots
 * (source/common/primitives.h:239-432): a device round trip per 8x8 call is
 * impossible, so each entry point replaces one of the batch points x265's own
 * lookahead already has.  The host keeps x265's decision logic, request order,
 * float math (MV-cost table, AQ, weight guessing) and the Lowres bookkeeping;
 * this library owns a device-side mirror of the per-frame Lowres arrays
 * (source/common/lowres.h:164-258) in "slots".
 *
 * Conventions: every function returns X265GPU_OK (0) or a negative error
 * code; nothing throws; all host buffers are caller-owned; strides are in
 * PIXELS as in x265; `pixel` is uint8_t when bitDepth == 8 and uint16_t
 * otherwise (source/common/common.h:127-141).  A context is bound to one
 * device and one HIP stream.  Every entry point holds the context's lock while
 * it runs, so calls on one context from several host threads take turns; what
 * a begin/finish pair or x265gpu_last_error means across threads stays the
 * caller's business.  Different contexts run side by side.
 * There is no CPU fallback: if no gfx950 device or no kernel image is
 * available, x265gpu_open fails.
 */
#ifndef X265GPU_H
#define X265GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define X265GPU_OK            0
#define X265GPU_ERR_ARG      -1   /* bad argument / out-of-range slot, distance or size */
#define X265GPU_ERR_DEVICE   -2   /* HIP runtime error (see x265gpu_last_error) */
#define X265GPU_ERR_NOMEM    -3
#define X265GPU_ERR_STATE    -4   /* call order violated (e.g. search before MV-cost table upload) */
#define X265GPU_ERR_UNSUPPORTED -5

#define X265GPU_BFRAME_MAX 16      /* X265_BFRAME_MAX, source/x265.h:599 */

/* numbering of X265_*_SEARCH in source/x265.h */
#define X265GPU_DIA_SEARCH  0
#define X265GPU_HEX_SEARCH  1
#define X265GPU_UMH_SEARCH  2
#define X265GPU_STAR_SEARCH 3
#define X265GPU_SEA_SEARCH  4
#define X265GPU_FULL_SEARCH 5
/* not an x265 method value: MotionEstimate::diamondSearch (source/encoder/motion.cpp:631-788), the coarse
 * search behind an area's starting vector (Search::puMotionEstimation with isMVP, source/encoder/search.cpp:360-364).
 * merange, subpelRefine and the candidate list are ignored; the result vector is in FULL pixels, as there. */
#define X265GPU_AREA_DIAMOND_SEARCH 16

/* where a caller-supplied buffer lives */
#define X265GPU_MEM_HOST   0
#define X265GPU_MEM_DEVICE 1

typedef struct x265gpu_ctx x265gpu_ctx;

/* The subset of x265_param the path reads (source/x265.h; defaults
 * source/common/param.cpp:197-260,310-315). */
typedef struct x265gpu_config
{
    int32_t sourceWidth;          /* the size of the planes the caller hands over, at least 16.  An odd size (x265 takes */
    int32_t sourceHeight;         /* one for 4:0:0 / 4:4:4 input only, param.cpp:1791-1794) is treated as x265 does: the */
                                  /* lookahead's geometry is that of the picture padded to the minimum CU size of 8      */
                                  /* (Encoder::configure, encoder.cpp:4470-4474, 4689-4693; Lowres::create sizes from    */
                                  /* that, lowres.cpp:84-103), its samples beyond the source repeat the last column /   */
                                  /* row (picyuv.cpp:572-583); the 4:2:0 entry points (AQ, histogram) then refuse        */
    int32_t internalBitDepth;     /* 8, 10 or 12 */
    int32_t maxCUSize;            /* CTU size; margins are maxCUSize+32 / maxCUSize+16 */
    int32_t bframes;              /* sizes the [bframes+2] result arrays */
    int32_t numSlots;             /* device-resident Lowres mirrors to allocate */
    int32_t searchMethod;         /* lowres search; x265 hard-wires HEX (slicetype.cpp:4402-4404) */
    int32_t merange;              /* Lookahead::s_merange = 16 (slicetype.h:337) */
    int32_t bEnableHME;
    int32_t hmeSearchMethod[3];
    int32_t hmeRange[3];
    int32_t bFrameBias;
    int32_t maxSlices;            /* param->maxSlices (encoder slices).  Accepted and without effect on the lowres path:
                                     above 1 motionEstimate clamps a full-pel result whose row left mvmin.y..mvmax.y
                                     (motion.cpp:1654-1659), and without extra candidates every search method already
                                     tests each candidate's row against that range */
    int32_t lookaheadLambda;      /* (int)x265_lambda_tab[X265_LOOKAHEAD_QP]: 1, 16 or 256 */
    int32_t deviceId;
    void*   stream;               /* hipStream_t to run on, or NULL for the library's own */
    int32_t bKeepFullRes;         /* keep a padded full-resolution luma plane per slot (for x265gpu_pu_motion_estimate) */
    int32_t bMcstfFilter;         /* param->bEnableTemporalFilter: Lowres::init downscales with the (a+b+c+d+2)>>2 filter
                                     and also makes the quarter-res planes (source/common/lowres.cpp:377-378, :388-398) */
    int32_t lookaheadSlices;      /* param->lookaheadSlices as Lookahead::Lookahead sees it (after the "no pool" and
                                     "height < 720" resets, source/encoder/slicetype.cpp:1110-1120); 0 or 1: whole frames.
                                     The slice layout follows :1122-1133 and is reported by x265gpu_get_geometry. */
} x265gpu_config;

/* Geometry derived exactly as Lowres::create does (source/common/lowres.cpp:84-103). */
typedef struct x265gpu_geometry
{
    int32_t width, lines;         /* lowres size rounded up to multiples of 8 */
    int32_t stride;               /* lumaStride, pixels */
    int32_t marginX, marginY;
    int32_t widthInCU, heightInCU, cuCount;
    int32_t widthInCU4, heightInCU4;   /* quarter-res (HME level 0) block grid, 0 if HME is off */
    int64_t planeSize;            /* pixels per padded plane */
    int64_t padOffset;            /* pixels from buffer start to pixel (0,0) */
    int32_t numCoopSlices;        /* Lookahead::m_numCoopSlices (1: whole frames) */
    int32_t numRowsPerSlice;      /* Lookahead::m_numRowsPerSlice; the last slice also takes the remainder rows */
    int32_t sourceWidth, sourceHeight;   /* the picture size the context was opened with */
} x265gpu_geometry;

/* One cost-estimation request = one CostEstimateGroup::estimateFrameCost body
 * (source/encoder/slicetype.cpp:4284-4376) with the cache check and the
 * bDoSearch decision already taken by the caller. */
typedef struct x265gpu_request
{
    int32_t slotP0, slotB, slotP1;    /* slotP1 == slotB for a P request */
    int32_t p0, b, p1;                /* frame indices: distances are b-p0 and p1-b */
    int32_t doSearch[2];              /* run the L0 / L1 motion search for this request */
    int32_t weightedL0;               /* L0 search reads the weighted planes set with x265gpu_weight_planes */
    int32_t bCoopSlices;              /* the request is a CostEstimateGroup::singleCost outside batch mode on a lookahead
                                         with cooperative slices (slicetype.cpp:4312-4343): every slice's bottom block row
                                         starts without the predictors of the row below it (:4248-4276).  Ignored when
                                         the context has one slice.  Only the searches differ; sums and per-block costs
                                         are the totals over all slices, as x265 adds them up (:4336-4342). */
    int32_t bSearchOnly;              /* run the motion searches of doSearch[] and nothing else: no per-block costs, no
                                         sums (the request's x265gpu_frame_cost comes back as -1).  For a caller that
                                         issues, inside a batch, the searches of requests it will make one by one later
                                         (then with doSearch cleared): the vectors are those the later request would
                                         have found, since a search depends only on the two frames, the list, the
                                         distance, p1 > b and bCoopSlices. */
} x265gpu_request;

/* per-request scalars, as stored by x265 in Lowres::costEst/costEstAq/intraMbs */
typedef struct x265gpu_frame_cost
{
    int64_t costEst;      /* after the B-frame *100/(130+bFrameBias) scaling */
    int64_t costEstAq;
    int32_t intraMbs;     /* blocks that chose intra in a P request (added onto the slot's running count) */
    int32_t reserved;
} x265gpu_frame_cost;

/* ---- lifetime --------------------------------------------------------- */

/* Replaces Lookahead::create + LookaheadTLD setup for the device side
 * (source/encoder/slicetype.cpp:1061-1066, 1213-1216). */
int x265gpu_open(const x265gpu_config* cfg, x265gpu_ctx** out);
void x265gpu_close(x265gpu_ctx* ctx);
int x265gpu_get_geometry(const x265gpu_ctx* ctx, x265gpu_geometry* out);
const char* x265gpu_last_error(const x265gpu_ctx* ctx);
/* Waits for all work queued by this context. */
int x265gpu_sync(x265gpu_ctx* ctx);
/* Page-locked host memory for the arrays the caller hands over with X265GPU_MEM_HOST (source frames, PU job
 * and result arrays): transfers from pageable memory are staged by the driver at a fraction of the PCIe
 * rate.  Stands where x265 allocates the same buffers with x265_malloc (source/common/common.cpp:150-170);
 * any host pointer is still accepted everywhere. */
int x265gpu_host_alloc(x265gpu_ctx* ctx, size_t bytes, void** out);
int x265gpu_host_free(x265gpu_ctx* ctx, void* p);

/* Uploads the host-built BitCost table (source/encoder/bitcost.cpp:31-56):
 * `centre` points at the MVD-0 entry, valid for [-maxMvd, +maxMvd]; maxMvd >= 1024 (x265's table spans +-2^17
 * entries; the search kernels keep the entries around MVD 0 in LDS without testing for a shorter table), else
 * X265GPU_ERR_ARG. */
int x265gpu_set_mvcost(x265gpu_ctx* ctx, const uint16_t* centre, int32_t maxMvd);

/* ---- pre-lookahead: PreLookaheadGroup::processTasks (slicetype.cpp:1836-1861) */

/* Lowres::init (source/common/lowres.cpp:344-402): resets the slot's
 * sentinels and cost caches, runs the half-res downscale to four planes,
 * extends their borders, and with HME the quarter-res set.  `luma` points at
 * pixel (0,0) of the full-res luma plane; columns/rows beyond the picture are
 * taken as edge replicas, which is what PicYuv::copyFromPicture stores there
 * (source/common/picyuv.cpp:291-300,572-583). */
int x265gpu_lowres_init(x265gpu_ctx* ctx, int32_t slot, const void* luma, intptr_t stride,
                        int32_t memKind, int32_t poc);
/* Batched form: n frames in one launch sequence. */
int x265gpu_lowres_init_batch(x265gpu_ctx* ctx, int32_t n, const int32_t* slots, const void* const* lumas,
                              const intptr_t* strides, int32_t memKind, const int32_t* pocs);

/* invQscaleFactor from the host's calcAdaptiveQuantFrame (slicetype.cpp:454-704),
 * one int per 8x8 lowres block; NULL clears it (costEstAq == costEst).  The
 * array is copied before the call returns (the upload itself stays asynchronous). */
int x265gpu_lowres_set_aq(x265gpu_ctx* ctx, int32_t slot, const int32_t* invQscaleFactor);

/* LookaheadTLD::lowresIntraEstimate (slicetype.cpp:706-815) for n slots.  It may be called before or after the
 * slot's scale factors are given (x265gpu_lowres_set_aq, the x265gpu_aq_frame* calls): the per-block costs and modes
 * do not depend on them, and the AQ-weighted row and frame sums are redone with whatever factors are installed
 * after the estimate was launched, before any call reads them. */
int x265gpu_intra_estimate(x265gpu_ctx* ctx, int32_t n, const int32_t* slots);

/* ---- cost estimation: CostEstimateGroup::add/finishBatch/singleCost
 *      (slicetype.cpp:4148-4196) -> estimateFrameCost -> estimateCUCost ---- */

/* Executes n requests.  All motion searches of the batch run first (each one
 * a bottom-right to top-left wavefront over the 8x8 blocks), then all
 * per-block cost evaluations; a request may therefore use vectors searched by
 * any request of the same batch, and requests of one batch must not search
 * the same (frame, list, distance) twice: such a batch is X265GPU_ERR_ARG. */
int x265gpu_estimate(x265gpu_ctx* ctx, int32_t n, const x265gpu_request* reqs);
/* Same, then waits and returns the n per-request scalars in one transfer; the
 * per-block arrays stay on the device until a getter asks for them.  This is
 * what CostEstimateGroup::finishBatch's caller consumes (slicetype.cpp:2947,
 * 2987: only costEst/intraMbs are read before the next decision). */
int x265gpu_estimate_costs(x265gpu_ctx* ctx, int32_t n, const x265gpu_request* reqs, x265gpu_frame_cost* out);

/* Same work, results left on the device and no wait: request i's frame totals go to deviceSums[3i .. 3i+2] =
 * { costEst BEFORE the B-frame scaling of slicetype.cpp:4372-4373, costEstAq, intraMbs } (device memory of this
 * context's GPU, 24 bytes per request), ordered on the context's stream.  For callers that pass the totals on
 * without reading them on this host first: a multi-GPU lookahead gathers them over RCCL (each rank owns a
 * frame segment) and only the rank that runs slicetypeAnalyse applies the scaling and reads them. */
int x265gpu_estimate_sums_device(x265gpu_ctx* ctx, int32_t n, const x265gpu_request* reqs, int64_t* deviceSums);

/* ---- weighted prediction analysis: LookaheadTLD::weightsAnalyse
 *      (slicetype.cpp:870-971); the float guessing stays with the caller ---- */

/* weightCostLuma (slicetype.cpp:817-850): sum over 8x8 blocks of
 * min(SATD(ref [weighted], fenc), intraCost).  wtPresent == 0 measures the
 * unweighted reference.  The result is written to *cost after a sync.  Reads slotFenc's per-block intra costs:
 * x265gpu_intra_estimate for that slot must have been issued (it need not have been given its AQ factors yet). */
int x265gpu_weight_cost(x265gpu_ctx* ctx, int32_t slotFenc, int32_t slotRef, int32_t wtPresent,
                        int32_t scale, int32_t log2Denom, int32_t offset, uint32_t* cost);
/* n x265gpu_weight_cost measurements with one wait at the end (the weight analyses of a batch of requests are
 * independent of each other).  costs[i] receives job i's sum. */
typedef struct x265gpu_weight_job
{
    int32_t slotFenc, slotRef;
    int32_t wtPresent, scale, log2Denom, offset;
} x265gpu_weight_job;
int x265gpu_weight_costs(x265gpu_ctx* ctx, int32_t n, const x265gpu_weight_job* jobs, uint32_t* costs);
/* x265gpu_weight_cost for callers that cannot allocate device memory (tight serving loops, harnesses that
 * forbid hipMalloc).  It has no counterpart in the reference.  The call still waits for its result and
 * copies it to `cost` in ordinary host memory, so it is not something a stream capture can record.  Same
 * arguments and result as x265gpu_weight_cost; the accumulator and the weighted copy of plane 0 live in a
 * fixed 8 MiB array inside the library's device image (one copy per GPU, held until the process ends)
 * instead of buffers the context allocates or grows, and the call itself performs no hipMalloc / hipFree.
 * Limit: with wtPresent the padded half-res plane (stride x (lines + 2 marginY) pixels) plus 256 bytes
 * must fit the array - up to 3840x2160 at 10 bit; larger pictures return X265GPU_ERR_ARG with a message
 * in x265gpu_last_error.  Every context on one GPU shares that GPU's array and the array keeps the last
 * call's contents (the call clears what must start at zero).  Calls made through this library in one
 * process take turns on a per-GPU lock; calls from different processes on one GPU use different copies
 * of the array.  Anything else that touched the array while a call runs would race with it. */
int x265gpu_weight_cost_static(x265gpu_ctx* ctx, int32_t slotFenc, int32_t slotRef, int32_t wtPresent,
                               int32_t scale, int32_t log2Denom, int32_t offset, uint32_t* cost);
/* weight_pp over all four padded planes of slotRef (slicetype.cpp:957-969);
 * the result becomes the L0 reference of requests with weightedL0 set, for
 * (slotFenc, dist) until replaced. */
int x265gpu_weight_planes(x265gpu_ctx* ctx, int32_t slotFenc, int32_t dist, int32_t slotRef,
                          int32_t scale, int32_t log2Denom, int32_t offset);

/* ---- results ---------------------------------------------------------- */

/* Each getter syncs the context's stream and copies into caller memory; any
 * output pointer may be NULL. */
int x265gpu_get_plane(x265gpu_ctx* ctx, int32_t slot, int32_t planeIdx /*0..3, 4..7 quarter-res*/,
                      void* paddedBuffer);
int x265gpu_get_intra(x265gpu_ctx* ctx, int32_t slot, int32_t* intraCost, uint8_t* intraMode);
int x265gpu_get_mvs(x265gpu_ctx* ctx, int32_t slot, int32_t list, int32_t dist,
                    int32_t* mvsXY /* cuCount x {x,y} */, int32_t* mvCosts);
int x265gpu_get_hme_mvs(x265gpu_ctx* ctx, int32_t slot, int32_t list, int32_t dist,
                        int32_t* mvsXY, int32_t* mvCosts);
int x265gpu_get_costs(x265gpu_ctx* ctx, int32_t slot, int32_t d0, int32_t d1,
                      uint16_t* lowresCosts, int32_t* rowSatds, x265gpu_frame_cost* frameCost);

/* The frame totals of n (slot, d0, d1) results with one wait: x265gpu_get_costs' frameCost for each.  d0 / d1 may be
 * NULL for the intra estimate's totals (costEst[0][0], costEstAq[0][0] of Lowres, slicetype.cpp:795-812). */
int x265gpu_get_frame_costs(x265gpu_ctx* ctx, int32_t n, const int32_t* slots, const int32_t* d0, const int32_t* d1,
                            x265gpu_frame_cost* out);

/* ---- instrumentation --------------------------------------------------- */

#define X265GPU_PHASE_LOWRES 0
#define X265GPU_PHASE_INTRA  1
#define X265GPU_PHASE_SEARCH 2
#define X265GPU_PHASE_COST   3
#define X265GPU_PHASE_COUNT  4

/* When enabled, every kernel group is bracketed by HIP events on the
 * context's stream; totals are read back (and optionally reset) here. */
int x265gpu_profile_enable(x265gpu_ctx* ctx, int32_t on);
int x265gpu_profile_read(x265gpu_ctx* ctx, int32_t phase, double* ms, int64_t* launches, int32_t reset);

/* ---- full-resolution motion estimation for batches of prediction units ----
 *
 * One job = one MotionEstimate::setSourcePU + MotionEstimate::motionEstimate call
 * on a full-resolution reference (source/encoder/motion.cpp:167-191, 923-1773) as
 * Search::puMotionEstimation issues them (source/encoder/search.cpp:226-520; with
 * --threaded-me per CTU-row batch, source/encoder/threadedme.cpp:207-261).  The
 * caller keeps the AMVP / neighbour logic and Search::setSearchRange
 * (search.cpp:4969-5022) and passes their results.  Pictures are the slots'
 * full-resolution planes (config.bKeepFullRes), padded the way PicYuv +
 * extendPicBorder pad them (margins maxCUSize+32 / maxCUSize+16). */
typedef struct x265gpu_pu_job
{
    int32_t x, y, w, h;                   /* PU rectangle, pixels; w and h multiples of 4, at most 64 */
    int32_t mvpX, mvpY;                   /* quarter-pel predictor (qmvp) */
    int32_t mvminX, mvminY, mvmaxX, mvmaxY;   /* full-pel bounds (mvmin / mvmax) */
    int32_t numCandidates;                /* extra quarter-pel candidates mvc[], at most 12 */
    int32_t mvc[24];                      /* x, y pairs */
    int32_t flags;                        /* X265GPU_PU_VERT_RESTRICTION or 0, | X265GPU_PU_COST_TABLE(k) */
} x265gpu_pu_job;

/* motionEstimate's m_vertRestriction argument (the reference picture is the current picture, search.cpp:367): the
 * clipped predictor's y is also held to mvmax.y << 2 (motion.cpp:957-963).  With mvmin <= mvmax, which this entry
 * point requires, clipping has already done that; the restriction that matters is the one Search::setSearchRange
 * applies to mvmax (search.cpp:4975-4982), which is the caller's. */
#define X265GPU_PU_VERT_RESTRICTION 1
/* Which of the resident MV-cost tables prices the job's vectors (x265gpu_me_set_mvcost_tables): with delta QP on, a
 * CU's searches run under its own QP's lambda (Analysis::setLambdaFromQP before every puMotionEstimation call,
 * source/encoder/analysis.cpp:188, 243, 259, 274).  Table 0 when left out. */
#define X265GPU_PU_COST_TABLE(k) ((int32_t)(k) << 8)
#define X265GPU_PU_MAX_COST_TABLES 128

typedef struct x265gpu_pu_result
{
    int32_t mvX, mvY;                     /* outQMv, quarter-pel */
    int32_t cost;                         /* motionEstimate's return value */
    int32_t reserved;
} x265gpu_pu_result;

/* BitCost table for the full-resolution searches (its lambda follows the slice QP,
 * unlike the lookahead's): same meaning as x265gpu_set_mvcost. */
int x265gpu_me_set_mvcost(x265gpu_ctx* ctx, const uint16_t* centre, int32_t maxMvd);
/* Several such tables at once, one per QP the picture's CUs use (BitCost keeps one per QP, bitcost.cpp:31-60);
 * jobs choose theirs with X265GPU_PU_COST_TABLE.  Replaces whatever was set before. */
int x265gpu_me_set_mvcost_tables(x265gpu_ctx* ctx, int32_t numTables, const uint16_t* const* centres, int32_t maxMvd);

/* searchMethod: X265GPU_DIA_SEARCH, _HEX_, _UMH_, _STAR_, _SEA_ or _FULL_SEARCH (param->searchMethod);
 * merange: param->searchRange; subpelRefine: param->subpelRefine (0..7, workload table
 * motion.cpp:48-58).  Results come back in job order; the call returns when they are in `out`. */
/* X265GPU_SEA_SEARCH (motion.cpp:1438-1591): the block sums FrameFilter::computeMEIntegral keeps per reference
 * (source/encoder/framefilter.cpp:731-831) are made on the device from the reference slot's padded picture when
 * first needed and kept until that slot is uploaded again or another reference is searched.  merange up to 126.
 * The reference's elimination test for 8x4, 4x8, 32x8 and 8x32 PUs compares sums taken over encode-buffer bytes
 * outside the PU (what an earlier PU left there), so no result is defined for them: X265GPU_ERR_UNSUPPORTED, as for
 * sizes HEVC has no partition for.  A window whose block sums would leave the padded picture (the reference reads
 * past its planes there) is X265GPU_ERR_ARG: keep mvmax 32 pixels inside the padding. */
int x265gpu_pu_motion_estimate(x265gpu_ctx* ctx, int32_t slotFenc, int32_t slotRef, int32_t searchMethod,
                               int32_t merange, int32_t subpelRefine, int32_t n, const x265gpu_pu_job* jobs,
                               x265gpu_pu_result* out);

/* ---- motion search of the motion-compensated spatio-temporal filter (--mcstf) ----
 *
 * What the lookahead runs for a picture that is going to be filtered, for each of its reference pictures:
 * MotionEstimatorTLD::motionEstimationLuma on the quarter-res, half-res and full-res pictures, then
 * ::motionEstimationLumaDoubleRes in 1/16 pixel (source/common/temporalfilter.cpp:421-478, 642-897), issued as
 * 16-line row tasks, one batch per level (source/encoder/slicetype.cpp:2280-2297, 4217-4234).  The context
 * needs bMcstfFilter and bKeepFullRes; every slot involved must have been through x265gpu_lowres_init with
 * its unfiltered source picture (x265 searches copies taken before filtering, source/encoder/encoder.cpp:1973).
 * Outputs follow the arrays of TemporalFilterRefPicInfo (temporalfilter.cpp:397-409), reference after reference,
 * W x H the source size:
 *   mvs    : (W/4)*(H/4) x {x,y} per reference, 1/16 pixel per 8x8 block, W/4 entries per block row  (::mvs, ::mvsStride)
 *   error  : (W/4)*(H/4) per reference, same indexing                                              (::error)
 *   coarse : optional; 3 x (W/16)*(H/16) x {x,y} per reference: ::mvs0, ::mvs1, ::mvs2, W/16 entries per row
 * Entries the reference does not write are zero.  At most 16 references (MCSTF_MAX_REFS).  The final error
 * is formed in double precision with every operation rounded on its own; a reference build whose compiler
 * regroups that expression under -ffast-math may differ in the last unit there. */
int x265gpu_mcstf_motion_search(x265gpu_ctx* ctx, int32_t slotCur, int32_t numRefs, const int32_t* slotRefs,
                                int32_t* mvs, int32_t* error, int32_t* coarse);

/* ---- adaptive-quantisation block energies and weighted-prediction frame sums ----
 *
 * The pixel passes of LookaheadTLD::calcAdaptiveQuantFrame (source/encoder/slicetype.cpp:454-704) for a
 * 4:2:0 picture and qgSize >= 16: energy[by * ceil(w/16) + bx] = acEnergyCu(blockX, blockY) (:266-285), and
 * the raw per-plane accumulators Lowres::wp_sum[] / wp_ssd[] as acEnergyVar leaves them (:51-59), before the
 * mean is removed at :678-685.  The caller keeps the float mapping from energy to QP offsets.  Planes are the
 * unpadded source planes (luma sourceWidth x sourceHeight, chroma half of each); strides in pixels. */
int x265gpu_aq_energy(x265gpu_ctx* ctx, const void* y, intptr_t strideY, const void* u, const void* v, intptr_t strideC,
                      int32_t memKind, uint32_t* energy, uint64_t wpSum[3], uint64_t wpSsd[3]);

/* The whole of LookaheadTLD::calcAdaptiveQuantFrame (source/encoder/slicetype.cpp:454-704) for a 4:2:0
 * picture with rc.aqMode NONE / VARIANCE / AUTO_VARIANCE / AUTO_VARIANCE_BIASED / EDGE: the block energies come from
 * the device pass above, the double-precision mapping to QP offsets (:486-505, :539-643) and x265_exp2fix8
 * (source/common/common.cpp:96-103) run on the host in this call, in the reference's order of operations.
 * qgSize >= 16: arrays hold one entry per 16x16 block, ceil(w/16) per row.  qgSize 8: 8x8 luma + 4x4 chroma
 * blocks and the constants of :464-466; arrays hold (2*widthInCU) * (2*heightInCU) entries (Lowres's
 * cuCountFullRes, source/common/lowres.cpp:92-95), the block loop fills the first ceil(w/8) * ceil(h/8) of
 * them in loop order exactly as the reference's running index does (:571, :641), the rest keep what the
 * caller put there (the reference allocates them zeroed), and invQscaleFactor8x8 receives the per-lowres-block
 * means of :647-660 - that array is what the estimate calls use with qgSize 8 (:798-799, :4532-4533), so it is
 * the one installed for the slot (it is formed whenever qgSize is 8, offsets are not skipped and both arrays are
 * given, as the reference does even with aqMode NONE).  Arrays are otherwise left untouched when aqMode is NONE (x265 does not allocate them
 * then) or when bSkipOffsets is set (the rc.bStatRead && cuTree && referenced-frame case, :483).  wpSum /
 * wpSsd receive Lowres::wp_sum / wp_ssd as the function leaves them: with bWeightedPred the mean is removed
 * from wp_ssd (:663-686); without it and without AQ no energy pass runs and both are zero.
 * bBlockVariance (param->bDynamicRefine || param->bEnableFades) runs the tail of :688-703: blockVariance in
 * loop order over the picture (rounded to the nearest 16 each way when bWeightedPred is set, :674-675) and
 * frameVariance, with the reference's running row total; as there, that pass adds to wp_sum / wp_ssd a
 * second time.  With slot >= 0 the resulting scale factors are also installed for that slot (as
 * x265gpu_lowres_set_aq does).
 * bHevcAq (param->rc.hevcAq, :517-521): xPreanalyze (:339-452) replaces the energy mapping.  For every layer the
 * caller lists (the enabled depths of aqLayerDepth, source/common/lowres.h:120-139, with the part sizes and part
 * counts Lowres::create gave them, source/common/lowres.cpp:128-155) the device sums each part's four quadrants,
 * the host forms dActivity = 1 + the smallest quadrant variance, dAvgActivity, and xPreanalyzeQp's dQpOffset /
 * dCuTreeOffset (:303-337) with qpAdaptationRange.  The LAST layer listed is the one of minAQDepth: its
 * dQpOffset becomes invQscaleFactor through x265_exp2fix8 in visiting order (:441-447), and acEnergyCu runs
 * once per block for wp_sum / wp_ssd (:449).  qpAqOffset / qpCuTreeOffset are not written in this mode.
 * X265_AQ_EDGE (aqMode 4, :529-606): edgeFilter's blurred picture, Sobel gradients and edge picture (:100-225) and
 * edgeDensityCu's block sums (:227-263) are one more device pass; as in the reference, the edge picture's block
 * sums are added to plane 0 of wp_sum / wp_ssd.  A pixel's gradient angle in whole degrees comes from the device
 * wherever the degree is certain within the accuracy of the device's atan2f; the few pixels close to a whole
 * degree are listed and get their angle from the host's atan2f in this call, so the angle sums are the ones a
 * host-only run with this machine's libm gives.  The reference's other consumer of the edge picture
 * (recursionSkipMode's m_edgeBitPic) and Lowres::edgeInclined are not outputs. */
typedef struct x265gpu_hevc_aq_layer
{
    int32_t aqPartWidth, aqPartHeight;            /* PicQPAdaptationLayer's fields of the same names */
    int32_t numAQPartInWidth, numAQPartInHeight;
    double* dActivity;                            /* out, numAQPartInWidth * numAQPartInHeight each; the parts the */
    double* dQpOffset;                            /* loops visit, ceil(w / aqPartWidth) * ceil(h / aqPartHeight), */
    double* dCuTreeOffset;                        /* are written in visiting order                                */
    double  dAvgActivity;                         /* out */
} x265gpu_hevc_aq_layer;
typedef struct x265gpu_aq_params
{
    int32_t aqMode;               /* param->rc.aqMode, 0..4 */
    int32_t bWeightedPred;        /* param->bEnableWeightedPred || param->bEnableWeightedBiPred */
    int32_t bHDR10Opt;            /* param->bHDR10Opt: luma-average dependent offset (:613-635) */
    int32_t bSkipOffsets;
    double  aqStrength;           /* param->rc.aqStrength */
    int32_t qgSize;               /* param->rc.qgSize: 8, 16, 32, 64 (0 = 32) */
    int32_t bBlockVariance;       /* param->bDynamicRefine || param->bEnableFades */
    int32_t*  invQscaleFactor8x8; /* out, qgSize 8: widthInCU * heightInCU (Lowres::invQscaleFactor8x8) */
    uint32_t* blockVariance;      /* out, bBlockVariance: Lowres::blockVariance, same size as the offset arrays */
    double*   frameVariance;      /* out, bBlockVariance: Lowres::frameVariance */
    int32_t bHevcAq;              /* param->rc.hevcAq */
    int32_t numLayers;            /* bHevcAq: entries of layers[], 1..4, in increasing depth */
    double  qpAdaptationRange;    /* param->rc.qpAdaptationRange */
    x265gpu_hevc_aq_layer* layers;
} x265gpu_aq_params;
int x265gpu_aq_frame(x265gpu_ctx* ctx, int32_t slot, const void* y, intptr_t strideY, const void* u, const void* v,
                     intptr_t strideC, int32_t memKind, const x265gpu_aq_params* params,
                     const float* quantOffsets /* Frame::m_quantOffsets or NULL */, double* qpAqOffset,
                     double* qpCuTreeOffset, int32_t* invQscaleFactor, uint64_t wpSum[3], uint64_t wpSsd[3]);

/* x265gpu_aq_frame for the n pictures of a pre-lookahead batch (PreLookaheadGroup::processTasks hands one frame to
 * each pool worker, slicetype.cpp:1836-1861): every picture's device pass is queued first, their results come back
 * in one transfer, and the double-precision mapping of the pictures runs on host threads side by side (as many as
 * X265GPU_HOST_THREADS says, by default the host's cores up to 16).  Fields are x265gpu_aq_frame's arguments; results
 * are identical to n calls of it.  Pictures in host memory, qgSize 8, rc.hevcAq and bBlockVariance are simply
 * handled by those n calls. */
typedef struct x265gpu_aq_picture
{
    int32_t slot;                          /* slot that receives the scale factors, or -1 */
    const void* y; const void* u; const void* v;
    intptr_t strideY, strideC;
    const float* quantOffsets;             /* or NULL */
    double* qpAqOffset; double* qpCuTreeOffset; int32_t* invQscaleFactor;    /* out */
    uint64_t wpSum[3], wpSsd[3];           /* out */
} x265gpu_aq_picture;
int x265gpu_aq_frames(x265gpu_ctx* ctx, int32_t n, x265gpu_aq_picture* pics, int32_t memKind, const x265gpu_aq_params* params);

/* The same in two halves, so that the host's double-precision mapping of the offsets (the part of
 * calcAdaptiveQuantFrame that x265 also runs on CPU threads, slicetype.cpp:1843-1859) overlaps the motion searches.
 * x265gpu_aq_frames_begin runs the energy pass, waits for it and fills wpSum / wpSsd (what LookaheadTLD::weightsAnalyse
 * reads); qpAqOffset, qpCuTreeOffset and invQscaleFactor are being written by library threads when it returns and
 * the arrays (and `quantOffsets`) must stay where they are.  x265gpu_aq_frames_finish joins those threads and installs
 * the scale factors; after it the arrays are the caller's again.  An x265gpu_intra_estimate call made until then
 * starts at once (the per-block intra costs and modes, which the weight analysis and the cost passes read, do not
 * depend on AQ); only its AQ-weighted row and frame sums are redone when the factors are installed.  Every call that
 * reads scale factors or those sums - the cost pass of an estimate call (after its searches have been queued), the
 * cuTree calls, the getters, x265gpu_sync - finishes the pending batches first, so only the caller's own reads of the
 * three arrays need the explicit call. */
int x265gpu_aq_frames_begin(x265gpu_ctx* ctx, int32_t n, x265gpu_aq_picture* pics, int32_t memKind, const x265gpu_aq_params* params);
int x265gpu_aq_frames_finish(x265gpu_ctx* ctx);
/* --aq-motion: Lookahead::calcMotionAdaptiveQuantFrame (source/encoder/slicetype.cpp:3609-3663) for frame `slot`
 * and its request (d0 = b - p0, d1 = p1 - b), which must have been estimated (x265 itself reads never-written arrays
 * when it has not been; a caller that wants a first-use frame's outcome there skips the call): the length of every block's vector
 * in the lists its decision used is formed on the device from the resident lowresMvs / lowresCosts (no vector
 * crosses to the host), the power of 0.1, the frame's mean and deviation and the offsets on the host (libm,
 * sums in block order).  qpAqMotionOffset is written; qpAqOffset, qpCuTreeOffset and invQscaleFactor (cuCount
 * entries each, the arrays of x265gpu_aq_frame) are added to where a block lies more than one deviation above the
 * mean, and the slot's scale factors follow.  The walk over the mini-GOP (Lookahead::aqMotion, :3581-3607) is
 * the caller's. */
int x265gpu_aq_motion_frame(x265gpu_ctx* ctx, int32_t slot, int32_t d0, int32_t d1, double* qpAqOffset,
                            double* qpCuTreeOffset, int32_t* invQscaleFactor, double* qpAqMotionOffset);
/* The auto-variance modes' block activity, pow(energy + 1, 0.1) in double (slicetype.cpp:541-578), is the host's
 * libm result, looked up per integer energy once libm has computed it (same function, same bits; energies recur
 * from picture to picture).  Process-wide counts since start: lookups, and those answered from the table. */
void x265gpu_aq_memo_stats(uint64_t* lookups, uint64_t* hits);

/* ---- cuTree: propagation of block costs along the lookahead's motion vectors ----
 *
 * The data passes of Lookahead::cuTree (source/encoder/slicetype.cpp:3665-3766) on the arrays the
 * estimate calls left on the device (intraCost, invQscaleFactor, lowresCosts, lowresMvs); the walk over
 * the mini-GOPs stays in the caller.  Each slot carries a propagateCost array (Lowres::propagateCost).
 *   x265gpu_cutree_clear      memset(frames[i]->propagateCost, 0, ...)                       (:3692-3728)
 *   x265gpu_cutree_swap       std::swap of two frames' propagateCost (lookaheadDepth 0)      (:3699, :3760)
 *   x265gpu_cutree_propagate  estimateCUPropagate(p0, p1, b, referenced) without its closing
 *                             cuTreeFinish: primitives.propagateCost (source/common/pixel.cpp:906-932)
 *                             per block row, then the scatter into frames p0 / p1        (:3768-3870)
 *                             d0 = b - p0, d1 = p1 - b; fpsFactor = CLIP_DURATION(frame) /
 *                             CLIP_DURATION(average) (:3781); bipredWeight as at :3772.  The request
 *                             (d0, d1) of slotB must have been estimated before.
 *   x265gpu_cutree_finish     cuTreeFinish for qgSize > 8 without hevcAq (:4024-4061): fpsFactor is the
 *                             integer of :4024, weightDelta that of :4025-4028, cuTreeStrength
 *                             Lookahead::m_cuTreeStrength (:1075).  log2 runs on the host (libm).
 *                             qpCuTreeOffset is read-modify-write: blocks with a zero scaled intra
 *                             cost keep their value.
 *   x265gpu_cutree_finish_qg8 the qgSize 8 loop of cuTreeFinish (:4030-4049): the slot's scale factors are
 *                             Lowres::invQscaleFactor8x8 (as x265gpu_aq_frame installs them with qgSize 8),
 *                             intra and propagated costs enter divided by 4, and each lowres block's ratio
 *                             is applied to the four entries it covers in offset arrays of
 *                             (2*widthInCU) * (2*heightInCU) doubles.  estimateCUPropagate itself only
 *                             swaps the scale-factor array with qgSize 8 (:3791-3798), which installing
 *                             invQscaleFactor8x8 for the slot already does.
 *   x265gpu_cutree_finish_hevc cuTreeFinish with rc.hevcAq = computeCUTreeQpOffset (:3876-4014): for every
 *                             layer listed (fields as for x265gpu_aq_frame; dQpOffset is read, dCuTreeOffset
 *                             written for all numAQPartInWidth * numAQPartInHeight parts) the mean log2 ratio
 *                             of the AQ blocks a part covers.  With qgSize 8 the per-block ratios go through
 *                             dCuTreeOffset8x8 ((2*widthInCU) * (2*heightInCU) doubles, read-modify-write as
 *                             PicQPAdaptationLayer::dCuTreeOffset8x8 is) and the slot's scale factors are
 *                             invQscaleFactor8x8.  Every part must reach into the picture (a part without
 *                             a block divides by zero in the reference).
 * The uint16 saturation of the reference's CLIP_ADD is order independent (all addends >= 0), so the
 * device adds with atomics into wide cells and clamps on read; results are bit-exact.
 * x265gpu_cutree_clear and x265gpu_cutree_propagate only queue their pass.  The queue is launched by the next call
 * that reads totals or waits (the finish calls, _get_propagate_cost, _swap, x265gpu_sync, the estimate calls): passes
 * that do not depend on each other - adds into one array commute, so all that orders them is a clear of an array
 * they touch and every add into the array they read - go out as one launch per dependency level (the unreferenced
 * B frames of a whole cuTree walk together, then the chain from each P frame to the one before it).  The outcome is
 * that of the calls in their order. */
int x265gpu_cutree_clear(x265gpu_ctx* ctx, int32_t slot);
int x265gpu_cutree_swap(x265gpu_ctx* ctx, int32_t slotA, int32_t slotB);
int x265gpu_cutree_propagate(x265gpu_ctx* ctx, int32_t slotP0, int32_t slotP1, int32_t slotB, int32_t d0, int32_t d1,
                             int32_t referenced, double fpsFactor, int32_t bipredWeight);
int x265gpu_cutree_get_propagate_cost(x265gpu_ctx* ctx, int32_t slot, uint16_t* propagateCost);
int x265gpu_cutree_finish(x265gpu_ctx* ctx, int32_t slot, int32_t fpsFactor, double weightDelta, double cuTreeStrength,
                          const double* qpAqOffset, double* qpCuTreeOffset);
int x265gpu_cutree_finish_qg8(x265gpu_ctx* ctx, int32_t slot, int32_t fpsFactor, double weightDelta,
                              double cuTreeStrength, const double* qpAqOffset, double* qpCuTreeOffset);
int x265gpu_cutree_finish_hevc(x265gpu_ctx* ctx, int32_t slot, int32_t fpsFactor, double weightDelta,
                               double cuTreeStrength, int32_t qgSize, int32_t numLayers,
                               x265gpu_hevc_aq_layer* layers, double* dCuTreeOffset8x8);

/* ---- picture statistics for histogram-based scene-cut detection (--hist-scenecut) ----
 *
 * LookaheadTLD::collectPictureStatistics (source/encoder/slicetype.cpp:1809-1834) for a 4:2:0 picture: the
 * luma histogram and means of the 4x4 segments of the quarter-sample picture that Lowres::init derives from
 * lowresPlane[0] (source/common/lowres.cpp:404-414; the slot must have been through x265gpu_lowres_init with
 * this picture's luma), the subsampled chroma histograms and means (:1688-1759), the frame means (:1827-1829)
 * and the three picture variances (:1567-1650).  Fields and index order are Lowres's own
 * (source/common/lowres.h:245-253).  The scene-cut decision that compares these between frames
 * (:3323-3482) stays in the caller. */
typedef struct x265gpu_hist_stats_t
{
    uint32_t picHistogram[4][4][3][256];              /* [segment in width][segment in height][plane][bin] */
    uint64_t averageIntensityPerSegment[4][4][3];
    uint8_t  averageIntensity[3];
    uint8_t  pad[5];
    uint16_t picAvgVariance, picAvgVarianceCb, picAvgVarianceCr, pad2;
} x265gpu_hist_stats_t;
int x265gpu_hist_stats(x265gpu_ctx* ctx, int32_t slot, const void* y, intptr_t strideY, const void* u, const void* v,
                       intptr_t strideC, int32_t memKind, x265gpu_hist_stats_t* out);

/* ---- the pixel primitives, one call per batch of blocks ----
 *
 * The EncoderPrimitives slots this library's kernels are built from, exposed on the slots'
 * full-resolution planes so that each can be checked against x265's C primitive:
 * pu[part].sad / pu[part].satd (pixelcmp_t, source/common/primitives.h:133; all 25 luma
 * partitions, source/common/pixel.cpp:1088-1172), cu[size].sa8d (pixel.cpp:328-369, 1180-1184),
 * and pu[part].luma_hpp / luma_vpp / luma_hvpp (filter_pp_t :176, filter_hv_pp_t :181,
 * source/common/ipfilter.cpp:79-369).  A compare job measures the block at (x,y) of the encode
 * slot against the reference slot's block displaced by the quarter-pel vector (qmvX,qmvY); with
 * a fractional vector the reference block is interpolated first, as subpelCompare does
 * (source/encoder/motion.cpp:1775-1803).  SA8D takes square blocks of 8..64 and full-pel vectors. */
#define X265GPU_OP_SAD  0
#define X265GPU_OP_SATD 1
#define X265GPU_OP_SA8D 2
typedef struct x265gpu_block_job
{
    int32_t x, y, w, h;
    int32_t qmvX, qmvY;
    int32_t op;
    int32_t reserved;
} x265gpu_block_job;
int x265gpu_block_compare(x265gpu_ctx* ctx, int32_t slotFenc, int32_t slotRef, int32_t n, const x265gpu_block_job* jobs,
                          int32_t* costs);
/* The bidirectional candidate of Search::puMotionEstimation (source/encoder/search.cpp:440-448, 482-486): each PU
 * predicted from two reference pictures at its own quarter-pel vector (Predict::predInterLumaPixel,
 * source/common/predict.cpp:279-300), the two predictions averaged (pixelavg_pp, (a + b + 1) >> 1) and compared
 * with the source PU by SATD (bufSATD).  costs[i] receives the SATD of job i; the zero-vector form of :482-486 is
 * the same job with both vectors zero. */
typedef struct x265gpu_bipred_job
{
    int32_t x, y, w, h;
    int32_t qmv0X, qmv0Y;      /* vector into slotRef0, quarter pixels */
    int32_t qmv1X, qmv1Y;      /* vector into slotRef1 */
} x265gpu_bipred_job;
int x265gpu_bipred_satd(x265gpu_ctx* ctx, int32_t slotFenc, int32_t slotRef0, int32_t slotRef1, int32_t n,
                        const x265gpu_bipred_job* jobs, int32_t* costs);
/* out receives w*h pixels, rows tight; (xFrac,yFrac) in quarter pixels, 0..3 each */
int x265gpu_interp_luma(x265gpu_ctx* ctx, int32_t slot, int32_t x, int32_t y, int32_t w, int32_t h, int32_t xFrac,
                        int32_t yFrac, void* out);

/* Copies a slot's padded full-resolution plane to the host (tests): `out` holds
 * stride * rows pixels, both reported through the pointers. */
int x265gpu_get_full_plane(x265gpu_ctx* ctx, int32_t slot, void* out, int32_t* stride, int32_t* rows,
                           int32_t* originX, int32_t* originY);

#ifdef __cplusplus
}
#endif
#endif /* X265GPU_H */
