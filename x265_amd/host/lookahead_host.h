// Host side of the lookahead above the x265gpu C-ABI, in C++ like the code it stands in for.
//
// Each class mirrors, name for name, the piece of x265's host code that sits directly above the
// offload boundary, with the pixel work replaced by calls into libx265gpu:
//
//   PreLookaheadGroup::processTasks     source/encoder/slicetype.cpp:1836-1861
//   CostEstimateGroup::singleCost / add / finishBatch / estimateFrameCost
//                                       source/encoder/slicetype.cpp:4148-4196, 4284-4383
//   LookaheadTLD::weightsAnalyse        source/encoder/slicetype.cpp:870-971
//   Lookahead::cuTree / estimateCUPropagate / cuTreeFinish
//                                       source/encoder/slicetype.cpp:3665-3874, 4016-4064
//   Lookahead::addPicture / flush / getDecidedPicture, slicetypeDecide, slicetypeAnalyse, scenecut,
//   scenecutInternal, slicetypePath, slicetypePathCost, vbvLookahead, vbvFrameCost, frameCostRecalculate
//                                       source/encoder/slicetype.cpp:1299-1431, 1982-2751, 2753-3321,
//                                       3484-3579, 4068-4145
//
// What stays here is what x265 keeps on the host: the cost cache and the bDoSearch decision from the
// 0x7FFF sentinel, request order, the B-frame bias and the intra penalty, the single-precision weight
// guess, the walk over the mini-GOPs.  Pixels never come here.  Every frame of the window owns one
// device slot.  Calls return X265GPU_OK or the library's negative status (nothing throws).
#ifndef X265GPU_LOOKAHEAD_HOST_H
#define X265GPU_LOOKAHEAD_HOST_H

#include "../../include/x265gpu.h"

#include <stdint.h>
#include <deque>
#include <memory>
#include <string>
#include <vector>

namespace x265gpu_host {

// X265_TYPE_* (source/x265.h)
enum { TYPE_AUTO = 0, TYPE_IDR = 1, TYPE_I = 2, TYPE_P = 3, TYPE_BREF = 4, TYPE_B = 5 };

constexpr int MAX_D = X265GPU_BFRAME_MAX + 2;
constexpr int LOOKAHEAD_MAX = 250;    // X265_LOOKAHEAD_MAX (source/x265.h:103)
enum { PRE_NONE = 0, PRE_P = 1, PRE_B = 2 };

// a 4:2:0 source picture; planes point at pixel (0,0), strides in pixels.  memKind says where the planes
// live (X265GPU_MEM_HOST / X265GPU_MEM_DEVICE); u and v may be null when no AQ pass reads them
struct Picture
{
    const void* y = nullptr;
    const void* u = nullptr;
    const void* v = nullptr;
    intptr_t strideY = 0, strideC = 0;
    int memKind = X265GPU_MEM_HOST;
};

// one computed (not cached) cost request, in issue order: what a test compares against the oracle's order
struct RequestTrace
{
    int32_t p0, p1, b;                // frame numbers
    int32_t doSearch[2];
    int32_t batchMode;
};

// What x265 keeps in Lowres besides pixels (source/common/lowres.h:164-258)
struct Lowres
{
    int slot = 0;
    int frameNum = 0;
    int sliceType = TYPE_AUTO;
    bool searched[2][MAX_D];          // lowresMvs[l][d][0].x != 0x7FFF
    // A search run ahead of the request that will ask for it (CostEstimateGroup::addPresearch): the device arrays
    // of (list, distance) already hold that request's vectors, while `searched` still says what x265's sentinel
    // would.  0: none; PRE_P: searched as a P request's list 0; PRE_B: as part of a bidirectional request.
    uint8_t presearched[2][MAX_D];
    bool presearchWeighted[MAX_D];    // weightsAnalyse's outcome for the presearched list-0 search
    // Costs measured ahead of x265's order over vectors that are already on the device (searched or presearched),
    // kept out of costEst until x265's order asks for them.  A search that rewrites one of the two vector sets drops
    // the entry.
    struct PreCost { int64_t costEst, costEstAq; int32_t intraMbs; bool valid; };
    PreCost preCost[MAX_D][MAX_D];
    int64_t costEst[MAX_D][MAX_D];    // -1: not measured (lowres.cpp:354-371)
    int64_t costEstAq[MAX_D][MAX_D];
    bool pending[MAX_D][MAX_D];       // queued in a batch that has not finished
    bool intraTotalsPending = false;  // costEst[0][0] / costEstAq[0][0] not read back yet (resolveIntraTotals)
    int intraMbs[MAX_D];
    uint64_t wp_sum = 0, wp_ssd = 0;  // plane 0, as calcAdaptiveQuantFrame leaves them (slicetype.cpp:663-686)
    bool weighted[MAX_D];             // weightedRef[d].isWeighted
    double weightedCostDelta[MAX_D];
    double* qpAqOffset = nullptr;     // caller-owned, one entry per 8x8 lowres block (16x16 source block); with qgSize 8
                                      // four times as many, one per 8x8 source block (lowres.cpp:92-95)
    double* qpCuTreeOffset = nullptr;
    int32_t* invQscaleFactor = nullptr;   // caller-owned, same shape; filled by the AQ form of processTasks
    std::vector<double> qpAqMotionOffset; // --aq-motion (lowres.cpp:114-115), sized on first use
    std::shared_ptr<x265gpu_hist_stats_t> hist;   // --hist-scenecut: collectPictureStatistics' results (lowres.h:245-253)
    bool bHistScenecutAnalyzed = false;
    double frameVariance = 0;             // --fades: calcAdaptiveQuantFrame's tail (slicetype.cpp:688-703)
    bool bIsFadeEnd = false;
    std::vector<uint32_t> blockVariance;  // its per-block output, sized on first use
    std::vector<int32_t> invQscaleFactor8x8;      // qgSize 8 (lowres.cpp:99-100)
    // rc.hevcAq: the PicQPAdaptationLayers of the enabled depths (lowres.cpp:128-155), the one of minAQDepth last
    struct AqLayer
    {
        int partWidth, partHeight, numPartInWidth, numPartInHeight;
        std::vector<double> dActivity, dQpOffset, dCuTreeOffset;
        double dAvgActivity;
    };
    std::vector<AqLayer> aqLayers;
    std::vector<double> dCuTreeOffset8x8;         // of that last layer, qgSize 8
    // Lowres::create's layer loop for a picture of the context's size
    void createAqLayers(const x265gpu_geometry& g, int maxCUSize, int qgSize);

    // what the slice-type decision keeps per frame (source/common/lowres.h:166-213)
    int sliceTypeReq = TYPE_AUTO;
    bool bScenecut = false, bKeyframe = false, bLastMiniGopBFrame = false;
    int leadingBframes = 0;
    // --temporal-layers 3..5: the picture's place in its hierarchical mini-GOP (Frame::m_gopOffset, m_gopId,
    // m_tempLayer; source/common/frame.h:96, 167-168)
    int gopOffset = 0, gopId = 0, tempLayer = 0;
    int indB = 0;
    std::vector<int64_t> plannedSatd;     // [LOOKAHEAD_MAX + 1] once the VBV lookahead has run
    std::vector<int32_t> plannedType;
    std::vector<int32_t> recalcRowSatds;  // row sums of frameCostRecalculate (the reference overwrites rowSatds in place)
    Picture pic;                          // Frame::m_fencPic: read when the frame is pre-analysed
    bool lowresInit = false;              // Frame::m_lowresInit

    Lowres() { reset(); }
    void reset();                     // the host half of Lowres::init
};

class LookaheadTLD
{
public:
    explicit LookaheadTLD(x265gpu_ctx* ctx) : m_ctx(ctx) { x265gpu_get_geometry(ctx, &m_geom); }
    int weightsAnalyse(Lowres& fenc, Lowres& ref, int bitDepth);
    // the same for n independent (fenc, ref) pairs: their two device passes each run as one batch
    int weightsAnalyse(int n, Lowres* const* fencs, Lowres* const* refs, int bitDepth);

private:
    x265gpu_ctx* m_ctx;
    x265gpu_geometry m_geom;
};

// costEst[0][0] / costEstAq[0][0] of the frames whose pre-lookahead left them on the device (the AQ form of
// PreLookaheadGroup::processTasks returns while the host threads that map the AQ offsets are still running, so
// that the mapping overlaps the first batch of searches); one transfer for all of them.  Also ends that overlap:
// after it qpAqOffset / qpCuTreeOffset / invQscaleFactor of every frame are complete.
int resolveIntraTotals(x265gpu_ctx* ctx, Lowres* const* frames, int n);

class PreLookaheadGroup
{
public:
    explicit PreLookaheadGroup(x265gpu_ctx* ctx) : m_ctx(ctx) {}
    // Lowres::init + lowresIntraEstimate for n frames; lumas are host planes (pixel (0,0), stride in pixels),
    // invQscale[i] the frame's AQ factors or null
    int processTasks(int n, Lowres* const* frames, const void* const* lumas, const intptr_t* strides,
                     const int32_t* const* invQscale);
    // The whole of PreLookaheadGroup::processTasks (slicetype.cpp:1836-1861): Lowres::init,
    // calcAdaptiveQuantFrame (block energies on the device, offsets into the frame's qpAqOffset /
    // qpCuTreeOffset / invQscaleFactor arrays, wp_sum / wp_ssd of plane 0 into the frame) and
    // lowresIntraEstimate, for 4:2:0 host pictures.
    typedef x265gpu_host::Picture Picture;
    int processTasks(int n, Lowres* const* frames, const Picture* pics, const x265gpu_aq_params& aq,
                     const float* const* quantOffsets);

private:
    x265gpu_ctx* m_ctx;
};

class CostEstimateGroup
{
public:
    // frames: the window, indexed like x265's frames[]; bitDepth only matters with weightp
    CostEstimateGroup(x265gpu_ctx* ctx, Lowres* const* frames, bool bEnableWeightedPred = false, int bitDepth = 8,
                      std::vector<RequestTrace>* trace = nullptr);

    int64_t singleCost(int p0, int p1, int b, bool bIntraPenalty = false);
    void add(int p0, int p1, int b);
    // Queues, into the current batch, the motion search(es) that a later singleCost(p0, p1, b) outside batch mode
    // would run for the lists in `lists` (bit 0: list 0, bit 1: list 1), and nothing else.  When that request
    // comes, it finds its vectors on the device and only measures costs.  A search's outcome depends on the frame
    // pair, the list, the distance, whether the request is bidirectional and whether it runs in cooperative slices;
    // a presearched list 0 is therefore only used by a later request of the same kind (p1 > b or not).
    void addPresearch(int p0, int p1, int b, int lists);
    void addPrecost(int p0, int p1, int b);
    void finishBatch();
    int status() const { return m_status; }      // first failure of any call, or X265GPU_OK
    int presearchHits() const { return m_presearchHits; }
    int precostHits() const { return m_precostHits; }

private:
    struct Pending { x265gpu_request req; int b, d0, d1; int p0; bool deferWeights, presearch, precost; };
    void analyseDeferredWeights();
    bool makeRequest(int p0, int p1, int b, Pending& out);
    bool analyseWeights(Lowres& fenc, Lowres& ref);
    void collect(const Pending& p, const x265gpu_frame_cost& fc);

    x265gpu_ctx* m_ctx;
    Lowres* const* m_frames;
    LookaheadTLD m_tld;
    bool m_weightp;
    int m_bitDepth;
    bool m_batchMode = false;
    int m_ncu;                                    // LookaheadTLD::ncu (slicetype.cpp:1065-1066)
    int m_numCoopSlices = 1;
    int m_status = X265GPU_OK;
    std::vector<Pending> m_batch;
    std::vector<RequestTrace>* m_trace;
    int m_presearchHits = 0;
    int m_precostHits = 0;
};

// The subset of x265_param that cuTree reads
struct CuTreeParam
{
    int fpsNum = 30, fpsDenom = 1;
    bool bBPyramid = false;
    bool bEnableWeightedBiPred = false;
    int lookaheadDepth = 20;
    int vbvBufferSize = 0;
    double qCompress = 0.6;
    bool bEnableWeightedPred = false;             // cuTree's own requests analyse weights like any other
    int bitDepth = 8;
    int qgSize = 32;                              // rc.qgSize: 8 selects the qgSize 8 loop of cuTreeFinish (:4030-4049)
    bool bHevcAq = false;                         // rc.hevcAq: cuTreeFinish is computeCUTreeQpOffset over the frame's layers
    std::vector<RequestTrace>* trace = nullptr;
};

class CuTree
{
public:
    CuTree(x265gpu_ctx* ctx, const CuTreeParam& p);
    int cuTree(Lowres* const* frames, int numframes, bool bIntra);

private:
    void estimateCUPropagate(Lowres* const* frames, double averageDuration, int p0, int p1, int b, int referenced);
    void cuTreeFinish(Lowres& frame, double averageDuration, int ref0Distance);
    void note(int rc) { if (rc != X265GPU_OK && m_status == X265GPU_OK) m_status = rc; }

    x265gpu_ctx* m_ctx;
    CuTreeParam m_p;
    double m_frameDuration;
    double m_cuTreeStrength;                      // slicetype.cpp:1075
    int m_status = X265GPU_OK;
};

// The subset of x265_param the slice-type decision reads (fields named as there; values as
// Encoder::configure leaves them: keyframeMin never 0, scenecutBias already divided by 100).
struct LookaheadParam
{
    int bframes = 4;
    int lookaheadDepth = 20;
    int bFrameAdaptive = 2;                       // X265_B_ADAPT_NONE / FAST / TRELLIS
    bool bBPyramid = true;
    int scenecutThreshold = 40;
    double scenecutBias = 0.05;
    int keyframeMin = 25, keyframeMax = 250;
    bool bOpenGOP = true;
    bool bIntraRefresh = false;
    int radl = 0;
    int gopLookahead = 0;
    bool cuTree = true;                           // rc.cuTree
    int bFrameBias = 0;
    int vbvBufferSize = 0, vbvMaxBitrate = 0;
    bool bCqp = false;                            // rc.rateControlMode == X265_RC_CQP
    int bEnableTemporalSubLayers = 0;             // 0, 2, or 3..5 with bframes 3 / 7 / 15 and no B-adapt, as
                                                  // Encoder::configure leaves them (encoder.cpp:4303-4332)
    int maxNumReferences = 3;
    bool bEnableWeightedPred = true;
    bool bEnableWeightedBiPred = false;
    int fpsNum = 30, fpsDenom = 1;
    double qCompress = 0.6;
    int bitDepth = 8;
    bool bPresearch = true;                       // with batches: also queue, in batch 1, the searches of the window's first
                                                  // and last frames that x265 leaves to single requests (results unchanged)
    int lookaheadSlices = 0;                      // as the context was opened with; only read to know whether single requests
                                                  // run in cooperative slices
    int poolWorkers = 0;                          // ThreadPool::m_numWorkers of the lookahead's pool; 0: no pool, so
                                                  // requests are issued one by one in the order the decision needs them;
                                                  // > 0 with trellis B-adapt: the two batches of slicetypeAnalyse
    bool bHistBasedSceneCut = false;              // --hist-scenecut: scene cuts from histogram differences of the 4x4 segments
    bool bAQMotion = false;                       // --aq-motion: offsets raised where a block moves unlike the rest of its frame
    int maxCUSize = 64;                           // read for the layers of rc.hevcAq
    bool bEnableFades = false;                    // --fades: the picture that ends a fade-in of a second or more becomes a keyframe
    bool bAdaptiveQuant = true;                   // Lookahead::m_bAdaptiveQuant: pre-analysis runs calcAdaptiveQuantFrame
    x265gpu_aq_params aq;                         // its parameters (aqMode, strength, qgSize, ...)
    LookaheadParam();
};

// Lookahead: the queue interface and the serial decision logic, over device-side cost estimation.
// Frames are the caller's Lowres objects (x265: Frame::m_lowres), each with a device slot of its own for as
// long as it is inside the lookahead or referenced as the last non-B frame, and with qpAqOffset /
// qpCuTreeOffset / invQscaleFactor arrays when AQ or cuTree is on.
class Lookahead
{
public:
    Lookahead(x265gpu_ctx* ctx, const LookaheadParam& param);
    int  unsupported() const;                     // X265GPU_ERR_UNSUPPORTED when param asks for what is not covered
    void addPicture(Lowres& frame, int sliceType);
    void flush();
    Lowres* getDecidedPicture();
    int64_t frameCostRecalculate(Lowres* const* frames, int p0, int p1, int b);
    int status() const { return m_status; }
    const std::vector<RequestTrace>& trace() const { return m_trace; }
    int decideCalls() const { return m_decideCalls; }
    int presearchHits() const { return m_presearchHits; }
    int precostHits() const { return m_precostHits; }
    // wall-clock seconds spent inside slicetypeDecide, split by what was running (profiling aid)
    enum { T_PRE, T_BATCH1, T_BATCH2, T_SINGLE, T_CUTREE, T_DECIDE, T_COUNT };
    const double* seconds() const { return m_seconds; }

private:
    double m_seconds[T_COUNT] = { 0, 0, 0, 0, 0, 0 };
    void slicetypeDecide();
    void slicetypeAnalyse(Lowres** frames, bool bKeyframe);
    bool scenecut(Lowres** frames, int p0, int p1, bool bRealScenecut, int numFrames);
    bool scenecutInternal(Lowres** frames, int p0, int p1, bool bRealScenecut);
    void slicetypePath(Lowres** frames, int length, std::vector<std::string>& bestPaths);
    int64_t slicetypePathCost(Lowres** frames, const std::string& path, int64_t threshold);
    void vbvLookahead(Lowres** frames, int numFrames, bool keyframe);
    int64_t vbvFrameCost(Lowres** frames, int p0, int p1, int b);
    void placeBref(Lowres** list, int start, int end, int num, int& brefs);
    void finishMiniGop(Lowres** frames, Lowres** list, int bframes, int& brefs, int& maxSearch);
    void finishTemporalMiniGop(Lowres** frames, Lowres** list, int bframes, int& brefs, int& maxSearch);
    void subGopCosts(Lowres** frames, int first, int nonB);
    void dequeue(int count, int& maxSearch);
    void compCostBref(Lowres** frames, int start, int end, int num);
    void cuTree(Lowres** frames, int numframes, bool bIntra);
    void aqMotion(Lowres** frames, bool bIntra);
    bool histBasedScenecut(Lowres** frames, int p0, int p1, int numFrames);
    bool detectHistBasedSceneChange(Lowres** frames, int p0, int p1, int p2);
    void calcMotionAdaptiveQuantFrame(Lowres** frames, int p0, int p1, int b);
    int64_t cost(Lowres** frames, int p0, int p1, int b, bool bIntraPenalty = false);
    void note(int rc) { if (rc != X265GPU_OK && m_status == X265GPU_OK) m_status = rc; }

    x265gpu_ctx* m_ctx;
    LookaheadParam m_param;
    x265gpu_geometry m_geom;
    int m_8x8Blocks;
    Lowres* m_lastNonB = nullptr;
    int m_lastKeyframe;
    int m_gopId = 0;                  // --temporal-layers 3..5: which of the 4 / 8 / 16 picture orders a full mini-GOP takes
    bool m_isSceneTransition = false, m_extendGopBoundary = false;
    bool m_bBatchMotionSearch, m_bBatchFrameCosts;
    // --hist-scenecut (slicetype.cpp:1143-1174): running averages of the histogram differences, [plane][segment]
    uint32_t m_accHistDiffRunningAvg[3][4][4] = {};
    bool m_resetRunningAvg = true;
    uint32_t m_segmentCountThreshold = (uint32_t)(((float)((4 * 4) * 50) / 100) + 0.5);
    // --fades (slicetype.cpp:1067-1069, slicetype.h:196-200)
    double m_frameVariance[X265GPU_BFRAME_MAX + 4];
    bool m_isFadeIn = false;
    uint64_t m_fadeCount = 0;
    int m_fadeStart = -1;
    void detectFades(Lowres** list);
    bool m_filled = false;
    size_t m_fullQueueSize;
    int m_inputCount = 0;
    std::deque<Lowres*> m_inputQueue, m_outputQueue;
    std::vector<RequestTrace> m_trace;
    int m_status = X265GPU_OK;
    int m_decideCalls = 0;
    int m_presearchHits = 0;
    int m_precostHits = 0;
    void presearchEdges(CostEstimateGroup& est, Lowres** frames, int numFrames);
    void precostEdges(CostEstimateGroup& est, int numFrames);
};

} // namespace x265gpu_host

// ---- a plain-C face of the same classes, for tests and non-C++ callers ----
extern "C" {
typedef struct xgh_window xgh_window;
xgh_window* xgh_window_create(x265gpu_ctx* ctx, int32_t nframes, int32_t bEnableWeightedPred, int32_t bitDepth);
void xgh_new_group(xgh_window* w, int32_t bEnableWeightedPred, int32_t bitDepth);
void xgh_window_destroy(xgh_window* w);
int xgh_pre_lookahead(xgh_window* w, int32_t n, const int32_t* frames, const void* const* lumas, const intptr_t* strides,
                      const int32_t* const* invQscale);
/* the AQ form: 4:2:0 pictures; invQscale[i] receives the frame's scale factors (qgSize >= 16), the offsets go to the
 * arrays given with xgh_set_qp_offsets, wp_sum / wp_ssd of the luma plane are kept for weightsAnalyse */
int xgh_pre_lookahead_aq(xgh_window* w, int32_t n, const int32_t* frames, const void* const* y, const void* const* u,
                         const void* const* v, const intptr_t* strideY, const intptr_t* strideC,
                         const x265gpu_aq_params* aq, int32_t* const* invQscale, int32_t memKind);
void xgh_get_wp(const xgh_window* w, int32_t frame, uint64_t* sum, uint64_t* ssd);
void xgh_set_wp(xgh_window* w, int32_t frame, uint64_t sum, uint64_t ssd);
void xgh_set_slice_type(xgh_window* w, int32_t frame, int32_t sliceType);
void xgh_set_qp_offsets(xgh_window* w, int32_t frame, double* qpAqOffset, double* qpCuTreeOffset);
int64_t xgh_single_cost(xgh_window* w, int32_t p0, int32_t p1, int32_t b, int32_t bIntraPenalty);
void xgh_add(xgh_window* w, int32_t p0, int32_t p1, int32_t b);
void xgh_add_many(xgh_window* w, int32_t n, const int32_t* p0p1b);   /* n x { p0, p1, b } */
// finishes what xgh_pre_lookahead_aq left running: after it the frames' AQ arrays and intra totals are complete
int xgh_resolve(xgh_window* w);
int xgh_finish_batch(xgh_window* w);
int64_t xgh_cost_est(const xgh_window* w, int32_t b, int32_t d0, int32_t d1);
/* { costEst, costEstAq, intraMbs } of n requests { p0, p1, b } into out[3n]: what a multi-GPU lookahead gathers */
void xgh_collect_costs(const xgh_window* w, int32_t n, const int32_t* p0p1b, int64_t* out);
int32_t xgh_intra_mbs(const xgh_window* w, int32_t b, int32_t d0);
int32_t xgh_is_weighted(const xgh_window* w, int32_t b, int32_t d0);
int xgh_cu_tree(xgh_window* w, int32_t fpsNum, int32_t fpsDenom, int32_t bBPyramid, int32_t bEnableWeightedBiPred,
                int32_t lookaheadDepth, int32_t vbvBufferSize, double qCompress, int32_t numframes, int32_t bIntra);
int xgh_status(const xgh_window* w);
/* BitCost::setQP's table for `lambda` (source/encoder/bitcost.cpp:31-56): out holds 4 * 32768 + 1 entries, entry
 * 2 * 32768 is MVD 0.  For callers that have no BitCost of their own; x265 passes BitCost::m_cost instead. */
void xgh_bitcost_table(double lambda, uint16_t* out);

/* the whole lookahead: queue in, decided frames out.  Every frame gets a device slot of its own (the context
 * needs as many slots as frames are alive at once: here, as frames are added in total). */
typedef struct xgh_lookahead xgh_lookahead;
typedef struct xgh_lookahead_param
{
    int32_t bframes, lookaheadDepth, bFrameAdaptive, bBPyramid, scenecutThreshold;
    double  scenecutBias;
    int32_t keyframeMin, keyframeMax, bOpenGOP, bIntraRefresh, radl, gopLookahead, cuTree, bFrameBias;
    int32_t vbvBufferSize, vbvMaxBitrate, bCqp, bEnableTemporalSubLayers, maxNumReferences;
    int32_t bEnableWeightedPred, bEnableWeightedBiPred, fpsNum, fpsDenom;
    double  qCompress;
    int32_t bitDepth, poolWorkers, bAdaptiveQuant, aqMode;
    double  aqStrength;
    int32_t bPresearch, lookaheadSlices;
    int32_t bAQMotion, bHistBasedSceneCut;
    int32_t bEnableFades;
    int32_t qgSize, bHevcAq;      /* rc.qgSize (0 = 32), rc.hevcAq */
    double  qpAdaptationRange;    /* rc.qpAdaptationRange */
    int32_t maxCUSize;            /* 0 = 64 */
} xgh_lookahead_param;
typedef struct xgh_decided
{
    int32_t frame;                /* index in order of xgh_lookahead_add_picture */
    int32_t sliceType, bScenecut, bKeyframe, bLastMiniGopBFrame, leadingBframes;
    int32_t bIsFadeEnd;
    int32_t gopOffset, gopId, tempLayer;
} xgh_decided;
xgh_lookahead* xgh_lookahead_create(x265gpu_ctx* ctx, const xgh_lookahead_param* p);
void xgh_lookahead_destroy(xgh_lookahead* la);
/* planes stay the caller's and must live until the frame comes out; returns the frame's index or a negative status */
int xgh_lookahead_add_picture(xgh_lookahead* la, const void* y, const void* u, const void* v, intptr_t strideY,
                              intptr_t strideC, int32_t memKind, int32_t sliceType, int32_t sliceTypeReq);
void xgh_lookahead_flush(xgh_lookahead* la);
/* 1: *out filled; 0: nothing decided yet (or the lookahead is empty); < 0: status */
int xgh_lookahead_get_decided(xgh_lookahead* la, xgh_decided* out);
int xgh_lookahead_trace(const xgh_lookahead* la, const int32_t** out);    /* 6 ints per request, as RequestTrace */
int64_t xgh_lookahead_cost_est(const xgh_lookahead* la, int32_t frame, int32_t d0, int32_t d1);
int64_t xgh_lookahead_cost_est_aq(const xgh_lookahead* la, int32_t frame, int32_t d0, int32_t d1);
int32_t xgh_lookahead_intra_mbs(const xgh_lookahead* la, int32_t frame, int32_t d0);
int32_t xgh_lookahead_slot(const xgh_lookahead* la, int32_t frame);
const double* xgh_lookahead_qp_offsets(const xgh_lookahead* la, int32_t frame, int32_t which /*0 aq, 1 cuTree*/);
const int32_t* xgh_lookahead_inv_qscale(const xgh_lookahead* la, int32_t frame);
/* rc.hevcAq: layer `layer` of the frame (0 .. count - 1, the one of minAQDepth last); which: 0 dQpOffset, 1 dCuTreeOffset;
 * returns the layer count, *out / *n the array and its length */
int xgh_lookahead_aq_layer(const xgh_lookahead* la, int32_t frame, int32_t layer, int32_t which, const double** out, int32_t* n);
/* VBV lookahead results: counts of valid entries come back through n */
const int64_t* xgh_lookahead_planned_satd(const xgh_lookahead* la, int32_t frame, const int32_t** types, int32_t* indB);
const int32_t* xgh_lookahead_recalc_row_satds(const xgh_lookahead* la, int32_t frame);
int xgh_lookahead_decide_calls(const xgh_lookahead* la);
int xgh_lookahead_presearch_hits(const xgh_lookahead* la);
int xgh_lookahead_precost_hits(const xgh_lookahead* la);
/* seconds inside slicetypeDecide: { pre-analysis, batch 1, batch 2, single requests, cuTree, whole decide calls } */
void xgh_lookahead_seconds(const xgh_lookahead* la, double out[6]);
}

#endif
