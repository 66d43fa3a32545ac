// Host side of --threaded-me above the x265gpu C-ABI: the CTU job generator.
//
// x265 derives the motion vectors of every enabled prediction-unit shape of a CTU ahead of analysis
// (ThreadedME::findJob -> Analysis::deriveMVsForCTU, source/encoder/threadedme.cpp:207-261,
// source/encoder/analysis.cpp:161-306), one Search::puMotionEstimation call per shape and CU
// (source/encoder/search.cpp:226-560).  A CTU reads nothing of its neighbour CTUs on that path: the AMVP
// neighbours are entries of the same CTU's result table (search.cpp:287-306), so every CTU of a picture may
// run at once, and inside a CTU the calls fall into dependency levels by the table entries they read.
// ThreadedME below walks those levels for a batch of CTU rows: per level one x265gpu_block_compare call
// (Search::selectMVP's candidate SADs) and one x265gpu_pu_motion_estimate call per reference picture, with
// the predictor logic (CUData::getPMV, Search::setSearchRange, getLowresMV, updateMVP, checkBestMVP and the
// bit accounting) kept here, as x265 keeps it on the host.  Pixels never come here.
//
// Covered: P and B slices (any number of references per list; in a B slice the bidirectional candidate of
// search.cpp:415-511 through x265gpu_bipred_satd, and the list-selection bits that depend on what the
// partition's first part chose), every shape of g_puLookup (source/encoder/threadedme.h:67-92) under
// bEnableAMP / bEnableRectInter, one slice, no frame-parallel lag window beyond refLagPixels.  With delta QP on,
// every CU searches under its own QP's lambda (Analysis::setLambdaFromQP, analysis.cpp:188, 243, 259, 274): the
// per-CU QPs are an input (setCuQp; x265 derives them from rate-control state in calculateQpforCuSize), the
// MV-cost tables of the QPs in use live on the device side by side.  The temporal candidate
// (neighbours[MD_COLLOCATED]) and CUData::getMedianColMV read the reference picture's coded CUs, which
// only the encoder has: both are optional inputs (setTemporalCandidates, medianMv); without them the temporal
// candidate is unavailable (--no-temporal-mvp).
#ifndef X265GPU_THREADED_ME_HOST_H
#define X265GPU_THREADED_ME_HOST_H

#include "../../include/x265gpu.h"

#include <stdint.h>
#include <vector>

namespace x265gpu_host {

constexpr int MAX_NUM_PUS_PER_CTU = 593;   // source/common/common.h:355
constexpr int MAX_NUM_PU_SIZES = 24;       // source/common/common.h:356
constexpr int REF_NOT_VALID = -1;

// PartSize (source/common/common.h:266-279)
enum { SIZE_2Nx2N = 0, SIZE_2NxN, SIZE_Nx2N, SIZE_NxN, SIZE_2NxnU, SIZE_2NxnD, SIZE_nLx2N, SIZE_nRx2N };
// MVP_DIR (source/common/cudata.h:67-75)
enum { MD_LEFT = 0, MD_ABOVE, MD_ABOVE_RIGHT, MD_BELOW_LEFT, MD_ABOVE_LEFT, MD_COLLOCATED };

struct MV { int32_t x, y; };

// MEData (source/encoder/threadedme.h:122-130)
struct MEData
{
    MV       mv[2];
    MV       mvp[2];
    uint32_t mvCost[2];
    int32_t  ref[2];
    int32_t  bits;
    uint32_t cost;
};

struct TmeParam
{
    int32_t searchMethod;        // param->searchMethod (X265GPU_*_SEARCH)
    int32_t searchRange;         // param->searchRange
    int32_t subpelRefine;        // param->subpelRefine
    int32_t bEnableAMP;          // param->bEnableAMP
    int32_t bEnableRectInter;    // param->bEnableRectInter
    int32_t maxCUSize;           // param->maxCUSize, equal to the context's CTU size
    int32_t minCUSize;           // param->minCUSize
    int32_t bframes;             // param->bframes: reach of the lookahead vectors (Search::getLowresMV)
    int32_t picWidth;            // sps->picWidthInLumaSamples / picHeightInLumaSamples: the source size
    int32_t picHeight;           //   padded to a multiple of minCUSize
    int32_t refLagPixels;        // Search::m_refLagPixels (search.cpp:96): sourceHeight without frame threads
    double  lambda;              // x265_lambda_tab[slice QP]: MV cost table and RDCost::getCost
    int32_t sourceWidth;         // param->sourceWidth / sourceHeight (the zero-vector test of the bidirectional
    int32_t sourceHeight;        //   candidate, search.cpp:458); 0: picWidth / picHeight
};

// one entry of slice->m_refPOCList[list] / m_mref[list]
struct TmeRef
{
    int32_t slot;                // device slot holding the reference picture (its full-resolution plane)
    int32_t poc;
    int32_t isIntraSlice;        // the reference picture was coded as an I slice (search.cpp:315)
    const MEData* ctuMV;         // that picture's own result table (all CTUs), or NULL
    const int32_t* lowresMvs;    // m_frame->m_lowres.lowresMvs[list][|poc - ref poc|] as x,y pairs, or NULL when that
                                 //   distance was never searched (the 0x7FFF sentinel of search.cpp:2331)
};

// neighbours[MD_COLLOCATED] of one PU and list as CUData::getNeighbourMV leaves it (source/common/cudata.cpp:1987-2080):
// whether a co-located vector exists (refIdx != -1), the vector, and the two POCs CUData::getPMV scales it by
// (the co-located picture's own POC and the POC that vector refers to, cudata.cpp:1959-1977)
struct TmeTemporal
{
    int32_t valid;
    MV      mv;
    int32_t colPoc, colRefPoc;
};

// One prediction-unit search call of Analysis::computeMVForPUs: a CU and one g_puLookup entry.
struct PuTask
{
    int cuX, cuY, cuSize;        // CU inside the CTU
    int partSize, numPart;
    int finalIdx, puOffset, areaId;
    int nb[5];                   // neighborIdx[], already -1 where the entry is written later in call order
    int level;                   // dependency level: every entry it reads belongs to a lower level
    struct { int x, y, w, h; } pu[2];   // relative to the CTU
};

class ThreadedME
{
public:
    ThreadedME(x265gpu_ctx* ctx, const TmeParam& p);
    ~ThreadedME();
    ThreadedME(const ThreadedME&) = delete;
    ThreadedME& operator=(const ThreadedME&) = delete;

    // Why this configuration cannot be run, or NULL
    const char* unsupported() const { return m_unsupported; }

    // deriveMVsForCTU for every CTU of rows [rowBegin, rowEnd) of the P picture in slotFenc.
    // medianMv: NULL, or per CTU of the picture and reference {valid, x, y} (CUData::getMedianColMV's result).
    // out: the picture's table, numCtuInPicture x MAX_NUM_PUS_PER_CTU entries; the rows' entries are reset
    // (FrameData::reinit, source/common/framedata.cpp:93-102) and filled.
    int estimateRows(int slotFenc, int poc, int numRefs, const TmeRef* refs, int rowBegin, int rowEnd,
                     const int32_t* medianMv, MEData* out);
    // The same for a B picture: refs holds list 0's numRefsL0 references, then list 1's numRefsL1 (medianMv in
    // that order too).  numRefsL1 = 0 is the P form above.
    int estimateRowsB(int slotFenc, int poc, int numRefsL0, int numRefsL1, const TmeRef* refs, int rowBegin, int rowEnd,
                      const int32_t* medianMv, MEData* out);

    // Per-CU QPs of the picture for the calls that follow (NULL: the one lambda of TmeParam again).  qp holds, per
    // CTU of the picture, cusPerCtu() entries: the CUs of depth 0, then those of depth 1 in raster order, and so on
    // down to minCUSize - what Analysis::computeMVForPUs would pass to setLambdaFromQP for that CU (a CU deeper than
    // pps->maxCuDQPDepth inherits its parent's, analysis.cpp:183-190).  The area searches take areaQp's five QPs per
    // CTU (the CTU, then its quadrants: analysis.cpp:259, 274 compute them whatever maxCuDQPDepth is), or, with
    // areaQp NULL, those of the CTU's depth-0 and depth-1 CUs.  lambdaByQp: x265_lambda_tab
    // (source/common/constants.cpp), numQp entries, copied.  The QP arrays stay the caller's and must outlive the calls.
    int setCuQp(const int8_t* qp, const int8_t* areaQp, const double* lambdaByQp, int numQp);
    int cusPerCtu() const;
    // The temporal AMVP candidates of the picture (sps->bTemporalMVPEnabled), or NULL: per CTU of the picture
    // MAX_NUM_PUS_PER_CTU table positions x 2 lists, indexed like the result table.
    void setTemporalCandidates(const TmeTemporal* t) { m_temporal = t; }

    const std::vector<PuTask>& tasks() const { return m_tasks; }
    int numLevels() const { return m_numLevels; }
    int numCuInWidth() const { return m_numCuInWidth; }
    int numCuInHeight() const { return m_numCuInHeight; }
    int64_t searches() const { return m_searches; }        // PU searches issued so far (area searches included)
    int64_t deviceCalls() const { return m_deviceCalls; }
    // wall-clock seconds so far: [1] inside device calls (upload + kernels + download), [2] inside estimateRows
    const double* seconds() const { return m_seconds; }

private:
    void buildTasks();
    void addCostSet(double lambda);
    void visit(int cuX, int cuY, int cuSize, std::vector<int>& writer);
    uint32_t bitcost(MV mv, MV mvp) const;
    uint32_t mvcost(MV mv, MV mvp, int set) const;
    uint32_t getCost(uint32_t bits, int set) const { return (uint32_t)(((uint64_t)bits * m_costSets[set].lambdaFix8 + 128) >> 8); }
    int costSetOfQp(int qp);                         // builds the QP's table on first use; -1: no such QP
    int cuIndex(const PuTask& t) const;
    void clipMv(int cuPelX, int cuPelY, MV& mv) const;
    void setSearchRange(int cuPelX, int cuPelY, MV mvp, int merange, MV& mvmin, MV& mvmax) const;

    x265gpu_ctx* m_ctx;
    TmeParam m_p;
    const char* m_unsupported;
    int m_numCuInWidth, m_numCuInHeight;
    int m_puStartIdx[2 * 64 + 1][8];                // g_puStartIdx
    std::vector<PuTask> m_tasks;                    // call order of computeMVForPUs
    int m_numLevels;
    std::vector<float> m_bitsizes;                  // BitCost::s_bitsizes, centred
    // BitCost::m_cost (centred) and RDCost::m_lambda per lambda in use: set 0 is TmeParam::lambda, the others the
    // QPs that setCuQp's array has named so far; a job's set is also its MV-cost table's index on the device
    struct CostSet { std::vector<uint16_t> table; uint64_t lambdaFix8; };
    std::vector<CostSet> m_costSets;
    std::vector<int> m_setOfQp;
    std::vector<double> m_lambdaByQp;
    const int8_t* m_cuQp;
    const int8_t* m_areaQp;
    const TmeTemporal* m_temporal;
    size_t m_setsOnDevice;
    int64_t m_searches, m_deviceCalls;
    double m_seconds[3];
    class Pool;
    x265gpu_pu_job* m_jobs;                         // page-locked job / result arrays of the level batches
    x265gpu_pu_result* m_results;
    size_t m_jobCap;
    Pool* m_pool;                                   // host threads for the per-PU loops (X265GPU_HOST_THREADS, default up to 16)
};

}  // namespace x265gpu_host

extern "C" {

typedef x265gpu_host::MEData xgh_me_data;
typedef x265gpu_host::TmeParam xgh_tme_param;
typedef x265gpu_host::TmeRef xgh_tme_ref;
typedef x265gpu_host::TmeTemporal xgh_tme_temporal;
typedef struct xgh_tme xgh_tme;

xgh_tme* xgh_tme_create(x265gpu_ctx* ctx, const xgh_tme_param* p);
void xgh_tme_destroy(xgh_tme* t);
const char* xgh_tme_unsupported(const xgh_tme* t);
int xgh_tme_estimate_rows(xgh_tme* t, int32_t slotFenc, int32_t poc, int32_t numRefs, const xgh_tme_ref* refs,
                          int32_t rowBegin, int32_t rowEnd, const int32_t* medianMv, xgh_me_data* out);
// B picture: refs = list 0's numRefsL0 references, then list 1's numRefsL1
int xgh_tme_estimate_rows_b(xgh_tme* t, int32_t slotFenc, int32_t poc, int32_t numRefsL0, int32_t numRefsL1,
                            const xgh_tme_ref* refs, int32_t rowBegin, int32_t rowEnd, const int32_t* medianMv,
                            xgh_me_data* out);
// per-CU QPs / temporal candidates of the picture for the calls that follow (see ThreadedME::setCuQp,
// setTemporalCandidates); NULL switches them off again
int xgh_tme_set_cu_qp(xgh_tme* t, const int8_t* qp, const int8_t* areaQp, const double* lambdaByQp, int32_t numQp);
int xgh_tme_cus_per_ctu(const xgh_tme* t);
void xgh_tme_set_temporal(xgh_tme* t, const xgh_tme_temporal* candidates);
// seconds so far: {host logic, inside device calls (upload + kernels + download), whole estimateRows calls}
void xgh_tme_seconds(const xgh_tme* t, double seconds[3]);
// counts: {tasks per CTU, dependency levels, CTUs per row, CTU rows, searches so far, device calls so far}
void xgh_tme_counts(const xgh_tme* t, int64_t counts[6]);
}

#endif
