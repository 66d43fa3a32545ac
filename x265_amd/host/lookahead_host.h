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
//
// What stays here is what x265 keeps on the host: the cost cache and the bDoSearch decision from the
// 0x7FFF sentinel, request order, the B-frame bias and the intra penalty, the single-precision weight
// guess, the walk over the mini-GOPs.  Pixels never come here.  Every frame of the window owns one
// device slot.  Calls return X265GPU_OK or the library's negative status (nothing throws).
#ifndef X265GPU_LOOKAHEAD_HOST_H
#define X265GPU_LOOKAHEAD_HOST_H

#include "../../include/x265gpu.h"

#include <stdint.h>
#include <vector>

namespace x265gpu_host {

// X265_TYPE_* (source/x265.h)
enum { TYPE_AUTO = 0, TYPE_IDR = 1, TYPE_I = 2, TYPE_P = 3, TYPE_BREF = 4, TYPE_B = 5 };

constexpr int MAX_D = X265GPU_BFRAME_MAX + 2;

// What x265 keeps in Lowres besides pixels (source/common/lowres.h:164-258)
struct Lowres
{
    int slot = 0;
    int frameNum = 0;
    int sliceType = TYPE_AUTO;
    bool searched[2][MAX_D];          // lowresMvs[l][d][0].x != 0x7FFF
    int64_t costEst[MAX_D][MAX_D];    // -1: not measured (lowres.cpp:354-371)
    int64_t costEstAq[MAX_D][MAX_D];
    bool pending[MAX_D][MAX_D];       // queued in a batch that has not finished
    int intraMbs[MAX_D];
    uint64_t wp_sum = 0, wp_ssd = 0;  // plane 0, as calcAdaptiveQuantFrame leaves them (slicetype.cpp:663-686)
    bool weighted[MAX_D];             // weightedRef[d].isWeighted
    double weightedCostDelta[MAX_D];
    double* qpAqOffset = nullptr;     // caller-owned, one entry per 8x8 lowres block (16x16 source block)
    double* qpCuTreeOffset = nullptr;
    int32_t* invQscaleFactor = nullptr;   // caller-owned, same shape; filled by the AQ form of processTasks

    Lowres() { reset(); }
    void reset();                     // the host half of Lowres::init
};

class LookaheadTLD
{
public:
    explicit LookaheadTLD(x265gpu_ctx* ctx) : m_ctx(ctx) { x265gpu_get_geometry(ctx, &m_geom); }
    int weightsAnalyse(Lowres& fenc, Lowres& ref, int bitDepth);

private:
    x265gpu_ctx* m_ctx;
    x265gpu_geometry m_geom;
};

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
    struct Picture { const void* y; const void* u; const void* v; intptr_t strideY, strideC; };
    int processTasks(int n, Lowres* const* frames, const Picture* pics, const x265gpu_aq_params& aq,
                     const float* const* quantOffsets);

private:
    x265gpu_ctx* m_ctx;
};

class CostEstimateGroup
{
public:
    // frames: the window, indexed like x265's frames[]; bitDepth only matters with weightp
    CostEstimateGroup(x265gpu_ctx* ctx, Lowres* const* frames, bool bEnableWeightedPred = false, int bitDepth = 8);

    int64_t singleCost(int p0, int p1, int b, bool bIntraPenalty = false);
    void add(int p0, int p1, int b);
    void finishBatch();
    int status() const { return m_status; }      // first failure of any call, or X265GPU_OK

private:
    struct Pending { x265gpu_request req; int b, d0, d1; };
    bool makeRequest(int p0, int p1, int b, Pending& out);
    void collect(const Pending& p, const x265gpu_frame_cost& fc);

    x265gpu_ctx* m_ctx;
    Lowres* const* m_frames;
    LookaheadTLD m_tld;
    bool m_weightp;
    int m_bitDepth;
    bool m_batchMode = false;
    int m_ncu;                                    // LookaheadTLD::ncu (slicetype.cpp:1065-1066)
    int m_status = X265GPU_OK;
    std::vector<Pending> m_batch;
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

} // namespace x265gpu_host

// ---- a plain-C face of the same classes, for tests and non-C++ callers ----
extern "C" {
typedef struct xgh_window xgh_window;
xgh_window* xgh_window_create(x265gpu_ctx* ctx, int32_t nframes, int32_t bEnableWeightedPred, int32_t bitDepth);
void xgh_window_destroy(xgh_window* w);
int xgh_pre_lookahead(xgh_window* w, int32_t n, const int32_t* frames, const void* const* lumas, const intptr_t* strides,
                      const int32_t* const* invQscale);
/* the AQ form: 4:2:0 pictures; invQscale[i] receives the frame's scale factors (qgSize >= 16), the offsets go to the
 * arrays given with xgh_set_qp_offsets, wp_sum / wp_ssd of the luma plane are kept for weightsAnalyse */
int xgh_pre_lookahead_aq(xgh_window* w, int32_t n, const int32_t* frames, const void* const* y, const void* const* u,
                         const void* const* v, const intptr_t* strideY, const intptr_t* strideC,
                         const x265gpu_aq_params* aq, int32_t* const* invQscale);
void xgh_get_wp(const xgh_window* w, int32_t frame, uint64_t* sum, uint64_t* ssd);
void xgh_set_wp(xgh_window* w, int32_t frame, uint64_t sum, uint64_t ssd);
void xgh_set_slice_type(xgh_window* w, int32_t frame, int32_t sliceType);
void xgh_set_qp_offsets(xgh_window* w, int32_t frame, double* qpAqOffset, double* qpCuTreeOffset);
int64_t xgh_single_cost(xgh_window* w, int32_t p0, int32_t p1, int32_t b, int32_t bIntraPenalty);
void xgh_add(xgh_window* w, int32_t p0, int32_t p1, int32_t b);
int xgh_finish_batch(xgh_window* w);
int64_t xgh_cost_est(const xgh_window* w, int32_t b, int32_t d0, int32_t d1);
int32_t xgh_intra_mbs(const xgh_window* w, int32_t b, int32_t d0);
int32_t xgh_is_weighted(const xgh_window* w, int32_t b, int32_t d0);
int xgh_cu_tree(xgh_window* w, int32_t fpsNum, int32_t fpsDenom, int32_t bBPyramid, int32_t bEnableWeightedBiPred,
                int32_t lookaheadDepth, int32_t vbvBufferSize, double qCompress, int32_t numframes, int32_t bIntra);
int xgh_status(const xgh_window* w);
}

#endif
