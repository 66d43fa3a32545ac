// See lookahead_host.h.  Plain C++, no device code: everything that touches pixels is a call into libx265gpu.
#include "lookahead_host.h"

#include <math.h>
#include <string.h>

namespace x265gpu_host {

void Lowres::reset()
{
    for (int l = 0; l < 2; l++)
        for (int d = 0; d < MAX_D; d++)
            searched[l][d] = false;
    for (int i = 0; i < MAX_D; i++)
    {
        for (int j = 0; j < MAX_D; j++)
        {
            costEst[i][j] = costEstAq[i][j] = -1;
            pending[i][j] = false;
        }
        intraMbs[i] = 0;
        weighted[i] = false;
        weightedCostDelta[i] = 0;
    }
}

// ---- LookaheadTLD::weightsAnalyse: the float guess stays here, in single precision as in the reference;
//      weightCostLuma and weight_pp run on the device ----
int LookaheadTLD::weightsAnalyse(Lowres& fenc, Lowres& ref, int bitDepth)
{
    static const float epsilon = 1.f / 128.f;
    const int deltaIndex = fenc.frameNum - ref.frameNum;
    fenc.weighted[deltaIndex] = false;

    float guessScale, fencMean, refMean;
    if (fenc.wp_ssd && ref.wp_ssd)
        guessScale = sqrtf((float)fenc.wp_ssd / ref.wp_ssd);
    else
        guessScale = 1.0f;
    fencMean = (float)fenc.wp_sum / (m_geom.lines * m_geom.width) / (1 << (bitDepth - 8));
    refMean = (float)ref.wp_sum / (m_geom.lines * m_geom.width) / (1 << (bitDepth - 8));

    /* Early termination */
    if (fabsf(refMean - fencMean) < 0.5f && fabsf(1.f - guessScale) < epsilon)
        return X265GPU_OK;

    // WeightParam::setFromWeightAndOffset(w, 0, 7, true) (source/common/slice.h)
    int w = (int)(guessScale * 128 + 0.5f), mindenom = 7;
    while (mindenom > 0 && w > 127)
    {
        mindenom--;
        w >>= 1;
    }
    int minscale = w < 127 ? w : 127, minoff = 0;

    uint32_t origscore = 0, minscore = 0, s = 0;
    int rc = x265gpu_weight_cost(m_ctx, fenc.slot, ref.slot, 0, 0, 0, 0, &origscore);
    if (rc != X265GPU_OK)
        return rc;
    minscore = origscore;
    if (!minscore)
        return X265GPU_OK;

    int curScale = minscale;
    int curOffset = (int)(fencMean - refMean * curScale / (1 << mindenom) + 0.5f);
    if (curOffset < -128 || curOffset > 127)
    {
        /* Rescale considering the constraints on curOffset (slicetype.cpp:926-931) */
        curOffset = curOffset < -128 ? -128 : 127;
        curScale = (int)((1 << mindenom) * (fencMean - curOffset) / refMean + 0.5f);
        curScale = curScale < 0 ? 0 : (curScale > 127 ? 127 : curScale);
    }
    rc = x265gpu_weight_cost(m_ctx, fenc.slot, ref.slot, 1, curScale, mindenom, curOffset, &s);
    if (rc != X265GPU_OK)
        return rc;
    bool found = false;
    if (s < minscore)
    {
        minscore = s;
        minscale = curScale;
        minoff = curOffset;
        found = true;
    }

    /* Use a smaller denominator if possible */
    if (mindenom > 0 && minscale && !(minscale & 1))
    {
        int idx = __builtin_ctz((unsigned)minscale);
        int shift = idx < mindenom ? idx : mindenom;
        mindenom -= shift;
        minscale >>= shift;
    }

    if (!found || (minscale == (1 << mindenom) && minoff == 0) || (float)minscore / origscore > 0.998f)
        return X265GPU_OK;

    fenc.weightedCostDelta[deltaIndex] = minscore / origscore;    // integer division, slicetype.cpp:955
    rc = x265gpu_weight_planes(m_ctx, fenc.slot, deltaIndex, ref.slot, minscale, mindenom, minoff);
    if (rc != X265GPU_OK)
        return rc;
    fenc.weighted[deltaIndex] = true;
    return X265GPU_OK;
}

// ---- PreLookaheadGroup ----
int PreLookaheadGroup::processTasks(int n, Lowres* const* frames, const void* const* lumas, const intptr_t* strides,
                                    const int32_t* const* invQscale)
{
    std::vector<int32_t> slots(n);
    for (int i = 0; i < n; i++)
    {
        Lowres& fr = *frames[i];
        int rc = x265gpu_lowres_init(m_ctx, fr.slot, lumas[i], strides[i], X265GPU_MEM_HOST, fr.frameNum);
        if (rc == X265GPU_OK)
            rc = x265gpu_lowres_set_aq(m_ctx, fr.slot, invQscale ? invQscale[i] : nullptr);
        if (rc != X265GPU_OK)
            return rc;
        fr.reset();
        slots[i] = fr.slot;
    }
    int rc = x265gpu_intra_estimate(m_ctx, n, slots.data());
    if (rc != X265GPU_OK)
        return rc;
    for (int i = 0; i < n; i++)
    {
        x265gpu_frame_cost fc;
        rc = x265gpu_get_costs(m_ctx, frames[i]->slot, 0, 0, nullptr, nullptr, &fc);
        if (rc != X265GPU_OK)
            return rc;
        frames[i]->costEst[0][0] = fc.costEst;
        frames[i]->costEstAq[0][0] = fc.costEstAq;
    }
    return X265GPU_OK;
}

int PreLookaheadGroup::processTasks(int n, Lowres* const* frames, const Picture* pics, const x265gpu_aq_params& aq,
                                    const float* const* quantOffsets)
{
    std::vector<int32_t> slots(n);
    for (int i = 0; i < n; i++)
    {
        Lowres& fr = *frames[i];
        const Picture& p = pics[i];
        int rc = x265gpu_lowres_init(m_ctx, fr.slot, p.y, p.strideY, X265GPU_MEM_HOST, fr.frameNum);
        if (rc != X265GPU_OK)
            return rc;
        fr.reset();
        // tld.calcAdaptiveQuantFrame(preFrame, param) when the lookahead has AQ or weighted prediction
        // (slicetype.cpp:1849-1850); the call installs the scale factors for the slot as well
        uint64_t wpSum[3], wpSsd[3];
        rc = x265gpu_aq_frame(m_ctx, fr.slot, p.y, p.strideY, p.u, p.v, p.strideC, X265GPU_MEM_HOST, &aq,
                              quantOffsets ? quantOffsets[i] : nullptr, fr.qpAqOffset, fr.qpCuTreeOffset, fr.invQscaleFactor,
                              wpSum, wpSsd);
        if (rc != X265GPU_OK)
            return rc;
        fr.wp_sum = wpSum[0];
        fr.wp_ssd = wpSsd[0];
        slots[i] = fr.slot;
    }
    int rc = x265gpu_intra_estimate(m_ctx, n, slots.data());
    if (rc != X265GPU_OK)
        return rc;
    for (int i = 0; i < n; i++)
    {
        x265gpu_frame_cost fc;
        rc = x265gpu_get_costs(m_ctx, frames[i]->slot, 0, 0, nullptr, nullptr, &fc);
        if (rc != X265GPU_OK)
            return rc;
        frames[i]->costEst[0][0] = fc.costEst;
        frames[i]->costEstAq[0][0] = fc.costEstAq;
    }
    return X265GPU_OK;
}

// ---- CostEstimateGroup ----
CostEstimateGroup::CostEstimateGroup(x265gpu_ctx* ctx, Lowres* const* frames, bool bEnableWeightedPred, int bitDepth)
    : m_ctx(ctx), m_frames(frames), m_tld(ctx), m_weightp(bEnableWeightedPred), m_bitDepth(bitDepth)
{
    x265gpu_geometry g;
    x265gpu_get_geometry(ctx, &g);
    const int W = g.widthInCU, H = g.heightInCU;
    m_ncu = (W > 2 && H > 2) ? W * H + 4 - 2 * (W + H) : W * H;
}

// The first lines of estimateFrameCost (slicetype.cpp:4290-4310): cache check, bDoSearch, weights
bool CostEstimateGroup::makeRequest(int p0, int p1, int b, Pending& out)
{
    Lowres& fenc = *m_frames[b];
    const int d0 = b - p0, d1 = p1 - b;
    if (fenc.costEst[d0][d1] >= 0 || fenc.pending[d0][d1])
        return false;
    const bool do0 = !fenc.searched[0][d0];
    const bool do1 = p1 > b && !fenc.searched[1][d1];
    if (do0)
        fenc.searched[0][d0] = true;
    if (do1)
        fenc.searched[1][d1] = true;
    fenc.pending[d0][d1] = true;
    bool weighted = false;
    if (m_weightp)
    {
        fenc.weighted[d0] = false;
        if (do0)
        {
            int rc = m_tld.weightsAnalyse(fenc, *m_frames[p0], m_bitDepth);
            if (rc != X265GPU_OK && m_status == X265GPU_OK)
                m_status = rc;
            weighted = fenc.weighted[d0];
        }
    }
    memset(&out.req, 0, sizeof(out.req));
    out.req.slotP0 = m_frames[p0]->slot;
    out.req.slotB = fenc.slot;
    out.req.slotP1 = m_frames[p1]->slot;
    out.req.p0 = p0;
    out.req.b = b;
    out.req.p1 = p1;
    out.req.doSearch[0] = do0;
    out.req.doSearch[1] = do1;
    out.req.weightedL0 = weighted;
    out.req.bCoopSlices = !m_batchMode;        // slicetype.cpp:4312
    out.b = b;
    out.d0 = d0;
    out.d1 = d1;
    return true;
}

void CostEstimateGroup::collect(const Pending& p, const x265gpu_frame_cost& fc)
{
    Lowres& fenc = *m_frames[p.b];
    fenc.costEst[p.d0][p.d1] = fc.costEst;
    fenc.costEstAq[p.d0][p.d1] = fc.costEstAq;
    if (p.d1 == 0)
        fenc.intraMbs[p.d0] += fc.intraMbs;
    fenc.pending[p.d0][p.d1] = false;
}

int64_t CostEstimateGroup::singleCost(int p0, int p1, int b, bool bIntraPenalty)
{
    Pending p;
    if (makeRequest(p0, p1, b, p))
    {
        x265gpu_frame_cost fc;
        int rc = x265gpu_estimate_costs(m_ctx, 1, &p.req, &fc);
        if (rc != X265GPU_OK)
        {
            if (m_status == X265GPU_OK)
                m_status = rc;
            m_frames[b]->pending[p.d0][p.d1] = false;
            return -1;
        }
        collect(p, fc);
    }
    Lowres& fenc = *m_frames[b];
    int64_t score = fenc.costEst[b - p0][p1 - b];
    if (bIntraPenalty)
        // arbitrary penalty for I-blocks after B-frames (slicetype.cpp:4378-4380)
        score += score * fenc.intraMbs[b - p0] / (m_ncu * 8);
    return score;
}

void CostEstimateGroup::add(int p0, int p1, int b)
{
    m_batchMode = true;
    Pending p;
    if (makeRequest(p0, p1, b, p))
        m_batch.push_back(p);
}

void CostEstimateGroup::finishBatch()
{
    if (m_batch.empty())
        return;
    std::vector<x265gpu_request> reqs(m_batch.size());
    std::vector<x265gpu_frame_cost> out(m_batch.size());
    for (size_t i = 0; i < m_batch.size(); i++)
        reqs[i] = m_batch[i].req;
    int rc = x265gpu_estimate_costs(m_ctx, (int32_t)reqs.size(), reqs.data(), out.data());
    if (rc != X265GPU_OK)
    {
        if (m_status == X265GPU_OK)
            m_status = rc;
        for (const Pending& p : m_batch)
            m_frames[p.b]->pending[p.d0][p.d1] = false;
    }
    else
        for (size_t i = 0; i < m_batch.size(); i++)
            collect(m_batch[i], out[i]);
    m_batch.clear();
}

// ---- cuTree ----
static double clipDuration(double f) { return f < 0.01 ? 0.01 : (f > 1.0 ? 1.0 : f); }   // ratecontrol.h:44-49

CuTree::CuTree(x265gpu_ctx* ctx, const CuTreeParam& p) : m_ctx(ctx), m_p(p)
{
    m_frameDuration = (double)p.fpsDenom / p.fpsNum;
    m_cuTreeStrength = 5.0 * (1.0 - p.qCompress);
}

void CuTree::cuTreeFinish(Lowres& frame, double averageDuration, int ref0Distance)
{
    const int fpsFactor = (int)(clipDuration(averageDuration) / clipDuration(m_frameDuration) * 256);
    double weightdelta = 0.0;
    if (ref0Distance && frame.weightedCostDelta[ref0Distance - 1] > 0)
        weightdelta = 1.0 - frame.weightedCostDelta[ref0Distance - 1];
    note(x265gpu_cutree_finish(m_ctx, frame.slot, fpsFactor, weightdelta, m_cuTreeStrength, frame.qpAqOffset,
                               frame.qpCuTreeOffset));
}

void CuTree::estimateCUPropagate(Lowres* const* frames, double averageDuration, int p0, int p1, int b, int referenced)
{
    const int distScaleFactor = (((b - p0) << 8) + ((p1 - p0) >> 1)) / (p1 - p0);
    const int bipredWeight = m_p.bEnableWeightedBiPred ? 64 - (distScaleFactor >> 2) : 32;
    const double fpsFactor = clipDuration(m_frameDuration) / clipDuration(averageDuration);
    note(x265gpu_cutree_propagate(m_ctx, frames[p0]->slot, frames[p1]->slot, frames[b]->slot, b - p0, p1 - b, referenced,
                                  fpsFactor, bipredWeight));
    if (m_p.vbvBufferSize && m_p.lookaheadDepth && referenced)
        cuTreeFinish(*frames[b], averageDuration, b == p1 ? b - p0 : 0);
}

// Lookahead::cuTree, slicetype.cpp:3665-3766
int CuTree::cuTree(Lowres* const* frames, int numframes, bool bIntra)
{
    const int idx = !bIntra;
    int lastnonb, curnonb = 1;
    int bframes = 0;

    double totalDuration = 0.0;
    for (int j = 0; j <= numframes; j++)
        totalDuration += m_frameDuration;
    const double averageDuration = totalDuration / (numframes + 1);

    int i = numframes;
    while (i > 0 && frames[i]->sliceType == TYPE_B)
        i--;
    lastnonb = i;

    /* Lookaheadless MB-tree is not a theoretically distinct case; the same extrapolation could
     * be applied to the end of a lookahead buffer of any size. */
    if (!m_p.lookaheadDepth)
    {
        if (bIntra)
        {
            note(x265gpu_cutree_clear(m_ctx, frames[0]->slot));
            x265gpu_geometry g;
            x265gpu_get_geometry(m_ctx, &g);
            memcpy(frames[0]->qpCuTreeOffset, frames[0]->qpAqOffset, sizeof(double) * g.cuCount);
            return m_status;
        }
        note(x265gpu_cutree_swap(m_ctx, frames[lastnonb]->slot, frames[0]->slot));
        note(x265gpu_cutree_clear(m_ctx, frames[0]->slot));
    }
    else
    {
        if (lastnonb < idx)
            return m_status;
        note(x265gpu_cutree_clear(m_ctx, frames[lastnonb]->slot));
    }

    CostEstimateGroup estGroup(m_ctx, frames);

    while (i-- > idx)
    {
        curnonb = i;
        while (frames[curnonb]->sliceType == TYPE_B && curnonb > 0)
            curnonb--;
        if (curnonb < idx)
            break;

        estGroup.singleCost(curnonb, lastnonb, lastnonb);
        note(x265gpu_cutree_clear(m_ctx, frames[curnonb]->slot));
        bframes = lastnonb - curnonb - 1;
        if (m_p.bBPyramid && bframes > 1)
        {
            const int middle = (bframes + 1) / 2 + curnonb;
            estGroup.singleCost(curnonb, lastnonb, middle);
            note(x265gpu_cutree_clear(m_ctx, frames[middle]->slot));
            while (i > curnonb)
            {
                const int p0 = i > middle ? middle : curnonb;
                const int p1 = i < middle ? middle : lastnonb;
                if (i != middle)
                {
                    estGroup.singleCost(p0, p1, i);
                    estimateCUPropagate(frames, averageDuration, p0, p1, i, 0);
                }
                i--;
            }
            estimateCUPropagate(frames, averageDuration, curnonb, lastnonb, middle, 1);
        }
        else
        {
            while (i > curnonb)
            {
                estGroup.singleCost(curnonb, lastnonb, i);
                estimateCUPropagate(frames, averageDuration, curnonb, lastnonb, i, 0);
                i--;
            }
        }
        estimateCUPropagate(frames, averageDuration, curnonb, lastnonb, lastnonb, 1);
        lastnonb = curnonb;
    }

    if (!m_p.lookaheadDepth)
    {
        estGroup.singleCost(0, lastnonb, lastnonb);
        estimateCUPropagate(frames, averageDuration, 0, lastnonb, lastnonb, 1);
        note(x265gpu_cutree_swap(m_ctx, frames[lastnonb]->slot, frames[0]->slot));
    }

    cuTreeFinish(*frames[lastnonb], averageDuration, lastnonb);
    if (m_p.bBPyramid && bframes > 1 && !m_p.vbvBufferSize)
        cuTreeFinish(*frames[lastnonb + (bframes + 1) / 2], averageDuration, 0);
    note(estGroup.status());
    return m_status;
}

} // namespace x265gpu_host

// ---- plain-C face ----
using namespace x265gpu_host;

struct xgh_window
{
    x265gpu_ctx* ctx;
    std::vector<Lowres> store;
    std::vector<Lowres*> frames;
    CostEstimateGroup* est;
    int status;
};

extern "C" {

xgh_window* xgh_window_create(x265gpu_ctx* ctx, int32_t nframes, int32_t bEnableWeightedPred, int32_t bitDepth)
{
    if (!ctx || nframes < 1)
        return nullptr;
    xgh_window* w = new xgh_window;
    w->ctx = ctx;
    w->store.resize(nframes);
    w->frames.resize(nframes);
    for (int i = 0; i < nframes; i++)
    {
        w->store[i].slot = i;
        w->store[i].frameNum = i;
        w->frames[i] = &w->store[i];
    }
    w->est = new CostEstimateGroup(ctx, w->frames.data(), bEnableWeightedPred != 0, bitDepth);
    w->status = X265GPU_OK;
    return w;
}

void xgh_window_destroy(xgh_window* w)
{
    if (!w)
        return;
    delete w->est;
    delete w;
}

int xgh_pre_lookahead(xgh_window* w, int32_t n, const int32_t* frames, const void* const* lumas, const intptr_t* strides,
                      const int32_t* const* invQscale)
{
    std::vector<Lowres*> fr(n);
    for (int i = 0; i < n; i++)
        fr[i] = w->frames[frames[i]];
    return PreLookaheadGroup(w->ctx).processTasks(n, fr.data(), lumas, strides, invQscale);
}

int xgh_pre_lookahead_aq(xgh_window* w, int32_t n, const int32_t* frames, const void* const* y, const void* const* u,
                         const void* const* v, const intptr_t* strideY, const intptr_t* strideC,
                         const x265gpu_aq_params* aq, int32_t* const* invQscale)
{
    std::vector<Lowres*> fr(n);
    std::vector<PreLookaheadGroup::Picture> pics(n);
    for (int i = 0; i < n; i++)
    {
        fr[i] = w->frames[frames[i]];
        fr[i]->invQscaleFactor = invQscale ? invQscale[i] : nullptr;
        pics[i] = { y[i], u[i], v[i], strideY[i], strideC[i] };
    }
    return PreLookaheadGroup(w->ctx).processTasks(n, fr.data(), pics.data(), *aq, nullptr);
}

void xgh_get_wp(const xgh_window* w, int32_t frame, uint64_t* sum, uint64_t* ssd)
{
    *sum = w->frames[frame]->wp_sum;
    *ssd = w->frames[frame]->wp_ssd;
}

void xgh_set_wp(xgh_window* w, int32_t frame, uint64_t sum, uint64_t ssd)
{
    w->frames[frame]->wp_sum = sum;
    w->frames[frame]->wp_ssd = ssd;
}

void xgh_set_slice_type(xgh_window* w, int32_t frame, int32_t sliceType) { w->frames[frame]->sliceType = sliceType; }

void xgh_set_qp_offsets(xgh_window* w, int32_t frame, double* qpAqOffset, double* qpCuTreeOffset)
{
    w->frames[frame]->qpAqOffset = qpAqOffset;
    w->frames[frame]->qpCuTreeOffset = qpCuTreeOffset;
}

int64_t xgh_single_cost(xgh_window* w, int32_t p0, int32_t p1, int32_t b, int32_t bIntraPenalty)
{
    return w->est->singleCost(p0, p1, b, bIntraPenalty != 0);
}

void xgh_add(xgh_window* w, int32_t p0, int32_t p1, int32_t b) { w->est->add(p0, p1, b); }

int xgh_finish_batch(xgh_window* w)
{
    w->est->finishBatch();
    return w->est->status();
}

int64_t xgh_cost_est(const xgh_window* w, int32_t b, int32_t d0, int32_t d1) { return w->frames[b]->costEst[d0][d1]; }
int32_t xgh_intra_mbs(const xgh_window* w, int32_t b, int32_t d0) { return w->frames[b]->intraMbs[d0]; }
int32_t xgh_is_weighted(const xgh_window* w, int32_t b, int32_t d0) { return w->frames[b]->weighted[d0]; }

int xgh_cu_tree(xgh_window* w, int32_t fpsNum, int32_t fpsDenom, int32_t bBPyramid, int32_t bEnableWeightedBiPred,
                int32_t lookaheadDepth, int32_t vbvBufferSize, double qCompress, int32_t numframes, int32_t bIntra)
{
    CuTreeParam p;
    p.fpsNum = fpsNum;
    p.fpsDenom = fpsDenom;
    p.bBPyramid = bBPyramid != 0;
    p.bEnableWeightedBiPred = bEnableWeightedBiPred != 0;
    p.lookaheadDepth = lookaheadDepth;
    p.vbvBufferSize = vbvBufferSize;
    p.qCompress = qCompress;
    return CuTree(w->ctx, p).cuTree(w->frames.data(), numframes, bIntra != 0);
}

int xgh_status(const xgh_window* w) { return w->est->status(); }

} // extern "C"
