// See lookahead_host.h.  Plain C++, no device code: everything that touches pixels is a call into libx265gpu.
#include "lookahead_host.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>

namespace x265gpu_host {

void Lowres::reset()
{
    for (int l = 0; l < 2; l++)
        for (int d = 0; d < MAX_D; d++)
        {
            searched[l][d] = false;
            presearched[l][d] = PRE_NONE;
        }
    memset(preCost, 0, sizeof(preCost));
    for (int d = 0; d < MAX_D; d++)
        presearchWeighted[d] = false;
    intraTotalsPending = false;
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
    bLastMiniGopBFrame = false;
    bKeyframe = false;
    bIsFadeEnd = false;               // lowres.cpp:348
    leadingBframes = 0;
    indB = 0;
    if (!plannedType.empty())
        plannedType.assign(plannedType.size(), TYPE_AUTO);
}

// ---- LookaheadTLD::weightsAnalyse (slicetype.cpp:870-971) with two substitutions: weightCostLuma ->
//      x265gpu_weight_cost, the weight_pp loop -> x265gpu_weight_planes.  The guess arithmetic is x265's own, in
//      single precision, so that the same pictures give the same scale / offset / denominator. ----
int LookaheadTLD::weightsAnalyse(Lowres& fenc, Lowres& ref, int bitDepth)
{
    Lowres* f = &fenc;
    Lowres* r = &ref;
    return weightsAnalyse(1, &f, &r, bitDepth);
}

int LookaheadTLD::weightsAnalyse(int n, Lowres* const* fencs, Lowres* const* refs, int bitDepth)
{
    static const float epsilon = 1.f / 128.f;
    struct Guess { int idx; int mindenom, minscale; float fencMean, refMean; int curScale, curOffset; uint32_t orig, s; };
    std::vector<Guess> live;
    for (int i = 0; i < n; i++)
    {
        Lowres& fenc = *fencs[i];
        Lowres& ref = *refs[i];
        const int deltaIndex = fenc.frameNum - ref.frameNum;
        fenc.weighted[deltaIndex] = false;

        float guessScale, fencMean, refMean;
        if (fenc.wp_ssd && ref.wp_ssd)
            guessScale = sqrtf((float)fenc.wp_ssd / ref.wp_ssd);
        else
            guessScale = 1.0f;
        fencMean = (float)fenc.wp_sum / (m_geom.lines * m_geom.width) / (1 << (bitDepth - 8));
        refMean = (float)ref.wp_sum / (m_geom.lines * m_geom.width) / (1 << (bitDepth - 8));

        // nothing to gain when means and spread already agree (:906)
        if (fabsf(refMean - fencMean) < 0.5f && fabsf(1.f - guessScale) < epsilon)
            continue;

        // WeightParam::setFromWeightAndOffset(w, 0, 7, true) (source/common/slice.h)
        int w = (int)(guessScale * 128 + 0.5f), mindenom = 7;
        while (mindenom > 0 && w > 127)
        {
            mindenom--;
            w >>= 1;
        }
        Guess g;
        g.idx = i;
        g.mindenom = mindenom;
        g.minscale = w < 127 ? w : 127;
        g.fencMean = fencMean;
        g.refMean = refMean;
        g.orig = g.s = 0;
        g.curScale = g.curOffset = 0;
        live.push_back(g);
    }
    if (live.empty())
        return X265GPU_OK;

    // first device pass: the unweighted cost of every pair left
    std::vector<x265gpu_weight_job> jobs(live.size());
    std::vector<uint32_t> costs(live.size());
    for (size_t k = 0; k < live.size(); k++)
        jobs[k] = { fencs[live[k].idx]->slot, refs[live[k].idx]->slot, 0, 0, 0, 0 };
    int rc = x265gpu_weight_costs(m_ctx, (int32_t)jobs.size(), jobs.data(), costs.data());
    if (rc != X265GPU_OK)
        return rc;
    std::vector<Guess> second;
    for (size_t k = 0; k < live.size(); k++)
    {
        Guess g = live[k];
        g.orig = costs[k];
        if (!g.orig)
            continue;
        g.curScale = g.minscale;
        g.curOffset = (int)(g.fencMean - g.refMean * g.curScale / (1 << g.mindenom) + 0.5f);
        if (g.curOffset < -128 || g.curOffset > 127)
        {
            /* Rescale considering the constraints on curOffset (slicetype.cpp:926-931) */
            g.curOffset = g.curOffset < -128 ? -128 : 127;
            g.curScale = (int)((1 << g.mindenom) * (g.fencMean - g.curOffset) / g.refMean + 0.5f);
            g.curScale = g.curScale < 0 ? 0 : (g.curScale > 127 ? 127 : g.curScale);
        }
        second.push_back(g);
    }
    if (second.empty())
        return X265GPU_OK;

    // second device pass: the cost with the guessed weight
    jobs.resize(second.size());
    costs.resize(second.size());
    for (size_t k = 0; k < second.size(); k++)
        jobs[k] = { fencs[second[k].idx]->slot, refs[second[k].idx]->slot, 1, second[k].curScale, second[k].mindenom, second[k].curOffset };
    rc = x265gpu_weight_costs(m_ctx, (int32_t)jobs.size(), jobs.data(), costs.data());
    if (rc != X265GPU_OK)
        return rc;
    for (size_t k = 0; k < second.size(); k++)
    {
        const Guess& g = second[k];
        Lowres& fenc = *fencs[g.idx];
        Lowres& ref = *refs[g.idx];
        const int deltaIndex = fenc.frameNum - ref.frameNum;
        uint32_t minscore = g.orig;
        int minscale = g.minscale, minoff = 0, mindenom = g.mindenom;
        bool found = false;
        if (costs[k] < minscore)
        {
            minscore = costs[k];
            minscale = g.curScale;
            minoff = g.curOffset;
            found = true;
        }

        // drop common factors of two from scale and denominator (:943-950)
        if (mindenom > 0 && minscale && !(minscale & 1))
        {
            int idx = __builtin_ctz((unsigned)minscale);
            int shift = idx < mindenom ? idx : mindenom;
            mindenom -= shift;
            minscale >>= shift;
        }

        if (!found || (minscale == (1 << mindenom) && minoff == 0) || (float)minscore / g.orig > 0.998f)
            continue;

        fenc.weightedCostDelta[deltaIndex] = minscore / g.orig;    // integer division, slicetype.cpp:955
        rc = x265gpu_weight_planes(m_ctx, fenc.slot, deltaIndex, ref.slot, minscale, mindenom, minoff);
        if (rc != X265GPU_OK)
            return rc;
        fenc.weighted[deltaIndex] = true;
    }
    return X265GPU_OK;
}

// aqLayerDepth (source/common/lowres.h:120-139) as one bit per enabled depth, by CTU size (64, 32, 16) and
// log2(maxCUSize) - log2(qgSize); minAQSize (:142) is the depth whose layer has the doubled lowres grid
static const unsigned char kAqDepthMask[3][4] = { { 0x5, 0x7, 0x7, 0xf }, { 0x3, 0x3, 0x7, 0x0 }, { 0x1, 0x3, 0x0, 0x0 } };
static const int kMinAqSize[3] = { 3, 2, 1 };

void Lowres::createAqLayers(const x265gpu_geometry& g, int maxCUSize, int qgSize)
{
    aqLayers.clear();
    int log2Ctu = 0, log2Qg = 0;
    while ((1 << log2Ctu) < maxCUSize)
        log2Ctu++;
    while ((1 << log2Qg) < qgSize)
        log2Qg++;
    const int ctuSizeIdx = 6 - log2Ctu, aqDepth = log2Ctu - log2Qg;
    if (ctuSizeIdx < 0 || ctuSizeIdx > 2 || aqDepth < 0 || aqDepth > 3)
        return;
    for (int d = 0; d < 4; d++)
    {
        if (!(kAqDepthMask[ctuSizeIdx][aqDepth] >> d & 1))
            continue;
        AqLayer l;
        l.partWidth = l.partHeight = maxCUSize >> d;
        if (kMinAqSize[ctuSizeIdx] == d)
        {
            l.numPartInWidth = g.widthInCU * 2;
            l.numPartInHeight = g.heightInCU * 2;
        }
        else
        {
            l.numPartInWidth = (g.sourceWidth + l.partWidth - 1) / l.partWidth;
            l.numPartInHeight = (g.sourceHeight + l.partHeight - 1) / l.partHeight;
        }
        const size_t n = (size_t)l.numPartInWidth * l.numPartInHeight;
        l.dActivity.assign(n, 0.0);
        l.dQpOffset.assign(n, 0.0);
        l.dCuTreeOffset.assign(n, 0.0);
        l.dAvgActivity = 0;
        aqLayers.push_back(std::move(l));
    }
    if (qgSize == 8)
        dCuTreeOffset8x8.assign((size_t)g.cuCount * 4, 0.0);
}

// the frame's layers as the library's call takes them
static std::vector<x265gpu_hevc_aq_layer> layerDescriptors(Lowres& fr)
{
    std::vector<x265gpu_hevc_aq_layer> d(fr.aqLayers.size());
    for (size_t k = 0; k < d.size(); k++)
    {
        Lowres::AqLayer& l = fr.aqLayers[k];
        d[k].aqPartWidth = l.partWidth;
        d[k].aqPartHeight = l.partHeight;
        d[k].numAQPartInWidth = l.numPartInWidth;
        d[k].numAQPartInHeight = l.numPartInHeight;
        d[k].dActivity = l.dActivity.data();
        d[k].dQpOffset = l.dQpOffset.data();
        d[k].dCuTreeOffset = l.dCuTreeOffset.data();
        d[k].dAvgActivity = l.dAvgActivity;
    }
    return d;
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
    if (n < 1)
        return X265GPU_OK;
    std::vector<int32_t> slots(n);
    std::vector<x265gpu_aq_picture> ap(n);
    // Lowres::init of every frame, then calcAdaptiveQuantFrame of every frame (when the lookahead has AQ or weighted
    // prediction, slicetype.cpp:1849-1850; the call installs the scale factors for the slots as well), then
    // lowresIntraEstimate of every frame: the per-frame task of x265 regrouped by stage, which changes no result
    // because a frame's task reads nothing of another frame's
    std::vector<const void*> lumas(n);
    std::vector<intptr_t> strides(n);
    std::vector<int32_t> pocs(n);
    for (int i = 0; i < n; i++)
    {
        slots[i] = frames[i]->slot;
        lumas[i] = pics[i].y;
        strides[i] = pics[i].strideY;
        pocs[i] = frames[i]->frameNum;
    }
    int rc = x265gpu_lowres_init_batch(m_ctx, n, slots.data(), lumas.data(), strides.data(), pics[0].memKind, pocs.data());
    if (rc != X265GPU_OK)
        return rc;
    for (int i = 0; i < n; i++)
    {
        Lowres& fr = *frames[i];
        const Picture& p = pics[i];
        fr.reset();
        x265gpu_aq_picture& a = ap[i];
        memset(&a, 0, sizeof(a));
        a.slot = fr.slot;
        a.y = p.y;
        a.u = p.u;
        a.v = p.v;
        a.strideY = p.strideY;
        a.strideC = p.strideC;
        a.quantOffsets = quantOffsets ? quantOffsets[i] : nullptr;
        a.qpAqOffset = fr.qpAqOffset;
        a.qpCuTreeOffset = fr.qpCuTreeOffset;
        a.invQscaleFactor = fr.invQscaleFactor;
    }
    // The intra estimate goes first: its per-block costs need no AQ (weightsAnalyse reads them), so it runs on the
    // library's side stream beside the AQ energy pass; its AQ-weighted sums follow the scale factors by themselves.
    rc = x265gpu_intra_estimate(m_ctx, n, slots.data());
    if (rc != X265GPU_OK)
        return rc;
    if (aq.bBlockVariance || aq.qgSize == 8 || aq.bHevcAq)
    {
        // forms of calcAdaptiveQuantFrame with per-picture outputs beyond the three arrays (the variance tail of
        // --fades / --dynamic-refine, invQscaleFactor8x8 of qgSize 8, the layers of rc.hevcAq): one call per picture
        x265gpu_geometry g;
        x265gpu_get_geometry(m_ctx, &g);
        for (int i = 0; i < n; i++)
        {
            Lowres& fr = *frames[i];
            x265gpu_aq_params q = aq;
            const size_t blocks = (size_t)g.cuCount * (aq.qgSize == 8 ? 4 : 1);
            if (aq.bBlockVariance)
            {
                fr.blockVariance.assign(blocks, 0);
                q.blockVariance = fr.blockVariance.data();
                q.frameVariance = &fr.frameVariance;
            }
            if (aq.qgSize == 8)
            {
                fr.invQscaleFactor8x8.assign((size_t)g.cuCount, 0);
                q.invQscaleFactor8x8 = fr.invQscaleFactor8x8.data();
            }
            std::vector<x265gpu_hevc_aq_layer> layers;
            if (aq.bHevcAq)
            {
                if (fr.aqLayers.empty())
                    return X265GPU_ERR_ARG;             // the caller creates them (Lowres::createAqLayers)
                layers = layerDescriptors(fr);
                q.numLayers = (int32_t)layers.size();
                q.layers = layers.data();
            }
            uint64_t wpSum[3], wpSsd[3];
            rc = x265gpu_aq_frame(m_ctx, fr.slot, pics[i].y, pics[i].strideY, pics[i].u, pics[i].v, pics[i].strideC,
                                  pics[i].memKind, &q, ap[i].quantOffsets, fr.qpAqOffset, fr.qpCuTreeOffset,
                                  fr.invQscaleFactor, wpSum, wpSsd);
            if (rc != X265GPU_OK)
                return rc;
            for (size_t k = 0; k < layers.size(); k++)
                fr.aqLayers[k].dAvgActivity = layers[k].dAvgActivity;
            fr.wp_sum = wpSum[0];
            fr.wp_ssd = wpSsd[0];
            fr.intraTotalsPending = true;
        }
        return X265GPU_OK;
    }
    // pictures of one batch live in one kind of memory.  The sums that weightsAnalyse reads are back when this
    // returns; the offsets are still being mapped on the library's host threads and are finished from inside the
    // next estimate call, after its searches have been queued: resolveIntraTotals() then reads the frames' intra totals.
    rc = x265gpu_aq_frames_begin(m_ctx, n, ap.data(), pics[0].memKind, &aq);
    if (rc != X265GPU_OK)
        return rc;
    for (int i = 0; i < n; i++)
    {
        frames[i]->wp_sum = ap[i].wpSum[0];
        frames[i]->wp_ssd = ap[i].wpSsd[0];
        frames[i]->intraTotalsPending = true;
    }
    return X265GPU_OK;
}

int resolveIntraTotals(x265gpu_ctx* ctx, Lowres* const* frames, int n)
{
    std::vector<int32_t> slots;
    std::vector<Lowres*> which;
    for (int i = 0; i < n; i++)
        if (frames[i] && frames[i]->intraTotalsPending)
        {
            slots.push_back(frames[i]->slot);
            which.push_back(frames[i]);
        }
    if (which.empty())
        return X265GPU_OK;
    std::vector<x265gpu_frame_cost> fc(which.size());
    const int rc = x265gpu_get_frame_costs(ctx, (int32_t)which.size(), slots.data(), nullptr, nullptr, fc.data());
    if (rc != X265GPU_OK)
        return rc;
    for (size_t i = 0; i < which.size(); i++)
    {
        which[i]->costEst[0][0] = fc[i].costEst;
        which[i]->costEstAq[0][0] = fc[i].costEstAq;
        which[i]->intraTotalsPending = false;
    }
    return X265GPU_OK;
}

// ---- CostEstimateGroup ----
CostEstimateGroup::CostEstimateGroup(x265gpu_ctx* ctx, Lowres* const* frames, bool bEnableWeightedPred, int bitDepth,
                                     std::vector<RequestTrace>* trace)
    : m_ctx(ctx), m_frames(frames), m_tld(ctx), m_weightp(bEnableWeightedPred), m_bitDepth(bitDepth), m_trace(trace)
{
    x265gpu_geometry g;
    x265gpu_get_geometry(ctx, &g);
    const int W = g.widthInCU, H = g.heightInCU;
    m_ncu = (W > 2 && H > 2) ? W * H + 4 - 2 * (W + H) : W * H;
    m_numCoopSlices = g.numCoopSlices;
}

// weightsAnalyse for the list-0 search of (fenc, ref); returns isWeighted
bool CostEstimateGroup::analyseWeights(Lowres& fenc, Lowres& ref)
{
    const int d0 = fenc.frameNum - ref.frameNum;
    int rc = m_tld.weightsAnalyse(fenc, ref, m_bitDepth);
    if (rc != X265GPU_OK && m_status == X265GPU_OK)
        m_status = rc;
    return fenc.weighted[d0];
}

// The first lines of estimateFrameCost (slicetype.cpp:4290-4310): cache check, bDoSearch, weights
bool CostEstimateGroup::makeRequest(int p0, int p1, int b, Pending& out)
{
    Lowres& fenc = *m_frames[b];
    const int d0 = b - p0, d1 = p1 - b;
    if (fenc.costEst[d0][d1] >= 0 || fenc.pending[d0][d1])
        return false;
    // what x265 would search now ...
    const bool need0 = !fenc.searched[0][d0];
    const bool need1 = p1 > b && !fenc.searched[1][d1];
    // ... and what is left of it after the searches that ran ahead (only outside batch mode: they were made for
    // single requests, with their cooperative slices)
    const bool usable = !m_batchMode || m_numCoopSlices <= 1;     // without slices a batch request searches the same way
    const bool pre0 = need0 && usable && fenc.presearched[0][d0] == (p1 > b ? PRE_B : PRE_P);
    const bool pre1 = need1 && usable && fenc.presearched[1][d1] == PRE_B;
    const bool do0 = need0 && !pre0, do1 = need1 && !pre1;
    m_presearchHits += (int)pre0 + (int)pre1;
    if (need0)
    {
        fenc.searched[0][d0] = true;
        fenc.presearched[0][d0] = PRE_NONE;       // a regular search overwrites whatever ran ahead
    }
    if (need1)
    {
        fenc.searched[1][d1] = true;
        fenc.presearched[1][d1] = PRE_NONE;
    }
    // a search that runs now rewrites vectors: costs measured ahead over the old ones are void
    if (do0)
        for (int j = 0; j < MAX_D; j++)
            fenc.preCost[d0][j].valid = false;
    if (do1)
        for (int i = 0; i < MAX_D; i++)
            fenc.preCost[i][d1].valid = false;
    fenc.pending[d0][d1] = true;
    if (m_trace)
        m_trace->push_back({ m_frames[p0]->frameNum, m_frames[p1]->frameNum, fenc.frameNum, { need0, need1 }, m_batchMode });
    bool weighted = false;
    out.deferWeights = false;
    out.presearch = false;
    out.precost = false;
    out.p0 = p0;
    if (m_weightp)
    {
        fenc.weighted[d0] = false;
        if (pre0)
            weighted = fenc.weighted[d0] = fenc.presearchWeighted[d0];    // the analysis ran with the search
        else if (do0 && m_batchMode)
            out.deferWeights = true;           // finishBatch analyses the batch's pairs together
        else if (do0)
            weighted = analyseWeights(fenc, *m_frames[p0]);
    }
    if (!do0 && !do1 && fenc.preCost[d0][d1].valid)
    {
        // nothing to search, and the cost pass over these vectors ran ahead (addPrecost): its result is this request's
        const Lowres::PreCost& pc = fenc.preCost[d0][d1];
        fenc.costEst[d0][d1] = pc.costEst;
        fenc.costEstAq[d0][d1] = pc.costEstAq;
        if (d1 == 0)
            fenc.intraMbs[d0] += pc.intraMbs;
        fenc.preCost[d0][d1].valid = false;
        fenc.pending[d0][d1] = false;
        m_precostHits++;
        return false;
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
    out.req.weightedL0 = weighted && do0;
    out.req.bCoopSlices = !m_batchMode;        // slicetype.cpp:4312
    out.b = b;
    out.d0 = d0;
    out.d1 = d1;
    return true;
}

void CostEstimateGroup::addPresearch(int p0, int p1, int b, int lists)
{
    Lowres& fenc = *m_frames[b];
    const int d0 = b - p0, d1 = p1 - b;
    const uint8_t kind = p1 > b ? PRE_B : PRE_P;
    const bool do0 = (lists & 1) && !fenc.searched[0][d0] && fenc.presearched[0][d0] != kind;
    const bool do1 = (lists & 2) && p1 > b && !fenc.searched[1][d1] && fenc.presearched[1][d1] != PRE_B;
    if (!do0 && !do1)
        return;
    m_batchMode = true;
    Pending p;
    memset(&p.req, 0, sizeof(p.req));
    const bool weighted = false;
    p.deferWeights = do0 && m_weightp;         // analysed with the batch's other pairs in finishBatch
    p.presearch = true;
    p.precost = false;
    p.p0 = p0;
    if (do0)
    {
        fenc.presearched[0][d0] = kind;
        fenc.presearchWeighted[d0] = false;
        for (int j = 0; j < MAX_D; j++)
            fenc.preCost[d0][j].valid = false;
    }
    if (do1)
    {
        fenc.presearched[1][d1] = PRE_B;
        for (int i = 0; i < MAX_D; i++)
            fenc.preCost[i][d1].valid = false;
    }
    p.req.slotP0 = m_frames[p0]->slot;
    p.req.slotB = fenc.slot;
    p.req.slotP1 = m_frames[p1]->slot;
    p.req.p0 = p0;
    p.req.b = b;
    p.req.p1 = p1;
    p.req.doSearch[0] = do0;
    p.req.doSearch[1] = do1;
    p.req.weightedL0 = weighted;
    p.req.bCoopSlices = 1;                     // the request this stands in for is a single one
    p.req.bSearchOnly = 1;
    p.b = b;
    p.d0 = d0;
    p.d1 = d1;
    m_batch.push_back(p);
}

// Not in x265: the cost pass of a request x265 has not made yet, over vectors that are on the device already - found
// by a regular search or by one that ran ahead (addPresearch), of the kind (P or bidirectional) the request would
// search with.  The result is kept aside (Lowres::preCost) and becomes the request's when x265's order makes it and
// finds nothing left to search (makeRequest); the per-block costs and row sums it leaves on the device are the ones
// the request would write.  Nothing x265 can observe changes before that moment.
void CostEstimateGroup::addPrecost(int p0, int p1, int b)
{
    Lowres& fenc = *m_frames[b];
    const int d0 = b - p0, d1 = p1 - b;
    if (fenc.costEst[d0][d1] >= 0 || fenc.pending[d0][d1] || fenc.preCost[d0][d1].valid)
        return;
    const uint8_t kind = p1 > b ? PRE_B : PRE_P;
    const bool have0 = fenc.searched[0][d0] || fenc.presearched[0][d0] == kind;
    const bool have1 = p1 == b || fenc.searched[1][d1] || fenc.presearched[1][d1] == PRE_B;
    if (!have0 || !have1)
        return;
    for (const Pending& q : m_batch)
        if (q.precost && q.b == b && q.d0 == d0 && q.d1 == d1)
            return;
    m_batchMode = true;
    Pending p;
    memset(&p.req, 0, sizeof(p.req));
    p.deferWeights = false;
    p.presearch = false;
    p.precost = true;
    p.p0 = p0;
    p.req.slotP0 = m_frames[p0]->slot;
    p.req.slotB = fenc.slot;
    p.req.slotP1 = m_frames[p1]->slot;
    p.req.p0 = p0;
    p.req.b = b;
    p.req.p1 = p1;
    p.b = b;
    p.d0 = d0;
    p.d1 = d1;
    m_batch.push_back(p);
}

void CostEstimateGroup::collect(const Pending& p, const x265gpu_frame_cost& fc)
{
    if (p.req.bSearchOnly)
        return;
    Lowres& fenc = *m_frames[p.b];
    if (p.precost)
    {
        fenc.preCost[p.d0][p.d1] = { fc.costEst, fc.costEstAq, fc.intraMbs, true };
        return;
    }
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

// weightsAnalyse of every list-0 search in the batch, two device passes for all of them together.  A regular
// request's verdict goes to fenc.weighted[d0] as in x265; a presearch's verdict is filed with the presearch, and
// fenc.weighted[d0] / weightedCostDelta[d0], which belong to the request current in x265's order, are put back.
void CostEstimateGroup::analyseDeferredWeights()
{
    std::vector<Lowres*> fencs, refs;
    std::vector<size_t> which;
    for (size_t i = 0; i < m_batch.size(); i++)
        if (m_batch[i].deferWeights)
        {
            fencs.push_back(m_frames[m_batch[i].b]);
            refs.push_back(m_frames[m_batch[i].p0]);
            which.push_back(i);
        }
    if (which.empty())
        return;
    std::vector<bool> keep(which.size());
    std::vector<double> keepDelta(which.size());
    for (size_t k = 0; k < which.size(); k++)
    {
        keep[k] = fencs[k]->weighted[m_batch[which[k]].d0];
        keepDelta[k] = fencs[k]->weightedCostDelta[m_batch[which[k]].d0];
    }
    int rc = m_tld.weightsAnalyse((int)which.size(), fencs.data(), refs.data(), m_bitDepth);
    if (rc != X265GPU_OK && m_status == X265GPU_OK)
        m_status = rc;
    for (size_t k = 0; k < which.size(); k++)
    {
        Pending& p = m_batch[which[k]];
        Lowres& fenc = *fencs[k];
        const bool weighted = fenc.weighted[p.d0];
        p.req.weightedL0 = weighted;
        if (p.presearch)
        {
            fenc.presearchWeighted[p.d0] = weighted;
            fenc.weighted[p.d0] = keep[k];
            fenc.weightedCostDelta[p.d0] = keepDelta[k];
        }
    }
}

void CostEstimateGroup::finishBatch()
{
    if (m_batch.empty())
        return;
    analyseDeferredWeights();
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
            if (!p.req.bSearchOnly && !p.precost)
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
    m_cuTreeStrength = (p.bHevcAq ? 6.0 : 5.0) * (1.0 - p.qCompress);
}

void CuTree::cuTreeFinish(Lowres& frame, double averageDuration, int ref0Distance)
{
    const int fpsFactor = (int)(clipDuration(averageDuration) / clipDuration(m_frameDuration) * 256);
    double weightdelta = 0.0;
    if (ref0Distance && frame.weightedCostDelta[ref0Distance - 1] > 0)
        weightdelta = 1.0 - frame.weightedCostDelta[ref0Distance - 1];
    if (m_p.bHevcAq)
    {
        // computeCUTreeQpOffset (:3876-4014): every layer's dCuTreeOffset from its dQpOffset
        std::vector<x265gpu_hevc_aq_layer> layers = layerDescriptors(frame);
        note(layers.empty() ? X265GPU_ERR_ARG
                            : x265gpu_cutree_finish_hevc(m_ctx, frame.slot, fpsFactor, weightdelta, m_cuTreeStrength, m_p.qgSize,
                                                         (int32_t)layers.size(), layers.data(),
                                                         m_p.qgSize == 8 ? frame.dCuTreeOffset8x8.data() : nullptr));
    }
    else if (m_p.qgSize == 8)
        note(x265gpu_cutree_finish_qg8(m_ctx, frame.slot, fpsFactor, weightdelta, m_cuTreeStrength, frame.qpAqOffset,
                                       frame.qpCuTreeOffset));
    else
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

// Lookahead::cuTree (slicetype.cpp:3665-3766) with these substitutions: memset of propagateCost ->
// x265gpu_cutree_clear, std::swap of two frames' propagateCost -> x265gpu_cutree_swap, estimateCUPropagate and
// cuTreeFinish -> the members above (x265gpu_cutree_propagate / _finish).  The walk is x265's.
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

    // without a lookahead the previous non-B frame's totals are carried over instead (:3690-3704)
    if (!m_p.lookaheadDepth)
    {
        if (bIntra)
        {
            note(x265gpu_cutree_clear(m_ctx, frames[0]->slot));
            x265gpu_geometry g;
            x265gpu_get_geometry(m_ctx, &g);
            memcpy(frames[0]->qpCuTreeOffset, frames[0]->qpAqOffset, sizeof(double) * g.cuCount * (m_p.qgSize == 8 ? 4 : 1));
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

    CostEstimateGroup estGroup(m_ctx, frames, m_p.bEnableWeightedPred, m_p.bitDepth, m_p.trace);

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


// ---- Lookahead: queue interface and slice-type decision ----
//
// The decision logic below is x265's (slicetype.cpp, lines cited per member) and has to stay so: which cost
// requests are issued, and in which order, decides which motion searches run inside a bidirectional request
// (the skip shortcut of estimateCUCost makes an L0 vector depend on that), so any other order gives other
// vectors.  What differs from x265 is only where a request is computed: CostEstimateGroup sends it to the device.
// Not covered (unsupported() says so): zones, analysis load, --mcstf
// batches, more than two temporal sub-layers, rc.hevcAq, --aq-motion together with qgSize 8.

// x265_exp2fix8 (source/common/common.cpp:96-103) over x265_exp2_lut (source/common/constants.cpp:552-558),
// whose entries are round(256 * (2^(i/64) - 1)); inside x265 this is the function itself
static int exp2fix8(double x)
{
    // filled once, by whichever thread comes first (instances on several host threads share it)
    struct Lut { int v[64]; Lut() { for (int k = 0; k < 64; k++) v[k] = (int)lround(256.0 * (exp2(k / 64.0) - 1.0)); } };
    static const Lut table;
    const int* lut = table.v;
    const int i = (int)(x * (-64.f / 6.f) + 512.5f);
    if (i < 0)
        return 0;
    if (i > 1023)
        return 0xffff;
    return (lut[i & 63] + 256) << (i >> 6) >> 8;
}

namespace {
struct ScopeTimer
{
    double& acc;
    std::chrono::steady_clock::time_point t0;
    explicit ScopeTimer(double& a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~ScopeTimer() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
}

static inline bool isTypeI(int t) { return t == TYPE_I || t == TYPE_IDR; }
static inline bool isTypeB(int t) { return t == TYPE_B || t == TYPE_BREF; }

LookaheadParam::LookaheadParam()
{
    memset(&aq, 0, sizeof(aq));
    aq.aqMode = 2;                 // X265_AQ_AUTO_VARIANCE (param.cpp:310)
    aq.aqStrength = 1.0;
    aq.qgSize = 32;
    aq.bWeightedPred = 1;
}

Lookahead::Lookahead(x265gpu_ctx* ctx, const LookaheadParam& param) : m_ctx(ctx), m_param(param)
{
    x265gpu_get_geometry(ctx, &m_geom);
    const int W = m_geom.widthInCU, H = m_geom.heightInCU;
    m_8x8Blocks = (W > 2 && H > 2) ? W * H + 4 - 2 * (W + H) : W * H;      // :1066
    m_lastKeyframe = -param.keyframeMax;                                   // :1077
    m_fullQueueSize = (size_t)(param.lookaheadDepth > 1 ? param.lookaheadDepth : 1);
    m_bBatchMotionSearch = param.poolWorkers > 0 && param.bFrameAdaptive == 2;   // :1091
    m_bBatchFrameCosts = m_bBatchMotionSearch;                             // :1100
    m_gopId = param.bEnableTemporalSubLayers > 2 ? param.bEnableTemporalSubLayers - 3 : 0;   // :1176-1192
}

int Lookahead::unsupported() const
{
    const LookaheadParam& p = m_param;
    // Encoder::configure (encoder.cpp:4303-4332) leaves 0, 2 or 3..5 layers, and with 3..5 a fixed B-frame count
    // and no adaptive placement; the decision below relies on all three
    const int layers = p.bEnableTemporalSubLayers;
    const bool badLayers = layers < 0 || layers == 1 || layers > 5 ||
                           (layers > 2 && (p.bFrameAdaptive || p.bframes != (4 << (layers - 3)) - 1));
    if (badLayers || p.bframes > X265GPU_BFRAME_MAX || p.lookaheadDepth > LOOKAHEAD_MAX ||
        (p.bframes && p.bframes >= p.lookaheadDepth) ||   // x265_check_params (param.cpp:1811): the decision reads frames[bframes + 1]
        (p.aq.bHevcAq && (p.bAQMotion || (p.maxCUSize != 64 && p.maxCUSize != 32 && p.maxCUSize != 16))) || p.keyframeMin < 1 ||
        // calcMotionAdaptiveQuantFrame indexes the qgSize 8 offset arrays by lowres block (:3648-3658) and changes
        // invQscaleFactor, which no estimate reads then: the device form would install those as the slot's factors
        (p.bAQMotion && p.aq.qgSize == 8) ||
        (p.cuTree && !p.lookaheadDepth))          // Encoder::configure turns cuTree off without a lookahead (encoder.cpp:4103)
        return X265GPU_ERR_UNSUPPORTED;
    return X265GPU_OK;
}

int64_t Lookahead::cost(Lowres** frames, int p0, int p1, int b, bool bIntraPenalty)
{
    if (frames[b]->intraTotalsPending)
        note(resolveIntraTotals(m_ctx, &frames[b], 1));
    // the trellis asks for the same costs thousands of times per decision: answer those from the cache
    // without setting a group up (estimateFrameCost's own first test, slicetype.cpp:4290)
    const int64_t cached = frames[b]->costEst[b - p0][p1 - b];
    if (cached >= 0 && !frames[b]->pending[b - p0][p1 - b])
        return bIntraPenalty ? cached + cached * frames[b]->intraMbs[b - p0] / (m_8x8Blocks * 8) : cached;
    ScopeTimer timer(m_seconds[T_SINGLE]);
    CostEstimateGroup est(m_ctx, frames, m_param.bEnableWeightedPred, m_param.bitDepth, &m_trace);
    const int64_t c = est.singleCost(p0, p1, b, bIntraPenalty);
    note(est.status());
    m_presearchHits += est.presearchHits();
    m_precostHits += est.precostHits();
    return c;
}

void Lookahead::cuTree(Lowres** frames, int numframes, bool bIntra)
{
    ScopeTimer timer(m_seconds[T_CUTREE]);
    note(resolveIntraTotals(m_ctx, frames, numframes + 1));
    note(x265gpu_aq_frames_finish(m_ctx));
    CuTreeParam cp;
    cp.fpsNum = m_param.fpsNum;
    cp.fpsDenom = m_param.fpsDenom;
    cp.bBPyramid = m_param.bBPyramid;
    cp.bEnableWeightedBiPred = m_param.bEnableWeightedBiPred;
    cp.lookaheadDepth = m_param.lookaheadDepth;
    cp.vbvBufferSize = m_param.vbvBufferSize;
    cp.qCompress = m_param.qCompress;
    cp.bEnableWeightedPred = m_param.bEnableWeightedPred;
    cp.bitDepth = m_param.bitDepth;
    cp.qgSize = m_param.bAdaptiveQuant ? m_param.aq.qgSize : 32;
    cp.bHevcAq = m_param.bAdaptiveQuant && m_param.aq.bHevcAq;
    cp.trace = &m_trace;
    note(CuTree(m_ctx, cp).cuTree(frames, numframes, bIntra));
}

// :3323-3454.  Histogram differences of the 4x4 segments between the previous and the current picture against their
// running averages; an abrupt change is a scene change unless the segment means say flash, fade or intensity change.
// The segment size that scales the thresholds grows as the loops go (the last column's / row's remainder is added to
// the running width / height and never taken off again): kept, in the reference's unsigned arithmetic.
bool Lookahead::detectHistBasedSceneChange(Lowres** frames, int p0, int p1, int p2)
{
    const x265gpu_hist_stats_t& prev = *frames[p0]->hist;
    const x265gpu_hist_stats_t& cur = *frames[p1]->hist;
    const x265gpu_hist_stats_t& fut = *frames[p2]->hist;
    frames[p1]->bHistScenecutAnalyzed = true;

    uint32_t abruptChangeCount = 0, sceneChangeCount = 0;
    const uint32_t widthFullRes = (uint32_t)m_geom.sourceWidth, heightFullRes = (uint32_t)m_geom.sourceHeight;
    uint32_t segmentWidth = widthFullRes / 4, segmentHeight = heightFullRes / 4;
    const int64_t var[3][2] = { { cur.picAvgVariance, prev.picAvgVariance }, { cur.picAvgVarianceCb, prev.picAvgVarianceCb },
                                { cur.picAvgVarianceCr, prev.picAvgVarianceCr } };
    for (uint32_t sw = 0; sw < 4; sw++)
        for (uint32_t sh = 0; sh < 4; sh++)
        {
            bool isAbruptChange = false, isSceneChange = false;
            segmentWidth += (sw == 3) ? widthFullRes - 4 * segmentWidth : 0;
            segmentHeight += (sh == 3) ? heightFullRes - 4 * segmentHeight : 0;
            const uint32_t num64 = (segmentWidth * segmentHeight) >> 12;      // NUM64x64INPIC (slicetype.h:64)
            uint32_t thresh[3];
            // PICTURE_DIFF_VARIANCE_TH 390, PICTURE_VARIANCE_TH 1500, HIGH / LOW_VAR_SCENE_CHANGE_TH 3500 / 2250; chroma:
            // 10, 20 and the same two divided by 4 (slicetype.h:50-58)
            thresh[0] = (std::llabs(var[0][0] - var[0][1]) > 390 && (var[0][0] > 1500 || var[0][1] > 1500)) ? 3500 * num64 : 2250 * num64;
            for (int c = 1; c < 3; c++)
                thresh[c] = (std::llabs(var[c][0] - var[c][1]) > 10 && (var[c][0] > 20 || var[c][1] > 20)) ? 3500 / 4 * num64
                                                                                                          : 2250 / 4 * num64;
            uint32_t accHistDiff[3] = { 0, 0, 0 };
            for (int bin = 0; bin < 256; bin++)
                for (int c = 0; c < 3; c++)
                    accHistDiff[c] += (uint32_t)std::abs((int32_t)cur.picHistogram[sw][sh][c][bin] - (int32_t)prev.picHistogram[sw][sh][c][bin]);
            if (m_resetRunningAvg)
                for (int c = 0; c < 3; c++)
                    m_accHistDiffRunningAvg[c][sw][sh] = accHistDiff[c];
            for (int c = 0; c < 3; c++)
            {
                const uint32_t err = (uint32_t)std::abs((int32_t)m_accHistDiffRunningAvg[c][sw][sh] - (int32_t)accHistDiff[c]);
                if (err > thresh[c] && accHistDiff[c] >= err)
                    isAbruptChange = true;
            }
            if (isAbruptChange)
            {
                const int16_t f = (int16_t)fut.averageIntensityPerSegment[sw][sh][0];
                const int16_t c = (int16_t)cur.averageIntensityPerSegment[sw][sh][0];
                const int16_t q = (int16_t)prev.averageIntensityPerSegment[sw][sh][0];
                const uint8_t futurePast = (uint8_t)std::abs(f - q), futurePresent = (uint8_t)std::abs(f - c);
                const uint8_t presentPast = (uint8_t)std::abs(c - q);
                if (futurePresent >= 1.5 * futurePast && presentPast >= 1.5 * futurePast)
                    ;                                       // FLASH_TH: a flash
                else if (futurePresent < 4 && presentPast < 4)
                    ;                                       // FADE_TH: a fade
                else if (std::abs(futurePresent - presentPast) < 4 && futurePresent + presentPast >= futurePast)
                    ;                                       // INTENSITY_CHANGE_TH
                else
                    isSceneChange = true;
            }
            else   // only the luma average follows the picture (:3430)
                m_accHistDiffRunningAvg[0][sw][sh] = (3 * m_accHistDiffRunningAvg[0][sw][sh] + accHistDiff[0]) / 4;
            abruptChangeCount += isAbruptChange;
            sceneChangeCount += isSceneChange;
        }
    m_resetRunningAvg = abruptChangeCount >= m_segmentCountThreshold;
    return sceneChangeCount >= m_segmentCountThreshold;
}

// :3456-3482
bool Lookahead::histBasedScenecut(Lowres** frames, int p0, int p1, int numFrames)
{
    if (m_param.bframes)
    {
        const int maxp1 = std::min(p0 + 1 + m_param.bframes, numFrames);
        for (int cp1 = p0; cp1 < maxp1; cp1++)
        {
            if (frames[cp1 + 1]->bHistScenecutAnalyzed)
                continue;
            if (frames[cp1 + 2] != nullptr && detectHistBasedSceneChange(frames, cp1, cp1 + 1, cp1 + 2))
                frames[cp1 + 1]->bScenecut = true;
        }
    }
    return frames[p1]->bScenecut;
}

// :3581-3607: the frames of the first mini-GOP, in the order and with the references the trellis costed them
void Lookahead::aqMotion(Lowres** frames, bool bIntra)
{
    if (bIntra)
        return;
    ScopeTimer timer(m_seconds[T_CUTREE]);
    int curnonb = 0, lastnonb = 1;
    while (frames[lastnonb]->sliceType != TYPE_P)
        lastnonb++;
    const int bframes = lastnonb - 1;
    note(resolveIntraTotals(m_ctx, frames, lastnonb + 1));          // the frames' offset arrays are complete
    if (m_param.bBPyramid && bframes > 1)
    {
        const int middle = (bframes + 1) / 2;
        for (int i = 1; i < lastnonb; i++)
        {
            const int p0 = i > middle ? middle : curnonb;
            const int p1 = i < middle ? middle : lastnonb;
            if (i != middle)
                calcMotionAdaptiveQuantFrame(frames, p0, p1, i);
        }
        calcMotionAdaptiveQuantFrame(frames, curnonb, lastnonb, middle);
    }
    else
        for (int i = 1; i < lastnonb; i++)
            calcMotionAdaptiveQuantFrame(frames, curnonb, lastnonb, i);
    calcMotionAdaptiveQuantFrame(frames, curnonb, lastnonb, lastnonb);
}

// :3609-3663: the vectors stay on the device (x265gpu_aq_motion_frame); what comes back is one double per block
void Lookahead::calcMotionAdaptiveQuantFrame(Lowres** frames, int p0, int p1, int b)
{
    Lowres& f = *frames[b];
    if (!f.qpAqOffset || !f.qpCuTreeOffset || !f.invQscaleFactor)
    {
        note(X265GPU_ERR_STATE);
        return;
    }
    if (f.qpAqMotionOffset.empty())
        f.qpAqMotionOffset.assign((size_t)m_geom.cuCount, 0.0);
    if (f.costEst[b - p0][p1 - b] < 0)
    {
        // Nobody has estimated this request yet (a B pyramid under fast or no B-adapt gets here before compCostBref
        // asks for it): x265 reads lowresCosts as Lowres::create's malloc left it, zero pages for a frame in first
        // use.  No list used, no displacement, no deviation: the offsets stay as they are.
        std::fill(f.qpAqMotionOffset.begin(), f.qpAqMotionOffset.end(), 0.0);
        return;
    }
    // costs measured ahead of x265's order used the factors as they were: a request that comes now is measured anew
    for (int i = 0; i < MAX_D; i++)
        for (int j = 0; j < MAX_D; j++)
            f.preCost[i][j].valid = false;
    note(x265gpu_aq_motion_frame(m_ctx, f.slot, b - p0, p1 - b, f.qpAqOffset, f.qpCuTreeOffset, f.invQscaleFactor,
                                 f.qpAqMotionOffset.data()));
}

// addPicture + checkLookaheadQueue (:1299-1342); no pool to wake here
void Lookahead::addPicture(Lowres& frame, int sliceType)
{
    if (!m_filled)
    {
        if (!m_param.bframes && !m_param.lookaheadDepth)
            m_filled = true;
        else if (m_inputCount >= m_param.lookaheadDepth + 2 + m_param.bframes)
            m_filled = true;
    }
    frame.sliceType = sliceType;
    frame.bScenecut = false;
    frame.lowresInit = false;
    m_inputQueue.push_back(&frame);
    m_inputCount++;
}

void Lookahead::flush()                      // :1345-1352
{
    m_fullQueueSize = 1;
    m_filled = true;
}

Lowres* Lookahead::getDecidedPicture()       // :1396-1431 with findJob (:1360-1393) run in place
{
    if (!m_filled)
        return nullptr;
    if (m_outputQueue.empty() && m_inputQueue.size() >= m_fullQueueSize)
        slicetypeDecide();
    if (m_outputQueue.empty())
        return nullptr;
    Lowres* out = m_outputQueue.front();
    m_outputQueue.pop_front();
    m_inputCount--;
    return out;
}

// :4068-4145 for qgSize > 8 without hevcAq.  The per-block costs come back from the device once; the row sums
// go to the frame's recalcRowSatds (x265 overwrites rowSatds of the request in place: consumers read
// recalcRowSatds where x265's rate control reads rowSatds after this call)
int64_t Lookahead::frameCostRecalculate(Lowres* const* frames, int p0, int p1, int b)
{
    Lowres& f = *frames[b];
    const int d0 = b - p0, d1 = p1 - b;
    if (f.sliceType == TYPE_B)
        return f.costEstAq[d0][d1];
    const int W = m_geom.widthInCU, H = m_geom.heightInCU;
    std::vector<uint16_t> costs((size_t)W * H);
    int rc = x265gpu_get_costs(m_ctx, f.slot, d0, d1, costs.data(), nullptr, nullptr);
    if (rc != X265GPU_OK)
    {
        note(rc);
        return -1;
    }
    f.recalcRowSatds.assign(H, 0);
    const bool qg8 = m_param.bAdaptiveQuant && m_param.aq.qgSize == 8;
    // rc.hevcAq: the offsets of the layer of minAQDepth stand in for qpCuTreeOffset (:4078-4112)
    const double* qpOffset = m_param.bAdaptiveQuant && m_param.aq.bHevcAq && !f.aqLayers.empty()
                                 ? f.aqLayers.back().dCuTreeOffset.data() : f.qpCuTreeOffset;
    int64_t score = 0;
    const bool tiny = W <= 2 || H <= 2;
    for (int y = H - 1; y >= 0; y--)
        for (int x = W - 1; x >= 0; x--)
        {
            const int i = x + y * W;
            int c = costs[i] & 0x3FFF;                                  // LOWRES_COST_MASK
            double qpAdj;
            if (qg8)
            {
                // the four 8x8 source blocks under the lowres block, rows of maxBlocksInRowFullRes = 2 W entries
                const double* q = qpOffset + x * 2 + y * W * 4;
                qpAdj = (q[0] + q[1] + q[2 * W] + q[2 * W + 1]) / 4;
            }
            else
                qpAdj = qpOffset[i];
            c = (c * exp2fix8(qpAdj) + 128) >> 8;
            f.recalcRowSatds[y] += c;
            if (tiny || (y > 0 && y < H - 1 && x > 0 && x < W - 1))
                score += c;
        }
    return score;
}

int64_t Lookahead::vbvFrameCost(Lowres** frames, int p0, int p1, int b)   // :2830-2844
{
    const int64_t c = cost(frames, p0, p1, b);
    if (!((m_param.bAdaptiveQuant && m_param.aq.aqMode) || m_param.bAQMotion))
        return c;
    return m_param.cuTree ? frameCostRecalculate(frames, p0, p1, b) : frames[b]->costEstAq[b - p0][p1 - b];
}

// :2753-2828.  Fills plannedSatd / plannedType of the next non-B frame (and of the B frames of the current
// mini-GOP) with the costs of the frames that follow in coded order.
void Lookahead::vbvLookahead(Lowres** frames, int numFrames, bool keyframe)
{
    auto ensure = [](Lowres* f) {
        if (f->plannedSatd.empty())
        {
            f->plannedSatd.assign(LOOKAHEAD_MAX + 1, 0);
            f->plannedType.assign(LOOKAHEAD_MAX + 1, TYPE_AUTO);
        }
    };
    auto plan = [&](Lowres* f, int at, int64_t satd, int type) {
        ensure(f);
        f->plannedSatd[at] = satd;
        f->plannedType[at] = type;
    };
    auto nextNonBAfter = [&](int from) {
        int k = from;
        while (k <= numFrames && isTypeB(frames[k]->sliceType))
            k++;
        return k;
    };
    int prevNonB = 0, idx = 0;
    int curNonB = 1;
    while (curNonB < numFrames && isTypeB(frames[curNonB]->sliceType))
        curNonB++;
    const int nextNonB = keyframe ? prevNonB : curNonB;
    const int miniGopEnd = nextNonB;
    const int nextB = prevNonB + 1;
    const int curBRef = (m_param.bBPyramid && curNonB - prevNonB > 1) ? (prevNonB + curNonB + 1) / 2 : 0;
    int nextBRef = 0;
    for (; curNonB <= numFrames; prevNonB = curNonB, curNonB = nextNonBAfter(curNonB + 1))
    {
        if (nextNonB != curNonB)
        {
            // the P / I frame that closes this mini-GOP (not nextNonB itself)
            const int p0 = isTypeI(frames[curNonB]->sliceType) ? curNonB : prevNonB;
            const int64_t satd = vbvFrameCost(frames, p0, curNonB, curNonB);
            const int type = frames[curNonB]->sliceType;
            plan(frames[nextNonB], idx, satd, type);
            if (curNonB > miniGopEnd)
                for (int j = nextB; j < miniGopEnd; j++)
                    plan(frames[j], frames[j]->indB++, satd, type);
            idx++;
        }
        if (m_param.bBPyramid && curNonB - prevNonB > 1)
            nextBRef = (prevNonB + curNonB + 1) / 2;
        for (int i = prevNonB + 1; i < curNonB; i++, idx++)
        {
            int type = TYPE_B;
            int64_t satd;
            if (!nextBRef)
                satd = vbvFrameCost(frames, prevNonB, curNonB, i);
            else if (i == nextBRef)
            {
                satd = vbvFrameCost(frames, prevNonB, curNonB, nextBRef);
                type = TYPE_BREF;
            }
            else if (i < nextBRef)
                satd = vbvFrameCost(frames, prevNonB, nextBRef, i);
            else
                satd = vbvFrameCost(frames, nextBRef, curNonB, i);
            plan(frames[nextNonB], idx, satd, type);
            for (int j = nextB; j < miniGopEnd; j++)
            {
                if (curBRef && curBRef == i)
                    break;
                if (j >= i && j != nextBRef)
                    continue;
                plan(frames[j], frames[j]->indB++, satd, type);
            }
        }
    }
    ensure(frames[nextNonB]);
    frames[nextNonB]->plannedType[idx] = TYPE_AUTO;          // end of the plan
}

// :3281-3321
bool Lookahead::scenecutInternal(Lowres** frames, int p0, int p1, bool bRealScenecut)
{
    Lowres& frame = *frames[p1];
    cost(frames, p0, p1, p1);
    const int64_t icost = frame.costEst[0][0];
    const int64_t pcost = frame.costEst[p1 - p0][0];
    const int gopSize = (frame.frameNum - m_lastKeyframe) % m_param.keyframeMax;
    // single precision thresholds, double bias: the mix is x265's and decides borderline frames
    const float threshMax = (float)(m_param.scenecutThreshold / 100.0);
    float threshMin = (float)(threshMax * 0.25);
    double bias = m_param.scenecutBias;
    if (bRealScenecut)
    {
        const int kmin = m_param.keyframeMin, kmax = m_param.keyframeMax;
        if (kmin == kmax)
            threshMin = threshMax;
        if (gopSize <= kmin / 4 || m_param.bIntraRefresh)
            bias = threshMin / 4;
        else if (gopSize <= kmin)
            bias = threshMin * gopSize / kmin;
        else
            bias = threshMin + (threshMax - threshMin) * (gopSize - kmin) / (kmax - kmin);
    }
    const bool res = pcost >= (1.0 - bias) * icost;
    if (res && bRealScenecut)
        frame.bScenecut = true;
    return res;
}

// :3186-3279
bool Lookahead::scenecut(Lowres** frames, int p0, int p1, bool bRealScenecut, int numFrames)
{
    const int bf = m_param.bframes;
    if (bRealScenecut && bf)
    {
        // look past p1 so that a flash shorter than a mini-GOP is not taken for a cut
        const int maxp1 = std::min(p0 + 1 + bf, numFrames);
        bool fluctuate = false, noScenecuts = false;
        int64_t avgSatdCost = frames[p0]->costEst[p1 - p0][0] > -1 ? frames[p0]->costEst[p1 - p0][0] : 0;
        int cnt = 1;
        for (int cp1 = p1; cp1 <= maxp1; cp1++, cnt++)
        {
            if (!scenecutInternal(frames, p0, cp1, false))
            {
                for (int i = cp1; i > p0; i--)
                    frames[i]->bScenecut = false;
                if (cp1 > p0)
                    noScenecuts = false;
            }
            else if (scenecutInternal(frames, cp1 - 1, cp1, false))
            {
                frames[cp1]->bScenecut = true;
                noScenecuts = true;
            }
            avgSatdCost += frames[cp1]->costEst[cp1 - p0][0];
        }
        if (noScenecuts)
        {
            avgSatdCost /= cnt;
            for (int i = p1; i <= maxp1; i++)
            {
                const int64_t cur = frames[i]->costEst[i - p0][0];
                const int64_t prev = frames[i - 1]->costEst[i - 1 - p0][0];
                if (fabs((double)(cur - avgSatdCost)) > 0.1 * avgSatdCost || fabs((double)(cur - prev)) > 0.1 * prev)
                {
                    fluctuate = true;
                    if (!m_isSceneTransition && frames[i]->bScenecut)
                    {
                        // only the first change of a transition counts as the cut
                        m_isSceneTransition = true;
                        for (int j = i + 1; j <= maxp1; j++)
                            frames[j]->bScenecut = false;
                        break;
                    }
                }
                frames[i]->bScenecut = false;
            }
        }
        if (!fluctuate && !noScenecuts)
            m_isSceneTransition = false;
    }
    if (!frames[p1]->bScenecut && bf)
        return false;
    return scenecutInternal(frames, p0, p1, bRealScenecut);
}

// :3534-3579.  path[k] is the type of frame k + 1.
int64_t Lookahead::slicetypePathCost(Lowres** frames, const std::string& path, int64_t threshold)
{
    int64_t total = 0;
    int cur_p = 0;
    const int n = (int)path.size();
    for (int loc = 1; loc <= n;)
    {
        int next_p = loc;
        while (path[next_p - 1] != 'P')
            next_p++;
        total += cost(frames, cur_p, next_p, next_p);
        if (total > threshold)
            break;
        if (m_param.bBPyramid && next_p - cur_p > 2)
        {
            const int middle = cur_p + (next_p - cur_p) / 2;
            total += cost(frames, cur_p, next_p, middle);
            for (int b = loc; b < middle && total < threshold; b++)
                total += cost(frames, cur_p, middle, b);
            for (int b = middle + 1; b < next_p && total < threshold; b++)
                total += cost(frames, middle, next_p, b);
        }
        else
            for (int b = loc; b < next_p && total < threshold; b++)
                total += cost(frames, cur_p, next_p, b);
        loc = next_p + 1;
        cur_p = next_p;
    }
    return total;
}

// :3484-3511.  bestPaths[k % (BFRAME_MAX + 1)] holds the best type string for the first k frames.
void Lookahead::slicetypePath(Lowres** frames, int length, std::vector<std::string>& bestPaths)
{
    const int ring = X265GPU_BFRAME_MAX + 1;
    const int numPaths = std::min(m_param.bframes + 1, length);
    int64_t bestCost = 1LL << 62;
    std::string best;
    for (int nb = 0; nb < numPaths; nb++)
    {
        const int len = length - (nb + 1);
        std::string cand = bestPaths[len % ring].substr(0, len);
        cand.resize(len, '\0');
        cand.append(nb, 'B');
        cand.push_back('P');
        const int64_t c = slicetypePathCost(frames, cand, bestCost);
        if (c < bestCost)
        {
            bestCost = c;
            best.swap(cand);
        }
    }
    bestPaths[length % ring] = best;
}

// :2846-3184
void Lookahead::slicetypeAnalyse(Lowres** frames, bool bKeyframe)
{
    const LookaheadParam& P = m_param;
    const int maxSearch = std::min(P.lookaheadDepth, LOOKAHEAD_MAX);
    const bool bIsVbvLookahead = P.vbvBufferSize && P.lookaheadDepth;

    int framecnt = 0;                              // undecided frames ahead
    while (framecnt < maxSearch && frames[framecnt + 1] && frames[framecnt + 1]->sliceType == TYPE_AUTO)
        framecnt++;
    if (!framecnt)
    {
        if (P.cuTree)
            cuTree(frames, 0, bKeyframe);
        return;
    }
    frames[framecnt + 1] = nullptr;

    const int keyFrameLimit = P.keyframeMax + m_lastKeyframe - frames[0]->frameNum - 1;
    const bool nearKeyint = P.gopLookahead && keyFrameLimit <= P.bframes + 1;
    int keyintLimit = nearKeyint ? keyFrameLimit + P.gopLookahead : keyFrameLimit;
    int numFrames = P.bIntraRefresh ? framecnt : std::min(framecnt, keyintLimit);
    const int origNumFrames = numFrames;

    if (bIsVbvLookahead)
        numFrames = framecnt;
    else if (P.bOpenGOP && numFrames < framecnt)
        numFrames++;
    else if (numFrames == 0)
    {
        frames[1]->sliceType = TYPE_I;
        return;
    }

    if (m_bBatchMotionSearch)
    {
        // every motion search the trellis can ask for, in one device batch (:2920-2947) ...
        CostEstimateGroup est(m_ctx, frames, P.bEnableWeightedPred, P.bitDepth, &m_trace);
        for (int b = 2; b < numFrames; b++)
            for (int d = 1; d <= P.bframes + 1; d++)
            {
                if (b - d < 0 || frames[b]->searched[0][d])
                    continue;
                const bool both = b + d < numFrames && !frames[b]->searched[1][d];
                est.add(b - d, both ? b + d : b, b);
            }
        if (P.bPresearch)
            presearchEdges(est, frames, numFrames);
        m_bBatchMotionSearch = m_bBatchMotionSearch && P.poolWorkers >= 4;
        {
            ScopeTimer timer(m_seconds[T_BATCH1]);
            est.finishBatch();
        }
        if (m_bBatchFrameCosts)
        {
            // ... then every frame cost those searches allow (:2949-2988)
            for (int b = 2; b < numFrames; b++)
                for (int d0 = 1; d0 <= P.bframes + 1; d0++)
                {
                    if (b < d0 || !frames[b]->searched[0][d0])
                        continue;
                    for (int d1 = 0; d1 <= P.bframes && b + d1 < numFrames; d1++)
                    {
                        if (d1 && !frames[b]->searched[1][d1])
                            continue;
                        if (frames[b]->costEst[d0][d1] >= 0)
                            continue;
                        est.add(b - d0, b + d1, b);
                    }
                }
            m_bBatchFrameCosts = m_bBatchFrameCosts && P.poolWorkers > 12;
            if (P.bPresearch)
                precostEdges(est, numFrames);
            ScopeTimer timer(m_seconds[T_BATCH2]);
            est.finishBatch();
        }
        note(est.status());
    }
    note(resolveIntraTotals(m_ctx, frames, numFrames + 1));

    int numBFrames = 0, numAnalyzed = numFrames, resetStart;
    const bool isScenecut = P.bHistBasedSceneCut ? histBasedScenecut(frames, 0, 1, origNumFrames)
                                                 : scenecut(frames, 0, 1, true, origNumFrames);
    if (P.scenecutThreshold && isScenecut)
    {
        frames[1]->sliceType = TYPE_I;
        return;
    }
    const bool gopExtension = P.gopLookahead && keyFrameLimit >= 0 && keyFrameLimit <= P.bframes + 1;
    if (gopExtension)
    {
        // a cut shortly after the keyframe that is due pushes that keyframe back to the cut (:3006-3026)
        const bool keep = m_isSceneTransition;
        m_extendGopBoundary = false;
        for (int i = P.bframes + 1; i < origNumFrames && !m_extendGopBoundary; i += P.bframes + 1)
        {
            scenecut(frames, i, i + 1, true, origNumFrames);
            for (int j = i + 1; j <= std::min(i + P.bframes + 1, origNumFrames); j++)
                if (frames[j]->bScenecut && scenecutInternal(frames, j - 1, j, true))
                {
                    m_extendGopBoundary = true;
                    break;
                }
        }
        m_isSceneTransition = keep;
    }

    if (P.bframes)
    {
        if (P.bFrameAdaptive == 2)
        {
            if (numFrames > 1)
            {
                std::vector<std::string> bestPaths(X265GPU_BFRAME_MAX + 1);
                bestPaths[1] = "P";
                for (int len = 2; len <= numFrames; len++)
                    slicetypePath(frames, len, bestPaths);
                const std::string& best = bestPaths[numFrames % (X265GPU_BFRAME_MAX + 1)];
                numBFrames = 0;
                while (numBFrames < (int)best.size() && best[numBFrames] == 'B')
                    numBFrames++;
                for (int j = 1; j < numFrames; j++)
                    frames[j]->sliceType = best[j - 1] == 'B' ? TYPE_B : TYPE_P;
            }
            frames[numFrames]->sliceType = TYPE_P;
        }
        else if (P.bFrameAdaptive == 1)
        {
            // pairwise comparison, then extension of the B run while P frames stay cheap (:3048-3098)
            const int interThresh = 300, pSensBias = 50 - P.bFrameBias;
            for (int i = 0; i <= numFrames - 2;)
            {
                const int64_t cost2p1 = cost(frames, i, i + 2, i + 2, true);
                if (frames[i + 2]->intraMbs[2] > m_8x8Blocks / 2)
                {
                    frames[i + 1]->sliceType = frames[i + 2]->sliceType = TYPE_P;
                    i += 2;
                    continue;
                }
                const int64_t cost1b1 = cost(frames, i, i + 2, i + 1);
                const int64_t cost1p0 = cost(frames, i, i + 1, i + 1);
                const int64_t cost2p0 = cost(frames, i + 1, i + 2, i + 2);
                if (cost1p0 + cost2p0 < cost1b1 + cost2p1)
                {
                    frames[i + 1]->sliceType = TYPE_P;
                    i += 1;
                    continue;
                }
                frames[i + 1]->sliceType = TYPE_B;
                int j = i + 2;
                for (; j <= std::min(i + P.bframes, numFrames - 1); j++)
                {
                    const int64_t pthresh = std::max(interThresh - pSensBias * (j - i - 1), interThresh / 10);
                    const int64_t pcost = cost(frames, i, j + 1, j + 1, true);
                    if (pcost > pthresh * m_8x8Blocks || frames[j + 1]->intraMbs[j - i + 1] > m_8x8Blocks / 3)
                        break;
                    frames[j]->sliceType = TYPE_B;
                }
                frames[j]->sliceType = TYPE_P;
                i = j;
            }
            frames[numFrames]->sliceType = TYPE_P;
            numBFrames = 0;
            while (numBFrames < numFrames && frames[numBFrames + 1]->sliceType == TYPE_B)
                numBFrames++;
        }
        else
        {
            numBFrames = std::min(numFrames - 1, P.bframes);
            for (int j = 1; j < numFrames; j++)
                frames[j]->sliceType = (j % (numBFrames + 1)) ? TYPE_B : TYPE_P;
            frames[numFrames]->sliceType = TYPE_P;
        }

        const bool bForceRADL = P.radl && P.keyframeMax == P.keyframeMin;
        const bool bLastMiniGop = framecnt < P.bframes + 1;
        const int preRADL = m_lastKeyframe + P.keyframeMax - P.radl - 1;     // frame before the RADL pictures
        if (bForceRADL && frames[0]->frameNum == preRADL && !bLastMiniGop)
        {
            numBFrames = P.radl;
            for (int j = 1; j <= numBFrames; j++)
                frames[j]->sliceType = TYPE_B;
            frames[numBFrames + 1]->sliceType = TYPE_I;
        }
        else
            for (int j = 1; j < numBFrames + 1; j++)
                if (scenecut(frames, j, j + 1, false, origNumFrames) || (bForceRADL && frames[j]->frameNum == preRADL))
                {
                    frames[j]->sliceType = TYPE_P;
                    numAnalyzed = j;
                    break;
                }
        resetStart = bKeyframe ? 1 : std::min(numBFrames + 2, numAnalyzed + 1);
    }
    else
    {
        for (int j = 1; j <= numFrames; j++)
            frames[j]->sliceType = TYPE_P;
        resetStart = bKeyframe ? 1 : 2;
    }

    if (P.bAQMotion)
        aqMotion(frames, bKeyframe);

    if (P.cuTree)
        cuTree(frames, std::min(numFrames, P.keyframeMax), bKeyframe);

    if (gopExtension && !m_extendGopBoundary)
        keyintLimit = keyFrameLimit;
    if (!P.bIntraRefresh)
        for (int j = keyintLimit + 1; j <= numFrames; j += P.keyframeMax)
        {
            frames[j]->sliceType = TYPE_I;
            resetStart = std::min(resetStart, j + 1);
        }

    if (bIsVbvLookahead)
        vbvLookahead(frames, numFrames, bKeyframe);

    // frames beyond the first mini-GOP go back to undecided (:3163-3171)
    const int maxp1 = std::min(P.bframes + 1, origNumFrames);
    for (int j = resetStart; j <= numFrames; j++)
    {
        frames[j]->sliceType = TYPE_AUTO;
        if (j <= maxp1 && frames[j]->bScenecut && m_isSceneTransition)
            m_isSceneTransition = false;
    }
}

// Not in x265: the searches its two batch loops leave out, queued with batch 1.  The loops start at frame 2 and stop
// before frame numFrames, take a list-1 search only towards a frame before numFrames, and only together with the
// list-0 search of the same distance; what is left x265 runs one request at a time when the scene-cut test and
// the trellis ask, each a whole wavefront over the frame by itself.
// On the device such a lone search is latency (one launch per anti-diagonal), while inside the batch it costs
// almost nothing.  Each is queued exactly as the single request would run it (list, distance, P or bidirectional,
// cooperative slices), kept aside, and handed to that request when it comes, so decisions, vectors and costs are
// the ones x265 gets; a search nobody asks for stays invisible (the host's `searched` flags say what x265's
// sentinel would).
void Lookahead::presearchEdges(CostEstimateGroup& est, Lowres** /*frames*/, int numFrames)
{
    const int bf = m_param.bframes;
    // the first frame: scenecut() always asks for (0, 1, 1) first, a P request
    est.addPresearch(0, 1, 1, 1);
    // the last frame of the window as the P frame that closes a path of the trellis
    for (int d = 1; d <= bf + 1 && numFrames - d >= 0; d++)
        est.addPresearch(numFrames - d, numFrames, numFrames, 1);
    // Every list-1 search inside the window that batch 1 did not queue: those of frame 1, those towards the last
    // frame, and those the batch loop passes over because the list-0 search of the same distance exists already
    // (:2934-2935; typically frames that were the window's last frames a decision ago) or would reach before the
    // window.  A list-1 search always belongs to a bidirectional request, whatever that request's p0.
    // (addPresearch drops what is searched or queued already.)
    for (int b = 1; b < numFrames; b++)
        for (int d = 1; d <= bf && b + d <= numFrames; d++)
            est.addPresearch(b - 1, b + d, b, 2);
}

// Not in x265: with the frame costs of batch 2, the cost of every other request of the window whose vectors are on
// the device by now (the window's first and last frames, and whatever the searches that ran ahead made possible).
// x265 asks for these one at a time, each a round trip to the device here; addPrecost keeps the answers aside.
void Lookahead::precostEdges(CostEstimateGroup& est, int numFrames)
{
    const int bf = m_param.bframes;
    for (int b = 1; b <= numFrames; b++)
        for (int d0 = 1; d0 <= bf + 1 && b - d0 >= 0; d0++)
            for (int d1 = 0; d1 <= bf && b + d1 <= numFrames; d1++)
                est.addPrecost(b - d0, b + d1, b);
}

void Lookahead::placeBref(Lowres** list, int start, int end, int num, int& brefs)      // :1864-1886
{
    const int avg = (start + end) / 2;
    if (m_param.bEnableTemporalSubLayers >= 2 && num <= 2)
        return;
    list[avg]->sliceType = TYPE_BREF;
    brefs++;
    if (m_param.bEnableTemporalSubLayers >= 2)
    {
        placeBref(list, start, avg, avg - start, brefs);
        placeBref(list, avg + 1, end, end - avg, brefs);
    }
}

void Lookahead::compCostBref(Lowres** frames, int start, int end, int num)              // :1889-1908
{
    if (num <= 2)
    {
        for (int i = start; i < end; i++)
            cost(frames, start, end + 1, i + 1);
        return;
    }
    const int avg = (start + end) / 2;
    cost(frames, start, end + 1, avg + 1);
    compCostBref(frames, start, avg, avg - start);
    compCostBref(frames, avg + 1, end, end - avg);
}

// :1982-2751 without zones, analysis load and MCSTF
// :2042-2088.  A run of pictures whose variance does not fall is a fade-in; the picture that ends a run of at least
// a second is marked, and the fix-ups of the mini-GOP make it a keyframe.  The ring index over POC modulo the array
// length, its wrap inside the loop and the use of list[endIndex] are the reference's.
void Lookahead::detectFades(Lowres** list)
{
    const int length = X265GPU_BFRAME_MAX + 4;
    int j, endIndex = 0;
    for (j = 0; j < length; j++)
        m_frameVariance[j] = -1;
    for (j = 0; list[j] != nullptr; j++)
        m_frameVariance[list[j]->frameNum % length] = list[j]->frameVariance;
    const int first = list[0]->frameNum, last = list[j - 1]->frameNum;
    for (int k = first % length; k <= last % length; k++)
    {
        if (m_frameVariance[k] == -1)
            break;
        const double before = k > 0 ? m_frameVariance[k - 1] : m_frameVariance[length - 1];
        if (m_frameVariance[k] >= before)
        {
            m_isFadeIn = true;
            if (m_fadeCount == 0 && m_fadeStart == -1)
                for (int poc = first; poc <= last; poc++)
                    if (k == poc % length)
                    {
                        m_fadeStart = poc ? poc - 1 : 0;
                        break;
                    }
            m_fadeCount = list[endIndex]->frameNum > m_fadeStart ? list[endIndex]->frameNum - m_fadeStart : 0;
            endIndex++;
        }
        else
        {
            if (m_isFadeIn && m_fadeCount >= (uint64_t)(m_param.fpsNum / m_param.fpsDenom))
                for (int t = 0; list[t] != nullptr; t++)
                    if (list[t]->frameNum == m_fadeStart + (int)m_fadeCount)
                    {
                        list[t]->bIsFadeEnd = true;
                        break;
                    }
            m_isFadeIn = false;
            m_fadeCount = 0;
            m_fadeStart = -1;
        }
        if (k == length - 1)
            k = -1;
    }
}

// The pictures that leave the window: off the input queue before they enter the output queue (:2672-2685)
void Lookahead::dequeue(int count, int& maxSearch)
{
    m_inputQueue.erase(m_inputQueue.begin(), m_inputQueue.begin() + count);
    maxSearch -= count;
}

// :2607-2716, the mini-GOP of list[0..bframes] without the temporal-layer orders
void Lookahead::finishMiniGop(Lowres** frames, Lowres** list, int bframes, int& brefs, int& maxSearch)
{
    const LookaheadParam& P = m_param;
    if (bframes)
        list[bframes - 1]->bLastMiniGopBFrame = true;
    list[bframes]->leadingBframes = bframes;
    m_lastNonB = list[bframes];

    if (P.bBPyramid && bframes > 1 && !brefs)
        placeBref(list, 0, bframes, bframes + 1, brefs);

    // the costs rate control will ask for, while the frames are still in the window (:2620-2670)
    if (!P.bCqp)
    {
        if (!maxSearch)
            for (int i = 0; i <= bframes; i++)
                frames[i + 1] = list[i];
        const int nonB = bframes + 1;
        cost(frames, isTypeI(frames[nonB]->sliceType) ? nonB : 0, nonB, nonB);
        if (P.bEnableTemporalSubLayers > 1 && bframes)
            compCostBref(frames, 0, bframes, bframes + 1);
        else
        {
            int p0 = 0;
            bool p0Available = frames[nonB]->sliceType != TYPE_IDR;
            for (int b = 1; b <= bframes; b++)
            {
                if (!p0Available)
                    p0 = b;
                int p1 = nonB;
                if (frames[b]->sliceType == TYPE_B)
                    for (p1 = b; frames[p1]->sliceType == TYPE_B; p1++)
                        ;                                       // next non-B or B-ref
                cost(frames, p0, p1, b);
                if (frames[b]->sliceType == TYPE_BREF)
                {
                    p0 = b;
                    p0Available = true;
                }
            }
        }
    }

    // hand the mini-GOP over in coded order: the non-B frame, B-refs, then the plain B frames (:2672-2715)
    dequeue(bframes + 1, maxSearch);
    m_outputQueue.push_back(list[bframes]);
    for (int pass = 0; pass < 2; pass++)
        for (int i = 0; i < bframes; i++)
            if ((list[i]->sliceType == TYPE_BREF) == (pass == 0) && (pass == 1 || brefs))
                m_outputQueue.push_back(list[i]);
}

// x265_gop_ra (source/x265.h:835-): the coding order of a hierarchical mini-GOP of 4, 8 or 16 pictures as (POC offset,
// temporal layer) pairs: the closing picture first, then of every interval its middle before its two halves, one
// layer down per halving
namespace {
struct GopEntry { int pocOffset, layer; };
const int kGopLength[3] = { 4, 8, 16 };

void fillGopOrder(GopEntry* out, int& n, int lo, int hi, int layer)
{
    if (hi - lo < 2)
        return;
    const int mid = (lo + hi) / 2;
    out[n++] = { mid, layer };
    fillGopOrder(out, n, lo, mid, layer + 1);
    fillGopOrder(out, n, mid, hi, layer + 1);
}

const GopEntry* gopOrder(int gopId)
{
    static GopEntry table[3][16];
    static bool built = false;
    if (!built)
    {
        for (int g = 0; g < 3; g++)
        {
            int n = 0;
            table[g][n++] = { kGopLength[g], 0 };
            fillGopOrder(table[g], n, 0, kGopLength[g], 1);
        }
        built = true;
    }
    return table[gopId];
}
}

// The costs rate control will ask for of the sub-GOP list[first..nonB] (:2339-2359, 2424-2444, 2510-2528)
void Lookahead::subGopCosts(Lowres** frames, int first, int nonB)
{
    if (m_param.bCqp)
        return;
    // (the reference's refill of frames[] for a lookahead without a window cannot run: the window holds at least the
    // pictures still to hand over)
    const int b = nonB + 1;
    cost(frames, isTypeI(frames[b]->sliceType) ? b : first, b, b);
    if (nonB)
        compCostBref(frames, first, nonB, nonB + 1);
}

// :2304-2571.  A full mini-GOP leaves in the hierarchical order of its length; a shorter one (a keyframe or the end
// of the stream cut it) is split into the longest orders that fit, and what is left over (under four pictures)
// leaves as a plain mini-GOP.  The indices that stay relative to the whole list (leadingBframes, the count handed
// to placeBref and compCostBref) are the reference's.
void Lookahead::finishTemporalMiniGop(Lowres** frames, Lowres** list, int bframes, int& brefs, int& maxSearch)
{
    const LookaheadParam& P = m_param;
    auto closeSubGop = [&](int first, int nonB)
    {
        Lowres& last = *list[nonB];
        if (!isTypeI(last.sliceType))
            last.sliceType = TYPE_P;
        if (nonB)
            list[nonB - 1]->bLastMiniGopBFrame = true;
        last.leadingBframes = nonB;
        m_lastNonB = &last;
        (void)first;
    };
    auto hierarchical = [&](int first, int nonB, int gopId)
    {
        const GopEntry* order = gopOrder(gopId);
        Lowres& last = *list[nonB];
        last.gopOffset = 0;
        last.gopId = gopId;
        last.tempLayer = order[0].layer;
        m_outputQueue.push_back(&last);
        // the B pictures in coding order.  (The reference stores the mini-GOP's id into the list's last picture
        // here, :2393, and leaves the B pictures' own field as it was; every picture carries its id here.)
        for (int j = 1; j < kGopLength[gopId]; j++)
        {
            Lowres& f = *list[first + order[j].pocOffset - 1];
            f.gopOffset = j;
            f.gopId = gopId;
            f.tempLayer = order[j].layer;
            m_outputQueue.push_back(&f);
        }
    };

    if (bframes == P.bframes)
    {
        list[bframes - 1]->bLastMiniGopBFrame = true;
        list[bframes]->leadingBframes = bframes;
        m_lastNonB = list[bframes];
        if (P.bBPyramid && !brefs)
            placeBref(list, 0, bframes, bframes + 1, brefs);
        subGopCosts(frames, 0, bframes);
        dequeue(bframes + 1, maxSearch);
        hierarchical(0, bframes, m_gopId);
        return;
    }

    int leftOver = bframes + 1, first = 0;
    int gopId = m_gopId - 1;
    while (gopId >= 0 && leftOver > 3)
    {
        const int gopLen = kGopLength[gopId];
        if (leftOver < gopLen)
        {
            gopId--;
            continue;
        }
        const int nonB = first + gopLen - 1;
        closeSubGop(first, nonB);
        if (P.bBPyramid && nonB)
            placeBref(list, first, nonB, nonB + 1, brefs);
        subGopCosts(frames, first, nonB);
        dequeue(gopLen, maxSearch);
        hierarchical(first, nonB, gopId);
        first += gopLen;
        leftOver -= gopLen;
        gopId--;
    }
    if (leftOver > 0 && leftOver < 4)
    {
        const int nonB = first + leftOver - 1;
        closeSubGop(first, nonB);
        if (P.bBPyramid && nonB - first > 1)
            placeBref(list, first, nonB, nonB + 1, brefs);
        subGopCosts(frames, first, nonB);
        dequeue(leftOver, maxSearch);
        // the non-B picture, the B-refs (if the list holds any at all), then the plain B pictures, one layer up
        for (int pass = 0; pass < 3; pass++)
            for (int i = pass ? first : nonB; i < (pass ? nonB : nonB + 1); i++)
            {
                Lowres& f = *list[i];
                const bool isBref = f.sliceType == TYPE_BREF;
                if ((pass == 1 && !(brefs && isBref)) || (pass == 2 && isBref))
                    continue;
                f.gopOffset = 0;
                f.gopId = -1;
                f.tempLayer = pass == 2 ? 1 : 0;
                m_outputQueue.push_back(&f);
            }
    }
}

void Lookahead::slicetypeDecide()
{
    const LookaheadParam& P = m_param;
    ScopeTimer decideTimer(m_seconds[T_DECIDE]);
    m_decideCalls++;
    std::vector<Lowres*> framesV(LOOKAHEAD_MAX + X265GPU_BFRAME_MAX + 4, nullptr);
    std::vector<Lowres*> listV(X265GPU_BFRAME_MAX + 4, nullptr);
    Lowres** frames = framesV.data();
    Lowres** list = listV.data();
    int maxSearch = std::max(1, std::min(P.lookaheadDepth, LOOKAHEAD_MAX));

    const int queued = (int)m_inputQueue.size();
    for (int j = 0; j < P.bframes + 2 && j < queued; j++)
        list[j] = m_inputQueue[j];
    frames[0] = m_lastNonB;
    maxSearch = std::min(maxSearch, queued);
    std::vector<Lowres*> pre;
    for (int j = 0; j < maxSearch; j++)
    {
        frames[j + 1] = m_inputQueue[j];
        if (!frames[j + 1]->lowresInit)
            pre.push_back(frames[j + 1]);
    }

    // pre-analysis of the frames that entered the window: PreLookaheadGroup::processTasks (:2033-2040)
    if (!pre.empty())
    {
        ScopeTimer timer(m_seconds[T_PRE]);
        PreLookaheadGroup group(m_ctx);
        std::vector<Picture> pics(pre.size());
        for (size_t i = 0; i < pre.size(); i++)
            pics[i] = pre[i]->pic;
        if (P.bAdaptiveQuant && P.aq.bHevcAq)
            for (Lowres* f : pre)
                if (f->aqLayers.empty())
                    f->createAqLayers(m_geom, P.maxCUSize, P.aq.qgSize);
        if (P.bAdaptiveQuant)
            note(group.processTasks((int)pre.size(), pre.data(), pics.data(), P.aq, nullptr));
        else
        {
            if (pics[0].memKind != X265GPU_MEM_HOST)
                note(X265GPU_ERR_UNSUPPORTED);     // the plain form takes host planes
            std::vector<const void*> lumas(pre.size());
            std::vector<intptr_t> strides(pre.size());
            for (size_t i = 0; i < pre.size(); i++)
            {
                lumas[i] = pics[i].y;
                strides[i] = pics[i].strideY;
            }
            note(group.processTasks((int)pre.size(), pre.data(), lumas.data(), strides.data(), nullptr));
        }
        if (P.bHistBasedSceneCut)
            for (size_t i = 0; i < pre.size(); i++)
            {
                // collectPictureStatistics (:1809-1834, the task's second step at :1852-1853) on the device
                if (!pre[i]->hist)
                    pre[i]->hist = std::make_shared<x265gpu_hist_stats_t>();
                note(x265gpu_hist_stats(m_ctx, pre[i]->slot, pics[i].y, pics[i].strideY, pics[i].u, pics[i].v, pics[i].strideC,
                                        pics[i].memKind, pre[i]->hist.get()));
                pre[i]->bHistScenecutAnalyzed = false;
            }
        for (Lowres* f : pre)
            f->lowresInit = true;
    }

    if (P.bEnableFades)
        detectFades(list);

    if (m_lastNonB && ((P.bFrameAdaptive && P.bframes) || P.cuTree || P.scenecutThreshold || P.bHistBasedSceneCut ||
                       (P.lookaheadDepth && P.vbvBufferSize)))
        slicetypeAnalyse(frames, false);

    // fix-ups of the first mini-GOP: forced types, keyframe interval, B-frame limit (:2111-2223)
    const bool isClosedGopRadl = P.radl && P.keyframeMax != P.keyframeMin;
    int bframes = 0, brefs = 0;
    for (;; bframes++)
    {
        Lowres& frm = *list[bframes];
        if (frm.sliceTypeReq != TYPE_AUTO && frm.sliceTypeReq != frm.sliceType)
            frm.sliceType = frm.sliceTypeReq;
        if (frm.sliceType == TYPE_BREF)
        {
            // a B-ref needs the pyramid, and with several of them enough references to keep the P frame
            const bool noPyramid = !P.bBPyramid && brefs == (int)P.bBPyramid;
            const bool fewRefs = P.bBPyramid && brefs && P.maxNumReferences <= brefs + 3;
            if (noPyramid || fewRefs)
                frm.sliceType = TYPE_B;
        }
        const int sinceKey = frm.frameNum - m_lastKeyframe;
        // chunkStart / chunkEnd at their defaults (0) only ever match frame 0, which this test already catches
        if ((!P.bIntraRefresh || frm.frameNum == 0) && sinceKey >= P.keyframeMax &&
            (!m_extendGopBoundary || sinceKey >= P.keyframeMax + P.gopLookahead))
        {
            const int keyType = P.bOpenGOP && m_lastKeyframe >= 0 ? TYPE_I : TYPE_IDR;
            if (frm.sliceType == TYPE_AUTO || frm.sliceType == TYPE_I)
                frm.sliceType = keyType;
            const bool compatible = frm.sliceType == TYPE_IDR || (P.bOpenGOP && frm.sliceType == TYPE_I);
            if (!compatible)
                frm.sliceType = keyType;
        }
        if (frm.bIsFadeEnd)                        // :2153-2155
            frm.sliceType = P.bOpenGOP && m_lastKeyframe >= 0 ? TYPE_I : TYPE_IDR;
        if (frm.sliceType == TYPE_I && sinceKey >= P.keyframeMin)
        {
            if (P.bOpenGOP)
            {
                m_lastKeyframe = frm.frameNum;
                frm.bKeyframe = true;
            }
            else
                frm.sliceType = TYPE_IDR;
        }
        if (frm.sliceType == TYPE_IDR && frm.bScenecut && isClosedGopRadl)
        {
            for (int i = bframes; i < bframes + P.radl; i++)
                list[i]->sliceType = TYPE_B;
            list[bframes + P.radl]->sliceType = TYPE_IDR;
        }
        if (frm.sliceType == TYPE_IDR)
        {
            m_lastKeyframe = frm.frameNum;
            frm.bKeyframe = true;
            if (bframes > 0 && !P.radl)
            {
                // a closed GOP cannot end on B frames
                list[bframes - 1]->sliceType = TYPE_P;
                bframes--;
            }
        }
        // (after the step back above frm is still the IDR frame, which ends the loop below)
        if (bframes == P.bframes || !list[bframes + 1])
            if (frm.sliceType == TYPE_AUTO || isTypeB(frm.sliceType))
                frm.sliceType = TYPE_P;
        if (frm.sliceType == TYPE_BREF)
            brefs++;
        if (frm.sliceType == TYPE_AUTO)
            frm.sliceType = TYPE_B;
        else if (!isTypeB(frm.sliceType))
            break;
    }

    if (P.bEnableTemporalSubLayers > 2)
        finishTemporalMiniGop(frames, list, bframes, brefs, maxSearch);
    else
        finishMiniGop(frames, list, bframes, brefs, maxSearch);

    // after a keyframe, run the analysis again from it so that cuTree / VBV planning start there (:2718-2747)
    if ((P.cuTree || (P.vbvBufferSize && P.lookaheadDepth)) && isTypeI(m_lastNonB->sliceType))
    {
        frames[0] = m_lastNonB;
        int j = 0;
        for (; j < maxSearch; j++)
            frames[j + 1] = m_inputQueue[j];
        frames[j + 1] = nullptr;
        slicetypeAnalyse(frames, true);
    }
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

// A fresh CostEstimateGroup over the window's frames, as x265 sets one up per call site (slicetype.cpp:2920, 3030,
// 3691): m_batchMode starts out false again
void xgh_new_group(xgh_window* w, int32_t bEnableWeightedPred, int32_t bitDepth)
{
    delete w->est;
    w->est = new CostEstimateGroup(w->ctx, w->frames.data(), bEnableWeightedPred != 0, bitDepth);
}

void xgh_window_destroy(xgh_window* w)
{
    if (!w)
        return;
    x265gpu_aq_frames_finish(w->ctx);     // library threads may still be writing the frames' AQ arrays
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
                         const x265gpu_aq_params* aq, int32_t* const* invQscale, int32_t memKind)
{
    std::vector<Lowres*> fr(n);
    std::vector<PreLookaheadGroup::Picture> pics(n);
    for (int i = 0; i < n; i++)
    {
        fr[i] = w->frames[frames[i]];
        fr[i]->invQscaleFactor = invQscale ? invQscale[i] : nullptr;
        pics[i] = { y[i], u[i], v[i], strideY[i], strideC[i], memKind };
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

void xgh_add_many(xgh_window* w, int32_t n, const int32_t* p0p1b)
{
    for (int i = 0; i < n; i++)
        w->est->add(p0p1b[3 * i], p0p1b[3 * i + 1], p0p1b[3 * i + 2]);
}

// what the AQ form of the pre-lookahead left running or on the device
static int window_resolve(const xgh_window* w)
{
    int rc = resolveIntraTotals(w->ctx, w->frames.data(), (int)w->frames.size());
    if (rc == X265GPU_OK)
        rc = x265gpu_aq_frames_finish(w->ctx);
    return rc;
}

int xgh_resolve(xgh_window* w) { return window_resolve(w); }

int xgh_finish_batch(xgh_window* w)
{
    w->est->finishBatch();
    const int rc = window_resolve(w);
    return w->est->status() != X265GPU_OK ? w->est->status() : rc;
}

int64_t xgh_cost_est(const xgh_window* w, int32_t b, int32_t d0, int32_t d1)
{
    window_resolve(w);
    return w->frames[b]->costEst[d0][d1];
}

void xgh_collect_costs(const xgh_window* w, int32_t n, const int32_t* p0p1b, int64_t* out)
{
    window_resolve(w);
    for (int i = 0; i < n; i++)
    {
        const int p0 = p0p1b[3 * i], p1 = p0p1b[3 * i + 1], b = p0p1b[3 * i + 2];
        const Lowres& f = *w->frames[b];
        out[3 * i] = f.costEst[b - p0][p1 - b];
        out[3 * i + 1] = f.costEstAq[b - p0][p1 - b];
        out[3 * i + 2] = f.intraMbs[b - p0];
    }
}
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
    const int rc = window_resolve(w);
    if (rc != X265GPU_OK)
        return rc;
    return CuTree(w->ctx, p).cuTree(w->frames.data(), numframes, bIntra != 0);
}

int xgh_status(const xgh_window* w) { return w->est->status(); }

// BitCost::setQP's table for one lambda (source/encoder/bitcost.cpp:31-56, 87-102): bits(|mvd|) in single
// precision as CalculateLogs forms them, scaled by lambda, rounded and capped at 2^15 - 1; symmetric around
// entry 2 * 32768.  Inside x265 the table is BitCost's own (m_cost); this builder serves callers without one.
void xgh_bitcost_table(double lambda, uint16_t* out)
{
    const int maxMv = 1 << 15;                       // BC_MAX_MV (bitcost.h:92)
    uint16_t* centre = out + 2 * maxMv;
    const float twoOverLn2 = 2.0f / logf(2.0f);
    for (int mvd = 0; mvd <= 2 * maxMv; mvd++)
    {
        const float bits = mvd ? logf((float)(mvd + 1)) * twoOverLn2 + 1.718f : 0.718f;
        const double scaled = bits * lambda + 0.5f;
        const uint16_t c = (uint16_t)(scaled < 32767.0 ? scaled : 32767.0);
        centre[mvd] = centre[-mvd] = c;
    }
}

struct LaFrame
{
    Lowres lowres;
    std::vector<double> qpAq, qpCuTree;
    std::vector<int32_t> invQ;
};

struct xgh_lookahead
{
    x265gpu_ctx* ctx;
    Lookahead* la;
    std::deque<LaFrame> frames;           // stable addresses: frames are only appended
    int cuCount;
    bool qg8;
};

xgh_lookahead* xgh_lookahead_create(x265gpu_ctx* ctx, const xgh_lookahead_param* p)
{
    if (!ctx || !p)
        return nullptr;
    LookaheadParam q;
    q.bframes = p->bframes;
    q.lookaheadDepth = p->lookaheadDepth;
    q.bFrameAdaptive = p->bFrameAdaptive;
    q.bBPyramid = p->bBPyramid != 0;
    q.scenecutThreshold = p->scenecutThreshold;
    q.scenecutBias = p->scenecutBias;
    q.keyframeMin = p->keyframeMin;
    q.keyframeMax = p->keyframeMax;
    q.bOpenGOP = p->bOpenGOP != 0;
    q.bIntraRefresh = p->bIntraRefresh != 0;
    q.radl = p->radl;
    q.gopLookahead = p->gopLookahead;
    q.cuTree = p->cuTree != 0;
    q.bFrameBias = p->bFrameBias;
    q.vbvBufferSize = p->vbvBufferSize;
    q.vbvMaxBitrate = p->vbvMaxBitrate;
    q.bCqp = p->bCqp != 0;
    q.bEnableTemporalSubLayers = p->bEnableTemporalSubLayers;
    q.maxNumReferences = p->maxNumReferences;
    q.bEnableWeightedPred = p->bEnableWeightedPred != 0;
    q.bEnableWeightedBiPred = p->bEnableWeightedBiPred != 0;
    q.fpsNum = p->fpsNum;
    q.fpsDenom = p->fpsDenom;
    q.qCompress = p->qCompress;
    q.bitDepth = p->bitDepth;
    q.poolWorkers = p->poolWorkers;
    q.bAdaptiveQuant = p->bAdaptiveQuant != 0;
    q.aq.aqMode = p->aqMode;
    q.aq.aqStrength = p->aqStrength;
    q.aq.bWeightedPred = q.bEnableWeightedPred || q.bEnableWeightedBiPred;
    q.bPresearch = p->bPresearch != 0;
    q.lookaheadSlices = p->lookaheadSlices;
    q.bAQMotion = p->bAQMotion != 0;
    q.bHistBasedSceneCut = p->bHistBasedSceneCut != 0;
    q.bEnableFades = p->bEnableFades != 0;
    q.aq.bBlockVariance = q.bEnableFades;     // param->bDynamicRefine || param->bEnableFades (slicetype.cpp:688)
    q.aq.qgSize = p->qgSize ? p->qgSize : 32;
    q.aq.bHevcAq = p->bHevcAq;
    q.aq.qpAdaptationRange = p->qpAdaptationRange;
    q.maxCUSize = p->maxCUSize ? p->maxCUSize : 64;
    if (q.aq.bHevcAq)
        q.bAdaptiveQuant = true;              // Lookahead::m_bAdaptiveQuant (slicetype.cpp:1080-1084)
    if (q.bAQMotion)
        q.bAdaptiveQuant = true;              // Lookahead::m_bAdaptiveQuant (slicetype.cpp:1080-1084)
    xgh_lookahead* h = new xgh_lookahead;
    h->ctx = ctx;
    h->la = new Lookahead(ctx, q);
    if (h->la->unsupported() != X265GPU_OK)
    {
        xgh_lookahead_destroy(h);
        return nullptr;
    }
    x265gpu_geometry g;
    x265gpu_get_geometry(ctx, &g);
    h->cuCount = g.cuCount;
    h->qg8 = q.bAdaptiveQuant && q.aq.qgSize == 8;
    return h;
}

void xgh_lookahead_destroy(xgh_lookahead* h)
{
    if (!h)
        return;
    x265gpu_aq_frames_finish(h->ctx);     // library threads may still be writing the frames' AQ arrays
    delete h->la;
    delete h;
}

int xgh_lookahead_add_picture(xgh_lookahead* h, const void* y, const void* u, const void* v, intptr_t strideY,
                              intptr_t strideC, int32_t memKind, int32_t sliceType, int32_t sliceTypeReq)
{
    if (!h || !y)
        return X265GPU_ERR_ARG;
    const int idx = (int)h->frames.size();
    h->frames.emplace_back();
    LaFrame& f = h->frames.back();
    // Lowres::create: cuCountFullRes entries with qgSize 8, zeroed (lowres.cpp:92-116)
    const size_t offsets = (size_t)h->cuCount * (h->qg8 ? 4 : 1);
    f.qpAq.assign(offsets, 0.0);
    f.qpCuTree.assign(offsets, 0.0);
    f.invQ.assign(offsets, 0);
    f.lowres.slot = idx;
    f.lowres.frameNum = idx;
    f.lowres.qpAqOffset = f.qpAq.data();
    f.lowres.qpCuTreeOffset = f.qpCuTree.data();
    f.lowres.invQscaleFactor = f.invQ.data();
    f.lowres.sliceTypeReq = sliceTypeReq;
    f.lowres.pic.y = y;
    f.lowres.pic.u = u;
    f.lowres.pic.v = v;
    f.lowres.pic.strideY = strideY;
    f.lowres.pic.strideC = strideC;
    f.lowres.pic.memKind = memKind;
    h->la->addPicture(f.lowres, sliceType);
    return idx;
}

void xgh_lookahead_flush(xgh_lookahead* h) { h->la->flush(); }

int xgh_lookahead_get_decided(xgh_lookahead* h, xgh_decided* out)
{
    Lowres* f = h->la->getDecidedPicture();
    if (h->la->status() != X265GPU_OK)
        return h->la->status();
    if (!f)
        return 0;
    out->frame = f->frameNum;
    out->sliceType = f->sliceType;
    out->bScenecut = f->bScenecut;
    out->bKeyframe = f->bKeyframe;
    out->bLastMiniGopBFrame = f->bLastMiniGopBFrame;
    out->leadingBframes = f->leadingBframes;
    out->bIsFadeEnd = f->bIsFadeEnd;
    out->gopOffset = f->gopOffset;
    out->gopId = f->gopId;
    out->tempLayer = f->tempLayer;
    return 1;
}

int xgh_lookahead_trace(const xgh_lookahead* h, const int32_t** out)
{
    static_assert(sizeof(RequestTrace) == 6 * sizeof(int32_t), "RequestTrace is read as 6 ints");
    const std::vector<RequestTrace>& t = h->la->trace();
    *out = t.empty() ? nullptr : &t[0].p0;
    return (int)t.size();
}

// the getters may be called while frames added after the last decision still have their AQ mapping running
static void lookahead_resolve(const xgh_lookahead* h)
{
    std::vector<Lowres*> fr;
    for (const LaFrame& f : h->frames)
        if (f.lowres.intraTotalsPending)
            fr.push_back(const_cast<Lowres*>(&f.lowres));
    if (!fr.empty())
        resolveIntraTotals(h->ctx, fr.data(), (int)fr.size());
    x265gpu_aq_frames_finish(h->ctx);
}

int64_t xgh_lookahead_cost_est(const xgh_lookahead* h, int32_t frame, int32_t d0, int32_t d1)
{
    lookahead_resolve(h);
    return h->frames[frame].lowres.costEst[d0][d1];
}
int64_t xgh_lookahead_cost_est_aq(const xgh_lookahead* h, int32_t frame, int32_t d0, int32_t d1)
{
    lookahead_resolve(h);
    return h->frames[frame].lowres.costEstAq[d0][d1];
}
int32_t xgh_lookahead_intra_mbs(const xgh_lookahead* h, int32_t frame, int32_t d0) { return h->frames[frame].lowres.intraMbs[d0]; }
int32_t xgh_lookahead_slot(const xgh_lookahead* h, int32_t frame) { return h->frames[frame].lowres.slot; }
const double* xgh_lookahead_qp_offsets(const xgh_lookahead* h, int32_t frame, int32_t which)
{
    lookahead_resolve(h);
    return which ? h->frames[frame].qpCuTree.data() : h->frames[frame].qpAq.data();
}
const int32_t* xgh_lookahead_inv_qscale(const xgh_lookahead* h, int32_t frame)
{
    lookahead_resolve(h);
    // what the estimates read: invQscaleFactor8x8 with qgSize 8
    return h->qg8 ? h->frames[frame].lowres.invQscaleFactor8x8.data() : h->frames[frame].invQ.data();
}
int xgh_lookahead_aq_layer(const xgh_lookahead* h, int32_t frame, int32_t layer, int32_t which, const double** out, int32_t* n)
{
    lookahead_resolve(h);
    const std::vector<Lowres::AqLayer>& l = h->frames[frame].lowres.aqLayers;
    if (layer >= 0 && layer < (int)l.size())
    {
        const std::vector<double>& a = which ? l[layer].dCuTreeOffset : l[layer].dQpOffset;
        *out = a.data();
        *n = (int32_t)a.size();
    }
    return (int)l.size();
}
const int64_t* xgh_lookahead_planned_satd(const xgh_lookahead* h, int32_t frame, const int32_t** types, int32_t* indB)
{
    const Lowres& f = h->frames[frame].lowres;
    *types = f.plannedType.empty() ? nullptr : f.plannedType.data();
    *indB = f.indB;
    return f.plannedSatd.empty() ? nullptr : f.plannedSatd.data();
}
const int32_t* xgh_lookahead_recalc_row_satds(const xgh_lookahead* h, int32_t frame)
{
    const Lowres& f = h->frames[frame].lowres;
    return f.recalcRowSatds.empty() ? nullptr : f.recalcRowSatds.data();
}
int xgh_lookahead_decide_calls(const xgh_lookahead* h) { return h->la->decideCalls(); }
int xgh_lookahead_presearch_hits(const xgh_lookahead* h) { return h->la->presearchHits(); }
int xgh_lookahead_precost_hits(const xgh_lookahead* h) { return h->la->precostHits(); }
void xgh_lookahead_seconds(const xgh_lookahead* h, double out[6])
{
    for (int i = 0; i < Lookahead::T_COUNT; i++)
        out[i] = h->la->seconds()[i];
}

} // extern "C"
