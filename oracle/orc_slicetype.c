/* CPU oracle (test infrastructure only; parity unpinned, see orc.h).
 * The serial slice-type decision of the lookahead, restated from source/encoder/slicetype.cpp:
 *   Lookahead::addPicture / flush / getDecidedPicture / findJob   :1299-1431
 *   Lookahead::slicetypeDecide                                    :1982-2751
 *   Lookahead::placeBref / compCostBref                           :1864-1908
 *   Lookahead::vbvLookahead / vbvFrameCost                        :2753-2844
 *   Lookahead::slicetypeAnalyse                                   :2846-3184
 *   Lookahead::scenecut / scenecutInternal                        :3186-3321
 *   Lookahead::slicetypePath / slicetypePathCost                  :3484-3579
 *   Lookahead::frameCostRecalculate                               :4068-4145
 *   Lookahead::aqMotion / calcMotionAdaptiveQuantFrame            :3581-3663
 *   Lookahead::histBasedScenecut / detectHistBasedSceneChange     :3323-3482
 *   the fade detection of slicetypeDecide (--fades)               :2042-2088, 2153-2155
 * Not restated (the product's host layer refuses the same settings): zones, analysis load,
 * --mcstf, more than two temporal sub-layers.  qgSize 8 and rc.hevcAq act through the frames (orc_lowres::qg8,
 * aqLayer[]) in cuTreeFinish and frameCostRecalculate.
 * Frames enter already pre-analysed (Lowres::init, AQ, intra estimate are per-frame functions of the
 * picture alone, so running them at addPicture instead of in PreLookaheadGroup changes nothing). */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IS_TYPE_I(x) ((x) == ORC_TYPE_I || (x) == ORC_TYPE_IDR)
#define IS_TYPE_B(x) ((x) == ORC_TYPE_B || (x) == ORC_TYPE_BREF)
#define QCAP 4096

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

struct orc_la
{
    orc_param p;
    orc_la_param lp;
    orc_cutree_param cp;
    orc_tld* tld;
    orc_lowres* lastNonB;
    int lastKeyframe;
    int isSceneTransition, extendGopBoundary;
    int batchMotionSearch, batchFrameCosts;
    int filled, fullQueueSize, inputCount;
    int blocks8x8;                     /* m_8x8Blocks */
    orc_lowres* inq[QCAP];
    int inCount;
    orc_lowres* outq[QCAP];
    int outHead, outCount;
    int batchOpen;                     /* a CostEstimateGroup in batch mode is collecting requests */
    /* --hist-scenecut (:1143-1174): running averages of the histogram differences per segment and plane */
    uint32_t accHistDiffRunningAvg[3][4][4];
    int resetRunningAvg;
    uint32_t segmentCountThreshold;
    /* --fades (:2042-2088, slicetype.h:196-200) */
    double frameVariance[ORC_BFRAME_MAX + 4];
    int isFadeIn, fadeStart;
    int gopId;                     /* --temporal-layers 3..5 (:1176-1192) */
    uint64_t fadeCount;
    int32_t* trace;
    int ntrace, captrace;
};

/* ---- request plumbing: CostEstimateGroup::singleCost / add / finishBatch ---- */

static void trace_hook(void* user, orc_lowres** frames, int p0, int p1, int b, const int* bDoSearch, int batchMode)
{
    orc_la* la = (orc_la*)user;
    if (la->ntrace == la->captrace)
    {
        la->captrace = la->captrace ? 2 * la->captrace : 1024;
        la->trace = (int32_t*)realloc(la->trace, sizeof(int32_t) * 6 * la->captrace);
    }
    int32_t* t = la->trace + 6 * la->ntrace++;
    t[0] = frames[p0]->frameNum;
    t[1] = frames[p1]->frameNum;
    t[2] = frames[b]->frameNum;
    t[3] = bDoSearch[0];
    t[4] = bDoSearch[1];
    t[5] = batchMode;
}

static int64_t single_cost(orc_la* la, orc_lowres** frames, int p0, int p1, int b, int bIntraPenalty)
{
    orc_set_request_hook(trace_hook, la);
    int64_t c = orc_estimate_frame_cost_single(la->tld, &la->p, frames, p0, p1, b, bIntraPenalty);
    orc_set_request_hook(NULL, NULL);
    return c;
}

/* a batch's requests are independent of each other (distinct searches, or none), so the worker threads'
 * order cannot matter and they run here as they are added */
static void batch_add(orc_la* la, orc_lowres** frames, int p0, int p1, int b)
{
    orc_set_request_hook(trace_hook, la);
    orc_estimate_frame_cost(la->tld, &la->p, frames, p0, p1, b, 0);
    orc_set_request_hook(NULL, NULL);
}

static void cu_tree(orc_la* la, orc_lowres** frames, int numframes, int bIntra)
{
    orc_set_request_hook(trace_hook, la);
    orc_cutree(la->tld, &la->p, &la->cp, frames, numframes, bIntra);
    orc_set_request_hook(NULL, NULL);
}

/* ---- :3609-3663.  The block vectors' lengths, raised to 0.1, normalised over the frame; blocks more than one
 * deviation above the mean get the excess added to their offsets and the matching factor added to invQscaleFactor. */
static void calc_motion_adaptive_quant_frame(orc_la* la, orc_lowres** frames, int p0, int p1, int b)
{
    orc_lowres* f = frames[b];
    const int listDist[2] = { b - p0, p1 - b };
    const int W = f->widthInCU, H = f->heightInCU;
    (void)la;
    if (!f->qpAqMotionOffset)
        f->qpAqMotionOffset = (double*)calloc(f->cuCount, sizeof(double));
    double qp_adj = 0, avg_adj = 0, avg_adj_pow2 = 0, sd;
    for (int blocky = 0; blocky < H; blocky++)
    {
        int cuIndex = blocky * W;
        for (int blockx = 0; blockx < W; blockx++, cuIndex++)
        {
            int lists_used = f->lowresCosts[b - p0][p1 - b][cuIndex] >> ORC_LOWRES_COST_SHIFT;
            double displacement = 0;
            for (int list = 0; list < 2; list++)
                if ((lists_used >> list) & 1)
                {
                    const orc_mv* mvs = f->lowresMvs[list][listDist[list]];
                    int32_t x = mvs[cuIndex].x, y = mvs[cuIndex].y;
                    displacement += sqrt((double)(abs(x) * abs(x)) + (double)(abs(y) * abs(y)));
                }
            if (lists_used == 3)
                displacement = displacement / 2;
            qp_adj = pow(displacement, 0.1);
            f->qpAqMotionOffset[cuIndex] = qp_adj;
            avg_adj += qp_adj;
            avg_adj_pow2 += qp_adj * qp_adj;
        }
    }
    avg_adj /= f->cuCount;
    avg_adj_pow2 /= f->cuCount;
    sd = sqrt((avg_adj_pow2 - (avg_adj * avg_adj)));
    if (sd > 0)
        for (int cuIndex = 0; cuIndex < W * H; cuIndex++)
        {
            qp_adj = f->qpAqMotionOffset[cuIndex];
            qp_adj = (qp_adj - avg_adj) / sd;
            if (qp_adj > 1)
            {
                f->qpAqOffset[cuIndex] += qp_adj;
                f->qpCuTreeOffset[cuIndex] += qp_adj;
                f->invQscaleFactor[cuIndex] += orc_exp2fix8(qp_adj);
            }
        }
}

/* ---- :3581-3607 ---- */
static void aq_motion(orc_la* la, orc_lowres** frames, int bIntra)
{
    if (bIntra)
        return;
    int curnonb = 0, lastnonb = 1;
    while (frames[lastnonb]->sliceType != ORC_TYPE_P)
        lastnonb++;
    int bframes = lastnonb - 1;
    if (la->lp.bBPyramid && bframes > 1)
    {
        int middle = (bframes + 1) / 2;
        for (int i = 1; i < lastnonb; i++)
        {
            int p0 = i > middle ? middle : curnonb;
            int p1 = i < middle ? middle : lastnonb;
            if (i != middle)
                calc_motion_adaptive_quant_frame(la, frames, p0, p1, i);
        }
        calc_motion_adaptive_quant_frame(la, frames, curnonb, lastnonb, middle);
    }
    else
        for (int i = 1; i < lastnonb; i++)
            calc_motion_adaptive_quant_frame(la, frames, curnonb, lastnonb, i);
    calc_motion_adaptive_quant_frame(la, frames, curnonb, lastnonb, lastnonb);
}

void orc_lowres_set_frame_variance(orc_lowres* lr, double v) { lr->frameVariance = v; }
int orc_lowres_is_fade_end(const orc_lowres* lr) { return lr->bIsFadeEnd; }

/* ---- :2042-2088: a run of pictures whose variance does not fall is a fade-in; the picture that ends a run of at
 * least a second is marked, and the fix-ups below make it a keyframe.  Kept to the letter, including the ring
 * index over POC modulo the array length. ---- */
static void detect_fades(orc_la* la, orc_lowres** list)
{
    int j, endIndex = 0, length = ORC_BFRAME_MAX + 4;
    for (j = 0; j < length; j++)
        la->frameVariance[j] = -1;
    for (j = 0; list[j] != NULL; j++)
        la->frameVariance[list[j]->frameNum % length] = list[j]->frameVariance;
    for (int k = list[0]->frameNum % length; k <= list[j - 1]->frameNum % length; k++)
    {
        if (la->frameVariance[k] == -1)
            break;
        if ((k > 0 && la->frameVariance[k] >= la->frameVariance[k - 1]) ||
            (k == 0 && la->frameVariance[k] >= la->frameVariance[length - 1]))
        {
            la->isFadeIn = 1;
            if (la->fadeCount == 0 && la->fadeStart == -1)
            {
                for (int temp = list[0]->frameNum; temp <= list[j - 1]->frameNum; temp++)
                    if (k == temp % length)
                    {
                        la->fadeStart = temp ? temp - 1 : 0;
                        break;
                    }
            }
            la->fadeCount = list[endIndex]->frameNum > la->fadeStart ? list[endIndex]->frameNum - la->fadeStart : 0;
            endIndex++;
        }
        else
        {
            if (la->isFadeIn && la->fadeCount >= (uint64_t)(la->cp.fpsNum / la->cp.fpsDenom))
            {
                for (int temp = 0; list[temp] != NULL; temp++)
                    if (list[temp]->frameNum == la->fadeStart + (int)la->fadeCount)
                    {
                        list[temp]->bIsFadeEnd = 1;
                        break;
                    }
            }
            la->isFadeIn = 0;
            la->fadeCount = 0;
            la->fadeStart = -1;
        }
        if (k == length - 1)
            k = -1;
    }
}

void orc_lowres_set_hist_stats(orc_lowres* lr, const orc_hist_stats* st)
{
    if (!lr->hist)
        lr->hist = (orc_hist_stats*)malloc(sizeof(orc_hist_stats));
    *lr->hist = *st;
    lr->bHistScenecutAnalyzed = 0;      /* collectPictureStatistics ends with this (:1833) */
}

/* ---- :3323-3454.  The segment size the thresholds are scaled by grows as the loops go: the remainder of the last
 * column / row is added to the running width / height and never taken off again (kept as the reference has it). */
static int detect_hist_based_scene_change(orc_la* la, orc_lowres** frames, int p0, int p1, int p2)
{
    const orc_hist_stats* prev = frames[p0]->hist;
    const orc_hist_stats* cur = frames[p1]->hist;
    const orc_hist_stats* fut = frames[p2]->hist;
    frames[p1]->bHistScenecutAnalyzed = 1;

    uint32_t abruptChangeCount = 0, sceneChangeCount = 0;
    const uint32_t widthFullRes = (uint32_t)la->p.sourceWidth, heightFullRes = (uint32_t)la->p.sourceHeight;
    uint32_t segmentWidth = widthFullRes / 4, segmentHeight = heightFullRes / 4;
    const int64_t var[3][2] = { { cur->picAvgVariance, prev->picAvgVariance }, { cur->picAvgVarianceCb, prev->picAvgVarianceCb },
                                { cur->picAvgVarianceCr, prev->picAvgVarianceCr } };

    for (uint32_t sw = 0; sw < 4; sw++)
        for (uint32_t sh = 0; sh < 4; sh++)
        {
            int isAbruptChange = 0, isSceneChange = 0;
            segmentWidth += (sw == 3) ? widthFullRes - 4 * segmentWidth : 0;
            segmentHeight += (sh == 3) ? heightFullRes - 4 * segmentHeight : 0;
            const uint32_t num64 = (segmentWidth * segmentHeight) >> 12;       /* NUM64x64INPIC */
            uint32_t thresh[3];
            thresh[0] = (llabs(var[0][0] - var[0][1]) > 390 && (var[0][0] > 1500 || var[0][1] > 1500)) ? 3500 * num64 : 2250 * num64;
            for (int c = 1; c < 3; c++)
                thresh[c] = (llabs(var[c][0] - var[c][1]) > 10 && (var[c][0] > 20 || var[c][1] > 20)) ? 3500 / 4 * num64
                                                                                                     : 2250 / 4 * num64;
            uint32_t accHistDiff[3] = { 0, 0, 0 };
            for (int bin = 0; bin < 256; bin++)
                for (int c = 0; c < 3; c++)
                    accHistDiff[c] += (uint32_t)abs((int32_t)cur->picHistogram[sw][sh][c][bin] - (int32_t)prev->picHistogram[sw][sh][c][bin]);
            if (la->resetRunningAvg)
                for (int c = 0; c < 3; c++)
                    la->accHistDiffRunningAvg[c][sw][sh] = accHistDiff[c];
            for (int c = 0; c < 3; c++)
            {
                const uint32_t err = (uint32_t)abs((int32_t)la->accHistDiffRunningAvg[c][sw][sh] - (int32_t)accHistDiff[c]);
                if (err > thresh[c] && accHistDiff[c] >= err)
                    isAbruptChange = 1;
            }
            if (isAbruptChange)
            {
                const int16_t f = (int16_t)fut->averageIntensityPerSegment[sw][sh][0];
                const int16_t c = (int16_t)cur->averageIntensityPerSegment[sw][sh][0];
                const int16_t q = (int16_t)prev->averageIntensityPerSegment[sw][sh][0];
                const uint8_t futurePast = (uint8_t)abs(f - q), futurePresent = (uint8_t)abs(f - c), presentPast = (uint8_t)abs(c - q);
                if (futurePresent >= 1.5 * futurePast && presentPast >= 1.5 * futurePast)
                    ;                                   /* a flash */
                else if (futurePresent < 4 && presentPast < 4)
                    ;                                   /* a fade */
                else if (abs(futurePresent - presentPast) < 4 && futurePresent + presentPast >= futurePast)
                    ;                                   /* an intensity change */
                else
                    isSceneChange = 1;
            }
            else
                la->accHistDiffRunningAvg[0][sw][sh] = (3 * la->accHistDiffRunningAvg[0][sw][sh] + accHistDiff[0]) / 4;
            abruptChangeCount += isAbruptChange;
            sceneChangeCount += isSceneChange;
        }
    la->resetRunningAvg = abruptChangeCount >= la->segmentCountThreshold;
    return sceneChangeCount >= la->segmentCountThreshold;
}

/* ---- :3456-3482 ---- */
static int hist_based_scenecut(orc_la* la, orc_lowres** frames, int p0, int p1, int numFrames)
{
    if (la->p.bframes)
    {
        int maxp1 = imin(p0 + 1 + la->p.bframes, numFrames);
        for (int cp1 = p0; cp1 < maxp1; cp1++)
        {
            if (frames[cp1 + 1]->bHistScenecutAnalyzed)
                continue;
            if (frames[cp1 + 2] != NULL && detect_hist_based_scene_change(la, frames, cp1, cp1 + 1, cp1 + 2))
                frames[cp1 + 1]->bScenecut = 1;
        }
    }
    return frames[p1]->bScenecut;
}

/* ---- :4068-4145 (no hevcAq) ---- */
int64_t orc_frame_cost_recalculate(orc_lowres** frames, int p0, int p1, int b)
{
    orc_lowres* f = frames[b];
    if (f->sliceType == ORC_TYPE_B)
        return f->costEstAq[b - p0][p1 - b];
    int64_t score = 0;
    const int W = f->widthInCU, H = f->heightInCU;
    if (!f->recalcRowSatds)
        f->recalcRowSatds = (int32_t*)malloc(sizeof(int32_t) * H);
    /* rc.hevcAq: the offsets of the layer of minAQDepth stand in for qpCuTreeOffset (:4078-4112) */
    const double* qp_offset = f->numAqLayers ? f->aqLayer[f->numAqLayers - 1].dCuTreeOffset : f->qpCuTreeOffset;
    for (int cuy = H - 1; cuy >= 0; cuy--)
    {
        f->recalcRowSatds[cuy] = 0;
        for (int cux = W - 1; cux >= 0; cux--)
        {
            int cuxy = cux + cuy * W;
            int cuCost = f->lowresCosts[b - p0][p1 - b][cuxy] & ORC_LOWRES_COST_MASK;
            double qp_adj;
            if (f->qg8)
                qp_adj = (qp_offset[cux * 2 + cuy * W * 4] + qp_offset[cux * 2 + cuy * W * 4 + 1] +
                          qp_offset[cux * 2 + cuy * W * 4 + 2 * W] + qp_offset[cux * 2 + cuy * W * 4 + 2 * W + 1]) / 4;
            else
                qp_adj = qp_offset[cuxy];
            cuCost = (cuCost * orc_exp2fix8(qp_adj) + 128) >> 8;
            f->recalcRowSatds[cuy] += cuCost;
            if ((cuy > 0 && cuy < H - 1 && cux > 0 && cux < W - 1) || W <= 2 || H <= 2)
                score += cuCost;
        }
    }
    return score;
}

/* ---- :2830-2844 ---- */
static int64_t vbv_frame_cost(orc_la* la, orc_lowres** frames, int p0, int p1, int b)
{
    int64_t cost = single_cost(la, frames, p0, p1, b, 0);
    if (la->lp.aqMode || la->lp.bAQMotion)
    {
        if (la->lp.cuTree)
            return orc_frame_cost_recalculate(frames, p0, p1, b);
        return frames[b]->costEstAq[b - p0][p1 - b];
    }
    return cost;
}

/* ---- :2753-2828 ---- */
static void vbv_lookahead(orc_la* la, orc_lowres** frames, int numFrames, int keyframe)
{
    int prevNonB = 0, curNonB = 1, idx = 0;
    while (curNonB < numFrames && IS_TYPE_B(frames[curNonB]->sliceType))
        curNonB++;
    int nextNonB = keyframe ? prevNonB : curNonB;
    int nextB = prevNonB + 1;
    int nextBRef = 0, curBRef = 0;
    if (la->lp.bBPyramid && curNonB - prevNonB > 1)
        curBRef = (prevNonB + curNonB + 1) / 2;
    int miniGopEnd = keyframe ? prevNonB : curNonB;
    while (curNonB <= numFrames)
    {
        if (nextNonB != curNonB)
        {
            int p0 = IS_TYPE_I(frames[curNonB]->sliceType) ? curNonB : prevNonB;
            frames[nextNonB]->plannedSatd[idx] = vbv_frame_cost(la, frames, p0, curNonB, curNonB);
            frames[nextNonB]->plannedType[idx] = frames[curNonB]->sliceType;
            if (curNonB > miniGopEnd)
            {
                for (int j = nextB; j < miniGopEnd; j++)
                {
                    frames[j]->plannedSatd[frames[j]->indB] = frames[nextNonB]->plannedSatd[idx];
                    frames[j]->plannedType[frames[j]->indB++] = frames[nextNonB]->plannedType[idx];
                }
            }
            idx++;
        }
        if (la->lp.bBPyramid && curNonB - prevNonB > 1)
            nextBRef = (prevNonB + curNonB + 1) / 2;
        for (int i = prevNonB + 1; i < curNonB; i++, idx++)
        {
            int64_t satdCost;
            int type = ORC_TYPE_B;
            if (nextBRef)
            {
                if (i == nextBRef)
                {
                    satdCost = vbv_frame_cost(la, frames, prevNonB, curNonB, nextBRef);
                    type = ORC_TYPE_BREF;
                }
                else if (i < nextBRef)
                    satdCost = vbv_frame_cost(la, frames, prevNonB, nextBRef, i);
                else
                    satdCost = vbv_frame_cost(la, frames, nextBRef, curNonB, i);
            }
            else
                satdCost = vbv_frame_cost(la, frames, prevNonB, curNonB, i);
            frames[nextNonB]->plannedSatd[idx] = satdCost;
            frames[nextNonB]->plannedType[idx] = type;
            for (int j = nextB; j < miniGopEnd; j++)
            {
                if (curBRef && curBRef == i)
                    break;
                if (j >= i && j != nextBRef)
                    continue;
                frames[j]->plannedSatd[frames[j]->indB] = satdCost;
                frames[j]->plannedType[frames[j]->indB++] = type;
            }
        }
        prevNonB = curNonB;
        curNonB++;
        while (curNonB <= numFrames && IS_TYPE_B(frames[curNonB]->sliceType))
            curNonB++;
    }
    frames[nextNonB]->plannedType[idx] = ORC_TYPE_AUTO;
}

/* ---- :3281-3321 ---- */
static int scenecut_internal(orc_la* la, orc_lowres** frames, int p0, int p1, int bRealScenecut)
{
    orc_lowres* frame = frames[p1];
    single_cost(la, frames, p0, p1, p1, 0);
    int64_t icost = frame->costEst[0][0];
    int64_t pcost = frame->costEst[p1 - p0][0];
    int gopSize = (frame->frameNum - la->lastKeyframe) % la->lp.keyframeMax;
    float threshMax = (float)(la->lp.scenecutThreshold / 100.0);
    float threshMin = (float)(threshMax * 0.25);
    double bias = la->lp.scenecutBias;
    if (bRealScenecut)
    {
        if (la->lp.keyframeMin == la->lp.keyframeMax)
            threshMin = threshMax;
        if (gopSize <= la->lp.keyframeMin / 4 || la->lp.bIntraRefresh)
            bias = threshMin / 4;
        else if (gopSize <= la->lp.keyframeMin)
            bias = threshMin * gopSize / la->lp.keyframeMin;
        else
            bias = threshMin + (threshMax - threshMin) * (gopSize - la->lp.keyframeMin) /
                   (la->lp.keyframeMax - la->lp.keyframeMin);
    }
    int res = pcost >= (1.0 - bias) * icost;
    if (res && bRealScenecut)
        frame->bScenecut = res;
    return res;
}

/* ---- :3186-3279 ---- */
static int scenecut(orc_la* la, orc_lowres** frames, int p0, int p1, int bRealScenecut, int numFrames)
{
    if (bRealScenecut && la->p.bframes)
    {
        int origmaxp1 = p0 + 1 + la->p.bframes;
        int maxp1 = imin(origmaxp1, numFrames);
        int fluctuate = 0, noScenecuts = 0;
        int64_t avgSatdCost = 0;
        if (frames[p0]->costEst[p1 - p0][0] > -1)
            avgSatdCost = frames[p0]->costEst[p1 - p0][0];
        int cnt = 1;
        for (int cp1 = p1; cp1 <= maxp1; cp1++)
        {
            if (!scenecut_internal(la, frames, p0, cp1, 0))
            {
                for (int i = cp1; i > p0; i--)
                {
                    frames[i]->bScenecut = 0;
                    noScenecuts = 0;
                }
            }
            else if (scenecut_internal(la, frames, cp1 - 1, cp1, 0))
            {
                frames[cp1]->bScenecut = 1;
                noScenecuts = 1;
            }
            avgSatdCost += frames[cp1]->costEst[cp1 - p0][0];
            cnt++;
        }
        if (noScenecuts)
        {
            fluctuate = 0;
            avgSatdCost /= cnt;
            for (int i = p1; i <= maxp1; i++)
            {
                int64_t curCost = frames[i]->costEst[i - p0][0];
                int64_t prevCost = frames[i - 1]->costEst[i - 1 - p0][0];
                if (fabs((double)(curCost - avgSatdCost)) > 0.1 * avgSatdCost ||
                    fabs((double)(curCost - prevCost)) > 0.1 * prevCost)
                {
                    fluctuate = 1;
                    if (!la->isSceneTransition && frames[i]->bScenecut)
                    {
                        la->isSceneTransition = 1;
                        for (int j = i + 1; j <= maxp1; j++)
                            frames[j]->bScenecut = 0;
                        break;
                    }
                }
                frames[i]->bScenecut = 0;
            }
        }
        if (!fluctuate && !noScenecuts)
            la->isSceneTransition = 0;
    }
    if (!frames[p1]->bScenecut && la->p.bframes)
        return 0;
    return scenecut_internal(la, frames, p0, p1, bRealScenecut);
}

/* ---- :3534-3579 ---- */
static int64_t slicetype_path_cost(orc_la* la, orc_lowres** frames, char* path, int64_t threshold)
{
    int64_t cost = 0;
    int loc = 1, cur_p = 0;
    path--;
    while (path[loc])
    {
        int next_p = loc;
        while (path[next_p] != 'P')
            next_p++;
        cost += single_cost(la, frames, cur_p, next_p, next_p, 0);
        if (cost > threshold)
            break;
        if (la->lp.bBPyramid && next_p - cur_p > 2)
        {
            int middle = cur_p + (next_p - cur_p) / 2;
            cost += single_cost(la, frames, cur_p, next_p, middle, 0);
            for (int next_b = loc; next_b < middle && cost < threshold; next_b++)
                cost += single_cost(la, frames, cur_p, middle, next_b, 0);
            for (int next_b = middle + 1; next_b < next_p && cost < threshold; next_b++)
                cost += single_cost(la, frames, middle, next_p, next_b, 0);
        }
        else
        {
            for (int next_b = loc; next_b < next_p && cost < threshold; next_b++)
                cost += single_cost(la, frames, cur_p, next_p, next_b, 0);
        }
        loc = next_p + 1;
        cur_p = next_p;
    }
    return cost;
}

/* ---- :3484-3511 ---- */
static void slicetype_path(orc_la* la, orc_lowres** frames, int length, char (*best_paths)[ORC_LOOKAHEAD_MAX + 1])
{
    char paths[2][ORC_LOOKAHEAD_MAX + 1];
    int num_paths = imin(la->p.bframes + 1, length);
    int64_t best_cost = 1LL << 62;
    int idx = 0;
    for (int path = 0; path < num_paths; path++)
    {
        int len = length - (path + 1);
        memcpy(paths[idx], best_paths[len % (ORC_BFRAME_MAX + 1)], len);
        memset(paths[idx] + len, 'B', path);
        strcpy(paths[idx] + len + path, "P");
        int64_t cost = slicetype_path_cost(la, frames, paths[idx], best_cost);
        if (cost < best_cost)
        {
            best_cost = cost;
            idx ^= 1;
        }
    }
    memcpy(best_paths[length % (ORC_BFRAME_MAX + 1)], paths[idx ^ 1], length);
}

/* ---- :2846-3184 ---- */
static void slicetype_analyse(orc_la* la, orc_lowres** frames, int bKeyframe)
{
    const orc_la_param* lp = &la->lp;
    const int bframesMax = la->p.bframes;
    int numFrames, origNumFrames, keyintLimit, framecnt;
    int maxSearch = imin(lp->lookaheadDepth, ORC_LOOKAHEAD_MAX);
    int cuCount = la->blocks8x8;
    int resetStart;
    int bIsVbvLookahead = lp->vbvBufferSize && lp->lookaheadDepth;

    for (framecnt = 0; framecnt < maxSearch; framecnt++)
    {
        orc_lowres* fenc = frames[framecnt + 1];
        if (!fenc || fenc->sliceType != ORC_TYPE_AUTO)
            break;
    }
    if (!framecnt)
    {
        if (lp->cuTree)
            cu_tree(la, frames, 0, bKeyframe);
        return;
    }
    frames[framecnt + 1] = NULL;

    int keylimit = lp->keyframeMax;
    int keyFrameLimit = keylimit + la->lastKeyframe - frames[0]->frameNum - 1;
    if (lp->gopLookahead && keyFrameLimit <= bframesMax + 1)
        keyintLimit = keyFrameLimit + lp->gopLookahead;
    else
        keyintLimit = keyFrameLimit;

    origNumFrames = numFrames = lp->bIntraRefresh ? framecnt : imin(framecnt, keyintLimit);

    if (bIsVbvLookahead)
        numFrames = framecnt;
    else if (lp->bOpenGOP && numFrames < framecnt)
        numFrames++;
    else if (numFrames == 0)
    {
        frames[1]->sliceType = ORC_TYPE_I;
        return;
    }

    if (la->batchMotionSearch)
    {
        for (int b = 2; b < numFrames; b++)
        {
            for (int i = 1; i <= bframesMax + 1; i++)
            {
                int p0 = b - i;
                if (p0 < 0)
                    continue;
                if (frames[b]->lowresMvs[0][i][0].x != 0x7FFF)
                    continue;
                int p1 = b + i;
                if (p1 >= numFrames || frames[b]->lowresMvs[1][i][0].x != 0x7FFF)
                    p1 = b;
                batch_add(la, frames, p0, p1, b);
            }
        }
        la->batchMotionSearch &= lp->poolWorkers >= 4;

        if (la->batchFrameCosts)
        {
            for (int b = 2; b < numFrames; b++)
            {
                for (int i = 1; i <= bframesMax + 1; i++)
                {
                    if (b < i)
                        continue;
                    if (frames[b]->lowresMvs[0][i][0].x == 0x7FFF)
                        continue;
                    int p0 = b - i;
                    for (int j = 0; j <= bframesMax; j++)
                    {
                        int p1 = b + j;
                        if (p1 >= numFrames)
                            break;
                        if (j && frames[b]->lowresMvs[1][j][0].x == 0x7FFF)
                            continue;
                        if (frames[b]->costEst[i][j] >= 0)
                            continue;
                        batch_add(la, frames, p0, p1, b);
                    }
                }
            }
            la->batchFrameCosts &= lp->poolWorkers > 12;
        }
    }

    int numBFrames = 0;
    int numAnalyzed = numFrames;
    int isScenecut = lp->bHistBasedSceneCut ? hist_based_scenecut(la, frames, 0, 1, origNumFrames)
                                            : scenecut(la, frames, 0, 1, 1, origNumFrames);

    if (lp->scenecutThreshold && isScenecut)
    {
        frames[1]->sliceType = ORC_TYPE_I;
        return;
    }
    if (lp->gopLookahead && keyFrameLimit >= 0 && keyFrameLimit <= bframesMax + 1)
    {
        int sceneTransition = la->isSceneTransition;
        la->extendGopBoundary = 0;
        for (int i = bframesMax + 1; i < origNumFrames; i += bframesMax + 1)
        {
            scenecut(la, frames, i, i + 1, 1, origNumFrames);
            for (int j = i + 1; j <= imin(i + bframesMax + 1, origNumFrames); j++)
            {
                if (frames[j]->bScenecut && scenecut_internal(la, frames, j - 1, j, 1))
                {
                    la->extendGopBoundary = 1;
                    break;
                }
            }
            if (la->extendGopBoundary)
                break;
        }
        la->isSceneTransition = sceneTransition;
    }
    if (bframesMax)
    {
        if (lp->bFrameAdaptive == 2)
        {
            if (numFrames > 1)
            {
                char best_paths[ORC_BFRAME_MAX + 1][ORC_LOOKAHEAD_MAX + 1];
                memset(best_paths, 0, sizeof(best_paths));
                strcpy(best_paths[1], "P");
                int best_path_index = numFrames % (ORC_BFRAME_MAX + 1);
                for (int j = 2; j <= numFrames; j++)
                    slicetype_path(la, frames, j, best_paths);
                numBFrames = (int)strspn(best_paths[best_path_index], "B");
                for (int j = 1; j < numFrames; j++)
                    frames[j]->sliceType = best_paths[best_path_index][j - 1] == 'B' ? ORC_TYPE_B : ORC_TYPE_P;
            }
            frames[numFrames]->sliceType = ORC_TYPE_P;
        }
        else if (lp->bFrameAdaptive == 1)
        {
            int64_t cost1p0, cost2p0, cost1b1, cost2p1;
            for (int i = 0; i <= numFrames - 2;)
            {
                cost2p1 = single_cost(la, frames, i + 0, i + 2, i + 2, 1);
                if (frames[i + 2]->intraMbs[2] > cuCount / 2)
                {
                    frames[i + 1]->sliceType = ORC_TYPE_P;
                    frames[i + 2]->sliceType = ORC_TYPE_P;
                    i += 2;
                    continue;
                }
                cost1b1 = single_cost(la, frames, i + 0, i + 2, i + 1, 0);
                cost1p0 = single_cost(la, frames, i + 0, i + 1, i + 1, 0);
                cost2p0 = single_cost(la, frames, i + 1, i + 2, i + 2, 0);
                if (cost1p0 + cost2p0 < cost1b1 + cost2p1)
                {
                    frames[i + 1]->sliceType = ORC_TYPE_P;
                    i += 1;
                    continue;
                }
                frames[i + 1]->sliceType = ORC_TYPE_B;
                int j;
                for (j = i + 2; j <= imin(i + bframesMax, numFrames - 1); j++)
                {
                    int64_t pthresh = imax(300 - (50 - la->p.bFrameBias) * (j - i - 1), 300 / 10);
                    int64_t pcost = single_cost(la, frames, i + 0, j + 1, j + 1, 1);
                    if (pcost > pthresh * cuCount || frames[j + 1]->intraMbs[j - i + 1] > cuCount / 3)
                        break;
                    frames[j]->sliceType = ORC_TYPE_B;
                }
                frames[j]->sliceType = ORC_TYPE_P;
                i = j;
            }
            frames[numFrames]->sliceType = ORC_TYPE_P;
            numBFrames = 0;
            while (numBFrames < numFrames && frames[numBFrames + 1]->sliceType == ORC_TYPE_B)
                numBFrames++;
        }
        else
        {
            numBFrames = imin(numFrames - 1, bframesMax);
            for (int j = 1; j < numFrames; j++)
                frames[j]->sliceType = (j % (numBFrames + 1)) ? ORC_TYPE_B : ORC_TYPE_P;
            frames[numFrames]->sliceType = ORC_TYPE_P;
        }

        int bForceRADL = lp->radl && lp->keyframeMax == lp->keyframeMin;
        int bLastMiniGop = framecnt >= bframesMax + 1 ? 0 : 1;
        int preRADL = la->lastKeyframe + lp->keyframeMax - lp->radl - 1;
        if (bForceRADL && frames[0]->frameNum == preRADL && !bLastMiniGop)
        {
            int j = 1;
            numBFrames = lp->radl;
            for (; j <= numBFrames; j++)
                frames[j]->sliceType = ORC_TYPE_B;
            frames[j]->sliceType = ORC_TYPE_I;
        }
        else
        {
            for (int j = 1; j < numBFrames + 1; j++)
            {
                if (scenecut(la, frames, j, j + 1, 0, origNumFrames) || (bForceRADL && frames[j]->frameNum == preRADL))
                {
                    frames[j]->sliceType = ORC_TYPE_P;
                    numAnalyzed = j;
                    break;
                }
            }
        }
        resetStart = bKeyframe ? 1 : imin(numBFrames + 2, numAnalyzed + 1);
    }
    else
    {
        for (int j = 1; j <= numFrames; j++)
            frames[j]->sliceType = ORC_TYPE_P;
        resetStart = bKeyframe ? 1 : 2;
    }

    if (lp->bAQMotion)
        aq_motion(la, frames, bKeyframe);

    if (lp->cuTree)
        cu_tree(la, frames, imin(numFrames, lp->keyframeMax), bKeyframe);

    if (lp->gopLookahead && keyFrameLimit >= 0 && keyFrameLimit <= bframesMax + 1 && !la->extendGopBoundary)
        keyintLimit = keyFrameLimit;

    if (!lp->bIntraRefresh)
        for (int j = keyintLimit + 1; j <= numFrames; j += lp->keyframeMax)
        {
            frames[j]->sliceType = ORC_TYPE_I;
            resetStart = imin(resetStart, j + 1);
        }

    if (bIsVbvLookahead)
        vbv_lookahead(la, frames, numFrames, bKeyframe);
    int maxp1 = imin(bframesMax + 1, origNumFrames);

    for (int j = resetStart; j <= numFrames; j++)
    {
        frames[j]->sliceType = ORC_TYPE_AUTO;
        if (j <= maxp1 && frames[j]->bScenecut && la->isSceneTransition)
            la->isSceneTransition = 0;
    }
}

/* ---- :1864-1908 ---- */
static void place_bref(orc_la* la, orc_lowres** list, int start, int end, int num, int* brefs)
{
    int avg = (start + end) / 2;
    if (la->lp.bEnableTemporalSubLayers < 2)
    {
        list[avg]->sliceType = ORC_TYPE_BREF;
        (*brefs)++;
        return;
    }
    if (num <= 2)
        return;
    list[avg]->sliceType = ORC_TYPE_BREF;
    (*brefs)++;
    place_bref(la, list, start, avg, avg - start, brefs);
    place_bref(la, list, avg + 1, end, end - avg, brefs);
}

static void comp_cost_bref(orc_la* la, orc_lowres** frames, int start, int end, int num)
{
    int avg = (start + end) / 2;
    if (num <= 2)
    {
        for (int i = start; i < end; i++)
            single_cost(la, frames, start, end + 1, i + 1, 0);
        return;
    }
    single_cost(la, frames, start, end + 1, avg + 1, 0);
    comp_cost_bref(la, frames, start, avg, avg - start);
    comp_cost_bref(la, frames, avg + 1, end, end - avg);
}

static void out_push(orc_la* la, orc_lowres* f) { la->outq[(la->outHead + la->outCount++) % QCAP] = f; }

/* ---- :2607-2716: the mini-GOP list[0..bframes] without the temporal-layer orders ---- */
static void plain_mini_gop(orc_la* la, orc_lowres** frames, orc_lowres** list, int bframes, int* brefs, int* maxSearch)
{
    const orc_la_param* lp = &la->lp;
    if (bframes)
        list[bframes - 1]->bLastMiniGopBFrame = 1;
    list[bframes]->leadingBframes = bframes;
    la->lastNonB = list[bframes];

    if (lp->bBPyramid && bframes > 1 && !*brefs)
        place_bref(la, list, 0, bframes, bframes + 1, brefs);

    if (!lp->bCqp)
    {
        int p0, p1, b;
        if (!*maxSearch)
            for (int i = 0; i <= bframes; i++)
                frames[i + 1] = list[i];
        p1 = b = bframes + 1;
        p0 = IS_TYPE_I(frames[bframes + 1]->sliceType) ? b : 0;
        single_cost(la, frames, p0, p1, b, 0);
        if (lp->bEnableTemporalSubLayers > 1 && bframes)
            comp_cost_bref(la, frames, 0, bframes, bframes + 1);
        else if (bframes)
        {
            p0 = 0;
            int isp0available = frames[bframes + 1]->sliceType == ORC_TYPE_IDR ? 0 : 1;
            for (b = 1; b <= bframes; b++)
            {
                if (!isp0available)
                    p0 = b;
                if (frames[b]->sliceType == ORC_TYPE_B)
                    for (p1 = b; frames[p1]->sliceType == ORC_TYPE_B; p1++)
                        ;
                else
                    p1 = bframes + 1;
                single_cost(la, frames, p0, p1, b, 0);
                if (frames[b]->sliceType == ORC_TYPE_BREF)
                {
                    p0 = b;
                    isp0available = 1;
                }
            }
        }
    }

    /* dequeue the mini-GOP: non-B first, then B-refs, then the other B frames (:2672-2715) */
    memmove(la->inq, la->inq + bframes + 1, sizeof(la->inq[0]) * (la->inCount - bframes - 1));
    la->inCount -= bframes + 1;
    *maxSearch -= bframes + 1;
    out_push(la, list[bframes]);
    if (*brefs)
        for (int i = 0; i < bframes; i++)
            if (list[i]->sliceType == ORC_TYPE_BREF)
                out_push(la, list[i]);
    for (int i = 0; i < bframes; i++)
        if (list[i]->sliceType != ORC_TYPE_BREF)
            out_push(la, list[i]);

}

/* x265_gop_ra / x265_gop_ra_length (source/x265.h:835-): coding order of the hierarchical mini-GOPs of 4, 8 and 16
 * pictures, POC offset and temporal layer of each position */
static const int8_t gop_ra_length[3] = { 4, 8, 16 };
static const int8_t gop_ra_poc[3][16] = {
    { 4, 2, 1, 3 },
    { 8, 4, 2, 1, 3, 6, 5, 7 },
    { 16, 8, 4, 2, 1, 3, 6, 5, 7, 12, 10, 9, 11, 14, 13, 15 },
};
static const int8_t gop_ra_layer[3][16] = {
    { 0, 1, 2, 2 },
    { 0, 1, 2, 3, 3, 2, 3, 3 },
    { 0, 1, 2, 3, 4, 4, 3, 4, 4, 2, 3, 4, 4, 3, 4, 4 },
};

static void dequeue(orc_la* la, int count, int* maxSearch)
{
    memmove(la->inq, la->inq + count, sizeof(la->inq[0]) * (la->inCount - count));
    la->inCount -= count;
    *maxSearch -= count;
}

/* ---- :2304-2571 ---- */
static void temporal_mini_gop(orc_la* la, orc_lowres** frames, orc_lowres** list, int bframes, int* brefs, int* maxSearch)
{
    const orc_la_param* lp = &la->lp;
    if (bframes < la->p.bframes)
    {
        /* a partial mini-GOP: sub-GOPs of the lengths that fit, longest first, then a plain remainder */
        int leftOver = bframes + 1;
        int gopId = la->gopId - 1;
        int gopLen = gopId >= 0 ? gop_ra_length[gopId] : 0;
        int listReset = 0;
        while (gopId >= 0 && leftOver > 3)
        {
            if (leftOver < gopLen)
            {
                gopId = gopId - 1;
                gopLen = gopId >= 0 ? gop_ra_length[gopId] : 0;
                continue;
            }
            int newbFrames = listReset + gopLen - 1;
            if (!IS_TYPE_I(list[newbFrames]->sliceType))
                list[newbFrames]->sliceType = ORC_TYPE_P;
            if (newbFrames)
                list[newbFrames - 1]->bLastMiniGopBFrame = 1;
            list[newbFrames]->leadingBframes = newbFrames;
            la->lastNonB = list[newbFrames];
            if (lp->bBPyramid && newbFrames)
                place_bref(la, list, listReset, newbFrames, newbFrames + 1, brefs);
            if (!lp->bCqp)
            {
                /* (:2343-2347 needs maxSearch == 0, which cannot be while pictures of this list remain) */
                int b = newbFrames + 1;
                int p0 = IS_TYPE_I(frames[b]->sliceType) ? b : listReset;
                single_cost(la, frames, p0, b, b, 0);
                if (newbFrames)
                    comp_cost_bref(la, frames, listReset, newbFrames, newbFrames + 1);
            }
            dequeue(la, gopLen, maxSearch);
            list[newbFrames]->gopOffset = 0;
            list[newbFrames]->gopId = gopId;
            list[newbFrames]->tempLayer = gop_ra_layer[gopId][0];
            out_push(la, list[newbFrames]);
            for (int j = 1; j < gopLen; j++)
            {
                int offset = listReset + (gop_ra_poc[gopId][j] - 1);
                list[offset]->gopOffset = j;
                /* :2393 writes list[bframes]->m_gopId here, which leaves the B pictures' field unset (it is never
                 * initialised); the id is stored where :2563 stores it for a full mini-GOP */
                list[offset]->gopId = gopId;
                list[offset]->tempLayer = gop_ra_layer[gopId][j];
                out_push(la, list[offset]);
            }
            listReset += gopLen;
            leftOver -= gopLen;
            gopId -= 1;
            gopLen = gopId >= 0 ? gop_ra_length[gopId] : 0;
        }
        if (leftOver > 0 && leftOver < 4)
        {
            int newbFrames = listReset + leftOver - 1;
            if (!IS_TYPE_I(list[newbFrames]->sliceType))
                list[newbFrames]->sliceType = ORC_TYPE_P;
            if (newbFrames)
                list[newbFrames - 1]->bLastMiniGopBFrame = 1;
            list[newbFrames]->leadingBframes = newbFrames;
            la->lastNonB = list[newbFrames];
            if (lp->bBPyramid && (newbFrames - listReset) > 1)
                place_bref(la, list, listReset, newbFrames, newbFrames + 1, brefs);
            if (!lp->bCqp)
            {
                int b = newbFrames + 1;
                int p0 = IS_TYPE_I(frames[b]->sliceType) ? b : listReset;
                single_cost(la, frames, p0, b, b, 0);
                if (newbFrames)
                    comp_cost_bref(la, frames, listReset, newbFrames, newbFrames + 1);
            }
            dequeue(la, leftOver, maxSearch);
            list[newbFrames]->gopOffset = 0;
            list[newbFrames]->gopId = -1;
            list[newbFrames]->tempLayer = 0;
            out_push(la, list[newbFrames]);
            if (*brefs)
                for (int i = listReset; i < newbFrames; i++)
                    if (list[i]->sliceType == ORC_TYPE_BREF)
                    {
                        list[i]->gopOffset = 0;
                        list[i]->gopId = -1;
                        list[i]->tempLayer = 0;
                        out_push(la, list[i]);
                    }
            for (int i = listReset; i < newbFrames; i++)
                if (list[i]->sliceType != ORC_TYPE_BREF)
                {
                    list[i]->gopOffset = 0;
                    list[i]->gopId = -1;
                    list[i]->tempLayer = 1;
                    out_push(la, list[i]);
                }
        }
        return;
    }

    /* the complete mini-GOP */
    list[bframes - 1]->bLastMiniGopBFrame = 1;
    list[bframes]->leadingBframes = bframes;
    la->lastNonB = list[bframes];
    if (lp->bBPyramid && !*brefs)
        place_bref(la, list, 0, bframes, bframes + 1, brefs);
    if (!lp->bCqp)
    {
        int b = bframes + 1;
        int p0 = IS_TYPE_I(frames[b]->sliceType) ? b : 0;
        single_cost(la, frames, p0, b, b, 0);
        comp_cost_bref(la, frames, 0, bframes, bframes + 1);
    }
    dequeue(la, bframes + 1, maxSearch);
    list[bframes]->gopOffset = 0;
    list[bframes]->gopId = la->gopId;
    list[bframes]->tempLayer = gop_ra_layer[la->gopId][0];
    out_push(la, list[bframes]);
    for (int j = 1; j <= bframes; j++)
    {
        int offset = gop_ra_poc[la->gopId][j] - 1;
        list[offset]->gopOffset = j;
        list[offset]->gopId = la->gopId;
        list[offset]->tempLayer = gop_ra_layer[la->gopId][j];
        out_push(la, list[offset]);
    }
}

/* ---- :1982-2751 ---- */
static void slicetype_decide(orc_la* la)
{
    const orc_la_param* lp = &la->lp;
    orc_lowres* frames[ORC_LOOKAHEAD_MAX + ORC_BFRAME_MAX + 4];
    orc_lowres* list[ORC_BFRAME_MAX + 4];
    memset(frames, 0, sizeof(frames));
    memset(list, 0, sizeof(list));
    int maxSearch = imax(1, imin(lp->lookaheadDepth, ORC_LOOKAHEAD_MAX));

    int j;
    for (j = 0; j < la->p.bframes + 2 && j < la->inCount; j++)
        list[j] = la->inq[j];
    frames[0] = la->lastNonB;
    for (j = 0; j < maxSearch && j < la->inCount; j++)
        frames[j + 1] = la->inq[j];
    maxSearch = j;

    if (lp->bEnableFades)
        detect_fades(la, list);

    if (la->lastNonB && ((lp->bFrameAdaptive && la->p.bframes) || lp->cuTree || lp->scenecutThreshold ||
                         lp->bHistBasedSceneCut || (lp->lookaheadDepth && lp->vbvBufferSize)))
        slicetype_analyse(la, frames, 0);

    int bframes, brefs;
    for (bframes = 0, brefs = 0;; bframes++)
    {
        orc_lowres* frm = list[bframes];
        if (frm->sliceTypeReq != ORC_TYPE_AUTO && frm->sliceTypeReq != frm->sliceType)
            frm->sliceType = frm->sliceTypeReq;
        if (frm->sliceType == ORC_TYPE_BREF && !lp->bBPyramid && brefs == lp->bBPyramid)
            frm->sliceType = ORC_TYPE_B;
        else if (frm->sliceType == ORC_TYPE_BREF && lp->bBPyramid && brefs && lp->maxNumReferences <= (brefs + 3))
            frm->sliceType = ORC_TYPE_B;
        /* chunkStart / chunkEnd stay at their default 0 here: their terms at :2139 and :2166 then only
         * match frame 0, which the keyframe-interval test makes a keyframe anyway */
        if ((!lp->bIntraRefresh || frm->frameNum == 0) && frm->frameNum - la->lastKeyframe >= lp->keyframeMax &&
            (!la->extendGopBoundary || frm->frameNum - la->lastKeyframe >= lp->keyframeMax + lp->gopLookahead))
        {
            if (frm->sliceType == ORC_TYPE_AUTO || frm->sliceType == ORC_TYPE_I)
                frm->sliceType = lp->bOpenGOP && la->lastKeyframe >= 0 ? ORC_TYPE_I : ORC_TYPE_IDR;
            int warn = frm->sliceType != ORC_TYPE_IDR;
            if (warn && lp->bOpenGOP)
                warn &= frm->sliceType != ORC_TYPE_I;
            if (warn)
                frm->sliceType = lp->bOpenGOP && la->lastKeyframe >= 0 ? ORC_TYPE_I : ORC_TYPE_IDR;
        }
        if (frm->bIsFadeEnd)           /* :2153-2155 */
            frm->sliceType = lp->bOpenGOP && la->lastKeyframe >= 0 ? ORC_TYPE_I : ORC_TYPE_IDR;
        if (frm->sliceType == ORC_TYPE_I && frm->frameNum - la->lastKeyframe >= lp->keyframeMin)
        {
            if (lp->bOpenGOP)
            {
                la->lastKeyframe = frm->frameNum;
                frm->bKeyframe = 1;
            }
            else
                frm->sliceType = ORC_TYPE_IDR;
        }
        int isClosedGopRadl = lp->radl && lp->keyframeMax != lp->keyframeMin;
        if (frm->sliceType == ORC_TYPE_IDR && frm->bScenecut && isClosedGopRadl)
        {
            for (int i = bframes; i < bframes + lp->radl; i++)
                list[i]->sliceType = ORC_TYPE_B;
            list[bframes + lp->radl]->sliceType = ORC_TYPE_IDR;
        }
        if (frm->sliceType == ORC_TYPE_IDR)
        {
            la->lastKeyframe = frm->frameNum;
            frm->bKeyframe = 1;
            if (bframes > 0 && !lp->radl)
            {
                list[bframes - 1]->sliceType = ORC_TYPE_P;
                bframes--;
            }
        }
        if (bframes == la->p.bframes || !list[bframes + 1])
        {
            if (frm->sliceType == ORC_TYPE_AUTO || IS_TYPE_B(frm->sliceType))
                frm->sliceType = ORC_TYPE_P;
        }
        if (frm->sliceType == ORC_TYPE_BREF)
            brefs++;
        if (frm->sliceType == ORC_TYPE_AUTO)
            frm->sliceType = ORC_TYPE_B;
        else if (!IS_TYPE_B(frm->sliceType))
            break;
    }

    if (lp->bEnableTemporalSubLayers > 2)
        temporal_mini_gop(la, frames, list, bframes, &brefs, &maxSearch);
    else
        plain_mini_gop(la, frames, list, bframes, &brefs, &maxSearch);

    int isKeyFrameAnalyse = lp->cuTree || (lp->vbvBufferSize && lp->lookaheadDepth);
    if (isKeyFrameAnalyse && IS_TYPE_I(la->lastNonB->sliceType))
    {
        frames[0] = la->lastNonB;
        for (j = 0; j < maxSearch; j++)
            frames[j + 1] = la->inq[j];
        frames[j + 1] = NULL;
        slicetype_analyse(la, frames, 1);
    }
}

/* ---- queue interface: :1299-1431 without the locks ---- */
orc_la* orc_la_create(const orc_param* p, const orc_la_param* lp, const orc_cutree_param* cp, orc_tld* tld)
{
    orc_la* la = (orc_la*)calloc(1, sizeof(orc_la));
    la->p = *p;
    la->lp = *lp;
    la->cp = *cp;
    la->tld = tld;
    la->lastKeyframe = -lp->keyframeMax;
    la->fullQueueSize = imax(1, lp->lookaheadDepth);
    la->batchMotionSearch = lp->poolWorkers > 0 && lp->bFrameAdaptive == 2;
    la->batchFrameCosts = la->batchMotionSearch;
    la->blocks8x8 = tld->ncu;
    la->resetRunningAvg = 1;
    la->fadeStart = -1;            /* :1067-1069 */
    la->gopId = lp->bEnableTemporalSubLayers > 2 ? lp->bEnableTemporalSubLayers - 3 : 0;
    la->segmentCountThreshold = (uint32_t)(((float)((4 * 4) * 50) / 100) + 0.5);
    return la;
}

void orc_la_destroy(orc_la* la)
{
    if (la)
    {
        free(la->trace);
        free(la);
    }
}

/* sliceType: Lookahead::addPicture's argument (AUTO unless a first pass dictates types); sliceTypeReq: the
 * picture's own request, which Encoder::encode stores beside it (source/encoder/encoder.cpp:1815) */
void orc_la_add_picture(orc_la* la, orc_lowres* frame, int sliceType, int sliceTypeReq)
{
    if (!la->filled)
    {
        if (!la->p.bframes & !la->lp.lookaheadDepth)
            la->filled = 1;
        else if (la->inputCount >= la->lp.lookaheadDepth + 2 + la->p.bframes)
            la->filled = 1;
    }
    frame->sliceType = sliceType;
    frame->sliceTypeReq = sliceTypeReq;
    frame->bScenecut = 0;              /* a Frame starts zeroed (source/common/frame.cpp:50, encoder.cpp:1690) */
    la->inq[la->inCount++] = frame;
    la->inputCount++;
}

void orc_la_flush(orc_la* la)
{
    la->fullQueueSize = 1;
    la->filled = 1;
}

static orc_lowres* out_pop(orc_la* la)
{
    if (!la->outCount)
        return NULL;
    orc_lowres* f = la->outq[la->outHead];
    la->outHead = (la->outHead + 1) % QCAP;
    la->outCount--;
    la->inputCount--;
    return f;
}

orc_lowres* orc_la_get_decided_picture(orc_la* la)
{
    if (!la->filled)
        return NULL;
    orc_lowres* out = out_pop(la);
    if (out)
        return out;
    if (la->inCount >= la->fullQueueSize)
        slicetype_decide(la);
    return out_pop(la);
}

int orc_la_trace(const orc_la* la, const int32_t** out)
{
    *out = la->trace;
    return la->ntrace;
}
