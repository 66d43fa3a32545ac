/* TEST ORACLE - not part of the product.  CPU restatement of x265's cuTree
 * propagation, used only by tests to check the HIP path.
 *
 * Follows Lookahead::cuTree, estimateCUPropagate and cuTreeFinish
 * (source/encoder/slicetype.cpp:3665-3874, 4016-4064) and the C primitive
 * estimateCUPropagateCost (source/common/pixel.cpp:906-932), without hevcAq; the frame walk for
 * qgSize >= 16, the closing offsets also for qgSize 8 (orc_cutree_finish_qg8). */
#include "orc.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int imin(int a, int b) { return a < b ? a : b; }

static double clip_duration(double f) /* CLIP_DURATION, source/encoder/ratecontrol.h:44-49 */
{
    return f < 0.01 ? 0.01 : (f > 1.00 ? 1.00 : f);
}

/* source/common/pixel.cpp:906-932 */
void orc_propagate_cost(int* dst, const uint16_t* propagateIn, const int32_t* intraCosts, const uint16_t* interCosts,
                        const int32_t* invQscales, const double* fpsFactor, int len)
{
    double fps = *fpsFactor / 256;
    for (int i = 0; i < len; i++)
    {
        int intraCost = intraCosts[i];
        int interCost = imin(intraCosts[i], interCosts[i] & ORC_LOWRES_COST_MASK);
        double propagateIntra = intraCost * invQscales[i];
        double propagateAmount = (double)propagateIn[i] + propagateIntra * fps;
        double propagateNum = (double)(intraCost - interCost);
        double propagateDenom = (double)intraCost;
        dst[i] = (int)(propagateAmount * propagateNum / propagateDenom + 0.5);
    }
}

static void clip_add(uint16_t* s, int x)
{
    int v = *s + x;
    *s = (uint16_t)(v < 65535 ? v : 65535);
}

/* source/encoder/slicetype.cpp:4016-4064 (the qgSize > 8, non-hevcAq branch) */
void orc_cutree_finish_qg8(int W, int H, const int32_t* intraCost, const int32_t* invQscale8x8,
                           const uint16_t* propagateCost, int fpsFactor, double weightdelta, double strength,
                           const double* qpAqOffset, double* qpCuTreeOffset);

void orc_lowres_set_hevc_layer(orc_lowres* lr, int k, int partW, int partH, int numW, int numH, const double* dQpOffset,
                               const double* dCuTreeOffset)
{
    size_t n = (size_t)numW * numH;
    lr->aqLayer[k].partW = partW;
    lr->aqLayer[k].partH = partH;
    lr->aqLayer[k].numW = numW;
    lr->aqLayer[k].numH = numH;
    free(lr->aqLayer[k].dQpOffset);
    free(lr->aqLayer[k].dCuTreeOffset);
    lr->aqLayer[k].dQpOffset = (double*)malloc(n * sizeof(double));
    lr->aqLayer[k].dCuTreeOffset = (double*)malloc(n * sizeof(double));
    memcpy(lr->aqLayer[k].dQpOffset, dQpOffset, n * sizeof(double));
    memcpy(lr->aqLayer[k].dCuTreeOffset, dCuTreeOffset, n * sizeof(double));
    if (k + 1 > lr->numAqLayers)
        lr->numAqLayers = k + 1;
    if (lr->qg8 && !lr->cuTreeOffset8x8)
        lr->cuTreeOffset8x8 = (double*)calloc((size_t)lr->cuCount * 4, sizeof(double));   /* PicQPAdaptationLayer::create */
}

double* orc_lowres_hevc_layer(orc_lowres* lr, int k, int which, int* count)
{
    *count = lr->aqLayer[k].numW * lr->aqLayer[k].numH;
    return which ? lr->aqLayer[k].dCuTreeOffset : lr->aqLayer[k].dQpOffset;
}

/* Lookahead::computeCUTreeQpOffset (source/encoder/slicetype.cpp:3876-4014) over the frame's layers */
static void compute_cutree_qp_offset(const orc_cutree_param* cp, orc_lowres* frame, double averageDuration, int ref0Distance)
{
    int fpsFactor = (int)(clip_duration(averageDuration) / clip_duration((double)cp->fpsDenom / cp->fpsNum) * 256);
    double weightdelta = 0.0;
    if (ref0Distance && frame->weightedCostDelta[ref0Distance - 1] > 0)
        weightdelta = 1.0 - frame->weightedCostDelta[ref0Distance - 1];
    double strength = 6.0 * (1.0 - cp->qCompress);       /* m_cuTreeStrength with rc.hevcAq, :1075 */
    const int W = frame->widthInCU, H = frame->heightInCU;
    if (frame->qg8)
        orc_cutree_ratio8x8(W, H, frame->intraCost, frame->invQscaleFactor, frame->propagateCost, fpsFactor, weightdelta,
                            frame->cuTreeOffset8x8);
    for (int k = 0; k < frame->numAqLayers; k++)
        orc_cutree_hevc_layer(frame->qg8, frame->qg8 ? 2 * W : W, frame->sourceWidth, frame->sourceHeight,
                              frame->aqLayer[k].partW, frame->aqLayer[k].partH, frame->aqLayer[k].numW, frame->aqLayer[k].numH,
                              frame->intraCost, frame->invQscaleFactor, frame->propagateCost, fpsFactor, weightdelta, strength,
                              frame->cuTreeOffset8x8, frame->aqLayer[k].dQpOffset, frame->aqLayer[k].dCuTreeOffset);
}

void orc_cutree_finish(const orc_cutree_param* cp, orc_lowres* frame, double averageDuration, int ref0Distance)
{
    if (frame->numAqLayers)
    {
        compute_cutree_qp_offset(cp, frame, averageDuration, ref0Distance);
        return;
    }
    if (frame->qg8)
    {
        int fpsFactor8 = (int)(clip_duration(averageDuration) / clip_duration((double)cp->fpsDenom / cp->fpsNum) * 256);
        double wd = 0.0;
        if (ref0Distance && frame->weightedCostDelta[ref0Distance - 1] > 0)
            wd = 1.0 - frame->weightedCostDelta[ref0Distance - 1];
        orc_cutree_finish_qg8(frame->widthInCU, frame->heightInCU, frame->intraCost, frame->invQscaleFactor,
                              frame->propagateCost, fpsFactor8, wd, 5.0 * (1.0 - cp->qCompress), frame->qpAqOffset,
                              frame->qpCuTreeOffset);
        return;
    }
    int fpsFactor = (int)(clip_duration(averageDuration) / clip_duration((double)cp->fpsDenom / cp->fpsNum) * 256);
    double strength = 5.0 * (1.0 - cp->qCompress);      /* m_cuTreeStrength, :1075 */
    double weightdelta = 0.0;
    if (ref0Distance && frame->weightedCostDelta[ref0Distance - 1] > 0)
        weightdelta = 1.0 - frame->weightedCostDelta[ref0Distance - 1];
    for (int i = 0; i < frame->cuCount; i++)
    {
        int intracost = (frame->intraCost[i] * frame->invQscaleFactor[i] + 128) >> 8;
        if (intracost)
        {
            int propagateCost = (frame->propagateCost[i] * fpsFactor + 128) >> 8;
            double log2_ratio = log2(intracost + propagateCost) - log2(intracost) + weightdelta;
            frame->qpCuTreeOffset[i] = frame->qpAqOffset[i] - strength * log2_ratio;
        }
    }
}

/* source/encoder/slicetype.cpp:4030-4049, the qgSize 8 loop of cuTreeFinish on plain arrays: invQscale8x8
 * is Lowres::invQscaleFactor8x8 (W x H), the offset arrays hold (2W) x (2H) doubles. */
void orc_cutree_finish_qg8(int W, int H, const int32_t* intraCost, const int32_t* invQscale8x8,
                           const uint16_t* propagateCost, int fpsFactor, double weightdelta, double strength,
                           const double* qpAqOffset, double* qpCuTreeOffset)
{
    const int rowFull = W * 2;
    for (int cuY = 0; cuY < H; cuY++)
        for (int cuX = 0; cuX < W; cuX++)
        {
            const int cuXY = cuX + cuY * W;
            int intracost = ((intraCost[cuXY]) / 4 * invQscale8x8[cuXY] + 128) >> 8;
            if (intracost)
            {
                int propagate = ((propagateCost[cuXY]) / 4 * fpsFactor + 128) >> 8;
                double log2_ratio = log2(intracost + propagate) - log2(intracost) + weightdelta;
                const int at = cuX * 2 + cuY * W * 4;
                qpCuTreeOffset[at] = qpAqOffset[at] - strength * (log2_ratio);
                qpCuTreeOffset[at + 1] = qpAqOffset[at + 1] - strength * (log2_ratio);
                qpCuTreeOffset[at + rowFull] = qpAqOffset[at + rowFull] - strength * (log2_ratio);
                qpCuTreeOffset[at + rowFull + 1] = qpAqOffset[at + rowFull + 1] - strength * (log2_ratio);
            }
        }
}

/* Lookahead::computeCUTreeQpOffset (source/encoder/slicetype.cpp:3876-4014) for ONE layer, on plain arrays.
 * qg8: pcCuTree8x8 holds the per-block ratios the caller formed with orc_cutree_ratio8x8. */
void orc_cutree_hevc_layer(int qg8, int maxCols, int widthFullRes, int heightFullRes, int aqPartWidth, int aqPartHeight,
                           int numAQPartInWidth, int numAQPartInHeight, const int32_t* intraCost,
                           const int32_t* invQscaleFactor, const uint16_t* propagateCostArr, int fpsFactor,
                           double weightdelta, double strength, const double* pcCuTree8x8, const double* pcQP,
                           double* pcCuTree)
{
    const uint32_t loopIncr = qg8 ? 8 : 16;
    for (uint32_t y = 0; y < (uint32_t)numAQPartInHeight; y++)
        for (uint32_t x = 0; x < (uint32_t)numAQPartInWidth; x++, pcQP++, pcCuTree++)
        {
            uint32_t block_x = x * aqPartWidth, block_y = y * aqPartHeight;
            uint32_t blockXY = 0;
            double log2_ratio = 0;
            for (uint32_t block_yy = block_y; block_yy < block_y + aqPartHeight && block_yy < (uint32_t)heightFullRes; block_yy += loopIncr)
                for (uint32_t block_xx = block_x; block_xx < block_x + aqPartWidth && block_xx < (uint32_t)widthFullRes; block_xx += loopIncr)
                {
                    uint32_t idx = ((block_yy / loopIncr) * (maxCols)) + (block_xx / loopIncr);
                    if (qg8)
                        log2_ratio += *(pcCuTree8x8 + idx);
                    else
                    {
                        int intra = (intraCost[idx] * invQscaleFactor[idx] + 128) >> 8;
                        int propagate = (propagateCostArr[idx] * fpsFactor + 128) >> 8;
                        log2_ratio += (log2(intra + propagate) - log2(intra) + weightdelta);
                    }
                    blockXY++;
                }
            double qp_offset = (strength * log2_ratio) / blockXY;
            *pcCuTree = *pcQP - qp_offset;
        }
}

/* the first loop of computeCUTreeQpOffset's qgSize 8 branch (:3895-3912) */
void orc_cutree_ratio8x8(int W, int H, const int32_t* intraCost, const int32_t* invQscale8x8, const uint16_t* propagateCostArr,
                         int fpsFactor, double weightdelta, double* pcCuTree8x8)
{
    const int rowFull = W * 2;
    for (int cuY = 0; cuY < H; cuY++)
        for (int cuX = 0; cuX < W; cuX++)
        {
            const int cuXY = cuX + cuY * W;
            int intracost = ((intraCost[cuXY] / 4) * invQscale8x8[cuXY] + 128) >> 8;
            if (intracost)
            {
                int propagate = ((propagateCostArr[cuXY] / 4) * fpsFactor + 128) >> 8;
                double log2_ratio = log2(intracost + propagate) - log2(intracost) + weightdelta;
                pcCuTree8x8[cuX * 2 + cuY * W * 4] = log2_ratio;
                pcCuTree8x8[cuX * 2 + cuY * W * 4 + 1] = log2_ratio;
                pcCuTree8x8[cuX * 2 + cuY * W * 4 + rowFull] = log2_ratio;
                pcCuTree8x8[cuX * 2 + cuY * W * 4 + rowFull + 1] = log2_ratio;
            }
        }
}

/* source/encoder/slicetype.cpp:3768-3874 */
void orc_estimate_cu_propagate(const orc_cutree_param* cp, orc_lowres** frames, int* scratch, double averageDuration,
                               int p0, int p1, int b, int referenced)
{
    orc_lowres* fb = frames[b];
    const int W = fb->widthInCU, H = fb->heightInCU;
    uint16_t* refCosts[2] = { frames[p0]->propagateCost, frames[p1]->propagateCost };
    int32_t distScaleFactor = (((b - p0) << 8) + ((p1 - p0) >> 1)) / (p1 - p0);
    int32_t bipredWeight = cp->bEnableWeightedBiPred ? 64 - (distScaleFactor >> 2) : 32;
    int32_t bipredWeights[2] = { bipredWeight, 64 - bipredWeight };
    int listDist[2] = { b - p0, p1 - b };
    const uint16_t* lowresCosts = fb->lowresCosts[b - p0][p1 - b];

    memset(scratch, 0, W * sizeof(int));
    uint16_t* propagateCost = fb->propagateCost;
    double fpsFactor = clip_duration((double)cp->fpsDenom / cp->fpsNum) / clip_duration(averageDuration);
    if (!referenced)
        memset(fb->propagateCost, 0, W * sizeof(uint16_t));

    for (int blocky = 0; blocky < H; blocky++)
    {
        int cuIndex = blocky * W;
        orc_propagate_cost(scratch, propagateCost, fb->intraCost + cuIndex, lowresCosts + cuIndex,
                           fb->invQscaleFactor + cuIndex, &fpsFactor, W);
        if (referenced)
            propagateCost += W;
        for (int blockx = 0; blockx < W; blockx++, cuIndex++)
        {
            int32_t amount = scratch[blockx];
            if (amount <= 0)
                continue;
            int32_t lists_used = lowresCosts[cuIndex] >> ORC_LOWRES_COST_SHIFT;
            for (int list = 0; list < 2; list++)
            {
                if (!((lists_used >> list) & 1))
                    continue;
                int32_t listamount = amount;
                if (lists_used == 3)
                    listamount = (listamount * bipredWeights[list] + 32) >> 6;
                const orc_mv* mvs = fb->lowresMvs[list][listDist[list]];
                if (!mvs[cuIndex].x && !mvs[cuIndex].y)
                {
                    clip_add(&refCosts[list][cuIndex], listamount);
                    continue;
                }
                int32_t x = mvs[cuIndex].x, y = mvs[cuIndex].y;
                int32_t cux = (x >> 5) + blockx, cuy = (y >> 5) + blocky;
                int32_t idx0 = cux + cuy * W;
                x &= 31;
                y &= 31;
                const int32_t weight[4] = { (32 - y) * (32 - x), (32 - y) * x, y * (32 - x), y * x };
                for (int k = 0; k < 4; k++)
                {
                    int tx = cux + (k & 1), ty = cuy + (k >> 1);
                    if (tx >= 0 && tx < W && ty >= 0 && ty < H)
                        clip_add(&refCosts[list][idx0 + (k & 1) + (k >> 1) * W], (listamount * weight[k] + 512) >> 10);
                }
            }
        }
    }
    if (cp->vbvBufferSize && cp->lookaheadDepth && referenced)
        orc_cutree_finish(cp, fb, averageDuration, b == p1 ? b - p0 : 0);
}

static void swap_propagate(orc_lowres* a, orc_lowres* b)
{
    uint16_t* t = a->propagateCost;
    a->propagateCost = b->propagateCost;
    b->propagateCost = t;
}

/* source/encoder/slicetype.cpp:3665-3766 */
void orc_cutree(orc_tld* t, const orc_param* p, const orc_cutree_param* cp, orc_lowres** frames, int numframes, int bIntra)
{
    int idx = !bIntra;
    int lastnonb, curnonb = 1;
    int bframes = 0;
    const int cuCount = frames[0]->cuCount;
    int* scratch = (int*)malloc(sizeof(int) * frames[0]->widthInCU);

    double totalDuration = 0.0;
    for (int j = 0; j <= numframes; j++)
        totalDuration += (double)cp->fpsDenom / cp->fpsNum;
    double averageDuration = totalDuration / (numframes + 1);

    int i = numframes;
    while (i > 0 && frames[i]->sliceType == ORC_TYPE_B)
        i--;
    lastnonb = i;

    if (!cp->lookaheadDepth)
    {
        if (bIntra)
        {
            memset(frames[0]->propagateCost, 0, cuCount * sizeof(uint16_t));
            memcpy(frames[0]->qpCuTreeOffset, frames[0]->qpAqOffset, cuCount * (frames[0]->qg8 ? 4 : 1) * sizeof(double));
            free(scratch);
            return;
        }
        swap_propagate(frames[lastnonb], frames[0]);
        memset(frames[0]->propagateCost, 0, cuCount * sizeof(uint16_t));
    }
    else
    {
        if (lastnonb < idx)
        {
            free(scratch);
            return;
        }
        memset(frames[lastnonb]->propagateCost, 0, cuCount * sizeof(uint16_t));
    }

    while (i-- > idx)
    {
        curnonb = i;
        while (frames[curnonb]->sliceType == ORC_TYPE_B && curnonb > 0)
            curnonb--;
        if (curnonb < idx)
            break;

        orc_estimate_frame_cost_single(t, p, frames, curnonb, lastnonb, lastnonb, 0);
        memset(frames[curnonb]->propagateCost, 0, cuCount * sizeof(uint16_t));
        bframes = lastnonb - curnonb - 1;
        if (cp->bBPyramid && bframes > 1)
        {
            int middle = (bframes + 1) / 2 + curnonb;
            orc_estimate_frame_cost_single(t, p, frames, curnonb, lastnonb, middle, 0);
            memset(frames[middle]->propagateCost, 0, cuCount * sizeof(uint16_t));
            while (i > curnonb)
            {
                int p0 = i > middle ? middle : curnonb;
                int p1 = i < middle ? middle : lastnonb;
                if (i != middle)
                {
                    orc_estimate_frame_cost_single(t, p, frames, p0, p1, i, 0);
                    orc_estimate_cu_propagate(cp, frames, scratch, averageDuration, p0, p1, i, 0);
                }
                i--;
            }
            orc_estimate_cu_propagate(cp, frames, scratch, averageDuration, curnonb, lastnonb, middle, 1);
        }
        else
        {
            while (i > curnonb)
            {
                orc_estimate_frame_cost_single(t, p, frames, curnonb, lastnonb, i, 0);
                orc_estimate_cu_propagate(cp, frames, scratch, averageDuration, curnonb, lastnonb, i, 0);
                i--;
            }
        }
        orc_estimate_cu_propagate(cp, frames, scratch, averageDuration, curnonb, lastnonb, lastnonb, 1);
        lastnonb = curnonb;
    }

    if (!cp->lookaheadDepth)
    {
        orc_estimate_frame_cost_single(t, p, frames, 0, lastnonb, lastnonb, 0);
        orc_estimate_cu_propagate(cp, frames, scratch, averageDuration, 0, lastnonb, lastnonb, 1);
        swap_propagate(frames[lastnonb], frames[0]);
    }

    orc_cutree_finish(cp, frames[lastnonb], averageDuration, lastnonb);
    if (cp->bBPyramid && bframes > 1 && !cp->vbvBufferSize)
        orc_cutree_finish(cp, frames[lastnonb + (bframes + 1) / 2], averageDuration, 0);
    free(scratch);
}
