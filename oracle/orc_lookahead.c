/* CPU oracle (test infrastructure only; parity unpinned, see orc.h).
 * Per-frame lookahead state, the 8x8 intra estimate and the per-block
 * inter/bidir cost estimation, restated from source/common/lowres.cpp:78-422
 * and source/encoder/slicetype.cpp:706-971, 4284-4557 (whole-frame mode,
 * i.e. no cooperative slices). */
#include "orc.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ---------------------------------------------------------------- frames */

/* source/common/lowres.cpp:78-219 (sizes, strides, array shapes) */
orc_lowres* orc_lowres_create(const orc_param* p, int withAq)
{
    orc_lowres* lr = (orc_lowres*)calloc(1, sizeof(orc_lowres));
    lr->rp.isLowres = 1;
    lr->rp.isHMELowres = p->bEnableHME ? 1 : 0;
    lr->bframes = p->bframes;
    lr->marginX = p->maxCUSize + 32;    /* source/common/picyuv.cpp:91-92 */
    lr->marginY = p->maxCUSize + 16;
    int width = p->sourceWidth / 2, lines = p->sourceHeight / 2;
    intptr_t stride = width + 2 * lr->marginX;
    if (stride & 31)
        stride += 32 - (stride & 31);
    lr->rp.stride = stride;
    lr->widthInCU = (width + ORC_CU - 1) >> 3;
    lr->heightInCU = (lines + ORC_CU - 1) >> 3;
    lr->cuCount = lr->widthInCU * lr->heightInCU;
    lr->width = lr->widthInCU * ORC_CU;
    lr->lines = lr->heightInCU * ORC_CU;
    lr->planesize = (size_t)stride * (lr->lines + 2 * lr->marginY);
    lr->padoffset = (size_t)stride * lr->marginY + lr->marginX;

    lr->buffer[0] = (pixel*)calloc(4 * lr->planesize, sizeof(pixel));
    for (int i = 0; i < 4; i++)
    {
        lr->buffer[i] = lr->buffer[0] + i * lr->planesize;
        lr->rp.plane[i] = lr->buffer[i] + lr->padoffset;
    }
    if (p->bEnableHME || p->bEnableTemporalFilter)
    {
        size_t half = lr->planesize / 2, padHalf = lr->padoffset / 2;
        lr->lowerResBuffer[0] = (pixel*)calloc(4 * half, sizeof(pixel));
        for (int i = 0; i < 4; i++)
        {
            lr->lowerResBuffer[i] = lr->lowerResBuffer[0] + i * half;
            lr->rp.lowerPlane[i] = lr->lowerResBuffer[i] + padHalf;
        }
        /* loop bounds follow Lookahead::m_4x4Width/Height
         * (source/encoder/slicetype.cpp:1063-1064); the arrays are sized for
         * the larger of that and the allocation formula of
         * source/common/lowres.cpp:211-213 */
        lr->widthInCU4 = ((p->sourceWidth / 4) + ORC_CU - 1) >> 3;
        lr->heightInCU4 = ((p->sourceHeight / 4) + ORC_CU - 1) >> 3;
    }
    if (withAq)
    {
        lr->invQscaleFactor = (int*)calloc(lr->cuCount, sizeof(int));
        lr->propagateCost = (uint16_t*)calloc(lr->cuCount, sizeof(uint16_t));
        lr->qg8 = p->qgSize == 8;
        lr->sourceWidth = p->sourceWidth;
        lr->sourceHeight = p->sourceHeight;
        lr->qpAqOffset = (double*)calloc((size_t)lr->cuCount * (lr->qg8 ? 4 : 1), sizeof(double));
        lr->qpCuTreeOffset = (double*)calloc((size_t)lr->cuCount * (lr->qg8 ? 4 : 1), sizeof(double));
    }
    lr->intraCost = (int32_t*)malloc(sizeof(int32_t) * lr->cuCount);
    lr->intraMode = (uint8_t*)malloc(lr->cuCount);
    for (int i = 0; i < p->bframes + 2; i++)
    {
        for (int j = 0; j < p->bframes + 2; j++)
        {
            lr->rowSatds[i][j] = (int32_t*)malloc(sizeof(int32_t) * lr->heightInCU);
            /* zeroed: --aq-motion reads the list bits of requests nobody has estimated yet (slicetype.cpp:3624 with a
             * B pyramid under fast or no B-adapt, before compCostBref); x265 then sees whatever Lowres::create's
             * malloc returned, which for a frame in first use is zero pages: "no list used" */
            lr->lowresCosts[i][j] = (uint16_t*)calloc(lr->cuCount, sizeof(uint16_t));
        }
        for (int l = 0; l < 2; l++)
        {
            lr->lowresMvs[l][i] = (orc_mv*)malloc(sizeof(orc_mv) * lr->cuCount);
            lr->lowresMvCosts[l][i] = (int32_t*)malloc(sizeof(int32_t) * lr->cuCount);
            if (p->bEnableHME)
            {
                int aw = ((lr->width / 2) + ORC_CU - 1) >> 3, ah = ((lr->lines / 2) + ORC_CU - 1) >> 3;
                int n4 = imax(aw, lr->widthInCU4) * imax(ah, lr->heightInCU4);
                /* Zeroed, unlike the reference's CHECKED_MALLOC (source/common/lowres.cpp:214-217): when the
                 * source height (or width) is 2 or 3 above a multiple of 32, the half-res grid's last block
                 * row maps to a quarter-res row (cuY / 2) that the quarter-res search never visits
                 * (m_4x4Height rows, slicetype.cpp:1063-1064), and estimateCUCost reads its cost anyway
                 * (:4450).  The reference reads uninitialised heap there; zero means "no such candidate". */
                lr->lowerResMvs[l][i] = (orc_mv*)calloc(n4, sizeof(orc_mv));
                lr->lowerResMvCosts[l][i] = (int32_t*)calloc(n4, sizeof(int32_t));
            }
        }
    }
    return lr;
}

void orc_lowres_destroy(orc_lowres* lr)
{
    if (!lr)
        return;
    free(lr->buffer[0]);
    free(lr->lowerResBuffer[0]);
    free(lr->invQscaleFactor);
    free(lr->propagateCost);
    for (int k = 0; k < lr->numAqLayers; k++)
    {
        free(lr->aqLayer[k].dQpOffset);
        free(lr->aqLayer[k].dCuTreeOffset);
    }
    free(lr->cuTreeOffset8x8);
    free(lr->qpAqOffset);
    free(lr->qpCuTreeOffset);
    free(lr->qpAqMotionOffset);
    free(lr->hist);
    free(lr->intraCost);
    free(lr->intraMode);
    free(lr->recalcRowSatds);
    for (int i = 0; i < ORC_BFRAME_MAX + 2; i++)
    {
        for (int j = 0; j < ORC_BFRAME_MAX + 2; j++)
        {
            free(lr->rowSatds[i][j]);
            free(lr->lowresCosts[i][j]);
        }
        for (int l = 0; l < 2; l++)
        {
            free(lr->lowresMvs[l][i]);
            free(lr->lowresMvCosts[l][i]);
            free(lr->lowerResMvs[l][i]);
            free(lr->lowerResMvCosts[l][i]);
        }
    }
    free(lr);
}

/* Lay a tight w x h plane out like PicYuv: CTU-rounded stride with margins,
 * right/bottom replication out to a multiple of 16 plus one extra row/column
 * (source/common/picyuv.cpp:88-103, 291-300, 572-583). */
pixel* orc_picyuv_pad(const orc_param* p, const pixel* tight, pixel** origin, intptr_t* strideOut)
{
    int w = p->sourceWidth, h = p->sourceHeight, ctu = p->maxCUSize;
    int marginX = ctu + 32, marginY = ctu + 16;
    intptr_t stride = ((w + ctu - 1) / ctu) * ctu + 2 * marginX;
    int maxH = ((h + ctu - 1) / ctu) * ctu;
    pixel* buf = (pixel*)calloc((size_t)stride * (maxH + 2 * marginY), sizeof(pixel));
    pixel* Y = buf + marginY * stride + marginX;
    int padx = (w & 15) ? 16 - (w & 15) : 0;
    int pady = (h & 15) ? 16 - (h & 15) : 0;
    padx++;
    pady++;
    for (int r = 0; r < h; r++)
    {
        memcpy(Y + r * stride, tight + (size_t)r * w, w * sizeof(pixel));
        for (int x = 0; x < padx; x++)
            Y[r * stride + w + x] = Y[r * stride + w - 1];
    }
    for (int i = 1; i <= pady; i++)
        memcpy(Y + (h - 1 + i) * stride, Y + (h - 1) * stride, (w + padx) * sizeof(pixel));
    *origin = Y;
    *strideOut = stride;
    return buf;
}

/* source/common/lowres.cpp:344-402 */
void orc_lowres_init(orc_lowres* lr, const orc_param* p, const pixel* luma, intptr_t stride, int poc)
{
    lr->frameNum = poc;
    lr->bLastMiniGopBFrame = 0;
    lr->bKeyframe = 0;
    lr->leadingBframes = 0;
    lr->gopOffset = lr->gopId = lr->tempLayer = 0;
    lr->indB = 0;
    for (int i = 0; i <= ORC_LOOKAHEAD_MAX; i++)
        lr->plannedType[i] = ORC_TYPE_AUTO;
    memset(lr->costEst, -1, sizeof(lr->costEst));
    memset(lr->weightedCostDelta, 0, sizeof(lr->weightedCostDelta));
    if (lr->invQscaleFactor)
        memset(lr->costEstAq, -1, sizeof(lr->costEstAq));
    for (int y = 0; y < lr->bframes + 2; y++)
        for (int x = 0; x < lr->bframes + 2; x++)
            lr->rowSatds[y][x][0] = -1;
    for (int i = 0; i < lr->bframes + 2; i++)
    {
        lr->lowresMvs[0][i][0].x = 0x7FFF;
        lr->lowresMvs[1][i][0].x = 0x7FFF;
        lr->intraMbs[i] = 0;
    }

    intptr_t ls = lr->rp.stride;
    void (*core)(const pixel*, pixel*, pixel*, pixel*, pixel*, intptr_t, intptr_t, int, int) =
        p->bEnableTemporalFilter ? orc_frame_init_lowres_core_mcstf : orc_frame_init_lowres_core;
    core(luma, lr->rp.plane[0], lr->rp.plane[1], lr->rp.plane[2], lr->rp.plane[3], stride, ls, lr->width, lr->lines);
    for (int i = 0; i < 4; i++)
        orc_extend_pic_border(lr->rp.plane[i], ls, lr->width, lr->lines, lr->marginX, lr->marginY);

    if (p->bEnableHME || p->bEnableTemporalFilter)
    {
        core(lr->rp.plane[0], lr->rp.lowerPlane[0], lr->rp.lowerPlane[1], lr->rp.lowerPlane[2], lr->rp.lowerPlane[3],
             ls, ls / 2, lr->width / 2, lr->lines / 2);
        for (int i = 0; i < 4; i++)
            orc_extend_pic_border(lr->rp.lowerPlane[i], ls / 2, lr->width / 2, lr->lines / 2,
                                  lr->marginX / 2, lr->marginY / 2);
    }
}

/* ---------------------------------------------------------------- thread-local state */

orc_tld* orc_tld_create(const orc_param* p, const uint16_t* centredCost)
{
    orc_tld* t = (orc_tld*)calloc(1, sizeof(orc_tld));
    orc_me_init(&t->me, centredCost);
    t->widthInCU = ((p->sourceWidth / 2) + ORC_CU - 1) >> 3;
    t->heightInCU = ((p->sourceHeight / 2) + ORC_CU - 1) >> 3;
    /* m_8x8Blocks: interior blocks only (source/encoder/slicetype.cpp:1065-1066) */
    int n = t->widthInCU * t->heightInCU;
    t->ncu = (t->widthInCU > 2 && t->heightInCU > 2) ? n + 4 - 2 * (t->widthInCU + t->heightInCU) : n;
    return t;
}

void orc_tld_destroy(orc_tld* t)
{
    if (t)
    {
        free(t->wbuffer[0]);
        free(t);
    }
}

/* ---------------------------------------------------------------- intra */

/* source/encoder/slicetype.cpp:706-815 */
void orc_intra_estimate(orc_tld* t, orc_lowres* fenc, const orc_param* p)
{
    pixel pred[64], cur[64], nbr[2][33];
    const int lambda = (int)orc_lookahead_lambda();
    const int intraPenalty = 5 * lambda, lowresPenalty = 4;
    const int W = t->widthInCU, H = t->heightInCU;
    const intptr_t ls = fenc->rp.stride;
    int costEst = 0, costEstAq = 0;
    (void)p;
    const int phaseWas = orc_op_phase;
    orc_op_phase = 0;
    orc_op_counters[0][2] += 12LL * W * H;      /* DC, planar, six coarse angles, two passes of two */

    for (int cuY = 0; cuY < H; cuY++)
    {
        fenc->rowSatds[0][0][cuY] = 0;
        for (int cuX = 0; cuX < W; cuX++)
        {
            const int cuXY = cuX + cuY * W;
            const pixel* pix = fenc->rp.plane[0] + 8 * cuX + 8 * cuY * ls;
            for (int y = 0; y < 8; y++)
                memcpy(cur + 8 * y, pix + y * ls, 8 * sizeof(pixel));

            const pixel* tl = pix - ls - 1;
            memcpy(nbr[0], tl, 17 * sizeof(pixel));
            for (int i = 1; i <= 16; i++)
                nbr[0][16 + i] = tl[i * ls];
            orc_intra_filter(8, nbr[0], nbr[1]);

            int cost, icost = ORC_COST_MAX;
            uint32_t ilowmode = 0;

            orc_intra_pred(8, 1, pred, 8, nbr[0], 1);
            cost = orc_satd(8, 8, cur, 8, pred, 8);
            if (cost < icost) { icost = cost; ilowmode = 1; }

            orc_intra_pred(8, 0, pred, 8, nbr[1], 0);
            cost = orc_satd(8, 8, cur, 8, pred, 8);
            if (cost < icost) { icost = cost; ilowmode = 0; }

            int acost = ORC_COST_MAX;
            uint32_t alowmode = 4;
            for (uint32_t mode = 5; mode < 35; mode += 5)
            {
                orc_intra_pred(8, mode, pred, 8, nbr[orc_intra_filter_flag(mode, 8)], 1);
                cost = orc_satd(8, 8, cur, 8, pred, 8);
                if (cost < acost) { acost = cost; alowmode = mode; }
            }
            for (uint32_t dist = 2; dist >= 1; dist--)
            {
                uint32_t two[2] = { alowmode - dist, alowmode + dist };
                for (int k = 0; k < 2; k++)
                {
                    uint32_t mode = two[k];
                    orc_intra_pred(8, mode, pred, 8, nbr[orc_intra_filter_flag(mode, 8)], 1);
                    cost = orc_satd(8, 8, cur, 8, pred, 8);
                    if (cost < acost) { acost = cost; alowmode = mode; }
                }
            }
            if (acost < icost) { icost = acost; ilowmode = alowmode; }

            icost += intraPenalty + lowresPenalty;
            fenc->lowresCosts[0][0][cuXY] = (uint16_t)imin(icost, ORC_LOWRES_COST_MASK);
            fenc->intraCost[cuXY] = icost;
            fenc->intraMode[cuXY] = (uint8_t)ilowmode;

            const int scored = (cuX > 0 && cuX < W - 1 && cuY > 0 && cuY < H - 1) || W <= 2 || H <= 2;
            int icostAq = (scored && fenc->invQscaleFactor)
                              ? ((icost * fenc->invQscaleFactor[cuXY] + 128) >> 8) : icost;
            if (scored)
            {
                costEst += icost;
                costEstAq += icostAq;
            }
            fenc->rowSatds[0][0][cuY] += icostAq;
        }
    }
    fenc->costEst[0][0] = costEst;
    fenc->costEstAq[0][0] = costEstAq;
    orc_op_phase = phaseWas;
}

/* ---------------------------------------------------------------- AQ block energies */

/* One plane's share of acEnergyCu: pixel_var of a size x size block whose samples beyond the
 * picture are the replicated edge (PicYuv::copyFromPicture pads the source that way,
 * source/common/picyuv.cpp:340-400), then acEnergyVar (source/encoder/slicetype.cpp:51-59). */
static uint32_t ac_energy_plane(const pixel* plane, intptr_t stride, int w, int h, int x0, int y0, int size, int shift,
                                uint64_t* wpSum, uint64_t* wpSsd)
{
    pixel blk[16 * 16];
    for (int y = 0; y < size; y++)
        for (int x = 0; x < size; x++)
            blk[y * size + x] = plane[(intptr_t)imin(y0 + y, h - 1) * stride + imin(x0 + x, w - 1)];
    uint64_t sum_ssd = orc_pixel_var(size, blk, size);
    uint32_t sum = (uint32_t)sum_ssd, ssd = (uint32_t)(sum_ssd >> 32);
    *wpSum += sum;
    *wpSsd += ssd;
    return ssd - (uint32_t)(((uint64_t)sum * sum) >> shift);
}

/* acEnergyCu over the frame, 4:2:0, 16x16 blocks: source/encoder/slicetype.cpp:266-285 and the
 * block loop of calcAdaptiveQuantFrame (:520-527, :560-566).  wp[0..2] = wp_sum, wp[3..5] = wp_ssd. */
void orc_aq_energy(const pixel* y, intptr_t sy, const pixel* u, const pixel* v, intptr_t sc, int w, int h,
                   uint32_t* energy, uint64_t* wp)
{
    memset(wp, 0, 6 * sizeof(uint64_t));
    int k = 0;
    for (int by = 0; by < h; by += 16)
        for (int bx = 0; bx < w; bx += 16, k++)
        {
            uint32_t e = ac_energy_plane(y, sy, w, h, bx, by, 16, 8, &wp[0], &wp[3]);
            e += ac_energy_plane(u, sc, w >> 1, h >> 1, bx >> 1, by >> 1, 8, 6, &wp[1], &wp[4]);
            e += ac_energy_plane(v, sc, w >> 1, h >> 1, bx >> 1, by >> 1, 8, 6, &wp[2], &wp[5]);
            energy[k] = e;
        }
}

/* ---------------------------------------------------------------- AQ edge mode */

/* edgeFilter (source/encoder/slicetype.cpp:191-225) on a tight w x h luma plane: gaussianBlur5x5 (:164-189), then
 * computeEdge with bcalcTheta (:100-162).  edgePic starts as a copy of the source and thetaPic as zeros (:199-215),
 * so the one-pixel frame the Sobel pass leaves out keeps source pixels in edgePic and 0 in thetaPic; the blurred
 * picture keeps source pixels in its two-pixel frame.  The reference's pictures are followed by zeroed padding
 * (:199-201); here everything beyond w x h is simply absent and readers treat it as 0. */
void orc_edge_filter(const pixel* src, intptr_t stride, int w, int h, pixel* edgePic, pixel* thetaPic)
{
    pixel* blur = (pixel*)malloc((size_t)w * h * sizeof(pixel));
    for (int y = 0; y < h; y++)
    {
        memcpy(edgePic + (size_t)y * w, src + y * stride, w * sizeof(pixel));
        memcpy(blur + (size_t)y * w, src + y * stride, w * sizeof(pixel));
        memset(thetaPic + (size_t)y * w, 0, w * sizeof(pixel));
    }
    static const int g[5][5] = { { 2, 4, 5, 4, 2 }, { 4, 9, 12, 9, 4 }, { 5, 12, 15, 12, 5 }, { 4, 9, 12, 9, 4 }, { 2, 4, 5, 4, 2 } };
    for (int r = 2; r < h - 2; r++)
        for (int c = 2; c < w - 2; c++)
        {
            int acc = 0;
            for (int j = 0; j < 5; j++)
                for (int i = 0; i < 5; i++)
                    acc += g[j][i] * src[(r + j - 2) * stride + c + i - 2];
            blur[(size_t)r * w + c] = (pixel)(acc / 159);
        }
#if ORC_DEPTH > 8
    const float edgeThreshold = 1023.0f;            /* EDGE_THRESHOLD, source/encoder/slicetype.h:69-73 */
#else
    const float edgeThreshold = 255.0f;
#endif
    const pixel whitePixel = (pixel)edgeThreshold;
    for (int r = 1; r < h - 1; r++)
        for (int c = 1; c < w - 1; c++)
        {
            const pixel* p = blur + (size_t)r * w + c;
            const float gradientH = (float)(-3 * p[-w - 1] + 3 * p[-w + 1] - 10 * p[-1] + 10 * p[1] - 3 * p[w - 1] + 3 * p[w + 1]);
            const float gradientV = (float)(-3 * p[-w - 1] - 10 * p[-w] - 3 * p[-w + 1] + 3 * p[w - 1] + 10 * p[w] + 3 * p[w + 1]);
            const float gradientMagnitude = sqrtf(gradientH * gradientH + gradientV * gradientV);
            /* C++ picks the float overload of atan2 for two floats */
            const float radians = atan2f(gradientV, gradientH);
            float theta = (float)((radians * 180) / 3.14159265);       /* PI, slicetype.h:74 */
            if (theta < 0)
                theta = 180 + theta;
            thetaPic[(size_t)r * w + c] = (pixel)theta;
            edgePic[(size_t)r * w + c] = (pixel)(gradientMagnitude >= edgeThreshold ? whitePixel : 0);
        }
    free(blur);
}

/* LookaheadTLD::edgeDensityCu + findAvgAngle (source/encoder/slicetype.cpp:227-263): variance of the edge picture's
 * block (acEnergyVar, adding the block's sums to plane 0 of wp_sum / wp_ssd as it does) and the mean angle */
static uint32_t edge_density_cu(const pixel* edgePic, const pixel* thetaPic, int w, int h, int bx, int by, int size,
                                uint32_t* avgAngle, uint64_t* wpSum, uint64_t* wpSsd)
{
    pixel blk[16 * 16];
    int angle = 0;
    for (int y = 0; y < size; y++)
        for (int x = 0; x < size; x++)
        {
            const int in = by + y < h && bx + x < w;
            blk[y * size + x] = in ? edgePic[(size_t)(by + y) * w + bx + x] : 0;
            angle += in ? thetaPic[(size_t)(by + y) * w + bx + x] : 0;
        }
    *avgAngle = angle / (size * size);
    uint64_t sum_ssd = orc_pixel_var(size, blk, size);
    uint32_t sum = (uint32_t)sum_ssd, ssd = (uint32_t)(sum_ssd >> 32);
    *wpSum += sum;
    *wpSsd += ssd;
    return ssd - (uint32_t)(((uint64_t)sum * sum) >> (size == 8 ? 6 : 8));
}

/* x265_exp2fix8, source/common/common.cpp:96-103; the fraction table is
 * x265_exp2_lut (source/common/constants.cpp:552-558) = round(256 * (2^(i/64) - 1)) */
int orc_exp2fix8(double x)
{
    static int lut[64], ready;
    if (!ready)
    {
        for (int i = 0; i < 64; i++)
            lut[i] = (int)lround(256.0 * (pow(2.0, i / 64.0) - 1.0));
        ready = 1;
    }
    int i = (int)(x * (-64.f / 6.f) + 512.5f);
    if (i < 0)
        return 0;
    if (i > 1023)
        return 0xffff;
    return (lut[i & 63] + 256) << (i >> 6) >> 8;
}

/* LookaheadTLD::calcAdaptiveQuantFrame, source/encoder/slicetype.cpp:454-704, for 4:2:0, without
 * hevcAq (aqMode 4 is the edge mode, :529-606).  qgSize 8 measures 8x8 luma + 4x4 chroma blocks with modeOneConst 11.427f /
 * modeTwoConst 8.f, everything else 16x16 + 8x8 with 14.427f / 11.f (:461-474).  Block by block in the
 * reference's order: acEnergyCu is called from the same places, so wp_sum / wp_ssd accumulate exactly
 * as they do there, including the second round of the bDynamicRefine / bEnableFades tail (:688-703).
 * wp[0..2] = wp_sum, wp[3..5] = wp_ssd.  Arrays hold blockCount entries (:463, :470) and are written
 * through the reference's running index; they may be null when the mode does not write them.
 * w8 x h8 is the lowres 8x8 grid (Lowres::maxBlocksInRow / maxBlocksInCol). */
void orc_calc_adaptive_quant_frame(const pixel* y, intptr_t sy, const pixel* u, const pixel* v, intptr_t sc,
                                   int w, int h, int aqMode, double aqStrength, int weightedPred, int hdr10,
                                   int skipOffsets, const float* quantOffsets, double* qpAqOffset,
                                   double* qpCuTreeOffset, int* invQscaleFactor, uint64_t* wp,
                                   int qgSize, int w8, int h8, int* invQscaleFactor8x8,
                                   int blockVarianceOn, uint32_t* blockVariance, double* frameVarianceOut)
{
    int maxCol = w, maxRow = h;
    int blockCount, loopIncr;
    float modeOneConst, modeTwoConst;
    if (qgSize == 8)
    {
        blockCount = (w8 * 2) * (h8 * 2);
        modeOneConst = 11.427f;
        modeTwoConst = 8.f;
        loopIncr = 8;
    }
    else
    {
        blockCount = w8 * h8;
        modeOneConst = 14.427f;
        modeTwoConst = 11.f;
        loopIncr = 16;
    }
    const int lumaShift = loopIncr == 8 ? 6 : 8, chromaShift = loopIncr == 8 ? 4 : 6, half = loopIncr >> 1;
    memset(wp, 0, 6 * sizeof(uint64_t));
#define ORC_ENERGY(bx, by) \
    (ac_energy_plane(y, sy, w, h, bx, by, loopIncr, lumaShift, &wp[0], &wp[3]) + \
     ac_energy_plane(u, sc, w >> 1, h >> 1, (bx) >> 1, (by) >> 1, half, chromaShift, &wp[1], &wp[4]) + \
     ac_energy_plane(v, sc, w >> 1, h >> 1, (bx) >> 1, (by) >> 1, half, chromaShift, &wp[2], &wp[5]))

    if (!skipOffsets)
    {
        if (aqMode == 0 || aqStrength == 0)
        {
            if (aqMode && aqStrength == 0)
            {
                for (int i = 0; i < blockCount; i++)
                {
                    if (quantOffsets)
                    {
                        qpCuTreeOffset[i] = qpAqOffset[i] = quantOffsets[i];
                        invQscaleFactor[i] = orc_exp2fix8(qpCuTreeOffset[i]);
                    }
                    else
                    {
                        qpCuTreeOffset[i] = qpAqOffset[i] = 0;
                        invQscaleFactor[i] = 256;
                    }
                }
            }
            if (weightedPred)
                for (int by = 0; by < maxRow; by += loopIncr)
                    for (int bx = 0; bx < maxCol; bx += loopIncr)
                        (void)ORC_ENERGY(bx, by);
        }
        else
        {
            int k = 0;
            double avg_adj_pow2 = 0, avg_adj = 0, qp_adj = 0, bias_strength = 0, strength = 0;
            pixel* edgePic = NULL;
            pixel* thetaPic = NULL;
            uint8_t* edgeInclined = NULL;
            if (aqMode == 4)
            {
                edgePic = (pixel*)malloc((size_t)w * h * sizeof(pixel));
                thetaPic = (pixel*)malloc((size_t)w * h * sizeof(pixel));
                edgeInclined = (uint8_t*)calloc(blockCount, 1);
                orc_edge_filter(y, sy, w, h, edgePic, thetaPic);
            }
            if (aqMode == 2 || aqMode == 3 || aqMode == 4)
            {
                double bit_depth_correction = 1.f / (1 << (2 * (ORC_DEPTH - 8)));
                for (int by = 0; by < maxRow; by += loopIncr)
                    for (int bx = 0; bx < maxCol; bx += loopIncr)
                    {
                        uint32_t energy = ORC_ENERGY(bx, by);
                        if (aqMode == 4)
                        {
                            uint32_t avgAngle;
                            uint32_t edgeDensity = edge_density_cu(edgePic, thetaPic, w, h, bx, by, loopIncr, &avgAngle, &wp[0], &wp[3]);
                            if (edgeDensity)
                            {
                                qp_adj = pow(edgeDensity * bit_depth_correction + 1, 0.1);
                                /* EDGE_INCLINATION 45 -+ 15 and 45 + 75 .. 45 + 105 */
                                edgeInclined[k] = (avgAngle >= 30 && avgAngle <= 60) || (avgAngle >= 120 && avgAngle <= 150);
                            }
                            else
                            {
                                qp_adj = pow(energy * bit_depth_correction + 1, 0.1);
                                edgeInclined[k] = 0;
                            }
                        }
                        else
                        qp_adj = pow(energy * bit_depth_correction + 1, 0.1);
                        qpCuTreeOffset[k++] = qp_adj;
                        avg_adj += qp_adj;
                        avg_adj_pow2 += qp_adj * qp_adj;
                    }
                avg_adj /= blockCount;
                avg_adj_pow2 /= blockCount;
                strength = aqStrength * avg_adj;
                avg_adj = avg_adj - 0.5f * (avg_adj_pow2 - modeTwoConst) / avg_adj;
                bias_strength = aqStrength;
            }
            else
                strength = aqStrength * 1.0397f;

            k = 0;
            for (int by = 0; by < maxRow; by += loopIncr)
                for (int bx = 0; bx < maxCol; bx += loopIncr, k++)
                {
                    if (aqMode == 3)
                    {
                        qp_adj = qpCuTreeOffset[k];
                        qp_adj = strength * (qp_adj - avg_adj) + bias_strength * (1.f - modeTwoConst / (qp_adj * qp_adj));
                    }
                    else if (aqMode == 2)
                    {
                        qp_adj = qpCuTreeOffset[k];
                        qp_adj = strength * (qp_adj - avg_adj);
                    }
                    else if (aqMode == 4)
                    {
                        qp_adj = qpCuTreeOffset[k];
                        if (edgeInclined[k] && (qp_adj - avg_adj > 0))
                            qp_adj = ((strength + 0.5 /* AQ_EDGE_BIAS */) * (qp_adj - avg_adj));
                        else
                            qp_adj = strength * (qp_adj - avg_adj);
                    }
                    else
                    {
                        uint32_t energy = ORC_ENERGY(bx, by);
                        qp_adj = strength * (log2((double)(energy > 1 ? energy : 1)) - (modeOneConst + 2 * (ORC_DEPTH - 8)));
                    }
                    if (hdr10)
                    {
                        /* lumaSumCu (:288-301): the low word of pixel_var over the same padded block */
                        uint64_t s0 = 0, s1 = 0;
                        (void)ac_energy_plane(y, sy, w, h, bx, by, loopIncr, lumaShift, &s0, &s1);
                        uint32_t lumaAvg = (uint32_t)s0 / (loopIncr * loopIncr);
                        if (lumaAvg < 301) qp_adj += 3;
                        else if (lumaAvg < 367) qp_adj += 2;
                        else if (lumaAvg < 434) qp_adj += 1;
                        else if (lumaAvg >= 501 && lumaAvg < 567) qp_adj -= 1;
                        else if (lumaAvg >= 567 && lumaAvg < 634) qp_adj -= 2;
                        else if (lumaAvg >= 634 && lumaAvg < 701) qp_adj -= 3;
                        else if (lumaAvg >= 701 && lumaAvg < 767) qp_adj -= 4;
                        else if (lumaAvg >= 767 && lumaAvg < 834) qp_adj -= 5;
                        else if (lumaAvg >= 834) qp_adj -= 6;
                    }
                    if (quantOffsets)
                        qp_adj += quantOffsets[k];
                    qpAqOffset[k] = qp_adj;
                    qpCuTreeOffset[k] = qp_adj;
                    invQscaleFactor[k] = orc_exp2fix8(qp_adj);
                }
            free(edgePic);
            free(thetaPic);
            free(edgeInclined);
        }
        if (qgSize == 8 && invQscaleFactor && invQscaleFactor8x8)
        {
            /* :647-660 */
            const int rowFull = w8 * 2;
            for (int cuY = 0; cuY < h8; cuY++)
                for (int cuX = 0; cuX < w8; cuX++)
                {
                    const int at = cuX * 2 + cuY * w8 * 4;
                    invQscaleFactor8x8[cuX + cuY * w8] = (invQscaleFactor[at] + invQscaleFactor[at + 1] +
                                                          invQscaleFactor[at + rowFull] + invQscaleFactor[at + rowFull + 1]) / 4;
                }
        }
    }
    if (weightedPred)
    {
        if (skipOffsets)
            for (int by = 0; by < maxRow; by += loopIncr)
                for (int bx = 0; bx < maxCol; bx += loopIncr)
                    (void)ORC_ENERGY(bx, by);
        maxCol = ((maxCol + 8) >> 4) << 4;
        maxRow = ((maxRow + 8) >> 4) << 4;
        int width[3] = { maxCol, maxCol >> 1, maxCol >> 1 };
        int height[3] = { maxRow, maxRow >> 1, maxRow >> 1 };
        for (int i = 0; i < 3; i++)
        {
            uint64_t sum = wp[i], ssd = wp[3 + i];
            wp[3 + i] = ssd - (sum * sum + (width[i] * height[i]) / 2) / (width[i] * height[i]);
        }
    }
    if (blockVarianceOn)
    {
        /* :688-703; maxCol / maxRow are the rounded ones when the weighted-prediction branch ran */
        uint64_t blockXY = 0, rowVariance = 0;
        double frameVariance = 0;
        for (int by = 0; by < maxRow; by += loopIncr)
        {
            for (int bx = 0; bx < maxCol; bx += loopIncr)
            {
                blockVariance[blockXY] = ORC_ENERGY(bx, by);
                rowVariance += blockVariance[blockXY];
                blockXY++;
            }
            frameVariance += (rowVariance / maxCol);
        }
        frameVariance /= maxRow;
        *frameVarianceOut = frameVariance;
    }
#undef ORC_ENERGY
}

/* One layer of LookaheadTLD::xPreanalyze (source/encoder/slicetype.cpp:352-428) followed by
 * xPreanalyzeQp for that layer (:315-335): quadrant variances of every part straight from the
 * pixels, activity = 1 + the smallest, the layer mean over numW * numH parts, then the QP offsets.
 * Returns dAvgActivity. */
double orc_hevc_aq_layer(const pixel* src, intptr_t stride, int width, int height, int aqPartWidth, int aqPartHeight,
                         int numW, int numH, double qpAdaptationRange, double* dActivity, double* dQpOffset,
                         double* dCuTreeOffset)
{
    double* pcAQU = dActivity;
    double dSumAct = 0.0;
    for (uint32_t y = 0; y < (uint32_t)height; y += aqPartHeight)
    {
        const uint32_t currAQPartHeight = imin(aqPartHeight, height - y);
        for (uint32_t x = 0; x < (uint32_t)width; x += aqPartWidth, pcAQU++)
        {
            const uint32_t currAQPartWidth = imin(aqPartWidth, width - x);
            const pixel* pBlkY = &src[x];
            uint64_t sum[4] = { 0, 0, 0, 0 };
            uint64_t sumSq[4] = { 0, 0, 0, 0 };
            uint32_t by = 0;
            for (; by < currAQPartHeight; by++)
            {
                const int lower = by >= (currAQPartHeight >> 1) ? 2 : 0;
                for (uint32_t bx = 0; bx < currAQPartWidth; bx++)
                {
                    const int q = lower + (bx >= (currAQPartWidth >> 1) ? 1 : 0);
                    sum[q] += pBlkY[bx];
                    sumSq[q] += pBlkY[bx] * pBlkY[bx];
                }
                pBlkY += stride;
            }
            const uint32_t numPixInAQPart = (currAQPartWidth >> 1) * (currAQPartHeight >> 1);
            double dMinVar = 1.7e+308;
            if (numPixInAQPart != 0)
            {
                for (int i = 0; i < 4; i++)
                {
                    const double dAverage = (double)(sum[i]) / numPixInAQPart;
                    const double dVariance = (double)(sumSq[i]) / numPixInAQPart - dAverage * dAverage;
                    dMinVar = dMinVar < dVariance ? dMinVar : dVariance;
                }
            }
            else
                dMinVar = 0.0;
            double dAct = 1.0 + dMinVar;
            *pcAQU = dAct;
            dSumAct += dAct;
        }
        src += stride * currAQPartHeight;
    }
    const double dAvgAct = dSumAct / (numW * numH);

    pcAQU = dActivity;
    for (uint32_t y = 0; y < (uint32_t)height; y += aqPartHeight)
        for (uint32_t x = 0; x < (uint32_t)width; x += aqPartWidth, pcAQU++, dQpOffset++, dCuTreeOffset++)
        {
            double dMaxQScale = pow(2.0, qpAdaptationRange / 6.0);
            double dCUAct = *pcAQU;
            double dNormAct = (dMaxQScale * dCUAct + dAvgAct) / (dCUAct + dMaxQScale * dAvgAct);
            double dQp = (log2(dNormAct) / log2(2.0)) * 6.0;
            *dQpOffset = dQp;
            *dCuTreeOffset = dQp;
        }
    return dAvgAct;
}

/* ---------------------------------------------------------------- weighted prediction */

/* source/encoder/slicetype.cpp:817-850 */
uint32_t orc_weight_cost_luma(orc_tld* t, orc_lowres* fenc, orc_lowres* ref,
                              int wtPresent, int scale, int denom, int offsetIn)
{
    const pixel* src = ref->rp.plane[0];
    intptr_t stride = fenc->rp.stride;
    if (wtPresent)
    {
        int offset = offsetIn * (1 << (ORC_DEPTH - 8));
        int round = denom ? 1 << (denom - 1) : 0;
        int correction = 14 - ORC_DEPTH;
        orc_weight_pp(ref->buffer[0], t->wbuffer[0], stride, (int)stride, t->paddedLines,
                      scale, round << correction, denom + correction, offset);
        src = t->wbuffer[0] + fenc->padoffset;
    }
    uint32_t cost = 0;
    int mb = 0;
    const int phaseWas = orc_op_phase;
    orc_op_phase = 3;
    for (int y = 0; y < fenc->lines; y += 8)
        for (int x = 0; x < fenc->width; x += 8, mb++)
        {
            intptr_t off = y * stride + x;
            int satd = orc_satd(8, 8, src + off, stride, fenc->rp.plane[0] + off, stride);
            cost += imin(satd, fenc->intraCost[mb]);
        }
    orc_op_phase = phaseWas;
    return cost;
}

/* WeightParam::setFromWeightAndOffset (source/common/slice.h): reduce the
 * weight to at most 7 bits by dropping denominator bits */
static void wp_set(int* inputWeight, int* log2Denom, int* inputOffset, int w, int o, int denom, int bNormalize)
{
    *inputOffset = o;
    *log2Denom = denom;
    *inputWeight = w;
    while (bNormalize && *log2Denom > 0 && (*inputWeight > 127))
    {
        (*log2Denom)--;
        *inputWeight >>= 1;
    }
    *inputWeight = imin(*inputWeight, 127);
}

/* source/encoder/slicetype.cpp:870-971; float scalar logic as in the reference */
void orc_weights_analyse(orc_tld* t, orc_lowres* fenc, orc_lowres* ref)
{
    static const float epsilon = 1.f / 128.f;
    int deltaIndex = fenc->frameNum - ref->frameNum;

    if (!t->wbuffer[0])
    {
        t->paddedLines = (int)(fenc->planesize / fenc->rp.stride);
        t->wbuffer[0] = (pixel*)malloc(4 * fenc->planesize * sizeof(pixel));
        for (int i = 1; i < 4; i++)
            t->wbuffer[i] = t->wbuffer[0] + i * fenc->planesize;
    }
    orc_refplanes* wr = &fenc->weightedRef[deltaIndex];
    memset(wr, 0, sizeof(*wr));
    for (int i = 0; i < 4; i++)
        wr->plane[i] = t->wbuffer[i] + fenc->padoffset;
    wr->stride = fenc->rp.stride;
    wr->isLowres = 1;
    wr->isHMELowres = ref->rp.isHMELowres;
    fenc->weightedRefIsWeighted[deltaIndex] = 0;

    float guessScale, fencMean, refMean;
    if (fenc->wp_ssd[0] && ref->wp_ssd[0])
        guessScale = sqrtf((float)fenc->wp_ssd[0] / ref->wp_ssd[0]);
    else
        guessScale = 1.0f;
    fencMean = (float)fenc->wp_sum[0] / (fenc->lines * fenc->width) / (1 << (ORC_DEPTH - 8));
    refMean = (float)ref->wp_sum[0] / (fenc->lines * fenc->width) / (1 << (ORC_DEPTH - 8));

    if (fabsf(refMean - fencMean) < 0.5f && fabsf(1.f - guessScale) < epsilon)
        return;

    int minoff = 0, minscale, mindenom;
    unsigned int minscore = 0, origscore = 1;
    int found = 0;
    int wW, wD, wO;

    wp_set(&wW, &wD, &wO, (int)(guessScale * 128 + 0.5f), 0, 7, 1);
    mindenom = wD;
    minscale = wW;

    origscore = minscore = orc_weight_cost_luma(t, fenc, ref, 0, 0, 0, 0);
    if (!minscore)
        return;

    unsigned int s = 0;
    int curScale = minscale;
    int curOffset = (int)(fencMean - refMean * curScale / (1 << mindenom) + 0.5f);
    if (curOffset < -128 || curOffset > 127)
    {
        curOffset = imax(-128, imin(127, curOffset));
        curScale = (int)((1 << mindenom) * (fencMean - curOffset) / refMean + 0.5f);
        curScale = imax(0, imin(127, curScale));
    }
    s = orc_weight_cost_luma(t, fenc, ref, 1, curScale, mindenom, curOffset);
    if (s < minscore) { minscore = s; minscale = curScale; minoff = curOffset; found = 1; }

    if (mindenom > 0 && minscale && !(minscale & 1))
    {
        int idx = __builtin_ctz((unsigned)minscale);
        int shift = imin(idx, mindenom);
        mindenom -= shift;
        minscale >>= shift;
    }

    if (!found || (minscale == 1 << mindenom && minoff == 0) || (float)minscore / origscore > 0.998f)
        return;

    /* integer division, as in the reference (source/encoder/slicetype.cpp:955) */
    fenc->weightedCostDelta[deltaIndex] = minscore / origscore;

    int offset = minoff * (1 << (ORC_DEPTH - 8));
    int round = mindenom ? 1 << (mindenom - 1) : 0;
    int correction = 14 - ORC_DEPTH;
    intptr_t stride = ref->rp.stride;
    for (int i = 0; i < 4; i++)
        orc_weight_pp(ref->buffer[i], t->wbuffer[i], stride, (int)stride, t->paddedLines,
                      minscale, round << correction, mindenom + correction, offset);
    fenc->weightedRefIsWeighted[deltaIndex] = 1;
}

/* ---------------------------------------------------------------- inter / bidir */

/* source/encoder/slicetype.cpp:4385-4557 */
static void estimate_cu(orc_tld* t, const orc_param* p, orc_lowres** frames, int cuX, int cuY,
                        int p0, int p1, int b, const int bDoSearch[2], int lastRow, int hme)
{
    orc_lowres* fref0 = frames[p0];
    orc_lowres* fref1 = frames[p1];
    orc_lowres* fenc = frames[b];
    const orc_refplanes* wfref0 =
        (fenc->weightedRefIsWeighted[b - p0] && !hme) ? &fenc->weightedRef[b - p0] : &fref0->rp;

    const int W = hme ? fenc->widthInCU4 : fenc->widthInCU;
    const int H = hme ? fenc->heightInCU4 : fenc->heightInCU;
    const int bBidir = b < p1;
    const int cuXY = cuX + cuY * W;
    const int cuXY4 = (cuX / 2) + (cuY / 2) * W / 2;
    const intptr_t ls = fenc->rp.stride;
    const intptr_t pelOffset = 8 * cuX + 8 * cuY * (hme ? ls / 2 : ls);

    if (bBidir || bDoSearch[0] || bDoSearch[1])
        orc_me_set_source(&t->me, hme ? fenc->rp.lowerPlane[0] : fenc->rp.plane[0], hme ? ls / 2 : ls,
                          pelOffset, 8, 8, p->lookaheadSearchMethod, p->hmeSearchMethod[0],
                          p->hmeSearchMethod[1], 1);

    const int lowresPenalty = 4;
    const int listDist[2] = { b - p0, p1 - b };
    int bcost = ORC_COST_MAX;
    int listused = 0;
    orc_op_phase = 2;       /* the decision half; the searches below name themselves */

    orc_mv mvmin = { -cuX * 8 - 8, -cuY * 8 - 8 };
    orc_mv mvmax = { (W - cuX - 1) * 8 + 8, (H - cuY - 1) * 8 + 8 };

    for (int i = 0; i < 1 + bBidir; i++)
    {
        int32_t* fencCost = hme ? &fenc->lowerResMvCosts[i][listDist[i]][cuXY]
                                : &fenc->lowresMvCosts[i][listDist[i]][cuXY];
        int skipCost = INT_MAX;

        if (!bDoSearch[i])
        {
            if (*fencCost < bcost) { bcost = *fencCost; listused = i + 1; }
            continue;
        }

        orc_op_phase = 1;
        int numc = 0;
        orc_mv mvc[5], mvp = { 0, 0 };
        orc_mv* fencMV = hme ? &fenc->lowerResMvs[i][listDist[i]][cuXY] : &fenc->lowresMvs[i][listDist[i]][cuXY];
        const orc_refplanes* fref = i ? &fref1->rp : wfref0;

        /* predictors from blocks already visited (right, below, below-left, below-right) */
        if (cuX < W - 1)
            mvc[numc++] = fencMV[1];
        if (!lastRow)
        {
            mvc[numc++] = fencMV[W];
            if (cuX > 0)
                mvc[numc++] = fencMV[W - 1];
            if (cuX < W - 1)
                mvc[numc++] = fencMV[W + 1];
        }
        if (fenc->lowerResMvs[0][0] && !hme && fenc->lowerResMvCosts[i][listDist[i]][cuXY4] > 0)
        {
            orc_mv m = fenc->lowerResMvs[i][listDist[i]][cuXY4];
            m.x *= 2;
            m.y *= 2;
            mvc[numc++] = m;
        }

        if (numc)
        {
            pixel subpelbuf[64];
            int mvpcost = ORC_COST_MAX;
            for (int idx = 0; idx < numc; idx++)
            {
                intptr_t stride = 8;
                const pixel* src = orc_lowres_mc(fref, pelOffset, mvc[idx], subpelbuf, &stride, hme);
                int cost = orc_me_buf_satd(&t->me, src, stride);
                if (cost < mvpcost) { mvpcost = cost; mvp = mvc[idx]; }
                /* evaluated after every candidate: tracks the cost of the latest
                 * candidate while the running best is still the zero vector */
                if (!(mvp.x | mvp.y) && bBidir)
                    skipCost = cost;
            }
        }

        int searchRange = p->bEnableHME ? (hme ? p->hmeRange[0] : p->hmeRange[1]) : 16;
        *fencCost = orc_motion_estimate(&t->me, fref, mvmin, mvmax, mvp, 0, NULL, searchRange, fencMV,
                                        (uint32_t)p->maxSlices, hme);
        if (skipCost < 64 && skipCost < *fencCost && bBidir)
        {
            *fencCost = skipCost;
            fencMV->x = 0;
            fencMV->y = 0;
        }
        if (*fencCost < bcost) { bcost = *fencCost; listused = i + 1; }
        orc_op_phase = 2;
    }
    if (hme)
        return;

    if (bBidir)
    {
        pixel buf0[64], buf1[64], avg[64];
        intptr_t s0 = 8, s1 = 8;
        const pixel* src0 = orc_lowres_mc(&fref0->rp, pelOffset, fenc->lowresMvs[0][listDist[0]][cuXY], buf0, &s0, 0);
        const pixel* src1 = orc_lowres_mc(&fref1->rp, pelOffset, fenc->lowresMvs[1][listDist[1]][cuXY], buf1, &s1, 0);
        orc_pixelavg_pp(8, 8, avg, 8, src0, s0, src1, s1);
        int bicost = orc_me_buf_satd(&t->me, avg, 8);
        if (bicost < bcost) { bcost = bicost; listused = 3; }
        orc_pixelavg_pp(8, 8, avg, 8, fref0->rp.plane[0] + pelOffset, fref0->rp.stride,
                        fref1->rp.plane[0] + pelOffset, fref1->rp.stride);
        bicost = orc_me_buf_satd(&t->me, avg, 8);
        if (bicost < bcost) { bcost = bicost; listused = 3; }
        bcost += lowresPenalty;
    }
    else
    {
        bcost += lowresPenalty;
        if (fenc->intraCost[cuXY] < bcost)
        {
            bcost = fenc->intraCost[cuXY];
            listused = 0;
        }
    }

    const int scored = (cuX > 0 && cuX < W - 1 && cuY > 0 && cuY < H - 1) || W <= 2 || H <= 2;
    int bcostAq = (scored && fenc->invQscaleFactor) ? ((bcost * fenc->invQscaleFactor[cuXY] + 128) >> 8) : bcost;
    if (scored)
    {
        fenc->costEst[b - p0][p1 - b] += bcost;
        fenc->costEstAq[b - p0][p1 - b] += bcostAq;
        if (!listused && !bBidir)
            fenc->intraMbs[b - p0]++;
    }
    fenc->rowSatds[b - p0][p1 - b][cuY] += bcostAq;
    fenc->lowresCosts[b - p0][p1 - b][cuXY] =
        (uint16_t)(imin(bcost, ORC_LOWRES_COST_MASK) | (listused << ORC_LOWRES_COST_SHIFT));
}

/* Lookahead::Lookahead, source/encoder/slicetype.cpp:1122-1133 */
int orc_coop_slices(const orc_param* p, int heightInCU, int* numRowsPerSlice)
{
    int rows = heightInCU, slices = 1;
    if (p->lookaheadSlices > 1)
    {
        rows = heightInCU / p->lookaheadSlices;
        rows = imax(rows, 10);
        rows = imin(rows, heightInCU);
        slices = heightInCU / rows;
    }
    if (numRowsPerSlice)
        *numRowsPerSlice = rows;
    return slices;
}

/* test instrumentation: called for every request that is computed rather than found cached */
static orc_request_hook g_requestHook;
static void* g_requestHookUser;
void orc_set_request_hook(orc_request_hook hook, void* user)
{
    g_requestHook = hook;
    g_requestHookUser = user;
}

/* source/encoder/slicetype.cpp:4284-4383.  batchMode: CostEstimateGroup::m_batchMode; outside it a
 * lookahead with cooperative slices splits the block loop (:4312-4343, the slice tasks of
 * processTasks :4248-4276 run here one after the other; their sums are order independent). */
static int64_t estimate_frame_cost(orc_tld* t, const orc_param* p, orc_lowres** frames,
                                   int p0, int p1, int b, int bIntraPenalty, int batchMode)
{
    orc_lowres* fenc = frames[b];
    int64_t score;

    if (fenc->costEst[b - p0][p1 - b] >= 0 && fenc->rowSatds[b - p0][p1 - b][0] != -1)
        score = fenc->costEst[b - p0][p1 - b];
    else
    {
        int bDoSearch[2];
        bDoSearch[0] = fenc->lowresMvs[0][b - p0][0].x == 0x7FFF;
        bDoSearch[1] = p1 > b && fenc->lowresMvs[1][p1 - b][0].x == 0x7FFF;
        if (g_requestHook)
            g_requestHook(g_requestHookUser, frames, p0, p1, b, bDoSearch, batchMode);

        fenc->weightedRefIsWeighted[b - p0] = 0;
        if (p->bEnableWeightedPred && bDoSearch[0])
            orc_weights_analyse(t, frames[b], frames[p0]);

        fenc->costEst[b - p0][p1 - b] = 0;
        fenc->costEstAq[b - p0][p1 - b] = 0;

        int rowsPerSlice;
        const int numCoopSlices = orc_coop_slices(p, fenc->heightInCU, &rowsPerSlice);
        if (!batchMode && numCoopSlices > 1 && (p1 > b || bDoSearch[0] || bDoSearch[1]))
        {
            /* with HME a slice's half-res blocks read quarter-res vectors that another slice's task
             * writes (:4250-4263, :4450-4453): timing dependent in x265, not restated */
            if (p->bEnableHME)
                return -1;
            for (int i = 0; i < numCoopSlices; i++)
            {
                const int firstY = rowsPerSlice * i;
                const int lastY = (i == numCoopSlices - 1) ? fenc->heightInCU - 1 : rowsPerSlice * (i + 1) - 1;
                int lastRow = 1;
                for (int cuY = lastY; cuY >= firstY; cuY--)
                {
                    fenc->rowSatds[b - p0][p1 - b][cuY] = 0;
                    for (int cuX = fenc->widthInCU - 1; cuX >= 0; cuX--)
                        estimate_cu(t, p, frames, cuX, cuY, p0, p1, b, bDoSearch, lastRow, 0);
                    lastRow = 0;
                }
            }
        }
        else
        {
            if (p->bEnableHME)
            {
                int lastRow = 1;
                for (int cuY = fenc->heightInCU4 - 1; cuY >= 0; cuY--)
                {
                    for (int cuX = fenc->widthInCU4 - 1; cuX >= 0; cuX--)
                        estimate_cu(t, p, frames, cuX, cuY, p0, p1, b, bDoSearch, lastRow, 1);
                    lastRow = 0;
                }
            }
            int lastRow = 1;
            for (int cuY = fenc->heightInCU - 1; cuY >= 0; cuY--)
            {
                fenc->rowSatds[b - p0][p1 - b][cuY] = 0;
                for (int cuX = fenc->widthInCU - 1; cuX >= 0; cuX--)
                    estimate_cu(t, p, frames, cuX, cuY, p0, p1, b, bDoSearch, lastRow, 0);
                lastRow = 0;
            }
        }

        score = fenc->costEst[b - p0][p1 - b];
        if (b != p1)
            score = score * 100 / (130 + p->bFrameBias);
        fenc->costEst[b - p0][p1 - b] = score;
    }

    if (bIntraPenalty)
        score += score * fenc->intraMbs[b - p0] / (t->ncu * 8);
    return score;
}

/* a request of a batch (CostEstimateGroup::add / finishBatch), or any request without cooperative slices */
int64_t orc_estimate_frame_cost(orc_tld* t, const orc_param* p, orc_lowres** frames,
                                int p0, int p1, int b, int bIntraPenalty)
{
    return estimate_frame_cost(t, p, frames, p0, p1, b, bIntraPenalty, 1);
}

/* CostEstimateGroup::singleCost outside batch mode: split into p->lookaheadSlices cooperative slices */
int64_t orc_estimate_frame_cost_single(orc_tld* t, const orc_param* p, orc_lowres** frames,
                                       int p0, int p1, int b, int bIntraPenalty)
{
    return estimate_frame_cost(t, p, frames, p0, p1, b, bIntraPenalty, 0);
}

/* ---------------------------------------------------------------- ctypes accessors */

void orc_lowres_set_slice_type(orc_lowres* lr, int sliceType) { lr->sliceType = sliceType; }

void* orc_lowres_ptr(orc_lowres* lr, const char* f, int i, int j)
{
    if (!strcmp(f, "plane")) return lr->rp.plane[i];
    if (!strcmp(f, "buffer")) return lr->buffer[i];
    if (!strcmp(f, "lowerPlane")) return lr->rp.lowerPlane[i];
    if (!strcmp(f, "lowerBuffer")) return lr->lowerResBuffer[i];
    if (!strcmp(f, "intraCost")) return lr->intraCost;
    if (!strcmp(f, "intraMode")) return lr->intraMode;
    if (!strcmp(f, "invQscaleFactor")) return lr->invQscaleFactor;
    if (!strcmp(f, "propagateCost")) return lr->propagateCost;
    if (!strcmp(f, "qpAqOffset")) return lr->qpAqOffset;
    if (!strcmp(f, "qpCuTreeOffset")) return lr->qpCuTreeOffset;
    if (!strcmp(f, "rowSatds")) return lr->rowSatds[i][j];
    if (!strcmp(f, "lowresCosts")) return lr->lowresCosts[i][j];
    if (!strcmp(f, "lowresMvs")) return lr->lowresMvs[i][j];
    if (!strcmp(f, "lowresMvCosts")) return lr->lowresMvCosts[i][j];
    if (!strcmp(f, "lowerResMvs")) return lr->lowerResMvs[i][j];
    if (!strcmp(f, "lowerResMvCosts")) return lr->lowerResMvCosts[i][j];
    if (!strcmp(f, "wp_sum")) return lr->wp_sum;
    if (!strcmp(f, "wp_ssd")) return lr->wp_ssd;
    if (!strcmp(f, "weightedPlane")) return lr->weightedRef[i].plane[j];
    if (!strcmp(f, "plannedSatd")) return lr->plannedSatd;
    if (!strcmp(f, "plannedType")) return lr->plannedType;
    if (!strcmp(f, "recalcRowSatds")) return lr->recalcRowSatds;
    return NULL;
}

int64_t orc_lowres_i64(orc_lowres* lr, const char* f, int i, int j)
{
    if (!strcmp(f, "costEst")) return lr->costEst[i][j];
    if (!strcmp(f, "costEstAq")) return lr->costEstAq[i][j];
    if (!strcmp(f, "intraMbs")) return lr->intraMbs[i];
    if (!strcmp(f, "width")) return lr->width;
    if (!strcmp(f, "lines")) return lr->lines;
    if (!strcmp(f, "stride")) return lr->rp.stride;
    if (!strcmp(f, "marginX")) return lr->marginX;
    if (!strcmp(f, "marginY")) return lr->marginY;
    if (!strcmp(f, "widthInCU")) return lr->widthInCU;
    if (!strcmp(f, "heightInCU")) return lr->heightInCU;
    if (!strcmp(f, "widthInCU4")) return lr->widthInCU4;
    if (!strcmp(f, "heightInCU4")) return lr->heightInCU4;
    if (!strcmp(f, "planesize")) return (int64_t)lr->planesize;
    if (!strcmp(f, "padoffset")) return (int64_t)lr->padoffset;
    if (!strcmp(f, "isWeighted")) return lr->weightedRefIsWeighted[i];
    if (!strcmp(f, "frameNum")) return lr->frameNum;
    if (!strcmp(f, "sliceType")) return lr->sliceType;
    if (!strcmp(f, "bScenecut")) return lr->bScenecut;
    if (!strcmp(f, "bKeyframe")) return lr->bKeyframe;
    if (!strcmp(f, "bLastMiniGopBFrame")) return lr->bLastMiniGopBFrame;
    if (!strcmp(f, "leadingBframes")) return lr->leadingBframes;
    if (!strcmp(f, "gopOffset")) return lr->gopOffset;
    if (!strcmp(f, "gopId")) return lr->gopId;
    if (!strcmp(f, "tempLayer")) return lr->tempLayer;
    if (!strcmp(f, "indB")) return lr->indB;
    if (!strcmp(f, "qg8")) return lr->qg8;
    if (!strcmp(f, "weightedCostDelta")) return (int64_t)lr->weightedCostDelta[i];
    return -1;
}
