/* TEST ORACLE - not part of the product.  CPU restatement of the motion search
 * of x265's motion-compensated spatio-temporal filter (--mcstf), used only by
 * tests to check the HIP path.
 *
 * Follows MotionEstimatorTLD::motionErrorLumaSSD, ::motionEstimationLuma and
 * ::motionEstimationLumaDoubleRes (source/common/temporalfilter.cpp:38-91,
 * 421-478, 642-897) and the way the lookahead issues them as 16-line row tasks,
 * one batch per pyramid level (source/encoder/slicetype.cpp:2280-2297,
 * 4217-4234).  Motion vectors are in 1/16 pixel. */
#include "orc.h"

#include <limits.h>
#include <string.h>

/* s_interpolationFilter, source/common/temporalfilter.h:36-54 (taps 0 and 7 are never read) */
static const int k_filter[16][8] = {
    { 0, 0,   0, 64,  0,   0, 0, 0 }, { 0, 1,  -3, 64,  4,  -2, 0, 0 }, { 0, 1,  -6, 62,  9,  -3, 1, 0 },
    { 0, 2,  -8, 60, 14,  -5, 1, 0 }, { 0, 2,  -9, 57, 19,  -7, 2, 0 }, { 0, 3, -10, 53, 24,  -8, 2, 0 },
    { 0, 3, -11, 50, 29,  -9, 2, 0 }, { 0, 3, -11, 44, 35, -10, 3, 0 }, { 0, 1,  -7, 38, 38,  -7, 1, 0 },
    { 0, 3, -10, 35, 44, -11, 3, 0 }, { 0, 2,  -9, 29, 50, -11, 3, 0 }, { 0, 2,  -8, 24, 53, -10, 3, 0 },
    { 0, 2,  -7, 19, 57,  -9, 2, 0 }, { 0, 1,  -5, 14, 60,  -8, 2, 0 }, { 0, 1,  -3,  9, 62,  -6, 1, 0 },
    { 0, 0,  -2,  4, 64,  -3, 1, 0 },
};

enum { MV_FACTOR = 16, SEARCH_RANGE = 3, BLOCK_SIZE = 16 };   /* temporalfilter.h:121-123 */

/* motionErrorLumaFrac_c, temporalfilter.cpp:38-91 */
static int motion_error_frac(const pixel* orig, intptr_t origStride, const pixel* buff, intptr_t buffStride,
                             int x, int y, int dx, int dy, int bs, int besterror)
{
    const int* xFilter = k_filter[dx & 0xF];
    const int* yFilter = k_filter[dy & 0xF];
    int temp[64 + 8][64];
    int error = 0;

    for (int y1 = 1; y1 < bs + 7; y1++)
    {
        const int yOffset = y + y1 + (dy >> 4) - 3;
        const pixel* sourceRow = buff + yOffset * buffStride;
        for (int x1 = 0; x1 < bs; x1++)
        {
            const pixel* rowStart = sourceRow + (x + x1 + (dx >> 4) - 3);
            int sum = 0;
            for (int k = 1; k <= 6; k++)
                sum += xFilter[k] * rowStart[k];
            temp[y1][x1] = sum;
        }
    }

    for (int y1 = 0; y1 < bs; y1++)
    {
        const pixel* origRow = orig + (y + y1) * origStride;
        for (int x1 = 0; x1 < bs; x1++)
        {
            int sum = 0;
            for (int k = 1; k <= 6; k++)
                sum += yFilter[k] * temp[y1 + k][x1];
            sum = (sum + (1 << 11)) >> 12;
            sum = sum < 0 ? 0 : (sum > ORC_PIXEL_MAX ? ORC_PIXEL_MAX : sum);
            int diff = sum - origRow[x + x1];
            error += diff * diff;
        }
        if (error > besterror)
            return error;
    }
    return error;
}

/* motionErrorLumaSSD, temporalfilter.cpp:421-478.  The full-pel branch copies the block and calls
 * sse_pp: the whole sum, cast to int. */
int orc_mcstf_motion_error(const pixel* src, int stride, const pixel* buf, int x, int y, int dx, int dy, int bs,
                           int besterror)
{
    if (((dx | dy) & 0xF) == 0)
    {
        dx /= MV_FACTOR;
        dy /= MV_FACTOR;
        const pixel* b = buf + (intptr_t)(y + dy) * stride + (x + dx);
        const pixel* o = src + (intptr_t)y * stride + x;
        uint64_t sse = 0;
        for (int y1 = 0; y1 < bs; y1++)
            for (int x1 = 0; x1 < bs; x1++)
            {
                int d = o[(intptr_t)y1 * stride + x1] - b[(intptr_t)y1 * stride + x1];
                sse += (uint64_t)(d * d);
            }
        return (int)sse;
    }
    return motion_error_frac(src, stride, buf, stride, x, y, dx, dy, bs, besterror);
}

/* the candidates shared by both searches: the nine surrounding vectors of the coarser level and the
 * zero vector (temporalfilter.cpp:676-711, 786-821) */
static void try_previous(const pixel* src, int stride, const pixel* buf, int blockX, int blockY, int blockSize,
                         int origWidth, int origHeight, const orc_mv* previous, uint32_t prevMvStride, int factor,
                         orc_mv* best, int* leastError)
{
    for (int py = -1; py <= 1; py++)
    {
        int testy = blockY / (2 * blockSize) + py;
        for (int px = -1; px <= 1; px++)
        {
            int testx = blockX / (2 * blockSize) + px;
            if (testx >= 0 && testx < origWidth / (2 * blockSize) && testy >= 0 && testy < origHeight / (2 * blockSize))
            {
                orc_mv old = previous[testy * prevMvStride + testx];
                int error = orc_mcstf_motion_error(src, stride, buf, blockX, blockY, old.x * factor, old.y * factor,
                                                   blockSize, *leastError);
                if (error < *leastError)
                {
                    best->x = old.x * factor;
                    best->y = old.y * factor;
                    *leastError = error;
                }
            }
        }
    }
    int error = orc_mcstf_motion_error(src, stride, buf, blockX, blockY, 0, 0, blockSize, *leastError);
    if (error < *leastError)
    {
        best->x = best->y = 0;
        *leastError = error;
    }
}

static void try_full_pel_square(const pixel* src, int stride, const pixel* buf, int blockX, int blockY, int blockSize,
                                int range, orc_mv* best, int* leastError)
{
    const orc_mv prevBest = *best;
    for (int y2 = prevBest.y / MV_FACTOR - range; y2 <= prevBest.y / MV_FACTOR + range; y2++)
        for (int x2 = prevBest.x / MV_FACTOR - range; x2 <= prevBest.x / MV_FACTOR + range; x2++)
        {
            int error = orc_mcstf_motion_error(src, stride, buf, blockX, blockY, x2 * MV_FACTOR, y2 * MV_FACTOR,
                                               blockSize, *leastError);
            if (error < *leastError)
            {
                best->x = x2 * MV_FACTOR;
                best->y = y2 * MV_FACTOR;
                *leastError = error;
            }
        }
}

static void try_vector(const pixel* src, int stride, const pixel* buf, int blockX, int blockY, int blockSize, orc_mv mv,
                       orc_mv* best, int* leastError)
{
    int error = orc_mcstf_motion_error(src, stride, buf, blockX, blockY, mv.x, mv.y, blockSize, *leastError);
    if (error < *leastError)
    {
        *best = mv;
        *leastError = error;
    }
}

/* motionEstimationLuma, temporalfilter.cpp:642-748 */
void orc_mcstf_me_luma(orc_mv* mvs, uint32_t mvStride, const pixel* src, int stride, int height, int width,
                       const pixel* buf, int row, int rowSize, const orc_mv* previous, uint32_t prevMvStride, int factor)
{
    const int range = SEARCH_RANGE, blockSize = BLOCK_SIZE, stepSize = BLOCK_SIZE;
    const int rowStart = row * rowSize;
    if (rowStart > height)
        return;
    const int rowEnd = !rowSize ? height : (rowStart + rowSize < height ? rowStart + rowSize : height);

    for (int blockY = rowStart; blockY + blockSize <= rowEnd; blockY += stepSize)
        for (int blockX = 0; blockX + blockSize <= width; blockX += stepSize)
        {
            orc_mv best = { 0, 0 };
            int leastError = INT_MAX;
            if (previous)
                try_previous(src, stride, buf, blockX, blockY, blockSize, width, height, previous, prevMvStride, factor,
                             &best, &leastError);
            try_full_pel_square(src, stride, buf, blockX, blockY, blockSize, range, &best, &leastError);
            if (blockX > 0)
                try_vector(src, stride, buf, blockX, blockY, blockSize,
                           mvs[(blockY / stepSize) * mvStride + (blockX - stepSize) / stepSize], &best, &leastError);
            mvs[(blockY / stepSize) * mvStride + blockX / stepSize] = best;
        }
}

/* motionEstimationLumaDoubleRes, temporalfilter.cpp:751-897 */
void orc_mcstf_me_luma_double_res(orc_mv* mvs, uint32_t mvStride, const pixel* orig, int stride, int origWidth,
                                  int origHeight, const pixel* buffer, const orc_mv* previous, uint32_t prevMvStride,
                                  int factor, int* minError, int row, int rowSize)
{
    const int blockSize = BLOCK_SIZE / 2, stepSize = BLOCK_SIZE / 2;
    const int rowStart = row * rowSize;
    if (rowStart > origHeight)
        return;
    const int rowEnd = !rowSize ? origHeight : (rowStart + rowSize < origHeight ? rowStart + rowSize : origHeight);

    for (int blockY = rowStart; blockY + blockSize <= rowEnd; blockY += stepSize)
        for (int blockX = 0; blockX + blockSize <= origWidth; blockX += stepSize)
        {
            orc_mv best = { 0, 0 };
            int leastError = INT_MAX;
            int range = 0;                       /* stays 0 when a coarser level is given (:755, :782-785) */
            if (!previous)
                range = 8;
            else
                try_previous(orig, stride, buffer, blockX, blockY, blockSize, origWidth, origHeight, previous,
                             prevMvStride, factor, &best, &leastError);
            try_full_pel_square(orig, stride, buffer, blockX, blockY, blockSize, range, &best, &leastError);

            const orc_mv prevBest = best;
            const int doubleRange = 3;
            for (int y2 = prevBest.y - doubleRange; y2 <= prevBest.y + doubleRange; y2++)
                for (int x2 = prevBest.x - doubleRange; x2 <= prevBest.x + doubleRange; x2++)
                {
                    orc_mv mv = { x2, y2 };
                    try_vector(orig, stride, buffer, blockX, blockY, blockSize, mv, &best, &leastError);
                }

            /* the block above, when the same row task produced it */
            if (blockY != rowStart)
                try_vector(orig, stride, buffer, blockX, blockY, blockSize,
                           mvs[((blockY - stepSize) / stepSize) * mvStride + blockX / stepSize], &best, &leastError);
            if (blockX > 0)
                try_vector(orig, stride, buffer, blockX, blockY, blockSize,
                           mvs[(blockY / stepSize) * mvStride + (blockX - stepSize) / stepSize], &best, &leastError);

            uint64_t sumSqr = orc_pixel_var(blockSize, orig + (intptr_t)blockY * stride + blockX, stride);
            uint32_t sum = (uint32_t)sumSqr, sqr = (uint32_t)(sumSqr >> 32);
            const int N = blockSize * blockSize;
            double variance = (double)sqr - ((double)sum * sum) / N;
            leastError = (int)(20 * ((leastError + 5.0) / (variance + 5.0)) + (leastError / (blockSize * blockSize)) / 50);

            const int mvIdx = (blockY / stepSize) * mvStride + blockX / stepSize;
            mvs[mvIdx] = best;
            minError[mvIdx] = leastError;
        }
}

/* One pyramid level for one reference picture, as the lookahead's row tasks run it
 * (slicetype.cpp:2280-2297 issues them, :4225-4233 executes them).  Planes are origin pointers:
 * level 0 the quarter-res planes, level 1 the half-res planes, levels 2 and 3 the padded pictures. */
void orc_mcstf_level(int level, const orc_mcstf_planes* cur, const orc_mcstf_planes* ref, orc_mcstf_mvs* out)
{
    enum { ROW_SIZE = 16 };                   /* PARALLEL_ME_ROWSIZE, slicetype.h:67 */
    const int origHeight = cur->picHeight;
    const int levelHeight[4] = { origHeight >> 2, origHeight >> 1, origHeight, origHeight };
    const int numBlockRows = (levelHeight[level] + ROW_SIZE - 1) / ROW_SIZE;
    for (int row = 0; row < numBlockRows; row++)
    {
        if (level == 0)
            orc_mcstf_me_luma(out->mvs0, out->mvsStride0, cur->lowerRes, (int)(cur->lumaStride / 2), cur->lines / 2,
                              cur->width / 2, ref->lowerRes, row, ROW_SIZE, NULL, 0, 1);
        else if (level == 1)
            orc_mcstf_me_luma(out->mvs1, out->mvsStride1, cur->lowres, (int)cur->lumaStride, cur->lines, cur->width,
                              ref->lowres, row, ROW_SIZE, out->mvs0, out->mvsStride0, 2);
        else if (level == 2)
            orc_mcstf_me_luma(out->mvs2, out->mvsStride2, cur->pic, (int)cur->picStride, cur->picHeight, cur->picWidth,
                              ref->pic, row, ROW_SIZE, out->mvs1, out->mvsStride1, 2);
        else
            orc_mcstf_me_luma_double_res(out->mvs, out->mvsStride, cur->pic, (int)cur->picStride, cur->picWidth,
                                         cur->picHeight, ref->pic, out->mvs2, out->mvsStride2, 1, out->error, row,
                                         ROW_SIZE);
    }
}
