/* TEST ORACLE - not part of the product.  CPU restatement of the picture
 * statistics behind x265's histogram-based scene-cut detection, used only by
 * tests to check the HIP path.
 *
 * Follows LookaheadTLD::collectPictureStatistics and its helpers
 * (source/encoder/slicetype.cpp:1550-1834) and the quarter-sample picture of
 * Lowres::init (source/common/lowres.cpp:40-57, 221-231, 404-414), 4:2:0. */
#include "orc.h"

#include <stdlib.h>
#include <string.h>

static inline int imin(int a, int b) { return a < b ? a : b; }

/* source/common/lowres.cpp:40-57 */
static void frame_lowres_core(const pixel* src0, pixel* dst0, intptr_t src_stride, intptr_t dst_stride, int width, int height)
{
    for (int y = 0; y < height; y++)
    {
        const pixel* src1 = src0 + src_stride;
        for (int x = 0; x < width; x++)
        {
            int a = (src0[2 * x] + src1[2 * x] + 1) >> 1, b = (src0[2 * x + 1] + src1[2 * x + 1] + 1) >> 1;
            dst0[x] = (pixel)((a + b + 1) >> 1);
        }
        src0 += src_stride * 2;
        dst0 += dst_stride;
    }
}

/* source/encoder/slicetype.cpp:1655-1683 */
static void calculate_histogram(const pixel* inputSrc, uint32_t inputWidth, uint32_t inputHeight, intptr_t stride,
                                uint8_t dsFactor, uint32_t* histogram, uint64_t* sum)
{
    *sum = 0;
    const int shift = ORC_DEPTH - 8;
    for (uint32_t v = 0; v < inputHeight; v += dsFactor)
    {
        for (uint32_t h = 0; h < inputWidth; h += dsFactor)
        {
            pixel val = (pixel)(inputSrc[h] >> shift);
            ++histogram[val];
            *sum += val;
        }
        inputSrc += (stride << (dsFactor >> 1));
    }
}

/* calcVariance (:1550-1562) on a block whose samples beyond the picture are the replicated
 * edge, as PicYuv::copyFromPicture pads them (source/common/picyuv.cpp:340-400) */
static uint32_t calc_variance(const pixel* plane, intptr_t stride, int w, int h, int x0, int y0, int size, int shift)
{
    pixel blk[64];
    for (int y = 0; y < size; y++)
        for (int x = 0; x < size; x++)
            blk[y * size + x] = plane[(intptr_t)imin(y0 + y, h - 1) * stride + imin(x0 + x, w - 1)];
    uint64_t sum_ssd = orc_pixel_var(size, blk, size);
    uint32_t sum = (uint32_t)sum_ssd, ssd = (uint32_t)(sum_ssd >> 32);
    return ssd - (uint32_t)(((uint64_t)sum * sum) >> shift);
}

/* one plane's share of computePictureStatistics (:1567-1650) */
static uint16_t picture_variance(const pixel* plane, intptr_t stride, int w, int h, int size, int shift)
{
    uint64_t picTotVariance = 0;
    for (int by = 0; by < h; by += size)
    {
        uint64_t rowVariance = 0;
        for (int bx = 0; bx < w; bx += size)
            rowVariance += calc_variance(plane, stride, w, h, bx, by, size, shift);
        picTotVariance += (uint16_t)(rowVariance / w);
    }
    return (uint16_t)(picTotVariance / h);
}

/* lowresPlane0: origin of the half-res plane 0 (Lowres::lowresPlane[0]); y/u/v: tight source planes */
void orc_collect_picture_statistics(const pixel* lowresPlane0, intptr_t lowresStride, const pixel* y, intptr_t sy,
                                    const pixel* u, const pixel* v, intptr_t sc, int w, int h, orc_hist_stats* out)
{
    memset(out, 0, sizeof(*out));
    /* Lowres::create (:221-231) and Lowres::init (:404-414) */
    const uint32_t qw = w / 4, qh = h / 4, org = 16;
    const intptr_t qstride = qw + 2 * org;
    pixel* quarter = (pixel*)calloc((size_t)qstride * (qh + 2 * org), sizeof(pixel));
    frame_lowres_core(lowresPlane0, quarter + org + org * qstride, lowresStride, qstride, qw, qh);

    uint64_t sumAverageIntensity = 0, sumAverageIntensityCb = 0, sumAverageIntensityCr = 0, sum;

    /* computeIntensityHistogramBinsLuma (:1764-1807) */
    {
        uint32_t segmentWidth = qw / 4, segmentHeight = qh / 4;
        for (uint32_t sx = 0; sx < 4; sx++)
            for (uint32_t sy_ = 0; sy_ < 4; sy_++)
            {
                uint32_t* hist = out->picHistogram[sx][sy_][0];
                for (int i = 0; i < 256; i++)
                    hist[i] = 1;
                uint32_t wo = (sx == 3) ? qw - 4 * segmentWidth : 0;
                uint32_t ho = (sy_ == 3) ? qh - 4 * segmentHeight : 0;
                calculate_histogram(quarter + (org + sx * segmentWidth) + (org + sy_ * segmentHeight) * qstride,
                                    segmentWidth + wo, segmentHeight + ho, qstride, 1, hist, &sum);
                out->averageIntensityPerSegment[sx][sy_][0] =
                    (uint8_t)((sum + (((segmentWidth + wo) * (segmentWidth + ho)) >> 1)) /
                              ((segmentWidth + wo) * (segmentHeight + ho)));
                sumAverageIntensity += (sum << 4);
                for (int i = 0; i < 256; i++)
                    hist[i] <<= 4;
            }
    }
    /* computeIntensityHistogramBinsChroma (:1688-1759) */
    {
        const uint8_t dsFactor = 4;
        uint32_t segmentWidth = (uint32_t)w / 4, segmentHeight = (uint32_t)h / 4;
        for (uint32_t sx = 0; sx < 4; sx++)
            for (uint32_t sy_ = 0; sy_ < 4; sy_++)
            {
                uint32_t wo = (sx == 3) ? w - 4 * segmentWidth : 0;
                uint32_t ho = (sy_ == 3) ? h - 4 * segmentHeight : 0;
                for (int plane = 1; plane <= 2; plane++)
                {
                    uint32_t* hist = out->picHistogram[sx][sy_][plane];
                    for (int i = 0; i < 256; i++)
                        hist[i] = 1;
                    const pixel* src = (plane == 1 ? u : v) + ((sx * segmentWidth) >> 1) + ((sy_ * segmentHeight) >> 1) * sc;
                    calculate_histogram(src, (segmentWidth + wo) >> 1, (segmentHeight + ho) >> 1, sc, dsFactor, hist, &sum);
                    sum <<= dsFactor;
                    /* the Cr divisor mixes the height remainder into the width term, as the reference does (:1749) */
                    uint32_t area = (segmentWidth + wo) * (segmentHeight + ho);
                    uint32_t div = plane == 1 ? area >> 2 : ((segmentWidth + ho) * (segmentHeight + ho)) >> 2;
                    if (plane == 1)
                        sumAverageIntensityCb += sum;
                    else
                        sumAverageIntensityCr += sum;
                    out->averageIntensityPerSegment[sx][sy_][plane] = (uint8_t)((sum + (area >> 3)) / div);
                    for (int i = 0; i < 256; i++)
                        hist[i] <<= dsFactor;
                }
            }
    }
    /* collectPictureStatistics (:1827-1829) */
    const uint32_t area = (uint32_t)w * (uint32_t)h;
    out->averageIntensity[0] = (uint8_t)((sumAverageIntensity + (area >> 1)) / area);
    out->averageIntensity[1] = (uint8_t)((sumAverageIntensityCb + (area >> 3)) / (area >> 2));
    out->averageIntensity[2] = (uint8_t)((sumAverageIntensityCr + (area >> 3)) / (area >> 2));

    out->picAvgVariance = picture_variance(y, sy, w, h, 8, 6);
    out->picAvgVarianceCb = picture_variance(u, sc, w >> 1, h >> 1, 4, 4);
    out->picAvgVarianceCr = picture_variance(v, sc, w >> 1, h >> 1, 4, 4);
    free(quarter);
}
