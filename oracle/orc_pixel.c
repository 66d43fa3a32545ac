/* CPU oracle (test infrastructure only; parity unpinned, see orc.h).
 * Pixel-compare and plane primitives, restated from source/common/pixel.cpp
 * with straight int arithmetic instead of the reference's SWAR packing. */
#include "orc.h"
#include <stdlib.h>
#include <string.h>

int orc_depth(void) { return ORC_DEPTH; }

/* sum of absolute differences: source/common/pixel.cpp:40-55 */
/* Executed block compares, for the benchmark's algorithmic-operation count: [phase][0] SADs and [phase][1] SATDs
 * in units of 64 pixels (an 8x8 block compare), [phase][2] 8x8 intra predictions.  The caller names the phase
 * (orc_op_phase: 0 intra estimate, 1 motion search, 2 cost decision, 3 weight analysis, 4 anything else). */
long long orc_op_counters[5][3];
int orc_op_phase = 4;

int orc_sad(int w, int h, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{
    int acc = 0;
    orc_op_counters[orc_op_phase][0] += (w * h + 63) >> 6;
    for (int y = 0; y < h; y++, a += sa, b += sb)
        for (int x = 0; x < w; x++)
            acc += abs((int)a[x] - (int)b[x]);
    return acc;
}

/* three / four SADs against one encode block whose stride is implicitly
 * FENC_STRIDE: source/common/pixel.cpp:74-119 */
void orc_sad_x3(int w, int h, const pixel* fenc, const pixel* r0, const pixel* r1, const pixel* r2,
                intptr_t rstride, int32_t* res)
{
    res[0] = orc_sad(w, h, fenc, ORC_FENC_STRIDE, r0, rstride);
    res[1] = orc_sad(w, h, fenc, ORC_FENC_STRIDE, r1, rstride);
    res[2] = orc_sad(w, h, fenc, ORC_FENC_STRIDE, r2, rstride);
}

void orc_sad_x4(int w, int h, const pixel* fenc, const pixel* r0, const pixel* r1, const pixel* r2,
                const pixel* r3, intptr_t rstride, int32_t* res)
{
    res[0] = orc_sad(w, h, fenc, ORC_FENC_STRIDE, r0, rstride);
    res[1] = orc_sad(w, h, fenc, ORC_FENC_STRIDE, r1, rstride);
    res[2] = orc_sad(w, h, fenc, ORC_FENC_STRIDE, r2, rstride);
    res[3] = orc_sad(w, h, fenc, ORC_FENC_STRIDE, r3, rstride);
}

/* four-point butterfly used by every Hadamard below (HADAMARD4 macro,
 * source/common/pixel.cpp:190-199) */
static void bfly4(int* d, int s0, int s1, int s2, int s3)
{
    int t0 = s0 + s1, t1 = s0 - s1, t2 = s2 + s3, t3 = s2 - s3;
    d[0] = t0 + t2;
    d[2] = t0 - t2;
    d[1] = t1 + t3;
    d[3] = t1 - t3;
}

/* sum of |H4 * D * H4| over one 4x4 block, not yet normalised */
static int had4x4_abs(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{
    int t[4][4], c[4], acc = 0;
    for (int y = 0; y < 4; y++, a += sa, b += sb)
        bfly4(t[y], a[0] - b[0], a[1] - b[1], a[2] - b[2], a[3] - b[3]);
    for (int x = 0; x < 4; x++)
    {
        bfly4(c, t[0][x], t[1][x], t[2][x], t[3][x]);
        acc += abs(c[0]) + abs(c[1]) + abs(c[2]) + abs(c[3]);
    }
    return acc;
}

/* SATD.  4-wide and 12-wide partitions are tiled with 4x4 transforms
 * (satd_4x4 returns sum>>1, source/common/pixel.cpp:210-236, tiling :263-275);
 * every other width is tiled with 8x4 pairs whose two 4x4 sums are added
 * before the single >>1 (source/common/pixel.cpp:239-261, :277-289).
 * Partition -> tiling map: source/common/pixel.cpp:1148-1172. */
int orc_satd(int w, int h, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{
    int acc = 0;
    orc_op_counters[orc_op_phase][1] += (w * h + 63) >> 6;
    if (w & 7)
    {
        for (int y = 0; y < h; y += 4)
            for (int x = 0; x < w; x += 4)
                acc += had4x4_abs(a + y * sa + x, sa, b + y * sb + x, sb) >> 1;
    }
    else
    {
        for (int y = 0; y < h; y += 4)
            for (int x = 0; x < w; x += 8)
                acc += (had4x4_abs(a + y * sa + x, sa, b + y * sb + x, sb) +
                        had4x4_abs(a + y * sa + x + 4, sa, b + y * sb + x + 4, sb)) >> 1;
    }
    return acc;
}

/* sum of |H8 * D * H8| over one 8x8 block (source/common/pixel.cpp:291-326);
 * the order of the coefficients does not matter for a sum of magnitudes */
static int had8x8_abs(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{
    int m[8][8];
    for (int y = 0; y < 8; y++, a += sa, b += sb)
    {
        int lo[4], hi[4];
        bfly4(lo, a[0] - b[0], a[1] - b[1], a[2] - b[2], a[3] - b[3]);
        bfly4(hi, a[4] - b[4], a[5] - b[5], a[6] - b[6], a[7] - b[7]);
        for (int k = 0; k < 4; k++)
        {
            m[y][k] = lo[k] + hi[k];
            m[y][k + 4] = lo[k] - hi[k];
        }
    }
    int acc = 0;
    for (int x = 0; x < 8; x++)
    {
        int lo[4], hi[4];
        bfly4(lo, m[0][x], m[1][x], m[2][x], m[3][x]);
        bfly4(hi, m[4][x], m[5][x], m[6][x], m[7][x]);
        for (int k = 0; k < 4; k++)
            acc += abs(lo[k] + hi[k]) + abs(lo[k] - hi[k]);
    }
    return acc;
}

/* cu[].sa8d by block size (table: source/common/pixel.cpp:1180-1184):
 * 4 -> satd_4x4 alias; 8 -> (sum+2)>>2 (:328-331); 16 -> four 8x8 sums rounded
 * once (:333-343); 32/64 -> sum of sa8d_16x16 tiles (:358-369) */
int orc_sa8d(int size, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{
    if (size == 4)
        return had4x4_abs(a, sa, b, sb) >> 1;
    if (size == 8)
        return (had8x8_abs(a, sa, b, sb) + 2) >> 2;
    int acc = 0;
    for (int y = 0; y < size; y += 16)
        for (int x = 0; x < size; x += 16)
        {
            const pixel* pa = a + y * sa + x;
            const pixel* pb = b + y * sb + x;
            int s = had8x8_abs(pa, sa, pb, sb) + had8x8_abs(pa + 8, sa, pb + 8, sb) +
                    had8x8_abs(pa + 8 * sa, sa, pb + 8 * sb, sb) +
                    had8x8_abs(pa + 8 * sa + 8, sa, pb + 8 * sb + 8, sb);
            acc += (s + 2) >> 2;
        }
    return acc;
}

/* rounding average of two blocks: source/common/pixel.cpp:537-549 */
void orc_pixelavg_pp(int w, int h, pixel* dst, intptr_t ds, const pixel* s0, intptr_t ss0,
                     const pixel* s1, intptr_t ss1)
{
    for (int y = 0; y < h; y++, dst += ds, s0 += ss0, s1 += ss1)
        for (int x = 0; x < w; x++)
            dst[x] = (pixel)((s0[x] + s1[x] + 1) >> 1);
}

static inline int ravg(int a, int b) { return (a + b + 1) >> 1; }

/* half-resolution plane plus H / V / HV half-pel planes by nested rounding
 * averages, vertical pairs first: source/common/pixel.cpp:596-620 */
void orc_frame_init_lowres_core(const pixel* src0, pixel* dst0, pixel* dsth, pixel* dstv, pixel* dstc,
                                intptr_t src_stride, intptr_t dst_stride, int width, int height)
{
    for (int y = 0; y < height; y++)
    {
        const pixel* r0 = src0 + (intptr_t)2 * y * src_stride;
        const pixel* r1 = r0 + src_stride;
        const pixel* r2 = r1 + src_stride;
        for (int x = 0; x < width; x++)
        {
            int v01_0 = ravg(r0[2 * x], r1[2 * x]);
            int v01_1 = ravg(r0[2 * x + 1], r1[2 * x + 1]);
            int v01_2 = ravg(r0[2 * x + 2], r1[2 * x + 2]);
            int v12_0 = ravg(r1[2 * x], r2[2 * x]);
            int v12_1 = ravg(r1[2 * x + 1], r2[2 * x + 1]);
            int v12_2 = ravg(r1[2 * x + 2], r2[2 * x + 2]);
            dst0[y * dst_stride + x] = (pixel)ravg(v01_0, v01_1);
            dsth[y * dst_stride + x] = (pixel)ravg(v01_1, v01_2);
            dstv[y * dst_stride + x] = (pixel)ravg(v12_0, v12_1);
            dstc[y * dst_stride + x] = (pixel)ravg(v12_1, v12_2);
        }
    }
}

/* the same four planes with the plain rounded 2x2 mean, used when the
 * temporal filter is on: source/common/pixel.cpp:1017-1039 */
void orc_frame_init_lowres_core_mcstf(const pixel* src0, pixel* dst0, pixel* dsth, pixel* dstv, pixel* dstc,
                                      intptr_t src_stride, intptr_t dst_stride, int width, int height)
{
    for (int y = 0; y < height; y++)
    {
        const pixel* r0 = src0 + (intptr_t)2 * y * src_stride;
        const pixel* r1 = r0 + src_stride;
        const pixel* r2 = r1 + src_stride;
        for (int x = 0; x < width; x++)
        {
            int c0 = 2 * x, c1 = c0 + 1, c2 = c0 + 2;
            dst0[y * dst_stride + x] = (pixel)((r0[c0] + r1[c0] + r0[c1] + r1[c1] + 2) >> 2);
            dsth[y * dst_stride + x] = (pixel)((r0[c1] + r1[c1] + r0[c2] + r1[c2] + 2) >> 2);
            dstv[y * dst_stride + x] = (pixel)((r1[c0] + r2[c0] + r1[c1] + r2[c1] + 2) >> 2);
            dstc[y * dst_stride + x] = (pixel)((r1[c1] + r2[c1] + r1[c2] + r2[c2] + 2) >> 2);
        }
    }
}

/* replicate edges into the margins: left/right per row
 * (source/common/ipfilter.cpp:59-77), then whole padded top and bottom rows,
 * `stride` pixels starting marginX left of the picture
 * (source/common/pixel.cpp:1044-1058) */
void orc_extend_pic_border(pixel* pic, intptr_t stride, int width, int height, int marginX, int marginY)
{
    for (int y = 0; y < height; y++)
    {
        pixel* row = pic + y * stride;
        for (int x = 0; x < marginX; x++)
        {
            row[-marginX + x] = row[0];
            row[width + x] = row[width - 1];
        }
    }
    pixel* top = pic - marginX;
    pixel* bot = pic - marginX + (intptr_t)(height - 1) * stride;
    for (int y = 1; y <= marginY; y++)
    {
        memcpy(top - y * stride, top, stride * sizeof(pixel));
        memcpy(bot + y * stride, bot, stride * sizeof(pixel));
    }
}

/* explicit weighted prediction of a whole plane; the pixel is first raised to
 * 14-bit precision and held in an int16: source/common/pixel.cpp:510-535 */
void orc_weight_pp(const pixel* src, pixel* dst, intptr_t stride, int width, int height,
                   int w0, int round, int shift, int offset)
{
    const int correction = 14 - ORC_DEPTH;
    for (int y = 0; y < height; y++, src += stride, dst += stride)
        for (int x = 0; x < width; x++)
        {
            int16_t val = (int16_t)(src[x] << correction);
            int v = ((w0 * val + round) >> shift) + offset;
            dst[x] = (pixel)(v < 0 ? 0 : (v > ORC_PIXEL_MAX ? ORC_PIXEL_MAX : v));
        }
}

/* sum and sum of squares packed as sum + (sqr << 32), both accumulated in
 * uint32: source/common/pixel.cpp:695-712 */
uint64_t orc_pixel_var(int size, const pixel* pix, intptr_t stride)
{
    uint32_t sum = 0, sqr = 0;
    for (int y = 0; y < size; y++, pix += stride)
        for (int x = 0; x < size; x++)
        {
            sum += pix[x];
            sqr += (uint32_t)pix[x] * pix[x];
        }
    return sum + ((uint64_t)sqr << 32);
}
