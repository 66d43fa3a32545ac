/* CPU oracle (test infrastructure only; parity unpinned, see orc.h).
 * HEVC intra reference smoothing and DC / planar / angular prediction,
 * restated from source/common/intrapred.cpp. */
#include "orc.h"
#include <string.h>

/* neighbour array layout used by every function here:
 *   [0] top-left, [1 .. 2N] top and top-right, [2N+1 .. 4N] left and bottom-left */

/* source/common/constants.cpp:561-567 */
static const uint8_t k_filter_flags[35] = {
    0x38, 0x00,
    0x38, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x20, 0x00, 0x20, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30,
    0x38, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x20, 0x00, 0x20, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30,
    0x38,
};

int orc_intra_filter_flag(int mode, int size) { return !!(k_filter_flags[mode] & size); }

/* 1:2:1 smoothing, both far ends kept: source/common/intrapred.cpp:31-51 */
void orc_intra_filter(int n, const pixel* s, pixel* f)
{
    const int n2 = 2 * n;
    f[0] = (pixel)((2 * s[0] + s[1] + s[n2 + 1] + 2) >> 2);
    for (int i = 1; i < n2; i++)
        f[i] = (pixel)((2 * s[i] + s[i - 1] + s[i + 1] + 2) >> 2);
    f[n2] = s[n2];
    f[n2 + 1] = (pixel)((2 * s[n2 + 1] + s[0] + s[n2 + 2] + 2) >> 2);
    for (int i = n2 + 2; i < 2 * n2; i++)
        f[i] = (pixel)((2 * s[i] + s[i - 1] + s[i + 1] + 2) >> 2);
    f[2 * n2] = s[2 * n2];
}

/* DC with optional edge smoothing: source/common/intrapred.cpp:53-85 */
static void pred_dc(int n, pixel* dst, intptr_t ds, const pixel* src, int bFilter)
{
    const pixel* above = src + 1;
    const pixel* left = src + 2 * n + 1;
    int dc = n;
    for (int i = 0; i < n; i++)
        dc += above[i] + left[i];
    dc /= 2 * n;
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)
            dst[y * ds + x] = (pixel)dc;
    if (bFilter)
    {
        dst[0] = (pixel)((above[0] + left[0] + 2 * dc + 2) >> 2);
        for (int x = 1; x < n; x++)
            dst[x] = (pixel)((above[x] + 3 * dc + 2) >> 2);
        for (int y = 1; y < n; y++)
            dst[y * ds] = (pixel)((left[y] + 3 * dc + 2) >> 2);
    }
}

/* planar: source/common/intrapred.cpp:87-100 */
static void pred_planar(int n, pixel* dst, intptr_t ds, const pixel* src)
{
    const pixel* above = src + 1;
    const pixel* left = src + 2 * n + 1;
    int log2n = 0;
    while ((1 << log2n) < n)
        log2n++;
    int tr = above[n], bl = left[n];
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)
            dst[y * ds + x] = (pixel)(((n - 1 - x) * left[y] + (n - 1 - y) * above[x] +
                                       (x + 1) * tr + (y + 1) * bl + n) >> (log2n + 1));
}

/* angular modes 2..34: source/common/intrapred.cpp:102-204.  Horizontal modes
 * (<18) are predicted as their vertical mirror on swapped neighbours and the
 * result is transposed. */
static void pred_angular(int n, pixel* dst, intptr_t ds, const pixel* src0, int mode, int bFilter)
{
    static const int8_t angles[17] = { -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
    static const int16_t invAngles[8] = { 4096, 1638, 910, 630, 482, 390, 315, 256 };
    const int n2 = 2 * n;
    const int hor = mode < 18;
    pixel nb[129];
    const pixel* src = src0;
    if (hor)
    {
        nb[0] = src0[0];
        for (int i = 0; i < n2; i++)
        {
            nb[1 + i] = src0[n2 + 1 + i];
            nb[n2 + 1 + i] = src0[1 + i];
        }
        src = nb;
    }
    const int angOff = hor ? 10 - mode : mode - 26;
    const int angle = angles[8 + angOff];
    pixel out[64 * 64];

    if (angle == 0)
    {
        for (int y = 0; y < n; y++)
            for (int x = 0; x < n; x++)
                out[y * n + x] = src[1 + x];
        if (bFilter)
        {
            int tl = src[0], top = src[1];
            for (int y = 0; y < n; y++)
            {
                int v = (int16_t)(top + ((src[n2 + 1 + y] - tl) >> 1));
                out[y * n] = (pixel)(v < 0 ? 0 : (v > ORC_PIXEL_MAX ? ORC_PIXEL_MAX : v));
            }
        }
    }
    else
    {
        pixel refbuf[64];
        const pixel* ref;
        if (angle < 0)
        {
            int nproj = -((n * angle) >> 5) - 1;
            pixel* rp = refbuf + nproj + 1;
            int inv = invAngles[-angOff - 1];
            int invSum = 128;
            for (int i = 0; i < nproj; i++)
            {
                invSum += inv;
                rp[-2 - i] = src[n2 + (invSum >> 8)];
            }
            for (int i = 0; i < n + 1; i++)
                rp[-1 + i] = src[i];
            ref = rp;
        }
        else
            ref = src + 1;

        int angSum = 0;
        for (int y = 0; y < n; y++)
        {
            angSum += angle;
            int off = angSum >> 5, frac = angSum & 31;
            for (int x = 0; x < n; x++)
                out[y * n + x] = frac ? (pixel)(((32 - frac) * ref[off + x] + frac * ref[off + x + 1] + 16) >> 5)
                                      : ref[off + x];
        }
    }

    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)
            dst[y * ds + x] = hor ? out[x * n + y] : out[y * n + x];
}

/* dispatch in the shape of primitives.cu[size].intra_pred[mode]
 * (mode 0 planar, 1 DC, 2..34 angular; source/common/intrapred.cpp:247-263) */
void orc_intra_pred(int size, int mode, pixel* dst, intptr_t dstStride, const pixel* srcPix, int bFilter)
{
    if (mode == 0)
        pred_planar(size, dst, dstStride, srcPix);
    else if (mode == 1)
        pred_dc(size, dst, dstStride, srcPix, bFilter);
    else
        pred_angular(size, dst, dstStride, srcPix, mode, bFilter);
}
