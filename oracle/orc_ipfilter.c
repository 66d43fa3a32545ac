/* CPU oracle (test infrastructure only; parity unpinned, see orc.h).
 * HEVC 8-tap luma interpolation, pixel-to-pixel forms, restated from
 * source/common/ipfilter.cpp (horizontal :79-118, vertical :164-203,
 * separable :362-369 via :120-162 and :319-360). */
#include "orc.h"
#include <stdlib.h>

/* source/common/constants.cpp:250-256 */
static const int16_t k_luma_taps[4][8] = {
    {  0, 0,   0, 64,  0,   0, 0,  0 },
    { -1, 4, -10, 58, 17,  -5, 1,  0 },
    { -1, 4, -11, 40, 40, -11, 4, -1 },
    {  0, 1,  -5, 17, 58, -10, 4, -1 },
};

#define IF_PREC 14                     /* IF_INTERNAL_PREC */
#define IF_FPREC 6                     /* IF_FILTER_PREC */
#define IF_OFFS (1 << (IF_PREC - 1))   /* IF_INTERNAL_OFFS */

static inline pixel clip_pix(int v)
{
    int16_t s = (int16_t)v;
    return (pixel)(s < 0 ? 0 : (s > ORC_PIXEL_MAX ? ORC_PIXEL_MAX : s));
}

void orc_luma_hpp(int w, int h, const pixel* src, intptr_t ss, pixel* dst, intptr_t ds, int coeffIdx)
{
    const int16_t* c = k_luma_taps[coeffIdx];
    src -= 3;
    for (int y = 0; y < h; y++, src += ss, dst += ds)
        for (int x = 0; x < w; x++)
        {
            int sum = 0;
            for (int k = 0; k < 8; k++)
                sum += src[x + k] * c[k];
            dst[x] = clip_pix((sum + (1 << (IF_FPREC - 1))) >> IF_FPREC);
        }
}

void orc_luma_vpp(int w, int h, const pixel* src, intptr_t ss, pixel* dst, intptr_t ds, int coeffIdx)
{
    const int16_t* c = k_luma_taps[coeffIdx];
    src -= 3 * ss;
    for (int y = 0; y < h; y++, src += ss, dst += ds)
        for (int x = 0; x < w; x++)
        {
            int sum = 0;
            for (int k = 0; k < 8; k++)
                sum += src[x + k * ss] * c[k];
            dst[x] = clip_pix((sum + (1 << (IF_FPREC - 1))) >> IF_FPREC);
        }
}

/* horizontal pass to 14-bit intermediates over h+7 rows, then vertical pass
 * with the internal offset folded back in */
void orc_luma_hvpp(int w, int h, const pixel* src, intptr_t ss, pixel* dst, intptr_t ds, int idxX, int idxY)
{
    const int headRoom = IF_PREC - ORC_DEPTH;
    const int shiftH = IF_FPREC - headRoom;
    const int offH = (int)((unsigned)-IF_OFFS << shiftH);
    const int shiftV = IF_FPREC + headRoom;
    const int offV = (1 << (shiftV - 1)) + (IF_OFFS << IF_FPREC);
    const int16_t* cx = k_luma_taps[idxX];
    const int16_t* cy = k_luma_taps[idxY];
    int16_t* imm = (int16_t*)malloc(sizeof(int16_t) * w * (h + 7));
    const pixel* s = src - 3 - 3 * ss;
    for (int y = 0; y < h + 7; y++, s += ss)
        for (int x = 0; x < w; x++)
        {
            int sum = 0;
            for (int k = 0; k < 8; k++)
                sum += s[x + k] * cx[k];
            imm[y * w + x] = (int16_t)((sum + offH) >> shiftH);
        }
    for (int y = 0; y < h; y++, dst += ds)
        for (int x = 0; x < w; x++)
        {
            int sum = 0;
            for (int k = 0; k < 8; k++)
                sum += imm[(y + k) * w + x] * cy[k];
            dst[x] = clip_pix((sum + offV) >> shiftV);
        }
    free(imm);
}
