/* CPU oracle (test infrastructure only; parity unpinned, see orc.h).
 * Integer motion search (diamond / hexagon / uneven multi-hexagon / star /
 * exhaustive) with sub-pel refinement, restated from
 * source/encoder/motion.cpp:923-1868, and the lowres quarter-pel sampling of
 * source/common/lowres.h:72-117.  The visiting order of candidates and every
 * strict '<' tie-break follow the reference; its quirks are kept and marked. */
#include "orc.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

/* source/encoder/motion.cpp:48-58 */
typedef struct { int hpel_iters, hpel_dirs, qpel_iters, qpel_dirs, hpel_satd; } subpel_workload;
static const subpel_workload k_workload[8] = {
    { 1, 4, 0, 4, 0 }, { 1, 4, 1, 4, 0 }, { 1, 4, 1, 4, 1 }, { 2, 4, 1, 4, 1 },
    { 2, 4, 2, 4, 1 }, { 1, 8, 1, 8, 1 }, { 2, 8, 1, 8, 1 }, { 2, 8, 2, 8, 1 },
};

/* source/encoder/motion.cpp:63-84 */
static const orc_mv k_hex2[8] = { {-1,-2}, {-2,0}, {-1,2}, {1,2}, {2,0}, {1,-2}, {-1,-2}, {-2,0} };
static const uint8_t k_mod6m1[8] = { 5, 0, 1, 2, 3, 4, 5, 0 };
static const orc_mv k_square1[9] = { {0,0}, {0,-1}, {0,1}, {-1,0}, {1,0}, {-1,-1}, {-1,1}, {1,-1}, {1,1} };
static const orc_mv k_hex4[16] = {
    {0,-4}, {0,4}, {-2,-3}, {2,-3}, {-4,-2}, {4,-2}, {-4,-1}, {4,-1},
    {-4,0}, {4,0}, {-4,1}, {4,1}, {-4,2}, {4,2}, {-2,3}, {2,3},
};
static const orc_mv k_twopoint[16] = {
    {-1,0}, {0,-1}, {-1,-1}, {1,-1}, {-1,0}, {1,0}, {-1,1}, {-1,-1},
    {1,-1}, {1,1}, {-1,0}, {0,1}, {-1,1}, {1,1}, {1,0}, {0,1},
};

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int in_range(orc_mv m, orc_mv lo, orc_mv hi)
{
    return m.x >= lo.x && m.x <= hi.x && m.y >= lo.y && m.y <= hi.y;
}

void orc_me_init(orc_me* me, const uint16_t* centredCost)
{
    memset(me, 0, sizeof(*me));
    me->cost = centredCost;
    me->searchMethod = me->searchMethodL0 = me->searchMethodL1 = ORC_HEX;
    me->subpelRefine = 2;
}

/* heap-allocated instance for callers that cannot size the struct (ctypes) */
orc_me* orc_me_create(const uint16_t* centredCost)
{
    orc_me* me = (orc_me*)malloc(sizeof(orc_me));
    orc_me_init(me, centredCost);
    return me;
}

void orc_me_destroy(orc_me* me) { free(me); }

/* source/encoder/motion.cpp:167-191 */
void orc_me_set_source(orc_me* me, const pixel* fencY, intptr_t stride, intptr_t offset, int w, int h,
                       int method, int searchL0, int searchL1, int refine)
{
    me->bw = w;
    me->bh = h;
    me->blockOffset = offset;
    me->searchMethod = method;
    me->searchMethodL0 = searchL0;
    me->searchMethodL1 = searchL1;
    me->subpelRefine = refine;
    const pixel* s = fencY + offset;
    for (int y = 0; y < h; y++)
        memcpy(me->fenc + y * ORC_FENC_STRIDE, s + y * stride, w * sizeof(pixel));
}

static int cmp_sad(orc_me* me, const pixel* ref, intptr_t stride)
{
    me->nSad++;
    return orc_sad(me->bw, me->bh, me->fenc, ORC_FENC_STRIDE, ref, stride);
}

static int cmp_satd(orc_me* me, const pixel* ref, intptr_t stride)
{
    me->nSatd++;
    return orc_satd(me->bw, me->bh, me->fenc, ORC_FENC_STRIDE, ref, stride);
}

int orc_me_buf_satd(orc_me* me, const pixel* ref, intptr_t stride) { return cmp_satd(me, ref, stride); }

void orc_me_set_vert_restriction(orc_me* me, int on) { me->vertRestriction = on; }

/* quarter-pel sample fetch by averaging the two nearest half-pel planes:
 * source/common/lowres.h:72-93 */
const pixel* orc_lowres_mc(const orc_refplanes* ref, intptr_t blockOffset, orc_mv qmv, pixel* buf,
                           intptr_t* outstride, int hme)
{
    intptr_t ys = hme ? ref->stride / 2 : ref->stride;
    pixel* const* plane = hme ? ref->lowerPlane : ref->plane;
    int hpelA = (qmv.y & 2) | ((qmv.x & 2) >> 1);
    const pixel* a = plane[hpelA] + blockOffset + (qmv.x >> 2) + (qmv.y >> 2) * ys;
    if ((qmv.x | qmv.y) & 1)
    {
        int qx = qmv.x + (qmv.x & 1), qy = qmv.y + (qmv.y & 1);
        int hpelB = (qy & 2) | ((qx & 2) >> 1);
        const pixel* b = plane[hpelB] + blockOffset + (qx >> 2) + (qy >> 2) * ys;
        orc_pixelavg_pp(8, 8, buf, *outstride, a, ys, b, ys);
        return buf;
    }
    *outstride = ys;
    return a;
}

/* source/common/lowres.h:95-117 */
int orc_lowres_qpel_cost(orc_me* me, const orc_refplanes* ref, intptr_t blockOffset, orc_mv qmv,
                         int useSatd, int hme)
{
    pixel buf[64];
    intptr_t st = 8;
    const pixel* src = orc_lowres_mc(ref, blockOffset, qmv, buf, &st, hme);
    return useSatd ? cmp_satd(me, src, st) : cmp_sad(me, src, st);
}

/* one motionEstimate() invocation */
typedef struct {
    orc_me* me;
    const orc_refplanes* ref;
    const pixel* fref;          /* full-pel plane + blockOffset */
    intptr_t stride;
    orc_mv mvmin, mvmax;
    const uint16_t* cx;         /* cost table shifted by the predictor: cx[q] = cost[q - mvp.x] */
    const uint16_t* cy;
    orc_mv bmv;
    int bcost;
    int hme;
} srch;

/* BitCost::mvcost returns uint16_t: source/encoder/bitcost.h:56 */
static inline int mvcost_q(const srch* s, int qx, int qy) { return (uint16_t)(s->cx[qx] + s->cy[qy]); }
static inline int mvcost_f(const srch* s, int fx, int fy) { return mvcost_q(s, fx * 4, fy * 4); }

/* compare counters for workload analysis (tests / benchmarks read them through orc.py):
 * [0] full-pel SADs, [1] those inside the star search's raster pass, [2] sub-pel compares */
long long orc_me_counters[3];
static int g_inRaster;

static inline int fpel_sad(srch* s, int mx, int my)
{
    orc_me_counters[0]++;
    orc_me_counters[1] += g_inRaster;
    return cmp_sad(s->me, s->fref + mx + my * s->stride, s->stride);
}
static inline int fpel_cost(srch* s, int mx, int my) { return fpel_sad(s, mx, my) + mvcost_f(s, mx, my); }
static inline int y_ok(const srch* s, int y) { return y >= s->mvmin.y && y <= s->mvmax.y; }

/* COST_MV, source/encoder/motion.cpp:263-269 */
static void try_mv(srch* s, int mx, int my)
{
    int c = fpel_cost(s, mx, my);
    if (c < s->bcost)
    {
        s->bcost = c;
        s->bmv.x = mx;
        s->bmv.y = my;
    }
}

/* COST_MV_X4 around origin o: four costs, each accepted only if its row is in
 * range (source/encoder/motion.cpp:302-323) */
static void try_x4(srch* s, orc_mv o, const int off[8])
{
    int c[4];
    for (int k = 0; k < 4; k++)
        c[k] = fpel_cost(s, o.x + off[2 * k], o.y + off[2 * k + 1]);
    for (int k = 0; k < 4; k++)
        if (y_ok(s, o.y + off[2 * k + 1]) && c[k] < s->bcost)
        {
            s->bcost = c[k];
            s->bmv.x = o.x + off[2 * k];
            s->bmv.y = o.y + off[2 * k + 1];
        }
}

/* 8-point square refinement shared by HEX and refineMV:
 * source/encoder/motion.cpp:1120-1138 */
static void square_refine(srch* s)
{
    int dir = 0;
    orc_mv b = s->bmv;
    int c[8];
    for (int k = 0; k < 8; k++)
        c[k] = fpel_cost(s, b.x + k_square1[k + 1].x, b.y + k_square1[k + 1].y);
    int up = y_ok(s, b.y - 1), dn = y_ok(s, b.y + 1);
    if (up && c[0] < s->bcost) { s->bcost = c[0]; dir = 1; }
    if (dn && c[1] < s->bcost) { s->bcost = c[1]; dir = 2; }
    if (c[2] < s->bcost) { s->bcost = c[2]; dir = 3; }
    if (c[3] < s->bcost) { s->bcost = c[3]; dir = 4; }
    if (up && c[4] < s->bcost) { s->bcost = c[4]; dir = 5; }
    if (dn && c[5] < s->bcost) { s->bcost = c[5]; dir = 6; }
    if (up && c[6] < s->bcost) { s->bcost = c[6]; dir = 7; }
    if (dn && c[7] < s->bcost) { s->bcost = c[7]; dir = 8; }
    s->bmv.x += k_square1[dir].x;
    s->bmv.y += k_square1[dir].y;
}

/* radius-2 hexagon walk with the direction carried in the low 3 bits of the
 * cost, then the square refinement: source/encoder/motion.cpp:1041-1140 */
static void hex_search(srch* s, int merange)
{
    orc_mv b = s->bmv;
    int bc = s->bcost;
    int c0, c1, c2;

    c0 = fpel_cost(s, b.x - 2, b.y);
    c1 = fpel_cost(s, b.x - 1, b.y + 2);
    c2 = fpel_cost(s, b.x + 1, b.y + 2);
    bc <<= 3;
    if (y_ok(s, b.y))
        bc = imin(bc, (c0 << 3) + 2);
    if (y_ok(s, b.y + 2))
    {
        bc = imin(bc, (c1 << 3) + 3);
        bc = imin(bc, (c2 << 3) + 4);
    }
    c0 = fpel_cost(s, b.x + 2, b.y);
    c1 = fpel_cost(s, b.x + 1, b.y - 2);
    c2 = fpel_cost(s, b.x - 1, b.y - 2);
    if (y_ok(s, b.y))
        bc = imin(bc, (c0 << 3) + 5);
    if (y_ok(s, b.y - 2))
    {
        bc = imin(bc, (c1 << 3) + 6);
        bc = imin(bc, (c2 << 3) + 7);
    }

    if (bc & 7)
    {
        int dir = (bc & 7) - 2;
        if (y_ok(s, b.y + k_hex2[dir + 1].y))
        {
            b.x += k_hex2[dir + 1].x;
            b.y += k_hex2[dir + 1].y;
            /* half hexagons, three new points per step */
            for (int i = (merange >> 1) - 1; i > 0 && in_range(b, s->mvmin, s->mvmax); i--)
            {
                c0 = fpel_cost(s, b.x + k_hex2[dir + 0].x, b.y + k_hex2[dir + 0].y);
                c1 = fpel_cost(s, b.x + k_hex2[dir + 1].x, b.y + k_hex2[dir + 1].y);
                c2 = fpel_cost(s, b.x + k_hex2[dir + 2].x, b.y + k_hex2[dir + 2].y);
                bc &= ~7;
                if (y_ok(s, b.y + k_hex2[dir + 0].y))
                    bc = imin(bc, (c0 << 3) + 1);
                if (y_ok(s, b.y + k_hex2[dir + 1].y))
                    bc = imin(bc, (c1 << 3) + 2);
                if (y_ok(s, b.y + k_hex2[dir + 2].y))
                    bc = imin(bc, (c2 << 3) + 3);
                if (!(bc & 7))
                    break;
                dir += (bc & 7) - 2;
                dir = k_mod6m1[dir + 1];
                b.x += k_hex2[dir + 1].x;
                b.y += k_hex2[dir + 1].y;
            }
        }
    }
    s->bcost = bc >> 3;
    s->bmv = b;
    square_refine(s);
}

/* radius-1 diamond with the step packed in the low 4 bits of the cost:
 * source/encoder/motion.cpp:1016-1039 */
static void dia_search(srch* s, int merange)
{
    int bc = s->bcost << 4;
    orc_mv b = s->bmv;
    int i = merange;
    do
    {
        int c0 = fpel_cost(s, b.x, b.y - 1);
        int c1 = fpel_cost(s, b.x, b.y + 1);
        int c2 = fpel_cost(s, b.x - 1, b.y);
        int c3 = fpel_cost(s, b.x + 1, b.y);
        if (y_ok(s, b.y - 1))
            bc = imin(bc, (c0 << 4) + 1);
        if (y_ok(s, b.y + 1))
            bc = imin(bc, (c1 << 4) + 3);
        bc = imin(bc, (c2 << 4) + 4);
        bc = imin(bc, (c3 << 4) + 12);
        if (!(bc & 15))
            break;
        b.x -= (int32_t)((uint32_t)bc << 28) >> 30;
        b.y -= (int32_t)((uint32_t)bc << 30) >> 30;
        bc &= ~15;
    }
    while (--i && in_range(b, s->mvmin, s->mvmax));
    s->bcost = bc >> 4;
    s->bmv = b;
}

/* CROSS macro, source/encoder/motion.cpp:361-385 */
static void umh_cross(srch* s, orc_mv o, int start, int x_max, int y_max)
{
    int16_t i = (int16_t)start;
    if (x_max <= imin(s->mvmax.x - o.x, o.x - s->mvmin.x))
        for (; i < x_max - 2; i += 4)
        {
            int off[8] = { i, 0, -i, 0, i + 2, 0, -i - 2, 0 };
            try_x4(s, o, off);
        }
    for (; i < x_max; i += 2)
    {
        if (o.x + i <= s->mvmax.x)
            try_mv(s, o.x + i, o.y);
        if (o.x - i >= s->mvmin.x)
            try_mv(s, o.x - i, o.y);
    }
    i = (int16_t)start;
    if (y_max <= imin(s->mvmax.y - o.y, o.y - s->mvmin.y))
        for (; i < y_max - 2; i += 4)
        {
            int off[8] = { 0, i, 0, -i, 0, i + 2, 0, -i - 2 };
            try_x4(s, o, off);
        }
    for (; i < y_max; i += 2)
    {
        if (o.y + i <= s->mvmax.y)
            try_mv(s, o.x, o.y + i);
        if (o.y - i >= s->mvmin.y)
            try_mv(s, o.x, o.y - i);
    }
}

/* uneven multi-hexagon: source/encoder/motion.cpp:1142-1326.  Returns 1 when
 * the search must continue with the hexagon walk. */
static int umh_search(srch* s, orc_mv pmvF, orc_mv qmvp, int numCandidates, const orc_mv* mvc, int merange)
{
    static const int dia1[8] = { 0, -1, 0, 1, -1, 0, 1, 0 };
    const int scale = (s->me->bh * s->me->bh) >> 4;   /* sizeScale[], :123-153 */
#define UMH_SAD_THRESH(v) (s->bcost < (((v) >> 4) * scale))
    int ucost1, ucost2;
    int cross_start = 1;
    orc_mv o;

    ucost1 = s->bcost;
    try_x4(s, pmvF, dia1);
    if (pmvF.x | pmvF.y)
    {
        orc_mv z = { 0, 0 };
        try_x4(s, z, dia1);
    }
    ucost2 = s->bcost;
    if ((s->bmv.x | s->bmv.y) && (s->bmv.x != pmvF.x || s->bmv.y != pmvF.y))
        try_x4(s, s->bmv, dia1);
    if (s->bcost == ucost2)
        cross_start = 3;

    o = s->bmv;
    if (s->bcost == ucost2 && UMH_SAD_THRESH(2000))
    {
        static const int oct_a[8] = { 0, -2, -1, -1, 1, -1, -2, 0 };
        static const int oct_b[8] = { 2, 0, -1, 1, 1, 1, 0, 2 };
        try_x4(s, o, oct_a);
        try_x4(s, o, oct_b);
        if (s->bcost == ucost1 && UMH_SAD_THRESH(500))
            return 0;
        if (s->bcost == ucost2)
        {
            static const int ring_a[8] = { -1, -2, 1, -2, -2, -1, 2, -1 };
            static const int ring_b[8] = { -2, 1, 2, 1, -1, 2, 1, 2 };
            int range = (int16_t)((merange >> 1) | 1);
            umh_cross(s, o, 3, range, range);
            try_x4(s, o, ring_a);
            try_x4(s, o, ring_b);
            if (s->bcost == ucost2)
                return 0;
            cross_start = range + 2;
        }
    }

    if (numCandidates)
    {
        static const uint8_t range_mul[4][4] = { { 3, 3, 4, 4 }, { 3, 4, 4, 4 }, { 4, 4, 4, 5 }, { 4, 4, 5, 6 } };
        int mvd, denom = 1;
        const int is64 = s->me->bw == 64 && s->me->bh == 64;
        if (numCandidates == 1)
            mvd = is64 ? 25 : abs(qmvp.x - mvc[0].x) + abs(qmvp.y - mvc[0].y);
        else
        {
            denom = numCandidates - 1;
            mvd = 0;
            if (!is64)
            {
                mvd = abs(qmvp.x - mvc[0].x) + abs(qmvp.y - mvc[0].y);
                denom++;
            }
            for (int i = 0; i < numCandidates - 1; i++)
                mvd += abs(mvc[i].x - mvc[i + 1].x) + abs(mvc[i].y - mvc[i + 1].y);
        }
        int sad_ctx = UMH_SAD_THRESH(1000) ? 0 : UMH_SAD_THRESH(2000) ? 1 : UMH_SAD_THRESH(4000) ? 2 : 3;
        int mvd_ctx = mvd < 10 * denom ? 0 : mvd < 20 * denom ? 1 : mvd < 40 * denom ? 2 : 3;
        merange = (merange * range_mul[mvd_ctx][sad_ctx]) >> 2;
    }

    /* the cross stays centred on the early-termination origin even if the
     * steps above found a better vector (acknowledged in the reference, :1238) */
    umh_cross(s, o, cross_start, merange, merange >> 1);
    {
        static const int corners[8] = { -2, -2, -2, 2, 2, -2, 2, 2 };
        try_x4(s, o, corners);
    }

    /* hexagon grid of growing radius */
    o = s->bmv;
    uint16_t i = 1;
    do
    {
        if (4 * i > imin(imin(s->mvmax.x - o.x, o.x - s->mvmin.x), imin(s->mvmax.y - o.y, o.y - s->mvmin.y)))
        {
            for (int j = 0; j < 16; j++)
            {
                orc_mv m = { o.x + k_hex4[j].x * i, o.y + k_hex4[j].y * i };
                if (in_range(m, s->mvmin, s->mvmax))
                    try_mv(s, m.x, m.y);
            }
        }
        else
        {
            int dir = 0;
            int c[16];
            for (int k = 0; k < 16; k++)
                c[k] = fpel_sad(s, o.x + k_hex4[k].x * i, o.y + k_hex4[k].y * i) +
                       s->cx[(o.x + k_hex4[k].x * i) * 4] + s->cy[(o.y + k_hex4[k].y * i) * 4];
            for (int k = 0; k < 16; k++)
            {
                /* the row test uses the unscaled pattern offset (MIN_MV, :1274) */
                int dy = k_hex4[k].y;
                if (y_ok(s, o.y + dy) && c[k] < s->bcost)
                {
                    s->bcost = c[k];
                    dir = k_hex4[k].x * 16 + (dy & 15);
                }
            }
            if (dir)
            {
                s->bmv.x = o.x + i * (dir >> 4);
                s->bmv.y = o.y + i * ((int32_t)((uint32_t)dir << 28) >> 28);
            }
        }
    }
    while (++i <= merange >> 2);
    return in_range(s->bmv, s->mvmin, s->mvmax);
#undef UMH_SAD_THRESH
}

typedef struct { int point, dist; } star_best;

/* COST_MV_PT_DIST, source/encoder/motion.cpp:249-261 */
static void star_try(srch* s, star_best* sb, int mx, int my, int point, int dist)
{
    int c = fpel_cost(s, mx, my);
    if (c < s->bcost)
    {
        s->bcost = c;
        s->bmv.x = mx;
        s->bmv.y = my;
        sb->point = point;
        sb->dist = dist;
    }
}

/* StarPatternSearch, source/encoder/motion.cpp:387-629 */
static void star_pattern(srch* s, star_best* sb, int earlyExitIters, int merange)
{
    const orc_mv o = s->bmv;
    const orc_mv lo = s->mvmin, hi = s->mvmax;
    int saved = s->bcost;
    int rounds = 0;

    {
        const int dist = 1;
        int top = o.y - dist, bottom = o.y + dist, left = o.x - dist, right = o.x + dist;
        if (top >= lo.y && left >= lo.x && right <= hi.x && bottom <= hi.y)
        {
            star_try(s, sb, o.x, top, 2, dist);
            star_try(s, sb, left, o.y, 4, dist);
            star_try(s, sb, right, o.y, 5, dist);
            star_try(s, sb, o.x, bottom, 7, dist);
        }
        else
        {
            if (top >= lo.y) star_try(s, sb, o.x, top, 2, dist);
            if (left >= lo.x) star_try(s, sb, left, o.y, 4, dist);
            if (right <= hi.x) star_try(s, sb, right, o.y, 5, dist);
            if (bottom <= hi.y) star_try(s, sb, o.x, bottom, 7, dist);
        }
        if (s->bcost < saved)
            rounds = 0;
        else if (++rounds >= earlyExitIters)
            return;
    }

    for (int dist = 2; dist <= 8; dist <<= 1)
    {
        int top = o.y - dist, bottom = o.y + dist, left = o.x - dist, right = o.x + dist;
        int h = dist >> 1;
        int top2 = o.y - h, bottom2 = o.y + h, left2 = o.x - h, right2 = o.x + h;
        saved = s->bcost;
        if (top >= lo.y && left >= lo.x && right <= hi.x && bottom <= hi.y)
        {
            star_try(s, sb, o.x, top, 2, dist);
            star_try(s, sb, left2, top2, 1, h);
            star_try(s, sb, right2, top2, 3, h);
            star_try(s, sb, left, o.y, 4, dist);
            star_try(s, sb, right, o.y, 5, dist);
            star_try(s, sb, left2, bottom2, 6, h);
            star_try(s, sb, right2, bottom2, 8, h);
            star_try(s, sb, o.x, bottom, 7, dist);
        }
        else
        {
            if (top >= lo.y) star_try(s, sb, o.x, top, 2, dist);
            if (top2 >= lo.y)
            {
                if (left2 >= lo.x) star_try(s, sb, left2, top2, 1, h);
                if (right2 <= hi.x) star_try(s, sb, right2, top2, 3, h);
            }
            if (left >= lo.x) star_try(s, sb, left, o.y, 4, dist);
            if (right <= hi.x) star_try(s, sb, right, o.y, 5, dist);
            if (bottom2 <= hi.y)
            {
                if (left2 >= lo.x) star_try(s, sb, left2, bottom2, 6, h);
                if (right2 <= hi.x) star_try(s, sb, right2, bottom2, 8, h);
            }
            if (bottom <= hi.y) star_try(s, sb, o.x, bottom, 7, dist);
        }
        if (s->bcost < saved)
            rounds = 0;
        else if (++rounds >= earlyExitIters)
            return;
    }

    for (int dist = 16; dist <= (int16_t)merange; dist <<= 1)
    {
        int top = o.y - dist, bottom = o.y + dist, left = o.x - dist, right = o.x + dist;
        int q = dist >> 2;
        saved = s->bcost;
        if (top >= lo.y && left >= lo.x && right <= hi.x && bottom <= hi.y)
        {
            star_try(s, sb, o.x, top, 0, dist);
            star_try(s, sb, left, o.y, 0, dist);
            star_try(s, sb, right, o.y, 0, dist);
            star_try(s, sb, o.x, bottom, 0, dist);
            for (int idx = 1; idx < 4; idx++)
            {
                int yT = top + q * idx, yB = bottom - q * idx, xL = o.x - q * idx, xR = o.x + q * idx;
                star_try(s, sb, xL, yT, 0, dist);
                star_try(s, sb, xR, yT, 0, dist);
                star_try(s, sb, xL, yB, 0, dist);
                star_try(s, sb, xR, yB, 0, dist);
            }
        }
        else
        {
            if (top >= lo.y) star_try(s, sb, o.x, top, 0, dist);
            if (left >= lo.x) star_try(s, sb, left, o.y, 0, dist);
            if (right <= hi.x) star_try(s, sb, right, o.y, 0, dist);
            if (bottom <= hi.y) star_try(s, sb, o.x, bottom, 0, dist);
            for (int idx = 1; idx < 4; idx++)
            {
                int yT = top + q * idx, yB = bottom - q * idx, xL = o.x - q * idx, xR = o.x + q * idx;
                if (yT >= lo.y)
                {
                    if (xL >= lo.x) star_try(s, sb, xL, yT, 0, dist);
                    if (xR <= hi.x) star_try(s, sb, xR, yT, 0, dist);
                }
                if (yB <= hi.y)
                {
                    if (xL >= lo.x) star_try(s, sb, xL, yB, 0, dist);
                    if (xR <= hi.x) star_try(s, sb, xR, yB, 0, dist);
                }
            }
        }
        if (s->bcost < saved)
            rounds = 0;
        else if (++rounds >= earlyExitIters)
            return;
    }
}

/* star search driver: source/encoder/motion.cpp:1328-1436 */
static void star_search(srch* s, int merange)
{
    star_best sb = { 0, 0 };
    star_pattern(s, &sb, 3, merange);
    if (sb.dist == 1)
    {
        if (!sb.point)
            return;
        int saved = s->bcost;
        orc_mv m1 = { s->bmv.x + k_twopoint[(sb.point - 1) * 2].x, s->bmv.y + k_twopoint[(sb.point - 1) * 2].y };
        orc_mv m2 = { s->bmv.x + k_twopoint[(sb.point - 1) * 2 + 1].x, s->bmv.y + k_twopoint[(sb.point - 1) * 2 + 1].y };
        if (in_range(m1, s->mvmin, s->mvmax)) try_mv(s, m1.x, m1.y);
        if (in_range(m2, s->mvmin, s->mvmax)) try_mv(s, m2.x, m2.y);
        if (s->bcost == saved)
            return;
    }

    const int raster = 5;
    if (sb.dist > raster)
    {
        g_inRaster = 1;
        for (int ty = s->mvmin.y; ty <= s->mvmax.y; ty += raster)
            for (int tx = s->mvmin.x; tx <= s->mvmax.x; tx += raster)
            {
                if (tx + raster * 3 <= s->mvmax.x)
                {
                    int c[4];
                    for (int k = 0; k < 4; k++)
                        c[k] = fpel_sad(s, tx + raster * k, ty);
                    for (int k = 0; k < 4; k++)
                    {
                        int mx = tx + raster * k;
                        /* the fourth point is priced with the vector scaled by 8
                         * instead of 4 (source/encoder/motion.cpp:1392) */
                        int mc = k < 3 ? mvcost_q(s, mx * 4, ty * 4) : mvcost_q(s, mx * 8, ty * 8);
                        if (c[k] + mc < s->bcost)
                        {
                            s->bcost = c[k] + mc;
                            s->bmv.x = mx;
                            s->bmv.y = ty;
                        }
                    }
                    tx += raster * 3;
                }
                else
                    try_mv(s, tx, ty);
            }
        g_inRaster = 0;
    }

    while (sb.dist > 0)
    {
        sb.dist = 0;
        sb.point = 0;
        star_pattern(s, &sb, 32, merange);
        if (sb.dist == 1)
        {
            if (!sb.point)
                break;
            orc_mv m1 = { s->bmv.x + k_twopoint[(sb.point - 1) * 2].x, s->bmv.y + k_twopoint[(sb.point - 1) * 2].y };
            orc_mv m2 = { s->bmv.x + k_twopoint[(sb.point - 1) * 2 + 1].x, s->bmv.y + k_twopoint[(sb.point - 1) * 2 + 1].y };
            if (in_range(m1, s->mvmin, s->mvmax)) try_mv(s, m1.x, m1.y);
            if (in_range(m2, s->mvmin, s->mvmax)) try_mv(s, m2.x, m2.y);
            break;
        }
    }
}

/* exhaustive search: source/encoder/motion.cpp:1593-1637 */
static void full_search(srch* s, int merange)
{
    int y0 = s->mvmin.y, x0 = s->mvmin.x, y1 = s->mvmax.y, x1 = s->mvmax.x;
    if (s->ref->isHMELowres)
    {
        merange = merange < 0 ? -merange : merange;
        y0 = imax(y0, -merange);
        x0 = imax(x0, -merange);
        y1 = imin(y1, merange);
        x1 = imin(x1, merange);
    }
    for (int ty = y0; ty <= y1; ty++)
        for (int tx = x0; tx <= x1; tx++)
            try_mv(s, tx, ty);
}

/* ---- successive elimination (X265_SEA) ---- */

/* integral_init{4,8,12,16,24,32}h_c, source/encoder/framefilter.cpp:39-104: running sum of w pixels plus the
 * entry one row up */
static void integral_inith(int w, uint32_t* sum, const pixel* pix, intptr_t stride)
{
    int32_t v = 0;
    for (int k = 0; k < w; k++)
        v += pix[k];
    for (int16_t x = 0; x < stride - w; x++)
    {
        sum[x] = v + sum[x - stride];
        v += pix[x + w] - pix[x];
    }
}

/* integral_init{4..32}v_c, framefilter.cpp:106-140 */
static void integral_initv(int h, uint32_t* sum, intptr_t stride)
{
    for (int x = 0; x < stride; x++)
        sum[x] = sum[x + h * stride] - sum[x];
}

static const int k_integralW[12] = { 32, 32, 32, 24, 16, 16, 16, 12, 8, 8, 4, 4 };
static const int k_integralH[12] = { 32, 24, 8, 32, 16, 12, 4, 16, 32, 8, 16, 4 };

/* FrameFilter::computeMEIntegral over every row of the padded picture at once: framefilter.cpp:760-828 */
void orc_sea_integral(const pixel* picOrg, intptr_t stride, int padX, int padY, int height, uint32_t* const planes[12])
{
    for (int i = 0; i < 12; i++)
        memset(planes[i] - padY * stride - padX, 0, stride * sizeof(uint32_t));
    const int last = height + padY - 1;
    for (int y = -padY; y < last; y++)
    {
        const pixel* pix = picOrg + y * stride - padX;
        for (int i = 0; i < 12; i++)
        {
            uint32_t* sum = planes[i] + (y + 1) * stride - padX;
            integral_inith(k_integralW[i], sum, pix, stride);
            if (y >= k_integralH[i] - padY)
                integral_initv(k_integralH[i], sum - k_integralH[i] * stride, stride);
        }
    }
}

void orc_me_set_integral(orc_me* me, uint32_t* const planes[12])
{
    for (int i = 0; i < 12; i++)
        me->integral[i] = planes[i];
}

/* ads_x4 / ads_x2 / ads_x1, source/common/pixel.cpp:121-165; lx is the PU width */
static int ads_n(int n, int lx, const int encDC[4], const uint32_t* sums, int delta, const uint16_t* costMvX,
                 int16_t* mvs, int width, int thresh)
{
    int nmv = 0;
    for (int16_t i = 0; i < width; i++, sums++)
    {
        int ads;
        if (n == 4)
            ads = (int)labs(encDC[0] - (long)sums[0]) + (int)labs(encDC[1] - (long)sums[lx >> 1]) +
                  (int)labs(encDC[2] - (long)sums[delta]) + (int)labs(encDC[3] - (long)sums[delta + (lx >> 1)]) + costMvX[i];
        else if (n == 2)
            ads = (int)labs(encDC[0] - (long)sums[0]) + (int)labs(encDC[1] - (long)sums[delta]) + costMvX[i];
        else
            ads = (int)labs(encDC[0] - (long)sums[0]) + costMvX[i];
        if (ads < thresh)
            mvs[nmv++] = i;
    }
    return nmv;
}

static int is_size(const orc_me* me, int w, int h) { return me->bw == w && me->bh == h; }

/* which ads primitive setSourcePU picks for the partition: source/common/pixel.cpp:1122-1146 */
static int ads_arity(const orc_me* me)
{
    static const int four[][2] = { {16,16}, {32,32}, {64,64}, {32,24}, {24,32}, {32,8}, {8,32}, {64,48}, {48,64}, {64,16}, {16,64} };
    static const int one[][2] = { {4,4}, {8,8}, {16,12}, {12,16}, {16,4}, {4,16} };
    for (size_t i = 0; i < sizeof(four) / sizeof(four[0]); i++)
        if (is_size(me, four[i][0], four[i][1]))
            return 4;
    for (size_t i = 0; i < sizeof(one) / sizeof(one[0]); i++)
        if (is_size(me, one[i][0], one[i][1]))
            return 1;
    return 2;
}

/* sum of a tw x th block of the staged encode block = sad_x4 against a zero block, motion.cpp:1501-1507 */
static int fenc_sum(const orc_me* me, int x, int y, int tw, int th)
{
    int s = 0;
    for (int r = 0; r < th; r++)
        for (int c = 0; c < tw; c++)
            s += me->fenc[(y + r) * ORC_FENC_STRIDE + x + c];
    return s;
}

/* Successive elimination: source/encoder/motion.cpp:1438-1591.  The DC of up to four sub-blocks is compared with
 * block sums of the reference before a candidate's SAD is taken.  Four partitions (8x4, 4x8, 32x8, 8x32) make the
 * reference compare DC terms taken over encode-buffer bytes outside the PU, left there by earlier PUs: those have
 * no defined result. */
static void sea_search(srch* s, int merange)
{
    orc_me* me = s->me;
    const intptr_t stride = s->stride;
    const orc_mv omv = s->bmv;
    const int minX = imax(omv.x - merange, s->mvmin.x), minY = imax(omv.y - merange, s->mvmin.y);
    const int maxX = imin(omv.x + merange, s->mvmax.x), maxY = imin(omv.y + merange, s->mvmax.y);
    const uint16_t* p_cost_mvx = s->cx;
    const uint16_t* p_cost_mvy = s->cy;
    int16_t* scratch = (int16_t*)calloc(merange * 2 + 4, sizeof(int16_t));
    const int meRangeWidth = (maxX - minX + 3) & ~3;
    const int w = me->bw, h = me->bh;
    if (is_size(me, 8, 4) || is_size(me, 4, 8) || is_size(me, 32, 8) || is_size(me, 8, 32))
        abort();
    int deltaX = w <= 8 ? w : w >> 1;
    int deltaY = h <= 8 ? h : h >> 1;
    const int smallRect = is_size(me, 4, 4) || is_size(me, 16, 12) || is_size(me, 12, 16) || is_size(me, 16, 4) || is_size(me, 4, 16);
    const int verticalRect = is_size(me, 32, 64) || is_size(me, 16, 32) || is_size(me, 8, 16) || is_size(me, 4, 8);
    const int horizontalRect = is_size(me, 64, 32) || is_size(me, 32, 16) || is_size(me, 16, 8) || is_size(me, 8, 4);
    const int asymVertical = is_size(me, 12, 16) || is_size(me, 4, 16) || is_size(me, 24, 32) || is_size(me, 8, 32) ||
                             is_size(me, 48, 64) || is_size(me, 16, 64);
    const int asymHorizontal = is_size(me, 16, 12) || is_size(me, 16, 4) || is_size(me, 32, 24) || is_size(me, 32, 8) ||
                               is_size(me, 64, 48) || is_size(me, 64, 16);
    int tw, th;             /* the partition the four DC terms are taken over */
    if (verticalRect)
        tw = w, th = h >> 1;
    else if (horizontalRect)
        tw = w >> 1, th = h;
    else if (asymVertical || asymHorizontal)
        tw = smallRect ? w : w >> 1, th = smallRect ? h : h >> 1;
    else
        tw = w <= 8 ? w : w >> 1, th = w <= 8 ? h : h >> 1;
    /* the four terms exactly where sad_x4 reads them, inside the 64x64 staging buffer; terms the partition's ads
     * does not use may cover bytes outside the PU */
    int encDC[4];
    encDC[0] = fenc_sum(me, 0, 0, tw, th);
    encDC[1] = deltaX + tw <= 64 ? fenc_sum(me, deltaX, 0, tw, th) : 0;
    encDC[2] = deltaY + th <= 64 ? fenc_sum(me, 0, deltaY, tw, th) : 0;
    encDC[3] = deltaX + tw <= 64 && deltaY + th <= 64 ? fenc_sum(me, deltaX, deltaY, tw, th) : 0;

    int plane;
    switch (deltaX)
    {
    case 32: plane = deltaY % 24 == 0 ? 1 : deltaY == 8 ? 2 : 0; break;
    case 24: plane = 3; break;
    case 16: plane = deltaY % 12 == 0 ? 5 : deltaY == 4 ? 6 : 4; break;
    case 12: plane = 7; break;
    case 8:  plane = deltaY == 32 ? 8 : 9; break;
    case 4:  plane = deltaY == 16 ? 10 : 11; break;
    default: plane = 11; break;
    }
    const uint32_t* sumsBase = me->integral[plane] + me->blockOffset;

    const int scaled = is_size(me, 64, 64) || is_size(me, 32, 32) || is_size(me, 16, 16) || is_size(me, 32, 64) ||
                       is_size(me, 16, 32) || is_size(me, 8, 16) || is_size(me, 4, 8) || is_size(me, 12, 16) ||
                       is_size(me, 4, 16) || is_size(me, 24, 32) || is_size(me, 8, 32) || is_size(me, 48, 64) ||
                       is_size(me, 16, 64);
    if (scaled)
        deltaY *= (int)stride;
    if (verticalRect)
        encDC[1] = encDC[2];
    if (horizontalRect)
        deltaY = deltaX;

    /* fpelCostMvX + minX (motion.cpp:1460, :1574): m_fpelMvCosts[j][i] = cost[4 * i + j] (bitcost.cpp:70-73), so
     * entry k of m_fpelMvCosts[-qmvp.x & 3] + (-qmvp.x >> 2) is cost[4 * k - qmvp.x] */
    uint16_t* costMvX = (uint16_t*)malloc((meRangeWidth + 1) * sizeof(uint16_t));
    for (int i = 0; i < meRangeWidth; i++)
        costMvX[i] = p_cost_mvx[4 * (minX + i)];

    const int arity = ads_arity(me);
    for (int ty = minY; ty <= maxY; ty++)
    {
        int i;
        const int ycost = p_cost_mvy[ty] << 2;       /* indexed by the full-pel row, as the reference does */
        if (s->bcost <= ycost)
            continue;
        s->bcost -= ycost;
        const int xn = ads_n(arity, w, encDC, sumsBase + minX + ty * stride, deltaY, costMvX, scratch, meRangeWidth, s->bcost);
        /* COST_MV_X3_ABS (motion.cpp:325-338): SAD plus the x cost only, against the reduced bcost */
        for (i = 0; i < xn - 2; i += 3)
        {
            int costs[3];
            for (int k = 0; k < 3; k++)
                costs[k] = fpel_sad(s, minX + scratch[i + k], ty) + p_cost_mvx[(minX + scratch[i + k]) * 4];
            for (int k = 0; k < 3; k++)
                if (costs[k] < s->bcost)
                {
                    s->bcost = costs[k];
                    s->bmv.x = minX + scratch[i + k];
                    s->bmv.y = ty;
                }
        }
        s->bcost += ycost;
        for (; i < xn; i++)
            try_mv(s, minX + scratch[i], ty);
    }
    free(costMvX);
    free(scratch);
}

/* the elimination search alone, from a given start (tests) */
int orc_sea_search_only(orc_me* me, const orc_refplanes* ref, orc_mv mvmin, orc_mv mvmax, orc_mv qmvp, int merange,
                        orc_mv* bmv, int bcost)
{
    srch S;
    memset(&S, 0, sizeof(S));
    S.me = me;
    S.ref = ref;
    S.stride = ref->stride;
    S.fref = ref->plane[0] + me->blockOffset;
    S.mvmin = mvmin;
    S.mvmax = mvmax;
    me->mvp = qmvp;
    S.cx = me->cost - qmvp.x;
    S.cy = me->cost - qmvp.y;
    S.bmv = *bmv;
    S.bcost = bcost;
    sea_search(&S, merange);
    *bmv = S.bmv;
    return S.bcost;
}

/* MotionEstimate::subpelCompare for luma: source/encoder/motion.cpp:1775-1803.
 * Always addresses fpelPlane[0] with the full lumaStride, also when the search
 * itself ran on the quarter-res planes. */
static int subpel_compare(srch* s, orc_mv qmv, int useSatd)
{
    orc_me* me = s->me;
    intptr_t rs = s->ref->stride;
    const pixel* fref = s->ref->plane[0] + me->blockOffset + (qmv.x >> 2) + (qmv.y >> 2) * rs;
    int xf = qmv.x & 3, yf = qmv.y & 3;
    orc_me_counters[2]++;
    if (!(xf | yf))
        return useSatd ? cmp_satd(me, fref, rs) : cmp_sad(me, fref, rs);
    pixel buf[64 * 64];
    if (!yf)
        orc_luma_hpp(me->bw, me->bh, fref, rs, buf, me->bw, xf);
    else if (!xf)
        orc_luma_vpp(me->bw, me->bh, fref, rs, buf, me->bw, yf);
    else
        orc_luma_hvpp(me->bw, me->bh, fref, rs, buf, me->bw, xf, yf);
    return useSatd ? cmp_satd(me, buf, me->bw) : cmp_sad(me, buf, me->bw);
}

/* One AMVP candidate's cost in Search::selectMVP (source/encoder/search.cpp:2372-2379): the PU is predicted
 * with Predict::predInterLumaPixel (source/common/predict.cpp:279-300: a copy, luma_hpp, luma_vpp or luma_hvpp
 * by the vector's fractions) and compared by SAD, which is subpelCompare's arithmetic without the MV cost. */
int orc_me_pred_sad(orc_me* me, const orc_refplanes* ref, orc_mv qmv)
{
    srch S;
    memset(&S, 0, sizeof(S));
    S.me = me;
    S.ref = ref;
    return subpel_compare(&S, qmv, 0);
}

/* Predict::predInterLumaPixel (source/common/predict.cpp:279-300) into a tight bw x bh buffer */
static void pred_luma_pixel(const orc_me* me, const orc_refplanes* ref, orc_mv qmv, pixel* dst)
{
    const intptr_t rs = ref->stride;
    const pixel* src = ref->plane[0] + me->blockOffset + (qmv.x >> 2) + (qmv.y >> 2) * rs;
    const int xf = qmv.x & 3, yf = qmv.y & 3;
    if (!(xf | yf))
        for (int y = 0; y < me->bh; y++)
            memcpy(dst + y * me->bw, src + y * rs, me->bw * sizeof(pixel));
    else if (!yf)
        orc_luma_hpp(me->bw, me->bh, src, rs, dst, me->bw, xf);
    else if (!xf)
        orc_luma_vpp(me->bw, me->bh, src, rs, dst, me->bw, yf);
    else
        orc_luma_hvpp(me->bw, me->bh, src, rs, dst, me->bw, xf, yf);
}

/* The bidirectional candidate of Search::puMotionEstimation (source/encoder/search.cpp:440-448): both predictions,
 * pixelavg_pp with weight 32, bufSATD against the source PU.  With zero vectors this is also :482-486, which
 * averages the two reference blocks in place. */
int orc_me_bipred_satd(orc_me* me, const orc_refplanes* ref0, orc_mv qmv0, const orc_refplanes* ref1, orc_mv qmv1)
{
    pixel p0[64 * 64], p1[64 * 64], avg[64 * 64];
    pred_luma_pixel(me, ref0, qmv0, p0);
    pred_luma_pixel(me, ref1, qmv1, p1);
    orc_pixelavg_pp(me->bw, me->bh, avg, me->bw, p0, me->bw, p1, me->bw);
    return cmp_satd(me, avg, me->bw);
}

/* MotionEstimate::diamondSearch, source/encoder/motion.cpp:631-788: the coarse
 * search Search::puMotionEstimation runs for an area's starting vector
 * (isMVP, source/encoder/search.cpp:360-364).  Full-pel in, full-pel out.
 *
 * COST_MV_X4 in this fork adds its offsets to omv (motion.cpp:302-323), while
 * diamondSearch hands it absolute coordinates: inside the range the measured
 * position is therefore omv + (the coordinate written at the call site), and
 * that sum is also what bmv receives.  try_x4() has the macro's semantics, so
 * the call sites below pass the same arguments as the reference's. */
int orc_diamond_search(orc_me* me, const orc_refplanes* ref, orc_mv mvmin, orc_mv mvmax, orc_mv qmvp, orc_mv* outMV)
{
    srch S;
    srch* s = &S;
    s->me = me;
    s->ref = ref;
    s->hme = 0;
    s->stride = ref->stride;
    s->fref = ref->plane[0] + me->blockOffset;
    s->mvmin = mvmin;
    s->mvmax = mvmax;
    me->mvp = qmvp;
    s->cx = me->cost - qmvp.x;
    s->cy = me->cost - qmvp.y;
    s->bcost = INT_MAX;
    s->bmv.x = s->bmv.y = 0;
    orc_mv omv = s->bmv;

    for (int dist = 1; dist <= 4; dist <<= 1)
    {
        const orc_mv bmv0 = s->bmv;
        const int top = omv.y - dist, bottom = omv.y + dist, left = omv.x - dist, right = omv.x + dist;
        const int top2 = omv.y - (dist >> 1), bottom2 = omv.y + (dist >> 1);
        const int left2 = omv.x - (dist >> 1), right2 = omv.x + (dist >> 1);
        if (top >= mvmin.y && left >= mvmin.x && right <= mvmax.x && bottom <= mvmax.y)
        {
            const int a[8] = { omv.x, top, omv.x, bottom, left, omv.y, right, omv.y };
            const int b[8] = { left2, top2, right2, top2, left2, bottom2, right2, bottom2 };
            try_x4(s, omv, a);
            try_x4(s, omv, b);
        }
        else
        {
            if (top >= mvmin.y)
                try_mv(s, omv.x, top);
            if (top2 >= mvmin.y)
            {
                if (left2 >= mvmin.x)
                    try_mv(s, left2, top2);
                if (right2 <= mvmax.x)
                    try_mv(s, right2, top2);
            }
            if (left >= mvmin.x)
                try_mv(s, left, omv.y);
            if (right <= mvmax.x)
                try_mv(s, right, omv.y);
            if (bottom2 <= mvmax.y)
            {
                if (left2 >= mvmin.x)
                    try_mv(s, left2, bottom2);
                if (right2 <= mvmax.x)
                    try_mv(s, right2, bottom2);
            }
            if (bottom <= mvmax.y)
                try_mv(s, omv.x, bottom);
        }
        if (s->bmv.x == bmv0.x && s->bmv.y == bmv0.y)
            break;
    }

    omv = s->bmv;
    for (int dist = 8; dist <= 64; dist += 8)
    {
        const orc_mv bmv0 = s->bmv;
        const int top = omv.y - dist, bottom = omv.y + dist, left = omv.x - dist, right = omv.x + dist;
        const int inside = top >= mvmin.y && left >= mvmin.x && right <= mvmax.x && bottom <= mvmax.y;
        if (inside)
        {
            const int a[8] = { omv.x, top, left, omv.y, right, omv.y, omv.x, bottom };
            try_x4(s, omv, a);
        }
        else
        {
            if (top >= mvmin.y)
                try_mv(s, omv.x, top);
            if (left >= mvmin.x)
                try_mv(s, left, omv.y);
            if (right <= mvmax.x)
                try_mv(s, right, omv.y);
            if (bottom <= mvmax.y)
                try_mv(s, omv.x, bottom);
        }
        for (int index = 1; index < 4; index++)
        {
            const int posYT = top + (dist >> 2) * index, posYB = bottom - (dist >> 2) * index;
            const int posXL = omv.x - (dist >> 2) * index, posXR = omv.x + (dist >> 2) * index;
            if (inside)
            {
                const int a[8] = { posXL, posYT, posXR, posYT, posXL, posYB, posXR, posYB };
                try_x4(s, omv, a);
                continue;
            }
            if (posYT >= mvmin.y)
            {
                if (posXL >= mvmin.x)
                    try_mv(s, posXL, posYT);
                if (posXR <= mvmax.x)
                    try_mv(s, posXR, posYT);
            }
            if (posYB <= mvmax.y)
            {
                if (posXL >= mvmin.x)
                    try_mv(s, posXL, posYB);
                if (posXR <= mvmax.x)
                    try_mv(s, posXR, posYB);
            }
        }
        if (s->bmv.x == bmv0.x && s->bmv.y == bmv0.y)
            break;
        omv = s->bmv;
    }
    *outMV = s->bmv;
    return s->bcost;
}

int orc_motion_estimate(orc_me* me, const orc_refplanes* ref, orc_mv mvmin, orc_mv mvmax, orc_mv qmvp,
                        int numCandidates, const orc_mv* mvc, int merange, orc_mv* outQMv,
                        uint32_t maxSlices, int hmePlane)
{
    srch S;
    srch* s = &S;
    const int hme = hmePlane;
    s->me = me;
    s->ref = ref;
    s->hme = hme;
    s->stride = hme ? ref->stride / 2 : ref->stride;
    s->fref = (hme ? ref->lowerPlane[0] : ref->plane[0]) + me->blockOffset;
    s->mvmin = mvmin;
    s->mvmax = mvmax;
    me->mvp = qmvp;
    s->cx = me->cost - qmvp.x;
    s->cy = me->cost - qmvp.y;

    const orc_mv qmin = { mvmin.x * 4, mvmin.y * 4 }, qmax = { mvmax.x * 4, mvmax.y * 4 };

    /* clipped quarter-pel predictor, measured by SAD (no MV cost on this term) */
    orc_mv pmv = { imax(imin(qmvp.x, qmax.x), qmin.x), imax(imin(qmvp.y, qmax.y), qmin.y) };
    if (me->vertRestriction && pmv.y > mvmax.y * 4)       /* motion.cpp:957-963 */
        pmv.y = mvmax.y * 4;
    orc_mv bestpre = pmv;
    int bprecost = ref->isLowres ? orc_lowres_qpel_cost(me, ref, me->blockOffset, pmv, 0, hme)
                                 : subpel_compare(s, pmv, 0);

    /* start point: predictor rounded to full pel */
    s->bmv.x = (pmv.x + 2) >> 2;
    s->bmv.y = (pmv.y + 2) >> 2;
    s->bcost = bprecost;
    if ((pmv.x | pmv.y) & 3)
        s->bcost = fpel_cost(s, s->bmv.x, s->bmv.y);

    /* zero vector, when the predictor is not zero */
    if (pmv.x | pmv.y)
    {
        int c = cmp_sad(me, s->fref, s->stride) + mvcost_q(s, 0, 0);
        if (c < s->bcost)
        {
            s->bcost = c;
            s->bmv.x = 0;
            s->bmv.y = imax(imin(0, mvmax.y), mvmin.y);
        }
    }

    /* extra quarter-pel candidates (never used by the lowres path) */
    for (int i = 0; i < numCandidates; i++)
    {
        orc_mv m = { imax(imin(mvc[i].x, qmax.x), qmin.x), imax(imin(mvc[i].y, qmax.y), qmin.y) };
        if ((m.x | m.y) && (m.x != pmv.x || m.y != pmv.y) && (m.x != bestpre.x || m.y != bestpre.y))
        {
            int c = subpel_compare(s, m, 0) + mvcost_q(s, m.x, m.y);
            if (c < bprecost)
            {
                bprecost = c;
                bestpre = m;
            }
        }
    }

    orc_mv pmvF = { (pmv.x + 2) >> 2, (pmv.y + 2) >> 2 };
    if (s->bcost == 0)
    {
        outQMv->x = s->bmv.x * 4;
        outQMv->y = s->bmv.y * 4;
        return mvcost_f(s, s->bmv.x, s->bmv.y);
    }

    int search = ref->isHMELowres ? (hme ? me->searchMethodL0 : me->searchMethodL1) : me->searchMethod;
    switch (search)
    {
    case ORC_DIA:
        dia_search(s, merange);
        break;
    case ORC_HEX:
        hex_search(s, merange);
        break;
    case ORC_UMH:
        if (umh_search(s, pmvF, qmvp, numCandidates, mvc, merange))
            hex_search(s, merange);
        break;
    case ORC_STAR:
        star_search(s, merange);
        break;
    case ORC_FULL:
        full_search(s, merange);
        break;
    case ORC_SEA:
        sea_search(s, merange);
        break;
    default:
        abort();
    }

    orc_mv bmv;
    int bcost = s->bcost;
    if (bprecost < bcost)
    {
        bmv = bestpre;
        bcost = bprecost;
    }
    else
    {
        bmv.x = s->bmv.x * 4;
        bmv.y = s->bmv.y * 4;
    }

    const subpel_workload* wl = &k_workload[me->subpelRefine];

    if ((maxSlices > 1) & ((bmv.y < qmin.y) | (bmv.y > qmax.y)))
    {
        bmv.y = imin(imax(bmv.y, qmin.y), qmax.y);
        bcost = subpel_compare(s, bmv, 1) + mvcost_q(s, bmv.x, bmv.y);
    }

    if (!bcost)
        bcost = mvcost_q(s, bmv.x, bmv.y);
    else if (ref->isLowres)
    {
        /* four half-pel SADs, one SATD, four quarter-pel SATDs: :1667-1699 */
        int bdir = 0;
        for (int i = 1; i <= wl->hpel_dirs; i++)
        {
            orc_mv q = { bmv.x + k_square1[i].x * 2, bmv.y + k_square1[i].y * 2 };
            if ((q.y < qmin.y) | (q.y > qmax.y))
                continue;
            int c = orc_lowres_qpel_cost(me, ref, me->blockOffset, q, 0, hme) + mvcost_q(s, q.x, q.y);
            if (c < bcost) { bcost = c; bdir = i; }
        }
        bmv.x += k_square1[bdir].x * 2;
        bmv.y += k_square1[bdir].y * 2;
        bcost = orc_lowres_qpel_cost(me, ref, me->blockOffset, bmv, 1, hme) + mvcost_q(s, bmv.x, bmv.y);
        bdir = 0;
        for (int i = 1; i <= wl->qpel_dirs; i++)
        {
            orc_mv q = { bmv.x + k_square1[i].x, bmv.y + k_square1[i].y };
            if ((q.y < qmin.y) | (q.y > qmax.y))
                continue;
            int c = orc_lowres_qpel_cost(me, ref, me->blockOffset, q, 1, hme) + mvcost_q(s, q.x, q.y);
            if (c < bcost) { bcost = c; bdir = i; }
        }
        bmv.x += k_square1[bdir].x;
        bmv.y += k_square1[bdir].y;
    }
    else
    {
        /* interpolated sub-pel refinement: :1700-1757 */
        int hpelSatd = wl->hpel_satd;
        if (hpelSatd)
            bcost = subpel_compare(s, bmv, 1) + mvcost_q(s, bmv.x, bmv.y);
        for (int iter = 0; iter < wl->hpel_iters; iter++)
        {
            int bdir = 0;
            for (int i = 1; i <= wl->hpel_dirs; i++)
            {
                orc_mv q = { bmv.x + k_square1[i].x * 2, bmv.y + k_square1[i].y * 2 };
                if ((q.y < qmin.y) | (q.y > qmax.y))
                    continue;
                int c = subpel_compare(s, q, hpelSatd) + mvcost_q(s, q.x, q.y);
                if (c < bcost) { bcost = c; bdir = i; }
            }
            if (!bdir)
                break;
            bmv.x += k_square1[bdir].x * 2;
            bmv.y += k_square1[bdir].y * 2;
        }
        if (!hpelSatd)
            bcost = subpel_compare(s, bmv, 1) + mvcost_q(s, bmv.x, bmv.y);
        for (int iter = 0; iter < wl->qpel_iters; iter++)
        {
            int bdir = 0;
            for (int i = 1; i <= wl->qpel_dirs; i++)
            {
                orc_mv q = { bmv.x + k_square1[i].x, bmv.y + k_square1[i].y };
                if ((q.y < qmin.y) | (q.y > qmax.y))
                    continue;
                int c = subpel_compare(s, q, 1) + mvcost_q(s, q.x, q.y);
                if (c < bcost) { bcost = c; bdir = i; }
            }
            if (!bdir)
                break;
            bmv.x += k_square1[bdir].x;
            bmv.y += k_square1[bdir].y;
        }
    }

    /* this fork's zero-vector override: the vector is replaced, the returned
     * cost is not (source/encoder/motion.cpp:1762-1768) */
    if (bmv.x | bmv.y)
    {
        orc_mv z = { 0, 0 };
        int c = subpel_compare(s, z, 1) + mvcost_q(s, 0, 0);
        if (c <= bcost)
            bmv = z;
    }

    *outQMv = bmv;
    return bcost;
}
