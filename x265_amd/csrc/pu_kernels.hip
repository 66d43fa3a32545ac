// Full-resolution motion estimation for batches of prediction units.
//
// Replaces MotionEstimate::motionEstimate (source/encoder/motion.cpp:923-1773)
// for a full-resolution reference: integer search (diamond, hexagon, star or
// exhaustive) on the reconstructed / source luma plane, then sub-pel refinement
// by true 8-tap interpolation (MotionEstimate::subpelCompare, motion.cpp:1775-1803,
// filters of source/common/ipfilter.cpp:79-118, 164-203, 362-369) following the
// subme workload table (motion.cpp:48-58), and the fork's closing zero-vector
// check (motion.cpp:1762-1768).  Every PU shape whose width and height are
// multiples of 4 up to 64x64 is handled.
//
// One wave per PU, or per 2 / 4 / 8 PUs of at most 8 / 4 / 2 tiles (see pu_search_kernel).  The PU is cut into tiles of 8x4 pixels (4x4 when the width
// is not a multiple of 8, which is also how the reference tiles its SATD,
// source/common/pixel.cpp:1148-1172); a quad of lanes holds the four rows of a
// tile, sixteen tiles are measured per pass and a 64x64 PU takes eight passes.
// The encode block sits in LDS in the packed column-pair form the SATD wants,
// next to each lane's pixel offset and validity mask, so one candidate costs,
// per pass, one unaligned row load, four byte permutes and four packed SADs.
// Candidates are visited one after the other in the reference's order with its
// strict '<' updates on wave-uniform scalars; the parallelism is across the
// block's pixels and across PUs.
#include "device_common.h"
#include "kernels.h"
#include "sea_shape.h"

#define XG_LDS __attribute__((address_space(3)))

namespace xg {

constexpr int PU_DIA = 0, PU_HEX = 1, PU_UMH = 2, PU_STAR = 3, PU_SEA = 4;   // x265gpu search method values (anything else: exhaustive)
constexpr int PU_AREA_DIAMOND = 16;                  // MotionEstimate::diamondSearch instead of motionEstimate

// luma interpolation taps, source/common/constants.cpp:250-256
__constant__ short k_lumaTaps[4][8] = {
    { 0, 0, 0, 64, 0, 0, 0, 0 },
    { -1, 4, -10, 58, 17, -5, 1, 0 },
    { -1, 4, -11, 40, 40, -11, 4, -1 },
    { 0, 1, -5, 17, 58, -10, 4, -1 },
};
// hex2, mod6m1, square1, offsets for the two-point refinement of the star search (motion.cpp:63-84)
__constant__ signed char k_hex2[8][2] = { {-1,-2}, {-2,0}, {-1,2}, {1,2}, {2,0}, {1,-2}, {-1,-2}, {-2,0} };
__constant__ signed char k_mod6m1[8] = { 5, 0, 1, 2, 3, 4, 5, 0 };
__constant__ signed char k_square1[9][2] = { {0,0}, {0,-1}, {0,1}, {-1,0}, {1,0}, {-1,-1}, {-1,1}, {1,-1}, {1,1} };
__constant__ signed char k_twopoint[16][2] = {
    {-1,0}, {0,-1}, {-1,-1}, {1,-1}, {-1,0}, {1,0}, {-1,1}, {-1,-1},
    {1,-1}, {1,1}, {-1,0}, {0,1}, {-1,1}, {1,1}, {1,0}, {0,1},
};
// the eight points of one star round in the reference's order (motion.cpp:420-470):
// direction, whether the point sits at half the round's distance, and its point number
__constant__ signed char k_star8[8][4] = {
    { 0, -1, 0, 2 }, { -1, -1, 1, 1 }, { 1, -1, 1, 3 }, { -1, 0, 0, 4 },
    { 1, 0, 0, 5 }, { -1, 1, 1, 6 }, { 1, 1, 1, 8 }, { 0, 1, 0, 7 },
};
// subme workload: half-pel iterations, half-pel directions, quarter-pel iterations, quarter-pel directions,
// SATD at the half-pel stage (motion.cpp:48-58)
__constant__ signed char k_workload[8][5] = {
    { 1, 4, 0, 4, 0 }, { 1, 4, 1, 4, 0 }, { 1, 4, 1, 4, 1 }, { 2, 4, 1, 4, 1 },
    { 2, 4, 2, 4, 1 }, { 1, 8, 1, 8, 1 }, { 2, 8, 1, 8, 1 }, { 2, 8, 2, 8, 1 },
};

#ifndef XG_PU_WAVES
#define XG_PU_WAVES 4
#endif
constexpr int PU_REC_DW = 8;   // LDS record per lane and pass: 4 dwords encode column pairs, pixel offset, mask, 2 spare

template <typename P>
struct PuCtx
{
    const XG_GLOBAL P* ref;        // reference picture, pixel (0,0)
    int stride;
    int x, y, w, h;                // the PU
    int minX, maxX, minY, maxY;    // legal block origins (inside the padded picture)
    int passes;
    int lane;                      // lane inside the PU's share of the wave
    int phys;                      // lane inside the wave
    int lpp;                       // lanes per PU: 64, or 32 / 16 / 8 when several small PUs share a wave
    const XG_LDS uint32_t* rec;    // this wave's records, [pass][lane of the wave][PU_REC_DW]
    int minx, miny, maxx, maxy;    // full-pel search bounds
    const XG_GLOBAL uint16_t* cx;  // MV-cost table shifted by the predictor: cx[q] = cost[q - mvp.x]
    const XG_GLOBAL uint16_t* cy;
    int lo, hi;                    // valid index range of cx/cy is [lo - mvp, hi - mvp]; see mvcost()
    int mvpx, mvpy;
    int depth;
    int bmx, bmy, bcost;           // running best of the integer search
    const XG_GLOBAL P* fencPic;    // encode picture, pixel (0,0) (same stride as the reference)
    bool tileForm;                 // lane layout of the staged encode block (stage_encode_block)
};

// sum over the wave of per-lane values; the total comes back as a wave-uniform scalar
__device__ __forceinline__ int wave_sum(int v)
{
    v += dpp<DPP_XOR1>(v);
    v += dpp<DPP_XOR2>(v);
    v += dpp<DPP_HALF_MIRROR>(v);
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);   // row_ror:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

// Sum over the lanes of one PU.  A PU that owns the wave gets a wave-uniform scalar back; PUs that share
// a wave get the total in each of their lanes (and everything decided from it becomes per-lane code, so
// the PUs of a wave walk their own candidate sequences).  c.lpp is a compile-time constant of the kernel
// instance once everything is inlined.
template <typename P> __device__ __forceinline__ int pu_sum(const PuCtx<P>& c, int v)
{
    if (c.lpp == 64)
        return wave_sum(v);
    v += dpp<DPP_XOR1>(v);
    v += dpp<DPP_XOR2>(v);
    v += dpp<DPP_HALF_MIRROR>(v);
    if (c.lpp >= 16)
        v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);   // row_ror:8
    if (c.lpp == 32)
        v += __builtin_amdgcn_ds_bpermute(4 * (c.phys ^ 16), v);          // the other row of the pair
    return v;
}

// BitCost::mvcost (source/encoder/bitcost.h:56): uint16 sum of the two table entries
template <typename P> __device__ __forceinline__ int mvcost(const PuCtx<P>& c, int qx, int qy)
{
    const int dx = clampi(qx - c.mvpx, c.lo, c.hi), dy = clampi(qy - c.mvpy, c.lo, c.hi);
    return ((int)c.cx[dx + c.mvpx] + (int)c.cy[dy + c.mvpy]) & 0xffff;
}

// the same two entries added as plain integers (ADD_MVCOST of the multi-hexagon grid, motion.cpp:1273)
template <typename P> __device__ __forceinline__ int mvcost_sum(const PuCtx<P>& c, int qx, int qy)
{
    const int dx = clampi(qx - c.mvpx, c.lo, c.hi), dy = clampi(qy - c.mvpy, c.lo, c.hi);
    return (int)c.cx[dx + c.mvpx] + (int)c.cy[dy + c.mvpy];
}

// origin of the block displaced by full-pel (mx,my), clamped into the padded picture
// (a no-op for every vector the reference itself can produce)
template <typename P> __device__ __forceinline__ const XG_GLOBAL P* block_at(const PuCtx<P>& c, int mx, int my)
{
    const int X = clampi(c.x + mx, c.minX, c.maxX), Y = clampi(c.y + my, c.minY, c.maxY);
    return c.ref + ((long long)Y * c.stride + X);
}

// ---- 8-tap interpolation of this lane's eight pixels --------------------------------
template <typename P> __device__ __forceinline__ void load16(const XG_GLOBAL P* p, int v[16])
{
    Row<P> a = load_row_g<P>(p, 0), b = load_row_g<P>(p, 8);
    unpack_row<P>(a, v);
    unpack_row<P>(b, v + 8);
}

// `p` points at this lane's first pixel at the vector's full-pel part; (xf,yf) its quarter-pel part
template <typename P>
__device__ __forceinline__ Row<P> interp_row(const XG_GLOBAL P* p, int stride, int xf, int yf, int depth)
{
    const int maxv = (1 << depth) - 1;
    int out[8];
    if (!yf)
    {
        // interp_horiz_pp_c, ipfilter.cpp:79-118
        int in[16];
        load16<P>(p - 3, in);
#pragma unroll
        for (int i = 0; i < 8; i++)
        {
            int s = 0;
#pragma unroll
            for (int k = 0; k < 8; k++)
                s += in[i + k] * k_lumaTaps[xf][k];
            out[i] = clampi((s + 32) >> 6, 0, maxv);
        }
    }
    else if (!xf)
    {
        // interp_vert_pp_c, ipfilter.cpp:164-203
        int acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            int in[8];
            unpack_row<P>(load_row_g<P>(p, (k - 3) * stride), in);
#pragma unroll
            for (int i = 0; i < 8; i++)
                acc[i] += in[i] * k_lumaTaps[yf][k];
        }
#pragma unroll
        for (int i = 0; i < 8; i++)
            out[i] = clampi((acc[i] + 32) >> 6, 0, maxv);
    }
    else
    {
        // interp_hv_pp_c, ipfilter.cpp:362-369: horizontal pass to 14-bit intermediates, vertical pass back
        const int headRoom = 14 - depth;
        const int shiftH = 6 - headRoom;
        const int offH = -(8192 << shiftH);
        const int shiftV = 6 + headRoom;
        const int offV = (1 << (shiftV - 1)) + (8192 << 6);
        int acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll 4
        for (int k = 0; k < 8; k++)
        {
            int in[16];
            load16<P>(p + (k - 3) * stride - 3, in);
            const int ty = k_lumaTaps[yf][k];
#pragma unroll
            for (int i = 0; i < 8; i++)
            {
                int s = 0;
#pragma unroll
                for (int t = 0; t < 8; t++)
                    s += in[i + t] * k_lumaTaps[xf][t];
                acc[i] += ((s + offH) >> shiftH) * ty;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++)
            out[i] = clampi((acc[i] + offV) >> shiftV, 0, maxv);
    }
    return pack_row<P>(out);
}

// ---- block compares -------------------------------------------------------------------

// SAD or SATD of the block at quarter-pel (qx,qy) (full-pel when both low bits pairs are 0)
template <typename P, bool SATD> __device__ __forceinline__ int block_cost(const PuCtx<P>& c, int qx, int qy)
{
    const XG_GLOBAL P* base = block_at<P>(c, qx >> 2, qy >> 2);
    const int xf = qx & 3, yf = qy & 3;
    uint32_t acc = 0;
    if (!SATD && !(xf | yf))
    {
        // full-pel SAD, the bulk of a search: the row loads of up to four passes are issued together
#pragma unroll 1
        for (int k = 0; k < c.passes; k += 4)
        {
            Row<P> rows[4];
            const XG_LDS uint32_t* r = c.rec + (k * 64 + c.phys) * PU_REC_DW;
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (k + i < c.passes)
                    rows[i] = load_row_g<P>(base, (int)r[i * 64 * PU_REC_DW + 4]);
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (k + i < c.passes)
                {
                    const XG_LDS uint32_t* ri = r + i * 64 * PU_REC_DW;
                    const uint32_t mask = ri[5];
                    const Cols d = pair_cols<P>(rows[i]);
                    acc = __builtin_amdgcn_sad_u16(ri[0], d.q[0] & mask, acc);
                    acc = __builtin_amdgcn_sad_u16(ri[1], d.q[1] & mask, acc);
                    acc = __builtin_amdgcn_sad_u16(ri[2], d.q[2] & mask, acc);
                    acc = __builtin_amdgcn_sad_u16(ri[3], d.q[3] & mask, acc);
                }
        }
        return pu_sum<P>(c, (int)acc);
    }
#pragma unroll 1
    for (int k = 0; k < c.passes; k++)
    {
        const XG_LDS uint32_t* r = c.rec + (k * 64 + c.phys) * PU_REC_DW;
        const uint32_t f0 = r[0], f1 = r[1], f2 = r[2], f3 = r[3], mask = r[5];
        const int off = (int)r[4];
        Row<P> row;
        if (xf | yf)
            row = interp_row<P>(base + off, c.stride, xf, yf, c.depth);
        else
            row = load_row_g<P>(base, off);
        const Cols d = pair_cols<P>(row);
        if (SATD)
        {
            // invalid lanes and masked halves carry zero differences
            acc += (uint32_t)satd8x4_cols(as_s16x2(f0) - as_s16x2(d.q[0] & mask), as_s16x2(f1) - as_s16x2(d.q[1] & mask),
                                          as_s16x2(f2) - as_s16x2(d.q[2] & mask), as_s16x2(f3) - as_s16x2(d.q[3] & mask),
                                          c.lane & 3);
        }
        else
        {
            acc = __builtin_amdgcn_sad_u16(f0, d.q[0] & mask, acc);
            acc = __builtin_amdgcn_sad_u16(f1, d.q[1] & mask, acc);
            acc = __builtin_amdgcn_sad_u16(f2, d.q[2] & mask, acc);
            acc = __builtin_amdgcn_sad_u16(f3, d.q[3] & mask, acc);
        }
    }
    if (SATD)
        acc = (c.lane & 3) ? 0u : acc;      // every lane of a quad holds its tile's value: count it once
    return pu_sum<P>(c, (int)acc);
}

// SADs of four blocks at full-pel vectors (mx[k], my[k]) (what sad_x4 does for the reference,
// source/common/pixel.cpp:113-136): the four row loads of a pass are in flight together and the four sums
// reduce side by side, which is what a wave that owns its candidates one after the other otherwise lacks.
template <typename P, int N>
__device__ __forceinline__ void sad_pts(const PuCtx<P>& c, const int mx[N], const int my[N], int out[N])
{
    const XG_GLOBAL P* base[N];
#pragma unroll
    for (int k = 0; k < N; k++)
        base[k] = block_at<P>(c, mx[k], my[k]);
    uint32_t acc[N];
#pragma unroll
    for (int k = 0; k < N; k++)
        acc[k] = 0;
#pragma unroll 1
    for (int p = 0; p < c.passes; p++)
    {
        const XG_LDS uint32_t* r = c.rec + (p * 64 + c.phys) * PU_REC_DW;
        const uint32_t f0 = r[0], f1 = r[1], f2 = r[2], f3 = r[3], mask = r[5];
        const int off = (int)r[4];
        Row<P> rows[N];
#pragma unroll
        for (int k = 0; k < N; k++)
            rows[k] = load_row_g<P>(base[k], off);
#pragma unroll
        for (int k = 0; k < N; k++)
        {
            const Cols d = pair_cols<P>(rows[k]);
            acc[k] = __builtin_amdgcn_sad_u16(f0, d.q[0] & mask, acc[k]);
            acc[k] = __builtin_amdgcn_sad_u16(f1, d.q[1] & mask, acc[k]);
            acc[k] = __builtin_amdgcn_sad_u16(f2, d.q[2] & mask, acc[k]);
            acc[k] = __builtin_amdgcn_sad_u16(f3, d.q[3] & mask, acc[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < N; k++)
        out[k] = pu_sum<P>(c, (int)acc[k]);
}

template <typename P>
__device__ __forceinline__ void sad_pts4(const PuCtx<P>& c, const int mx[4], const int my[4], int out[4])
{
    sad_pts<P, 4>(c, mx, my, out);
}

template <typename P> __device__ __forceinline__ int fpel_cost(const PuCtx<P>& c, int mx, int my)
{
    return block_cost<P, false>(c, mx * 4, my * 4) + mvcost<P>(c, mx * 4, my * 4);
}

// fpel_cost of N vectors at once (sad_pts plus the vector costs)
template <typename P, int N>
__device__ __forceinline__ void fpel_costs(const PuCtx<P>& c, const int mx[N], const int my[N], int cost[N])
{
    int mc[N];
#pragma unroll
    for (int k = 0; k < N; k++)
        mc[k] = mvcost<P>(c, mx[k] * 4, my[k] * 4);
    sad_pts<P, N>(c, mx, my, cost);
#pragma unroll
    for (int k = 0; k < N; k++)
        cost[k] += mc[k];
}

template <typename P> __device__ __forceinline__ void ensure_tile_form(PuCtx<P>& c);
template <typename P> __device__ __forceinline__ int subpel_cost(PuCtx<P>& c, int qx, int qy, bool satd)
{
    if (satd)
        ensure_tile_form<P>(c);
    return (satd ? block_cost<P, true>(c, qx, qy) : block_cost<P, false>(c, qx, qy)) + mvcost<P>(c, qx, qy);
}

template <typename P> __device__ __forceinline__ bool y_ok(const PuCtx<P>& c, int y) { return y >= c.miny && y <= c.maxy; }
template <typename P> __device__ __forceinline__ bool in_range(const PuCtx<P>& c, int x, int y)
{
    return x >= c.minx && x <= c.maxx && y >= c.miny && y <= c.maxy;
}

// COST_MV, motion.cpp:263-269
template <typename P> __device__ __forceinline__ void try_mv(PuCtx<P>& c, int mx, int my)
{
    const int cost = fpel_cost<P>(c, mx, my);
    if (cost < c.bcost)
    {
        c.bcost = cost;
        c.bmx = mx;
        c.bmy = my;
    }
}

// 8-point square refinement, motion.cpp:1120-1138
template <typename P> __device__ __forceinline__ void square_refine(PuCtx<P>& c)
{
    const int ox = c.bmx, oy = c.bmy;
    int dir = 0;
    int mxs[8], mys[8], cost[8];
#pragma unroll
    for (int k = 0; k < 8; k++)
    {
        mxs[k] = ox + k_square1[k + 1][0];
        mys[k] = oy + k_square1[k + 1][1];
    }
    fpel_costs<P, 8>(c, mxs, mys, cost);
#pragma unroll
    for (int k = 0; k < 8; k++)
        if ((mys[k] == oy || y_ok<P>(c, mys[k])) && cost[k] < c.bcost)
        {
            c.bcost = cost[k];
            dir = k + 1;
        }
    c.bmx = ox + k_square1[dir][0];
    c.bmy = oy + k_square1[dir][1];
}

// radius-2 hexagon walk + square refinement, motion.cpp:1041-1140
template <typename P> __device__ __forceinline__ void hex_search(PuCtx<P>& c, int merange)
{
    int bx = c.bmx, by = c.bmy;
    int bc = c.bcost << 3;
    {
        // (-2,0) (-1,2) (1,2) (2,0) (1,-2) (-1,-2) = hex2[1..6], direction values 2..7; measured together
        int mxs[6], mys[6], cost[6];
#pragma unroll
        for (int k = 0; k < 6; k++)
        {
            mxs[k] = bx + k_hex2[k + 1][0];
            mys[k] = by + k_hex2[k + 1][1];
        }
        fpel_costs<P, 6>(c, mxs, mys, cost);
#pragma unroll
        for (int k = 0; k < 6; k++)
            if (y_ok<P>(c, mys[k]))
                bc = min(bc, (cost[k] << 3) + k + 2);
    }
    if (bc & 7)
    {
        int dir = (bc & 7) - 2;
        if (y_ok<P>(c, by + k_hex2[dir + 1][1]))
        {
            bx += k_hex2[dir + 1][0];
            by += k_hex2[dir + 1][1];
#pragma unroll 1
            for (int i = (merange >> 1) - 1; i > 0 && in_range<P>(c, bx, by); i--)
            {
                bc &= ~7;
                int mxs[3], mys[3], cost[3];
#pragma unroll
                for (int k = 0; k < 3; k++)
                {
                    mxs[k] = bx + k_hex2[dir + k][0];
                    mys[k] = by + k_hex2[dir + k][1];
                }
                fpel_costs<P, 3>(c, mxs, mys, cost);
#pragma unroll
                for (int k = 0; k < 3; k++)
                    if (y_ok<P>(c, mys[k]))
                        bc = min(bc, (cost[k] << 3) + k + 1);
                if (!(bc & 7))
                    break;
                dir += (bc & 7) - 2;
                dir = k_mod6m1[dir + 1];
                bx += k_hex2[dir + 1][0];
                by += k_hex2[dir + 1][1];
            }
        }
    }
    c.bcost = bc >> 3;
    c.bmx = bx;
    c.bmy = by;
    square_refine<P>(c);
}

// radius-1 diamond, motion.cpp:1016-1039
template <typename P> __device__ __forceinline__ void dia_search(PuCtx<P>& c, int merange)
{
    int bx = c.bmx, by = c.bmy;
    int bc = c.bcost << 4;
    int i = merange;
    do
    {
        int mxs[4], mys[4], cost[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            mxs[k] = bx + k_square1[k + 1][0];
            mys[k] = by + k_square1[k + 1][1];
        }
        fpel_costs<P, 4>(c, mxs, mys, cost);
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const int step = (0xC431 >> (4 * k)) & 15;            // 1, 3, 4, 12
            if (mys[k] == by || y_ok<P>(c, mys[k]))
                bc = min(bc, (cost[k] << 4) + step);
        }
        if (!(bc & 15))
            break;
        bx -= (int)((uint32_t)bc << 28) >> 30;
        by -= (int)((uint32_t)bc << 30) >> 30;
        bc &= ~15;
    }
    while (--i && in_range<P>(c, bx, by));
    c.bcost = bc >> 4;
    c.bmx = bx;
    c.bmy = by;
}

// COST_MV_PT_DIST with the per-point range test of StarPatternSearch's border branch
// (motion.cpp:249-261, 387-629); inside the frame those tests all hold, so the one
// form covers both branches of the reference
template <typename P>
__device__ __forceinline__ void star_try(PuCtx<P>& c, int mx, int my, int point, int dist, int& bPoint, int& bDist)
{
    const int cost = fpel_cost<P>(c, mx, my);
    if (cost < c.bcost)
    {
        c.bcost = cost;
        c.bmx = mx;
        c.bmy = my;
        bPoint = point;
        bDist = dist;
    }
}

template <typename P>
__device__ __forceinline__ void star_pattern(PuCtx<P>& c, int earlyExitIters, int merange, int& bPoint, int& bDist)
{
    const int ox = c.bmx, oy = c.bmy;
    int rounds = 0;
    const int last = (int)(short)merange;
#pragma unroll 1
    for (int dist = 1; dist < 16 || dist <= last; dist <<= 1)
    {
        const int saved = c.bcost;
        // A round's points are measured four at a time (their costs do not depend on each other) and then
        // offered in the reference's order; a point outside the range is measured at its clamped place and dropped.
        const int npts = dist < 16 ? 8 : 16;
        const int h = dist >> 1, q = dist >> 2;
#pragma unroll 1
        for (int k0 = 0; k0 < npts; k0 += 4)
        {
            int mxs[4], mys[4], pts[4], ds[4], sad[4], mc[4];
            bool ok[4];
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                const int k = k0 + i;
                int sx, sy;
                bool take = true;
                if (dist < 16)
                {
                    sx = k_star8[k][0];
                    sy = k_star8[k][1];
                    const int half = k_star8[k][2];
                    pts[i] = k_star8[k][3];
                    take = !(half && dist == 1);    // the first round has only the four axis points
                    ds[i] = half ? h : dist;
                    mxs[i] = ox + sx * ds[i];
                    mys[i] = oy + sy * ds[i];
                }
                else if (k < 4)
                {
                    // top, left, right, bottom
                    sx = (k == 1) ? -1 : (k == 2) ? 1 : 0;
                    sy = (k == 0) ? -1 : (k == 3) ? 1 : 0;
                    pts[i] = 0;
                    ds[i] = dist;
                    mxs[i] = ox + sx * dist;
                    mys[i] = oy + sy * dist;
                }
                else
                {
                    // for idx = 1..3: (xL,yT) (xR,yT) (xL,yB) (xR,yB)
                    const int idx = ((k - 4) >> 2) + 1, corner = (k - 4) & 3;
                    sx = (corner & 1) ? 1 : -1;
                    sy = (corner & 2) ? 1 : -1;
                    pts[i] = 0;
                    ds[i] = dist;
                    mxs[i] = ox + sx * q * idx;
                    mys[i] = oy + sy * (dist - q * idx);
                }
                // the reference tests only the side a point can leave the range on
                ok[i] = take && (sx >= 0 || mxs[i] >= c.minx) && (sx <= 0 || mxs[i] <= c.maxx) &&
                        (sy >= 0 || mys[i] >= c.miny) && (sy <= 0 || mys[i] <= c.maxy);
                mc[i] = mvcost<P>(c, mxs[i] * 4, mys[i] * 4);
            }
            bool any = false;
#pragma unroll
            for (int i = 0; i < 4; i++)
                any |= ok[i];
            if (!any)
                continue;
            sad_pts4<P>(c, mxs, mys, sad);
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (ok[i] && sad[i] + mc[i] < c.bcost)
                {
                    c.bcost = sad[i] + mc[i];
                    c.bmx = mxs[i];
                    c.bmy = mys[i];
                    bPoint = pts[i];
                    bDist = ds[i];
                }
        }
        if (c.bcost < saved)
            rounds = 0;
        else if (++rounds >= earlyExitIters)
            return;
    }
}

template <typename P>
__device__ __forceinline__ bool star_two_points(PuCtx<P>& c, int point)
{
    const int saved = c.bcost;
    const int ox = c.bmx, oy = c.bmy;
#pragma unroll 1
    for (int k = 0; k < 2; k++)
    {
        const int mx = ox + k_twopoint[(point - 1) * 2 + k][0], my = oy + k_twopoint[(point - 1) * 2 + k][1];
        if (in_range<P>(c, mx, my))
            try_mv<P>(c, mx, my);
    }
    return c.bcost != saved;
}

// star search driver, motion.cpp:1328-1436
template <typename P> __device__ __forceinline__ void star_search(PuCtx<P>& c, int merange)
{
    int bPoint = 0, bDist = 0;
    bool first = true;
#pragma unroll 1
    for (;;)
    {
        if (!first)
        {
            bDist = 0;
            bPoint = 0;
        }
        star_pattern<P>(c, first ? 3 : 32, merange, bPoint, bDist);
        if (bDist == 1)
        {
            if (!bPoint)
                return;
            const bool moved = star_two_points<P>(c, bPoint);
            if (!first || !moved)
                return;
        }
        if (first)
        {
            first = false;
            const int raster = 5;
            if (bDist > raster)
            {
#pragma unroll 1
                for (int ty = c.miny; ty <= c.maxy; ty += raster)
#pragma unroll 1
                    for (int tx = c.minx; tx <= c.maxx; tx += raster)
                    {
                        if (tx + raster * 3 <= c.maxx)
                        {
                            int mxs[4], mys[4], sad[4], mc[4];
#pragma unroll
                            for (int k = 0; k < 4; k++)
                            {
                                mxs[k] = tx + raster * k;
                                mys[k] = ty;
                                // the fourth point is priced with the vector scaled by 8 instead of 4 (motion.cpp:1392)
                                mc[k] = k < 3 ? mvcost<P>(c, mxs[k] * 4, ty * 4) : mvcost<P>(c, mxs[k] * 8, ty * 8);
                            }
                            sad_pts4<P>(c, mxs, mys, sad);
#pragma unroll
                            for (int k = 0; k < 4; k++)
                                if (sad[k] + mc[k] < c.bcost)
                                {
                                    c.bcost = sad[k] + mc[k];
                                    c.bmx = mxs[k];
                                    c.bmy = ty;
                                }
                            tx += raster * 3;
                        }
                        else
                            try_mv<P>(c, tx, ty);
                    }
            }
        }
        if (bDist <= 0)
            return;
    }
}

// exhaustive search, motion.cpp:1593-1637
template <typename P> __device__ __forceinline__ void full_search(PuCtx<P>& c)
{
#pragma unroll 1
    for (int ty = c.miny; ty <= c.maxy; ty++)
#pragma unroll 1
        for (int tx = c.minx; tx <= c.maxx; tx++)
            try_mv<P>(c, tx, ty);
}

// Successive elimination, motion.cpp:1438-1591, for a PU that owns its wave.  Rows of the window are visited in
// order; in a row every lane takes one candidate column (two with a window wider than 64), adds up the absolute
// DC differences of the partition's terms plus the column's vector cost, and the ballot of "below the row's
// threshold" is the list ads() writes.  The survivors' SADs are then taken one after the other by the whole wave,
// in list order, with the reference's bookkeeping: triples are compared without the row's vector cost against a
// best cost reduced by `ycost`, the remainder as COST_MV; `ycost` is the cost table read at the full-pel row,
// times four, as the reference has it.
template <typename P>
__device__ __forceinline__ void sea_search(PuCtx<P>& c, const PuFrame& f, int merange)
{
    SeaShape sh;
    if (!sea_shape(c.w, c.h, sh))
        return;                              // the host refuses such jobs
    const int minX = max(c.bmx - merange, c.minx), minY = max(c.bmy - merange, c.miny);
    const int maxX = min(c.bmx + merange, c.maxx), maxY = min(c.bmy + merange, c.maxy);
    const int width = (maxX - minX + 3) & ~3;    // "fastest in multiples of 4": up to three columns past maxX
    // DC of the encode sub-blocks
    int encDC[4] = { 0, 0, 0, 0 };
    const int cells = sh.tw * sh.th;
#pragma unroll 1
    for (int k = 0; k < sh.terms; k++)
    {
        const XG_GLOBAL P* src = c.fencPic + ((long long)(c.y + sh.encY[k]) * c.stride + c.x + sh.encX[k]);
        int v = 0;
        for (int i = c.phys; i < cells; i += 64)
        {
            const int r = i / sh.tw;
            v += (int)src[r * c.stride + (i - r * sh.tw)];
        }
        encDC[k] = wave_sum(v);
    }
    const XG_GLOBAL uint32_t* sums = as_global<uint32_t>(f.seaSums[sh.plane]) + ((long long)c.y * c.stride + c.x);
#pragma unroll 1
    for (int ty = minY; ty <= maxY; ty++)
    {
        const int ycost = (int)c.cy[clampi(ty - c.mvpy, c.lo, c.hi) + c.mvpy] << 2;
        if (c.bcost <= ycost)
            continue;
        int reduced = c.bcost - ycost;
        const int thresh = reduced;
        int bx = c.bmx, by = c.bmy;
        int taken = 0;                        // survivors seen so far in this row
        int total = 0;
        // first pass over the row: how many survive (the split into triples and remainder needs the count)
        unsigned long long keep[4] = { 0, 0, 0, 0 };
#pragma unroll 1
        for (int base = 0, chunk = 0; base < width && chunk < 4; base += 64, chunk++)
        {
            const int i = base + c.phys;
            bool pass = false;
            if (i < width)
            {
                const int mx = minX + i;
                const XG_GLOBAL uint32_t* p = sums + ((long long)ty * c.stride + mx);
                int ads = (int)c.cx[clampi(4 * mx - c.mvpx, c.lo, c.hi) + c.mvpx];
#pragma unroll 1
                for (int k = 0; k < sh.terms; k++)
                    ads += abs(encDC[k] - (int)p[sh.sumY[k] * c.stride + sh.sumX[k]]);
                pass = ads < thresh;
            }
            keep[chunk] = __builtin_amdgcn_ballot_w64(pass);
            total += __builtin_popcountll(keep[chunk]);
        }
        const int triples = total >= 3 ? total - total % 3 : 0;
#pragma unroll 1
        for (int chunk = 0; chunk < 4; chunk++)
        {
            unsigned long long m = keep[chunk];
            while (m)
            {
                const int bit = __builtin_ctzll(m);
                m &= m - 1;
                const int mx = minX + chunk * 64 + bit;
                const int sad = block_cost<P, false>(c, mx * 4, ty * 4);
                if (taken < triples)
                {
                    const int cost = sad + (int)c.cx[clampi(4 * mx - c.mvpx, c.lo, c.hi) + c.mvpx];
                    if (cost < reduced)
                    {
                        reduced = cost;
                        bx = mx;
                        by = ty;
                    }
                }
                else
                {
                    if (taken == triples)
                    {
                        c.bcost = reduced + ycost;      // the triples are done: bcost gets the row cost back
                        reduced = -1;
                    }
                    const int cost = sad + mvcost<P>(c, mx * 4, ty * 4);
                    if (cost < c.bcost)
                    {
                        c.bcost = cost;
                        bx = mx;
                        by = ty;
                    }
                }
                taken++;
            }
        }
        if (reduced >= 0)
            c.bcost = reduced + ycost;
        c.bmx = bx;
        c.bmy = by;
    }
}

// COST_MV_X4 as this fork defines it (motion.cpp:302-323): the offset is added to `o`, the
// candidate is measured whatever its row, and accepted only when the row is in range.
template <typename P> __device__ __forceinline__ void try_rel(PuCtx<P>& c, int ox, int oy, int dx, int dy)
{
    const int x = ox + dx, y = oy + dy;
    const int cost = fpel_cost<P>(c, x, y);
    if (y_ok<P>(c, y) && cost < c.bcost)
    {
        c.bcost = cost;
        c.bmx = x;
        c.bmy = y;
    }
}

// four COST_MV_X4 offsets packed as biased nibbles
template <typename P> __device__ __forceinline__ void try_x4(PuCtx<P>& c, int ox, int oy, uint32_t px, uint32_t py)
{
    // measured together, offered in order
    int mxs[4], mys[4], cost[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
        mxs[k] = ox + nib(px, k);
        mys[k] = oy + nib(py, k);
    }
    fpel_costs<P, 4>(c, mxs, mys, cost);
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (y_ok<P>(c, mys[k]) && cost[k] < c.bcost)
        {
            c.bcost = cost[k];
            c.bmx = mxs[k];
            c.bmy = mys[k];
        }
}

// four COST_MV_X4 offsets given one by one
template <typename P>
__device__ __forceinline__ void try_rel4(PuCtx<P>& c, int ox, int oy, int dx0, int dy0, int dx1, int dy1, int dx2, int dy2,
                                         int dx3, int dy3)
{
    const int mxs[4] = { ox + dx0, ox + dx1, ox + dx2, ox + dx3 }, mys[4] = { oy + dy0, oy + dy1, oy + dy2, oy + dy3 };
    int cost[4];
    fpel_costs<P, 4>(c, mxs, mys, cost);
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (y_ok<P>(c, mys[k]) && cost[k] < c.bcost)
        {
            c.bcost = cost[k];
            c.bmx = mxs[k];
            c.bmy = mys[k];
        }
}

// CROSS(start, x_max, y_max) around (ox,oy): motion.cpp:361-385
template <typename P> __device__ __forceinline__ void umh_cross(PuCtx<P>& c, int ox, int oy, int start, int x_max, int y_max)
{
    int i = (short)start;
    if (x_max <= min(c.maxx - ox, ox - c.minx))
        for (; i < x_max - 2; i += 4)
        {
            try_rel4<P>(c, ox, oy, i, 0, -i, 0, i + 2, 0, -i - 2, 0);
        }
    for (; i < x_max; i += 2)
    {
        if (ox + i <= c.maxx)
            try_mv<P>(c, ox + i, oy);
        if (ox - i >= c.minx)
            try_mv<P>(c, ox - i, oy);
    }
    i = (short)start;
    if (y_max <= min(c.maxy - oy, oy - c.miny))
        for (; i < y_max - 2; i += 4)
        {
            try_rel4<P>(c, ox, oy, 0, i, 0, -i, 0, i + 2, 0, -i - 2);
        }
    for (; i < y_max; i += 2)
    {
        if (oy + i <= c.maxy)
            try_mv<P>(c, ox, oy + i);
        if (oy - i >= c.miny)
            try_mv<P>(c, ox, oy - i);
    }
}

// Uneven multi-hexagon search, motion.cpp:1142-1326; returns true when the hexagon walk must follow.
// (pfx,pfy): the clipped predictor rounded to full pel.
template <typename P>
__device__ __forceinline__ bool umh_search(PuCtx<P>& c, const PuJob& job, int pfx, int pfy, int merange)
{
    constexpr uint32_t D1X = 0x00009788u, D1Y = 0x00008897u;     // (0,-1)(0,1)(-1,0)(1,0)
    const int scale = (c.h * c.h) >> 4;                          // sizeScale[], motion.cpp:123-153
#define PU_SAD_THRESH(v) (c.bcost < (((v) >> 4) * scale))
    int cross_start = 1;
    const int ucost1 = c.bcost;
    try_x4<P>(c, pfx, pfy, D1X, D1Y);
    if (pfx | pfy)
        try_x4<P>(c, 0, 0, D1X, D1Y);
    const int ucost2 = c.bcost;
    if ((c.bmx | c.bmy) && (c.bmx != pfx || c.bmy != pfy))
        try_x4<P>(c, c.bmx, c.bmy, D1X, D1Y);
    if (c.bcost == ucost2)
        cross_start = 3;

    int ox = c.bmx, oy = c.bmy;
    if (c.bcost == ucost2 && PU_SAD_THRESH(2000))
    {
        try_x4<P>(c, ox, oy, 0x00006978u, 0x00008776u);          // (0,-2)(-1,-1)(1,-1)(-2,0)
        try_x4<P>(c, ox, oy, 0x0000897Au, 0x0000A998u);          // (2,0)(-1,1)(1,1)(0,2)
        if (c.bcost == ucost1 && PU_SAD_THRESH(500))
            return false;
        if (c.bcost == ucost2)
        {
            const int range = (short)((merange >> 1) | 1);
            umh_cross<P>(c, ox, oy, 3, range, range);
            try_x4<P>(c, ox, oy, 0x0000A697u, 0x00007766u);      // (-1,-2)(1,-2)(-2,-1)(2,-1)
            try_x4<P>(c, ox, oy, 0x000097A6u, 0x0000AA99u);      // (-2,1)(2,1)(-1,2)(1,2)
            if (c.bcost == ucost2)
                return false;
            cross_start = range + 2;
        }
    }

    // search range adapted to how well the extra candidates agree (motion.cpp:1184-1236)
    if (job.numCandidates)
    {
        const int n = job.numCandidates;
        const bool is64 = c.w == 64 && c.h == 64;
        int mvd, denom = 1;
        if (n == 1)
            mvd = is64 ? 25 : abs(c.mvpx - job.mvc[0]) + abs(c.mvpy - job.mvc[1]);
        else
        {
            denom = n - 1;
            mvd = 0;
            if (!is64)
            {
                mvd = abs(c.mvpx - job.mvc[0]) + abs(c.mvpy - job.mvc[1]);
                denom++;
            }
#pragma unroll 1
            for (int i = 0; i < n - 1; i++)
                mvd += abs(job.mvc[2 * i] - job.mvc[2 * i + 2]) + abs(job.mvc[2 * i + 1] - job.mvc[2 * i + 3]);
        }
        const int sad_ctx = PU_SAD_THRESH(1000) ? 0 : PU_SAD_THRESH(2000) ? 1 : PU_SAD_THRESH(4000) ? 2 : 3;
        const int mvd_ctx = mvd < 10 * denom ? 0 : mvd < 20 * denom ? 1 : mvd < 40 * denom ? 2 : 3;
        // range_mul[mvd_ctx][sad_ctx] = {3,3,4,4},{3,4,4,4},{4,4,4,5},{4,4,5,6}, one nibble each
        const uint32_t mul = mvd_ctx == 0 ? 0x4433u : mvd_ctx == 1 ? 0x4443u : mvd_ctx == 2 ? 0x5444u : 0x6544u;
        merange = (merange * (int)((mul >> (4 * sad_ctx)) & 15u)) >> 2;
    }

    // the cross stays centred on (ox,oy) even if the steps above moved the best vector (motion.cpp:1238)
    umh_cross<P>(c, ox, oy, cross_start, merange, merange >> 1);
    try_x4<P>(c, ox, oy, 0x0000AA66u, 0x0000A6A6u);              // (-2,-2)(-2,2)(2,-2)(2,2)

    // hexagon grid of growing radius around the best so far: hex4[16], motion.cpp:67-73
    constexpr uint32_t H4X0 = 0xC4C4A688u, H4Y0 = 0x776655C4u, H4X1 = 0xA6C4C4C4u, H4Y1 = 0xBBAA9988u;
    ox = c.bmx;
    oy = c.bmy;
    int i = 1;
    do
    {
        if (4 * i > min(min(c.maxx - ox, ox - c.minx), min(c.maxy - oy, oy - c.miny)))
        {
#pragma unroll 1
            for (int j = 0; j < 16; j++)
            {
                const int mx = ox + nib(j < 8 ? H4X0 : H4X1, j & 7) * i, my = oy + nib(j < 8 ? H4Y0 : H4Y1, j & 7) * i;
                if (in_range<P>(c, mx, my))
                    try_mv<P>(c, mx, my);
            }
        }
        else
        {
            // all sixteen measured; a point counts when the row of its UNSCALED offset is in range (MIN_MV,
            // motion.cpp:1274); the two table entries are added without the 16-bit wrap of mvcost()
            int dirx = 0, diry = 0;
            bool moved = false;
#pragma unroll 1
            for (int j0 = 0; j0 < 16; j0 += 4)
            {
                int dxs[4], dys[4], mxs[4], mys[4], sad[4], mc[4];
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    const int j = j0 + k;
                    dxs[k] = nib(j < 8 ? H4X0 : H4X1, j & 7);
                    dys[k] = nib(j < 8 ? H4Y0 : H4Y1, j & 7);
                    mxs[k] = ox + dxs[k] * i;
                    mys[k] = oy + dys[k] * i;
                    mc[k] = mvcost_sum<P>(c, mxs[k] * 4, mys[k] * 4);
                }
                sad_pts<P, 4>(c, mxs, mys, sad);
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (y_ok<P>(c, oy + dys[k]) && sad[k] + mc[k] < c.bcost)
                    {
                        c.bcost = sad[k] + mc[k];
                        dirx = dxs[k];
                        diry = dys[k];
                        moved = true;
                    }
            }
            if (moved)
            {
                c.bmx = ox + i * dirx;
                c.bmy = oy + i * diry;
            }
        }
    }
    while (++i <= (merange >> 2));
#undef PU_SAD_THRESH
    return in_range<P>(c, c.bmx, c.bmy);
}

// MotionEstimate::diamondSearch (motion.cpp:631-788), the coarse search behind an area's
// starting vector.  Inside the range the reference hands absolute coordinates to the relative
// COST_MV_X4, so the position measured (and stored) is o + coordinate; try_rel keeps that.
template <typename P> __device__ __forceinline__ void area_diamond_search(PuCtx<P>& c)
{
    c.bcost = 0x7fffffff;
    c.bmx = c.bmy = 0;
    int ox = 0, oy = 0;
#pragma unroll 1
    for (int dist = 1; dist <= 4; dist <<= 1)
    {
        const int sx = c.bmx, sy = c.bmy;
        const int h = dist >> 1;
        const int top = oy - dist, bottom = oy + dist, left = ox - dist, right = ox + dist;
        const int top2 = oy - h, bottom2 = oy + h, left2 = ox - h, right2 = ox + h;
        if (top >= c.miny && left >= c.minx && right <= c.maxx && bottom <= c.maxy)
        {
            try_rel<P>(c, ox, oy, ox, top);
            try_rel<P>(c, ox, oy, ox, bottom);
            try_rel<P>(c, ox, oy, left, oy);
            try_rel<P>(c, ox, oy, right, oy);
            try_rel<P>(c, ox, oy, left2, top2);
            try_rel<P>(c, ox, oy, right2, top2);
            try_rel<P>(c, ox, oy, left2, bottom2);
            try_rel<P>(c, ox, oy, right2, bottom2);
        }
        else
        {
            if (top >= c.miny)
                try_mv<P>(c, ox, top);
            if (top2 >= c.miny)
            {
                if (left2 >= c.minx)
                    try_mv<P>(c, left2, top2);
                if (right2 <= c.maxx)
                    try_mv<P>(c, right2, top2);
            }
            if (left >= c.minx)
                try_mv<P>(c, left, oy);
            if (right <= c.maxx)
                try_mv<P>(c, right, oy);
            if (bottom2 <= c.maxy)
            {
                if (left2 >= c.minx)
                    try_mv<P>(c, left2, bottom2);
                if (right2 <= c.maxx)
                    try_mv<P>(c, right2, bottom2);
            }
            if (bottom <= c.maxy)
                try_mv<P>(c, ox, bottom);
        }
        if (c.bmx == sx && c.bmy == sy)
            break;
    }
    ox = c.bmx;
    oy = c.bmy;
#pragma unroll 1
    for (int dist = 8; dist <= 64; dist += 8)
    {
        const int sx = c.bmx, sy = c.bmy;
        const int top = oy - dist, bottom = oy + dist, left = ox - dist, right = ox + dist;
        const bool inside = top >= c.miny && left >= c.minx && right <= c.maxx && bottom <= c.maxy;
        if (inside)
        {
            try_rel<P>(c, ox, oy, ox, top);
            try_rel<P>(c, ox, oy, left, oy);
            try_rel<P>(c, ox, oy, right, oy);
            try_rel<P>(c, ox, oy, ox, bottom);
        }
        else
        {
            if (top >= c.miny)
                try_mv<P>(c, ox, top);
            if (left >= c.minx)
                try_mv<P>(c, left, oy);
            if (right <= c.maxx)
                try_mv<P>(c, right, oy);
            if (bottom <= c.maxy)
                try_mv<P>(c, ox, bottom);
        }
#pragma unroll 1
        for (int index = 1; index < 4; index++)
        {
            const int q = (dist >> 2) * index;
            const int yt = top + q, yb = bottom - q, xl = ox - q, xr = ox + q;
            if (inside)
            {
                try_rel<P>(c, ox, oy, xl, yt);
                try_rel<P>(c, ox, oy, xr, yt);
                try_rel<P>(c, ox, oy, xl, yb);
                try_rel<P>(c, ox, oy, xr, yb);
                continue;
            }
            if (yt >= c.miny)
            {
                if (xl >= c.minx)
                    try_mv<P>(c, xl, yt);
                if (xr <= c.maxx)
                    try_mv<P>(c, xr, yt);
            }
            if (yb <= c.maxy)
            {
                if (xl >= c.minx)
                    try_mv<P>(c, xl, yb);
                if (xr <= c.maxx)
                    try_mv<P>(c, xr, yb);
            }
        }
        if (c.bmx == sx && c.bmy == sy)
            break;
        ox = c.bmx;
        oy = c.bmy;
    }
}

// The encode block into LDS (MotionEstimate::setSourcePU, motion.cpp:167-191): per pass and lane the
// row's column pairs (masked), the row's pixel offset inside the block, and the mask.  Two lane layouts:
// tile form (a quad of lanes holds the four rows of an 8x4 / 4x4 tile, what the SATD needs) and row form
// (consecutive lanes hold consecutive segments of one row, so the lanes of a row read neighbouring bytes of
// the reference and share cache lines; a SAD only sums over lanes and takes either).  A lane reads back
// only its own records, so switching layouts needs no agreement between the lanes of a wave.
template <typename P>
__device__ __forceinline__ void stage_encode_block(PuCtx<P>& c, const XG_GLOBAL P* fencPic, int stride, XG_LDS uint32_t* rec,
                                                   bool tileForm = true)
{
    const int segW = (c.w & 7) ? 4 : 8;
    const int segsX = c.w / segW, nsegs = segsX * c.h;
    c.passes = (nsegs + c.lpp - 1) / c.lpp;
    const XG_GLOBAL P* fenc = fencPic + ((long long)c.y * stride + c.x);
    for (int k = 0; k < c.passes; k++)
    {
        bool valid;
        int off;
        if (tileForm)
        {
            const int t = k * (c.lpp >> 2) + (c.lane >> 2);
            valid = t < (nsegs >> 2);
            const int ty = valid ? t / segsX : 0, tx = valid ? t - ty * segsX : 0;
            off = (ty * 4 + (c.lane & 3)) * stride + tx * segW;
        }
        else
        {
            const int i = k * c.lpp + c.lane;
            valid = i < nsegs;
            const int row = valid ? i / segsX : 0, sx = valid ? i - row * segsX : 0;
            off = row * stride + sx * segW;
        }
        const uint32_t mask = valid ? (segW == 8 ? 0xffffffffu : 0x0000ffffu) : 0u;
        const Cols fc = pair_cols<P>(load_row_g<P>(fenc, off));
        XG_LDS uint32_t* r = rec + (k * 64 + c.phys) * PU_REC_DW;
        r[0] = fc.q[0] & mask;
        r[1] = fc.q[1] & mask;
        r[2] = fc.q[2] & mask;
        r[3] = fc.q[3] & mask;
        r[4] = (uint32_t)off;
        r[5] = mask;
    }
    c.tileForm = tileForm;
    lds_handover();
}

// SATD compares want the tile form; the integer search and the SAD stages run in row form
template <typename P> __device__ __forceinline__ void ensure_tile_form(PuCtx<P>& c)
{
    if (!c.tileForm)
        stage_encode_block<P>(c, c.fencPic, c.stride, (XG_LDS uint32_t*)c.rec, true);
}

// LPP lanes per PU.  64: one PU per wave, `order` unused or a list of job indices.  Smaller: 64 / LPP
// PUs of at most LPP / 4 tiles share a wave, each in its own lanes with its own control flow; `order`
// lists the jobs of that size class and results go back to their own places.
template <typename P, int LPP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(XG_PU_WAVES))) pu_search_kernel(const PuJob* jobs, const int* order, int n, PuFrame f,
                                                        const uint16_t* costTab, int maxMvd, PuResult* out)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t puLds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int phys = threadIdx.x & 63;
    const int lane = phys & (LPP - 1);
    constexpr int PER_WAVE = 64 / LPP;
    if ((blockIdx.x * 4 + wave) * PER_WAVE >= n)
        return;                               // the whole wave has nothing to do
    int slotIdx = (blockIdx.x * 4 + wave) * PER_WAVE + phys / LPP;
    const bool live = slotIdx < n;            // a share without a PU repeats the last one and stores nothing
    slotIdx = live ? slotIdx : n - 1;
    const int pu = order ? order[slotIdx] : slotIdx;
    const PuJob& job = jobs[pu];      // read in place: indexing mvc[] of a register copy would go through scratch
    XG_LDS uint32_t* rec = (XG_LDS uint32_t*)puLds + wave * (f.maxPasses * 64 * PU_REC_DW);

    PuCtx<P> c;
    c.ref = as_global<P>(f.ref);
    c.stride = f.stride;
    c.x = job.x;
    c.y = job.y;
    c.w = job.w;
    c.h = job.h;
    c.minX = -f.marginX + 4;                       // room for the filter taps and the rows' trailing pixels
    c.maxX = f.width + f.marginX - job.w - 16;
    c.minY = -f.marginY + 4;
    c.maxY = f.height + f.marginY - job.h - 5;
    c.lane = lane;
    c.phys = phys;
    c.lpp = LPP;
    c.rec = rec;
    c.minx = job.minx;
    c.miny = job.miny;
    c.maxx = job.maxx;
    c.maxy = job.maxy;
    c.mvpx = job.mvpx;
    c.mvpy = job.mvpy;
    // the job's own MV-cost table (one per QP in use, 2 * maxMvd + 2 entries apart)
    const XG_GLOBAL uint16_t* table = as_global<uint16_t>(costTab) + (size_t)((unsigned)job.flags >> 8) * (2 * (size_t)maxMvd + 2);
    c.cx = table - job.mvpx;
    c.cy = table - job.mvpy;
    c.lo = -maxMvd;
    c.hi = maxMvd;
    c.depth = f.depth;

    c.fencPic = as_global<P>(f.fenc);
    stage_encode_block<P>(c, c.fencPic, f.stride, rec, f.method == PU_AREA_DIAMOND || !f.rowForm);

    if (f.method == PU_AREA_DIAMOND)
    {
        area_diamond_search<P>(c);
        if (live && lane == 0)
        {
            PuResult r;
            r.mvx = c.bmx;                 // full-pel, as diamondSearch returns it
            r.mvy = c.bmy;
            r.cost = c.bcost;
            r.reserved = 0;
            out[pu] = r;
        }
        return;
    }

    // ---- MotionEstimate::motionEstimate, full-resolution reference ----
    const int qminx = job.minx * 4, qminy = job.miny * 4, qmaxx = job.maxx * 4, qmaxy = job.maxy * 4;
    const int pmvx = clampi(job.mvpx, qminx, qmaxx);
    int pmvy = clampi(job.mvpy, qminy, qmaxy);
    if ((job.flags & 1) && pmvy > qmaxy)                      // m_vertRestriction, motion.cpp:957-963
        pmvy = qmaxy;
    int bestprex = pmvx, bestprey = pmvy;
    int bprecost = block_cost<P, false>(c, pmvx, pmvy);       // SAD, no MV cost on this term
    c.bmx = (pmvx + 2) >> 2;
    c.bmy = (pmvy + 2) >> 2;
    c.bcost = bprecost;
    if ((pmvx | pmvy) & 3)
        c.bcost = fpel_cost<P>(c, c.bmx, c.bmy);
    if (pmvx | pmvy)
    {
        const int cost = block_cost<P, false>(c, 0, 0) + mvcost<P>(c, 0, 0);
        if (cost < c.bcost)
        {
            c.bcost = cost;
            c.bmx = 0;
            c.bmy = max(min(0, job.maxy), job.miny);
        }
    }
    // extra quarter-pel candidates, motion.cpp:989-1003
#pragma unroll 1
    for (int i = 0; i < job.numCandidates; i++)
    {
        const int mx = clampi(job.mvc[2 * i], qminx, qmaxx), my = clampi(job.mvc[2 * i + 1], qminy, qmaxy);
        if ((mx | my) && (mx != pmvx || my != pmvy) && (mx != bestprex || my != bestprey))
        {
            const int cost = subpel_cost<P>(c, mx, my, false);
            if (cost < bprecost)
            {
                bprecost = cost;
                bestprex = mx;
                bestprey = my;
            }
        }
    }

    int bqx, bqy, bcost;
    // no residual at the start point: motionEstimate returns at once (motion.cpp:1007-1011), before the search, the
    // sub-pel steps and the closing zero-vector check
    const bool startIsExact = c.bcost == 0;
    if (startIsExact)
    {
        bqx = c.bmx * 4;
        bqy = c.bmy * 4;
        bcost = mvcost<P>(c, bqx, bqy);
    }
    else
    {
        if (f.method == PU_DIA)
            dia_search<P>(c, f.merange);
        else if (f.method == PU_HEX)
            hex_search<P>(c, f.merange);
        else if (f.method == PU_UMH)
        {
            if (umh_search<P>(c, job, (pmvx + 2) >> 2, (pmvy + 2) >> 2, f.merange))
                hex_search<P>(c, f.merange);
        }
        else if (f.method == PU_STAR)
            star_search<P>(c, f.merange);
        else if (f.method == PU_SEA)
        {
            if constexpr (LPP == 64)
                sea_search<P>(c, f, f.merange);
        }
        else
            full_search<P>(c);

        bcost = c.bcost;
        if (bprecost < bcost)
        {
            bqx = bestprex;
            bqy = bestprey;
            bcost = bprecost;
        }
        else
        {
            bqx = c.bmx * 4;
            bqy = c.bmy * 4;
        }

        if (!bcost)
            bcost = mvcost<P>(c, bqx, bqy);
        else
        {
            // interpolated sub-pel refinement, motion.cpp:1700-1757
            const int hpelIters = k_workload[f.subpelRefine][0], hpelDirs = k_workload[f.subpelRefine][1];
            const int qpelIters = k_workload[f.subpelRefine][2], qpelDirs = k_workload[f.subpelRefine][3];
            const bool hpelSatd = k_workload[f.subpelRefine][4] != 0;
            if (hpelSatd)
                bcost = subpel_cost<P>(c, bqx, bqy, true);
#pragma unroll 1
            for (int stage = 0; stage < 2; stage++)
            {
                // stage 0: half-pel steps (SAD or SATD); stage 1: quarter-pel steps (SATD)
                const int iters = stage ? qpelIters : hpelIters, dirs = stage ? qpelDirs : hpelDirs;
                const int step = stage ? 1 : 2;
                const bool satd = stage ? true : hpelSatd;
                if (stage && !hpelSatd)
                    bcost = subpel_cost<P>(c, bqx, bqy, true);
#pragma unroll 1
                for (int iter = 0; iter < iters; iter++)
                {
                    int bdir = 0;
#pragma unroll 1
                    for (int i = 1; i <= dirs; i++)
                    {
                        const int qx = bqx + k_square1[i][0] * step, qy = bqy + k_square1[i][1] * step;
                        if (qy < qminy || qy > qmaxy)
                            continue;
                        const int cost = subpel_cost<P>(c, qx, qy, satd);
                        if (cost < bcost)
                        {
                            bcost = cost;
                            bdir = i;
                        }
                    }
                    if (!bdir)
                        break;
                    bqx += k_square1[bdir][0] * step;
                    bqy += k_square1[bdir][1] * step;
                }
            }
        }
    }

    // the fork's closing zero-vector check: the vector is replaced, the cost is not (motion.cpp:1762-1768)
    if ((bqx | bqy) && !startIsExact)
    {
        const int cost = subpel_cost<P>(c, 0, 0, true);
        if (cost <= bcost)
            bqx = bqy = 0;
    }
    if (live && lane == 0)
    {
        PuResult r;
        r.mvx = bqx;
        r.mvy = bqy;
        r.cost = bcost;
        r.reserved = 0;
        out[pu] = r;
    }
}

// ---- primitive-level entry points ----------------------------------------------------

// sum of |H8 D H8| over the 8x8 block held by an 8-lane group (lane = row): sa8d_8x8 before
// its rounding, source/common/pixel.cpp:291-326.  Every lane of the group gets the total.
template <typename P> __device__ __forceinline__ int had8x8_abs(const Row<P>& fenc, const Row<P>& ref, int r)
{
    int a[8], b[8], d[8];
    unpack_row<P>(fenc, a);
    unpack_row<P>(ref, b);
#pragma unroll
    for (int i = 0; i < 8; i++)
        d[i] = a[i] - b[i];
    // horizontal 8-point transform (any butterfly order: only magnitudes are summed)
#pragma unroll
    for (int span = 1; span < 8; span <<= 1)
    {
        int t[8];
#pragma unroll
        for (int i = 0; i < 8; i++)
            t[i] = (i & span) ? d[i ^ span] - d[i] : d[i] + d[i ^ span];
#pragma unroll
        for (int i = 0; i < 8; i++)
            d[i] = t[i];
    }
    // vertical 8-point transform across the group's lanes: r ^ 1, r ^ 2, r ^ 4
    const int m1 = -(r & 1), m2 = -((r >> 1) & 1), m4 = -((r >> 2) & 1);
#pragma unroll
    for (int i = 0; i < 8; i++)
        d[i] = ((d[i] ^ m1) - m1) + dpp<DPP_XOR1>(d[i]);
#pragma unroll
    for (int i = 0; i < 8; i++)
        d[i] = ((d[i] ^ m2) - m2) + dpp<DPP_XOR2>(d[i]);
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
    {
        // lane r ^ 4 = the half-row mirror (r -> 7 - r) followed by the quad reversal (q -> 3 - q)
        const int partner = dpp<0x1B>(dpp<DPP_HALF_MIRROR>(d[i]));
        s += abs(((d[i] ^ m4) - m4) + partner);
    }
    return group_sum8(s);
}

// One wave per job: SAD / SATD of a w x h block against the reference block displaced by a
// quarter-pel vector (pixelcmp_t slots pu[].sad / pu[].satd behind MotionEstimate::subpelCompare),
// or SA8D of a square block (cu[].sa8d, source/common/pixel.cpp:328-369).
template <typename P>
__global__ void __launch_bounds__(256) block_compare_kernel(const BlockJob* jobs, int n, PuFrame f, int* out)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t puLds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + wave;
    if (idx >= n)
        return;
    const BlockJob& job = jobs[idx];
    XG_LDS uint32_t* rec = (XG_LDS uint32_t*)puLds + wave * (f.maxPasses * 64 * PU_REC_DW);
    PuCtx<P> c;
    c.ref = as_global<P>(f.ref);
    c.stride = f.stride;
    c.x = job.x;
    c.y = job.y;
    c.w = job.w;
    c.h = job.h;
    c.minX = -f.marginX + 4;
    c.maxX = f.width + f.marginX - job.w - 16;
    c.minY = -f.marginY + 4;
    c.maxY = f.height + f.marginY - job.h - 5;
    c.lane = lane;
    c.phys = lane;
    c.lpp = 64;
    c.rec = rec;
    c.depth = f.depth;
    int cost;
    if (job.op == 2)
    {
        const int size = job.w, grp = lane >> 3, r = lane & 7;
        const XG_GLOBAL P* fenc = as_global<P>(f.fenc) + ((long long)job.y * f.stride + job.x);
        const XG_GLOBAL P* ref = block_at<P>(c, job.qmvx >> 2, job.qmvy >> 2);
        if (size == 8)
        {
            const int s = had8x8_abs<P>(load_row_g<P>(fenc, r * f.stride), load_row_g<P>(ref, r * f.stride), r);
            cost = (group_value<0>(s) + 2) >> 2;
        }
        else
        {
            // two 16x16 tiles per pass: groups 0..3 and 4..7 hold a tile's four 8x8 blocks
            const int tiles = (size >> 4) * (size >> 4);
            cost = 0;
#pragma unroll 1
            for (int t0 = 0; t0 < tiles; t0 += 2)
            {
                const int t = min(t0 + (grp >> 2), tiles - 1);
                const int tx = t % (size >> 4), ty = t / (size >> 4);
                const int off = (16 * ty + 8 * ((grp >> 1) & 1) + r) * f.stride + 16 * tx + 8 * (grp & 1);
                const int s = had8x8_abs<P>(load_row_g<P>(fenc, off), load_row_g<P>(ref, off), r);
                cost += (group_value<0>(s) + group_value<1>(s) + group_value<2>(s) + group_value<3>(s) + 2) >> 2;
                if (t0 + 1 < tiles)
                    cost += (group_value<4>(s) + group_value<5>(s) + group_value<6>(s) + group_value<7>(s) + 2) >> 2;
            }
        }
    }
    else
    {
        stage_encode_block<P>(c, as_global<P>(f.fenc), f.stride, rec);
        cost = job.op == 1 ? block_cost<P, true>(c, job.qmvx, job.qmvy) : block_cost<P, false>(c, job.qmvx, job.qmvy);
    }
    if (lane == 0)
        out[idx] = cost;
}

// One wave per job: SATD of a PU against the average of two predictions, one from each of two reference pictures at
// its own quarter-pel vector - the bidirectional candidate of Search::puMotionEstimation (source/encoder/search.cpp:
// 440-448: Predict::predInterLumaPixel twice, pixelavg_pp with weight 32 = (a + b + 1) >> 1, bufSATD; and :482-486,
// the same average at the zero vectors).
struct BipredJob { int x, y, w, h; int qmv0x, qmv0y, qmv1x, qmv1y; };     // same layout as x265gpu_bipred_job

template <typename P>
__global__ void __launch_bounds__(256) bipred_satd_kernel(const BipredJob* jobs, int n, PuFrame f, const void* ref1, int* out)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t puLds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + wave;
    if (idx >= n)
        return;
    const BipredJob& job = jobs[idx];
    XG_LDS uint32_t* rec = (XG_LDS uint32_t*)puLds + wave * (f.maxPasses * 64 * PU_REC_DW);
    PuCtx<P> c;
    c.ref = as_global<P>(f.ref);
    c.stride = f.stride;
    c.x = job.x;
    c.y = job.y;
    c.w = job.w;
    c.h = job.h;
    c.minX = -f.marginX + 4;
    c.maxX = f.width + f.marginX - job.w - 16;
    c.minY = -f.marginY + 4;
    c.maxY = f.height + f.marginY - job.h - 5;
    c.lane = lane;
    c.phys = lane;
    c.lpp = 64;
    c.rec = rec;
    c.depth = f.depth;
    stage_encode_block<P>(c, as_global<P>(f.fenc), f.stride, rec);
    const XG_GLOBAL P* base0 = block_at<P>(c, job.qmv0x >> 2, job.qmv0y >> 2);
    const XG_GLOBAL P* base1 = as_global<P>(ref1) + (block_at<P>(c, job.qmv1x >> 2, job.qmv1y >> 2) - c.ref);
    const int xf0 = job.qmv0x & 3, yf0 = job.qmv0y & 3, xf1 = job.qmv1x & 3, yf1 = job.qmv1y & 3;
    uint32_t acc = 0;
#pragma unroll 1
    for (int k = 0; k < c.passes; k++)
    {
        const XG_LDS uint32_t* r = c.rec + (k * 64 + c.phys) * PU_REC_DW;
        const uint32_t f0 = r[0], f1 = r[1], f2 = r[2], f3 = r[3], mask = r[5];
        const int off = (int)r[4];
        const Row<P> row0 = (xf0 | yf0) ? interp_row<P>(base0 + off, c.stride, xf0, yf0, c.depth) : load_row_g<P>(base0, off);
        const Row<P> row1 = (xf1 | yf1) ? interp_row<P>(base1 + off, c.stride, xf1, yf1, c.depth) : load_row_g<P>(base1, off);
        int a[8], b[8];
        unpack_row<P>(row0, a);
        unpack_row<P>(row1, b);
#pragma unroll
        for (int i = 0; i < 8; i++)
            a[i] = (a[i] + b[i] + 1) >> 1;
        const Cols d = pair_cols<P>(pack_row<P>(a));
        acc += (uint32_t)satd8x4_cols(as_s16x2(f0) - as_s16x2(d.q[0] & mask), as_s16x2(f1) - as_s16x2(d.q[1] & mask),
                                      as_s16x2(f2) - as_s16x2(d.q[2] & mask), as_s16x2(f3) - as_s16x2(d.q[3] & mask), lane & 3);
    }
    acc = (lane & 3) ? 0u : acc;
    const int cost = wave_sum((int)acc);
    if (lane == 0)
        out[idx] = cost;
}

hipError_t launch_bipred_satd(hipStream_t s, int bpp, const PuFrame& f, const void* ref1, const void* dJobs, int n, int* dOut)
{
    if (n <= 0)
        return hipSuccess;
    const size_t lds = (size_t)4 * f.maxPasses * 64 * PU_REC_DW * sizeof(uint32_t);
    dim3 grid((n + 3) / 4);
    if (bpp == 1)
        hipLaunchKernelGGL(bipred_satd_kernel<uint8_t>, grid, dim3(256), lds, s, (const BipredJob*)dJobs, n, f, ref1, dOut);
    else
        hipLaunchKernelGGL(bipred_satd_kernel<uint16_t>, grid, dim3(256), lds, s, (const BipredJob*)dJobs, n, f, ref1, dOut);
    return hipGetLastError();
}

// luma_hpp / luma_vpp / luma_hvpp of one block (filter_pp_t, filter_hv_pp_t): one thread per
// 8-pixel row segment, output tight (stride = w)
template <typename P>
__global__ void __launch_bounds__(256) interp_block_kernel(const P* pic, int stride, int x, int y, int w, int h, int xf,
                                                           int yf, int depth, P* out)
{
    const int segs = (w + 7) >> 3;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= segs * h)
        return;
    const int row = i / segs, seg = i - row * segs;
    const XG_GLOBAL P* p = (const XG_GLOBAL P*)pic + ((long long)(y + row) * stride + x + 8 * seg);
    Row<P> r;
    if (xf | yf)
        r = interp_row<P>(p, stride, xf, yf, depth);
    else
        r = load_row_g<P>(p, 0);
    int v[8];
    unpack_row<P>(r, v);
    for (int k = 0; k < 8 && 8 * seg + k < w; k++)
        out[(long long)row * w + 8 * seg + k] = (P)v[k];
}

// (src coordinates clamped into the picture) = PicYuv padding + extendPicBorder
// (source/common/picyuv.cpp:340-400, source/common/pixel.cpp:1044-1058)
template <typename P>
__global__ void __launch_bounds__(256) pad_picture_kernel(const P* src, long long srcStride, P* dst, int stride, int rows,
                                                          int width, int height, int marginX, int marginY)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= stride || y >= rows)
        return;
    const int sx = clampi(x - marginX, 0, width - 1), sy = clampi(y - marginY, 0, height - 1);
    dst[(long long)y * stride + x] = src[(long long)sy * srcStride + sx];
}

hipError_t launch_pad_picture(hipStream_t s, int bpp, const void* src, long long srcStride, void* dst, int stride, int rows,
                              int width, int height, int marginX, int marginY)
{
    dim3 grid((stride + 255) / 256, rows);
    if (bpp == 1)
        hipLaunchKernelGGL(pad_picture_kernel<uint8_t>, grid, dim3(256), 0, s, (const uint8_t*)src, srcStride, (uint8_t*)dst,
                           stride, rows, width, height, marginX, marginY);
    else
        hipLaunchKernelGGL(pad_picture_kernel<uint16_t>, grid, dim3(256), 0, s, (const uint16_t*)src, srcStride,
                           (uint16_t*)dst, stride, rows, width, height, marginX, marginY);
    return hipGetLastError();
}

// ---- block-sum planes for the successive-elimination search ---------------------------
// What FrameFilter::computeMEIntegral leaves in FrameData::m_meIntegral once a picture is finished
// (source/encoder/framefilter.cpp:731-831): entry (x, y) of plane k is the sum of the bw x bh block of the padded
// reference picture whose top-left pixel is (x, y).  Two separable passes over the whole padded plane.
template <typename P>
__global__ void __launch_bounds__(256) sea_row_sums_kernel(const P* pic, int stride, int rows, int bw, uint32_t* out)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= stride || y >= rows)
        return;
    uint32_t v = 0;
    if (x + bw <= stride)
        for (int k = 0; k < bw; k++)
            v += pic[(long long)y * stride + x + k];
    out[(long long)y * stride + x] = v;
}

__global__ void __launch_bounds__(256) sea_col_sums_kernel(const uint32_t* rowSums, int stride, int rows, int bh, uint32_t* out)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= stride || y >= rows)
        return;
    uint32_t v = 0;
    if (y + bh <= rows)
        for (int r = 0; r < bh; r++)
            v += rowSums[(long long)(y + r) * stride + x];
    out[(long long)y * stride + x] = v;
}

// planes[k] (k in mask) <- block sums of the padded picture starting at `pic`; `tmp` holds one plane
hipError_t launch_sea_planes(hipStream_t s, int bpp, const void* pic, int stride, int rows, unsigned mask, uint32_t* tmp,
                             uint32_t* const planes[SEA_PLANES])
{
    dim3 grid((stride + 255) / 256, rows);
    int lastW = 0;
    for (int k = 0; k < SEA_PLANES; k++)
    {
        if (!(mask & (1u << k)))
            continue;
        int bw, bh;
        sea_plane_size(k, bw, bh);
        if (bw != lastW)       // planes of one width follow each other and share the row pass
        {
            if (bpp == 1)
                hipLaunchKernelGGL(sea_row_sums_kernel<uint8_t>, grid, dim3(256), 0, s, (const uint8_t*)pic, stride, rows, bw, tmp);
            else
                hipLaunchKernelGGL(sea_row_sums_kernel<uint16_t>, grid, dim3(256), 0, s, (const uint16_t*)pic, stride, rows, bw, tmp);
            lastW = bw;
        }
        hipLaunchKernelGGL(sea_col_sums_kernel, grid, dim3(256), 0, s, (const uint32_t*)tmp, stride, rows, bh, planes[k]);
    }
    return hipGetLastError();
}

template <typename P, int LPP>
static void launch_pu_class(hipStream_t s, const PuFrame& f, const PuJob* dJobs, const int* dOrder, int n, int maxPasses,
                            const uint16_t* costTab, int maxMvd, PuResult* dOut)
{
    if (n <= 0)
        return;
    PuFrame fc = f;
    fc.maxPasses = maxPasses;
    const size_t lds = (size_t)4 * maxPasses * 64 * PU_REC_DW * sizeof(uint32_t);
    const int perWave = 64 / LPP;
    const int waves = (n + perWave - 1) / perWave;
    hipLaunchKernelGGL((pu_search_kernel<P, LPP>), dim3((waves + 3) / 4), dim3(256), lds, s, dJobs, dOrder, n, fc, costTab,
                       maxMvd, dOut);
}

// Stream plan: the wave-per-PU classes on `big` (the longest-running waves first), the eight- and four-per-wave
// classes on `small`, the two-per-wave class on `third`.
template <typename P>
static void launch_pu_classes(hipStream_t small, hipStream_t big, hipStream_t third, const PuFrame& f, const PuJob* dJobs,
                              const PuClasses& k, const uint16_t* costTab, int maxMvd, PuResult* dOut)
{
    const int* o0 = k.order;
    const int* o1 = o0 + k.count[0];
    const int* o2 = o1 + k.count[1];
    const int* o3 = o2 + k.count[2];
    const int* o4 = o3 + k.count[3];
    launch_pu_class<P, 64>(big, f, dJobs, o4, k.count[4], k.maxPasses[1], costTab, maxMvd, dOut);
    launch_pu_class<P, 8>(small, f, dJobs, o0, k.count[0], 1, costTab, maxMvd, dOut);
    launch_pu_class<P, 32>(third, f, dJobs, o2, k.count[2], 1, costTab, maxMvd, dOut);
    launch_pu_class<P, 16>(small, f, dJobs, o1, k.count[1], 1, costTab, maxMvd, dOut);
    launch_pu_class<P, 64>(big, f, dJobs, o3, k.count[3], k.maxPasses[0], costTab, maxMvd, dOut);
}

// The size classes are independent: on side streams the shared-wave classes run beside the
// wave-per-PU classes and fill each other's tails.
hipError_t launch_pu_search(hipStream_t s, const SearchStreams& aux, int bpp, const PuFrame& f, const PuJob* dJobs,
                            const PuClasses& classes, const uint16_t* costTab, int maxMvd, PuResult* dOut)
{
    const int sides = aux.count > 2 ? 2 : aux.count;
    hipStream_t small = sides > 0 ? aux.side[0] : s, third = sides > 1 ? aux.side[1] : s;
    for (int k = 0; k < sides; k++)
    {
        hipError_t e = k ? hipSuccess : hipEventRecord(aux.fork, s);
        if (e == hipSuccess)
            e = hipStreamWaitEvent(aux.side[k], aux.fork, 0);
        if (e != hipSuccess)
            return e;
    }
    if (bpp == 1)
        launch_pu_classes<uint8_t>(small, s, third, f, dJobs, classes, costTab, maxMvd, dOut);
    else
        launch_pu_classes<uint16_t>(small, s, third, f, dJobs, classes, costTab, maxMvd, dOut);
    hipError_t e = hipGetLastError();
    for (int k = 0; k < sides && e == hipSuccess; k++)
    {
        e = hipEventRecord(aux.join[k], aux.side[k]);
        if (e == hipSuccess)
            e = hipStreamWaitEvent(s, aux.join[k], 0);
    }
    return e;
}

hipError_t launch_block_compare(hipStream_t s, int bpp, const PuFrame& f, const BlockJob* dJobs, int n, int* dOut)
{
    if (n <= 0)
        return hipSuccess;
    const size_t lds = (size_t)4 * f.maxPasses * 64 * PU_REC_DW * sizeof(uint32_t);
    dim3 grid((n + 3) / 4);
    if (bpp == 1)
        hipLaunchKernelGGL(block_compare_kernel<uint8_t>, grid, dim3(256), lds, s, dJobs, n, f, dOut);
    else
        hipLaunchKernelGGL(block_compare_kernel<uint16_t>, grid, dim3(256), lds, s, dJobs, n, f, dOut);
    return hipGetLastError();
}

hipError_t launch_interp_block(hipStream_t s, int bpp, const void* pic, int stride, int x, int y, int w, int h, int xf, int yf,
                               int depth, void* dOut)
{
    const int threads = ((w + 7) >> 3) * h;
    dim3 grid((threads + 255) / 256);
    if (bpp == 1)
        hipLaunchKernelGGL(interp_block_kernel<uint8_t>, grid, dim3(256), 0, s, (const uint8_t*)pic, stride, x, y, w, h, xf,
                           yf, depth, (uint8_t*)dOut);
    else
        hipLaunchKernelGGL(interp_block_kernel<uint16_t>, grid, dim3(256), 0, s, (const uint16_t*)pic, stride, x, y, w, h,
                           xf, yf, depth, (uint16_t*)dOut);
    return hipGetLastError();
}

} // namespace xg
