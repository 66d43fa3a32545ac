// Motion search of the motion-compensated spatio-temporal filter (--mcstf): a four-level
// pyramid of exhaustive +-3 searches by sum of squared differences, the last level in 1/16
// pixel with a 6-tap separable interpolation.
//
// Replaces MotionEstimatorTLD::motionEstimationLuma / ::motionEstimationLumaDoubleRes /
// ::motionErrorLumaSSD (source/common/temporalfilter.cpp:38-91, 421-478, 642-897) as the
// lookahead runs them: 16-line row tasks, one batch per level (source/encoder/slicetype.cpp:
// 2280-2297, 4217-4234).
//
// A block's own search (the coarser level's nine vectors, the zero vector, the +-3 square)
// does not depend on its neighbours; only the closing "try the left block's vector" (and, on
// the last level, the vector of the block above inside the same row task) does, and that
// candidate is accepted only if strictly better.  So each level is two kernels: one wave per
// block for the search proper, then one wave per row (task) that walks the row left to right
// and measures a neighbour's vector only where it differs from the block's own.
//
// The reference stops summing a candidate once it exceeds the best error so far; the partial
// sum it then returns is never accepted, so whole sums give the same decisions.
#include "device_common.h"
#include "kernels.h"

#define XG_LDS __attribute__((address_space(3)))

namespace xg {

// s_interpolationFilter taps 1..6 (source/common/temporalfilter.h:36-54)
__constant__ int kMcstfFilter[16][6] = {
    { 0,   0, 64,  0,   0, 0 }, { 1,  -3, 64,  4,  -2, 0 }, { 1,  -6, 62,  9,  -3, 1 }, { 2,  -8, 60, 14,  -5, 1 },
    { 2,  -9, 57, 19,  -7, 2 }, { 3, -10, 53, 24,  -8, 2 }, { 3, -11, 50, 29,  -9, 2 }, { 3, -11, 44, 35, -10, 3 },
    { 1,  -7, 38, 38,  -7, 1 }, { 3, -10, 35, 44, -11, 3 }, { 2,  -9, 29, 50, -11, 3 }, { 2,  -8, 24, 53, -10, 3 },
    { 2,  -7, 19, 57,  -9, 2 }, { 1,  -5, 14, 60,  -8, 2 }, { 1,  -3,  9, 62,  -6, 1 }, { 0,  -2,  4, 64,  -3, 1 },
};

constexpr int MV_FACTOR = 16;      // m_motionVectorFactor (temporalfilter.h:121)
constexpr int MCSTF_RANGE = 3;     // m_searchRange; also doubleRange of the last level

__device__ __forceinline__ int wave_sum(int v)
{
    v += dpp<DPP_XOR1>(v);
    v += dpp<DPP_XOR2>(v);
    v += dpp<DPP_HALF_MIRROR>(v);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);     // row_mirror: the row's other half
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);    // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);    // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

// C's truncating division by 16 for possibly negative vectors (prevBest.x / m_motionVectorFactor)
__device__ __forceinline__ int div16(int v) { return v / MV_FACTOR; }

template <typename P> __device__ __forceinline__ void load4(const XG_GLOBAL P* p, int v[4]);
template <> __device__ __forceinline__ void load4<uint8_t>(const XG_GLOBAL uint8_t* p, int v[4])
{
    typedef uint32_t u32_unaligned __attribute__((aligned(1)));
    const uint32_t w = *(const XG_GLOBAL u32_unaligned*)p;
    v[0] = (int)(w & 0xffu);
    v[1] = (int)((w >> 8) & 0xffu);
    v[2] = (int)((w >> 16) & 0xffu);
    v[3] = (int)(w >> 24);
}
template <> __device__ __forceinline__ void load4<uint16_t>(const XG_GLOBAL uint16_t* p, int v[4])
{
    const u32x2_unaligned w = *(const XG_GLOBAL u32x2_unaligned*)p;
    v[0] = (int)(w.x & 0xffffu);
    v[1] = (int)(w.x >> 16);
    v[2] = (int)(w.y & 0xffffu);
    v[3] = (int)(w.y >> 16);
}

// ---- 16x16 blocks, full-pel vectors (levels 0..2) ---------------------------------------------
// lane = (row, four columns); the vector is in 1/16 pixel and a multiple of 16
template <typename P>
__device__ __forceinline__ int error16(const int o[4], const XG_GLOBAL P* buf, int stride, int x, int y, int mvx, int mvy,
                                       int lane)
{
    int b[4];
    load4<P>(buf + (long long)(y + mvy / MV_FACTOR + (lane >> 2)) * stride + x + mvx / MV_FACTOR + 4 * (lane & 3), b);
    int e = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const int d = o[i] - b[i];
        e += d * d;
    }
    return wave_sum(e);
}

template <typename P>
__global__ void __launch_bounds__(256) mcstf_search16_kernel(McstfLevel lv, const McstfRef* refs)
{
    const int lane = threadIdx.x & 63;
    const int blk = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk >= lv.blocksX * lv.blocksY)
        return;
    const McstfRef rf = refs[blockIdx.y];
    const int bxi = blk % lv.blocksX, byi = blk / lv.blocksX;
    const int x = bxi * 16, y = byi * 16;
    const XG_GLOBAL P* src = as_global<P>(lv.src);
    const XG_GLOBAL P* buf = as_global<P>(rf.buf);
    int o[4];
    load4<P>(src + (long long)(y + (lane >> 2)) * lv.stride + x + 4 * (lane & 3), o);

    int bestx = 0, besty = 0, least = 0x7fffffff;
    if (rf.prev)
    {
        // the nine surrounding vectors of the coarser level, then the zero vector (temporalfilter.cpp:676-711)
        const XG_GLOBAL int32_t* prev = as_global<int32_t>(rf.prev);
        for (int py = -1; py <= 1; py++)
        {
            const int testy = y / 32 + py;
            for (int px = -1; px <= 1; px++)
            {
                const int testx = x / 32 + px;
                if (testx >= 0 && testx < lv.width / 32 && testy >= 0 && testy < lv.height / 32)
                {
                    const int ox = prev[2 * (testy * lv.prevStride + testx)] * lv.factor;
                    const int oy = prev[2 * (testy * lv.prevStride + testx) + 1] * lv.factor;
                    const int e = error16<P>(o, buf, lv.stride, x, y, ox, oy, lane);
                    if (e < least)
                    {
                        bestx = ox;
                        besty = oy;
                        least = e;
                    }
                }
            }
        }
        const int e = error16<P>(o, buf, lv.stride, x, y, 0, 0, lane);
        if (e < least)
        {
            bestx = besty = 0;
            least = e;
        }
    }
    const int cx = div16(bestx), cy = div16(besty);
    for (int y2 = cy - MCSTF_RANGE; y2 <= cy + MCSTF_RANGE; y2++)
#pragma unroll
        for (int x2 = -MCSTF_RANGE; x2 <= MCSTF_RANGE; x2++)
        {
            const int e = error16<P>(o, buf, lv.stride, x, y, (cx + x2) * MV_FACTOR, y2 * MV_FACTOR, lane);
            if (e < least)
            {
                bestx = (cx + x2) * MV_FACTOR;
                besty = y2 * MV_FACTOR;
                least = e;
            }
        }
    if (lane == 0)
    {
        const int idx = byi * lv.mvStride + bxi;
        rf.mvs[2 * idx] = bestx;
        rf.mvs[2 * idx + 1] = besty;
        rf.least[idx] = least;
    }
}

// one wave per block row: the left block's final vector, tried last (temporalfilter.cpp:730-742)
template <typename P>
__global__ void __launch_bounds__(64) mcstf_chain16_kernel(McstfLevel lv, const McstfRef* refs)
{
    const int lane = threadIdx.x;
    const int byi = blockIdx.x;
    const McstfRef rf = refs[blockIdx.y];
    const XG_GLOBAL P* src = as_global<P>(lv.src);
    const XG_GLOBAL P* buf = as_global<P>(rf.buf);
    const XG_GLOBAL int32_t* mvs = as_global<int32_t>(rf.mvs);
    const int row = byi * lv.mvStride;
    int leftx = mvs[2 * row], lefty = mvs[2 * row + 1];
    for (int bxi = 1; bxi < lv.blocksX; bxi++)
    {
        const int idx = row + bxi;
        int mx = mvs[2 * idx], my = mvs[2 * idx + 1];
        if (mx != leftx || my != lefty)        // the block's own vector measured again cannot be strictly better
        {
            int o[4];
            load4<P>(src + (long long)(byi * 16 + (lane >> 2)) * lv.stride + bxi * 16 + 4 * (lane & 3), o);
            const int e = error16<P>(o, buf, lv.stride, bxi * 16, byi * 16, leftx, lefty, lane);
            if (e < rf.least[idx])
            {
                mx = leftx;
                my = lefty;
                if (lane == 0)
                {
                    rf.mvs[2 * idx] = mx;
                    rf.mvs[2 * idx + 1] = my;
                    rf.least[idx] = e;
                }
            }
        }
        leftx = mx;
        lefty = my;
    }
}

// ---- 8x8 blocks, vectors in 1/16 pixel (last level) ---------------------------------------------
// lane = (y1, x1), one pixel.  A candidate set whose integer parts span at most two
// consecutive columns and rows is measured out of a 14x14 window staged in LDS: window
// column c is picture column x + (dxMin >> 4) - 2 + c (rows alike), which covers taps
// 1..6 of every candidate.  A full-pel vector goes through the same arithmetic: phase 0 is
// the single tap 64, and ((64 * 64 * p + 2048) >> 12) = p, so it equals the reference's
// plain sum of squared differences.
constexpr int WIN8 = 14, WIN8_S = 15;     // odd row stride: rows fall into different banks
struct Lds8
{
    int win[WIN8 * WIN8_S];
    int tmp[WIN8 * 8];
};

template <typename P>
__device__ __forceinline__ void stage8(XG_LDS Lds8* l, const XG_GLOBAL P* buf, int stride, int x, int y, int dxMin, int dyMin,
                                       int lane)
{
    const int wx = x + (dxMin >> 4) - 2, wy = y + (dyMin >> 4) - 2;
    for (int i = lane; i < WIN8 * WIN8; i += 64)
    {
        const int r = i / WIN8, c = i - r * WIN8;
        l->win[r * WIN8_S + c] = (int)buf[(long long)(wy + r) * stride + wx + c];
    }
    lds_handover();
}

// horizontal pass of phase dx & 15 for all 14 window rows; colOff = (dx >> 4) - (dxMin >> 4)
__device__ __forceinline__ void hpass8(XG_LDS Lds8* l, int dx, int colOff, int lane)
{
    const int* f = kMcstfFilter[dx & 15];
    for (int i = lane; i < WIN8 * 8; i += 64)
    {
        const int r = i >> 3, x1 = i & 7;
        const XG_LDS int* p = l->win + r * WIN8_S + x1 + colOff;
        int s = 0;
#pragma unroll
        for (int k = 0; k < 6; k++)
            s += f[k] * p[k];
        l->tmp[i] = s;
    }
    lds_handover();
}

// vertical pass, rounding, clipping and the squared difference against this lane's source pixel
__device__ __forceinline__ int vpass8(const XG_LDS Lds8* l, int dy, int rowOff, int o, int maxVal, int lane)
{
    const int* f = kMcstfFilter[dy & 15];
    const XG_LDS int* p = l->tmp + ((lane >> 3) + rowOff) * 8 + (lane & 7);
    int s = 0;
#pragma unroll
    for (int k = 0; k < 6; k++)
        s += f[k] * p[8 * k];
    s = (s + (1 << 11)) >> 12;
    s = min(max(s, 0), maxVal);
    const int d = s - o;
    return wave_sum(d * d);
}

// one candidate on its own
template <typename P>
__device__ __forceinline__ int error8(XG_LDS Lds8* l, const XG_GLOBAL P* buf, int stride, int x, int y, int dx, int dy, int o,
                                      int maxVal, int lane)
{
    lds_handover();
    stage8<P>(l, buf, stride, x, y, dx, dy, lane);
    hpass8(l, dx, 0, lane);
    return vpass8(l, dy, 0, o, maxVal, lane);
}

template <typename P>
__global__ void __launch_bounds__(256) mcstf_search8_kernel(McstfLevel lv, const McstfRef* refs)
{
    __shared__ Lds8 lds[4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = blockIdx.x * 4 + wave;
    if (blk >= lv.blocksX * lv.blocksY)
        return;
    XG_LDS Lds8* l = (XG_LDS Lds8*)&lds[wave];
    const McstfRef rf = refs[blockIdx.y];
    const int bxi = blk % lv.blocksX, byi = blk / lv.blocksX;
    const int x = bxi * 8, y = byi * 8;
    const XG_GLOBAL P* src = as_global<P>(lv.src);
    const XG_GLOBAL P* buf = as_global<P>(rf.buf);
    const int o = (int)src[(long long)(y + (lane >> 3)) * lv.stride + x + (lane & 7)];
    const int maxVal = (1 << lv.depth) - 1;

    int bestx = 0, besty = 0, least = 0x7fffffff;
    auto tryVector = [&](int ox, int oy) {
        int e;
        if (((ox | oy) & 15) == 0)
        {
            // full-pel: one pixel per lane straight from memory
            const int b = (int)buf[(long long)(y + oy / MV_FACTOR + (lane >> 3)) * lv.stride + x + ox / MV_FACTOR + (lane & 7)];
            e = wave_sum((o - b) * (o - b));
        }
        else
            e = error8<P>(l, buf, lv.stride, x, y, ox, oy, o, maxVal, lane);
        if (e < least)
        {
            bestx = ox;
            besty = oy;
            least = e;
        }
    };
    // the coarser level's nine vectors, then the zero vector (temporalfilter.cpp:786-821)
    const XG_GLOBAL int32_t* prev = as_global<int32_t>(rf.prev);
    for (int py = -1; py <= 1; py++)
    {
        const int testy = y / 16 + py;
        for (int px = -1; px <= 1; px++)
        {
            const int testx = x / 16 + px;
            if (testx >= 0 && testx < lv.width / 16 && testy >= 0 && testy < lv.height / 16)
                tryVector(prev[2 * (testy * lv.prevStride + testx)] * lv.factor,
                          prev[2 * (testy * lv.prevStride + testx) + 1] * lv.factor);
        }
    }
    tryVector(0, 0);
    // range 0 when a coarser level exists (temporalfilter.cpp:755, 782-785, 823-836): the best so far
    // truncated to full pel
    {
        const int fx = div16(bestx) * MV_FACTOR, fy = div16(besty) * MV_FACTOR;
        if (fx != bestx || fy != besty)      // the best vector itself cannot be strictly better than itself
            tryVector(fx, fy);
    }
    // +-3 sixteenths around it, rows outside, columns inside (temporalfilter.cpp:838-852)
    {
        const int px = bestx, py = besty;
        const int dxMin = px - MCSTF_RANGE, dyMin = py - MCSTF_RANGE;
        lds_handover();
        stage8<P>(l, buf, lv.stride, x, y, dxMin, dyMin, lane);
        // the reference walks y2 outside and x2 inside; the horizontal pass depends on x2 only, so
        // measure column by column and choose afterwards in the reference's order
        int e49[49];
#pragma unroll
        for (int ix = 0; ix < 7; ix++)
        {
            const int dx = dxMin + ix;
            lds_handover();
            hpass8(l, dx, (dx >> 4) - (dxMin >> 4), lane);
#pragma unroll
            for (int iy = 0; iy < 7; iy++)
            {
                const int dy = dyMin + iy;
                e49[iy * 7 + ix] = vpass8(l, dy, (dy >> 4) - (dyMin >> 4), o, maxVal, lane);
            }
        }
#pragma unroll
        for (int i = 0; i < 49; i++)
            if (e49[i] < least)
            {
                least = e49[i];
                bestx = dxMin + i % 7;
                besty = dyMin + i / 7;
            }
    }
    if (lane == 0)
    {
        const int idx = byi * lv.mvStride + bxi;
        rf.mvs[2 * idx] = bestx;
        rf.mvs[2 * idx + 1] = besty;
        rf.least[idx] = least;
    }
}

// The closing candidates of the last level are the FINAL vectors of the left block and, inside a
// row task, of the block above (temporalfilter.cpp:854-881): a chain along the row.  Most of the
// time a block's final vector is the one its own search found, so this pass measures, for every
// block at once, the errors of its left and above neighbours' own-search vectors and keeps a copy
// of those vectors; the chain below then only has to compare numbers unless a neighbour's vector
// was itself replaced on the way.  spec[4 * idx ..] = { own x, own y, error of left's, error of above's }.
template <typename P>
__global__ void __launch_bounds__(256) mcstf_spec8_kernel(McstfLevel lv, const McstfRef* refs)
{
    __shared__ Lds8 lds[4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = blockIdx.x * 4 + wave;
    if (blk >= lv.blocksX * lv.blocksY)
        return;
    XG_LDS Lds8* l = (XG_LDS Lds8*)&lds[wave];
    const McstfRef rf = refs[blockIdx.y];
    const int bxi = blk % lv.blocksX, byi = blk / lv.blocksX;
    const int idx = byi * lv.mvStride + bxi;
    const XG_GLOBAL P* src = as_global<P>(lv.src);
    const XG_GLOBAL P* buf = as_global<P>(rf.buf);
    const XG_GLOBAL int32_t* mvs = as_global<int32_t>(rf.mvs);
    const int maxVal = (1 << lv.depth) - 1;
    const int o = (int)src[(long long)(byi * 8 + (lane >> 3)) * lv.stride + bxi * 8 + (lane & 7)];
    const int mx = mvs[2 * idx], my = mvs[2 * idx + 1];
    int eL = -1, eA = -1;
    if (bxi > 0)
    {
        const int vx = mvs[2 * (idx - 1)], vy = mvs[2 * (idx - 1) + 1];
        if (vx != mx || vy != my)
            eL = error8<P>(l, buf, lv.stride, bxi * 8, byi * 8, vx, vy, o, maxVal, lane);
    }
    if (byi & 1)
    {
        const int vx = mvs[2 * (idx - lv.mvStride)], vy = mvs[2 * (idx - lv.mvStride) + 1];
        if (vx != mx || vy != my)
            eA = error8<P>(l, buf, lv.stride, bxi * 8, byi * 8, vx, vy, o, maxVal, lane);
    }
    if (lane == 0)
    {
        rf.spec[4 * idx] = mx;
        rf.spec[4 * idx + 1] = my;
        rf.spec[4 * idx + 2] = eL;
        rf.spec[4 * idx + 3] = eA;
    }
}

// One wave per 16-line row task, 64 blocks of its two rows at a time: a lane holds one block's
// numbers, the walk reads them lane by lane.  A candidate is measured here only when the
// neighbour's final vector is not the one the pass above measured.
template <typename P>
__global__ void __launch_bounds__(64) mcstf_chain8_kernel(McstfLevel lv, const McstfRef* refs)
{
    __shared__ Lds8 lds;
    XG_LDS Lds8* l = (XG_LDS Lds8*)&lds;
    const int lane = threadIdx.x;
    const McstfRef rf = refs[blockIdx.y];
    const XG_GLOBAL P* src = as_global<P>(lv.src);
    const XG_GLOBAL P* buf = as_global<P>(rf.buf);
    const XG_GLOBAL int32_t* spec = as_global<int32_t>(rf.spec);
    const XG_GLOBAL int32_t* leastIn = as_global<int32_t>(rf.least);
    const int maxVal = (1 << lv.depth) - 1;
    const int rows = min(2, lv.blocksY - 2 * (int)blockIdx.x);
    int leftx[2] = { 0, 0 }, lefty[2] = { 0, 0 };          // final vector of the previous block of each row
    int leftOwnX[2] = { 0, 0 }, leftOwnY[2] = { 0, 0 };    // and what its own search had found
    for (int x0 = 0; x0 < lv.blocksX; x0 += 64)
    {
        const int cnt = min(64, lv.blocksX - x0);
        int finx0 = 0, finy0 = 0;                           // row 0's final vectors of this chunk, one per lane
        for (int sub = 0; sub < rows; sub++)
        {
            const int byi = 2 * blockIdx.x + sub;
            const int idx0 = byi * lv.mvStride + x0;
            int ownx = 0, owny = 0, eL = -1, eA = -1, least = 0;
            if (lane < cnt)
            {
                ownx = spec[4 * (idx0 + lane)];
                owny = spec[4 * (idx0 + lane) + 1];
                eL = spec[4 * (idx0 + lane) + 2];
                eA = spec[4 * (idx0 + lane) + 3];
                least = leastIn[idx0 + lane];
            }
            // what the block above found on its own: valid for eA only if that is still its final vector
            int aboveOwnX = 0, aboveOwnY = 0;
            if (sub && lane < cnt)
            {
                aboveOwnX = spec[4 * (idx0 - lv.mvStride + lane)];
                aboveOwnY = spec[4 * (idx0 - lv.mvStride + lane) + 1];
            }
            int outx = ownx, outy = owny, outLeast = least;
            for (int i = 0; i < cnt; i++)
            {
                const int bxi = x0 + i;
                const int ox = __builtin_amdgcn_readlane(ownx, i), oy = __builtin_amdgcn_readlane(owny, i);
                const int ownLeast = __builtin_amdgcn_readlane(least, i);
                int mx = ox, my = oy, cur = ownLeast;
                int o = 0;
                bool haveO = false;
                auto measure = [&](int vx, int vy) {
                    if (!haveO)
                    {
                        o = (int)src[(long long)(byi * 8 + (lane >> 3)) * lv.stride + bxi * 8 + (lane & 7)];
                        haveO = true;
                    }
                    return error8<P>(l, buf, lv.stride, bxi * 8, byi * 8, vx, vy, o, maxVal, lane);
                };
                if (sub)
                {
                    const int ax = __builtin_amdgcn_readlane(finx0, i), ay = __builtin_amdgcn_readlane(finy0, i);
                    if (ax != mx || ay != my)
                    {
                        const bool asMeasured = ax == __builtin_amdgcn_readlane(aboveOwnX, i) &&
                                                ay == __builtin_amdgcn_readlane(aboveOwnY, i);
                        const int e = asMeasured ? __builtin_amdgcn_readlane(eA, i) : measure(ax, ay);
                        if (e < cur)
                        {
                            mx = ax;
                            my = ay;
                            cur = e;
                        }
                    }
                }
                if (bxi > 0 && (leftx[sub] != mx || lefty[sub] != my))
                {
                    int e;
                    if (leftx[sub] == ox && lefty[sub] == oy)
                        e = ownLeast;                        // the block's own vector again (it lost to the one above)
                    else if (leftx[sub] == leftOwnX[sub] && lefty[sub] == leftOwnY[sub])
                        e = __builtin_amdgcn_readlane(eL, i);
                    else
                        e = measure(leftx[sub], lefty[sub]);
                    if (e < cur)
                    {
                        mx = leftx[sub];
                        my = lefty[sub];
                        cur = e;
                    }
                }
                outx = lane == i ? mx : outx;
                outy = lane == i ? my : outy;
                outLeast = lane == i ? cur : outLeast;
                leftx[sub] = mx;
                lefty[sub] = my;
                leftOwnX[sub] = ox;
                leftOwnY[sub] = oy;
            }
            if (lane < cnt)
            {
                rf.mvs[2 * (idx0 + lane)] = outx;
                rf.mvs[2 * (idx0 + lane) + 1] = outy;
                rf.least[idx0 + lane] = outLeast;
            }
            if (!sub)
            {
                finx0 = outx;
                finy0 = outy;
            }
        }
    }
}

// error = (int)(20 * ((least + 5.0) / (variance + 5.0)) + (least / 64) / 50) with the block's
// variance = sqr - sum * sum / 64 (temporalfilter.cpp:883-894); every product, quotient and sum
// rounded on its own, as the reference's scalar double code is
template <typename P>
__global__ void __launch_bounds__(256) mcstf_finish8_kernel(McstfLevel lv, const McstfRef* refs)
{
    const int lane = threadIdx.x & 63;
    const int blk = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk >= lv.blocksX * lv.blocksY)
        return;
    const McstfRef rf = refs[blockIdx.y];
    const int bxi = blk % lv.blocksX, byi = blk / lv.blocksX;
    const XG_GLOBAL P* src = as_global<P>(lv.src);
    const int o = (int)src[(long long)(byi * 8 + (lane >> 3)) * lv.stride + bxi * 8 + (lane & 7)];
    const unsigned sum = (unsigned)wave_sum(o), sqr = (unsigned)wave_sum(o * o);
    if (lane == 0)
    {
        const int idx = byi * lv.mvStride + bxi;
        const int least = rf.least[idx];
        const double variance = __dsub_rn((double)sqr, __ddiv_rn(__dmul_rn((double)sum, (double)sum), 64.0));
        const double q = __ddiv_rn(__dadd_rn((double)least, 5.0), __dadd_rn(variance, 5.0));
        rf.error[idx] = (int)__dadd_rn(__dmul_rn(20.0, q), (double)((least / 64) / 50));
    }
}

hipError_t launch_mcstf_level(hipStream_t s, int bpp, int level, const McstfLevel& lv, const McstfRef* dRefs, int numRefs)
{
    const int blocks = lv.blocksX * lv.blocksY;
    if (blocks <= 0 || numRefs <= 0)
        return hipSuccess;
    const dim3 gridBlocks((blocks + 3) / 4, numRefs);
    if (level < 3)
    {
        const dim3 gridRows(lv.blocksY, numRefs);
        if (bpp == 1)
        {
            hipLaunchKernelGGL(mcstf_search16_kernel<uint8_t>, gridBlocks, dim3(256), 0, s, lv, dRefs);
            hipLaunchKernelGGL(mcstf_chain16_kernel<uint8_t>, gridRows, dim3(64), 0, s, lv, dRefs);
        }
        else
        {
            hipLaunchKernelGGL(mcstf_search16_kernel<uint16_t>, gridBlocks, dim3(256), 0, s, lv, dRefs);
            hipLaunchKernelGGL(mcstf_chain16_kernel<uint16_t>, gridRows, dim3(64), 0, s, lv, dRefs);
        }
    }
    else
    {
        const dim3 gridTasks((lv.blocksY + 1) / 2, numRefs);
        if (bpp == 1)
        {
            hipLaunchKernelGGL(mcstf_search8_kernel<uint8_t>, gridBlocks, dim3(256), 0, s, lv, dRefs);
            hipLaunchKernelGGL(mcstf_spec8_kernel<uint8_t>, gridBlocks, dim3(256), 0, s, lv, dRefs);
            hipLaunchKernelGGL(mcstf_chain8_kernel<uint8_t>, gridTasks, dim3(64), 0, s, lv, dRefs);
            hipLaunchKernelGGL(mcstf_finish8_kernel<uint8_t>, gridBlocks, dim3(256), 0, s, lv, dRefs);
        }
        else
        {
            hipLaunchKernelGGL(mcstf_search8_kernel<uint16_t>, gridBlocks, dim3(256), 0, s, lv, dRefs);
            hipLaunchKernelGGL(mcstf_spec8_kernel<uint16_t>, gridBlocks, dim3(256), 0, s, lv, dRefs);
            hipLaunchKernelGGL(mcstf_chain8_kernel<uint16_t>, gridTasks, dim3(64), 0, s, lv, dRefs);
            hipLaunchKernelGGL(mcstf_finish8_kernel<uint16_t>, gridBlocks, dim3(256), 0, s, lv, dRefs);
        }
    }
    return hipGetLastError();
}

} // namespace xg
