// Lowres motion search: one launch per anti-diagonal of the 8x8 block grid.
//
// Replaces, per block, the search half of CostEstimateGroup::estimateCUCost
// (source/encoder/slicetype.cpp:4422-4491): predictor choice by SATD over the
// already-searched neighbours, MotionEstimate::motionEstimate
// (source/encoder/motion.cpp:923-1773) on the half-pel plane set with
// ReferencePlanes::lowresQPelCost sub-pel refinement, the fork's closing
// zero-vector check and the bidir skip shortcut.
//
// Dependencies: a block reads the vectors of its right, below, below-left and
// below-right neighbours (slicetype.cpp:4438-4449).  With the grid flipped
// (x' = W-1-cuX, y' = H-1-cuY) block (x',y') depends only on blocks with a
// smaller t = x' + 2y', so all blocks of one t are independent: launch t runs
// them for every job of the batch, and the kernel boundary is the only
// synchronisation.  Occupancy comes from batching many (frame, list, distance)
// jobs per launch, never from splitting a block's serial decision chain.
//
// One wave per block.  The wave keeps the reference's candidate ORDER and
// strict-'<' tie-breaks on wave-uniform scalars, but measures the candidates
// of one step (sad_x3 / sad_x4 / square refine / four sub-pel points) in
// parallel, one per 8-lane group.
#include "device_common.h"
#include "kernels.h"

namespace xg {

// pattern tables as biased nibbles (nib()): index 0 in the low nibble
constexpr uint32_t HEX2_X = 0x679A9767u;   // hex2[8].x = -1,-2,-1,1,2,1,-1,-2   (motion.cpp:64)
constexpr uint32_t HEX2_Y = 0x8668AA86u;   // hex2[8].y = -2,0,2,2,0,-2,-2,0
constexpr uint32_t HEX6_X = 0x8879A976u;   // first six points: (-2,0)(-1,2)(1,2)(2,0)(1,-2)(-1,-2)
constexpr uint32_t HEX6_Y = 0x88668AA8u;
constexpr uint32_t SQ1_X = 0x99779788u;    // square1[1..8].x = 0,0,-1,1,-1,-1,1,1      (motion.cpp:66)
constexpr uint32_t SQ1_Y = 0x97978897u;    // square1[1..8].y = -1,1,0,0,-1,1,-1,1
constexpr uint32_t MOD6M1 = 0x05432105u;   // mod6m1[8] = 5,0,1,2,3,4,5,0 (unbiased)          (motion.cpp:65)

template <typename P>
struct BlockCtx
{
    PlaneSet<P> ps;        // reference plane set being searched
    Row<P> fr;             // this lane's encode row, packed
    int f[8];              // and unpacked
    int bx, by;            // block origin in pixels
    int g, r;              // candidate group and row of this lane
    int minx, miny, maxx, maxy;   // full-pel search bounds (mvmin / mvmax)
    int mvpx, mvpy;        // quarter-pel predictor the MV cost is centred on
    const uint16_t* cost;  // centred MV-cost table
    int maxMvd;
};

// BitCost::mvcost (source/encoder/bitcost.h:56): uint16 sum of two table reads
template <typename P> __device__ __forceinline__ int mvcost_q(const BlockCtx<P>& c, int qx, int qy)
{
    const int ix = clampi(qx - c.mvpx, -c.maxMvd, c.maxMvd);
    const int iy = clampi(qy - c.mvpy, -c.maxMvd, c.maxMvd);
    return ((int)c.cost[ix] + (int)c.cost[iy]) & 0xffff;
}

template <typename P> __device__ __forceinline__ int cand_sad(const BlockCtx<P>& c, int qx, int qy, bool anyOdd)
{
    return sad8x8<P>(c.fr, mc_row<P>(c.ps, c.bx, c.by, c.r, qx, qy, anyOdd));
}

template <typename P> __device__ __forceinline__ int cand_satd(const BlockCtx<P>& c, int qx, int qy, bool anyOdd)
{
    return satd8x8<P>(c.f, mc_row<P>(c.ps, c.bx, c.by, c.r, qx, qy, anyOdd), c.r);
}

template <typename P> __device__ __forceinline__ bool y_ok(const BlockCtx<P>& c, int y) { return y >= c.miny && y <= c.maxy; }
template <typename P> __device__ __forceinline__ bool in_range(const BlockCtx<P>& c, int x, int y)
{
    return x >= c.minx && x <= c.maxx && y >= c.miny && y <= c.maxy;
}

// full-pel cost of this lane's group candidate: SAD + mvcost(mv << 2)
template <typename P> __device__ __forceinline__ int fpel_cost(const BlockCtx<P>& c, int mx, int my)
{
    return cand_sad<P>(c, mx * 4, my * 4, false) + mvcost_q<P>(c, mx * 4, my * 4);
}

// 8-point square refinement around (bmx,bmy): motion.cpp:1120-1138
template <typename P> __device__ void square_refine(const BlockCtx<P>& c, int& bmx, int& bmy, int& bcost)
{
    const int cost = fpel_cost<P>(c, bmx + nib(SQ1_X, c.g), bmy + nib(SQ1_Y, c.g));
    const int c0 = group_value<0>(cost), c1 = group_value<1>(cost), c2 = group_value<2>(cost),
              c3 = group_value<3>(cost), c4 = group_value<4>(cost), c5 = group_value<5>(cost),
              c6 = group_value<6>(cost), c7 = group_value<7>(cost);
    const bool up = y_ok<P>(c, bmy - 1), dn = y_ok<P>(c, bmy + 1);
    int dir = 0;
    if (up && c0 < bcost) { bcost = c0; dir = 1; }
    if (dn && c1 < bcost) { bcost = c1; dir = 2; }
    if (c2 < bcost) { bcost = c2; dir = 3; }
    if (c3 < bcost) { bcost = c3; dir = 4; }
    if (up && c4 < bcost) { bcost = c4; dir = 5; }
    if (dn && c5 < bcost) { bcost = c5; dir = 6; }
    if (up && c6 < bcost) { bcost = c6; dir = 7; }
    if (dn && c7 < bcost) { bcost = c7; dir = 8; }
    if (dir)
    {
        bmx += nib(SQ1_X, dir - 1);
        bmy += nib(SQ1_Y, dir - 1);
    }
}

// radius-2 hexagon walk, direction carried in the low 3 bits of the cost:
// motion.cpp:1041-1118, followed by the square refinement
template <typename P> __device__ void hex_search(const BlockCtx<P>& c, int merange, int& bmx, int& bmy, int& bcost)
{
    int bc;
    {
        const int cost = fpel_cost<P>(c, bmx + nib(HEX6_X, c.g), bmy + nib(HEX6_Y, c.g));
        const int c0 = group_value<0>(cost), c1 = group_value<1>(cost), c2 = group_value<2>(cost),
                  c3 = group_value<3>(cost), c4 = group_value<4>(cost), c5 = group_value<5>(cost);
        bc = bcost << 3;
        const bool y0 = y_ok<P>(c, bmy), yp = y_ok<P>(c, bmy + 2), ym = y_ok<P>(c, bmy - 2);
        if (y0) bc = min(bc, (c0 << 3) + 2);
        if (yp) { bc = min(bc, (c1 << 3) + 3); bc = min(bc, (c2 << 3) + 4); }
        if (y0) bc = min(bc, (c3 << 3) + 5);
        if (ym) { bc = min(bc, (c4 << 3) + 6); bc = min(bc, (c5 << 3) + 7); }
    }
    if (bc & 7)
    {
        int dir = (bc & 7) - 2;
        if (y_ok<P>(c, bmy + nib(HEX2_Y, dir + 1)))
        {
            bmx += nib(HEX2_X, dir + 1);
            bmy += nib(HEX2_Y, dir + 1);
            // half hexagons: three new points per step, one per group 0..2
            for (int i = (merange >> 1) - 1; i > 0 && in_range<P>(c, bmx, bmy); i--)
            {
                const int idx = dir + (c.g < 3 ? c.g : 0);
                const int cost = fpel_cost<P>(c, bmx + nib(HEX2_X, idx), bmy + nib(HEX2_Y, idx));
                const int c0 = group_value<0>(cost), c1 = group_value<1>(cost), c2 = group_value<2>(cost);
                bc &= ~7;
                if (y_ok<P>(c, bmy + nib(HEX2_Y, dir + 0))) bc = min(bc, (c0 << 3) + 1);
                if (y_ok<P>(c, bmy + nib(HEX2_Y, dir + 1))) bc = min(bc, (c1 << 3) + 2);
                if (y_ok<P>(c, bmy + nib(HEX2_Y, dir + 2))) bc = min(bc, (c2 << 3) + 3);
                if (!(bc & 7))
                    break;
                dir += (bc & 7) - 2;
                dir = (int)((MOD6M1 >> (4 * (dir + 1))) & 15u);
                bmx += nib(HEX2_X, dir + 1);
                bmy += nib(HEX2_Y, dir + 1);
            }
        }
    }
    bcost = bc >> 3;
    square_refine<P>(c, bmx, bmy, bcost);
}

// radius-1 diamond, step packed in the low 4 bits of the cost: motion.cpp:1016-1039
template <typename P> __device__ void dia_search(const BlockCtx<P>& c, int merange, int& bmx, int& bmy, int& bcost)
{
    int bc = bcost << 4;
    int i = merange;
    do
    {
        // points (0,-1) (0,1) (-1,0) (1,0) = square1[1..4]
        const int gi = c.g & 3;
        const int cost = fpel_cost<P>(c, bmx + nib(SQ1_X, gi), bmy + nib(SQ1_Y, gi));
        const int c0 = group_value<0>(cost), c1 = group_value<1>(cost), c2 = group_value<2>(cost),
                  c3 = group_value<3>(cost);
        if (y_ok<P>(c, bmy - 1)) bc = min(bc, (c0 << 4) + 1);
        if (y_ok<P>(c, bmy + 1)) bc = min(bc, (c1 << 4) + 3);
        bc = min(bc, (c2 << 4) + 4);
        bc = min(bc, (c3 << 4) + 12);
        if (!(bc & 15))
            break;
        bmx -= (int)((uint32_t)bc << 28) >> 30;
        bmy -= (int)((uint32_t)bc << 30) >> 30;
        bc &= ~15;
    }
    while (--i && in_range<P>(c, bmx, bmy));
    bcost = bc >> 4;
}

// MotionEstimate::motionEstimate for a lowres reference (isLowres, no extra
// candidates, subpelRefine 1).  Returns the cost; the quarter-pel vector comes
// back in (outx, outy).  Everything returned is wave-uniform.
template <typename P>
__device__ int motion_estimate_lowres(BlockCtx<P>& c, int mvpx, int mvpy, int method, int merange,
                                      const P* zeroRef, int zeroStride, int& outx, int& outy)
{
    c.mvpx = mvpx;
    c.mvpy = mvpy;
    const int qminx = c.minx * 4, qminy = c.miny * 4, qmaxx = c.maxx * 4, qmaxy = c.maxy * 4;
    const int pmvx = clampi(mvpx, qminx, qmaxx), pmvy = clampi(mvpy, qminy, qmaxy);
    int bmx = (pmvx + 2) >> 2, bmy = (pmvy + 2) >> 2;
    int bcost, bprecost;
    {
        // group 0: clipped quarter-pel predictor (SAD only, no MV cost);
        // group 1: predictor rounded to full pel; group 2: zero vector
        const int qx = (c.g == 0) ? pmvx : (c.g == 1) ? bmx * 4 : 0;
        const int qy = (c.g == 0) ? pmvy : (c.g == 1) ? bmy * 4 : 0;
        const bool odd = ((pmvx | pmvy) & 1) != 0;
        int cost = cand_sad<P>(c, qx, qy, odd);
        if (c.g != 0)
            cost += mvcost_q<P>(c, qx, qy);
        const int c0 = group_value<0>(cost), c1 = group_value<1>(cost), c2 = group_value<2>(cost);
        bprecost = c0;
        bcost = ((pmvx | pmvy) & 3) ? c1 : c0;
        if ((pmvx | pmvy) && c2 < bcost)
        {
            bcost = c2;
            bmx = 0;
            bmy = max(min(0, c.maxy), c.miny);
        }
    }
    if (bcost == 0)
    {
        outx = bmx * 4;
        outy = bmy * 4;
        return mvcost_q<P>(c, bmx * 4, bmy * 4);
    }

    if (method == 0)
        dia_search<P>(c, merange, bmx, bmy, bcost);
    else
        hex_search<P>(c, merange, bmx, bmy, bcost);

    int bqx, bqy;
    if (bprecost < bcost)
    {
        bqx = pmvx;
        bqy = pmvy;
        bcost = bprecost;
    }
    else
    {
        bqx = bmx * 4;
        bqy = bmy * 4;
    }

    if (!bcost)
        bcost = mvcost_q<P>(c, bqx, bqy);
    else
    {
        // workload[1]: four half-pel SADs, SATD at the winner, four quarter-pel
        // SATDs (motion.cpp:48-58, 1667-1699)
        const int gi = c.g & 3;
        {
            const int qx = bqx + 2 * nib(SQ1_X, gi), qy = bqy + 2 * nib(SQ1_Y, gi);
            const bool odd = ((bqx | bqy) & 1) != 0;
            const int cost = cand_sad<P>(c, qx, qy, odd) + mvcost_q<P>(c, qx, qy);
            const int c0 = group_value<0>(cost), c1 = group_value<1>(cost), c2 = group_value<2>(cost),
                      c3 = group_value<3>(cost);
            const bool up = (bqy - 2 >= qminy) && (bqy - 2 <= qmaxy), dn = (bqy + 2 >= qminy) && (bqy + 2 <= qmaxy);
            const bool mid = (bqy >= qminy) && (bqy <= qmaxy);
            int bdir = 0;
            if (up && c0 < bcost) { bcost = c0; bdir = 1; }
            if (dn && c1 < bcost) { bcost = c1; bdir = 2; }
            if (mid && c2 < bcost) { bcost = c2; bdir = 3; }
            if (mid && c3 < bcost) { bcost = c3; bdir = 4; }
            if (bdir)
            {
                bqx += 2 * nib(SQ1_X, bdir - 1);
                bqy += 2 * nib(SQ1_Y, bdir - 1);
            }
        }
        {
            const bool odd = ((bqx | bqy) & 1) != 0;
            const int cost = cand_satd<P>(c, bqx, bqy, odd) + mvcost_q<P>(c, bqx, bqy);
            bcost = group_value<0>(cost);
        }
        {
            const int qx = bqx + nib(SQ1_X, gi), qy = bqy + nib(SQ1_Y, gi);
            const int cost = cand_satd<P>(c, qx, qy, true) + mvcost_q<P>(c, qx, qy);
            const int c0 = group_value<0>(cost), c1 = group_value<1>(cost), c2 = group_value<2>(cost),
                      c3 = group_value<3>(cost);
            const bool up = (bqy - 1 >= qminy) && (bqy - 1 <= qmaxy), dn = (bqy + 1 >= qminy) && (bqy + 1 <= qmaxy);
            const bool mid = (bqy >= qminy) && (bqy <= qmaxy);
            int bdir = 0;
            if (up && c0 < bcost) { bcost = c0; bdir = 1; }
            if (dn && c1 < bcost) { bcost = c1; bdir = 2; }
            if (mid && c2 < bcost) { bcost = c2; bdir = 3; }
            if (mid && c3 < bcost) { bcost = c3; bdir = 4; }
            if (bdir)
            {
                bqx += nib(SQ1_X, bdir - 1);
                bqy += nib(SQ1_Y, bdir - 1);
            }
        }
    }

    // closing zero-vector check: SATD at (0,0) + mvcost(0,0) <= bcost replaces
    // the vector but not the cost (motion.cpp:1762-1768).  It addresses
    // fpelPlane[0] with the full lumaStride whatever plane set was searched.
    if (bqx | bqy)
    {
        const P* p = zeroRef + (long long)c.by * c.ps.stride + c.bx + (long long)c.r * zeroStride;
        const int cost = group_value<0>(satd8x8<P>(c.f, load_row<P>(p), c.r)) + mvcost_q<P>(c, 0, 0);
        if (cost <= bcost)
            bqx = bqy = 0;
    }
    outx = bqx;
    outy = bqy;
    return bcost;
}

template <typename P>
__device__ void search_block(const SearchJob& job, const Geometry& g, int level0, int cuX, int cuY, int lane,
                             const uint16_t* costTab, int maxMvd)
{
    const int W = level0 ? g.W4 : g.W, H = level0 ? g.H4 : g.H;
    const int stride = level0 ? g.stride4 : g.stride;
    const int width = level0 ? g.width4 : g.width, lines = level0 ? g.lines4 : g.lines;
    const int marginX = level0 ? g.marginX4 : g.marginX, marginY = level0 ? g.marginY4 : g.marginY;
    const int cuXY = cuX + cuY * W;

    BlockCtx<P> c;
    c.ps.origin = (const P*)job.refOrigin;
    c.ps.planeSize = level0 ? g.planeSize4 : g.planeSize;
    c.ps.stride = stride;
    c.ps.minX = -marginX;
    c.ps.maxX = width + marginX - 8;
    c.ps.minY = -marginY;
    c.ps.maxY = lines + marginY - 8;
    c.g = lane >> 3;
    c.r = lane & 7;
    c.bx = 8 * cuX;
    c.by = 8 * cuY;
    c.cost = costTab;
    c.maxMvd = maxMvd;
    c.mvpx = c.mvpy = 0;
    // search bounds that stay inside the extended frame (slicetype.cpp:4417-4420)
    c.minx = -cuX * 8 - 8;
    c.miny = -cuY * 8 - 8;
    c.maxx = (W - cuX - 1) * 8 + 8;
    c.maxy = (H - cuY - 1) * 8 + 8;
    c.fr = load_row<P>((const P*)job.fencPlane0 + (long long)(c.by + c.r) * stride + c.bx);
    unpack_row<P>(c.fr, c.f);

    // ---- predictor: lowest-SATD neighbour vector (slicetype.cpp:4438-4476) ----
    const bool lastRow = cuY == H - 1;
    const bool v0 = cuX < W - 1, v1 = !lastRow, v2 = !lastRow && cuX > 0, v3 = !lastRow && cuX < W - 1;
    bool v4 = false;
    int hx = 0, hy = 0;
    if (job.hmeMvs && !level0)
    {
        const int cuXY4 = (cuX / 2) + (cuY / 2) * W / 2;
        if (job.hmeMvCosts[cuXY4] > 0)
        {
            v4 = true;
            hx = job.hmeMvs[2 * cuXY4] * 2;
            hy = job.hmeMvs[2 * cuXY4 + 1] * 2;
        }
    }
    int mvpx = 0, mvpy = 0, skipCost = 0x7fffffff;
    if (v0 || v1 || v4)
    {
        const int off = (c.g == 0) ? 1 : (c.g == 1) ? W : (c.g == 2) ? W - 1 : W + 1;
        const bool mine = (c.g == 0) ? v0 : (c.g == 1) ? v1 : (c.g == 2) ? v2 : (c.g == 3) ? v3 : false;
        int cx = 0, cy = 0;
        if (mine)
        {
            cx = job.mvs[2 * (cuXY + off)];
            cy = job.mvs[2 * (cuXY + off) + 1];
        }
        if (c.g == 4)
        {
            cx = hx;
            cy = hy;
        }
        const bool odd = __any((cx | cy) & 1) != 0;
        const int cost = cand_satd<P>(c, cx, cy, odd);
        int mvpcost = COST_MAX;
#define TAKE(G, VALID)                                                                      \
        if (VALID)                                                                          \
        {                                                                                   \
            const int cg = group_value<G>(cost);                                            \
            if (cg < mvpcost)                                                               \
            {                                                                               \
                mvpcost = cg;                                                               \
                mvpx = group_value<G>(cx);                                                  \
                mvpy = group_value<G>(cy);                                                  \
            }                                                                               \
            /* tracks the latest candidate while the running best is the zero vector */    \
            if (!(mvpx | mvpy) && job.bBidir)                                               \
                skipCost = cg;                                                              \
        }
        TAKE(0, v0)
        TAKE(1, v1)
        TAKE(2, v2)
        TAKE(3, v3)
        TAKE(4, v4)
#undef TAKE
    }

    int mx, my;
    int fencCost = motion_estimate_lowres<P>(c, mvpx, mvpy, job.method, job.merange, (const P*)job.zeroRef,
                                             job.zeroStride, mx, my);
    // bidir skip shortcut (slicetype.cpp:4485-4489)
    if (skipCost < 64 && skipCost < fencCost && job.bBidir)
    {
        fencCost = skipCost;
        mx = my = 0;
    }
    if (lane == 0)
    {
        job.mvs[2 * cuXY] = mx;
        job.mvs[2 * cuXY + 1] = my;
        job.mvCosts[cuXY] = fencCost;
    }
}

// grid: (ceil(blocksOnDiagonal / 4), jobs); 4 waves per workgroup, one block each
template <typename P>
__global__ void __launch_bounds__(256) search_diag_kernel(const SearchJob* jobs, Geometry g, int level0, int t,
                                                          const uint16_t* costTab, int maxMvd)
{
    const int W = level0 ? g.W4 : g.W, H = level0 ? g.H4 : g.H;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int ylo = max(0, (t - (W - 1) + 1) >> 1), yhi = min(H - 1, t >> 1);
    const int yq = ylo + k;
    if (yq > yhi)
        return;                       // whole wave leaves together
    const int xq = t - 2 * yq;
    const SearchJob job = jobs[blockIdx.y];
    search_block<P>(job, g, level0, W - 1 - xq, H - 1 - yq, threadIdx.x & 63, costTab, maxMvd);
}

hipError_t launch_search(hipStream_t s, const Geometry& g, int bpp, const SearchJob* dJobs, int n, int level0,
                         const uint16_t* costTab, int maxMvd, int maxSlices, int* launches)
{
    (void)maxSlices;
    if (n <= 0)
        return hipSuccess;
    const int W = level0 ? g.W4 : g.W, H = level0 ? g.H4 : g.H;
    const int tmax = (W - 1) + 2 * (H - 1);
    for (int t = 0; t <= tmax; t++)
    {
        const int ylo = (t - (W - 1) + 1) >> 1 > 0 ? (t - (W - 1) + 1) >> 1 : 0;
        const int yhi = (t >> 1) < H - 1 ? (t >> 1) : H - 1;
        const int cnt = yhi - ylo + 1;
        if (cnt <= 0)
            continue;
        dim3 grid((cnt + 3) / 4, n);
        if (bpp == 1)
            hipLaunchKernelGGL(search_diag_kernel<uint8_t>, grid, dim3(256), 0, s, dJobs, g, level0, t, costTab, maxMvd);
        else
            hipLaunchKernelGGL(search_diag_kernel<uint16_t>, grid, dim3(256), 0, s, dJobs, g, level0, t, costTab, maxMvd);
        if (launches)
            (*launches)++;
    }
    return hipGetLastError();
}

} // namespace xg
