// 8x8 intra cost estimate of the half-res plane.
//
// Replaces LookaheadTLD::lowresIntraEstimate (source/encoder/slicetype.cpp:
// 706-815) and the primitives it calls for BLOCK_8x8: intra_filter, intra_pred
// DC / planar / angular (source/common/intrapred.cpp:31-204) and satd 8x8.
//
// One wave per four neighbouring 8x8 blocks.  The 33 neighbour samples of each
// block and their 1:2:1 smoothed copy sit in LDS; each 8-lane group predicts
// one mode (lane = row).  The eight first-pass modes (DC, planar, 5,10,..,30)
// of a block fill one pass, so the four blocks take four; the two refinement
// passes (best-2/best+2, then best-1/best+1) measure two modes per block and
// therefore run once for all four blocks, two groups each.  The reference's
// order and strict '<' tie-breaks are taken on wave-uniform scalars.  Block
// rows are independent workgroups; row totals are reduced in LDS, frame totals
// in a second tiny kernel, so results do not depend on scheduling.
#include "device_common.h"
#include "kernels.h"

namespace xg {

// SATD of the encode row `f` (column pairs) against the predicted row p[0..7]
__device__ __forceinline__ int satd_pred(const Cols& f, const int p[8], int r)
{
    return satd8x8_cols(as_s16x2(f.q[0]) - as_s16x2((uint32_t)p[0] | ((uint32_t)p[4] << 16)),
                        as_s16x2(f.q[1]) - as_s16x2((uint32_t)p[1] | ((uint32_t)p[5] << 16)),
                        as_s16x2(f.q[2]) - as_s16x2((uint32_t)p[2] | ((uint32_t)p[6] << 16)),
                        as_s16x2(f.q[3]) - as_s16x2((uint32_t)p[3] | ((uint32_t)p[7] << 16)), r, 0);
}

// g_intraFilterFlags[mode] & 8 (source/common/constants.cpp:561-567): for 8x8
// blocks only planar and the three diagonal modes read the smoothed samples
__device__ __forceinline__ bool uses_filtered(int mode) { return mode == 0 || mode == 2 || mode == 18 || mode == 34; }

// Row r of the 8x8 prediction for `mode` (0 planar, 1 DC, 2..34 angular).
// nb[0..32] raw neighbours, nb[40..72] smoothed; layout [0] corner,
// [1..16] top + top-right, [17..32] left + bottom-left.
//
// Angular modes (intra_pred_ang_c, source/common/intrapred.cpp:123-204): the eight lanes of
// the group first lay the mode's reference array ref[-8..16] out in `ext` (ext[j] = ref[j - 8]):
// the main side (top for vertical modes, left for horizontal ones), the corner at ref[-1], and
// for negative angles the other side projected through the inverse angle.  Every predicted
// sample is then the 1/32-pel interpolation of two neighbouring entries.  Horizontal modes are
// the vertical formula on swapped sides, transposed: lane r is column r there, so the entry
// offset changes per sample instead of per lane; one loop serves both.  `ext` is this group's
// own 32 ints of LDS.  DC / planar (WITH_FLAT) only occur in the first pass.
template <bool WITH_FLAT>
__device__ __forceinline__ void predict_row(const int* nb, int* ext, int mode, int r, int pixelMax, int p[8])
{
    const bool hor = mode < 18;
    const int angOff = hor ? 10 - mode : mode - 26;
    const int aabs = angOff < 0 ? -angOff : angOff;
    // |angle| for offsets 0..8: 0,2,5,9,13,17,21,26,32 (angleTable, intrapred.cpp:123)
    const int angMag = (int)((0x201A15110D090502ull >> (8 * ((aabs - 1) & 7))) & 0xff) & -(int)(aabs != 0);
    const int angle = angOff < 0 ? -angMag : angMag;
    // inverse angles for offsets -1..-8: 4096,1638,910,630,482,390,315,256 (invAngleTable, :124)
    const int inv = (aabs <= 4) ? ((aabs <= 2) ? (aabs <= 1 ? 4096 : 1638) : (aabs == 3 ? 910 : 630))
                                : ((aabs <= 6) ? (aabs == 5 ? 482 : 390) : (aabs == 7 ? 315 : 256));
    const bool ang = !WITH_FLAT || mode >= 2;
    const int* s = (mode == 0 || mode == 2 || mode == 18 || mode == 34) ? nb + 40 : nb;
    // main side first sample index in `s`, side first sample index
    const int mainBase = hor ? 17 : 1, sideBase = hor ? 1 : 17;
    // ---- reference array: 25 entries, lane r writes entries r, r + 8, r + 16 (and 24 by lane 0) ----
#pragma unroll
    for (int j0 = 0; j0 < 32; j0 += 8)
    {
        const int j = j0 + r, i = j - 8;                  // ref index
        if (j <= 24)
        {
            int v;
            if (i >= 0)
                v = s[mainBase + i];
            else if (i == -1)
                v = s[0];
            else
                v = s[sideBase - 1 + ((128 + __mul24(-1 - i, inv)) >> 8)];     // ref[i] = side[(i * invAngle + 128) >> 8] mirrored
            ext[j] = v;
        }
    }
    lds_handover();
    if (ang)
    {
#pragma unroll
        for (int x = 0; x < 8; x++)
        {
            // (yy, xx): position in the un-transposed prediction
            const int yy = hor ? x : r, xx = hor ? r : x;
            // small operands throughout: 24-bit multiplies issue at full rate, 32-bit ones at a quarter of it
            const int angSum = __mul24(yy + 1, angle);
            const int idx = (angSum >> 5) + xx + 8, frac = angSum & 31;
            const int a = ext[idx], b = ext[idx + 1];
            p[x] = (__mul24(frac, b - a) + (a << 5) + 16) >> 5;      // = ((32 - frac) * a + frac * b + 16) >> 5
        }
        if (angle == 0)
        {
            // pure vertical / horizontal with the edge filter (bFilter, size <= 16): :170-180
            // un-transposed column 0 = corner-relative gradient along the side
            if (hor)
            {
                if (r == 0)
                {
#pragma unroll
                    for (int x = 0; x < 8; x++)
                        p[x] = clampi((int)(short)(ext[8] + ((s[sideBase + x] - s[0]) >> 1)), 0, pixelMax);
                }
            }
            else
                p[0] = clampi((int)(short)(ext[8] + ((s[sideBase + r] - s[0]) >> 1)), 0, pixelMax);
        }
    }
    if (WITH_FLAT && mode == 1)
    {
        // intra_pred_dc_c with the edge filter on (bFilter = size <= 16)
        const int* raw = nb;
        int dc = 8;
#pragma unroll
        for (int i = 0; i < 8; i++)
            dc += raw[1 + i] + raw[17 + i];
        dc >>= 4;
#pragma unroll
        for (int x = 0; x < 8; x++)
            p[x] = (r == 0) ? ((raw[1 + x] + 3 * dc + 2) >> 2) : dc;
        p[0] = (r == 0) ? ((raw[1] + raw[17] + 2 * dc + 2) >> 2) : ((raw[17 + r] + 3 * dc + 2) >> 2);
    }
    if (WITH_FLAT && mode == 0)
    {
        const int tr = s[9], bl = s[25], lf = s[17 + r];
#pragma unroll
        for (int x = 0; x < 8; x++)
            p[x] = (__mul24(7 - x, lf) + __mul24(7 - r, s[1 + x]) + __mul24(x + 1, tr) + __mul24(r + 1, bl) + 8) >> 4;
    }
    lds_handover();     // `ext` is rewritten by the next call
}

template <typename P>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6))) intra_kernel(const IntraJob* jobs, Geometry g, int lambda, int pixelMax)
{
    __shared__ int nbr[4][4][80];          // [wave][block][raw 0..32, smoothed 40..72]
    // [wave][group]: the angular reference array being used (25 entries).  Groups 40 dwords apart: the eight lanes of
    // a group read within eight consecutive entries, and the LDS serves 32 lanes (four groups) a cycle from 32 banks,
    // so a spacing of 8 banks between groups keeps them out of each other's way (32 apart, all groups shared banks)
    __shared__ int extRef[4][8][40];
    __shared__ long long red[4][3];

    const IntraJob job = jobs[blockIdx.y];
    const int cuY = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = lane >> 3, r = lane & 7;
    const int W = g.W, H = g.H;
    const P* plane = (const P*)job.plane0;
    const int by = 8 * cuY;
    int (*nb)[80] = nbr[wave];
    int* ext = extRef[wave][grp];

    long long accRow = 0, accEst = 0, accAq = 0;
    const int iters = (W + 15) / 16;
    for (int it = 0; it < iters; it++)
    {
        const int cuX0 = it * 16 + wave * 4;       // this wave's four blocks: cuX0 .. cuX0 + 3 (clamped to the row)
        // ---- neighbours: 16 lanes per block, lane k takes top sample k+1 and left sample k+1 ----
        {
            const int b = lane >> 4, k = lane & 15;
            const P* tl = plane + (long long)(by - 1) * g.stride + 8 * min(cuX0 + b, W - 1) - 1;
            nb[b][1 + k] = (int)tl[1 + k];
            nb[b][17 + k] = (int)tl[(long long)(1 + k) * g.stride];
            if (k == 0)
                nb[b][0] = (int)tl[0];
        }
        lds_handover();
        // intraFilter<8>: source/common/intrapred.cpp:31-51, 33 samples for each of the four blocks
        for (int i = lane; i < 132; i += 64)
        {
            const int b = i / 33, j = i - 33 * b;
            const int* n = nb[b];
            int f;
            if (j == 0)
                f = (2 * n[0] + n[1] + n[17] + 2) >> 2;
            else if (j == 16 || j == 32)
                f = n[j];
            else if (j == 17)
                f = (2 * n[17] + n[0] + n[18] + 2) >> 2;
            else
                f = (2 * n[j] + n[j - 1] + n[j + 1] + 2) >> 2;
            nb[b][40 + j] = f;
        }
        lds_handover();

        int p[8];
        int icost[4], ilow[4], acost[4], alow[4];
        // pass 1, once per block: DC, planar, 5, 10, ..., 30
        const int mode1 = (grp == 0) ? 1 : (grp == 1) ? 0 : 5 * (grp - 1);
#define XG_INTRA_PASS1(B)                                                                                      \
        {                                                                                                      \
            const Cols f = pair_cols<P>(load_row<P>(plane + (long long)(by + r) * g.stride + 8 * min(cuX0 + B, W - 1))); \
            predict_row<true>(nb[B], ext, mode1, r, pixelMax, p);                                                         \
            const int c = satd_pred(f, p, r);                                                                  \
            int ic = COST_MAX, il = 0, ac = COST_MAX, al = 4;                                                  \
            const int cDC = group_value<0>(c), cPl = group_value<1>(c);                                        \
            if (cDC < ic) { ic = cDC; il = 1; }                                                                \
            if (cPl < ic) { ic = cPl; il = 0; }                                                                \
            const int c5 = group_value<2>(c), c10 = group_value<3>(c), c15 = group_value<4>(c);                \
            const int c20 = group_value<5>(c), c25 = group_value<6>(c), c30 = group_value<7>(c);               \
            if (c5 < ac) { ac = c5; al = 5; }                                                                  \
            if (c10 < ac) { ac = c10; al = 10; }                                                               \
            if (c15 < ac) { ac = c15; al = 15; }                                                               \
            if (c20 < ac) { ac = c20; al = 20; }                                                               \
            if (c25 < ac) { ac = c25; al = 25; }                                                               \
            if (c30 < ac) { ac = c30; al = 30; }                                                               \
            icost[B] = ic; ilow[B] = il; acost[B] = ac; alow[B] = al;                                          \
        }
        XG_INTRA_PASS1(0)
        XG_INTRA_PASS1(1)
        XG_INTRA_PASS1(2)
        XG_INTRA_PASS1(3)
#undef XG_INTRA_PASS1

        // passes 2 and 3 for all four blocks at once: groups 2b and 2b+1 measure best-dist and
        // best+dist of block b, the best being updated in between as in the reference
        const int b2 = grp >> 1;
        const Cols f2 = pair_cols<P>(load_row<P>(plane + (long long)(by + r) * g.stride + 8 * min(cuX0 + b2, W - 1)));
#pragma unroll
        for (int dist = 2; dist >= 1; dist--)
        {
            const int best = (b2 == 0) ? alow[0] : (b2 == 1) ? alow[1] : (b2 == 2) ? alow[2] : alow[3];
            const int mode = (grp & 1) ? best + dist : best - dist;
            predict_row<false>(nb[b2], ext, mode, r, pixelMax, p);
            const int c = satd_pred(f2, p, r);
#define XG_INTRA_REFINE(B)                                                                 \
            {                                                                              \
                const int cm = group_value<2 * B>(c), cp = group_value<2 * B + 1>(c);      \
                const int minus = alow[B] - dist, plus = alow[B] + dist;                   \
                if (cm < acost[B]) { acost[B] = cm; alow[B] = minus; }                     \
                if (cp < acost[B]) { acost[B] = cp; alow[B] = plus; }                      \
            }
            XG_INTRA_REFINE(0)
            XG_INTRA_REFINE(1)
            XG_INTRA_REFINE(2)
            XG_INTRA_REFINE(3)
#undef XG_INTRA_REFINE
        }

#pragma unroll
        for (int b = 0; b < 4; b++)
        {
            const int cuX = cuX0 + b;
            if (cuX >= W)          // wave-uniform
                break;
            int ic = icost[b], il = ilow[b];
            if (acost[b] < ic) { ic = acost[b]; il = alow[b]; }
            ic += 5 * lambda + 4;   // intraPenalty + lowresPenalty
            const int cuXY = cuX + cuY * W;
            const bool scored = (cuX > 0 && cuX < W - 1 && cuY > 0 && cuY < H - 1) || W <= 2 || H <= 2;
            int icAq = ic;
            if (scored && job.invQscale)
                icAq = (ic * job.invQscale[cuXY] + 128) >> 8;
            if (lane == 0)
            {
                job.lowresCosts[cuXY] = (uint16_t)min(ic, LOWRES_COST_MASK);
                job.intraCost[cuXY] = ic;
                job.intraMode[cuXY] = (uint8_t)il;
            }
            accRow += icAq;
            if (scored)
            {
                accEst += ic;
                accAq += icAq;
            }
        }
        lds_handover();   // the neighbour buffer is rewritten next iteration
    }

    if (lane == 0)
    {
        red[wave][0] = accRow;
        red[wave][1] = accEst;
        red[wave][2] = accAq;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        job.rowSatds[cuY] = (int32_t)(red[0][0] + red[1][0] + red[2][0] + red[3][0]);
        job.rowPartials[2 * cuY + 0] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
        job.rowPartials[2 * cuY + 1] = red[0][2] + red[1][2] + red[2][2] + red[3][2];
    }
}

// frame totals: costEst[0][0], costEstAq[0][0] (slicetype.cpp:813-814)
__global__ void intra_finalize_kernel(const IntraJob* jobs, int H)
{
    const IntraJob job = jobs[blockIdx.x];
    long long a = 0, b = 0;
    for (int y = threadIdx.x; y < H; y += 64)
    {
        a += job.rowPartials[2 * y];
        b += job.rowPartials[2 * y + 1];
    }
    for (int o = 32; o > 0; o >>= 1)
    {
        a += __shfl_down(a, o);
        b += __shfl_down(b, o);
    }
    if (threadIdx.x == 0)
    {
        job.sums[0] = a;
        job.sums[1] = b;
    }
}

// The AQ-weighted sums of an intra estimate whose costs are in place, redone with the scale factors the job points
// at now: the row totals (all blocks) and the frame partial (scored blocks) of slicetype.cpp:798-811, then the frame
// totals.  One 64-lane workgroup per block row.  Lets the estimate itself run before the factors exist.
__global__ void __launch_bounds__(64) intra_aq_rows_kernel(const IntraJob* jobs, Geometry g)
{
    const IntraJob job = jobs[blockIdx.y];
    const int W = g.W, H = g.H, cuY = blockIdx.x;
    long long accRow = 0, accAq = 0;
    for (int cuX = threadIdx.x; cuX < W; cuX += 64)
    {
        const int cuXY = cuX + cuY * W;
        const int ic = job.intraCost[cuXY];
        const bool scored = (cuX > 0 && cuX < W - 1 && cuY > 0 && cuY < H - 1) || W <= 2 || H <= 2;
        int icAq = ic;
        if (scored && job.invQscale)
            icAq = (ic * job.invQscale[cuXY] + 128) >> 8;
        accRow += icAq;
        if (scored)
            accAq += icAq;
    }
    for (int o = 32; o > 0; o >>= 1)
    {
        accRow += __shfl_down(accRow, o);
        accAq += __shfl_down(accAq, o);
    }
    if (threadIdx.x == 0)
    {
        job.rowSatds[cuY] = (int32_t)accRow;
        job.rowPartials[2 * cuY + 1] = accAq;
    }
}

hipError_t launch_intra_aq_sums(hipStream_t s, const Geometry& g, const IntraJob* dJobs, int n)
{
    if (n <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(intra_aq_rows_kernel, dim3(g.H, n), dim3(64), 0, s, dJobs, g);
    hipLaunchKernelGGL(intra_finalize_kernel, dim3(n), dim3(64), 0, s, dJobs, g.H);
    return hipGetLastError();
}

hipError_t launch_intra(hipStream_t s, const Geometry& g, int bpp, int depth, int lambda, const IntraJob* dJobs, int n)
{
    if (n <= 0)
        return hipSuccess;
    dim3 grid(g.H, n);
    const int pixelMax = (1 << depth) - 1;
    if (bpp == 1)
        hipLaunchKernelGGL(intra_kernel<uint8_t>, grid, dim3(256), 0, s, dJobs, g, lambda, pixelMax);
    else
        hipLaunchKernelGGL(intra_kernel<uint16_t>, grid, dim3(256), 0, s, dJobs, g, lambda, pixelMax);
    hipLaunchKernelGGL(intra_finalize_kernel, dim3(n), dim3(64), 0, s, dJobs, g.H);
    return hipGetLastError();
}

} // namespace xg
